"""Generate golden vectors G1 (env step) and G2 (offline data) by IMPORTING the
reference (env/navigation1.py, env/navigation2.py) in this container.

Run:  python tests/golden/gen_env_golden.py
Writes tests/golden/nav_step_golden.npz and tests/golden/nav_offline_golden.npz.
The outputs are data only (inputs + the reference's outputs); no reference
source is stored.  SURVEY.md section 8c rows G1/G2.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_shims  # noqa: E402

_ref_shims.install()

import io  # noqa: E402
import contextlib  # noqa: E402
import importlib  # noqa: E402

BOXES = {
    "navigation1": [((-100, 150), (5, 10)), ((-100, -80), (-10, 10)),
                    ((-100, 150), (-10, -5))],
    "navigation2": [((-30, -20), (-7.5, 7.5))],
}


def _ulps(x, k):
    x = np.float64(x)
    for _ in range(abs(k)):
        x = np.nextafter(x, np.inf if k > 0 else -np.inf)
    return x


def build_rows(env_name, rng):
    """Return (s, a_raw, eps) test rows: random + adversarial edge rows."""
    S, A, E = [], [], []

    def add(s, a, e):
        S.append(np.asarray(s, dtype=np.float64))
        A.append(np.asarray(a, dtype=np.float32).astype(np.float64))
        E.append(np.asarray(e, dtype=np.float64))

    # (1) random rows near the start state and along typical trajectories
    for _ in range(600):
        add([-50, 0] + rng.randn(2) * [20, 3], rng.randn(2) * 1.5, rng.randn(2))
    # (2) random rows over the whole arena, including inside obstacles
    lo, hi = ((-110, -15), (160, 15)) if env_name == "navigation1" else ((-45, -30), (15, 30))
    for _ in range(800):
        add(rng.uniform(lo, hi), rng.uniform(-2, 2, 2), rng.randn(2))
    # (3) rows around the goal (|s| straddling 4, the success/termination radius)
    for _ in range(300):
        th = rng.uniform(0, 2 * np.pi)
        r = 4.0 * (1 + rng.choice([0, 1e-16, -1e-16, 1e-12, -1e-12, 1e-9, -1e-9, 1e-3, -1e-3]))
        add([r * np.cos(th), r * np.sin(th)], rng.uniform(-1, 1, 2), rng.randn(2))
    for sx, sy in [(4, 0), (0, 4), (-4, 0), (0, -4), (2.4, 3.2), (3.2, -2.4),
                   (np.sqrt(8), np.sqrt(8)), (_ulps(4, 1), 0), (_ulps(4, -1), 0),
                   (0, _ulps(4, 1)), (0, _ulps(-4, 1)), (0, 0), (1e-300, 0)]:
        add([sx, sy], [0.25, -0.5], [0.3, -0.7])
    # (4) next_state landing exactly on / 1-2 ulp / 1e-12 / 1e-9 around every box edge
    for (x0, x1), (y0, y1) in BOXES[env_name]:
        xm = 0.5 * (max(x0, -60) + min(x1, 20))
        ym = 0.5 * (y0 + y1)
        for k in (0, 1, -1, 2, -2):
            for d in (0.0, 1e-12, -1e-12, 1e-9, -1e-9):
                for edge_y in (y0, y1):
                    t = _ulps(edge_y, k) + d
                    add([xm, t], [0, 0], [0, 0])            # s itself on the edge (no-noise branch or exact copy)
                    add([xm, t - 0.5], [0, 0.5], [0, 0])    # step onto the edge from outside/inside
                    add([xm, t + 0.5], [0.25, -0.5], [0, 0])
                for edge_x in (x0, x1):
                    t = _ulps(edge_x, k) + d
                    add([t, ym], [0, 0], [0, 0])
                    add([t - 1.0, ym], [3.0, 0], [0, 0])    # clipped to +1 -> lands on the edge
                    add([t + 1.0, ym], [-7.0, 0.125], [0, 0])
        # corners
        for cx in (x0, x1):
            for cy in (y0, y1):
                for kx in (0, 1, -1):
                    for ky in (0, 1, -1):
                        add([_ulps(cx, kx), _ulps(cy, ky)], [0, 0], [0, 0])
                        add([_ulps(cx, kx) - 0.5, _ulps(cy, ky) + 0.25], [0.5, -0.25], [0, 0])
    # (5) noise carries the state across an edge
    for _ in range(300):
        (x0, x1), (y0, y1) = BOXES[env_name][rng.randint(len(BOXES[env_name]))]
        edge = rng.choice([y0, y1])
        x = rng.uniform(max(x0, -60), min(x1, 20))
        add([x, edge + rng.uniform(-1.2, 1.2)], rng.uniform(-1.5, 1.5, 2), rng.randn(2) * 3)
    return np.stack(S), np.stack(A), np.stack(E)


def run_reference_steps(env_name, S, A, E):
    mod = importlib.import_module("env." + env_name)
    cls = getattr(mod, {"navigation1": "Navigation1", "navigation2": "Navigation2"}[env_name])
    env = cls()
    M = len(S)
    out = dict(s2=np.zeros((M, 2)), reward=np.zeros(M), done=np.zeros(M, np.uint8),
               constraint=np.zeros(M, np.uint8), success=np.zeros(M, np.uint8),
               a_clip=np.zeros((M, 2)), noise_drawn=np.zeros(M, np.uint8))
    real_randn = np.random.randn
    sink = io.StringIO()
    try:
        for i in range(M):
            drawn = []

            def fake_randn(n, _i=i, _d=drawn):
                _d.append(1)
                return E[_i].copy()

            np.random.randn = fake_randn
            env.reset()
            drawn.clear()
            env.state = S[i].copy()
            env.time = 0
            with contextlib.redirect_stdout(sink):
                obs, r, done, info = env.step(A[i].astype(np.float32))
            out["s2"][i] = obs
            out["reward"][i] = r
            out["done"][i] = bool(done)
            out["constraint"][i] = bool(info["constraint"])
            out["success"][i] = bool(info["success"])
            out["a_clip"][i] = info["action"]
            out["noise_drawn"][i] = len(drawn)
            assert np.array_equal(info["state"], S[i]) and np.array_equal(info["next_state"], obs)
            assert info["reward"] == r
    finally:
        np.random.randn = real_randn
    return out


def run_reference_offline(env_name, seed, num):
    mod = importlib.import_module("env." + env_name)
    np.random.seed(seed)
    sink = io.StringIO()
    with contextlib.redirect_stdout(sink):
        tr = mod.get_offline_data(num)
    s = np.array([t[0] for t in tr], dtype=np.float64)
    a = np.array([t[1] for t in tr], dtype=np.float64)
    c = np.array([int(t[2]) for t in tr], dtype=np.uint8)
    s2 = np.array([t[3] for t in tr], dtype=np.float64)
    m = np.array([int(bool(t[4])) for t in tr], dtype=np.uint8)
    return s, a, c, s2, m


def main():
    rng = np.random.RandomState(20260928)
    step = {}
    for env_name in ("navigation1", "navigation2"):
        S, A, E = build_rows(env_name, rng)
        out = run_reference_steps(env_name, S, A, E)
        step[env_name + "_s"] = S
        step[env_name + "_a"] = A.astype(np.float32)
        step[env_name + "_eps"] = E
        for k, v in out.items():
            step[env_name + "_" + k] = v
        print(env_name, "rows", len(S), "done", int(out["done"].sum()), "constraint",
              int(out["constraint"].sum()), "success", int(out["success"].sum()),
              "no-noise rows", int((out["noise_drawn"] == 0).sum()))
    # anchor from SURVEY section 8c (nav1, seed 0)
    import env.navigation1 as n1
    np.random.seed(0)
    e = n1.Navigation1()
    s0 = e.reset()
    o1, r1, d1, _ = e.step(np.array([1, 0]))
    o2, r2, d2, _ = e.step(np.array([2, -3]))
    step["anchor"] = np.array([s0[0], s0[1], o1[0], o1[1], r1, o2[0], o2[1], r2])
    np.savez_compressed(os.path.join(HERE, "nav_step_golden.npz"), **step)

    off = {}
    for env_name in ("navigation1", "navigation2"):
        for seed in (0, 1):
            s, a, c, s2, m = run_reference_offline(env_name, seed, 1000)
            key = "%s_seed%d_n1000_" % (env_name, seed)
            off[key + "s"], off[key + "a"], off[key + "c"], off[key + "s2"], off[key + "m"] = s, a, c, s2, m
        s, a, c, s2, m = run_reference_offline(env_name, 1, 20000)
        off[env_name + "_seed1_n20000_stats"] = np.array(
            [len(s), int(c.sum()), s.mean(0)[0], s.mean(0)[1], s.var(0)[0], s.var(0)[1],
             a.mean(0)[0], a.mean(0)[1], a.var(0)[0], a.var(0)[1]])
        print(env_name, "offline 20000 @seed1:", len(s), "transitions,", int(c.sum()), "violations")
    np.savez_compressed(os.path.join(HERE, "nav_offline_golden.npz"), **off)


if __name__ == "__main__":
    main()
