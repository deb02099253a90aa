"""CPU ORACLE (numpy) for navigation1 / navigation2 -- TEST INFRASTRUCTURE ONLY.

A sequential restatement of `env/navigation1.py` / `env/navigation2.py` that consumes
the global `np.random` stream in the same order as the reference, so that with the same
seed it reproduces the reference's outputs exactly (pinned by
tests/golden/nav_step_golden.npz and nav_offline_golden.npz, which were captured by
importing the reference; generator: tests/golden/gen_env_golden.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from fractions import Fraction

import numpy as np

# env/navigation1.py:41-42 and env/navigation2.py:41 -- [[x0,x1],[y0,y1]] closed boxes
BOXES = {
    "navigation1": (((-100.0, 150.0), (5.0, 10.0)), ((-100.0, -80.0), (-10.0, 10.0)),
                    ((-100.0, 150.0), (-10.0, -5.0))),
    "navigation2": (((-30.0, -20.0), (-7.5, 7.5)),),
}
START = np.array([-50.0, 0.0])   # navigation1.py:27
NOISE_SCALE = 0.05               # navigation1.py:36
HORIZON = 100                    # navigation1.py:34


def obstacle(env_name, s):
    """env/obstacle.py:13-15 + :44-45 -- closed-interval box membership, max over boxes."""
    hit = False
    for (x0, x1), (y0, y1) in BOXES[env_name]:
        hit = hit or (x0 <= s[0] <= x1 and y0 <= s[1] <= y1)
    return int(hit)


def next_state(env_name, s, a, randn=None):
    """navigation1.py:99-104: stuck (and NO noise drawn) inside an obstacle."""
    if obstacle(env_name, s):
        return s
    eps = (np.random.randn if randn is None else randn)(2)
    return (s + a) + NOISE_SCALE * eps


def _fma(a, b, c):
    """Correctly rounded a*b+c (exact rational arithmetic, one rounding)."""
    return float(Fraction(float(a)) * Fraction(float(b)) + Fraction(float(c)))


def step_cost(s):
    """navigation1.py:106-110 (HARD_MODE False): -||s - goal||, goal = origin.
    np.linalg.norm = sqrt(ddot(s, s)); OpenBLAS' ddot accumulates with FMA, i.e.
    sqrt(fma(y, y, x*x)) -- this form matches every golden reward bit-for-bit."""
    return -np.sqrt(_fma(s[1], s[1], s[0] * s[0]))


class NavOracleEnv:
    """Sequential 1-env oracle with the reference's step/reset protocol
    (navigation1.py:71-97)."""

    def __init__(self, env_name):
        self.env_name = env_name
        self.state = None
        self.time = 0
        self._max_episode_steps = HORIZON

    def reset(self):
        self.state = START + np.random.randn(2)
        self.time = 0
        return self.state

    def step(self, a):
        a = np.clip(a, -1, 1)                      # process_action :50-51
        old = self.state.copy()
        nxt = next_state(self.env_name, self.state, a)
        cost = step_cost(old)
        self.state = nxt
        self.time += 1
        cons = obstacle(self.env_name, nxt)
        done = bool(cost > -4 or cons)             # :80
        info = {"constraint": cons, "reward": cost, "state": old, "next_state": nxt,
                "action": a, "success": bool(cost > -4)}
        return nxt, cost, done, info


def _rollout(env_name, state, action_fn, out):
    """<=10 scripted steps, break on constraint (navigation1.py:148-159)."""
    for _ in range(10):
        action = action_fn()
        nxt = next_state(env_name, state, action)
        cons = obstacle(env_name, nxt)
        out.append((state, action, cons, nxt, not cons))
        state = nxt
        if cons:
            break


def get_offline_data(env_name, num_transitions):
    """navigation1.py:133-164 / navigation2.py:133-243, same np.random draw order."""
    U, N = np.random.uniform, np.random.randn
    out = []

    def rand_action():
        return np.clip(N(2), -1, 1)

    if env_name == "navigation1":
        for _ in range(num_transitions // 10):
            if U(0, 1) < 0.5:
                state = np.array([U(-80, 50), U(-5, -2)])
            else:
                state = np.array([U(-80, 50), U(2, 5)])
            _rollout(env_name, state, rand_action, out)
        return out
    for _ in range(num_transitions // 10 // 3):
        state = np.array([U(-40, 10), U(-25, 25)])
        while obstacle(env_name, state):
            state = np.array([U(-40, 10), U(-25, 25)])
        _rollout(env_name, state, rand_action, out)
    n4 = num_transitions // 10 * 1 // 4
    phases = (  # (x range, y range, action generator)  navigation2.py:160-239
        ((-35, -30), (-12, 12), lambda: np.clip(np.array([U(0.5, 1, 1), N(1)]), -1, 1).ravel()),
        ((-20, -15), (-12, 12), lambda: np.clip(np.array([U(-1, -0.5, 1), N(1)]), -1, 1).ravel()),
        ((-30, -20), (10, 15), lambda: np.clip(np.array([N(1), U(-1, -0.5, 1)]), -1, 1).ravel()),
        ((-30, -20), (-15, -10), lambda: np.clip(np.array([N(1), U(0.5, 1, 1)]), -1, 1).ravel()),
    )
    for xr, yr, act in phases:
        for _ in range(n4):
            state = np.array([U(*xr), U(*yr)])
            _rollout(env_name, state, act, out)
    return out
