"""Golden vectors for the Maze env captured from the REFERENCE's env/maze.py, imported in this container over the
stand-in MjSim of maze_stub_sim.py (MuJoCo is absent; the stand-in implements the documented kinematic surrogate,
DESIGN.md section 6).  What this pins: every line of env/maze.py around `sim.step()` -- step (:139-168), reset incl.
the wall moves and the contact re-draw (:184-213), get_distance_score (:215-220), expert_action (:222-232) and
get_offline_data (:34-107).  What it cannot pin: MuJoCo's trajectories.

Run: python tests/golden/gen_maze_ref_golden.py -> tests/golden/maze_ref_golden.npz (data only).
np.random.uniform is replaced from outside by `lo + (hi - lo) * u` over a recorded uniform stream (numpy's own
formula), and env.action_space.sample by a recorded float32 stream, so that the oracle can be fed the same draws.
"""
import contextlib
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_shims  # noqa: E402
import maze_stub_sim  # noqa: E402

_ref_shims.install()
sys.modules["mujoco_py"].load_model_from_path = maze_stub_sim.load_model_from_path
sys.modules["mujoco_py"].MjSim = maze_stub_sim.MjSim

MODES = {0: 'h', 1: 'e', 2: 'm', 3: None}


class UniformStream:
    """Replacement for np.random.uniform that records the underlying [0,1) draws."""

    def __init__(self, seed):
        self.rng = np.random.RandomState(seed)
        self.u = []

    def __call__(self, low=0.0, high=1.0, size=None):
        n = 1 if size is None else int(np.prod(size))
        u = self.rng.random_sample(n)
        self.u.extend(u.tolist())
        out = low + (high - low) * u                      # numpy: loc + scale * random_sample()
        return float(out[0]) if size is None else out.reshape(size)


@contextlib.contextmanager
def patched_uniform(stream):
    real = np.random.uniform
    np.random.uniform = stream
    try:
        yield stream
    finally:
        np.random.uniform = real


def main():
    import env.maze as ref
    rng = np.random.RandomState(11)
    out = {}
    env = ref.MazeNavigation()
    model = env.sim.model
    out["model.geom_names"] = np.array(model.geom_names)
    out["model.gain"] = np.array(model.free_gain())

    # ---- reset: wall placement, ranges per difficulty, contact re-draw ----
    res_mode, res_check, res_u, res_pos, res_used = [], [], [], [], []
    for k in range(600):
        mode, check = k % 4, bool((k // 4) % 4)            # mostly with the contact check
        st = UniformStream(1000 + k)
        with patched_uniform(st):
            obs = env.reset(MODES[mode], check_constraint=check)
        u = st.u + [0.5] * (40 - len(st.u))
        assert len(st.u) <= 40 and env.steps == 0
        res_mode.append(mode), res_check.append(int(check)), res_u.append(u[:40]), res_pos.append(obs.copy())
        res_used.append(len(st.u))
    out["reset.mode"], out["reset.check"] = np.array(res_mode), np.array(res_check)
    out["reset.u"], out["reset.pos"], out["reset.used"] = np.array(res_u), np.array(res_pos), np.array(res_used)
    out["model.wall_pos_after_reset"] = model.geom_pos[5:9, :2].copy()

    # ---- step: (pos, raw float64 action, steps) -> (obs, reward, done, info) ----
    n = 2500
    pos = np.c_[rng.uniform(-0.29, 0.29, n), rng.uniform(-0.29, 0.29, n)]
    edge = []                                                # rows hugging every wall face, corner and the arena planes
    for cx, cy in ((-0.1, 0.42), (0.1, 0.48), (-0.1, -0.33), (0.1, -0.17)):
        for d in (0.0249, 0.025, 0.0251, 0.03, 0.04):
            for side in (-1, 1):
                edge.append([cx + side * (0.005 + d), cy - 0.1])
                edge.append([cx + side * (0.005 + d * 0.7), cy - side * (0.2 + d * 0.7)])
            edge.append([cx, cy - 0.2 - d])
            edge.append([cx, cy + 0.2 + d])
    for d in (0.2749, 0.275, 0.2751, 0.26):
        edge += [[d, 0.0], [-d, 0.0], [0.0, d], [0.0, -d], [d, d], [-d, -d]]
    for gx in (0.25, 0.22, 0.28):                            # around the goal (dist < 0.03, success)
        for gy in (0.0, 0.02, -0.03, 0.045):
            edge.append([gx, gy])
    pos = np.vstack([pos, np.array(edge)])
    n = len(pos)
    act = rng.uniform(-0.15, 0.15, (n, 2))                   # float64, beyond the +-0.1 clip in places
    act[5::50] = 0.0
    toward = np.array([[-0.1, 0.42], [0.1, 0.48], [-0.1, -0.33], [0.1, -0.17]])
    for i in range(0, n, 3):                                 # a third of the rows head for the nearest wall
        w = toward[np.argmin(np.abs(toward[:, 0] - pos[i, 0]))]
        act[i, 0] = np.sign(w[0] - pos[i, 0]) * rng.uniform(0.02, 0.12)
    act[::2] = act[::2].astype(np.float32)                   # float32-valued rows (what the policy emits)
    steps = rng.randint(0, 100, n)
    steps[::11] = 99                                          # the horizon row (steps + 1 >= 100)
    o_next, o_rew, o_done, o_cons, o_succ, o_state, o_act = [], [], [], [], [], [], []
    for i in range(n):
        env.reset(pos=(pos[i, 0], pos[i, 1]))
        env.steps = int(steps[i])
        obs, reward, done, info = env.step(act[i].copy())
        assert np.array_equal(info["next_state"], obs) and info["reward"] == reward
        o_next.append(obs.copy()), o_rew.append(float(reward)), o_done.append(int(bool(done)))
        o_cons.append(int(info["constraint"])), o_succ.append(int(bool(info["success"])))
        o_state.append(info["state"].copy()), o_act.append(np.asarray(info["action"], dtype=np.float64).copy())
    out["step.pos"], out["step.act"], out["step.steps"] = pos, act, steps
    out["step.next"], out["step.reward"] = np.array(o_next), np.array(o_rew)
    out["step.done"], out["step.constraint"], out["step.success"] = (np.array(x, dtype=np.uint8)
                                                                      for x in (o_done, o_cons, o_succ))
    out["step.info_state"], out["step.info_action"] = np.array(o_state), np.array(o_act)

    # ---- multi-step episodes with the expert (trajectory-level: stuck-in-contact, goal reached) ----
    ep_pos, ep_act, ep_rew, ep_done, ep_cons, ep_dist = [], [], [], [], [], []
    for k in range(12):
        st = UniformStream(5000 + k)
        with patched_uniform(st):
            env.reset(MODES[k % 4])
        traj_p, traj_a, traj_r, traj_d, traj_c, traj_g = [env._get_obs().copy()], [], [], [], [], []
        for _ in range(40):
            a = env.expert_action() if k % 3 else env.expert_action() * 2.5     # some over-driven runs hit walls
            a = np.array(a, dtype=np.float64)
            obs, reward, done, info = env.step(a)
            traj_p.append(obs.copy()), traj_a.append(a), traj_r.append(float(reward)), traj_d.append(int(bool(done)))
            traj_c.append(int(info["constraint"])), traj_g.append(float(env.get_distance_score()))
        ep_pos.append(traj_p), ep_act.append(traj_a), ep_rew.append(traj_r), ep_done.append(traj_d)
        ep_cons.append(traj_c), ep_dist.append(traj_g)
    out["ep.pos"], out["ep.act"], out["ep.reward"] = np.array(ep_pos), np.array(ep_act), np.array(ep_rew)
    out["ep.done"], out["ep.constraint"], out["ep.dist"] = np.array(ep_done), np.array(ep_cons), np.array(ep_dist)

    # ---- expert_action / get_distance_score on a grid ----
    q = np.c_[rng.uniform(-0.3, 0.3, 400), rng.uniform(-0.3, 0.3, 400)]
    q[:6, 0] = [-0.151, -0.1510001, -0.1509999, 0.149, 0.1490001, 0.1489999]
    ex, ds = [], []
    for x, y in q:
        env.sim.data.qpos[0], env.sim.data.qpos[1] = x, y
        ex.append(np.array(env.expert_action(), dtype=np.float64))
        ds.append(float(env.get_distance_score()))
    out["expert.pos"], out["expert.act"], out["expert.dist"] = q, np.array(ex), np.array(ds)

    # ---- get_offline_data (env/maze.py:34-107) with both random streams recorded ----
    for num in (1000, 90):                                    # 90: the last segment of each half is short (45 = 2*20 + 5)
        st = UniformStream(77 + num)
        arng = np.random.RandomState(78 + num)
        acts = []
        real_init = ref.MazeNavigation.__init__

        def init(self, *a, **k):
            real_init(self, *a, **k)
            box = self.action_space

            def sample():
                v = arng.uniform(box.low, box.high).astype(np.float32)       # gym Box.sample: float32 uniform
                acts.append(v)
                return v
            box.sample = sample
        ref.MazeNavigation.__init__ = init
        try:
            with patched_uniform(st), contextlib.redirect_stdout(io.StringIO()):
                data = ref.get_offline_data(num)
        finally:
            ref.MazeNavigation.__init__ = real_init
        assert len(data) == 2 * (num // 2) and len(acts) == num // 2
        pre = "off%d." % num
        out[pre + "u"], out[pre + "rand_actions"] = np.array(st.u), np.array(acts)
        out[pre + "s"] = np.array([np.asarray(t[0], dtype=np.float64) for t in data])
        out[pre + "a"] = np.array([np.asarray(t[1], dtype=np.float64) for t in data])
        out[pre + "c"] = np.array([int(t[2]) for t in data], dtype=np.uint8)
        out[pre + "s2"] = np.array([np.asarray(t[3], dtype=np.float64) for t in data])
        out[pre + "m"] = np.array([int(bool(t[4])) for t in data], dtype=np.uint8)

    np.savez_compressed(os.path.join(HERE, "maze_ref_golden.npz"), **out)
    print("step rows", n, "contacts", int(out["step.constraint"].sum()), "done", int(out["step.done"].sum()),
          "success", int(out["step.success"].sum()), "| reset redraws", int((out["reset.used"] > 2).sum()),
          "| offline violations", int(out["off1000.c"].sum()), "of", len(out["off1000.c"]),
          "| episode constraint steps", int(out["ep.constraint"].sum()))


if __name__ == "__main__":
    main()
