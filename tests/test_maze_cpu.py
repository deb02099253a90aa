"""CPU suite for the Maze oracle.  Control flow pinned to the REFERENCE's env/maze.py:34-232: the golden vectors of
maze_ref_golden.npz were produced by importing env/maze.py over a stand-in MjSim that implements the documented
kinematic surrogate (tests/golden/gen_maze_ref_golden.py, maze_stub_sim.py).  The physics itself (MuJoCo 1.50) is
not reproduced.  maze_oracle_golden.npz (the oracle's own output) only guards the Philox-driven paths."""
import os

import numpy as np

from oracle import c_oracle as co


def test_oracle_is_stable_against_its_own_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "maze_oracle_golden.npz"))
    o = co.maze_step(g["pos"], g["act"], g["t"], seed=5, counter=3, auto_reset=True)
    for k in ("next_pos64", "reward64", "done", "constraint", "success", "ep_done", "pos", "t", "obs"):
        assert np.array_equal(o[k], g["out_" + k]), k
    s, a, c, s2, m = co.maze_offline(2000, 9)
    for got, key in ((s, "off_s"), (a, "off_a"), (c, "off_c"), (s2, "off_s2"), (m, "off_m")):
        assert np.array_equal(got, g[key])


def test_contact_geometry_follows_the_xml():
    r = 0.025
    # wall1A after reset: x in [-0.105,-0.095], y in [0.22,0.62] (env/maze.py:201-203)
    assert co.maze_contact(-0.105 - r + 1e-9, 0.25) == 1 and co.maze_contact(-0.105 - r - 1e-6, 0.25) == 0
    assert co.maze_contact(-0.1, 0.22 - r + 1e-9) == 1 and co.maze_contact(-0.1, 0.22 - r - 1e-6) == 0
    # gaps: wall1 open for y in (-0.13, 0.22), wall2 for y in (0.03, 0.28)
    assert co.maze_contact(-0.1, 0.05) == 0 and co.maze_contact(0.1, 0.15) == 0
    assert co.maze_contact(0.1, -0.1) == 1 and co.maze_contact(-0.1, -0.2) == 1
    # arena planes at +-0.3
    assert co.maze_contact(0.3 - r + 1e-9, 0.0) == 1 and co.maze_contact(0.3 - r - 1e-6, 0.0) == 0
    assert co.maze_contact(0.0, -0.3 + r - 1e-9) == 1
    # corner rounding: beyond the wall end the disc must come within r of the corner
    assert co.maze_contact(-0.105 - 0.02, 0.22 - 0.02) == 0 and co.maze_contact(-0.105 - 0.017, 0.22 - 0.017) == 1


def test_step_semantics_follow_env_maze_py():
    pos = np.array([[-0.2, 0.0], [0.24, 0.0], [0.0, 0.0], [-0.13, -0.2], [0.0, 0.0]])
    act = np.array([[0.1, 0.0], [0.1, 0.0], [0.5, -0.5], [0.1, 0.0], [0.0, 0.0]], np.float32)
    t = np.array([0, 0, 0, 0, 99], np.int32)
    o = co.maze_step(pos, act, t)
    gain = 0.24667750873451577
    assert np.allclose(o["next_pos64"][0], [-0.2 + gain * 0.1, 0.0])                 # free motion
    assert np.allclose(o["next_pos64"][2], [gain * 0.1, -gain * 0.1])                # clipped to +-0.1
    d = np.sqrt(np.mean((np.array([0.25, 0.0]) - o["next_pos64"]) ** 2, axis=1))
    assert np.allclose(o["reward64"], -d, rtol=0, atol=1e-16)                        # -sqrt(mean(sq)), :215-220
    assert o["success"][1] == 1 and o["done"][1] == 1 and o["constraint"][1] == 0    # within 0.03 of the goal
    assert o["constraint"][3] == 1 and o["done"][3] == 1                              # ran into wall1B
    assert o["next_pos64"][3][0] < -0.1                                               # stopped at the wall
    assert o["done"][4] == 1 and o["constraint"][4] == 0 and o["t"][4] == 100         # env-internal horizon (:153)
    # in contact at the start of a step: no motion (env/maze.py:144-147)
    o2 = co.maze_step(o["next_pos64"][3:4], np.array([[-0.1, 0.0]], np.float32), np.zeros(1, np.int32))
    assert np.array_equal(o2["next_pos64"], o["next_pos64"][3:4]) and o2["constraint"][0] == 1


def test_reset_ranges_and_expert():
    for mode, (lo, hi) in {0: (-0.22, -0.13), 1: (0.14, 0.22), 2: (-0.04, 0.04), 3: (-0.27, 0.27)}.items():
        pos, obs, t = co.maze_reset(4000, mode=mode, seed=mode + 1)
        assert lo <= pos[:, 0].min() and pos[:, 0].max() <= hi
        assert -0.22 <= pos[:, 1].min() and pos[:, 1].max() <= 0.22
        assert not any(co.maze_contact(x, y) for x, y in pos)
        assert abs(pos[:, 0].mean() - (lo + hi) / 2) < 0.01
    assert np.allclose(co.maze_expert_action(-0.2, 0.1), 1.05 * (np.array([-0.15, -0.125]) - [-0.2, 0.1]))
    assert np.allclose(co.maze_expert_action(0.0, 0.0), 1.05 * np.array([0.15, 0.125]))
    assert np.allclose(co.maze_expert_action(0.2, 0.1), 1.05 * (np.array([0.25, 0.0]) - [0.2, 0.1]))
    # the scripted expert solves the surrogate from the default start most of the time
    p, _, t = co.maze_reset(500, seed=3)
    alive = np.ones(500, bool)
    succ = 0
    for _ in range(100):
        act = np.array([co.maze_expert_action(x, y) for x, y in p]).astype(np.float32)
        o = co.maze_step(p, act, t)
        end = alive & (o["done"] > 0)
        succ += int((end & (o["success"] > 0)).sum())
        alive &= o["done"] == 0
        p, t = np.where(alive[:, None], o["pos"], p), o["t"]
    assert succ > 400


def test_offline_data_layout():
    s, a, c, s2, m = co.maze_offline(10001, 4)
    assert len(s) == 10000                                      # 2 * (num // 2), env/maze.py:41,72
    assert 0.03 < c.mean() < 0.4 and set(np.unique(c)) <= {0.0, 1.0}
    assert np.all(m[c == 1] == 0)                               # mask = not done, done on constraint
    assert np.abs(a[:5000]).max() <= 0.1                        # random half: action_space.sample()
    assert np.abs(a[5000:]).max() > 0.1                         # expert half stores the raw 1.05*delta
    assert np.array_equal(s[1:20], s2[0:19])                    # 20-step segments are contiguous
    assert not np.array_equal(s[20], s2[19])                    # reset between segments


# ---- pinned to env/maze.py (reference imported over the stand-in MjSim) ---------------------------------------------
import pytest  # noqa: E402


@pytest.fixture(scope="module")
def R(golden_dir):
    return np.load(os.path.join(golden_dir, "maze_ref_golden.npz"))


def test_reference_model_constants(R):
    """What the stand-in read from the reference's simple_maze.xml and what reset() did to the walls."""
    assert list(R["model.geom_names"][5:9]) == ["wall1A", "wall2A", "wall1B", "wall2B"]
    assert abs(float(R["model.gain"]) - 0.24667750873451577) < 1e-15          # 500 x 2 ms semi-implicit Euler
    assert np.array_equal(R["model.wall_pos_after_reset"],
                          np.array([[-0.1, 0.5 + -0.08], [0.1, 0.4 + 0.08], [-0.1, -0.25 + -0.08], [0.1, -0.25 + 0.08]]))


def test_step_equals_env_maze_step(R):
    """env/maze.py:139-168 row by row: next state and reward bit-for-bit in float64, the three masks, the clipped
    action and the pre-step state of the info dict."""
    pos, act, steps = R["step.pos"], R["step.act"], R["step.steps"]
    for i in range(len(pos)):
        o = co.maze_step64(pos[i, 0], pos[i, 1], act[i, 0], act[i, 1], int(steps[i]))
        assert (o["x"], o["y"]) == tuple(R["step.next"][i]), i
        assert o["reward"] == R["step.reward"][i], i
        assert (o["done"], o["constraint"], o["success"]) == (R["step.done"][i], R["step.constraint"][i],
                                                               R["step.success"][i]), i
    assert np.array_equal(R["step.info_state"], pos)
    assert np.array_equal(R["step.info_action"], np.clip(act, -0.1, 0.1))
    assert R["step.constraint"].sum() > 500 and R["step.success"].sum() > 20 and (R["step.done"] > R["step.constraint"]).any()


def test_batched_f32_action_step_equals_env_maze_step(R):
    """The batched entry (float32 actions, what the kernels take) on the rows whose action is float32-valued."""
    act = R["step.act"]
    rows = np.where((act == act.astype(np.float32)).all(1))[0]
    assert len(rows) > 300
    o = co.maze_step(R["step.pos"][rows], act[rows].astype(np.float32), R["step.steps"][rows].astype(np.int32))
    assert np.array_equal(o["next_pos64"], R["step.next"][rows])
    assert np.array_equal(o["reward64"], R["step.reward"][rows])
    for k in ("done", "constraint", "success"):
        assert np.array_equal(o[k], R["step." + k][rows]), k


def test_expert_episodes_equal_the_reference(R):
    """40-step closed-loop episodes driven by expert_action (:222-232), incl. over-driven runs that end in a wall
    and stay stuck there (:144-147), positions bit-for-bit."""
    for k in range(R["ep.pos"].shape[0]):
        x, y = R["ep.pos"][k, 0]
        steps = 0
        for j in range(R["ep.act"].shape[1]):
            want = co.maze_expert_action(x, y) * (1.0 if k % 3 else 2.5)
            assert np.array_equal(want, R["ep.act"][k, j]), (k, j)
            o = co.maze_step64(x, y, want[0], want[1], steps)
            x, y, steps = o["x"], o["y"], o["steps"]
            assert (x, y) == tuple(R["ep.pos"][k, j + 1]) and o["reward"] == R["ep.reward"][k, j]
            assert (o["done"], o["constraint"]) == (R["ep.done"][k, j], R["ep.constraint"][k, j])
            assert co.maze_distance(x, y) == R["ep.dist"][k, j]
    assert R["ep.constraint"].sum() > 0 and (R["ep.reward"] > -0.03).any()


def test_expert_and_distance_equal_the_reference(R):
    for (x, y), a, d in zip(R["expert.pos"], R["expert.act"], R["expert.dist"]):
        assert np.array_equal(co.maze_expert_action(x, y), a)
        assert co.maze_distance(x, y) == d


def test_reset_equals_env_maze_reset(R):
    """:184-213 with the reference's own uniforms: ranges per difficulty, y range, re-draw while in contact."""
    for mode, check, u, pos, used in zip(R["reset.mode"], R["reset.check"], R["reset.u"], R["reset.pos"], R["reset.used"]):
        x, y, n = co.maze_reset_explicit(int(mode), bool(check), u)
        assert (x, y) == tuple(pos) and n == used
    assert (R["reset.used"] > 2).sum() >= 5                           # the re-draw branch is exercised


@pytest.mark.parametrize("num", (1000, 90))
def test_offline_data_equals_env_maze_get_offline_data(R, num):
    """:34-107 row for row (float64 states and next states, actions, constraint flags, masks) given the reference's
    uniform stream and its action_space.sample() stream."""
    pre = "off%d." % num
    (s, a, c, s2, m), (s64, a64, s2_64), used = co.maze_offline_explicit(num, R[pre + "u"], R[pre + "rand_actions"])
    assert used == len(R[pre + "u"])
    assert np.array_equal(s64, R[pre + "s"]) and np.array_equal(s2_64, R[pre + "s2"])
    assert np.array_equal(a64, R[pre + "a"])
    assert np.array_equal(c, R[pre + "c"].astype(np.float32)) and np.array_equal(m, R[pre + "m"].astype(np.float32))
    assert np.array_equal(s, R[pre + "s"].astype(np.float32)) and np.array_equal(a, R[pre + "a"].astype(np.float32))
