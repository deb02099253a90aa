"""CPU suite for the Maze surrogate oracle (PARITY UNPINNED -- see tests/golden/gen_maze_golden.py):
control flow / reward / termination follow env/maze.py; geometry follows simple_maze.xml."""
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
