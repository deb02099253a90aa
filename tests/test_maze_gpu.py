"""GPU parity tests for the Maze kernels: bit-exact against the C oracle, and against the rows recorded from the
REFERENCE's env/maze.py run over the stand-in MjSim (tests/golden/maze_ref_golden.npz).  The control flow is pinned
to env/maze.py:34-232; the physics is the documented surrogate, not MuJoCo."""
import os

import numpy as np
import pytest
import torch

from oracle import c_oracle as co
from recovery_rl_amd import _lib
from recovery_rl_amd.env import make_env, make_vec_env, register_env
from recovery_rl_amd.env.maze import offline_data

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def hip_step(pos, action, t, seed=0, counter=0, horizon=100, auto_reset=False):
    lib = _lib.load()
    n = len(pos)
    d = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a, dtype=dt), device=DEV)
    pos_t, act_t, t_t = d(pos, np.float64), d(action, np.float32), d(t, np.int32)
    o = dict(next_obs=torch.zeros(n, 2, device=DEV), obs=torch.zeros(n, 2, device=DEV),
             reward=torch.zeros(n, device=DEV))
    for k in ("done", "constraint", "success", "ep_done"):
        o[k] = torch.zeros(n, dtype=torch.uint8, device=DEV)
    rc = lib.rrl_maze_step(n, _lib.ptr(pos_t), _lib.ptr(act_t), seed, counter, None, 0,
                           _lib.ptr(o["next_obs"]), _lib.ptr(o["obs"]), _lib.ptr(o["reward"]),
                           _lib.ptr(o["done"]), _lib.ptr(o["constraint"]), _lib.ptr(o["success"]),
                           _lib.ptr(o["ep_done"]), _lib.ptr(t_t), horizon, int(auto_reset),
                           _lib.current_stream())
    assert rc == 0
    torch.cuda.synchronize()
    out = {k: v.cpu().numpy() for k, v in o.items()}
    out["pos"], out["t"] = pos_t.cpu().numpy(), t_t.cpu().numpy()
    return out


def test_step_matches_oracle_golden_rows(golden_dir):
    g = np.load(os.path.join(golden_dir, "maze_oracle_golden.npz"))
    got = hip_step(g["pos"], g["act"], g["t"], seed=5, counter=3, auto_reset=True)
    for k in ("done", "constraint", "success", "ep_done", "pos", "t", "obs"):
        assert np.array_equal(got[k], g["out_" + k]), k
    assert np.array_equal(got["next_obs"], g["out_next_pos64"].astype(np.float32))
    assert np.array_equal(got["reward"], g["out_reward64"].astype(np.float32))


def test_step_equals_env_maze_step_rows(golden_dir):
    """rrl_maze_step against env/maze.py:139-168 itself: the float32-action rows of the reference golden -- next state
    bit-for-bit in float64, reward, the three masks."""
    R = np.load(os.path.join(golden_dir, "maze_ref_golden.npz"))
    act = R["step.act"]
    rows = np.where((act == act.astype(np.float32)).all(1))[0]
    assert len(rows) > 1000
    got = hip_step(R["step.pos"][rows], act[rows].astype(np.float32), R["step.steps"][rows].astype(np.int32))
    assert np.array_equal(got["pos"], R["step.next"][rows])                       # no auto-reset: pos = s'
    assert np.array_equal(got["next_obs"], R["step.next"][rows].astype(np.float32))
    assert np.array_equal(got["reward"], R["step.reward"][rows].astype(np.float32))
    for k in ("done", "constraint", "success"):
        assert np.array_equal(got[k], R["step." + k][rows]), k
    assert R["step.constraint"][rows].sum() > 200


def test_expert_episodes_equal_the_reference(golden_dir):
    """Closed-loop episodes of the reference (expert_action :222-232 + step) replayed through the kernel: the float32
    rounding of the expert action is the only difference, so positions agree to 1e-6 and every mask is equal."""
    R = np.load(os.path.join(golden_dir, "maze_ref_golden.npz"))
    K, T = R["ep.act"].shape[:2]
    pos = R["ep.pos"][:, 0].copy()
    t = np.zeros(K, np.int32)
    for j in range(T):
        got = hip_step(pos, R["ep.act"][:, j].astype(np.float32), t)
        assert np.abs(got["pos"] - R["ep.pos"][:, j + 1]).max() < 1e-6, j
        assert np.array_equal(got["constraint"], R["ep.constraint"][:, j].astype(np.uint8)), j
        assert np.array_equal(got["done"], R["ep.done"][:, j].astype(np.uint8)), j
        pos, t = R["ep.pos"][:, j + 1].copy(), got["t"]


@pytest.mark.parametrize("n", (1, 65, 4096, 70001))
def test_step_matches_oracle_random(n):
    rng = np.random.RandomState(n)
    pos = rng.uniform(-0.29, 0.29, (n, 2))
    act = rng.uniform(-0.12, 0.12, (n, 2)).astype(np.float32)
    t = rng.randint(0, 100, n).astype(np.int32)
    for auto in (False, True):
        ref = co.maze_step(pos, act, t, seed=77, counter=5, auto_reset=auto)
        got = hip_step(pos, act, t, seed=77, counter=5, auto_reset=auto)
        for k in ("pos", "t", "next_obs", "obs", "reward", "done", "constraint", "success", "ep_done"):
            assert np.array_equal(got[k], ref[k]), k


def test_bracketed_collision_search_equals_the_scan_near_walls():
    """The kernel finds the first of the 64 sub-steps in contact by a bracketed search; the oracle scans them one by
    one.  400 000 moves that start within a few sub-steps of a wall face, wall end, corner arc or arena plane."""
    rng = np.random.RandomState(3)
    n = 400000
    wx, wy = np.array([-0.1, 0.1, -0.1, 0.1]), np.array([0.42, 0.48, -0.33, -0.17])
    j = rng.randint(0, 4, n)
    d = 0.02501 + 0.01 * rng.uniform(size=n)
    side = np.where(rng.uniform(size=n) < 0.5, -1.0, 1.0)
    face = rng.uniform(size=n) < 0.4
    x = np.where(face, wx[j] + side * (0.005 + d), wx[j] + rng.uniform(-0.04, 0.04, n))
    y = np.where(face, wy[j] + rng.uniform(-0.24, 0.24, n), wy[j] + side * (0.2 + d))
    ang, rad = rng.uniform(0, 2 * np.pi, n), 0.025 + rng.uniform(0, 1e-3, n)
    corner = rng.uniform(size=n) < 0.25
    x = np.where(corner, wx[j] + side * 0.005 + rad * np.cos(ang), x)
    y = np.where(corner, wy[j] + np.where(rng.uniform(size=n) < 0.5, -0.2, 0.2) + rad * np.sin(ang), y)
    plane = rng.uniform(size=n) < 0.1
    x = np.where(plane, side * rng.uniform(0.27, 0.2749, n), x)
    y = np.where((y > 0.274) | (y < -0.274), rng.uniform(-0.27, 0.27, n), y)
    pos = np.c_[x, y]
    act = rng.uniform(-0.12, 0.12, (n, 2)).astype(np.float32)
    act[::5] *= 0.02                                             # moves of a few sub-step lengths in total
    act[1::7, 1] = 0.0
    t = np.zeros(n, np.int32)
    ref = co.maze_step(pos, act, t)
    got = hip_step(pos, act, t)
    for k in ("pos", "next_obs", "reward", "done", "constraint", "success"):
        assert np.array_equal(got[k], ref[k]), k
    started_free = np.array([co.maze_contact(a, b) for a, b in pos[:20000]]) == 0
    assert (ref["constraint"][:20000][started_free] == 1).sum() > 2000      # many of the moves do run into something


def test_vec_env_episode_matches_oracle():
    n = 512
    env = make_vec_env("maze", n, device=DEV, seed=21)
    env.reset()
    pos, _, t = co.maze_reset(n, seed=21, counter=0)
    assert np.array_equal(env.pos.cpu().numpy(), pos)
    for k in range(150):
        act = (env.expert_action() if k % 3 else env.sample_actions()).contiguous()
        obs, rew, done, info = env.step(act)
        ref = co.maze_step(pos, act.cpu().numpy(), t, seed=21, counter=1 + k, auto_reset=True)
        pos, t = ref["pos"], ref["t"]
        assert np.array_equal(env.pos.cpu().numpy(), pos), k
        assert np.array_equal(rew.cpu().numpy(), ref["reward"])
        assert np.array_equal(done.cpu().numpy(), ref["done"])
        assert np.array_equal(info["constraint"].cpu().numpy(), ref["constraint"])
        assert np.array_equal(info["success"].cpu().numpy(), ref["success"])


@pytest.mark.parametrize("mode,name", ((0, 'h'), (1, 'e'), (2, 'm'), (3, None)))
def test_reset_modes_match_oracle(mode, name):
    env = make_vec_env("maze", 3000, device=DEV, seed=8)
    env.reset(difficulty=name)
    pos, obs, _ = co.maze_reset(3000, mode=mode, seed=8, counter=0)
    assert np.array_equal(env.pos.cpu().numpy(), pos)
    assert np.array_equal(env.obs.cpu().numpy(), obs)


@pytest.mark.parametrize("num", (0, 41, 10000))
def test_offline_data_matches_oracle(num):
    got = [x.cpu().numpy() for x in offline_data(num, seed=4, device=DEV)]
    ref = co.maze_offline(num, 4)
    assert len(got[0]) == len(ref[0]) == 2 * (num // 2)
    for a, b in zip(got, ref):
        assert np.array_equal(a, b)


def test_single_env_protocol():
    register_env("maze")
    env = make_env("maze", device=DEV, seed=2)
    assert env._max_episode_steps == 100 and env.action_space.high[0] == pytest.approx(0.1)
    assert env.observation_space.shape == (2,)
    s = env.reset()
    assert -0.22 <= s[0] <= -0.13
    env.reset(pos=(0.2, 0.0))
    obs, r, done, info = env.step(env.expert_action())
    assert set(info) == {"constraint", "reward", "state", "next_state", "action", "success"}
    assert r == pytest.approx(-env.get_distance_score(), abs=1e-6)
    for _ in range(5):
        obs, r, done, info = env.step(env.expert_action())
    assert done and info["success"] and not info["constraint"]


def test_fused_step_push_matches_oracle_step_plus_pushes():
    """rrl_maze_step_push == checker maze_step + two checker replay pushes + counters, over a ring wrap-around."""
    import ctypes as C
    from recovery_rl_amd.replay_memory import ConstraintReplayMemory, ReplayMemory
    lib = _lib.load()
    n, cap = 1500, 4000
    rng = np.random.RandomState(4)
    venv = make_vec_env("maze", n, device=DEV, seed=31)
    venv.reset()
    mem, rmem = ReplayMemory(cap, 1, device=DEV), ConstraintReplayMemory(cap, 1, device=DEV)
    omem, ormem = co.OracleReplay(cap), co.OracleReplay(cap)
    stats = torch.zeros(10, dtype=torch.int64, device=DEV)
    sums = torch.zeros(2, dtype=torch.float64, device=DEV)
    ep_reward = torch.zeros(n, device=DEV)
    pos, t = venv.pos.cpu().numpy().copy(), np.zeros(n, np.int32)
    tick0 = int(venv.tick[0].item())
    ref_stats = np.zeros(8, np.int64)
    ref_ep, ref_sums = np.zeros(n, np.float32), np.zeros(2)
    for k in range(5):
        obs_prev = venv.obs.cpu().numpy().copy()
        task = torch.as_tensor(rng.uniform(-0.1, 0.1, (n, 2)).astype(np.float32), device=DEV)
        real = torch.as_tensor(rng.uniform(-0.13, 0.13, (n, 2)).astype(np.float32), device=DEV)
        rec = torch.as_tensor((rng.uniform(size=n) < 0.3).astype(np.uint8), device=DEV)
        rc = lib.rrl_maze_step_push(
            n, _lib.ptr(venv.pos), _lib.ptr(venv.t), _lib.ptr(venv.obs), _lib.ptr(task), _lib.ptr(real),
            _lib.ptr(rec), venv.seed_value, 0, _lib.ptr(venv.tick), 1, 100, 1, 2.5, 0, C.byref(mem._desc),
            C.byref(rmem._desc), _lib.ptr(venv.next_obs), _lib.ptr(venv.reward), _lib.ptr(venv.done),
            _lib.ptr(venv.constraint), _lib.ptr(venv.success), _lib.ptr(venv.ep_done), _lib.ptr(stats),
            _lib.ptr(sums), _lib.ptr(ep_reward), _lib.current_stream())
        assert rc == 0
        ref = co.maze_step(pos, real.cpu().numpy(), t, seed=venv.seed_value, counter=tick0 + k, auto_reset=True)
        pos, t = ref["pos"], ref["t"]
        mask = 1.0 - ref["done"].astype(np.float32)
        cons = ref["constraint"].astype(np.float32)
        omem.push(obs_prev, task.cpu().numpy(), ref["reward"] - 2.5 * cons, ref["next_obs"], mask)
        ormem.push(obs_prev, real.cpu().numpy(), cons, ref["next_obs"], mask)
        assert np.array_equal(venv.pos.cpu().numpy(), pos) and np.array_equal(venv.obs.cpu().numpy(), ref["obs"])
        assert np.array_equal(venv.t.cpu().numpy(), t)
        r = rec.cpu().numpy().astype(bool)
        epd, c, s_ = ref["ep_done"].astype(bool), ref["constraint"].astype(bool), ref["success"].astype(bool)
        ref_stats += [n, epd.sum(), (epd & c).sum(), (epd & c & r).sum(), (epd & c & ~r).sum(), (epd & s_).sum(),
                      r.sum(), c.sum()]
        ref_ep += ref["reward"]
        ref_sums += [ref["reward"].astype(np.float64).sum(), ref_ep[epd].astype(np.float64).sum()]
        ref_ep[epd] = 0
    assert ref_stats[1] > 0 and ref_stats[7] > 0            # episodes ended and walls were hit
    for got, want in ((mem, omem), (rmem, ormem)):
        assert int(got.state[0].item()) == want.pos and int(got.state[1].item()) == want.size == cap
        for a_, b_ in ((got.s, want.s), (got.a, want.a), (got.r, want.r), (got.s2, want.s2), (got.m, want.m)):
            assert np.array_equal(a_.cpu().numpy(), b_)
    assert np.array_equal(stats.cpu().numpy()[:8], ref_stats)
    assert np.allclose(sums.cpu().numpy(), ref_sums, rtol=1e-9)
    assert np.allclose(ep_reward.cpu().numpy(), ref_ep, rtol=1e-6, atol=1e-5)
    assert int(venv.tick[0].item()) == tick0 + 5
