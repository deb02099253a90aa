"""GPU parity tests for the navigation kernels: HIP (through the C ABI) vs the golden vectors
captured from the reference, and vs the C oracle on seeded inputs.  Bit-exact throughout."""
import os

import numpy as np
import pytest
import torch

from oracle import c_oracle as co
from recovery_rl_amd import _lib
from recovery_rl_amd.env import make_env, make_vec_env, register_env
from recovery_rl_amd.env.navigation import offline_data

pytestmark = pytest.mark.gpu
ENVS = ("navigation1", "navigation2")
DEV = "cuda:0"


def hip_step(env_name, pos, action, t, noise=None, seed=0, counter=0, horizon=100, auto_reset=False,
             tick=None, inc=0):
    lib = _lib.load()
    n = len(pos)
    d = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a, dtype=dt), device=DEV)
    pos_t, act_t, t_t = d(pos, np.float64), d(action, np.float32), d(t, np.int32)
    noise_t = None if noise is None else d(noise, np.float64)
    o = dict(next_obs=torch.zeros(n, 2, device=DEV), obs=torch.zeros(n, 2, device=DEV),
             reward=torch.zeros(n, device=DEV))
    for k in ("done", "constraint", "success", "ep_done"):
        o[k] = torch.zeros(n, dtype=torch.uint8, device=DEV)
    rc = lib.rrl_nav_step(co.ENV_KIND[env_name], n, _lib.ptr(pos_t), _lib.ptr(act_t), _lib.ptr(noise_t),
                          seed, counter, _lib.ptr(tick), inc, _lib.ptr(o["next_obs"]), _lib.ptr(o["obs"]),
                          _lib.ptr(o["reward"]), _lib.ptr(o["done"]), _lib.ptr(o["constraint"]),
                          _lib.ptr(o["success"]), _lib.ptr(o["ep_done"]), _lib.ptr(t_t), horizon,
                          int(auto_reset), _lib.current_stream())
    assert rc == 0
    torch.cuda.synchronize()
    out = {k: v.cpu().numpy() for k, v in o.items()}
    out["pos"], out["t"] = pos_t.cpu().numpy(), t_t.cpu().numpy()
    return out


def assert_same(a, b, keys=("pos", "t", "next_obs", "obs", "reward", "done", "constraint", "success", "ep_done")):
    for k in keys:
        assert np.array_equal(a[k], b[k]), "%s differs in %d rows" % (k, int((a[k] != b[k]).sum()))


@pytest.mark.parametrize("env", ENVS)
def test_step_matches_reference_golden_bit_exact(env, golden_dir):
    """G1: explicit (s, a, eps) rows incl. every box edge +-ulp, the stuck-in-obstacle branch and
    ||s|| straddling 4.  Masks bit-exact; s' and reward equal to the fp64 reference rounded to f32."""
    g = np.load(os.path.join(golden_dir, "nav_step_golden.npz"))
    S, A, E = g[env + "_s"], g[env + "_a"], g[env + "_eps"]
    o = hip_step(env, S, A, np.zeros(len(S), np.int32), noise=E)
    assert np.array_equal(o["pos"], g[env + "_s2"])                     # fp64 state, exact
    assert np.array_equal(o["next_obs"], g[env + "_s2"].astype(np.float32))
    assert np.array_equal(o["reward"], g[env + "_reward"].astype(np.float32))
    for k in ("done", "constraint", "success"):
        assert np.array_equal(o[k], g[env + "_" + k]), k


@pytest.mark.parametrize("env", ENVS)
@pytest.mark.parametrize("n", (1, 63, 64, 257, 4096, 100003, 1 << 19, 600000, (1 << 22) + 2))   # >= 2^19, n % 4 == 0: 4 envs/thread; >= 2^22, even: 2
def test_step_matches_oracle_with_philox_noise(env, n):
    rng = np.random.RandomState(n)
    pos = np.c_[rng.uniform(-60, 10, n), rng.uniform(-12, 12, n)]
    act = rng.uniform(-1.5, 1.5, (n, 2)).astype(np.float32)
    t = rng.randint(0, 100, n).astype(np.int32)
    for auto in (False, True):
        ref = co.nav_step(env, pos, act, t, seed=0xDEADBEEF12345, counter=17, auto_reset=auto)
        got = hip_step(env, pos, act, t, seed=0xDEADBEEF12345, counter=17, auto_reset=auto)
        assert_same(got, ref)


def hip_step_compact(env_name, pos, action, t, noise=None, seed=0, counter=0, horizon=100, auto_reset=False,
                     stale_flags=0):
    """rrl_nav_step_compact; returns the same dict as hip_step (flags and count unpacked from the status words,
    obs = reset_obs where ep_done and auto_reset, next_obs elsewhere) plus the raw outputs."""
    lib = _lib.load()
    n = len(pos)
    d = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a, dtype=dt), device=DEV)
    pos_t, act_t = d(pos, np.float64), d(action, np.float32)
    status = torch.as_tensor((np.asarray(t).astype(np.int64) | stale_flags).astype(np.int16), device=DEV)
    noise_t = None if noise is None else d(noise, np.float64)
    next_obs = torch.zeros(n, 2, device=DEV)
    reset_obs = torch.full((n, 2), -7.0, device=DEV)
    reward = torch.zeros(n, device=DEV)
    rc = lib.rrl_nav_step_compact(co.ENV_KIND[env_name], n, _lib.ptr(pos_t), _lib.ptr(act_t), _lib.ptr(noise_t), seed,
                                  counter, None, 0, _lib.ptr(next_obs), _lib.ptr(reset_obs), _lib.ptr(reward),
                                  _lib.ptr(status), horizon, int(auto_reset), _lib.current_stream())
    assert rc == 0
    torch.cuda.synchronize()
    st = status.cpu().numpy().astype(np.uint16).astype(np.int64)
    out = dict(pos=pos_t.cpu().numpy(), next_obs=next_obs.cpu().numpy(), reward=reward.cpu().numpy(),
               t=(st & 0x0fff).astype(np.int32), reset_obs=reset_obs.cpu().numpy())
    for bit, k in ((12, "done"), (13, "constraint"), (14, "success"), (15, "ep_done")):
        out[k] = ((st >> bit) & 1).astype(np.uint8)
    fin = (out["ep_done"] != 0) & bool(auto_reset)
    out["obs"] = np.where(fin[:, None], out["reset_obs"], out["next_obs"])
    return out


@pytest.mark.parametrize("env", ENVS)
@pytest.mark.parametrize("n", (1, 2, 3, 4, 5, 63, 64, 257, 4096, 100003, 1 << 19, 600001, (1 << 22) + 3))
def test_compact_step_matches_oracle_with_philox_noise(env, n):
    """rrl_nav_step_compact (u16 status words, sparse post-reset observation, finished episodes of a wave re-drawn
    once per workgroup and pass) against the C oracle: every field bit-exact; reset_obs rows of continuing episodes are
    untouched or repeat next_obs."""
    rng = np.random.RandomState(n + 5)
    pos = np.c_[rng.uniform(-60, 10, n), rng.uniform(-12, 12, n)]
    act = rng.uniform(-1.5, 1.5, (n, 2)).astype(np.float32)
    t = rng.randint(0, 100, n).astype(np.int32)
    for auto in (False, True):
        ref = co.nav_step(env, pos, act, t, seed=0xDEADBEEF12345, counter=17, auto_reset=auto)
        got = hip_step_compact(env, pos, act, t, seed=0xDEADBEEF12345, counter=17, auto_reset=auto,
                               stale_flags=0xF000)            # flag bits of the previous step are ignored on input
        assert_same(got, ref)
        fin = (ref["ep_done"] != 0) & auto
        rest = got["reset_obs"][~fin]                      # untouched, or a copy of next_obs (whole 32-byte groups)
        assert np.all((rest == -7.0) | (rest == got["next_obs"][~fin]))
        assert np.any(rest == -7.0) or fin.mean() > 0.2 or n < 8


@pytest.mark.parametrize("env", ENVS)
def test_compact_step_every_episode_ending_at_once(env):
    """All envs reach the horizon on the same step (what a synchronously started sweep does): more finished episodes
    per wave than its reset list holds -> the per-lane path; and a mixed case around the 64-entry limit."""
    n = 8192
    rng = np.random.RandomState(3)
    pos = np.c_[rng.uniform(-60, -40, n), rng.uniform(-3, 3, n)]
    act = rng.uniform(-1, 1, (n, 2)).astype(np.float32)
    for t in (np.full(n, 99, np.int32), np.where(np.arange(n) % 4 == 0, 99, 5).astype(np.int32),
              np.where(np.arange(n) % 5 == 0, 99, 5).astype(np.int32)):
        ref = co.nav_step(env, pos, act, t, seed=11, counter=3, auto_reset=True)
        assert_same(hip_step_compact(env, pos, act, t, seed=11, counter=3, auto_reset=True), ref)
        assert_same(hip_step(env, pos, act, t, seed=11, counter=3, auto_reset=True), ref)


@pytest.mark.parametrize("env", ENVS)
def test_compact_step_matches_reference_golden_bit_exact(env, golden_dir):
    """G1 rows (explicit noise) through the compact entry."""
    g = np.load(os.path.join(golden_dir, "nav_step_golden.npz"))
    S, A, E = g[env + "_s"], g[env + "_a"], g[env + "_eps"]
    o = hip_step_compact(env, S, A, np.zeros(len(S), np.int32), noise=E)
    assert np.array_equal(o["pos"], g[env + "_s2"])
    assert np.array_equal(o["next_obs"], g[env + "_s2"].astype(np.float32))
    assert np.array_equal(o["reward"], g[env + "_reward"].astype(np.float32))
    for k in ("done", "constraint", "success"):
        assert np.array_equal(o[k], g[env + "_" + k]), k


def test_compact_step_rejects_what_the_status_word_cannot_hold():
    lib = _lib.load()
    z = torch.zeros(8, 2, dtype=torch.float64, device=DEV)
    f = torch.zeros(8, 2, device=DEV)
    st = torch.zeros(8, dtype=torch.int16, device=DEV)
    args = lambda horizon, pos=z: (0, 8, _lib.ptr(pos), _lib.ptr(f), None, 0, 0, None, 0, _lib.ptr(f), None,
                                   _lib.ptr(f[:, 0].contiguous()), _lib.ptr(st), horizon, 0, _lib.current_stream())
    assert lib.rrl_nav_step_compact(*args(4096)) == -3
    assert lib.rrl_nav_step_compact(*args(4095)) == 0
    assert lib.rrl_nav_step_compact(*args(100, pos=z.view(-1)[1:9].view(4, 2))) == -1   # 8-byte aligned pos
    torch.cuda.synchronize()


@pytest.mark.parametrize("env", ENVS)
def test_vector_path_with_explicit_noise_and_device_tick(env):
    """The 4-envs-per-thread kernel (n >= 2^19) with caller-supplied noise, and its device-side tick."""
    n = 1 << 19
    rng = np.random.RandomState(5)
    pos = np.c_[rng.uniform(-60, 10, n), rng.uniform(-12, 12, n)]
    act = rng.uniform(-1.5, 1.5, (n, 2)).astype(np.float32)
    t = rng.randint(0, 100, n).astype(np.int32)
    noise = rng.randn(n, 2)
    ref = co.nav_step(env, pos, act, t, noise=noise, seed=3, counter=9, auto_reset=True)
    got = hip_step(env, pos, act, t, noise=noise, seed=3, counter=9, auto_reset=True)
    assert_same(got, ref)
    tick = torch.tensor([4, 0], dtype=torch.int64, device=DEV)
    got = hip_step(env, pos, act, t, seed=3, counter=5, auto_reset=True, tick=tick, inc=2)
    assert_same(got, co.nav_step(env, pos, act, t, seed=3, counter=9, auto_reset=True))
    assert tick.tolist() == [6, 0]


@pytest.mark.parametrize("env", ENVS)
def test_trajectories_stay_bit_identical_over_an_episode(env):
    """300 lock-step vector steps with auto-reset through the host mirror (VecEnv) vs the oracle:
    state, masks and the device-side RNG tick must never drift."""
    n = 1024
    venv = make_vec_env(env, n, device=DEV, seed=99)
    venv.reset()
    pos = venv.pos.cpu().numpy().copy()
    t = np.zeros(n, np.int32)
    ref_pos, _, _ = co.nav_reset(env, n, seed=99, counter=0)
    assert np.array_equal(pos, ref_pos)
    g = torch.Generator(device=DEV)
    g.manual_seed(5)
    viol = succ = 0
    for k in range(300):
        act = venv.sample_actions(g) * 1.2 + torch.tensor([0.9, 0.0], device=DEV)
        obs, rew, done, info = venv.step(act.contiguous())
        ref = co.nav_step(env, pos, act.cpu().numpy(), t, seed=99, counter=1 + k, auto_reset=True)
        pos, t = ref["pos"], ref["t"]
        assert np.array_equal(venv.pos.cpu().numpy(), pos), k
        assert np.array_equal(venv.t.cpu().numpy(), t)
        assert np.array_equal(rew.cpu().numpy(), ref["reward"])
        assert np.array_equal(done.cpu().numpy(), ref["done"])
        assert np.array_equal(info["constraint"].cpu().numpy(), ref["constraint"])
        assert np.array_equal(info["ep_done"].cpu().numpy(), ref["ep_done"])
        assert np.array_equal(info["next_state"].cpu().numpy(), ref["next_obs"])
        assert np.array_equal(obs.cpu().numpy(), ref["obs"])
        viol += int(ref["constraint"].sum())
        succ += int(ref["success"].sum())
    assert int(venv.tick[0].item()) == 301 and int(venv.tick[1].item()) == 0
    assert viol > 0


def test_reset_matches_oracle_and_mask():
    lib = _lib.load()
    n = 5000
    pos = torch.full((n, 2), 7.0, dtype=torch.float64, device=DEV)
    obs = torch.zeros(n, 2, device=DEV)
    t = torch.full((n,), 5, dtype=torch.int32, device=DEV)
    mask = (torch.arange(n, device=DEV) % 3 == 0).to(torch.uint8)
    assert lib.rrl_nav_reset(0, n, _lib.ptr(pos), _lib.ptr(obs), _lib.ptr(t), _lib.ptr(mask), None, 42, 9,
                             None, _lib.current_stream()) == 0
    ref_pos, ref_obs, _ = co.nav_reset("navigation1", n, seed=42, counter=9)
    m = mask.cpu().numpy().astype(bool)
    assert np.array_equal(pos.cpu().numpy()[m], ref_pos[m])
    assert np.array_equal(obs.cpu().numpy()[m], ref_obs[m])
    assert (pos.cpu().numpy()[~m] == 7.0).all() and (t.cpu().numpy()[~m] == 5).all()
    assert (t.cpu().numpy()[m] == 0).all()


@pytest.mark.parametrize("env", ENVS)
def test_rollout_kernel_matches_oracle(env):
    lib = _lib.load()
    n, T = 3000, 12
    rng = np.random.RandomState(3)
    pos = np.c_[rng.uniform(-60, 10, n), rng.uniform(-6, 6, n)]
    acts = rng.uniform(-1.2, 1.2, (T, n, 2)).astype(np.float32)
    ref = co.nav_rollout(env, pos, acts, seed=77, counter=1000)
    pos_t = torch.as_tensor(pos, device=DEV)
    acts_t = torch.as_tensor(acts, device=DEV)
    obs = torch.zeros(T, n, 2, device=DEV)
    rew = torch.zeros(T, n, device=DEV)
    cons = torch.zeros(T, n, dtype=torch.uint8, device=DEV)
    done = torch.zeros(T, n, dtype=torch.uint8, device=DEV)
    assert lib.rrl_nav_rollout(co.ENV_KIND[env], n, T, _lib.ptr(pos_t), _lib.ptr(acts_t), 77, 1000, None,
                               _lib.ptr(obs), _lib.ptr(rew), _lib.ptr(cons), _lib.ptr(done),
                               _lib.current_stream()) == 0
    torch.cuda.synchronize()
    assert np.array_equal(pos_t.cpu().numpy(), ref["pos"])
    assert np.array_equal(obs.cpu().numpy(), ref["obs"])
    assert np.array_equal(rew.cpu().numpy(), ref["reward"])
    assert np.array_equal(cons.cpu().numpy(), ref["constraint"])
    assert np.array_equal(done.cpu().numpy(), ref["done"])


@pytest.mark.parametrize("env", ENVS)
@pytest.mark.parametrize("num", (0, 9, 1000, 20000))
def test_offline_data_matches_oracle(env, num):
    s, a, c, s2, m = (x.cpu().numpy() for x in offline_data(env, num, seed=1, device=DEV))
    rs, ra, rc_, rs2, rm = co.nav_offline(env, num, 1)
    assert len(s) == len(rs)
    for got, ref in ((s, rs), (a, ra), (c, rc_), (s2, rs2), (m, rm)):
        assert np.array_equal(got, ref)


def test_single_env_gym_protocol_reads_like_the_reference():
    """make_env/register_env (env/make_utils.py:23-31) and reset/step/info keys
    (env/navigation1.py:71-97)."""
    with pytest.raises(KeyError):
        register_env("no_such_env")
    register_env("navigation1")
    env = make_env("navigation1", device=DEV, seed=0)
    assert env._max_episode_steps == 100 and env.action_space.shape == (2,)
    assert env.observation_space.shape == (2,) and list(env.goal) == [0, 0]
    s = env.reset()
    assert s.shape == (2,) and s.dtype == np.float64 and abs(s[0] + 50) < 6
    obs, r, done, info = env.step(np.array([2.0, -3.0]))
    assert set(info) == {"constraint", "reward", "state", "next_state", "action", "success"}
    assert np.array_equal(info["action"], [1.0, -1.0]) and np.array_equal(info["state"], s)
    assert np.isclose(r, -np.hypot(*s)) and not done
    assert np.allclose(obs - s, [1, -1], atol=0.4)
    data = env.transition_function(200)
    assert len(data) > 50 and len(data[0]) == 5


def test_full_size_properties_4096x100():
    """BASELINE size (4096 envs, 100-step horizon): properties that need no oracle run."""
    n = 4096
    env = make_vec_env("navigation1", n, device=DEV, seed=1)
    obs = env.reset()
    ep_ends = torch.zeros(n, device=DEV)
    for k in range(100):
        prev = obs.clone()
        act = torch.zeros(n, 2, device=DEV)
        act[:, 0] = 1.0
        obs, rew, done, info = env.step(act)
        # reward is -||previous obs|| (f64 state rounded): tight tolerance
        assert torch.allclose(rew, -prev.double().norm(dim=1).float(), rtol=0, atol=2e-5)
        assert torch.equal(info["ep_done"] >= done, torch.ones_like(done, dtype=torch.bool))
        moved = info["next_state"] - prev
        stuck = info["constraint"].bool() & (moved.abs().sum(1) == 0)
        free = ~stuck
        assert (moved[free, 0] - 1.0).abs().max() < 0.4          # 0.05 * |N(0,1)| < 0.4
        ep_ends += info["ep_done"].float()
    assert (env.t <= 100).all() and (env.t >= 0).all()
    # driving right at unit speed from x=-50 reaches the goal radius (4) after ~46 steps
    assert ep_ends.min() >= 1 and ep_ends.max() <= 4


@pytest.mark.parametrize("env", ENVS)
@pytest.mark.parametrize("with_outputs,n,cap", [(True, 1500, 4000), (False, 1500, 4000), (True, 40000, 100000),
                                                (False, 300000, 700000)])
def test_fused_step_push_matches_oracle_step_plus_pushes(env, with_outputs, n, cap):
    """rrl_nav_step_push == oracle nav_step + two oracle replay pushes + counters, over a wrap-around; with and without
    the optional per-env output arrays (next_obs, reward, flags: NULL = not written, everything else unchanged)."""
    import ctypes as C
    from recovery_rl_amd.replay_memory import ConstraintReplayMemory, ReplayMemory
    lib = _lib.load()
    rng = np.random.RandomState(4)      # n = 40000 / 300000: the bandwidth-regime instances (256- / 1024-thread workgroups: reset
                                        # after the step, counts per workgroup, cursors advanced by the launch behind the step)
    venv = make_vec_env(env, n, device=DEV, seed=31)
    venv.reset()
    mem, rmem = ReplayMemory(cap, 1, device=DEV), ConstraintReplayMemory(cap, 1, device=DEV)
    omem, ormem = co.OracleReplay(cap), co.OracleReplay(cap)
    stats = torch.zeros(10, dtype=torch.int64, device=DEV)
    sums = torch.zeros(2, dtype=torch.float64, device=DEV)
    ep_reward = torch.zeros(n, device=DEV)
    pos, t = venv.pos.cpu().numpy().copy(), np.zeros(n, np.int32)
    ref_stats = np.zeros(8, np.int64)
    ref_ep, ref_sums = np.zeros(n, np.float32), np.zeros(2)
    for k in range(5):
        obs_prev = venv.obs.cpu().numpy().copy()
        task = torch.as_tensor(rng.uniform(-1, 1, (n, 2)).astype(np.float32), device=DEV)
        real = torch.as_tensor(rng.uniform(-1.3, 1.3, (n, 2)).astype(np.float32), device=DEV)
        rec = torch.as_tensor((rng.uniform(size=n) < 0.3).astype(np.uint8), device=DEV)
        rc = lib.rrl_nav_step_push(
            co.ENV_KIND[env], n, _lib.ptr(venv.pos), _lib.ptr(venv.t), _lib.ptr(venv.obs), _lib.ptr(task),
            _lib.ptr(real), _lib.ptr(rec), 31, 0, _lib.ptr(venv.tick), 1, 100, 1, 2.5, 0, C.byref(mem._desc),
            C.byref(rmem._desc), *([_lib.ptr(venv.next_obs), _lib.ptr(venv.reward), _lib.ptr(venv.done),
                                    _lib.ptr(venv.constraint), _lib.ptr(venv.success), _lib.ptr(venv.ep_done)]
                                   if with_outputs else [None] * 6), _lib.ptr(stats),
            _lib.ptr(sums), _lib.ptr(ep_reward), _lib.current_stream())
        assert rc == 0
        ref = co.nav_step(env, pos, real.cpu().numpy(), t, seed=31, counter=1 + k, auto_reset=True)
        if with_outputs:
            assert np.array_equal(venv.next_obs.cpu().numpy(), ref["next_obs"])
            assert np.array_equal(venv.reward.cpu().numpy(), ref["reward"])
            for key in ("done", "constraint", "success", "ep_done"):
                assert np.array_equal(getattr(venv, key).cpu().numpy(), ref[key]), key
        else:
            assert not venv.next_obs.any() and not venv.reward.any() and not venv.done.any()
        pos, t = ref["pos"], ref["t"]
        mask = 1.0 - ref["done"].astype(np.float32)
        cons = ref["constraint"].astype(np.float32)
        omem.push(obs_prev, task.cpu().numpy(), ref["reward"] - 2.5 * cons, ref["next_obs"], mask)
        ormem.push(obs_prev, real.cpu().numpy(), cons, ref["next_obs"], mask)
        assert np.array_equal(venv.pos.cpu().numpy(), pos) and np.array_equal(venv.obs.cpu().numpy(), ref["obs"])
        r = rec.cpu().numpy().astype(bool)
        epd, c, s_ = ref["ep_done"].astype(bool), ref["constraint"].astype(bool), ref["success"].astype(bool)
        ref_stats += [n, epd.sum(), (epd & c).sum(), (epd & c & r).sum(), (epd & c & ~r).sum(), (epd & s_).sum(),
                      r.sum(), c.sum()]
        ref_ep += ref["reward"]
        ref_sums += [ref["reward"].astype(np.float64).sum(), ref_ep[epd].astype(np.float64).sum()]
        ref_ep[epd] = 0
    for got, want in ((mem, omem), (rmem, ormem)):
        assert int(got.state[0].item()) == want.pos and int(got.state[1].item()) == want.size == cap
        for a_, b_ in ((got.s, want.s), (got.a, want.a), (got.r, want.r), (got.s2, want.s2), (got.m, want.m)):
            assert np.array_equal(a_.cpu().numpy(), b_)
    filled = ormem.r != 0
    assert np.array_equal(rmem.pos_cnt.cpu().numpy()[:cap // 64 + 1][: (cap + 63) // 64],
                          np.add.reduceat(np.r_[filled, np.zeros((-cap) % 64, bool)].astype(np.int32),
                                          np.arange(0, cap, 64)))
    from test_replay_gpu import assert_count_tables       # chunk counts, super-chunk counts, slot masks (wrapped ring)
    assert_count_tables(rmem, ormem, cap)
    assert np.array_equal(stats.cpu().numpy()[:8], ref_stats)
    assert np.allclose(sums.cpu().numpy(), ref_sums, rtol=1e-9)
    assert np.allclose(ep_reward.cpu().numpy(), ref_ep, rtol=1e-6, atol=1e-4)
    assert int(venv.tick[0].item()) == 6


def test_fused_step_push_counts_when_a_workgroup_wraps_over_three_super_chunks():
    """Bandwidth-regime instance (n > 16384: second-level positive counts summed per workgroup) on a ring whose capacity
    is NOT a multiple of the 1024-slot super-chunk: the workgroup that straddles the wrap touches the last two super-chunks
    and super-chunk 0.  cap = 100424 = 98 * 1024 + 72; the third push starts at slot 80000, its workgroup 79 covers slots
    100224 .. 100423 (super-chunks 97 and 98) and 0 .. 55 (super-chunk 0).  Half of the envs start inside a wall and stay
    there (no auto-reset: `_next_state` returns the state unchanged), so about half of all rows are constraint positives.  Tables (chunk counts, super-chunk counts, slot masks) = a function of the stored rows."""
    import ctypes as C
    from recovery_rl_amd.replay_memory import ConstraintReplayMemory, ReplayMemory
    from test_replay_gpu import assert_count_tables
    lib = _lib.load()
    n, cap = 40000, 100424
    assert 0 < cap % 1024 < 255 and (cap - 80000) % 256 > cap % 1024
    rng = np.random.RandomState(11)
    venv = make_vec_env("navigation1", n, device=DEV, seed=5)
    venv.reset()
    start = np.stack([rng.uniform(-70, 40, n), rng.uniform(3.5, 6.5, n) * rng.choice([-1, 1], n)], 1)
    venv.pos.copy_(torch.as_tensor(start, device=DEV))
    venv.obs.copy_(torch.as_tensor(start.astype(np.float32), device=DEV))
    mem, rmem = ReplayMemory(cap, 1, device=DEV), ConstraintReplayMemory(cap, 1, device=DEV)
    ormem = co.OracleReplay(cap)
    stats = torch.zeros(10, dtype=torch.int64, device=DEV)
    sums = torch.zeros(2, dtype=torch.float64, device=DEV)
    ep_reward = torch.zeros(n, device=DEV)
    pos, t = start.copy(), np.zeros(n, np.int32)
    for k in range(6):
        obs_prev = venv.obs.cpu().numpy().copy()
        real = torch.as_tensor(rng.uniform(-1, 1, (n, 2)).astype(np.float32), device=DEV)
        rec = torch.zeros(n, dtype=torch.uint8, device=DEV)
        rc = lib.rrl_nav_step_push(0, n, _lib.ptr(venv.pos), _lib.ptr(venv.t), _lib.ptr(venv.obs), _lib.ptr(real),
                                   _lib.ptr(real), _lib.ptr(rec), 5, 0, _lib.ptr(venv.tick), 1, 100, 0, 0.0, 0,
                                   C.byref(mem._desc), C.byref(rmem._desc), None, None, None, None, None, None,
                                   _lib.ptr(stats), _lib.ptr(sums), _lib.ptr(ep_reward), _lib.current_stream())
        assert rc == 0
        ref = co.nav_step("navigation1", pos, real.cpu().numpy(), t, seed=5, counter=1 + k, auto_reset=False)
        pos, t = ref["pos"], ref["t"]
        ormem.push(obs_prev, real.cpu().numpy(), ref["constraint"].astype(np.float32), ref["next_obs"],
                   1.0 - ref["done"].astype(np.float32))
        assert np.array_equal(rmem.r.cpu().numpy(), ormem.r)
        assert_count_tables(rmem, ormem, cap)
    assert 0.1 < float((ormem.r != 0).mean()) < 0.9
    rmem.rebuild_pos_cnt()                  # the torch re-computation used when a checkpoint carries another layout
    assert_count_tables(rmem, ormem, cap)
