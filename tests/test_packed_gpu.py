"""Seed packing (recovery_rl_amd/packed.py, rrl_*_packed): S independent learners share every launch of the lock-step
iteration.  Every packed seed must end exactly where its solo run ends."""
import numpy as np
import pytest
import torch

import arg_utils
import bench
from recovery_rl_amd import _lib
from recovery_rl_amd.packed import PackedLoop

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def make_loop(env, seed, n_envs):
    cfg = arg_utils.get_args(bench.config_argv(env, seed, n_envs, 1) + ["--num_unsafe_transitions", "3000"])
    return bench.build_loop(cfg, DEV, pretrain=10)


def state_of(loop):
    f = loop.agent.fast
    env = loop.env
    env.refresh_arrays()
    out = {"pos": env.pos, "obs": env.obs, "t": env.t, "flags": env._flags, "tick": env.tick, "stats": loop.stats,
           "reward_sums": loop.reward_sums, "ep_reward": loop.ep_reward, "noise_tick": f.noise_tick}
    for name in ("critic", "critic_target", "policy", "qrisk", "qrisk_target", "recpolicy"):
        net = getattr(f, name)
        out[name + ".flat"], out[name + ".m"], out[name + ".v"], out[name + ".step"] = net.flat, net.m, net.v, net.step
    for tag, mem in (("mem", loop.memory), ("rmem", loop.recovery_memory)):
        for k in ("s", "a", "r", "s2", "m", "state", "tick"):
            out[tag + "." + k] = getattr(mem, k)
    out["rmem.pos_cnt"] = loop.recovery_memory.pos_cnt
    return {k: v.clone() for k, v in out.items()}


@pytest.mark.parametrize("env,S,n_envs", [("navigation1", 3, 256), ("maze", 2, 384), ("navigation1", 5, 4096)])
def test_every_packed_seed_equals_its_solo_run(env, S, n_envs):
    K = 25
    packed = PackedLoop([make_loop(env, 1 + s, n_envs) for s in range(S)])
    done = packed.capture()
    assert [op[0] for op in packed.tapes[0]].count("forward") >= 5 and len(packed.stages) <= 22
    for _ in range(K):
        packed.replay()
    torch.cuda.synchronize()
    got = [state_of(l) for l in packed.loops]
    stats = packed.read_stats()
    del packed
    for s in range(S):
        solo = make_loop(env, 1 + s, n_envs)
        for _ in range(done + K):
            solo.vector_step(True, False, True)
        torch.cuda.synchronize()
        want = state_of(solo)
        for k in want:
            assert torch.equal(got[s][k], want[k]), (env, s, k)
        st = solo.read_stats()
        assert st == stats[s] and st["sac_updates"] == done + K
        assert int(solo.agent.fast.critic.step[0].item()) == done + K
    # the seeds are different learners
    assert not torch.equal(got[0]["critic.flat"], got[1]["critic.flat"]) and not torch.equal(got[0]["pos"], got[1]["pos"])


def test_packed_entry_points_reject_what_they_cannot_pack():
    lib = _lib.load()
    assert lib.rrl_sample_multi_packed(0, None, None) != 0
    assert lib.rrl_mlp3_forward_multi_packed(17, None, None, None) != 0
    assert lib.rrl_nav_step_push_packed(2, 5, None, None) != 0
