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


@pytest.mark.parametrize("env,S,n_envs", [("navigation1", 3, 256), ("maze", 2, 384), ("navigation1", 5, 4096),
                                           ("navigation1", 11, 128), ("navigation1", 14, 128),    # > 8 seeds: linear / two seeds per XCD
                                           ("navigation1", 2, 2048), ("navigation1", 9, 1056), ("navigation1", 16, 1056)])   # acting pass in the stream form: ragged blocks, > 8 seeds
def test_every_packed_seed_equals_its_solo_run(env, S, n_envs):
    K = 25
    packed = PackedLoop([make_loop(env, 1 + s, n_envs) for s in range(S)])
    done = packed.capture()
    kinds = [op[0] for op in packed.tapes[0]]
    # the launch list of the solo graph: 16 launches (every head + hidden backward pair is one launch: tile form up to two
    # seeds, 32 x 64 blocks beyond)
    # (16 since round 6: two of the acting pass's three forwards ride in the Q_risk update's forward launches)
    assert kinds.count("forward") >= 5 and kinds.count("pair_bwd") == 5 and packed.launches == (16 if S <= 8 else 21)
    packed.replay()
    packed.advance(K - 1)            # six four-iteration graphs (--graph_iterations 4) + single iterations for the rest
    assert packed.graph_many_iters == 4 and packed.graph_many is not None
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


def test_seeds_per_gpu_flag_runs_packed_experiments(tmp_path, capsys):
    """`rrl_main --seeds_per_gpu 3`: three experiments (seeds 4, 5, 6: own log directories, pre-training, buffers) advanced
    by one shared graph; each one's counters equal the solo run of the same seed at the same iteration."""
    import os
    import pickle
    from recovery_rl_amd.experiment import Experiment, run_packed
    argv = ["--env-name", "navigation1", "--cuda", "--use_recovery", "--MF_recovery", "--gamma_safe", "0.8", "--eps_safe",
            "0.3", "--num_unsafe_transitions", "3000", "--critic_safe_pretraining_steps", "30", "--num_envs", "128",
            "--seed", "4", "--log_every", "20", "--num_eps", "100000", "--num_steps", str(128 * 60 - 1)]
    cfg = arg_utils.get_args(argv + ["--seeds_per_gpu", "3", "--logdir", str(tmp_path / "packed")])
    hists = run_packed(cfg)
    assert len(hists) == 3 and all(h[-1]["iteration"] == 60 and h[-1]["env_steps"] == 60 * 128 for h in hists)
    dirs = sorted(os.listdir(tmp_path / "packed"))
    assert len(dirs) == 3 and dirs[0].endswith("_seed4") and dirs[2].endswith("_seed6")
    saved = pickle.load(open(tmp_path / "packed" / dirs[1] / "run_stats.pkl", "rb"))
    assert saved["seeds_per_gpu"] == 3 and saved["vector_stats"][-1] == hists[1][-1]
    assert hists[0][-1]["sac_updates"] > 40 and hists[0][-1] != hists[1][-1]
    # the solo run of the middle seed, iteration by iteration through the same phases
    solo_cfg = arg_utils.get_args(argv[:-8] + ["--seed", "5", "--log_every", "20", "--num_eps", "100000", "--num_steps",
                                               str(128 * 60 - 1), "--logdir", str(tmp_path / "solo")])
    solo = Experiment(solo_cfg)
    solo.pretrain_critic_recovery()
    loop = solo.loop
    loop.start()
    for _ in range(60):
        loop.vector_step(do_update=len(solo.memory) > solo_cfg.batch_size,
                         random_actions=solo_cfg.start_steps > loop.total_numsteps, online_qrisk=solo.online_qrisk_enabled())
    want = loop.read_stats()
    got = {k: v for k, v in hists[1][-1].items() if k != "iteration"}
    assert got == want


def test_packed_seeds_write_the_files_of_their_solo_runs(tmp_path, capsys):
    """Per-episode table (run_stats.pkl `episode_stats`, episode_stats.bin) and, with --info_envs, the reference-schema
    per-step `train_stats` of a packed seed equal what the solo lock-step run of that seed writes, record for record."""
    import os
    import pickle
    import numpy as np
    from recovery_rl_amd.episode_log import EPISODE_DTYPE
    from recovery_rl_amd.experiment import Experiment, run_packed
    argv = ["--env-name", "navigation1", "--cuda", "--use_recovery", "--MF_recovery", "--gamma_safe", "0.8", "--eps_safe",
            "0.3", "--num_unsafe_transitions", "3000", "--critic_safe_pretraining_steps", "30", "--num_envs", "64",
            "--log_every", "25", "--num_eps", "100000", "--num_steps", str(64 * 100 - 1), "--start_steps", "640",
            "--info_envs", "2"]
    run_packed(arg_utils.get_args(argv + ["--seed", "7", "--seeds_per_gpu", "2", "--logdir", str(tmp_path / "packed")]))
    dirs = sorted(os.listdir(tmp_path / "packed"))
    solo = Experiment(arg_utils.get_args(argv + ["--seed", "8", "--logdir", str(tmp_path / "solo")]))
    solo.run()
    want = pickle.load(open(os.path.join(solo.logdir, "run_stats.pkl"), "rb"))
    got = pickle.load(open(tmp_path / "packed" / dirs[1] / "run_stats.pkl", "rb"))
    assert dirs[1].endswith("_seed8") and got["seeds_per_gpu"] == 2 and got["info_envs"] == 2
    assert len(want["episode_stats"]) > 100
    for name in EPISODE_DTYPE.names:
        np.testing.assert_array_equal(got["episode_stats"][name], want["episode_stats"][name], err_msg=name)
    raw = np.fromfile(tmp_path / "packed" / dirs[1] / "episode_stats.bin", dtype=EPISODE_DTYPE)
    np.testing.assert_array_equal(raw["ret"], want["episode_stats"]["ret"])
    assert len(got["train_stats"]) == len(want["train_stats"]) > 4
    for ep_a, ep_b in zip(got["train_stats"], want["train_stats"]):
        assert len(ep_a) == len(ep_b)
        for a, b in zip(ep_a, ep_b):
            assert a["reward"] == b["reward"] and a["constraint"] == b["constraint"] and a["recovery"] == b["recovery"]
            np.testing.assert_array_equal(a["state"], b["state"])
            np.testing.assert_array_equal(a["action"], b["action"])
            np.testing.assert_array_equal(a["next_state"], b["next_state"])
    assert {k: v for k, v in got["vector_stats"][-1].items()} == {k: v for k, v in want["vector_stats"][-1].items()}


def test_packed_launch_that_would_build_its_argument_block_inside_a_capture_says_so():
    """A packed launch builds the device copy of its argument blocks the first time it sees them (hipMalloc + copy).  Inside a
    stream capture that would invalidate the graph with an opaque launch error: the entry points return RRL_ECAPTURE instead
    (PackedLoop.capture launches every stage eagerly first, so the captured launches only look their blocks up); afterwards
    the regular capture works, and close() frees the cached blocks."""
    packed = PackedLoop([make_loop("navigation1", 21 + s, 128) for s in range(2)])
    for _ in range(2):                       # the first iterations size the noise buffer and build the acting workspace
        for loop in packed.loops:
            loop.vector_step(True, False, True)
    packed.record()
    g = torch.cuda.CUDAGraph()
    with pytest.raises(_lib.RRLError, match="capturing"):
        with torch.cuda.graph(g):
            packed.launch()
    del g
    torch.cuda.synchronize()
    packed.capture()
    for _ in range(3):
        packed.replay()
    torch.cuda.synchronize()
    assert packed.loops[0].read_stats()["env_steps"] > 0
    assert packed.close() >= 7          # one cached block per packed stage kind at least


def test_iterations_from_the_many_iteration_graph_equal_single_replays():
    """VectorLoop.advance(n): --graph_iterations iterations per hipGraph wherever they fit, single-iteration graphs for the rest --
    the same launches in the same order as n replay() calls, so every tensor of the run is the same."""
    a, b = make_loop("navigation1", 3, 512), make_loop("navigation1", 3, 512)
    assert a.capture(online_qrisk=True) == b.capture(online_qrisk=True, iters=1)
    assert a.graph_many_iters == 4 and a.graph_many is not None and b.graph_many is None
    a.advance(11)                    # 4 + 4 + 1 + 1 + 1
    a.advance(2)
    for _ in range(13):
        b.replay()
    torch.cuda.synchronize()
    sa, sb = state_of(a), state_of(b)
    for k in sb:
        assert torch.equal(sa[k], sb[k]), k
    assert a.read_stats() == b.read_stats() and a.total_numsteps == b.total_numsteps and a.updates == b.updates
    assert len(a.memory) == len(b.memory) and len(a.recovery_memory) == len(b.recovery_memory)
    assert a.host_updates == b.host_updates
