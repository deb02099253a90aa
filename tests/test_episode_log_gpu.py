"""rrl_episode_log_append on MI355X: bit-exact against the sequential checker (integer fields and the f64
returns), hipGraph replay, overflow, and the run_stats.pkl written by the lock-step driver."""
import os
import pickle

import numpy as np
import pytest
import torch

import arg_utils
from oracle.log_oracle import EpisodeLogOracle
from recovery_rl_amd import _lib
from recovery_rl_amd.episode_log import (EPISODE_DTYPE, FLAG_CONSTRAINT, FLAG_SUCCESS, EpisodeLog, episode_metrics)
from recovery_rl_amd.experiment import Experiment

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _inputs(rng, n):
    reward = (-rng.uniform(0, 70, n)).astype(np.float32)
    cons = (rng.rand(n) < 0.05).astype(np.uint8)
    succ = (reward > -4).astype(np.uint8)
    done = ((rng.rand(n) < 0.08) | (cons > 0)).astype(np.uint8)
    rec = (rng.rand(n) < 0.2).astype(np.uint8)
    return reward, cons, succ, done, rec


@pytest.mark.parametrize("n,with_recovery", [(1, True), (1000, True), (5000, False)])
def test_episode_log_equals_checker(n, with_recovery):
    rng = np.random.RandomState(n)
    log = EpisodeLog(n, n * 20, DEV)
    chk = EpisodeLogOracle(n)
    drained = []
    for it in range(40):
        r, c, s, d, rec = _inputs(rng, n)
        t = [torch.as_tensor(x, device=DEV) for x in (r, c, s, d, rec)]
        log.append(t[0], t[1], t[2], t[3], t[4] if with_recovery else None)
        chk.append(r, c, s, d, rec if with_recovery else None)
        if it % 20 == 19:
            drained.append(log.drain())
    got = np.concatenate(drained)
    want = np.array(chk.records, dtype=EPISODE_DTYPE)
    assert len(got) == len(want) > 0
    for name in EPISODE_DTYPE.names:
        np.testing.assert_array_equal(got[name], want[name], err_msg=name)
    # open episodes: the accumulators equal the checker's
    np.testing.assert_array_equal(log.ep_len.cpu().numpy(), chk.len)
    np.testing.assert_array_equal(log.ep_ret.cpu().numpy(), chk.ret)
    assert int(log.state[1].item()) == 40


def test_episode_log_in_a_replayed_graph():
    n = 300
    rng = np.random.RandomState(5)
    r, c, s, d, rec = _inputs(rng, n)
    t = [torch.as_tensor(x, device=DEV) for x in (r, c, s, d, rec)]
    log = EpisodeLog(n, n * 8, DEV)
    chk = EpisodeLogOracle(n)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        log.append(*t)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        log.append(*t)
    for _ in range(6):
        g.replay()
    torch.cuda.synchronize()
    for _ in range(7):
        chk.append(r, c, s, d, rec)
    got, want = log.drain(), np.array(chk.records, dtype=EPISODE_DTYPE)
    for name in EPISODE_DTYPE.names:
        np.testing.assert_array_equal(got[name], want[name], err_msg=name)


def test_episode_log_overflow_is_reported():
    n = 64
    log = EpisodeLog(n, 100, DEV)
    ones = torch.ones(n, dtype=torch.uint8, device=DEV)
    r = torch.zeros(n, device=DEV)
    log.append(r, ones, ones, ones)
    assert len(log.drain()) == 64
    log.append(r, ones, ones, ones)
    log.append(r, ones, ones, ones)
    with pytest.raises(_lib.RRLError, match="overflow"):
        log.drain()


def test_lockstep_driver_writes_episode_table(tmp_path):
    cfg = arg_utils.get_args(["--env-name", "navigation1", "--cuda", "--hidden_size", "32", "--logdir", str(tmp_path),
                              "--seed", "3", "--num_unsafe_transitions", "2000", "--critic_safe_pretraining_steps",
                              "20", "--use_recovery", "--MF_recovery", "--gamma_safe", "0.8", "--eps_safe", "0.3",
                              "--num_envs", "64", "--num_eps", "150", "--log_every", "25"])
    exp = Experiment(cfg)
    exp.run()
    data = pickle.load(open(os.path.join(exp.logdir, "run_stats.pkl"), "rb"))
    rec = data["episode_stats"]
    last = data["vector_stats"][-1]
    assert len(rec) == last["episodes"] > 150
    assert int((rec["flags"] & FLAG_CONSTRAINT > 0).sum()) == last["num_viols"]
    assert int((rec["flags"] & FLAG_SUCCESS > 0).sum()) == last["num_successes"]
    assert int(rec["constraint_steps"].sum()) <= last["constraint_steps"]
    assert rec["length"].min() >= 1 and rec["length"].max() <= 100
    assert rec["length"].sum() <= last["env_steps"]
    np.testing.assert_allclose(rec["ret"].sum(), last["episode_return_sum"], rtol=1e-5)
    # nav1 success is judged on the reward of the last step (plot_runs.py:231 / navigation1.py:80)
    np.testing.assert_array_equal(rec["last_reward"] > -4, rec["flags"] & FLAG_SUCCESS > 0)
    raw = np.fromfile(os.path.join(exp.logdir, "episode_stats.bin"), dtype=EPISODE_DTYPE)
    for name in EPISODE_DTYPE.names:
        np.testing.assert_array_equal(raw[name], rec[name])
    m = episode_metrics(data, "navigation1")
    assert m["task_successes"][-1] == last["num_successes"]
    assert m["train_violations"][-1] == int((rec["constraint_steps"] > 0).sum())


def test_info_envs_writes_the_reference_per_step_schema(tmp_path):
    """--info_envs K: run_stats.pkl `train_stats` holds the first K envs' episodes as lists of the reference's step dicts
    (env/navigation1.py:82-89 + `recovery`, experiment.py:421-427); every episode equals its row of the per-episode table and
    its transitions chain (next_state of step t = state of step t+1; the dynamics of navigation1.py:103-106 hold step by step)."""
    K = 4
    cfg = arg_utils.get_args(["--env-name", "navigation1", "--cuda", "--logdir", str(tmp_path),
                              "--seed", "5", "--num_unsafe_transitions", "2000", "--critic_safe_pretraining_steps",
                              "20", "--use_recovery", "--MF_recovery", "--gamma_safe", "0.8", "--eps_safe", "0.3",
                              "--num_envs", "64", "--num_eps", "400", "--log_every", "25", "--start_steps", "640",
                              "--info_envs", str(K)])
    exp = Experiment(cfg)
    exp.run()
    data = pickle.load(open(os.path.join(exp.logdir, "run_stats.pkl"), "rb"))
    assert data["info_envs"] == K and data["test_stats"] == []
    ts, rec = data["train_stats"], data["episode_stats"]
    rec = rec[rec["env"] < K]
    assert len(ts) == len(rec) > 2 * K                       # several episodes per env, past the graph capture
    from recovery_rl_amd.episode_log import records_from_train_stats
    mine = records_from_train_stats(ts)
    for name in ("length", "constraint_steps", "recovery_steps", "flags"):
        np.testing.assert_array_equal(mine[name], rec[name], err_msg=name)
    np.testing.assert_allclose(mine["ret"], rec["ret"], rtol=1e-6)
    np.testing.assert_allclose(mine["last_reward"], rec["last_reward"], rtol=1e-6)
    assert any(s["recovery"] for ep in ts for s in ep)
    for ep in ts:
        assert set(ep[0]) == {"constraint", "reward", "state", "next_state", "action", "success", "recovery"}
        for a, b in zip(ep[:-1], ep[1:]):
            np.testing.assert_array_equal(a["next_state"], b["state"])
        for s in ep:
            assert np.abs(s["action"]).max() <= 1.0
            # noise scale 0.05 (navigation1.py:24,103-106): the step is the action plus a small perturbation
            assert np.abs(s["next_state"] - s["state"] - s["action"]).max() < 0.5
            assert s["reward"] == pytest.approx(-np.linalg.norm(s["state"] - np.array([0.0, 0.0])), rel=1e-5)
        assert not any(s["constraint"] for s in ep[:-1])      # an episode ends at its first violation
    # plotting/plot_runs.py:147-235 reads exactly these two keys per step
    m = episode_metrics({"train_stats": ts}, "navigation1")
    assert len(m["ep_lengths"]) == len(ts) and m["train_violations"][-1] == int((rec["constraint_steps"] > 0).sum())


def test_info_envs_on_the_unfused_step_of_the_maze(tmp_path):
    """The same stream when the iteration is NOT the fused step + push launch (--add_both_transitions keeps env.step() and
    the masked second push, experiment.py:446-448) and on the other env family (scripts/maze.sh:7)."""
    K = 3
    cfg = arg_utils.get_args(["--env-name", "maze", "--cuda", "--logdir", str(tmp_path), "--seed", "2",
                              "--use_recovery", "--MF_recovery", "--gamma_safe", "0.5", "--eps_safe", "0.15",
                              "--pos_fraction", "0.3", "--num_unsafe_transitions", "2000", "--critic_safe_pretraining_steps",
                              "20", "--add_both_transitions", "--num_envs", "32", "--num_eps", "100", "--log_every", "20",
                              "--start_steps", "320", "--info_envs", str(K)])
    exp = Experiment(cfg)
    exp.run()
    data = pickle.load(open(os.path.join(exp.logdir, "run_stats.pkl"), "rb"))
    from recovery_rl_amd.episode_log import records_from_train_stats
    ts, rec = data["train_stats"], data["episode_stats"]
    rec = rec[rec["env"] < K]
    assert len(ts) == len(rec) >= K
    mine = records_from_train_stats(ts)
    for name in ("length", "constraint_steps", "recovery_steps", "flags"):
        np.testing.assert_array_equal(mine[name], rec[name], err_msg=name)
    np.testing.assert_allclose(mine["ret"], rec["ret"], rtol=1e-6)
    for ep in ts:
        for a, b in zip(ep[:-1], ep[1:]):
            np.testing.assert_array_equal(a["next_state"], b["state"])


@pytest.mark.parametrize("env_name,extra,n", [("navigation1", [], 768), ("maze", ["--pos_fraction=0.3"], 768),
                                              ("navigation1", [], 20480), ("navigation1", [], 300000)])
def test_log_advanced_by_the_fused_step_equals_the_stand_alone_launch(env_name, extra, n):
    """rrl_step_push_t.log_*: the env-step launch advances the episode table from its registers (what the lock-step driver
    runs: compact env state, no per-env outputs).  Against the same loop on the array path with the stand-alone
    rrl_episode_log_append fed from the step's outputs: same records, same accumulators, same iteration counter -- eagerly
    and through a replayed graph.  n = 768: the latency variant of the step kernel (slots reserved per wave, early);
    20480 / 300000: its two bandwidth variants (256- / 1024-thread workgroups, slots reserved once per workgroup, cursors advanced
    by the one-thread launch behind the step)."""
    import bench
    loops, logs = [], []
    for fused in (True, False):
        cfg = arg_utils.get_args(["--env-name", env_name, "--cuda", "--use_recovery", "--MF_recovery", "--num_envs", str(n),
                                  "--seed", "9"] + extra)
        loop = bench.build_loop(cfg, torch.device(DEV), pretrain=5)
        log = EpisodeLog(n, n * (40 if n < 10000 else 24), DEV)
        if fused:
            loop.episode_log = log
        else:
            loop.step_outputs = True
        loops.append(loop)
        logs.append(log)
    for phase in range(2):
        for loop, log, fused in zip(loops, logs, (True, False)):
            def one(replay):
                if replay:
                    loop.replay()
                else:
                    loop.vector_step(True, False, True)
                if not fused:       # the old path: a launch of its own, fed from the step's per-env outputs
                    env = loop.env
                    log.append(env.reward, env.constraint, env.success, env.ep_done, loop._last_recovery)
            if phase == 0:
                for _ in range(12):
                    one(False)
            else:
                if fused:
                    loop.capture(online_qrisk=True)
                    for _ in range(9):
                        one(True)
                else:
                    for _ in range(12):
                        one(False)
        torch.cuda.synchronize()
        a, b = logs
        for name in ("ep_len", "ep_ret", "ep_viol", "ep_rec"):
            assert torch.equal(getattr(a, name), getattr(b, name)), (phase, name)
        assert int(a.state[1].item()) == int(b.state[1].item()) == 12 * (phase + 1)
        got, want = a.drain(), b.drain()
        assert len(got) == len(want) > 0
        for name in EPISODE_DTYPE.names:
            np.testing.assert_array_equal(got[name], want[name], err_msg=name)
    assert loops[0].env._status_live and not loops[1].env._status_live
    assert loops[0].read_stats() == loops[1].read_stats()
