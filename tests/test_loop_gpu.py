"""GPU tests of the driver: lock-step loop bookkeeping, hipGraph replay, and the reference-order
single-env loop (run_stats.pkl schema, counters)."""
import os
import pickle

import numpy as np
import pytest
import torch

import arg_utils
from recovery_rl_amd.experiment import STAT_KEYS, Experiment, VectorLoop

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def make_cfg(tmp_path, extra=()):
    return arg_utils.get_args(["--env-name", "navigation1", "--cuda", "--hidden_size", "32", "--logdir",
                               str(tmp_path), "--seed", "3", "--num_unsafe_transitions", "2000",
                               "--critic_safe_pretraining_steps", "20"] + list(extra))


def test_vector_loop_bookkeeping_eager(tmp_path, capsys):
    cfg = make_cfg(tmp_path, ["--use_recovery", "--MF_recovery", "--gamma_safe", "0.8", "--eps_safe", "0.3",
                              "--num_envs", "128"])
    exp = Experiment(cfg)
    exp.pretrain_critic_recovery()
    assert len(exp.recovery_memory) == exp.num_unsafe_transitions > 1000
    assert exp.num_constraint_violations == int(exp.constraint_demo_data[2].sum().item())
    loop = exp.loop
    loop.step_outputs = True        # this test reads the env's per-step arrays after every step
    loop.start()
    ep_done_total = 0
    for k in range(30):
        loop.vector_step(do_update=len(exp.memory) > cfg.batch_size, random_actions=k < 2,
                         online_qrisk=True)
        ep_done_total += int(exp.env.ep_done.sum().item())
    st = loop.read_stats()
    assert st["env_steps"] == 30 * 128 == loop.total_numsteps
    assert len(exp.memory) == 30 * 128
    assert len(exp.recovery_memory) == exp.num_unsafe_transitions + 30 * 128
    assert st["episodes"] == ep_done_total
    assert st["num_viols"] == st["viol_and_recovery"] + st["viol_and_no_recovery"]
    assert st["sac_updates"] == 27 and st["qrisk_updates"] == 27        # len(memory) > 256 from iteration 3 on
    # the rows in memory are what the env produced on the last step
    m = exp.memory
    last = slice(29 * 128, 30 * 128)
    assert torch.equal(m.s2[last], exp.env.next_obs)
    assert torch.equal(m.r[last], exp.env.reward)
    assert torch.equal(m.m[last], 1.0 - exp.env.done.float())
    rm = exp.recovery_memory
    off = exp.num_unsafe_transitions
    assert torch.equal(rm.r[off + 29 * 128: off + 30 * 128], exp.env.constraint.float())


def test_graph_replay_advances_everything(tmp_path):
    cfg = make_cfg(tmp_path, ["--use_recovery", "--MF_recovery", "--gamma_safe", "0.8", "--eps_safe", "0.3",
                              "--num_envs", "256"])
    exp = Experiment(cfg)
    exp.pretrain_critic_recovery()
    loop = exp.loop
    loop.start()
    for k in range(3):
        loop.vector_step(do_update=k > 1, random_actions=True)
    loop.capture(online_qrisk=True)
    st0 = loop.read_stats()
    w0 = exp.agent.critic.linear2.weight.clone()
    tick0 = int(exp.env.tick[0].item())
    mtick0 = int(exp.memory.tick[0].item())
    pos0 = exp.env.pos.clone()
    size0 = int(exp.memory.state[1].item())
    for _ in range(10):
        loop.replay()
    torch.cuda.synchronize()
    st1 = loop.read_stats()
    assert st1["env_steps"] - st0["env_steps"] == 10 * 256
    assert st1["sac_updates"] - st0["sac_updates"] == 10
    assert st1["qrisk_updates"] - st0["qrisk_updates"] == 10
    assert int(exp.env.tick[0].item()) == tick0 + 10            # in-kernel RNG tick advanced by the graph
    assert int(exp.memory.tick[0].item()) == mtick0 + 10
    assert int(exp.memory.state[1].item()) == size0 + 10 * 256
    assert not torch.equal(exp.env.pos, pos0)
    assert not torch.equal(exp.agent.critic.linear2.weight, w0)
    assert torch.isfinite(exp.agent.critic.linear2.weight).all()
    exp.memory.check_error()
    exp.recovery_memory.check_error()
    # successive replays draw different noise: per-step positions are not a fixed increment
    p1 = exp.env.pos.clone()
    loop.replay()
    p2 = exp.env.pos.clone()
    loop.replay()
    p3 = exp.env.pos.clone()
    assert not torch.equal(p2 - p1, p3 - p2)


def test_reference_order_single_env_run(tmp_path, capsys):
    """num_envs == 1: same prints, counters and run_stats.pkl schema as the reference
    (experiment.py:483-490, 540-543; info keys env/navigation1.py:82-89 + 'recovery')."""
    cfg = make_cfg(tmp_path, ["--use_recovery", "--MF_recovery", "--gamma_safe", "0.8", "--eps_safe", "0.3",
                              "--num_eps", "11", "--start_steps", "20", "--batch_size", "64"])
    exp = Experiment(cfg)
    exp.run()
    out = capsys.readouterr().out
    assert "LOGDIR: " in out and "CRITIC SAFE UPDATE STEP:  0" in out
    assert "Number of Constraint Transitions: " in out and "Number of Constraint Violations: " in out
    assert "Episode: 1, total numsteps: " in out and "Num Violations So Far: " in out
    assert "Violations with Recovery: " in out and "Num Successes So Far: " in out
    assert "Avg. Reward: " in out                                    # eval after episode 10
    assert os.path.basename(exp.logdir).endswith("_SAC_navigation1_Gaussian_")
    args = pickle.load(open(os.path.join(exp.logdir, "args.pkl"), "rb"))
    assert args.env_name == "navigation1" and args.seed == 3
    data = pickle.load(open(os.path.join(exp.logdir, "run_stats.pkl"), "rb"))
    assert set(data) == {"test_stats", "train_stats"}
    assert len(data["train_stats"]) == 11 and len(data["test_stats"]) == 1
    steps = sum(len(ep) for ep in data["train_stats"])
    # the reference runs episode num_eps+1 before breaking and skips that last dump (experiment.py:375-377)
    assert steps < exp.total_numsteps <= steps + 100 and exp.total_numsteps == len(exp.memory)
    info = data["train_stats"][0][0]
    assert set(info) == {"constraint", "reward", "state", "next_state", "action", "success", "recovery"}
    assert info["state"].dtype == np.float64 and info["state"].shape == (2,)
    for ep in data["train_stats"]:
        assert 1 <= len(ep) <= 100
        # only the last step may be a violation (done on constraint)
        assert not any(s["constraint"] for s in ep[:-1])
    viols = sum(int(ep[-1]["constraint"]) for ep in data["train_stats"])
    assert viols <= exp.num_viols <= viols + 1
    assert exp.num_viols == exp.viol_and_recovery + exp.viol_and_no_recovery
    assert exp.updates == exp.total_numsteps - 65                    # one update per step once len(memory) > 64
    assert len(exp.recovery_memory) == exp.num_unsafe_transitions + exp.total_numsteps


def test_vectorized_run_stops_and_logs(tmp_path, capsys):
    cfg = make_cfg(tmp_path, ["--num_envs", "64", "--num_steps", "3000", "--log_every", "10"])
    exp = Experiment(cfg)
    hist = exp.run()
    assert hist[-1]["env_steps"] > 3000 and hist[-1]["env_steps"] % 64 == 0
    assert set(STAT_KEYS) <= set(hist[-1])
    data = pickle.load(open(os.path.join(exp.logdir, "run_stats.pkl"), "rb"))
    assert data["num_envs"] == 64 and len(data["vector_stats"]) == len(hist)
    assert exp.loop.graph is not None                                # steady state ran from the hipGraph
    assert "Num Violations So Far: " in capsys.readouterr().out


def test_model_based_recovery_runs_single_and_vectorised(tmp_path, capsys):
    """RRL-MB (scripts/navigation2.sh:14): PETS ensemble pre-trained on the demos, CEM planner queried
    where Q_risk > eps_safe, ensemble re-fitted online."""
    base = ["--env-name", "navigation2", "--cuda", "--hidden_size", "32", "--logdir", str(tmp_path),
            "--seed", "2", "--num_unsafe_transitions", "1500", "--critic_safe_pretraining_steps", "30",
            "--use_recovery", "--gamma_safe", "0.65", "--eps_safe", "0.05", "--batch_size", "64",
            "--start_steps", "10"]
    cfg = arg_utils.get_args(base + ["--num_eps", "1"])
    exp = Experiment(cfg)
    assert exp.recovery_policy is not None and exp.recovery_policy.value_func is exp.agent.safety_critic
    exp.run()
    assert exp.recovery_policy.has_been_trained
    n_demo = exp.num_unsafe_transitions
    assert exp.recovery_policy.train_in.shape[0] >= n_demo          # demos + online episodes
    data = pickle.load(open(os.path.join(exp.logdir, "run_stats.pkl"), "rb"))
    assert len(data["train_stats"]) == 1
    cfg = arg_utils.get_args(base + ["--num_envs", "16", "--num_steps", "1700", "--log_every", "20"])
    exp = Experiment(cfg)
    hist = exp.run()
    assert hist[-1]["env_steps"] > 1700
    assert exp.recovery_policy.train_in.shape[0] > exp.num_unsafe_transitions   # online re-fit happened
    assert exp.loop.graph is None                                    # MB planning is not graph-captured
    # the online rows are real transitions (state, clipped executed action) -> s' - s, not the zero rows an unwritten
    # prev_obs / action_clipped would give (the fused step kernel does not write those buffers itself)
    rp, n_demo = exp.recovery_policy, exp.num_unsafe_transitions
    online_in, online_targ = rp.train_in[n_demo:], rp.train_targs[n_demo:]
    assert online_in.shape[0] % 16 == 0 and online_in.shape[0] >= 16 * 100
    assert float(online_in[:, :2].abs().min(dim=1).values.max()) > 1.0          # states are around (-50 .. 0, .)
    assert float(online_in[:, 2:].abs().max()) <= 1.0 and float(online_in[:, 2:].abs().mean()) > 0.05
    free = online_targ.abs().sum(1) > 0                                           # rows that moved (not stuck in the box)
    resid = (online_targ - online_in[:, 2:])[free]
    assert float(resid.abs().max()) < 0.5 and 0.02 < float(resid.std()) < 0.08   # s' - s = a + 0.05 N(0, I)


def test_fused_step_keeps_state_and_clipped_action_for_the_refit(tmp_path):
    """One fused lock-step iteration of the model-based configuration: env.prev_obs / env.action_clipped (what the
    online ensemble re-fit reads, experiment.py:464-480) equal the pre-step observation and the clipped action."""
    cfg = arg_utils.get_args(["--env-name", "navigation2", "--cuda", "--hidden_size", "32", "--logdir", str(tmp_path),
                              "--seed", "2", "--num_unsafe_transitions", "600", "--use_recovery", "--gamma_safe",
                              "0.65", "--eps_safe", "0.2", "--num_envs", "64"])
    exp = Experiment(cfg)
    loop = exp.loop
    assert loop._can_fuse_step() and loop.recovery_policy is not None
    obs0 = loop.start().clone()
    act = torch.rand(64, 2, device=DEV) * 4 - 2
    loop.step_and_store(act, act.clone(), torch.zeros(64, dtype=torch.uint8, device=DEV))
    assert torch.equal(exp.env.prev_obs, obs0)
    assert torch.equal(exp.env.action_clipped, act.clamp(-1, 1))
    moved = exp.env.next_obs - obs0
    assert float((moved - act.clamp(-1, 1)).abs().max()) < 0.5


def test_maze_config3_vectorised_with_stratified_replay(tmp_path, capsys):
    """BASELINE config 3 at reduced size: Maze, model-free recovery, pos_fraction 0.3 (scripts/maze.sh:7)."""
    cfg = arg_utils.get_args(["--env-name", "maze", "--cuda", "--logdir", str(tmp_path), "--seed", "5",
                              "--use_recovery", "--MF_recovery", "--gamma_safe", "0.5", "--eps_safe", "0.15",
                              "--pos_fraction", "0.3", "--num_unsafe_transitions", "4000",
                              "--critic_safe_pretraining_steps", "20", "--num_envs", "256", "--num_steps", "8000",
                              "--log_every", "10"])
    exp = Experiment(cfg)
    assert exp.agent.fast is not None and exp.agent.safety_critic.pos_fraction == 0.3
    hist = exp.run()
    assert exp.num_constraint_violations > 77                       # enough positives for the stratified batches
    assert hist[-1]["env_steps"] > 8000 and hist[-1]["qrisk_updates"] > 0
    assert hist[-1]["episodes"] > 0
    exp.recovery_memory.check_error()
    exp.memory.check_error()
    assert exp.loop.graph is not None
    for p in exp.agent.safety_critic.safety_critic.parameters():
        assert torch.isfinite(p).all()


def test_learning_level_parity_with_the_reference_run(tmp_path, golden_dir, capsys):
    """scripts/navigation1.sh:7 (RRL-MF, seed 1, 400 episodes) end to end, against the statistics of the
    REFERENCE's own run of the same command line (tests/golden/ref_learning_nav1_seed1.json, produced by
    tests/golden/run_reference_training.py on the CPU: 397 successes, 0 violations, 19 924 env-steps).
    RNG streams differ, so the comparison is at the level plotting/plot_runs.py:214-235 defines."""
    import json
    ref = json.load(open(os.path.join(golden_dir, "ref_learning_nav1_seed1.json")))
    cfg = arg_utils.get_args(["--cuda", "--env-name", "navigation1", "--use_recovery", "--MF_recovery",
                              "--gamma_safe", "0.8", "--eps_safe", "0.3", "--logdir", str(tmp_path),
                              "--logdir_suffix", "RRL_MF", "--num_eps", "400", "--num_unsafe_transitions", "20000",
                              "--seed", "1", "--eval", ""])
    exp = Experiment(cfg)
    exp.run()
    capsys.readouterr()
    data = pickle.load(open(os.path.join(exp.logdir, "run_stats.pkl"), "rb"))["train_stats"]
    assert len(data) == ref["episodes"] == 400
    viol = sum(int(any(s["constraint"] for s in ep)) for ep in data)
    succ = sum(int(ep[-1]["reward"] > -4) for ep in data)
    steps = sum(len(ep) for ep in data)
    assert viol <= ref["total_violations"] + 4                      # reference: 0 violations
    assert succ >= ref["total_successes"] - 12                      # reference: 397 / 400 successes
    assert abs(steps - ref["env_steps"]) < 0.1 * ref["env_steps"]   # reference: 19 924 env-steps
    # the offline constraint data is distributed like the reference's (14 664 transitions, 903 violations)
    assert abs(exp.num_unsafe_transitions - ref["num_constraint_transitions"]) < 0.04 * ref["num_constraint_transitions"]
    assert abs(exp.num_constraint_violations - ref["num_constraint_violations_offline"]) < 0.15 * ref["num_constraint_violations_offline"]


def test_learning_at_the_headline_size_4096_envs(tmp_path, capsys):
    """BASELINE config 2 (scripts/navigation1.sh:7 + --num_envs 4096) with 16 updates per lock-step iteration: within
    250 iterations (1 M env-steps, 4 k grad steps, ~1 s) most episodes reach the goal and violations are rare -- the
    success / violation definitions of plotting/plot_runs.py:214-235 on the device-side counters."""
    cfg = arg_utils.get_args(["--cuda", "--env-name", "navigation1", "--use_recovery", "--MF_recovery", "--gamma_safe",
                              "0.8", "--eps_safe", "0.3", "--logdir", str(tmp_path), "--num_unsafe_transitions", "20000",
                              "--seed", "1", "--num_envs", "4096", "--updates_per_step", "16", "--num_steps",
                              str(4096 * 250), "--num_eps", "100000000", "--log_every", "25"])
    exp = Experiment(cfg)
    hist = exp.run()
    capsys.readouterr()
    assert exp.loop.graph is not None and exp.agent.fast.grouped
    last, mid = hist[-1], hist[len(hist) // 2]
    assert last["sac_updates"] >= 16 * 200 and last["qrisk_updates"] >= 16 * 200
    ep = last["episodes"] - mid["episodes"]
    succ = last["num_successes"] - mid["num_successes"]
    viol = last["num_viols"] - mid["num_viols"]
    assert ep > 5000 and succ / ep > 0.8, (ep, succ, viol)
    assert viol / ep < 0.1, (ep, succ, viol)
    exp.memory.check_error()
    exp.recovery_memory.check_error()


@pytest.mark.parametrize("hidden", (512, 40))
def test_loop_at_hidden_widths_outside_the_one_launch_stack_kernel(tmp_path, hidden):
    """--hidden_size 512 (> 256) and 40 (not a multiple of 16): the grouped launches and the fused stack forward do not
    cover these widths; the fused update path must fall back to its per-layer kernels (not assert), eagerly and from a graph,
    and stay equal to the autograd path."""
    argv = ["--use_recovery", "--MF_recovery", "--gamma_safe", "0.8", "--eps_safe", "0.3", "--num_envs", "64",
            "--hidden_size", str(hidden), "--batch_size", "64"]
    cfg = make_cfg(tmp_path, argv)
    exp = Experiment(cfg)
    assert exp.agent.fast is not None and not exp.agent.fast.grouped
    exp.pretrain_critic_recovery()
    loop = exp.loop
    loop.start()
    for k in range(4):
        loop.vector_step(do_update=len(exp.memory) > cfg.batch_size, random_actions=k < 2)
    loop.capture(online_qrisk=True)
    for _ in range(5):
        loop.replay()
    st = loop.read_stats()
    assert st["env_steps"] == loop.total_numsteps and st["sac_updates"] == st["qrisk_updates"] >= 5
    for net in (exp.agent.critic, exp.agent.policy, exp.agent.safety_critic.safety_critic, exp.agent.safety_critic.policy):
        assert all(torch.isfinite(p).all() for p in net.parameters())


@pytest.mark.parametrize("env_name,extra", [("navigation1", ["--gamma_safe", "0.8", "--eps_safe", "0.3"]),
                                            ("maze", ["--gamma_safe", "0.5", "--eps_safe", "0.15", "--pos_fraction", "0.3"])])
def test_compact_env_state_loop_equals_the_array_state_loop(tmp_path, env_name, extra):
    """The steady-state loop runs on the compact env state (u16 status word instead of step count + four flags, stored
    state taken from pos, no per-env outputs; rrl_*_step_push_x with `status`); with `step_outputs` it runs on the arrays.
    Same replay rows, env state, counters and networks, eagerly and from the graph; the decoded status word equals the
    arrays; switching representation in mid-run (as env.step / a checkpoint does) loses nothing."""
    loops = []
    for outputs in (False, True):
        cfg = arg_utils.get_args(["--env-name", env_name, "--cuda", "--hidden_size", "32", "--logdir", str(tmp_path),
                                  "--seed", "5", "--num_unsafe_transitions", "2000", "--critic_safe_pretraining_steps", "20",
                                  "--use_recovery", "--MF_recovery", "--num_envs", "192", "--batch_size", "64"] + extra)
        exp = Experiment(cfg)
        exp.pretrain_critic_recovery()
        loop = exp.loop
        loop.step_outputs = outputs
        loop.start()
        for k in range(40):           # > horizon / 3: episodes end by violation, success and (maze) the step limit
            loop.vector_step(do_update=len(exp.memory) > cfg.batch_size, random_actions=k < 3)
        if not outputs:               # a representation switch in mid-run: arrays, then back to the word
            assert exp.env._status_live
            exp.env.use_arrays()
            assert not exp.env._status_live
        loop.capture(online_qrisk=True)
        for _ in range(30):
            loop.replay()
        torch.cuda.synchronize()
        loops.append((exp, loop))
    (ea, la), (eb, lb) = loops
    assert ea.env._status_live and not eb.env._status_live
    ea.env.refresh_arrays()
    assert ea.env._status_live                                   # decoding does not switch the live representation
    assert torch.equal(ea.env.pos, eb.env.pos) and torch.equal(ea.env.obs, eb.env.obs)
    assert torch.equal(ea.env.t, eb.env.t) and int(ea.env.t.max()) > 3
    assert torch.equal(ea.env._flags, eb.env._flags) and int(eb.env.ep_done.sum()) >= 0
    assert torch.equal(la.stats, lb.stats) and torch.equal(la.reward_sums, lb.reward_sums)
    assert torch.equal(la.ep_reward, lb.ep_reward)
    for ma, mb in ((ea.memory, eb.memory), (ea.recovery_memory, eb.recovery_memory)):
        assert torch.equal(ma.state, mb.state)
        for x, y in ((ma.s, mb.s), (ma.a, mb.a), (ma.r, mb.r), (ma.s2, mb.s2), (ma.m, mb.m)):
            assert torch.equal(x, y)
    assert torch.equal(ea.recovery_memory.pos_cnt, eb.recovery_memory.pos_cnt)
    for name in ("critic", "policy", "qrisk", "recpolicy"):
        assert torch.equal(getattr(ea.agent.fast, name).flat, getattr(eb.agent.fast, name).flat), name
    st = la.read_stats()
    assert st["episodes"] > 0 and st["env_steps"] == 73 * 192     # 40 eager + 3 capture warm-up + 30 replays


def test_q_sampling_recovery_at_many_envs_is_captured_with_its_own_generator(tmp_path):
    """--Q_sampling_recovery with N > 1 (advisor, round 3): the fused update path is on, acting goes through the torch modules and
    draws the task policy's noise from the loop's own generator INSIDE the captured graph.  An unregistered generator made torch
    raise at capture ('Attempt to increase offset for a CUDA generator not in capture mode') -- or, without that check, replay the
    same noise for ever.  Registered with the graph (VectorLoop.capture), its offset advances per replay."""
    cfg = make_cfg(tmp_path, ["--use_recovery", "--Q_sampling_recovery", "--gamma_safe", "0.8", "--eps_safe", "0.3",
                              "--num_envs", "128"])
    exp = Experiment(cfg)
    exp.pretrain_critic_recovery()
    loop = exp.loop
    assert exp.agent.fast is not None and loop.n == 128
    loop.start()
    for k in range(3):
        loop.vector_step(do_update=k > 1, random_actions=True)
    loop.capture(online_qrisk=True)
    acts = []
    for _ in range(4):
        loop.replay()
        acts.append(loop._last_real_action.clone() if torch.is_tensor(loop._last_real_action) else None)
    torch.cuda.synchronize()
    st = loop.read_stats()
    assert st["env_steps"] >= 4 * 128 and st["sac_updates"] >= 4
    # the policy noise differs from replay to replay: the executed actions are not a replayed constant pattern
    assert all(a is not None for a in acts)
    assert not torch.equal(acts[1], acts[2]) and not torch.equal(acts[2], acts[3])
    assert torch.isfinite(exp.env.pos).all()


def test_replay_refuses_a_graph_whose_env_representation_changed(tmp_path):
    """The captured graph bakes in which of the env's two state representations it steps (u16 status word / step count + flag
    arrays).  An eager env.step() / reset() after the capture switches the env to the arrays: replay() must not keep stepping
    the stale status word (advisor, round 3)."""
    cfg = make_cfg(tmp_path, ["--use_recovery", "--MF_recovery", "--gamma_safe", "0.8", "--eps_safe", "0.3", "--num_envs", "256"])
    exp = Experiment(cfg)
    exp.pretrain_critic_recovery()
    loop = exp.loop
    loop.start()
    for k in range(3):
        loop.vector_step(do_update=k > 1, random_actions=True)
    loop.capture(online_qrisk=True)
    loop.replay()
    assert exp.env._status_live                     # the production loop runs on the compact state
    exp.env.reset()                                 # array API: decodes the status word, makes the arrays live
    assert not exp.env._status_live
    with pytest.raises(RuntimeError, match="representation changed"):
        loop.replay()
    loop.capture(online_qrisk=True)                 # a fresh capture picks the loop up again
    loop.replay()
    torch.cuda.synchronize()


def test_buffers_of_a_lockstep_run_hold_the_whole_run(tmp_path):
    """Vectorisation rule 4: --num_envs > 1 with a step budget above the default capacities -> both buffers are sized for the run
    (the reference's buffers never wrap within a run: replay_size == num_steps == 1e6); one env and --keep_replay_size keep the
    reference's capacities."""
    big = make_cfg(tmp_path, ["--use_recovery", "--MF_recovery", "--num_envs", "64", "--num_steps", "2000000"])
    exp = Experiment(big)
    assert exp.memory.capacity == 2000000 + 128 and exp.recovery_memory.capacity == 2000000 + 128 + 2000
    assert exp.memory.s.shape[0] == exp.memory.capacity
    keep = make_cfg(tmp_path, ["--use_recovery", "--MF_recovery", "--num_envs", "64", "--num_steps", "2000000", "--keep_replay_size"])
    exp = Experiment(keep)
    assert exp.memory.capacity == 1000000 and exp.recovery_memory.capacity == 1000000


@pytest.mark.parametrize("share,want", [(None, 0.5), ("0", 0.0), ("0.25", 0.25)])
def test_demo_share_rule_is_printed_recorded_and_can_be_switched_off(tmp_path, capsys, share, want):
    """Vectorisation rule 3 changes the safety critic's training distribution at N > 1: the run says so at start-up, writes
    the effective share into run_stats.pkl and the checkpoint, and `--demo_share 0` keeps the reference's single uniform
    draw over recovery_memory (replay_memory.py:54-72, qrisk.py:100-105) through the same captured loop."""
    cfg = make_cfg(tmp_path, ["--use_recovery", "--MF_recovery", "--gamma_safe", "0.8", "--eps_safe", "0.3", "--num_envs", "64",
                              "--num_steps", "1500", "--log_every", "10"] + (["--demo_share", share] if share else []))
    exp = Experiment(cfg)
    hist = exp.run()
    out = capsys.readouterr().out
    assert exp.loop.graph is not None and hist[-1]["qrisk_updates"] > 10
    assert (exp.agent.safety_critic.demo_share or 0.0) == want
    data = pickle.load(open(os.path.join(exp.logdir, "run_stats.pkl"), "rb"))
    pinned = exp.recovery_memory.pinned
    assert 1000 < pinned == exp.num_unsafe_transitions <= 2000
    assert data["vector_rules"]["demo_share"] == want and data["vector_rules"]["pinned_demonstrations"] == pinned
    assert data["vector_rules"]["replay_capacities"] == (exp.memory.capacity, exp.recovery_memory.capacity)
    ck = torch.load(os.path.join(exp.logdir, "checkpoint.pt"), map_location="cpu", weights_only=False)
    assert ck["extra"]["vector_rules"] == data["vector_rules"]
    if want:
        assert "Q_risk batch: %d of 256 rows from the %d pinned demonstrations" % (int(256 * want), pinned) in out
    else:
        assert "Q_risk batch: one uniform draw over the safety buffer" in out
        # the draw the graph replays is the reference's: uniform over [0, size), no split
        d, _ = exp.recovery_memory.draw_desc(256, demo_share=exp.agent.safety_critic.demo_share)
        assert d.stratified == 0 and (d.n_pos, d.n_neg) == (0, 256)


@pytest.mark.parametrize("updates_per_step", (1, 2))
def test_acting_forwards_riding_in_the_update_launches_change_no_bit(tmp_path, updates_per_step):
    """Round 6: the acting pass's task-policy forward rides in the Q_risk update's first forward launch and its Q_risk forward
    in the forward at the updated critic (VectorLoop.actor_rider, FastUpdater.qrisk_update_grouped: 17 -> 16 launches, 4096-row
    and 256-row members in one multi-row-tile launch).  Same kernels on the same inputs in another launch: the loop with the
    riders equals the loop without them in every replay row, env state, counter and network parameter -- eagerly and from the
    graph, at the bench's shape (hidden 256, batch 256, 4096 envs), also with two update pairs per iteration (the riders go
    with the last one)."""
    import bench
    from recovery_rl_amd import fast_update
    loops = []
    for carry in (True, False):
        cfg = arg_utils.get_args(bench.config_argv("navigation1", 7, 4096, updates_per_step) +
                                 ["--num_unsafe_transitions", "3000", "--logdir", str(tmp_path)])
        loop = bench.build_loop(cfg, torch.device(DEV), pretrain=10)
        loop.carry_actor = carry
        for _ in range(2):                # (the first acting pass sizes the iteration's noise fill: a launch of its own)
            loop.vector_step(True, False, True)
        tape = []
        fast_update.set_tape(tape)
        try:
            loop.vector_step(True, False, True)
        finally:
            fast_update.set_tape(None)
        launches = [op for op in tape if op[0] != "unsupported"]
        assert not [op for op in tape if op[0] == "unsupported"]
        assert len(launches) == (16 if carry else 17) + 14 * (updates_per_step - 1)
        rows = [[op[1][k].M for k in range(op[2])] for op in tape if op[0] == "forward"]
        if carry:
            assert rows[-4:] == [[4096, 256, 256], [256, 256], [4096, 256], [4096]]
        else:
            assert rows[-4:] == [[256, 256], [256], [4096, 4096], [4096]]
        for _ in range(3):
            loop.vector_step(True, False, True)
        loop.capture(online_qrisk=True)
        loop.advance(13)
        torch.cuda.synchronize()
        loops.append(loop)
    la, lb = loops
    assert torch.equal(la.env.pos, lb.env.pos) and torch.equal(la.env.obs, lb.env.obs)
    assert torch.equal(la.stats, lb.stats) and torch.equal(la.reward_sums, lb.reward_sums)
    for ma, mb in ((la.memory, lb.memory), (la.recovery_memory, lb.recovery_memory)):
        assert torch.equal(ma.state, mb.state)
        for x, y in ((ma.s, mb.s), (ma.a, mb.a), (ma.r, mb.r), (ma.s2, mb.s2), (ma.m, mb.m)):
            assert torch.equal(x, y)
    for name in ("critic", "critic_target", "policy", "qrisk", "qrisk_target", "recpolicy"):
        assert torch.equal(getattr(la.agent.fast, name).flat, getattr(lb.agent.fast, name).flat), name
    assert int(la.read_stats()["env_steps"]) == la.total_numsteps == lb.total_numsteps
