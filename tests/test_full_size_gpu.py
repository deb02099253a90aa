"""Full-size lock-step loops of BASELINE configs 3 and 4 through the driver (`Experiment.run`, 4096 envs, the reference's
command lines scripts/maze.sh:7 and scripts/navigation2.sh:14 + --num_envs 4096), ~100 iterations each: sampler error flags,
composition of the stratified draw, size of the recovery set, counters, the online ensemble re-fit of config 4."""
import numpy as np
import pytest
import torch

import arg_utils
from recovery_rl_amd.experiment import Experiment

pytestmark = pytest.mark.gpu
N = 4096


def run(argv, tmp_path, iters=100, log_every=25):
    cfg = arg_utils.get_args(argv + ["--cuda", "--num_envs", str(N), "--seed", "1", "--logdir", str(tmp_path), "--eval", "",
                                     "--log_every", str(log_every), "--num_steps", str(iters * N - 1)])
    exp = Experiment(cfg)
    hist = exp.run()                      # read_stats() at every log point raises if a sampler set its error flag
    assert hist[-1]["iteration"] >= iters and hist[-1]["env_steps"] == hist[-1]["iteration"] * N
    return exp, hist


def test_config3_maze_model_free_recovery_at_4096_envs(tmp_path):
    exp, hist = run(["--env-name", "maze", "--use_recovery", "--MF_recovery", "--gamma_safe", "0.5", "--eps_safe", "0.15",
                     "--pos_fraction", "0.3"], tmp_path)
    last, cfg = hist[-1], exp.exp_cfg
    assert exp.loop.graph is not None                                   # steady state replayed from one hipGraph
    assert int(exp.memory.state[3].item()) == 0 and int(exp.recovery_memory.state[3].item()) == 0    # error flags
    assert last["sac_updates"] >= 95 and last["qrisk_updates"] >= 95
    assert int(exp.agent.fast.critic.step[0].item()) == last["sac_updates"]          # device-side optimiser counters
    assert 0 < last["recovery_steps"] < last["env_steps"] and last["episodes"] > 0
    assert last["num_viols"] == last["viol_and_recovery"] + last["viol_and_no_recovery"]
    # the stratified draw of the Q_risk update (replay_memory.py:54-72): int(B * 0.3) = 76 positives first, then 180 negatives
    rm = exp.recovery_memory
    size = int(rm.state[1].item())
    assert size == len(rm) == exp.num_unsafe_transitions + last["iteration"] * N
    n_pos = int((rm.r[:size] != 0).sum().item())
    assert n_pos >= 76, "the positive class is not starved after 100 iterations"
    s, a, c, s2, m = rm.sample(cfg.batch_size, pos_fraction=cfg.pos_fraction)
    c = c.reshape(-1).cpu().numpy()
    assert c.shape == (256,) and (c[:76] == 1).all() and (c[76:] == 0).all()
    rm.check_error()
    # pos_cnt (three count levels) is consistent with the rows after 100 fused pushes
    n_chunks = (rm.capacity + 63) // 64
    filled = torch.zeros(n_chunks * 64, dtype=torch.int32, device=rm.r.device)
    filled[:size] = (rm.r[:size] != 0).to(torch.int32)
    assert torch.equal(rm.pos_cnt[:n_chunks], filled.view(-1, 64).sum(1).to(torch.int32))


@pytest.mark.parametrize("precision", ("f32", "f16x3"))
def test_config4_navigation2_model_based_recovery_at_4096_envs(tmp_path, precision):
    exp, hist = run(["--env-name", "navigation2", "--use_recovery", "--gamma_safe", "0.65", "--eps_safe", "0.2",
                     "--num_unsafe_transitions", "20000", "--plan_precision", precision], tmp_path)
    last, mpc = hist[-1], exp.recovery_policy
    assert int(exp.memory.state[3].item()) == 0 and int(exp.recovery_memory.state[3].item()) == 0
    assert mpc.fused is not None and mpc.fused.f16x3 == (precision == "f16x3") and mpc.has_been_trained
    assert mpc.device_count and mpc.last_count is not None              # planning set counted on the device
    assert exp.loop.graph is not None                                   # ... so the model-based iteration is ONE hipGraph too
    # the gate is the pre-trained Q_risk: some, not all, env-steps are under the recovery controller
    assert 0 < last["recovery_steps"] < 0.6 * last["env_steps"], last
    per_log = np.diff([0] + [h["recovery_steps"] for h in hist])
    assert (per_log >= 0).all() and per_log.sum() == last["recovery_steps"]
    assert last["sac_updates"] >= 95 and last["qrisk_updates"] >= 95 and last["episodes"] > 0
    # the online re-fit (experiment.py:464-480): every recovery_policy_update_freq * horizon iterations on ALL data
    rows = exp.num_unsafe_transitions + 100 * N
    assert mpc.train_in.shape[0] == rows and mpc.train_targs.shape[0] == rows
    for name in ("lin0_w", "lin1_w", "lin2_w", "lin3_w", "max_logvar", "min_logvar"):
        assert torch.isfinite(getattr(mpc.model, name)).all()
    with torch.no_grad():                  # the re-fitted ensemble predicts the nav2 dynamics s' = s + a (navigation2.py:98-103)
        s = torch.tensor([[-45.0, 1.0], [-10.0, -12.0]], device=exp.device)
        a = torch.tensor([[1.0, 0.0], [0.0, 1.0]], device=exp.device)
        mean, _ = mpc.model(torch.cat([s, a], 1).unsqueeze(0).expand(mpc.model.num_nets, -1, -1))
        assert (mean.mean(0) - a).abs().max() < 0.2


def test_config4_stays_safe_over_1600_iterations_at_4096_envs(tmp_path):
    """The behavioural guard of vectorisation rules 3 + 4 (demonstration share of the Q_risk batch; buffers that hold the run):
    config 4 (Navigation2, model-based recovery, scripts/navigation2.sh:14 + --num_envs 4096, f16x3 planner, 16 updates per
    iteration), seed 3, 1 675 iterations = 6.9 M env-steps, ~40 s.  With 1e6-row rings and one uniform draw (round 3's loop) this
    seed learns, stays clean for ~1 300 iterations and then ends in violation bursts: 760 violations, last window 55 %
    (profiles/round4_learning_vec4096_config4_controls.json; 1 587 with the share alone).  As shipped: none.  The reference's
    one-env run of this command line has 0 violations on every seed it was run for."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles"))
    import learning_vec4096 as lv
    r = lv.run(3, 16, 1650, config=4, precision="f16x3")
    assert r["iterations"] >= 1650 and r["env_steps"] >= 1650 * N
    worst = max(w["violation_rate"] for w in r["windows"])
    assert r["violations"] <= 60 and worst <= 0.01, (r["violations"], worst)        # shipped: 0 violations
    assert r["final_success_rate"] > 0.97 and r["first_window_with_90pct_success"]["iteration"] <= 300
    assert r["offline_violations"] > 1000


def test_config4_seed_1_has_no_violation_under_the_recovery_controller(tmp_path):
    """The behavioural guard of vectorisation rule 5 (an env's CEM warm start does not survive its episode, MPC.forget_plans):
    seed 1 of config 4 -- the seed whose violation bursts followed the planner's Philox key through rounds 4 and 5 (2 077 / 1 734
    / 0 violations with the f16x3 planner under keys 0 / 1 / 2, 1 825 / 1 700 of them WITH the recovery controller active; 0 / 5 175
    / 6 278 with the f32 one).  Round 6 found the mechanism (profiles/round6_config4_seed1/): an env plans once in hundreds of
    iterations, its warm start is a plan for another place and sits at the action bound, where CEM's variance clamp freezes it;
    inside a burst the critic rates a safe direction at 0.03 on every row under the recovery controller while the executed action
    is rated unsafe on 16-50 % of them.  With the rule: 43 / 0 / 0 (f16x3) and 0 / 20 (f32, keys 1 / 2) violations of ~125 k
    episodes, NONE under the recovery controller.  The reference's one-env run of this command line has 0 violations."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles"))
    import learning_vec4096 as lv
    r = lv.run(1, 16, 1650, config=4, precision="f16x3")
    assert r["iterations"] >= 1650 and r["episodes"] > 100000
    assert r["viol_and_recovery"] == 0, r["viol_and_recovery"]                    # the recovery controller never drives an env in
    assert r["violations"] <= 150, r["violations"]                                  # (43: gate misses just below eps_safe)
    assert max(w["violation_rate"] for w in r["windows"]) <= 0.06
    assert r["final_success_rate"] > 0.85


def test_plan_warm_start_rule_resets_exactly_the_ended_envs(tmp_path):
    """MPC.forget_plans (rule 5) in isolation + its switches: the rows whose episode ended go back to the mid-point sequence, the
    others keep their shifted solution; one env and --keep_plan_warm_start keep the reference's carry-over."""
    from recovery_rl_amd.experiment import Experiment
    base = ["--env-name", "navigation2", "--cuda", "--use_recovery", "--gamma_safe", "0.65", "--eps_safe", "0.2", "--logdir",
            str(tmp_path), "--num_unsafe_transitions", "2000", "--critic_safe_pretraining_steps", "10", "--seed", "2"]
    exp = Experiment(arg_utils.get_args(base + ["--num_envs", "64"]))
    mpc = exp.recovery_policy
    assert exp.loop.forget_plans and exp.vector_rules["plan_warm_start"].startswith("per episode")
    mpc.prev_sol.copy_(torch.rand_like(mpc.prev_sol) - 0.5)
    before = mpc.prev_sol.clone()
    ended = torch.zeros(64, dtype=torch.uint8, device=mpc.prev_sol.device)
    ended[[3, 17, 63]] = 1
    mpc.forget_plans(ended)
    keep = ended == 0
    assert torch.equal(mpc.prev_sol[keep], before[keep]) and bool((mpc.prev_sol[~keep] == 0).all())
    kept = Experiment(arg_utils.get_args(base + ["--num_envs", "64", "--keep_plan_warm_start"]))
    one = Experiment(arg_utils.get_args(base))
    assert not kept.loop.forget_plans and not one.loop.forget_plans
    assert kept.vector_rules["plan_warm_start"].startswith("kept") and one.vector_rules["plan_warm_start"].startswith("kept")


def test_bench_stage_table_prices_the_flops_the_iteration_model_states():
    """bench.roofline_stages re-issues the recorded launches of ONE config-2 iteration with their algorithmic FLOPs; their sum is
    what bench.iteration_flops (the figure `roofline_mlp` is priced with) states analytically -- to 1 % --, and every kernel of
    the timed graph appears in `by_kernel` with a roof."""
    import argparse
    import bench
    a = argparse.Namespace(env="navigation1", num_envs=N)
    st = bench.roofline_stages(a, torch.device("cuda:0"), 0.15)
    executed, survey = bench.iteration_flops(N, 1)
    assert abs(st["mlp_flops"] / executed - 1.0) < 0.01, (st["mlp_flops"], executed)
    assert survey > executed                                  # the reference's unused safety_critic(s, pi) is not executed
    names = {k["kernel"] for k in st["by_kernel"]}
    assert {"backward_pair_kernel<1>", "backward_pair_kernel<4>", "backward_pair_kernel<2>", "adam_multi_kernel",
            "step_push_kernel", "sample_group_kernel"} <= names
    assert all(0 < k["frac"] < 1 and k["bound"] in ("mfma", "hbm") for k in st["by_kernel"])
    assert st["by_kernel"][0]["us"] == max(k["us"] for k in st["by_kernel"])
