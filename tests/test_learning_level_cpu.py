"""Learning-level parity (SURVEY 8d "parity gates"): cumulative task successes / constraint violations over 400
episodes of scripts/navigation1.sh:7, this stack (profiles/round2_learning_seeds.json = the round-2 build, and round 1's file; 8 seeds on one MI355X) against
the REFERENCE's own runs of the same command line (tests/golden/ref_learning_nav1_seed*.json, produced by
tests/golden/run_reference_training.py on the CPU of the build container).  RNG streams differ, so the comparison
is between the seed-to-seed distributions, with the definitions of plotting/plot_runs.py:214-235."""
import glob
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


import pytest

RUNS = ("round2_learning_seeds.json", "round1_learning_seeds.json")


def _load(name=RUNS[0]):
    ref = [json.load(open(p)) for p in sorted(glob.glob(os.path.join(HERE, "golden", "ref_learning_nav1_seed*.json")))]
    mine = json.load(open(os.path.join(HERE, "..", "profiles", name)))["runs"]
    return ref, mine


@pytest.mark.parametrize("name", RUNS)
def test_success_and_violation_counts_lie_in_the_reference_spread(name):
    ref, mine = _load(name)
    assert len(ref) >= 3 and len(mine) == 8
    for key, slack in (("total_successes", 6), ("total_violations", 2), ("env_steps", 800)):
        r = np.array([x[key] for x in ref], dtype=np.float64)
        m = np.array([x[key] for x in mine], dtype=np.float64)
        spread = max(r.std(), m.std(), 1.0)
        # means within two pooled standard deviations (+ a small absolute slack for the nearly constant counts)
        assert abs(r.mean() - m.mean()) <= 2.0 * spread + slack, (key, r.tolist(), m.tolist())
        assert m.min() >= r.min() - 3 * spread - slack and m.max() <= r.max() + 3 * spread + slack, (key, r, m)


@pytest.mark.parametrize("name", RUNS)
def test_learning_curves_have_the_reference_shape(name):
    """Cumulative successes after 100 / 200 / 400 episodes (the y axis of plot_runs.py PLOT_TYPE 'success')."""
    ref, mine = _load(name)
    for upto in (100, 200, 400):
        r = np.array([sum(x["successes"][:upto]) for x in ref], dtype=np.float64)
        m = np.array([sum(x["successes"][:upto]) for x in mine], dtype=np.float64)
        assert abs(r.mean() - m.mean()) <= 2.0 * max(r.std(), m.std(), 1.0) + 0.03 * upto, (upto, r, m)
    # offline constraint data: same generator statistics
    r = np.array([x["num_constraint_transitions"] for x in ref], dtype=np.float64)
    m = np.array([x["num_constraint_transitions"] for x in mine], dtype=np.float64)
    assert abs(r.mean() - m.mean()) < 0.03 * r.mean()


def test_navigation2_runs_are_in_the_range_of_the_reference_runs():
    """scripts/navigation2.sh:7 (model-free recovery), seeds 1..8: this stack's one-env runs (`python
    profiles/learning_other_configs.py nav2_mf <seeds>` on one MI355X: profiles/round1_learning_other_configs.jsonl seeds 1-4,
    round4_learning_nav2_mf_one_env.jsonl seeds 5-8) next to the reference's own (tests/golden/ref_learning_nav2_seed*, recorded
    by tests/golden/run_reference_training.py).  This configuration has a large seed-to-seed spread on BOTH stacks (reference
    238 .. 397 successes, this stack 106 .. 391 over seeds 1-4; seed 4 is the worst run of both), so the distributions are
    compared, not single runs."""
    ref = [json.load(open(p)) for p in sorted(glob.glob(os.path.join(HERE, "golden", "ref_learning_nav2_seed*.json")))]
    mine = []
    for name in ("round1_learning_other_configs.jsonl", "round4_learning_nav2_mf_one_env.jsonl"):
        path = os.path.join(HERE, "..", "profiles", name)
        if os.path.exists(path):
            mine += [json.loads(line) for line in open(path) if line.strip()]
    mine = [m for m in mine if m["config"] == "nav2_mf"]
    assert len(ref) >= 4 and len(mine) >= 4
    seeds = sorted(set(r["seed"] for r in ref) & set(m.get("seed", 1) for m in mine))
    assert len(seeds) >= 4, seeds                           # the same seeds on both sides
    ref = [r for r in ref if r["seed"] in seeds]
    mine = [m for m in mine if m.get("seed", 1) in seeds]
    r = np.array([x["total_successes"] for x in ref], dtype=np.float64)
    m = np.array([x["total_successes"] for x in mine], dtype=np.float64)
    assert abs(r.mean() - m.mean()) <= 2.0 * max(r.std(), m.std(), 20.0), (r, m)
    rv = np.array([x["total_violations"] for x in ref], dtype=np.float64)
    mv = np.array([x["total_violations"] for x in mine], dtype=np.float64)
    assert mv.max() <= rv.max() + 3 and mv.mean() <= rv.mean() + 2
    for x in mine:
        assert x["episodes"] == 400


def _mb_runs():
    ref = {r["seed"]: r for r in (json.load(open(p)) for p in
                                  sorted(glob.glob(os.path.join(HERE, "golden", "ref_learning_nav2_mb_seed*.json"))))}
    mine = {}
    for line in open(os.path.join(HERE, "..", "profiles", "round3_learning_nav2_mb_one_env.jsonl")):
        m = json.loads(line)
        mine[m["seed"]] = m
    return ref, mine


def test_model_based_recovery_runs_next_to_the_reference_runs():
    """scripts/navigation2.sh:14 (PETS/CEM recovery: MPC.py:213-347 with the ensemble re-fit of experiment.py:464-480 after
    every episode), seeds 1..4.  The reference's own runs (tests/golden/ref_learning_nav2_mb_seed*.json, recorded by
    tests/golden/run_reference_training.py ... nav2_mb; the CPU runs take > 8 h for 400 episodes, so the fixtures hold the
    episodes that were finished -- `partial`, read from the run_stats.pkl the reference re-writes after every episode) against
    this stack's one-env runs of the same command line (profiles/round3_learning_nav2_mb_one_env.jsonl), compared over the
    SAME number of episodes per seed.  RNG streams differ; network initialisation (torch.manual_seed(seed)) is shared."""
    ref, mine = _mb_runs()
    assert sorted(ref)[:4] == sorted(mine)[:4] == [1, 2, 3, 4]
    ref = {k: ref[k] for k in (1, 2, 3, 4)}
    rates_ref, rates_mine = [], []
    for seed in ref:
        r, m = ref[seed], mine[seed]
        K = r["episodes"]
        assert K >= 50 and m["episodes"] == 400
        # no constraint violation in any run of either stack
        assert r["total_violations"] == 0 and sum(m["violations"][:K]) == 0
        rates_ref.append(sum(r["successes"]) / K)
        rates_mine.append(sum(m["successes"][:K]) / K)
        # the recovery controller is busy in the first episodes of both stacks to the same degree (same gate: same
        # pre-trained safety critic up to the offline data's RNG stream)
        early_r, early_m = np.mean(r["recovery_steps"][:10]), np.mean(m["recovery_steps_per_episode"][:10])
        assert abs(early_r - early_m) <= 0.35 * max(early_r, early_m) + 4, (seed, early_r, early_m)
    # seed 2: the pre-trained safety critic extrapolates ABOVE eps_safe to the start region on both stacks
    # (tests/golden/ref_qrisk_gate_seed2.json: 0.339; this stack 0.337, profiles/round3_qrisk_gate_16_seeds.json): the gate is
    # closed from the first step and the recovery controller holds the agent back -- no success in the first 75 episodes of
    # either stack, 35+ recovery steps per episode, every episode ended by the horizon.  (Online Q_risk updates lower the
    # extrapolated values at one update per env-step: the reference's run breaks through in episodes 80-93 -- 14 successes --
    # and stalls again; this stack's run opens the gate later and has no success within its 400 episodes.)
    for run, rec_key in ((ref[2], "recovery_steps"), (mine[2], "recovery_steps_per_episode")):
        assert sum(run["successes"][:75]) == 0 and sum(run["violations"][:75]) == 0
        assert np.mean(run[rec_key][:75]) > 35 and set(run["episode_lengths"][:75]) == {100}
    assert rates_ref[1] < 0.2 and rates_mine[1] < 0.2
    # the seeds that learn: this stack is not worse than the reference over the same episodes, and the reference's rates lie
    # inside the spread Navigation2 shows on BOTH stacks (model-free line: 0.27 .. 0.99 of 400 episodes)
    learn_r = np.array([rates_ref[i] for i in (0, 2, 3)])
    learn_m = np.array([rates_mine[i] for i in (0, 2, 3)])
    assert (learn_r > 0.3).all() and (learn_m > 0.3).all()
    assert learn_m.mean() >= learn_r.mean() - 2.0 * max(learn_r.std(), learn_m.std(), 0.05)


def test_model_based_seeds_that_stall_at_scale_stall_on_the_reference_too():
    """Seeds 6 and 8 (the two further seeds of 5..8 that do not reach the goal at 4096 envs): the reference's
    own one-env runs (120 episodes since round 5) next to this stack's.  Seed 6 is seed 2's case (closed gate at the start);
    seed 8 has its gate open, is held back ~25-30 steps per episode by the recovery controller on both stacks.  Over the first
    80 episodes neither stack has a success or a violation on either seed.  What the longer reference runs of round 5 added:
    the reference's seed 6 leaves the start after ~100 episodes (first success in episode 103, 7 by episode 120, one violation),
    as its seed 2 does after 80 -- this stack's seed 6 does not within 400 episodes, its seed 8 does from episode 174 on and the
    reference's seed 8 not within its 120.  Which of the threshold seeds opens first differs between the stacks."""
    ref, mine = _mb_runs()
    for seed in (6, 8):
        r, m = ref[seed], mine[seed]
        K = 80
        assert r["episodes"] >= K and m["episodes"] == 400
        assert sum(r["successes"][:K]) == 0 and sum(m["successes"][:K]) == 0
        assert sum(r["violations"][:K]) == 0 and sum(m["violations"][:K]) == 0
        assert set(r["episode_lengths"][:K]) == {100} and set(m["episode_lengths"][:K]) == {100}
        early_r, early_m = np.mean(r["recovery_steps"][:10]), np.mean(m["recovery_steps_per_episode"][:10])
        assert abs(early_r - early_m) <= 0.35 * max(early_r, early_m) + 4, (seed, early_r, early_m)
    if ref[6]["episodes"] >= 120:
        first = int(np.argmax(np.array(ref[6]["successes"]) > 0))
        assert 95 <= first <= 115 and ref[6]["total_successes"] <= 10 and ref[6]["total_violations"] <= 1
        assert ref[8]["total_successes"] == 0
    assert sum(mine[6]["successes"]) == 0 and sum(mine[8]["successes"]) > 100


def test_safety_critic_gate_at_the_start_state_equals_the_reference_seed_by_seed():
    """Q_risk(start state, task action) after `pretrain_critic_recovery` (10 000 steps on 20 000 offline Navigation2
    transitions): the reference's value per seed (tests/golden/ref_qrisk_gate_seed*.json, ref_qrisk_gate_probe.py) next to this
    stack's (profiles/round3_qrisk_gate_16_seeds.json, qrisk_gate_probe.py).  The start region x = -50 lies outside the offline
    data's x range [-40, 10] (env/navigation2.py:133-243): the value is an extrapolation that depends on the seed -- the same
    way on both stacks, because torch.manual_seed(seed) gives both the same initial network."""
    ref = {r["seed"]: r for r in (json.load(open(p)) for p in
                                  glob.glob(os.path.join(HERE, "golden", "ref_qrisk_gate_seed*.json")))}
    mine = {m["seed"]: m for m in json.load(open(os.path.join(HERE, "..", "profiles", "round3_qrisk_gate_16_seeds.json")))}
    common = sorted(set(ref) & set(mine))
    assert len(common) >= 12
    r = np.array([ref[s]["q_start_mean"] for s in common])
    m = np.array([mine[s]["q_start_mean"] for s in common])
    assert np.corrcoef(r, m)[0, 1] > 0.9 and np.abs(r - m).max() < 0.08, (r.round(3), m.round(3))   # measured 0.950, 0.061
    closed_r = {s for s in common if ref[s]["q_start_share_above_eps"] > 0.5}
    closed_m = {s for s in common if mine[s]["q_start_share_above_eps"] > 0.5}
    assert 2 in closed_r and 2 in closed_m
    for s_ in closed_r ^ closed_m:         # only seeds whose value is next to eps_safe fall on different sides
        assert min(abs(ref[s_]["q_start_mean"] - 0.2), abs(mine[s_]["q_start_mean"] - 0.2)) < 0.03, s_
    assert len(closed_r ^ closed_m) <= 4, (closed_r, closed_m)
    assert 0.15 <= len(closed_r) / len(common) <= 0.6                # a third of the seeds start with a closed gate


def test_model_based_line_like_for_like_windows_over_eight_seeds():
    """scripts/navigation2.sh:14, seeds 1..8, compared over the SAME window on both stacks: the first K episodes, K = what every
    reference fixture covers (>= 80; the reference's CPU runs of this line cost hours: 120-episode runs of seeds 2, 5, 6, 7, 8
    were recorded in round 5).  This stack's side is regenerated by the committed script on the GPU box (`python
    profiles/learning_other_configs.py nav2_mb 1,2,3,4,5,6,7,8 120` -> profiles/round5_learning_nav2_mb_one_env.jsonl); the
    -m gpu twin of this test (tests/test_learning_level_gpu.py) runs two seeds of it on the build under test.
    What the window shows: no violation in any run of this stack and one in the reference's (seed 6, episode 109); the same seeds
    start behind a closed gate (2, 5, 6, 8: no success in the first 80 episodes on either stack; the reference's 2 and 6 open
    after 80 / 103 episodes, this stack's do not inside the window); the learning seeds learn on both, this stack EARLIER (its recovery controller's
    regime ends around episode 20-40, the reference's around episode 60-80) on three of the four learning seeds and later on the
    fourth, so the rates are compared as distributions, one-sided."""
    path = os.path.join(HERE, "..", "profiles", "round5_learning_nav2_mb_one_env.jsonl")
    ref, mine = _mb_runs()
    if os.path.exists(path):
        mine = {m["seed"]: m for m in (json.loads(line) for line in open(path) if line.strip())}
    seeds = sorted(set(ref) & set(mine))
    assert len(seeds) >= 6, seeds
    K = min(min(ref[s]["episodes"] for s in seeds), min(mine[s]["episodes"] for s in seeds), 120)
    assert K >= 80
    W0 = 80           # "held at the start": no success in the first 80 episodes (the reference's seeds 2 / 6 leave it after 80 / 103)
    stalled_r = {s for s in seeds if sum(ref[s]["successes"][:W0]) == 0}
    stalled_m = {s for s in seeds if sum(mine[s]["successes"][:W0]) == 0}
    for s in seeds:
        assert sum(ref[s]["violations"][:W0]) == 0 and sum(mine[s]["violations"][:K]) == 0, s
        assert sum(ref[s]["violations"][:K]) <= 1, s            # (the reference's seed 6: one, in episode 109)
        early_r, early_m = np.mean(ref[s]["recovery_steps"][:10]), np.mean(mine[s]["recovery_steps_per_episode"][:10])
        assert abs(early_r - early_m) <= 0.35 * max(early_r, early_m) + 4, (s, early_r, early_m)
    # the seeds that do not leave the start inside the first 80 episodes: the same ones, up to seeds whose gate sits at the threshold
    assert 6 in stalled_r and 6 in stalled_m and len(stalled_r & stalled_m) >= 3
    assert len(stalled_r ^ stalled_m) <= 2, (stalled_r, stalled_m)
    learn = [s for s in seeds if s not in stalled_r and s not in stalled_m]
    assert len(learn) >= 3
    rate_r = np.array([sum(ref[s]["successes"][:K]) / K for s in learn])
    rate_m = np.array([sum(mine[s]["successes"][:K]) / K for s in learn])
    # seed by seed either stack can be ahead (seed 7: the reference); the distributions are compared
    assert rate_m.mean() >= rate_r.mean() - 2.0 * max(rate_r.std(), rate_m.std(), 0.05), (learn, rate_r, rate_m)
    # ... and level with the reference once both are past the recovery regime (the window's last 20 episodes)
    if K >= 100:
        tail_r = np.array([sum(ref[s]["successes"][K - 20:K]) for s in learn])
        tail_m = np.array([sum(mine[s]["successes"][K - 20:K]) for s in learn])
        assert abs(tail_r.mean() - tail_m.mean()) <= 6, (tail_r, tail_m)


def test_model_based_escape_time_follows_the_demonstration_set_not_the_kernels():
    """Round 6: why this stack left the recovery controller's regime earlier than the reference on the
    model-based line's learning seeds.  The committed record (profiles/round6_mb_diag.json: 57 runs of 46 episodes through the
    same probes on both stacks -- tests/golden/mb_diag_common.py, run_reference_mb_diag.py, profiles/mb_diag.py) says:
    (1) no hand-written path moves the first success by more than a few episodes (updates / planner / re-fit each swapped for
    torch modules + autograd, and all three); (2) neither does the planner's own Philox key; (3) the reference's reruns of
    one seed differ by more than the stacks do; (4) THIS stack on the REFERENCE's offline demonstrations of seed 1 does what
    the reference does -- no success in the window, ~40 recovery steps per episode."""
    rec = json.load(open(os.path.join(HERE, "..", "profiles", "round6_mb_diag.json")))
    runs = rec["runs"]
    mine = [r for r in runs if r["stack"] == "recovery_rl_amd"]
    first = lambda r: 99 if r["first_success_episode"] is None else r["first_success_episode"]
    paths = [r for r in mine if r["group"].startswith("paths")]
    assert {r["variant"] for r in paths} == {"-", "updates=autograd", "planner=torch", "fit=torch", "all=torch"}
    for seed in (1, 3, 4):
        eps = [first(r) for r in paths if r["seed"] == seed]
        assert len(eps) == 5 and max(eps) - min(eps) <= 6 and max(eps) <= 20, (seed, eps)          # (1)
        keys = [first(r) for r in mine if r["group"].startswith("planner's own") and r["seed"] == seed]
        assert len(keys) == 3 and max(keys) - min(eps) <= 8, (seed, keys)                            # (2)
    ref = {}
    for r in runs:
        if r["stack"] == "reference":
            ref.setdefault(r["seed"], []).append(first(r))
    assert max(abs(a - b) for a, b in (ref[s] for s in (1, 3, 4))) >= 15                            # (3): 19 against none in 46
    demos = [r for r in mine if r["offline_demonstrations"] == "the reference's draws"]
    seed1 = [r for r in demos if r["seed"] == 1]
    assert len(seed1) == 3 and all(r["successes"] == 0 for r in seed1)                                # (4)
    assert all(35 <= np.mean(r["recovery_steps"][8:20]) <= 45 for r in seed1)
    ref1 = [r for r in runs if r["stack"] == "reference" and r["seed"] == 1 and r["group"].startswith("reference rerun")][0]
    assert 33 <= np.mean(ref1["recovery_steps"][8:20]) <= 45
    own1 = [r for r in paths if r["seed"] == 1 and r["variant"] == "-"][0]
    assert own1["first_success_episode"] <= 12 and np.mean(own1["recovery_steps"][12:20]) < 20
    assert not any(r["violations"] for r in mine)                     # no constraint violation in any of this stack's 46 runs
    # the ensemble is accurate from the pre-training on, on every run: one-step error at the env's noise floor (2 x 0.05^2)
    assert all(0.003 < np.mean(r["prefit_mse"]) < 0.008 and 0.045 < np.mean(r["pred_sd"]) < 0.06 for r in mine)


def test_headline_update_to_data_ratio_learns_on_every_seed():
    """Round 6: the bench's headline runs ONE update pair per 4096 env-steps.  The committed record
    (profiles/round6_utd_trade.json: config 2 at 4096 envs, seeds 1-8, U in {1, 2, 4, 8, 16}, 600 iterations through
    Experiment.run; regenerate with `python profiles/utd_trade.py 600 1 8 1,2,4,8,16` on the GPU box) says that every seed at
    every U reaches a 25-iteration window with >= 90 % successes and ends at 100 %; U = 1 gets there in the least loop time, U = 8
    in the fewest env-steps; violations stay below 0.05 % of the episodes."""
    rec = json.load(open(os.path.join(HERE, "..", "profiles", "round6_utd_trade.json")))
    by = {}
    for r in rec["runs"]:
        by.setdefault(r["updates_per_step"], []).append(r)
    assert sorted(by) == [1, 2, 4, 8, 16] and all(len(v) == 8 for v in by.values())
    med = {}
    for U, runs in by.items():
        assert all("to_90pct" in r and r["final_success_rate"] > 0.99 for r in runs), U
        assert all(r["violations"] <= 0.0005 * r["episodes"] for r in runs), U
        med[U] = (np.median([r["to_90pct"]["loop_seconds"] for r in runs]), np.median([r["to_90pct"]["env_steps"] for r in runs]),
                  np.median([r["to_90pct"]["grad_steps"] for r in runs]))
    assert med[1][0] < 0.5 * med[16][0]                       # wall clock: U = 1 learns the task > 2x sooner than U = 16 (4.7x)
    assert med[8][1] <= min(m[1] for m in med.values())       # env-steps: U = 8 (and 16) need the fewest
    assert med[1][2] < med[16][2] / 5                          # gradient steps: a seventh
