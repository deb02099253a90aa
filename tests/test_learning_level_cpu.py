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
    """scripts/navigation2.sh:7 (model-free recovery), seeds 1..4: this stack's runs
    (profiles/round1_learning_other_configs.jsonl) next to the reference's own (tests/golden/ref_learning_nav2_seed*).
    This configuration has a large seed-to-seed spread on BOTH stacks (reference 238 .. 397 successes, this stack
    106 .. 391; seed 4 is the worst run of both), so the distributions are compared, not single runs."""
    ref = [json.load(open(p)) for p in sorted(glob.glob(os.path.join(HERE, "golden", "ref_learning_nav2_seed*.json")))]
    mine = [json.loads(line) for line in open(os.path.join(HERE, "..", "profiles", "round1_learning_other_configs.jsonl"))]
    mine = [m for m in mine if m["config"] == "nav2_mf"]
    assert len(ref) >= 1 and len(mine) >= len(ref)
    r = np.array([x["total_successes"] for x in ref], dtype=np.float64)
    m = np.array([x["total_successes"] for x in mine], dtype=np.float64)
    assert abs(r.mean() - m.mean()) <= 2.0 * max(r.std(), m.std(), 20.0), (r, m)
    rv = np.array([x["total_violations"] for x in ref], dtype=np.float64)
    mv = np.array([x["total_violations"] for x in mine], dtype=np.float64)
    assert mv.max() <= rv.max() + 3 and mv.mean() <= rv.mean() + 2
    for x in mine:
        assert x["episodes"] == 400
