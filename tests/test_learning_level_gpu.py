"""Learning level of the model-based line, produced by THIS build on the GPU box (the CPU twin of this file compares
committed records): scripts/navigation2.sh:14 (PETS/CEM recovery, MPC.py:213-347, ensemble re-fit after every episode
experiment.py:464-480), one env, the first 60 episodes of seeds 3 and 2 through the committed script
profiles/learning_other_configs.py, next to the reference's own first 60 episodes of the same seeds
(tests/golden/ref_learning_nav2_mb_seed*.json, recorded by tests/golden/run_reference_training.py).  RNG streams differ,
network initialisation (torch.manual_seed(seed)) is shared: seed 2 starts behind a closed gate on both stacks, seed 3 learns."""
import json
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "profiles"))
K = 60


def _ref(seed):
    return json.load(open(os.path.join(HERE, "golden", "ref_learning_nav2_mb_seed%d.json" % seed)))


@pytest.mark.parametrize("seed", (3, 2))
def test_first_episodes_of_the_model_based_line_next_to_the_reference(seed):
    import learning_other_configs as loc
    mine, ref = loc.run("nav2_mb", seed, num_eps=K), _ref(seed)
    assert mine["episodes"] == K and ref["episodes"] >= K
    # no constraint violation in the window on either stack (the reference: none in any of its runs of this line)
    assert sum(mine["violations"]) == 0 and sum(ref["violations"][:K]) == 0
    early_m, early_r = np.mean(mine["recovery_steps_per_episode"][:10]), np.mean(ref["recovery_steps"][:10])
    assert abs(early_m - early_r) <= 0.35 * max(early_m, early_r) + 4, (early_m, early_r)     # the same gate at the start
    s_m, s_r = sum(mine["successes"]), sum(ref["successes"][:K])
    if seed == 2:
        # the pre-trained safety critic extrapolates above eps_safe to the start region on both stacks: held at the start,
        # every episode ended by the horizon
        assert s_m == 0 and s_r == 0 and set(mine["episode_lengths"]) == {100}
        assert np.mean(mine["recovery_steps_per_episode"]) > 35
    else:
        # a learning seed: successes appear inside the window on both stacks.  This stack's runs leave the recovery
        # controller's regime EARLIER than the reference's on most seeds (recorded: 49 against 7 successes in the first 60
        # episodes of this seed, 0.1 against 38.3 recovery steps per episode in episodes 20-40; both without a violation and level from episode
        # ~80 on -- DESIGN section 7 says so): the comparison is one-sided
        assert s_r >= 5 and s_m >= s_r - 10, (s_m, s_r)


def test_leaving_the_recovery_regime_does_not_depend_on_which_path_computes(tmp_path):
    """Round 6: seed 1 of the model-based line through the hand-written kernels and with the updates and the
    planner swapped for torch modules + autograd (`math=torch`: the reference's mathematics line by line for everything the gate
    and the recovery action depend on; the recorded runs include the re-fit as well) reaches its first success in the same early
    episodes -- what decides how long the stalemate between the task policy and the recovery controller lasts
    is the offline demonstration set, not the arithmetic (profiles/round6_mb_diag.json holds the full record, the reference's
    side included)."""
    import mb_diag
    first = {}
    for variant in ("-", "math=torch"):
        r = mb_diag.run(1, 20, variant=variant)
        assert r["planner_fused"] == (variant == "-") and r["updates_fused"] == (variant == "-")
        assert not any(e["violation"] for e in r["episodes"])
        first[variant] = next((i for i, e in enumerate(r["episodes"]) if e["success"]), None)
        # the ensemble is at the env's noise floor from the pre-training on
        assert 0.003 < np.mean([f["before"]["mse"] for f in r["refits"]]) < 0.008
    assert first["-"] is not None and first["math=torch"] is not None, first
    assert first["-"] <= 16 and first["math=torch"] <= 16 and abs(first["-"] - first["math=torch"]) <= 6, first


def test_on_the_references_demonstrations_this_stack_stalls_like_the_reference():
    """Round 6: a seed fixes the same initial networks on both stacks but not the same offline demonstrations
    (Philox here, the global MT19937 stream there).  Seed 1 of the model-based line on this stack's own draws reaches its first
    success by episode ~11; on the REFERENCE's demonstration set of seed 1 (tests/golden/ref_demos_nav2_seed1.npz:
    Experiment.constraint_demo_data of the imported reference, gen_ref_demos.py) it stays in the stalemate between task policy
    and recovery controller exactly as the reference's own run does -- no success, ~40 recovery steps per episode
    (tests/golden/ref_mb_diag_seed1.json: [.. 31, 34, 35, 33, 39, 40, 38, 40, 41 ..])."""
    import mb_diag
    os.environ["RRL_MB_DIAG_DEMOS"] = os.path.join(HERE, "golden", "ref_demos_nav2_seed1.npz")
    try:
        r = mb_diag.run(1, 18)
    finally:
        del os.environ["RRL_MB_DIAG_DEMOS"]
    rec = [e["recovery_steps"] for e in r["episodes"]]
    assert not any(e["success"] or e["violation"] for e in r["episodes"])
    assert 33 <= np.mean(rec[8:18]) <= 46, rec
    ref = json.load(open(os.path.join(HERE, "golden", "ref_mb_diag_seed1.json")))
    ref_rec = [e["recovery_steps"] for e in ref["episodes"]]
    assert abs(np.mean(rec[8:18]) - np.mean(ref_rec[8:18])) <= 6, (rec, ref_rec)
    assert not any(e["success"] for e in ref["episodes"][:18])
