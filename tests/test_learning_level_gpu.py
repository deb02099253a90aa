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
