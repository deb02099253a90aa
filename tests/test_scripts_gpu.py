"""GPU smoke of EVERY algorithm command line of the reference's scripts/navigation1.sh,
navigation2.sh and maze.sh (captured in tests/golden/cli_golden.json): each must run unmodified on
this stack (shortened to 2 episodes and a few pre-training steps)."""
import json
import os
import pickle

import pytest

import arg_utils
from recovery_rl_amd.experiment import Experiment

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cli_golden.json")
CASES = json.load(open(GOLDEN))["scripts"]


def shorten(argv, tmp):
    out, skip = [], False
    for i, a in enumerate(argv):
        if skip:
            skip = False
            continue
        if a in ("--num_eps", "--logdir"):
            skip = True
            continue
        out.append(a)
    if "--cuda" not in out:
        out.append("--cuda")           # RP / unconstrained lines of the scripts omit it; this stack is GPU-only
    return out + ["--num_eps", "2", "--logdir", str(tmp), "--critic_safe_pretraining_steps", "5",
                  "--num_unsafe_transitions", "600", "--hidden_size", "32", "--batch_size", "4", "--start_steps", "3",
                  "--eval", ""]


@pytest.mark.parametrize("case", CASES, ids=["%s:%s" % (c["script"], c["parsed"]["logdir_suffix"]) for c in CASES])
def test_reference_script_line_runs(case, tmp_path):
    cfg = arg_utils.get_args(shorten(case["argv"], tmp_path))
    assert cfg.env_name == case["parsed"]["env_name"]
    exp = Experiment(cfg)
    exp.run()
    data = pickle.load(open(os.path.join(exp.logdir, "run_stats.pkl"), "rb"))
    assert len(data["train_stats"]) == 2 and exp.total_numsteps >= 3
    assert exp.updates > 0 or exp.total_numsteps <= 5
