"""CPU suite: the command line stays flag-for-flag identical to the reference
(G10: tests/golden/cli_golden.json captured from the reference's arg_utils.py)."""
import json
import os

import pytest

import arg_utils


@pytest.fixture(scope="module")
def G(golden_dir):
    return json.load(open(os.path.join(golden_dir, "cli_golden.json")))


def test_defaults_of_all_55_reference_flags(G):
    mine = vars(arg_utils.get_args([]))
    ref = G["defaults"]
    assert len(ref) == 55
    for k, v in ref.items():
        assert k in mine, k
        assert mine[k] == v and type(mine[k]) is type(v), (k, mine[k], v)
    extra = set(mine) - set(ref)
    assert extra == {"num_envs", "log_every", "mb_dynamics", "checkpoint_every", "resume", "no_fast_path", "dp_mode", "plan_precision", "no_pin_demos", "seeds_per_gpu", "info_envs", "demo_share", "keep_replay_size", "plan_seed", "graph_iterations", "keep_plan_warm_start"}
    assert mine["num_envs"] == 1 and mine["mb_dynamics"] == "model"


def test_every_reference_script_command_line_parses_identically(G):
    assert len(G["scripts"]) >= 20
    for case in G["scripts"]:
        mine = vars(arg_utils.get_args(case["argv"]))
        for k, v in case["parsed"].items():
            assert mine[k] == v, (case["script"], case["argv"], k)


def test_argparse_quirks_are_preserved():
    a = arg_utils.get_args(["--eval", "False", "--lambda", "1000", "--pos_fraction=0.3"])
    assert a.eval is True                     # type=bool: any non-empty string is True
    assert a.lambda_RCPO == 1000.0            # prefix matching, scripts/navigation1.sh:56
    assert a.pos_fraction == 0.3
    a = arg_utils.get_args(["-ca", "k", "v", "-o", "x", "y"])
    assert a.ctrl_arg == [["k", "v"]] and a.override == [["x", "y"]]
    with pytest.raises(SystemExit):
        arg_utils.get_args(["--use_qvalue"])  # stale flag of scripts/ablations.sh is rejected
