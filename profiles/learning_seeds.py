"""scripts/navigation1.sh:7 (Recovery RL, model-free recovery, 400 episodes) on this stack for seeds 1..8, one env,
reference-order loop -- the learning-level counterpart of tests/golden/ref_learning_nav1_seed*.json (the REFERENCE's
own runs of the same command line).  Prints / writes per-seed successes, violations, env-steps, wall seconds.

    python profiles/learning_seeds.py [first_seed] [last_seed] > gpurun_out/learning_seeds.json
"""
import contextlib
import io
import json
import os
import pickle
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import arg_utils  # noqa: E402
from recovery_rl_amd.experiment import Experiment  # noqa: E402


def run(seed):
    tmp = tempfile.mkdtemp()
    cfg = arg_utils.get_args(["--cuda", "--env-name", "navigation1", "--use_recovery", "--MF_recovery", "--gamma_safe",
                              "0.8", "--eps_safe", "0.3", "--logdir", tmp, "--logdir_suffix", "RRL_MF", "--num_eps", "400",
                              "--num_unsafe_transitions", "20000", "--seed", str(seed), "--eval", ""])
    t0 = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        exp = Experiment(cfg)
        exp.run()
    data = pickle.load(open(os.path.join(exp.logdir, "run_stats.pkl"), "rb"))["train_stats"]
    viol = [int(any(s["constraint"] for s in ep)) for ep in data]
    succ = [int(ep[-1]["reward"] > -4) for ep in data]
    return {"seed": seed, "episodes": len(data), "total_violations": sum(viol), "total_successes": sum(succ),
            "env_steps": sum(len(ep) for ep in data), "wall_seconds": time.time() - t0,
            "num_constraint_transitions": exp.num_unsafe_transitions,
            "num_constraint_violations_offline": exp.num_constraint_violations,
            "violations": viol, "successes": succ, "episode_lengths": [len(ep) for ep in data]}


if __name__ == "__main__":
    lo = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    hi = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    out = [run(s) for s in range(lo, hi + 1)]
    for r in out:
        print({k: v for k, v in r.items() if not isinstance(v, list)}, file=sys.stderr)
    print(json.dumps(out))
