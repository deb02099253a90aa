"""Learning-level runs of the other reference command lines on this stack (one env, reference-order loop):
navigation2 model-free and model-based recovery (scripts/navigation2.sh:7,14), maze model-free (scripts/maze.sh:7).
The reference's own CPU runs of these lines are tests/golden/ref_learning_*.json (run_reference_training.py; the model-based
line takes hours on the CPU: 120-episode windows).  Usage:
    python profiles/learning_other_configs.py [nav2_mf|nav2_mb|maze_mf] [seed,seed,...] [num_eps]"""
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

LINES = {
    "nav2_mf": ["--env-name", "navigation2", "--use_recovery", "--MF_recovery", "--gamma_safe", "0.65", "--eps_safe", "0.2",
                "--num_eps", "400", "--num_unsafe_transitions", "20000"],
    "nav2_mb": ["--env-name", "navigation2", "--use_recovery", "--gamma_safe", "0.65", "--eps_safe", "0.2",
                "--num_eps", "400", "--num_unsafe_transitions", "20000"],
    "maze_mf": ["--env-name", "maze", "--use_recovery", "--MF_recovery", "--gamma_safe", "0.5", "--eps_safe", "0.15",
                "--pos_fraction=0.3", "--num_eps", "500"],
}


def run(name, seed, num_eps=None):
    tmp = tempfile.mkdtemp()
    cfg = arg_utils.get_args(["--cuda"] + LINES[name] + ["--logdir", tmp, "--logdir_suffix", name, "--seed", str(seed)]
                             + (["--num_eps", str(num_eps)] if num_eps else []))
    t0 = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        exp = Experiment(cfg)
        exp.run()
    data = pickle.load(open(os.path.join(exp.logdir, "run_stats.pkl"), "rb"))["train_stats"]
    viol = sum(int(any(s["constraint"] for s in ep)) for ep in data)
    if name == "maze_mf":
        succ = sum(int(-ep[-1]["reward"] < 0.03) for ep in data)           # plot_runs.py:225-226
    else:
        succ = sum(int(ep[-1]["reward"] > -4) for ep in data)
    rec = sum(int(s.get("recovery", False)) for ep in data for s in ep)
    thr = 0.03 if name == "maze_mf" else 4.0
    return {"config": name, "seed": seed, "episodes": len(data), "total_violations": viol, "total_successes": succ,
            "env_steps": sum(len(ep) for ep in data), "recovery_steps": rec, "wall_seconds": round(time.time() - t0, 1),
            # per episode (plotting/plot_runs.py:214-235), so that prefixes can be compared with partial reference runs
            "episode_lengths": [len(ep) for ep in data],
            "violations": [int(any(s["constraint"] for s in ep)) for ep in data],
            "successes": [int(-ep[-1]["reward"] < thr) for ep in data],
            "recovery_steps_per_episode": [sum(int(s.get("recovery", False)) for s in ep) for ep in data]}


if __name__ == "__main__":
    names = [sys.argv[1]] if len(sys.argv) > 1 else list(LINES)
    seeds = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1]
    eps = int(sys.argv[3]) if len(sys.argv) > 3 else None
    for n in names:
        for seed in seeds:
            print(json.dumps(run(n, seed, eps)), flush=True)
