"""Learning evidence at the headline size: config 2 (scripts/navigation1.sh:7) with 4096 lock-step envs and U updates
per iteration, several seeds.  Per log window: episodes finished, the share that reached the goal and the share that
ended in a constraint violation (definitions of plotting/plot_runs.py:214-235: success = last reward > -4, violation =
any constraint in the episode -- an episode ends at its first violation, so the per-episode flag of the last step);
plus the env-steps and grad-steps spent until the first window with >= 90 % successes.

    python profiles/learning_vec4096.py [updates_per_step=16] [iterations=1500] [first_seed=1] [last_seed=4] [config=2]
                                        [plan_precision|-] [extra flags ...]
config 2 = Navigation1 model-free recovery (scripts/navigation1.sh:7), 3 = Maze model-free recovery (scripts/maze.sh:7),
4 = Navigation2 model-based recovery (scripts/navigation2.sh:14; plan_precision f32 | f16x3).
"""
import contextlib
import io
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import arg_utils  # noqa: E402
from recovery_rl_amd.experiment import Experiment  # noqa: E402

N = 4096


CONFIGS = {
    2: ["--env-name", "navigation1", "--use_recovery", "--MF_recovery", "--gamma_safe", "0.8", "--eps_safe", "0.3",
        "--logdir_suffix", "RRL_MF", "--num_unsafe_transitions", "20000"],
    3: ["--env-name", "maze", "--use_recovery", "--MF_recovery", "--gamma_safe", "0.5", "--eps_safe", "0.15",
        "--pos_fraction=0.3", "--logdir_suffix", "RRL_MF"],
    4: ["--env-name", "navigation2", "--use_recovery", "--gamma_safe", "0.65", "--eps_safe", "0.2",
        "--logdir_suffix", "RRL_MB", "--num_unsafe_transitions", "20000"],
}


def run(seed, U, iterations, log_every=25, config=2, precision="", extra=()):
    tmp = tempfile.mkdtemp()
    cfg = arg_utils.get_args(["--cuda"] + CONFIGS[config] +
                             ["--logdir", tmp, "--seed", str(seed), "--num_envs", str(N),
                              "--updates_per_step", str(U), "--num_steps", str(N * iterations), "--num_eps", "100000000",
                              "--log_every", str(log_every)] + (["--plan_precision", precision] if precision else []) + list(extra))
    t0 = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        exp = Experiment(cfg)
        hist = exp.run()
    wall = time.time() - t0
    windows, prev = [], {"episodes": 0, "num_successes": 0, "num_viols": 0, "env_steps": 0, "sac_updates": 0}
    first90 = None
    for h in hist:
        d = {k: h[k] - prev[k] for k in prev}
        prev = {k: h[k] for k in prev}
        if d["episodes"] == 0:
            continue
        w = {"iteration": h["iteration"], "env_steps": h["env_steps"], "sac_updates": h["sac_updates"],
             "episodes": d["episodes"], "success_rate": d["num_successes"] / d["episodes"],
             "violation_rate": d["num_viols"] / d["episodes"]}
        windows.append(w)
        if first90 is None and w["success_rate"] >= 0.9:
            first90 = {"env_steps": h["env_steps"], "grad_steps": h["sac_updates"], "iteration": h["iteration"]}
    last = hist[-1]
    tail = windows[-max(1, len(windows) // 5):]
    return {"config": config, "plan_precision": precision or None,
            "seed": seed, "num_envs": N, "updates_per_step": U, "iterations": last["iteration"],
            "env_steps": last["env_steps"], "sac_grad_steps": last["sac_updates"], "episodes": last["episodes"],
            "successes": last["num_successes"], "violations": last["num_viols"],
            "viol_and_recovery": last["viol_and_recovery"], "viol_and_no_recovery": last["viol_and_no_recovery"],
            "recovery_steps": last["recovery_steps"],
            "final_success_rate": sum(w["success_rate"] * w["episodes"] for w in tail) / sum(w["episodes"] for w in tail),
            "final_violation_rate": sum(w["violation_rate"] * w["episodes"] for w in tail) / sum(w["episodes"] for w in tail),
            "first_window_with_90pct_success": first90, "wall_seconds": wall,
            "offline_transitions": exp.num_unsafe_transitions, "offline_violations": exp.num_constraint_violations,
            "windows": windows}


if __name__ == "__main__":
    U = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
    lo = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    hi = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    config = int(sys.argv[5]) if len(sys.argv) > 5 else 2
    precision = sys.argv[6] if len(sys.argv) > 6 else ""
    precision = "" if precision == "-" else precision
    extra = sys.argv[7:]                       # further command-line flags of the run, e.g. --demo_share 0
    out = [dict(run(s, U, iters, config=config, precision=precision, extra=extra), extra_flags=extra) for s in range(lo, hi + 1)]
    for r in out:
        print({k: v for k, v in r.items() if k != "windows"}, file=sys.stderr)
    print(json.dumps(out))
