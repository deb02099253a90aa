"""Seed 1 of BASELINE config 4 (Navigation2, 4096 lock-step envs, model-based recovery, 16 updates per iteration) had ONE
violation burst with the f16x3 planner in round 4 (iterations 1 000 - 1 150) and none to speak of with the f32 planner.  Is that
the f16x3 costs' 2e-5 disagreement picking other elites, or the run-to-run divergence of a chaotic learner?  The same run with
BOTH planner kernels under a SECOND Philox key of the planner's own streams (--plan_seed: CEM samples and particle noise;
env noise, replay draws, policy noise untouched): if the burst follows the key rather than the precision, it is divergence.
Per log window (25 iterations): violations (all / under the recovery controller), successes, where the envs are.
    python profiles/config4_seed1_keys.py <f32|f16x3> <plan_seed> [seed=1] [iterations=1675]
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
import numpy as np  # noqa: E402
import torch  # noqa: E402

import arg_utils  # noqa: E402
from recovery_rl_amd.experiment import Experiment  # noqa: E402

N = 4096


def run(precision, plan_seed, seed=1, iterations=1675, U=16):
    tmp = tempfile.mkdtemp()
    cfg = arg_utils.get_args(["--cuda", "--env-name", "navigation2", "--use_recovery", "--gamma_safe", "0.65", "--eps_safe",
                              "0.2", "--logdir_suffix", "RRL_MB", "--num_unsafe_transitions", "20000", "--logdir", tmp,
                              "--seed", str(seed), "--num_envs", str(N), "--updates_per_step", str(U), "--num_steps",
                              str(N * iterations), "--num_eps", "100000000", "--log_every", "25", "--plan_precision",
                              precision, "--plan_seed", str(plan_seed)])
    rows = []
    t0 = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        exp = Experiment(cfg)
        loop, mpc = exp.loop, exp.recovery_policy
        real_read = loop.read_stats

        def read_stats():
            st = real_read()
            row = {"iteration": int(loop.total_numsteps // N)}
            if mpc.last_count is not None:
                row["planning_set"] = int(mpc.last_count[0].item())      # of the log point's own iteration
            x = exp.env.pos[:, 0].float()
            row["x_quantiles"] = [round(float(v), 2) for v in torch.quantile(x, torch.tensor([0.1, 0.5, 0.9], device=x.device))]
            rows.append(row)
            return st
        loop.read_stats = read_stats
        hist = exp.run()
    prev = {"episodes": 0, "num_successes": 0, "num_viols": 0, "viol_and_recovery": 0}
    windows = []
    for h, row in zip(hist, rows[-len(hist):]):
        d = {k: h[k] - prev[k] for k in prev}
        prev = {k: h[k] for k in prev}
        windows.append(dict(row, episodes=d["episodes"], successes=d["num_successes"], violations=d["num_viols"],
                            violations_under_recovery=d["viol_and_recovery"]))
    last = hist[-1]
    worst = max(windows, key=lambda w: w["violations"] / max(w["episodes"], 1))
    return {"plan_precision": precision, "plan_seed": plan_seed, "seed": seed, "iterations": last["iteration"],
            "episodes": last["episodes"], "successes": last["num_successes"], "violations": last["num_viols"],
            "viol_and_recovery": last["viol_and_recovery"],
            "worst_window": {"iteration": worst["iteration"], "violation_rate": worst["violations"] / max(worst["episodes"], 1),
                             "violations": worst["violations"]},
            "windows_with_violations": sum(1 for w in windows if w["violations"]), "wall_seconds": time.time() - t0,
            "windows": windows}


if __name__ == "__main__":
    r = run(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 1,
            int(sys.argv[4]) if len(sys.argv) > 4 else 1675)
    print(json.dumps({k: v for k, v in r.items() if k != "windows"}), file=sys.stderr)
    print(json.dumps(r))
