"""Which point of the throughput / update-to-data line is the useful one: config 2 (scripts/navigation1.sh:7) at 4096
lock-step envs, seeds 1-4, U in {1, 4, 16} updates per iteration (experiment.py:397-416 runs `updates_per_step` update pairs
per env step; arg_utils.py:36-39).  Per run: env-steps, gradient steps and wall-seconds of the training loop
  * to the first 25-iteration window with >= 90 % successes, and
  * to 1 M env-steps (245 iterations),
and constraint violations throughout (success / violation as plotting/plot_runs.py:214-235 defines them).

    python profiles/utd_trade.py [iterations=400] [first_seed=1] [last_seed=4] [U list=1,4,16]  ->  JSON on stdout
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import learning_vec4096 as L  # noqa: E402

MILLION = 1000000


def summarise(r, hist_wall):
    """hist_wall: {iteration: seconds since the loop started} of the run's log points."""
    first = r["first_window_with_90pct_success"]
    at_1m = next((w for w in r["windows"] if w["env_steps"] >= MILLION), None)
    viol_to = lambda it: sum(round(w["violation_rate"] * w["episodes"]) for w in r["windows"] if w["iteration"] <= it)
    out = {"seed": r["seed"], "updates_per_step": r["updates_per_step"], "iterations": r["iterations"],
           "env_steps": r["env_steps"], "grad_steps": r["sac_grad_steps"], "episodes": r["episodes"],
           "violations": r["violations"], "final_success_rate": r["final_success_rate"],
           "loop_seconds": hist_wall.get(r["iterations"]), "ms_per_iteration": 1e3 * hist_wall[r["iterations"]] / r["iterations"]}
    if first is not None:
        out["to_90pct"] = {"iteration": first["iteration"], "env_steps": first["env_steps"], "grad_steps": first["grad_steps"],
                           "loop_seconds": hist_wall.get(first["iteration"]), "violations": viol_to(first["iteration"])}
    if at_1m is not None:
        out["to_1M_env_steps"] = {"iteration": at_1m["iteration"], "env_steps": at_1m["env_steps"],
                                  "grad_steps": at_1m["sac_updates"], "loop_seconds": hist_wall.get(at_1m["iteration"]),
                                  "violations": viol_to(at_1m["iteration"]), "success_rate_of_that_window": at_1m["success_rate"]}
    return out


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    lo = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    hi = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    us = [int(u) for u in (sys.argv[4] if len(sys.argv) > 4 else "1,4,16").split(",")]
    # the run's history carries the loop's wall clock per log point: take it through Experiment.run's return value
    real_run = L.Experiment.run
    walls = {}

    def run_and_keep(self):
        hist = real_run(self)
        walls.clear()
        walls.update(dict(self.log_wall))
        return hist
    L.Experiment.run = run_and_keep
    L.run(99, 1, 60)              # warm-up, discarded: the first run of a process pays the library / allocator / clock ramp
    rows = []
    for U in us:
        for seed in range(lo, hi + 1):
            r = L.run(seed, U, iters)
            rows.append(summarise(r, dict(walls)))
            print(rows[-1], file=sys.stderr)
    print(json.dumps({"config": "config 2: Navigation1, 4096 envs, SAC + Q_risk + model-free recovery (scripts/navigation1.sh:7 + "
                                "--num_envs 4096), log window 25 iterations", "runs": rows}))


if __name__ == "__main__":
    main()
