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


def step_probe(exp):
    """Round 6: what the two learned models say about the LAST step of all envs (the fused step keeps state, executed action
    and next state for the online re-fit): the ensemble's one-step error and predicted sd, and the safety critic's value of the
    executed action -- on the rows under the recovery controller and on the others -- beside what happened (constraint flags)."""
    env, loop, mpc, qr = exp.env, exp.loop, exp.recovery_policy, exp.agent.safety_critic
    with torch.no_grad():
        s, a, s2 = env.prev_obs.float(), env.action_clipped.float(), env.next_obs.float()
        rec = loop._last_recovery.bool() if loop._last_recovery is not None else torch.zeros(s.shape[0], dtype=torch.bool, device=s.device)
        x = torch.cat([s, a], 1)[None].expand(mpc.model.num_nets, -1, -1).contiguous()
        mean, var = mpc.model(x)
        err = ((mean - (s2 - s)[None]) ** 2).sum(-1).mean(0)              # per env, mean over members
        sd = var.sqrt().mean(-1).mean(0)
        q = qr.get_value(s, a).reshape(-1)
        cons = env.constraint.bool()
        out = {"recovery_rows": int(rec.sum()), "violations_this_step": int(cons.sum()),
               "violations_this_step_under_recovery": int((cons & rec).sum()),
               "ens_mse_all": float(err.mean()), "ens_sd_all": float(sd.mean()), "q_exec_all": float(q.mean())}
        near = (s[:, 0] > -34) & (s[:, 0] < -16) & (s[:, 1].abs() < 12)    # the obstacle's caution zone (navigation2.py:43)
        out["rows_near_obstacle"] = int(near.sum())
        for name, m in (("recovery", rec), ("near", near), ("violating", cons)):
            if bool(m.any()):
                out["ens_mse_" + name] = float(err[m].mean())
                out["q_exec_" + name] = float(q[m].mean())
                out["q_exec_" + name + "_min"] = float(q[m].min())
        if bool(cons.any()):
            out["q_exec_violating_below_eps"] = float((q[cons] < exp.exp_cfg.eps_safe).float().mean())
        # the critic's landscape over the action circle on the recovery rows: is there a safe action (min over 16 directions
        # of Q_risk(s, a)) -- the planner's miss -- or is every direction rated unsafe -- the critic's?
        if bool(rec.any()):
            import math
            sr, ar = s[rec], a[rec]
            dirs = torch.tensor([[math.cos(k * math.pi / 8), math.sin(k * math.pi / 8)] for k in range(16)], device=s.device)
            qd = torch.stack([qr.get_value(sr, d.expand_as(sr)).reshape(-1) for d in dirs], 1)          # [rows, 16]
            qe = q[rec]
            away = torch.stack([-torch.ones_like(sr[:, 0]), torch.zeros_like(sr[:, 0])], 1)           # straight back (-x)
            out.update(q_dir_min_recovery=float(qd.min(1).values.mean()), q_dir_max_recovery=float(qd.max(1).values.mean()),
                       q_away_recovery=float(qr.get_value(sr, away).mean()),
                       planner_regret_recovery=float((qe - qd.min(1).values).mean()),
                       recovery_rows_with_a_safe_direction=float((qd.min(1).values < exp.exp_cfg.eps_safe).float().mean()),
                       recovery_rows_executing_unsafe=float((qe > exp.exp_cfg.eps_safe).float().mean()),
                       exec_action_norm_recovery=float(ar.norm(dim=1).mean()),
                       exec_action_mean_recovery=[round(float(v), 3) for v in ar.mean(0)],
                       recovery_rows_x=[round(float(v), 2) for v in torch.quantile(sr[:, 0], torch.tensor([0.1, 0.5, 0.9], device=s.device))],
                       recovery_rows_abs_y=float(sr[:, 1].abs().mean()))
    return out


def run(precision, plan_seed, seed=1, iterations=1675, U=16, probe=False):
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
            if probe:
                row.update(step_probe(exp))
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
            int(sys.argv[4]) if len(sys.argv) > 4 else 1675, probe=len(sys.argv) > 5 and sys.argv[5] == "probe")
    print(json.dumps({k: v for k, v in r.items() if k != "windows"}), file=sys.stderr)
    print(json.dumps(r))
