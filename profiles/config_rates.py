"""Rates of SURVEY 8d configs 3 (Maze, MF recovery) and 4 (Navigation2, model-based recovery) at 4096 envs on
one MI355X; config 2 is bench.py's line.  Prints one JSON object per config.

    python profiles/config_rates.py [3|4] [num_envs] [Q_risk pre-training steps = 10000, the reference's default] [f16x3]

Config 4 is reported twice: with the safety critic as pre-trained on the offline data (the regime the reference runs
in: only the envs whose Q_risk exceeds eps_safe plan) and -- `untrained` -- with every env planning (worst case).
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import arg_utils  # noqa: E402
from recovery_rl_amd.experiment import Experiment  # noqa: E402

ARGV = {
    3: ["--env-name", "maze", "--use_recovery", "--MF_recovery", "--gamma_safe", "0.5", "--eps_safe", "0.15",
        "--pos_fraction", "0.3"],                                            # scripts/maze.sh:7
    4: ["--env-name", "navigation2", "--use_recovery", "--gamma_safe", "0.65", "--eps_safe", "0.2"],  # navigation2.sh:14
}


def run(config, n, pretrain=10000, precision=""):
    cfg = arg_utils.get_args(ARGV[config] + ["--cuda", "--num_envs", str(n), "--seed", "1", "--logdir", "/tmp/rrl_rates",
                                             "--critic_safe_pretraining_steps", str(pretrain)] +
                             (["--plan_precision", precision] if precision else []))
    exp = Experiment(cfg)
    t0 = time.time()
    exp.pretrain_critic_recovery()
    torch.cuda.synchronize()
    pre_s = time.time() - t0
    loop = exp.loop
    loop.start()
    while not (len(exp.memory) > cfg.batch_size and loop.total_numsteps >= cfg.start_steps):
        loop.vector_step(do_update=False, random_actions=True)
    out = {"config": config, "num_envs": n, "qrisk_pretraining_steps": pretrain, "pretrain_s": round(pre_s, 2)}
    if config == 4:
        out["plan_precision"] = "f16x3" if exp.recovery_policy.fused.f16x3 else "f32"
    if config == 3:
        loop.capture(online_qrisk=True)
        for _ in range(20):
            loop.replay()
        torch.cuda.synchronize()
        k = 300
        t0 = time.perf_counter()
        for _ in range(k):
            loop.replay()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / k
        out.update(ms_per_iteration=dt * 1e3, env_steps_per_s=n / dt, grad_steps_per_s=1 / dt, graph=True)
    else:
        mpc = exp.recovery_policy
        sizes, k = [], 6 if pretrain < 1000 else 30
        for _ in range(2):
            loop.vector_step(do_update=True, online_qrisk=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            loop.vector_step(do_update=True, online_qrisk=True)
            sizes.append(int(loop._last_recovery.sum().item()) if getattr(loop, "_last_recovery", None) is not None
                         else -1)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / k
        rows_per_env = mpc.optimizer.popsize * mpc.npart * mpc.plan_hor * mpc.optimizer.max_iters
        mean_m = sum(sizes) / len(sizes)
        out.update(ms_per_iteration=dt * 1e3, env_steps_per_s=n / dt, grad_steps_per_s=1 / dt, graph=False,
                   recovery_set_sizes=sizes, planner_rows_per_s=mean_m * rows_per_env / dt,
                   planner_tflops=mean_m * rows_per_env * 2 * (81_800 + 133_632) / dt / 1e12)
    print(json.dumps(out))
    return out


if __name__ == "__main__":
    which = [int(sys.argv[1])] if len(sys.argv) > 1 else [3, 4]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    pre = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
    prec = sys.argv[4] if len(sys.argv) > 4 else ""
    for c in which:
        run(c, n, pre, prec)
