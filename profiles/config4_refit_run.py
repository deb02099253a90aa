"""BASELINE config 4 (Navigation2, 4096 envs, model-based recovery) through the driver for 110 lock-step iterations: offline
data, Q_risk and ensemble pre-training, the loop with the device-counted planning set, and ONE online ensemble re-fit at
iteration 100 (429 k rows, batch 131 072: the large-batch HIP kernels).  Profiling target: a rocprofv3 kernel trace of this
run must show no vendor GEMM (Cijk_*) and no autograd kernels.  Usage: python profiles/config4_refit_run.py [f32|f16x3]"""
import contextlib
import io
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import arg_utils  # noqa: E402
from recovery_rl_amd.experiment import Experiment  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
N, iters = 4096, 110
cfg = arg_utils.get_args(["--cuda", "--env-name", "navigation2", "--use_recovery", "--gamma_safe", "0.65", "--eps_safe", "0.2",
                          "--num_unsafe_transitions", "20000", "--num_envs", str(N), "--seed", "1", "--logdir",
                          tempfile.mkdtemp(), "--eval", "", "--log_every", "55", "--num_steps", str(iters * N - 1),
                          "--plan_precision", prec, "--updates_per_step", "16"])
t0 = time.time()
with contextlib.redirect_stdout(io.StringIO()):
    exp = Experiment(cfg)
    real_train = exp.recovery_policy.train
    refits = []

    def train(*a, **k):
        torch.cuda.synchronize()
        t = time.time()
        out = real_train(*a, **k)
        torch.cuda.synchronize()
        refits.append({"rows": int(exp.recovery_policy.train_in.shape[0]), "batch_size": k.get("batch_size", 32),
                       "epochs": k.get("epochs") or exp.recovery_policy.model_train_cfg["epochs"], "seconds": time.time() - t})
        return out
    exp.recovery_policy.train = train
    hist = exp.run()
last = hist[-1]
print(json.dumps({"plan_precision": prec, "iterations": last["iteration"], "env_steps": last["env_steps"],
                  "recovery_steps": last["recovery_steps"], "sac_updates": last["sac_updates"], "wall_seconds": time.time() - t0,
                  "ensemble_fits": refits}))
