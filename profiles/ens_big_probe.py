"""One optimiser step of the PETS ensemble at the config-4 re-fit size (batch 131 072 = 32 x 4096 envs per member,
experiment.py:659): the large-batch HIP kernels (rrl_ens_train_grad_big + rrl_adam_step_multi) next to the PyTorch step
(autograd + vendor GEMMs, MPC._train_step).  Usage: python profiles/ens_big_probe.py [batch] [reps]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from recovery_rl_amd.MPC import MPC  # noqa: E402
from recovery_rl_amd.config import create_config  # noqa: E402
from recovery_rl_amd.ensemble_train import FusedEnsembleTrainer  # noqa: E402
from recovery_rl_amd.env import make_vec_env  # noqa: E402

DEV = "cuda:0"
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
torch.manual_seed(0)
env = make_vec_env("navigation2", 2, device=DEV, seed=1)
mpc = MPC(create_config("navigation2", "MPC", {}, [], "/tmp", env=env).ctrl_cfg, seed=1)
n = 424000
s = torch.rand(n, 2, device=DEV) * 40 - 45
ac = torch.rand(n, 2, device=DEV) * 2 - 1
mpc.train_in, mpc.train_targs = torch.cat([s, ac], 1).contiguous(), (ac + 0.05 * torch.randn(n, 2, device=DEV)).contiguous()
mpc.model.fit_input_stats(mpc.train_in)
idxs = torch.randint(n, (mpc.model.num_nets, n), device=DEV)
bi = idxs[:, :batch]
tr = FusedEnsembleTrainer(mpc.model)
tr.begin(mpc.train_in, mpc.train_targs)


def timed(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


rows = batch * mpc.model.num_nets
flop = rows * (3 * 2 * (4 * 200 + 200 * 200 + 200 * 200 + 200 * 4) - 2 * 4 * 200)    # fwd + two products per layer backward
t_big = timed(lambda: tr.step_big(bi))
t_grad = timed(lambda: tr.gradients_big(bi))
t_torch = timed(lambda: mpc._train_step(bi))
print(json.dumps({"batch_per_member": batch, "members": mpc.model.num_nets, "algorithmic_GFLOP": flop / 1e9,
                  "hip_step_ms": t_big * 1e3, "hip_grad_only_ms": t_grad * 1e3, "hip_TFLOPs": flop / t_grad / 1e12,
                  "frac_of_f32_mfma_peak_157.3": flop / t_grad / 1e12 / 157.3,
                  "torch_step_ms": t_torch * 1e3, "torch_TFLOPs": flop / t_torch / 1e12,
                  "speedup": t_torch / t_big}))
