"""Per-step time of the fused ensemble optimiser step (rrl_ens_train_epoch over 200 batches of 32)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from test_ens_train_gpu import build  # noqa: E402
from recovery_rl_amd.ensemble_train import FusedEnsembleTrainer  # noqa: E402

mpc, _ = build()
tr = FusedEnsembleTrainer(mpc.model)
tr.begin(mpc.train_in, mpc.train_targs)
idxs = torch.randint(mpc.train_in.shape[0], (5, 6400), device="cuda:0")
tr.epoch(idxs, 32)
torch.cuda.synchronize()
t0 = time.perf_counter()
tr.epoch(idxs, 32)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("fused ensemble step: %.1f us per optimiser step (200 steps)" % (dt / 200 * 1e6))
if os.environ.get("RRL_HIP_LIB", "").endswith("_ab_enstiming.so"):       # built with -DRRL_ENS_TIMING
    tr.gradients(idxs[:, :32])
    torch.cuda.synchronize()
    st = tr.scratch[-64:].view(torch.int64).cpu().tolist()
    st = [x for x in st if x > 0][:16]
    print("phase cycles (s_memtime, 100 MHz):", [st[i + 1] - st[i] for i in range(len(st) - 1)])
