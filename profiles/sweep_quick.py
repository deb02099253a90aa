"""Env kernels in the bandwidth regime for the library in RRL_HIP_LIB / the in-tree one: compact env step at 2^20 .. 2^24 envs
and the fused step + pushes at 2^20.    python profiles/sweep_quick.py [tag]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device("cuda:0")
tag = sys.argv[1] if len(sys.argv) > 1 else "lib"
out = []
for logn in (20, 22, 24):
    n = 1 << logn
    tc = bench.time_nav_step_compact_kernel(dev, n, reps=20)
    out.append("compact 2^%d %.1f us frac %.3f" % (logn, tc * 1e6, n * bench.NAV_STEP_ALGO_BYTES / tc / 1e9 / bench.HBM_PEAK_GBS))
    torch.cuda.empty_cache()
ts = bench.time_step_push_kernel(dev, "navigation1", 1 << 20, reps=20)
out.append("step_push 2^20 %.1f us frac %.3f" % (ts * 1e6, (1 << 20) * bench.STEP_PUSH_ALGO_BYTES / ts / 1e9 / bench.HBM_PEAK_GBS))
print(tag, " | ".join(out))
