"""The bench's config-4 leg alone (Navigation2, 4096 envs, model-based recovery, pre-trained gate): ms per iteration, both kernels.
    python profiles/config4_leg.py [iters=30]"""
import contextlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
out = {}
with contextlib.redirect_stdout(sys.stderr):
    for prec in ("f32", "f16x3"):
        r = bench.run_config4_leg(torch.device("cuda:0"), prec, iters=iters)
        out[prec] = {k: r[k] for k in ("ms_per_step", "env_steps_per_s", "planned_actions", "recovery_set_sizes")}
print(json.dumps(out))
