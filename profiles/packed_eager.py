"""S packed seeds, eager packed iterations (every launch its own dispatch record for rocprofv3 --pmc).
    python profiles/packed_eager.py S U iters"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402
import arg_utils  # noqa: E402
from recovery_rl_amd.packed import PackedLoop  # noqa: E402

S, U, iters = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda:0")
loops = [bench.build_loop(arg_utils.get_args(bench.config_argv("navigation1", 1 + k, 4096, U)), dev) for k in range(S)]
packed = PackedLoop(loops)
for loop in loops:
    loop.vector_step(True, False, True)
    loop.vector_step(True, False, True)
packed.record()
for _ in range(iters):
    packed.step()
torch.cuda.synchronize()
print("done", S, U, iters)
