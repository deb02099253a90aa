"""Latency of the stratified replay sampler (rrl_creplay_sample_gather, B = 256) at two capacities, HIP events around a
captured graph of 50 launches.  With RRL_HIP_LIB pointing at a build with -DRRL_CREPLAY_ABLATE=n the kernel stops early
(1 = after the count scan, 2 = after the distinct draws, 3 = after the rank -> chunk walk, 4 = after the reward scan)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from recovery_rl_amd.replay_memory import ConstraintReplayMemory  # noqa: E402

dev = "cuda:0"
out = {}
for cap in (1000000, 1 << 21):
    mem = ConstraintReplayMemory(cap, 1, device=dev)
    n = 1 << 19
    g = torch.Generator(device=dev).manual_seed(1)
    for _ in range(cap // n + 1):
        r = (torch.rand(n, device=dev, generator=g) < 0.1).float()
        mem.push(torch.randn(n, 2, device=dev), torch.randn(n, 2, device=dev), r, torch.randn(n, 2, device=dev),
                 torch.ones(n, device=dev))
    f = lambda: mem.sample(256, pos_fraction=0.3)
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(50):
            f()
    gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    gr.replay()
    e1.record()
    torch.cuda.synchronize()
    out["cap_%d_us" % cap] = round(e0.elapsed_time(e1) * 1e3 / 50, 2)
print(out)
