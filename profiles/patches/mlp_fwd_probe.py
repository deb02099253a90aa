"""Forward of the 2-hidden-layer stacks at the acting pass's size (4096 rows): plain tiling vs the column-split kernel
(RRL_SPLIT_MAX_M lifts its batch limit), one head (policies) and two heads (Q_risk), alone and grouped.
    RRL_SPLIT_MAX_M=8192 python profiles/mlp_fwd_probe.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from recovery_rl_amd import _lib  # noqa: E402
from recovery_rl_amd.fast_update import FlatNet, Stack, forward_multi  # noqa: E402

dev = torch.device("cuda:0")


def net(G, din, dout, H=256):
    f = FlatNet([("W1", (G, H, din)), ("b1", (G, H)), ("W2", (G, H, H)), ("b2", (G, H)), ("W3", (G, dout, H)),
                 ("b3", (G, dout))], dev)
    f.flat.normal_(0, 0.05)
    f.G, f.H, f.din, f.dout = G, H, din, dout
    return f


def timeit(fn, reps=200):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


out = {"RRL_SPLIT_MAX_M": os.environ.get("RRL_SPLIT_MAX_M", "1024")}
for M in (256, 1024, 4096):
    x2, x4 = torch.randn(M, 2, device=dev), torch.randn(M, 4, device=dev)
    pol, rec, qr = net(1, 2, 4), net(1, 2, 2), net(2, 4, 1)
    sp, sr, sq = Stack(pol, M), Stack(rec, M), Stack(qr, M)
    out["M%d" % M] = {
        "split": bool(sp.split),
        "policy_us": timeit(lambda: sp.forward(x2, save=False)),
        "qrisk_us": timeit(lambda: sq.forward(x4, save=False)),
        "policy+recovery grouped_us": timeit(lambda: forward_multi([sp.forward_desc(x2, save=False),
                                                                   sr.forward_desc(x2, save=False)])),
        "policy+recovery+qrisk grouped_us (not the real dependency)": timeit(lambda: forward_multi(
            [sp.forward_desc(x2, save=False), sr.forward_desc(x2, save=False), sq.forward_desc(x4, save=False)]))
        if sp.split else None,
    }
print(json.dumps(out))
