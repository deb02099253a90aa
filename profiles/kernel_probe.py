"""Time individual hot-path kernels in isolation (200 back-to-back launches from one hipGraph)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recovery_rl_amd import _lib, fused  # noqa: E402

dev = "cuda:0"


def bench(fn, n=200):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


r = lambda *s: torch.randn(*s, device=dev)
print("empty-ish kernel (counter_add): %.2f us" % bench(lambda: _lib.load().rrl_counter_add(
    torch.zeros(2, dtype=torch.int64, device=dev).data_ptr(), 1, _lib.current_stream())))
ctr = torch.zeros(2, dtype=torch.int64, device=dev)
print("counter_add persistent: %.2f us" % bench(lambda: _lib.load().rrl_counter_add(ctr.data_ptr(), 1, _lib.current_stream())))
for (M, H, din, dout, G) in ((256, 256, 4, 1, 2), (256, 256, 2, 4, 1), (4096, 256, 4, 1, 2), (4096, 256, 2, 4, 1)):
    x = r(M, din)
    W1, b1, W2, b2, W3, b3 = r(G, H, din), r(G, H), r(G, H, H), r(G, H), r(G, dout, H), r(G, dout)
    out, h1, h2 = torch.empty(G, M, dout, device=dev), torch.empty(G, M, H, device=dev), torch.empty(G, M, H, device=dev)
    t1 = bench(lambda: fused.mlp3_forward(x, W1, b1, W2, b2, W3, b3, out=out))
    t2 = bench(lambda: fused.mlp3_forward(x, W1, b1, W2, b2, W3, b3, out=out, h1=h1, h2=h2))
    scr = torch.empty(4, G, M, dout, device=dev)
    t3 = bench(lambda: fused.mlp3_forward(x, W1, b1, W2, b2, W3, b3, out=out, h1=h1, h2=h2, scratch=scr))
    print("mlp3_fwd M=%4d G=%d dout=%d: %.2f us (no save) %.2f us (save h1,h2) %.2f us (split+sum, save)" % (M, G, dout, t1, t2, t3))
n = 134666
p, g_, m, v = r(n), r(n), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
tgt = r(n)
step = torch.zeros(2, dtype=torch.int64, device=dev)
lib = _lib.load()
print("adam %d: %.2f us" % (n, bench(lambda: lib.rrl_adam_step(n, p.data_ptr(), g_.data_ptr(), m.data_ptr(), v.data_ptr(),
                                                              step.data_ptr(), 3e-4, 0.9, 0.999, 1e-8, tgt.data_ptr(), 0.005,
                                                              _lib.current_stream()))))
for mode, name in ((fused.NT, "NT"), (fused.NN, "NN"), (fused.TN, "TN")):
    for G in (1, 2):
        A, B = r(G, 256, 256), r(G, 256, 256)
        out = torch.empty(G, 256, 256, device=dev)
        print("gemm %s 256^3 G=%d: %.2f us" % (name, G, bench(lambda: fused.gemm(mode, A, B, out=out))))
for G, dout, din in ((2, 1, 4), (1, 4, 2)):
    B, H = 256, 256
    dO, h2, W3 = r(G, B, dout), r(G, B, H), r(G, dout, H)
    dW3, db3, dh2 = torch.empty(G, dout, H, device=dev), torch.empty(G, dout, device=dev), torch.empty(G, B, H, device=dev)
    print("head_bwd G=%d dout=%d: %.2f us" % (G, dout, bench(lambda: lib.rrl_mlp_head_backward(
        G, B, H, dout, dO.data_ptr(), h2.data_ptr(), W3.data_ptr(), dW3.data_ptr(), db3.data_ptr(), dh2.data_ptr(),
        _lib.current_stream()))))
    x, W1 = r(B, din), r(G, H, din)
    dW1, db1, dx = torch.empty(G, H, din, device=dev), torch.empty(G, H, device=dev), torch.empty(G, B, din, device=dev)
    print("input_bwd G=%d din=%d (w+x): %.2f us" % (G, din, bench(lambda: lib.rrl_mlp_input_backward(
        G, B, H, din, dh2.data_ptr(), x.data_ptr(), din, W1.data_ptr(), dW1.data_ptr(), db1.data_ptr(), dx.data_ptr(),
        _lib.current_stream()))))
for H in (256, 128, 64, 32):
    M, din, dout, G = 256, 4, 1, 2
    x = r(M, din)
    W1, b1, W2, b2, W3, b3 = r(G, H, din), r(G, H), r(G, H, H), r(G, H), r(G, dout, H), r(G, dout)
    out = torch.empty(G, M, dout, device=dev)
    print("mlp3_fwd M=256 G=2 H=%d: %.2f us" % (H, bench(lambda: fused.mlp3_forward(x, W1, b1, W2, b2, W3, b3, out=out))))
from recovery_rl_amd.replay_memory import ReplayMemory
for cap, fill in ((1000000, 300000), (4096, 4096)):
    mem = ReplayMemory(cap, 1, device=dev)
    rows = (r(fill, 2), r(fill, 2), r(fill), r(fill, 2), r(fill))
    mem.push(*rows)
    for B in (64, 256, 1024):
        print("sample_gather cap=%d size=%d B=%d: %.2f us" % (cap, fill, B, bench(lambda: mem.sample(B))))
