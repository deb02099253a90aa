"""Probe rocBLAS vs hipBLASLt (and TunableOp) on the MLP GEMM shapes of the hot path."""
import os
import sys
import time

import torch

dev = "cuda:0"
shapes = [(256, 256, 256), (256, 4, 256), (256, 256, 1), (4096, 256, 256), (4096, 4, 256), (4096, 256, 2)]


def bench(fn, n=200):
    for _ in range(20):
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


def run(tag):
    for (B, K, N) in shapes:
        x = torch.randn(B, K, device=dev)
        w = torch.randn(N, K, device=dev)
        b = torch.randn(N, device=dev)
        gy = torch.randn(B, N, device=dev)
        t_f = bench(lambda: torch.nn.functional.linear(x, w, b))
        t_dx = bench(lambda: gy @ w)
        t_dw = bench(lambda: gy.t() @ x)
        print("%-10s B=%4d K=%3d N=%3d  fwd %6.1f us  dX %6.1f us  dW %6.1f us" % (tag, B, K, N, t_f, t_dx, t_dw))


mode = sys.argv[1] if len(sys.argv) > 1 else "default"
if mode == "hipblas":
    torch.backends.cuda.preferred_blas_library("hipblas")
elif mode == "hipblaslt":
    torch.backends.cuda.preferred_blas_library("hipblaslt")
print("preferred:", torch.backends.cuda.preferred_blas_library(), "tunable:", os.environ.get("PYTORCH_TUNABLEOP_ENABLED"))
run(mode)

if mode == "rrl":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from recovery_rl_amd import fused
    for (B, K, N) in shapes:
        for G in (1, 2):
            x = torch.randn(G, B, K, device=dev)
            w = torch.randn(G, N, K, device=dev)
            b = torch.randn(G, N, device=dev)
            gy = torch.randn(G, B, N, device=dev)
            y = torch.empty(G, B, N, device=dev)
            dx = torch.empty(G, B, K, device=dev)
            dw = torch.empty(G, N, K, device=dev)
            db = torch.empty(G, N, device=dev)
            t_f = bench(lambda: fused.gemm(fused.NT, x, w, out=y, bias=b, relu=True))
            t_dx = bench(lambda: fused.gemm(fused.NN, gy, w, out=dx, mask=x))
            t_dw = bench(lambda: fused.gemm(fused.TN, gy, x, out=dw, colsum=db))
            print("rrl G=%d   B=%4d K=%3d N=%3d  fwd %6.1f us  dX %6.1f us  dW+db %6.1f us" % (G, B, K, N, t_f, t_dx, t_dw))
