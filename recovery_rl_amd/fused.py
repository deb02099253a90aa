"""Thin Python wrappers over the fused HIP building blocks (csrc/mlp_kernels.hip)."""
import torch

from . import _lib

NT, NN, TN = 0, 1, 2


def gemm(mode, A, B, out=None, bias=None, relu=False, mask=None, colsum=None, accumulate=False):
    """Batched (leading dim = heads) or plain 2-D f32 GEMM on the hand-written MFMA kernel.
    NT: A[G,M,K] B[G,N,K] -> [G,M,N];  NN: A[G,M,K] B[G,K,N];  TN: A[G,K,M] B[G,K,N] -> [G,M,N]."""
    lib = _lib.load()
    squeeze = A.dim() == 2
    if squeeze:
        A, B = A.unsqueeze(0), B.unsqueeze(0)
        out = None if out is None else out.unsqueeze(0)
        bias = None if bias is None else bias.unsqueeze(0)
        mask = None if mask is None else mask.unsqueeze(0)
        colsum = None if colsum is None else colsum.unsqueeze(0)
    G = A.shape[0]
    if mode == NT:
        M, K, N = A.shape[1], A.shape[2], B.shape[1]
    elif mode == NN:
        M, K, N = A.shape[1], A.shape[2], B.shape[2]
    else:
        K, M, N = A.shape[1], A.shape[2], B.shape[2]
    for t in (A, B):
        assert t.dtype == torch.float32 and t.stride(2) == 1
    if out is None:
        out = torch.empty(G, M, N, dtype=torch.float32, device=A.device)
    assert out.stride(2) == 1
    rc = lib.rrl_gemm_f32(
        mode, G, M, N, K, A.data_ptr(), A.stride(1), A.stride(0) if G > 1 else 0,
        B.data_ptr(), B.stride(1), B.stride(0) if G > 1 else 0,
        out.data_ptr(), out.stride(1), out.stride(0) if G > 1 else 0,
        _lib.ptr(bias), (bias.stride(0) if G > 1 else 0) if bias is not None else 0, int(relu),
        _lib.ptr(mask), mask.stride(1) if mask is not None else 0,
        (mask.stride(0) if G > 1 else 0) if mask is not None else 0,
        _lib.ptr(colsum), (colsum.stride(0) if G > 1 else 0) if colsum is not None else 0,
        int(accumulate), _lib.current_stream())
    _lib.check(rc, "rrl_gemm_f32")
    return out[0] if squeeze else out


def mlp3_forward(x, W1, b1, W2, b2, W3, b3, out=None, h1=None, h2=None, scratch=None, finalize=True):
    """Fused stack forward (rrl_mlp3_forward).  x [M,din]; W1 [G,H,din] ... W3 [G,dout,H] contiguous.
    Returns out [G,M,dout]."""
    lib = _lib.load()
    G, H, din = W1.shape
    dout = W3.shape[1]
    M = x.shape[0]
    assert x.stride(1) == 1 and W1.is_contiguous() and W2.is_contiguous() and W3.is_contiguous()
    if out is None:
        out = torch.empty(G, M, dout, dtype=torch.float32, device=x.device)
    rc = lib.rrl_mlp3_forward(G, M, H, din, dout, x.data_ptr(), x.stride(0), W1.data_ptr(), b1.data_ptr(),
                              W2.data_ptr(), b2.data_ptr(), W3.data_ptr(), b3.data_ptr(), _lib.ptr(h1),
                              _lib.ptr(h2), out.data_ptr(), _lib.ptr(scratch), int(finalize),
                              _lib.current_stream())
    _lib.check(rc, "rrl_mlp3_forward")
    return out


def mlp3_supported(H, din, dout):
    return H % 16 == 0 and H <= 256 and din <= 4 and dout <= 4
