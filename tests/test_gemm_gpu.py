"""GPU tests of the hand-written f32 MFMA GEMM (csrc/mlp_kernels.hip) against torch fp32 matmul.
f32 MFMA is an exact fmaf chain; tolerance covers summation-order differences only."""
import pytest
import torch

from recovery_rl_amd import fused

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SHAPES = [(256, 256, 256), (256, 256, 4), (256, 1, 256), (256, 4, 256), (8, 16, 4), (8, 1, 16), (8, 16, 16),
          (4096, 256, 256), (4096, 2, 256), (100, 70, 33), (33, 1, 1), (1, 5, 300), (64, 64, 17)]


def ref64(x):
    return x.double()


@pytest.mark.parametrize("M,N,K", SHAPES)
@pytest.mark.parametrize("G", (1, 2))
def test_all_modes_match_torch(M, N, K, G):
    g = torch.Generator(device=DEV).manual_seed(M * 7 + N * 3 + K + G)
    r = lambda *s: torch.randn(*s, device=DEV, generator=g)
    tol = dict(rtol=2e-5, atol=2e-5 * max(1, K) ** 0.5)
    # NT with bias + relu
    A, B, bias = r(G, M, K), r(G, N, K), r(G, N)
    want = torch.relu(torch.einsum("gmk,gnk->gmn", ref64(A), ref64(B)) + ref64(bias)[:, None]).float()
    assert torch.allclose(fused.gemm(fused.NT, A, B, bias=bias, relu=True), want, **tol)
    want = torch.einsum("gmk,gnk->gmn", ref64(A), ref64(B)).float()
    assert torch.allclose(fused.gemm(fused.NT, A, B), want, **tol)
    # NN with relu mask
    Bn, mask = r(G, K, N), r(G, M, N)
    want = (torch.einsum("gmk,gkn->gmn", ref64(A), ref64(Bn)) * (mask > 0)).float()
    assert torch.allclose(fused.gemm(fused.NN, A, Bn, mask=mask), want, **tol)
    # TN with column sums and accumulation
    At = r(G, K, M)
    out = r(G, M, N)
    base = out.clone()
    colsum = torch.zeros(G, M, device=DEV)
    want = (ref64(base) + torch.einsum("gkm,gkn->gmn", ref64(At), ref64(Bn))).float()
    fused.gemm(fused.TN, At, Bn, out=out, colsum=colsum, accumulate=True)
    assert torch.allclose(out, want, **tol)
    assert torch.allclose(colsum, At.double().sum(1).float(), **tol)


def test_2d_strided_views_and_unaligned_leading_dims():
    g = torch.Generator(device=DEV).manual_seed(1)
    big = torch.randn(300, 300, device=DEV, generator=g)
    A = big[3:131, 5:74]          # ld 300, offset not 16-byte aligned
    W = big[140:270, 1:70]
    got = fused.gemm(fused.NT, A, W)
    assert torch.allclose(got, (A.double() @ W.double().t()).float(), rtol=2e-5, atol=2e-4)
    out = torch.zeros(128, 140, device=DEV)
    fused.gemm(fused.NT, A, W, out=out[:, 5:135])
    assert torch.equal(out[:, 5:135], got) and out[:, :5].abs().sum() == 0 and out[:, 135:].abs().sum() == 0


def test_f32_mfma_is_an_exact_fma_chain_for_small_integers():
    A = torch.randint(-8, 8, (2, 64, 96), device=DEV).float()
    B = torch.randint(-8, 8, (2, 48, 96), device=DEV).float()
    assert torch.equal(fused.gemm(fused.NT, A, B), torch.einsum("gmk,gnk->gmn", A, B))


@pytest.mark.parametrize("M,H,din,dout,G", ((256, 256, 4, 1, 2), (256, 256, 2, 4, 1), (4096, 256, 4, 1, 2),
                                            (8, 16, 4, 1, 2), (64, 32, 2, 2, 1), (100, 48, 3, 4, 3), (1, 256, 2, 4, 1)))
def test_fused_stack_forward_matches_torch(M, H, din, dout, G):
    g = torch.Generator(device=DEV).manual_seed(M + H)
    r = lambda *s: torch.randn(*s, device=DEV, generator=g)
    x = r(M, din) * 5
    W1, b1, W2, b2, W3, b3 = r(G, H, din), r(G, H), r(G, H, H) / H ** 0.5, r(G, H), r(G, dout, H) / H ** 0.5, r(G, dout)
    h1, h2 = torch.empty(G, M, H, device=DEV), torch.empty(G, M, H, device=DEV)
    out = fused.mlp3_forward(x, W1, b1, W2, b2, W3, b3, h1=h1, h2=h2)
    # small-batch variant (hidden-2 columns split over 4 workgroups + fixed-order partial sums)
    h1s, h2s = torch.empty_like(h1), torch.empty_like(h2)
    outs = fused.mlp3_forward(x, W1, b1, W2, b2, W3, b3, h1=h1s, h2=h2s,
                              scratch=torch.empty(4, G, M, dout, device=DEV))
    assert torch.equal(h1s, h1) and torch.equal(h2s, h2)
    assert torch.allclose(outs, out, rtol=1e-5, atol=1e-5)
    xd = x.double()
    rh1 = torch.relu(torch.einsum("md,ghd->gmh", xd, W1.double()) + b1.double()[:, None])
    rh2 = torch.relu(torch.einsum("gmk,ghk->gmh", rh1, W2.double()) + b2.double()[:, None])
    ro = torch.einsum("gmk,gok->gmo", rh2, W3.double()) + b3.double()[:, None]
    assert torch.allclose(h1, rh1.float(), rtol=1e-5, atol=1e-5)
    assert torch.allclose(h2, rh2.float(), rtol=1e-4, atol=1e-4)
    assert torch.allclose(out, ro.float(), rtol=1e-4, atol=1e-4)
    # strided input view (columns of a wider buffer) and no saved activations
    wide = r(M, 7)
    out2 = fused.mlp3_forward(wide[:, 1:1 + din], W1, b1, W2, b2, W3, b3)
    xd = wide[:, 1:1 + din].double()
    rh1 = torch.relu(torch.einsum("md,ghd->gmh", xd, W1.double()) + b1.double()[:, None])
    rh2 = torch.relu(torch.einsum("gmk,ghk->gmh", rh1, W2.double()) + b2.double()[:, None])
    ro = torch.einsum("gmk,gok->gmo", rh2, W3.double()) + b3.double()[:, None]
    assert torch.allclose(out2, ro.float(), rtol=1e-4, atol=1e-4)
