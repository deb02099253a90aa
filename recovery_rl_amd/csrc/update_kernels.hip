// update_kernels.hip -- fused element-wise pieces of the SAC / Q_risk updates on gfx950.
//
// Each kernel replaces a chain of 5-30 PyTorch element-wise launches of recovery_rl/sac.py:192-239,
// qrisk.py:118-158 and model.py:324-340 (and their autograd backward) by one launch; together with
// rrl_gemm_f32 they give a hand-written forward + backward for the 2-hidden-layer MLPs.
// Batches are tiny (B = 256): one workgroup, latency-bound by design.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "rrl_device.hpp"
#include "pack.hpp"
#include "rrl_host.hpp"

namespace {

using rrl_host::check_launch;
using rrl_host::grid_for;
using rrl_host::kBlock;

constexpr float kLogSigMax = 2.f, kLogSigMin = -20.f, kEps = 1e-6f;   // model.py:14-16
constexpr float kHalfLog2Pi = 0.918938533204672742f;

__device__ __forceinline__ float sigmoidf(float z) { return 1.f / (1.f + expf(-z)); }

// A stack output may arrive as `np` partial sums (rrl_mlp3_forward with scratch and no final sum): element
// idx = p[idx] + p[ps + idx] + ... in that fixed order (the order of the stand-alone sum kernel).
__device__ __forceinline__ float psum(const float* p, long long idx, int np, long long ps) {
    // np <= 4; every load issued before the first add, sum in the fixed order ((p0 + p1) + p2) + p3
    const float v0 = p[idx];
    const float v1 = p[(np > 1 ? ps : 0) + idx];
    const float v2 = p[(np > 2 ? 2 * ps : 0) + idx];
    const float v3 = p[(np > 3 ? 3 * ps : 0) + idx];
    float v = v0;
    v = np > 1 ? v + v1 : v;
    v = np > 2 ? v + v2 : v;
    v = np > 3 ? v + v3 : v;
    return v;
}

// ---- tanh-Gaussian head (GaussianPolicy.sample, model.py:324-340) -------------------------------
// head[b] = (mean0, mean1, log_std0, log_std1) raw outputs of the last linear layer
// obs_in (nullable, [B,2]) is copied to obs_out (row stride ld_action): assembles the [s | a] critic input in place
__device__ __forceinline__ void gauss_head_fwd_row(int b, const float* head, int np, long long ps, const float* eps,
                                                   const float* scale, const float* bias, float* action,
                                                   int ld_action, float* logp, float* mean_action,
                                                   const float* obs_in, float* obs_out) {
    if (obs_in) {
        obs_out[(long long)b * ld_action] = obs_in[2 * b];
        obs_out[(long long)b * ld_action + 1] = obs_in[2 * b + 1];
    }
    float lp = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float mean = psum(head, 4 * b + j, np, ps);
        const float ls = fminf(fmaxf(psum(head, 4 * b + 2 + j, np, ps), kLogSigMin), kLogSigMax);
        const float e = eps[2 * b + j];
        const float y = tanhf(mean + expf(ls) * e);
        action[(long long)b * ld_action + j] = y * scale[j] + bias[j];
        lp += -0.5f * e * e - ls - kHalfLog2Pi - logf(scale[j] * (1.f - y * y) + kEps);
        if (mean_action) mean_action[2 * b + j] = tanhf(mean) * scale[j] + bias[j];
    }
    if (logp) logp[b] = lp;
}

__global__ void gauss_head_fwd_kernel(int B, const float* head, int np, long long ps, const float* eps,
                                      const float* scale, const float* bias, float* action, int ld_action,
                                      float* logp, float* mean_action, const float* obs_in, float* obs_out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    gauss_head_fwd_row(b, head, np, ps, eps, scale, bias, action, ld_action, logp, mean_action, obs_in, obs_out);
}

// backward of the head: given dL/d action[b,j] (d_action, leading dim ld) and dL/d logp[b] = dlogp
// (a constant, alpha / B) produce dL/d head[b, 0..3]
__global__ void gauss_head_bwd_kernel(int B, const float* head, int np, long long ps, const float* eps,
                                      const float* scale, const float* d_action, int ld, int n_heads, long long head_stride,
                                      float dlogp, float* dhead) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        float da = 0.f;   // dL/d action summed over the critic heads that consumed it
        for (int hd = 0; hd < n_heads; ++hd) da += d_action[hd * head_stride + (long long)b * ld + j];
        const float mean = psum(head, 4 * b + j, np, ps);
        const float raw = psum(head, 4 * b + 2 + j, np, ps);
        const float ls = fminf(fmaxf(raw, kLogSigMin), kLogSigMax);
        const float std = expf(ls), e = eps[2 * b + j];
        const float y = tanhf(mean + std * e);
        const float one_m = 1.f - y * y;
        // action = y scale + bias ; logp term = -log(scale (1 - y^2) + eps)
        const float dx = da * scale[j] * one_m + dlogp * (2.f * scale[j] * y * one_m) / (scale[j] * one_m + kEps);
        const bool inside = (raw >= kLogSigMin) & (raw <= kLogSigMax);   // clamp passes gradient inside
        dhead[4 * b + j] = dx;
        dhead[4 * b + 2 + j] = inside ? (dx * std * e - dlogp) : 0.f;
    }
}

// ---- SAC critic target + loss gradient (sac.py:192-214) -----------------------------------------
// q, qt: [2, B] (online on (s,a); target on (s', a')); dq = d(mse1 + mse2)/dq ; loss[0..1] = mse
__global__ void sac_critic_grad_kernel(int B, const float* q, const float* qt, int np, long long ps,
                                       const float* logp2,
                                       const float* r, const float* m, float gamma, const float* alpha,
                                       const float* penalty, float* dq, float* loss) {
    __shared__ float red[2][kBlock];
    float l0 = 0.f, l1 = 0.f;
    for (int b = threadIdx.x; b < B; b += kBlock) {
        float y = r[b] + m[b] * gamma * (fminf(psum(qt, b, np, ps), psum(qt, B + b, np, ps)) - alpha[0] * logp2[b]);
        if (penalty) y -= penalty[b];                 // RCPO: lambda * Q_risk (sac.py:202-205)
        const float e0 = psum(q, b, np, ps) - y, e1 = psum(q, B + b, np, ps) - y;
        dq[b] = 2.f * e0 / B;
        dq[B + b] = 2.f * e1 / B;
        l0 += e0 * e0;
        l1 += e1 * e1;
    }
    red[0][threadIdx.x] = l0;
    red[1][threadIdx.x] = l1;
    __syncthreads();
    for (int off = kBlock / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            red[0][threadIdx.x] += red[0][threadIdx.x + off];
            red[1][threadIdx.x] += red[1][threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0 && loss) { loss[0] = red[0][0] / B; loss[1] = red[1][0] / B; }
}

// ---- SAC policy loss gradient (sac.py:216-231) ---------------------------------------------------
// loss = mean(alpha logp - min(q1,q2)); dqp[i][b] = -1/B on the smaller head (ties split)
__global__ void sac_policy_grad_kernel(int B, const float* qp, int np, long long ps, const float* logp,
                                       const float* alpha, float* dqp, float* loss) {
    __shared__ float red[kBlock];
    float l = 0.f;
    for (int b = threadIdx.x; b < B; b += kBlock) {
        const float a = psum(qp, b, np, ps), c = psum(qp, B + b, np, ps);
        const float w0 = a < c ? 1.f : (a == c ? 0.5f : 0.f);
        dqp[b] = -w0 / B;
        dqp[B + b] = -(1.f - w0) / B;
        l += alpha[0] * logp[b] - fminf(a, c);
    }
    red[threadIdx.x] = l;
    __syncthreads();
    for (int off = kBlock / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0 && loss) loss[0] = red[0] / B;
}

// ---- Q_risk critic target + loss gradient (qrisk.py:118-148) ------------------------------------
// z, zt: [2,B] PRE-sigmoid outputs; q = sigmoid(z); y = c + m gamma_safe max(sigmoid(zt))
__global__ void qrisk_critic_grad_kernel(int B, const float* z, const float* zt, int np, long long ps,
                                         const float* c,
                                         const float* m, float gamma_safe, float* dz, float* loss) {
    __shared__ float red[2][kBlock];
    float l0 = 0.f, l1 = 0.f;
    for (int b = threadIdx.x; b < B; b += kBlock) {
        const float y = c[b] + m[b] * gamma_safe *
                                   fmaxf(sigmoidf(psum(zt, b, np, ps)), sigmoidf(psum(zt, B + b, np, ps)));
        const float q0 = sigmoidf(psum(z, b, np, ps)), q1 = sigmoidf(psum(z, B + b, np, ps));
        const float e0 = q0 - y, e1 = q1 - y;
        dz[b] = 2.f * e0 / B * q0 * (1.f - q0);
        dz[B + b] = 2.f * e1 / B * q1 * (1.f - q1);
        l0 += e0 * e0;
        l1 += e1 * e1;
    }
    red[0][threadIdx.x] = l0;
    red[1][threadIdx.x] = l1;
    __syncthreads();
    for (int off = kBlock / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            red[0][threadIdx.x] += red[0][threadIdx.x + off];
            red[1][threadIdx.x] += red[1][threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0 && loss) { loss[0] = red[0][0] / B; loss[1] = red[1][0] / B; }
}

// loss = mean(max(sigmoid(z1), sigmoid(z2))) (qrisk.py:150-154): dz on the larger head
__global__ void qrisk_policy_grad_kernel(int B, const float* zp, int np, long long ps, float* dzp, float* loss) {
    __shared__ float red[kBlock];
    float l = 0.f;
    for (int b = threadIdx.x; b < B; b += kBlock) {
        const float q0 = sigmoidf(psum(zp, b, np, ps)), q1 = sigmoidf(psum(zp, B + b, np, ps));
        const float w0 = q0 > q1 ? 1.f : (q0 == q1 ? 0.5f : 0.f);
        dzp[b] = w0 / B * q0 * (1.f - q0);
        dzp[B + b] = (1.f - w0) / B * q1 * (1.f - q1);
        l += fmaxf(q0, q1);
    }
    red[threadIdx.x] = l;
    __syncthreads();
    for (int off = kBlock / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0 && loss) loss[0] = red[0] / B;
}

// ---- model-free recovery policy head (StochasticPolicy, model.py:511-525) ------------------------
// raw[b] = last linear output (2); mean = tanh(raw) scale + bias; action = mean + exp(max(log_std, min)) eps
__device__ __forceinline__ void stoch_head_fwd_row(int b, const float* raw, int np, long long ps, const float* eps,
                                                   const float* log_std, float min_log_std, const float* scale,
                                                   const float* bias, float* action, int ld_action, float* mean_out) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float mean = tanhf(psum(raw, 2 * b + j, np, ps)) * scale[j] + bias[j];
        const float std = expf(fmaxf(log_std[j], min_log_std));
        const float e = eps ? eps[2 * b + j] : 0.f;
        action[(long long)b * ld_action + j] = mean + std * e;
        if (mean_out) mean_out[2 * b + j] = mean;
    }
}

__global__ void stoch_head_fwd_kernel(int B, const float* raw, int np, long long ps, const float* eps,
                                      const float* log_std,
                                      float min_log_std, const float* scale, const float* bias,
                                      float* action, int ld_action, float* mean_out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    stoch_head_fwd_row(b, raw, np, ps, eps, log_std, min_log_std, scale, bias, action, ld_action, mean_out);
}

// independent policy heads in one launch (e.g. a' = pi(s') and pi(s) of one SAC step; the task action and the
// recovery action of the acting pass): flat grid over (member, row block)
constexpr int kMaxHeads = 4;
struct HeadGroup {
    rrl_policy_head_t h[kMaxHeads];
    int first[kMaxHeads + 1];
    int n;
};

__global__ __launch_bounds__(kBlock) void policy_heads_group_kernel(HeadGroup hg) {
    int k = 0;
    while (k + 1 < hg.n && (int)blockIdx.x >= hg.first[k + 1]) ++k;
    const rrl_policy_head_t& h = hg.h[k];
    const int b = (blockIdx.x - hg.first[k]) * kBlock + threadIdx.x;
    if (b >= h.B) return;
    if (h.kind == RRL_HEAD_GAUSS)
        gauss_head_fwd_row(b, h.head, h.n_part, h.part_stride, h.eps, h.scale, h.bias, h.action, h.ld_action, h.logp,
                           h.mean_out, h.obs_in, h.obs_out);
    else
        stoch_head_fwd_row(b, h.head, h.n_part, h.part_stride, h.eps, h.log_std, h.min_log_std, h.scale, h.bias,
                           h.action, h.ld_action, h.mean_out);
}

// d_action [B,2] (leading dim ld) -> draw [B,2] and dlog_std[2] (sum over the batch; single workgroup)
__global__ void stoch_head_bwd_kernel(int B, const float* raw, int np, long long ps, const float* eps,
                                      const float* log_std,
                                      float min_log_std, const float* scale, const float* d_action, int ld,
                                      int n_heads, long long head_stride, float* draw, float* dlog_std) {
    __shared__ float red[2][kBlock];
    float s0 = 0.f, s1 = 0.f;
    for (int b = threadIdx.x; b < B; b += kBlock) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float t = tanhf(psum(raw, 2 * b + j, np, ps));
            float da = 0.f;
            for (int hd = 0; hd < n_heads; ++hd) da += d_action[hd * head_stride + (long long)b * ld + j];
            draw[2 * b + j] = da * scale[j] * (1.f - t * t);
            const float std = expf(fmaxf(log_std[j], min_log_std));
            const float g = (log_std[j] >= min_log_std) ? da * std * eps[2 * b + j] : 0.f;
            if (j == 0) s0 += g; else s1 += g;
        }
    }
    // the order of the loss-fused head backward (mlp_kernels.hip): DPP sums inside the 16-lane rows, then the 16 row
    // sums one after the other -- the two paths give the same bits
    s0 = rrl::row16_sum(s0);
    s1 = rrl::row16_sum(s1);
    if ((threadIdx.x & 15) == 0) {
        red[0][threadIdx.x >> 4] = s0;
        red[1][threadIdx.x >> 4] = s1;
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        float tot = 0.f;
#pragma unroll
        for (int r = 0; r < kBlock / 16; ++r) tot += red[threadIdx.x][r];
        dlog_std[threadIdx.x] = tot;
    }
}

// ---- Adam (+ Polyak target update) over one flat parameter buffer -------------------------------
// torch.optim.Adam semantics (no weight decay, no amsgrad):
//   m <- m + (g - m)(1 - b1) ; v <- b2 v + (1 - b2) g^2 ; p <- p - lr/(1 - b1^t) m / (sqrt(v)/sqrt(1 - b2^t) + eps)
// then, if target: target <- (1 - tau) target + tau p     (utils.soft_update, utils.py:46-49)
// step_dev = {t, ticket}: t is read by every workgroup, the last one to finish stores t + 1.
struct AdamSegs {
    rrl_adam_seg_t seg[RRL_ADAM_MAX_SEGS];
    int vec[RRL_ADAM_MAX_SEGS];                  // every pointer of the segment is 16-byte aligned
    int first_block[RRL_ADAM_MAX_SEGS + 1];      // workgroups [first_block[k], first_block[k+1]) serve segment k
};

__device__ __forceinline__ float adam_elem(float& p, float g, float& m, float& v, float step_size, float bc2_sqrt,
                                           float b1, float b2, float eps, float wd) {
    const float gi = g + wd * p;
    const float mi = m + (gi - m) * (1.f - b1);
    const float vi = b2 * v + (1.f - b2) * gi * gi;
    m = mi;
    v = vi;
    p = p - step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
    return p;
}

// `vec` (host-checked: every pointer 16-byte aligned): the body runs on float4 with the loads of two strided slots
// in flight per thread (135 K parameters over 64 workgroups are 2 float4 per thread; the scalar loop chained
// one memory round trip per element: 7.5 us); the arithmetic per element is the scalar one.
// the first `pe` gradients of a segment may arrive as np partial sums ps floats apart (rrl_first_layer_t): loads of up
// to kMaxGradParts parts issued together, added in the order 0, 1, 2, ...
constexpr int kMaxGradParts = 64;
__device__ __forceinline__ float4 part_sum4(const float* gp, int np, long long ps, long long i4) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t0 = 0; t0 < np; t0 += 16) {
        float4 v[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) v[t] = reinterpret_cast<const float4*>(gp + (long long)min(t0 + t, np - 1) * ps)[i4];
#pragma unroll
        for (int t = 0; t < 16; ++t)
            if (t0 + t < np) { acc.x += v[t].x; acc.y += v[t].y; acc.z += v[t].z; acc.w += v[t].w; }
    }
    return acc;
}
__device__ __forceinline__ float part_sum1(const float* gp, int np, long long ps, long long i) {
    float acc = 0.f;
    for (int t = 0; t < np; ++t) acc += gp[t * ps + i];
    return acc;
}

// one thread's share of a vectorised pass: two float4 slots `stride` apart, every operand of both requested together
struct AdamVec {
    float* p; const float* g; float* m; float* v; float* target; const float* g2;
    const float* gp; int np; long long ps, pe;          // partial-sum gradient range (pe = 0: none)
    long long n4, stride;
    float* w2p; float* tw2p; long long w2_lo4, w2_hi4;  // fragment-order copies of the [256, 256] matrices in float4s [w2_lo4, w2_hi4)
};
// float4 index of parameter float4 `i4` (inside the W2 range that starts at float4 lo4) in the fragment-order copy
// (rrl_w2_pack, H = 256): element (g, row, col .. col + 3) -> ((n 16 + j) 64 + 16 q + i), n = row / 16, i = row % 16, j = col / 16,
// q = col / 4 % 4
__device__ __forceinline__ long long w2_frag4(long long i4, long long lo4) {
    const unsigned l = unsigned(i4 - lo4);             // < heads * 16384
    const unsigned gq = l >> 14, row = (l >> 6) & 255, c4 = l & 63;      // 64 float4 per row
    return (long long)gq * 16384 + ((row >> 4) * 16 + (c4 >> 2)) * 64 + (c4 & 3) * 16 + (row & 15);
}
struct AdamSlot {
    float4 P[2], G[2], M[2], V[2], T[2], Hh[2];
    long long i0, i1;
    bool two;
};
__device__ __forceinline__ void adam_slot_load(const AdamVec& a, long long i0, AdamSlot& s) {
    const float4* p4 = reinterpret_cast<const float4*>(a.p);
    const float4* m4 = reinterpret_cast<const float4*>(a.m);
    const float4* v4 = reinterpret_cast<const float4*>(a.v);
    const float4* t4 = reinterpret_cast<const float4*>(a.target);
    const float4* g4 = reinterpret_cast<const float4*>(a.g);
    const float4* h4 = reinterpret_cast<const float4*>(a.g2);
    s.i0 = i0;
    s.i1 = i0 + a.stride;
    s.two = s.i1 < a.n4;
    const long long j1 = s.two ? s.i1 : i0;
    s.P[0] = p4[i0]; s.P[1] = p4[j1];
    s.G[0] = g4[i0]; s.G[1] = g4[j1];
    s.M[0] = m4[i0]; s.M[1] = m4[j1];
    s.V[0] = v4[i0]; s.V[1] = v4[j1];
    if (4 * i0 < a.pe) s.G[0] = part_sum4(a.gp, a.np, a.ps, i0);          // pe % 4 == 0: a float4 is inside or outside
    if (4 * j1 < a.pe) s.G[1] = part_sum4(a.gp, a.np, a.ps, j1);
    s.T[0] = s.P[0]; s.T[1] = s.P[1];
    s.Hh[0] = s.Hh[1] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.target) { s.T[0] = t4[i0]; s.T[1] = t4[j1]; }
    if (a.g2) { s.Hh[0] = h4[i0]; s.Hh[1] = h4[j1]; }
}
__device__ __forceinline__ void adam_slot_finish(const AdamVec& a, AdamSlot& s, float step_size, float bc2_sqrt, float b1,
                                                 float b2, float eps, float tau, float wd) {
    float4* p4 = reinterpret_cast<float4*>(a.p);
    float4* m4 = reinterpret_cast<float4*>(a.m);
    float4* v4 = reinterpret_cast<float4*>(a.v);
    float4* t4 = reinterpret_cast<float4*>(a.target);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        float* pp = reinterpret_cast<float*>(&s.P[u]);
        float* mm = reinterpret_cast<float*>(&s.M[u]);
        float* vv = reinterpret_cast<float*>(&s.V[u]);
        float* tt = reinterpret_cast<float*>(&s.T[u]);
        const float* gg = reinterpret_cast<const float*>(&s.G[u]);
        const float* hh = reinterpret_cast<const float*>(&s.Hh[u]);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float gi = a.g2 ? gg[c] + hh[c] : gg[c];
            const float pi = adam_elem(pp[c], gi, mm[c], vv[c], step_size, bc2_sqrt, b1, b2, eps, wd);
            tt[c] = tt[c] * (1.f - tau) + pi * tau;
        }
    }
    p4[s.i0] = s.P[0]; m4[s.i0] = s.M[0]; v4[s.i0] = s.V[0];
    if (a.target) t4[s.i0] = s.T[0];
    if (s.two) {
        p4[s.i1] = s.P[1]; m4[s.i1] = s.M[1]; v4[s.i1] = s.V[1];
        if (a.target) t4[s.i1] = s.T[1];
    }
    // (uniform) the forward kernels' copy of W2 -- and of the target's -- in fragment order.  (Lanes relabelled inside the W2 range so
    // that four neighbours hold rows i .. i + 3 and store 64 contiguous bytes instead of 16 per line: measured, no difference.)
    if (a.w2p) {
        float4* w4 = reinterpret_cast<float4*>(a.w2p);
        float4* tw4 = reinterpret_cast<float4*>(a.tw2p);
        if (s.i0 >= a.w2_lo4 && s.i0 < a.w2_hi4) {
            const long long f = w2_frag4(s.i0, a.w2_lo4);
            w4[f] = s.P[0];
            if (a.tw2p) tw4[f] = s.T[0];
        }
        if (s.two && s.i1 >= a.w2_lo4 && s.i1 < a.w2_hi4) {
            const long long f = w2_frag4(s.i1, a.w2_lo4);
            w4[f] = s.P[1];
            if (a.tw2p) tw4[f] = s.T[1];
        }
    }
}

// `first` (optional): the thread's first slot, already loaded by the caller (adam_seg_body requests it BEFORE it waits for
// the step count and the bias corrections: they are needed by the arithmetic only)
__device__ __forceinline__ void adam_range(long long n, float* p, const float* g, float* m, float* v,
                                           float step_size, float bc2_sqrt, float b1, float b2, float eps,
                                           float* target, float tau, float wd, const float* g2, int block,
                                           int blocks, bool vec, const float* gp, int np,
                                           long long ps, long long pe, AdamSlot& first, bool have_first,
                                           float* w2p = nullptr, float* tw2p = nullptr, long long w2_off = 0, int w2_heads = 0) {
    const long long stride = (long long)blocks * kBlock;
    long long done = 0;
    if (!gp) pe = 0;
    if (vec) {
        const AdamVec a{p, g, m, v, target, g2, gp, np, ps, pe, n >> 2, stride,
                        w2p, target ? tw2p : nullptr, w2_off >> 2, (w2_off >> 2) + (long long)w2_heads * 16384};
        long long i0 = (long long)block * kBlock + threadIdx.x;
        if (have_first && i0 < a.n4) {
            adam_slot_finish(a, first, step_size, bc2_sqrt, b1, b2, eps, tau, wd);
            i0 += 2 * stride;
        }
        for (; i0 < a.n4; i0 += 2 * stride) {
            AdamSlot s;
            adam_slot_load(a, i0, s);
            adam_slot_finish(a, s, step_size, bc2_sqrt, b1, b2, eps, tau, wd);
        }
        done = a.n4 << 2;
    }
    for (long long i = done + (long long)block * kBlock + threadIdx.x; i < n; i += stride) {
        float pi = p[i], mi = m[i], vi = v[i];
        const float g0 = i < pe ? part_sum1(gp, np, ps, i) : g[i];
        const float gi = g2 ? g0 + g2[i] : g0;
        adam_elem(pi, gi, mi, vi, step_size, bc2_sqrt, b1, b2, eps, wd);
        p[i] = pi; m[i] = mi; v[i] = vi;
        if (target) target[i] = target[i] * (1.f - tau) + pi * tau;
    }
}

// several flat buffers (e.g. critic + policy of one update) in ONE launch; every segment keeps its own step
// counter, advanced by the last of ITS workgroups.
__device__ __forceinline__ void adam_seg_body(const rrl_adam_seg_t& sg, bool vec, float lr, float b1, float b2, float eps,
                                              int block, int blocks, float* sh) {
    // the step count is requested first, then the thread's first two float4 slots of every operand -- before anything waits
    // for the step count (the partial-sum gradients are added up inside the slot load: a wait of their own)
    const uint64_t step = sg.step_dev[0];        // every thread (thread 0 uses it): a load under a branch is waited for at its end
    AdamSlot first;
    const bool pre = vec && (long long)block * kBlock + threadIdx.x < (sg.n >> 2);
    if (pre) {
        const AdamVec av{sg.p, sg.g, sg.m, sg.v, sg.target, sg.g2, sg.g_part, sg.n_part, sg.part_stride,
                         sg.g_part ? sg.part_elems : 0, sg.n >> 2, (long long)blocks * kBlock, nullptr, nullptr, 0, 0};
        adam_slot_load(av, (long long)block * kBlock + threadIdx.x, first);
    }
    unsigned long long ticket = ~0ULL;
    if (threadIdx.x == 0) {
        // plain load (above), then the ticket once it has returned ("memory" clobber: the compiler keeps the order): the read
        // of the step count cannot slip behind the ticket
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const double t = double(step + 1);
        // the ticket is taken as soon as this workgroup has READ the step count: the last of the segment's workgroups
        // to do so knows every other one has read it too and may store t + 1 -- the returning atomic's round trip
        // (~0.7 us) runs under the parameter loads: its value is looked at after them
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // the step count has returned (no agent-scope release: that
        ticket = __hip_atomic_fetch_add((unsigned long long*)&sg.step_dev[1], 1ULL, __ATOMIC_RELAXED,   // is an L2 write-back)
                                        __HIP_MEMORY_SCOPE_AGENT);
        sh[0] = lr / float(1.0 - pow(double(b1), t));
        sh[2] = __uint_as_float(uint32_t(step));
    } else if (threadIdx.x == 64) {
        // the second bias correction on another wave: each double pow is a ~230-instruction dependent chain (~1 us) that the
        // whole workgroup waits for at the barrier below -- side by side instead of one behind the other
        sh[1] = float(sqrt(1.0 - pow(double(b2), double(step + 1))));
        sh[3] = __uint_as_float(uint32_t(step));
    }
    __syncthreads();
    // only wave 0's read of the step count is ordered before this workgroup's ticket: a slow wave 1 may have read the count
    // AFTER the segment's last workgroup stored t + 1.  Wave 0's value is the step's; if wave 1 saw another one it redoes its
    // correction with wave 0's (workgroup-uniform branch, never taken in practice)
    if (__float_as_uint(sh[2]) != __float_as_uint(sh[3])) {
        if (threadIdx.x == 0) sh[1] = float(sqrt(1.0 - pow(double(b2), double(step + 1))));
        __syncthreads();
    }
    adam_range(sg.n, sg.p, sg.g, sg.m, sg.v, sh[0], sh[1], b1, b2, eps, sg.target, sg.tau, sg.weight_decay, sg.g2,
               block, blocks, vec, sg.g_part, sg.n_part, sg.part_stride, sg.part_elems, first, vec, sg.w2p, sg.target_w2p,
               sg.w2_off, sg.w2_heads);
    if (threadIdx.x == 0 && ticket == (unsigned long long)blocks - 1) {
        sg.step_dev[0] = step + 1;
        sg.step_dev[1] = 0;
    }
}

// solo launch: grid (workgroups of the largest segment, segments) -- blockIdx.y IS the segment, its block arrives in one batch
// of scalar loads (the flat grid of the packed form walks first_block[] first: one more dependent round trip)
__global__ __launch_bounds__(kBlock) void adam_multi_kernel(AdamSegs a, int n_seg, float lr, float b1, float b2,
                                                            float eps) {
    __shared__ float sh[4];
    const int k = blockIdx.y;
    rrl_adam_seg_t sg = a.seg[k];
    const int vec = a.vec[k], blocks = a.first_block[k + 1] - a.first_block[k];
    rrl_pack::to_global_all(sg.p, sg.g, sg.m, sg.v, sg.step_dev, sg.target, sg.g2, sg.g_part, sg.w2p, sg.target_w2p);
    asm volatile("" ::"s"(sg.n), "s"(sg.tau), "s"(sg.weight_decay), "s"(sg.n_part), "s"(sg.part_stride), "s"(sg.part_elems),
                 "s"(vec), "s"(blocks), "s"(sg.w2_off), "s"(sg.w2_heads));
    if ((int)blockIdx.x >= blocks) return;
    adam_seg_body(sg, vec != 0, lr, b1, b2, eps, blockIdx.x, blocks, sh);
}

// the same launch for S seeds (pack.hpp): grid (workgroups of the seeds' largest segments under the XCD-aware placement,
// segments) -- blockIdx.y IS the segment and the seed follows from blockIdx.x by arithmetic, so the segment's block arrives in
// one batch of scalar loads from the plan's device copy, as in the solo launch
struct AdamPack {
    AdamSegs a;
    int n_seg;
    float lr, b1, b2, eps;
};
__global__ __launch_bounds__(kBlock) void adam_pack_kernel(const AdamPack* __restrict__ packs, rrl_pack::Idx ix) {
    __shared__ float sh[4];
    int s, local;
    asm volatile("" ::"s"(ix.sp), "s"(ix.p), "s"(ix.r), "s"(ix.S));
    if (!rrl_pack::locate_grid(ix, blockIdx.x, s, local)) return;
    const AdamPack& pk = packs[s];
    const int k = blockIdx.y;
    rrl_adam_seg_t sg = pk.a.seg[k];
    const int vec = pk.a.vec[k], blocks = pk.a.first_block[k + 1] - pk.a.first_block[k];
    const float lr = pk.lr, b1 = pk.b1, b2 = pk.b2, eps = pk.eps;
    const int n_seg = pk.n_seg;
    rrl_pack::to_global_all(sg.p, sg.g, sg.m, sg.v, sg.step_dev, sg.target, sg.g2, sg.g_part, sg.w2p, sg.target_w2p);
    asm volatile("" ::"s"(sg.n), "s"(sg.tau), "s"(sg.weight_decay), "s"(sg.n_part), "s"(sg.part_stride), "s"(sg.part_elems),
                 "s"(sg.w2_off), "s"(sg.w2_heads), "s"(vec), "s"(blocks), "s"(lr), "s"(b1), "s"(b2), "s"(eps), "s"(n_seg));
    if (k >= n_seg || local >= blocks) return;
    adam_seg_body(sg, vec != 0, lr, b1, b2, eps, local, blocks, sh);
}

// ---- N(0,1) fill: out[2i], out[2i+1] = the Philox normal pair of index i (stream RRL_STREAM_NOISE) ----
__global__ __launch_bounds__(kBlock) void normal_fill_kernel(long long n_pairs, uint64_t seed, uint64_t counter,
                                                             uint64_t* counter_dev, uint64_t counter_inc,
                                                             float* __restrict__ out) {
    const uint64_t ctr = rrl::effective_counter(counter, counter_dev);
    for (long long i = blockIdx.x * (long long)kBlock + threadIdx.x; i < n_pairs; i += (long long)gridDim.x * kBlock) {
        double z0, z1;
        rrl::normal_at(seed, uint32_t(i), rrl::kStreamNoise, ctr, z0, z1);
        reinterpret_cast<float2*>(out)[i] = make_float2(float(z0), float(z1));
    }
    rrl::advance_counter(counter_dev, counter_inc);
}

__global__ __launch_bounds__(kBlock) void adam_kernel(long long n, float* p, const float* g, float* m,
                                                      float* v, uint64_t* step_dev, float lr, float b1,
                                                      float b2, float eps, float* target, float tau, int vec) {
    __shared__ float sh[2];
    if (threadIdx.x == 0) {   // bias corrections once per workgroup (double pow is ~100 instructions)
        const double t = double(step_dev[0] + 1);
        sh[0] = lr / float(1.0 - pow(double(b1), t));
        sh[1] = float(sqrt(1.0 - pow(double(b2), t)));
    }
    __syncthreads();
    AdamSlot none;
    adam_range(n, p, g, m, v, sh[0], sh[1], b1, b2, eps, target, tau, 0.f, nullptr, blockIdx.x, gridDim.x, vec != 0, nullptr, 0, 0,
               0, none, false);
    rrl::advance_counter(step_dev, 1);
}

// ---- acting: recovery gate (experiment.py:546-577) ------------------------------------------------
// z [2,N] pre-sigmoid Q_risk(s, a_task); recovery = max(sigmoid) > eps_safe; real = recovery ? rec : task
__global__ void recovery_select_kernel(int N, const float* z, float eps_safe, const float* task_action,
                                       int ld_task, const float* rec_action, float* real_action,
                                       uint8_t* recovery, float* task_out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= N) return;
    const bool rec = fmaxf(sigmoidf(z[b]), sigmoidf(z[N + b])) > eps_safe;
    recovery[b] = uint8_t(rec);
    const float t0 = task_action[(long long)b * ld_task], t1 = task_action[(long long)b * ld_task + 1];
    real_action[2 * b] = rec ? rec_action[2 * b] : t0;
    real_action[2 * b + 1] = rec ? rec_action[2 * b + 1] : t1;
    if (task_out) { task_out[2 * b] = t0; task_out[2 * b + 1] = t1; }
}

inline dim3 rows_grid(int B) { return dim3((B + kBlock - 1) / kBlock); }
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }   // null counts as aligned

}  // namespace

extern "C" {

int rrl_gauss_head_fwd(int B, const float* head, int n_part, long long part_stride, const float* eps,
                       const float* scale, const float* bias, float* action, int ld_action, float* logp,
                       float* mean_action, const float* obs_in, float* obs_out, void* stream) {
    if (!head || !eps || !scale || !bias || !action || B <= 0 || n_part <= 0 || (obs_in && !obs_out))
        return RRL_EINVAL;
    hipLaunchKernelGGL(gauss_head_fwd_kernel, rows_grid(B), dim3(kBlock), 0, (hipStream_t)stream, B, head, n_part,
                       part_stride, eps, scale, bias, action, ld_action, logp, mean_action, obs_in, obs_out);
    return check_launch();
}

int rrl_policy_heads_fwd_multi(int n, const rrl_policy_head_t* heads, void* stream) {
    if (!heads || n <= 0 || n > kMaxHeads) return RRL_EINVAL;
    HeadGroup hg{};
    hg.n = n;
    hg.first[0] = 0;
    for (int k = 0; k < n; ++k) {
        const rrl_policy_head_t& h = heads[k];
        if (!h.head || !h.scale || !h.bias || !h.action || h.B <= 0 || h.n_part <= 0 || h.n_part > 4) return RRL_EINVAL;
        if (h.kind == RRL_HEAD_GAUSS) {
            if (!h.eps || (h.obs_in && !h.obs_out)) return RRL_EINVAL;
        } else if (h.kind == RRL_HEAD_STOCH) {
            if (!h.log_std) return RRL_EINVAL;
        } else {
            return RRL_EINVAL;
        }
        hg.h[k] = h;
        hg.first[k + 1] = hg.first[k] + (h.B + kBlock - 1) / kBlock;
    }
    for (int k = n; k < kMaxHeads; ++k) hg.first[k + 1] = hg.first[n];
    hipLaunchKernelGGL(policy_heads_group_kernel, dim3(hg.first[n]), dim3(kBlock), 0, (hipStream_t)stream, hg);
    return check_launch();
}

int rrl_gauss_head_bwd(int B, const float* head, int n_part, long long part_stride, const float* eps,
                       const float* scale, const float* d_action, int ld, int n_heads, long long head_stride,
                       float dlogp, float* dhead, void* stream) {
    if (!head || !eps || !scale || !d_action || !dhead || B <= 0 || n_heads <= 0 || n_part <= 0 || n_part > 4) return RRL_EINVAL;
    hipLaunchKernelGGL(gauss_head_bwd_kernel, rows_grid(B), dim3(kBlock), 0, (hipStream_t)stream, B, head, n_part,
                       part_stride, eps, scale, d_action, ld, n_heads, head_stride, dlogp, dhead);
    return check_launch();
}

int rrl_sac_critic_grad(int B, const float* q, const float* qt, int n_part, long long part_stride,
                        const float* logp2, const float* r, const float* m, float gamma, const float* alpha,
                        const float* penalty, float* dq, float* loss, void* stream) {
    if (!q || !qt || !logp2 || !r || !m || !alpha || !dq || B <= 0 || n_part <= 0 || n_part > 4) return RRL_EINVAL;
    hipLaunchKernelGGL(sac_critic_grad_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream, B, q, qt, n_part,
                       part_stride, logp2, r, m, gamma, alpha, penalty, dq, loss);
    return check_launch();
}

int rrl_sac_policy_grad(int B, const float* qp, int n_part, long long part_stride, const float* logp,
                        const float* alpha, float* dqp, float* loss, void* stream) {
    if (!qp || !logp || !alpha || !dqp || B <= 0 || n_part <= 0 || n_part > 4) return RRL_EINVAL;
    hipLaunchKernelGGL(sac_policy_grad_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream, B, qp, n_part,
                       part_stride, logp, alpha, dqp, loss);
    return check_launch();
}

int rrl_qrisk_critic_grad(int B, const float* z, const float* zt, int n_part, long long part_stride,
                          const float* c, const float* m, float gamma_safe, float* dz, float* loss, void* stream) {
    if (!z || !zt || !c || !m || !dz || B <= 0 || n_part <= 0 || n_part > 4) return RRL_EINVAL;
    hipLaunchKernelGGL(qrisk_critic_grad_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream, B, z, zt, n_part,
                       part_stride, c, m, gamma_safe, dz, loss);
    return check_launch();
}

int rrl_qrisk_policy_grad(int B, const float* zp, int n_part, long long part_stride, float* dzp, float* loss,
                          void* stream) {
    if (!zp || !dzp || B <= 0 || n_part <= 0 || n_part > 4) return RRL_EINVAL;
    hipLaunchKernelGGL(qrisk_policy_grad_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream, B, zp, n_part,
                       part_stride, dzp, loss);
    return check_launch();
}

int rrl_stoch_head_fwd(int B, const float* raw, int n_part, long long part_stride, const float* eps,
                       const float* log_std, float min_log_std, const float* scale, const float* bias,
                       float* action, int ld_action, float* mean_out, void* stream) {
    if (!raw || !log_std || !scale || !bias || !action || B <= 0 || n_part <= 0 || n_part > 4) return RRL_EINVAL;
    hipLaunchKernelGGL(stoch_head_fwd_kernel, rows_grid(B), dim3(kBlock), 0, (hipStream_t)stream, B, raw, n_part,
                       part_stride, eps, log_std, min_log_std, scale, bias, action, ld_action, mean_out);
    return check_launch();
}

int rrl_stoch_head_bwd(int B, const float* raw, int n_part, long long part_stride, const float* eps,
                       const float* log_std, float min_log_std, const float* scale, const float* d_action, int ld,
                       int n_heads, long long head_stride, float* draw, float* dlog_std, void* stream) {
    if (!raw || !eps || !log_std || !scale || !d_action || !draw || !dlog_std || B <= 0 || n_heads <= 0 ||
        n_part <= 0 || n_part > 4)
        return RRL_EINVAL;
    hipLaunchKernelGGL(stoch_head_bwd_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream, B, raw, n_part,
                       part_stride, eps, log_std, min_log_std, scale, d_action, ld, n_heads, head_stride, draw,
                       dlog_std);
    return check_launch();
}

int rrl_adam_step(long long n, float* p, const float* g, float* m, float* v, uint64_t* step_dev, float lr,
                  float beta1, float beta2, float eps, float* target, float tau, void* stream) {
    if (!p || !g || !m || !v || !step_dev || n <= 0) return RRL_EINVAL;
    // <= 64 workgroups: the step ticket is one device-scope atomic per workgroup on a single word
    const int grid = grid_for(n) < 64 ? grid_for(n) : 64;
    const int vec = aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v) && aligned16(target);
    hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, n, p, g, m, v,
                       step_dev, lr, beta1, beta2, eps, target, tau, vec);
    return check_launch();
}

static int build_adam_segs(int n_seg, const rrl_adam_seg_t* segs, AdamSegs& a) {
    if (!segs || n_seg <= 0 || n_seg > RRL_ADAM_MAX_SEGS) return RRL_EINVAL;
    a.first_block[0] = 0;
    for (int k = 0; k < n_seg; ++k) {
        const rrl_adam_seg_t& sg = segs[k];
        if (!sg.p || !sg.g || !sg.m || !sg.v || !sg.step_dev || sg.n <= 0) return RRL_EINVAL;
        if (sg.g_part && (sg.n_part <= 0 || sg.n_part > kMaxGradParts || sg.part_elems <= 0 || sg.part_elems > sg.n ||
                          (sg.part_elems & 3) || (sg.part_stride & 3) || !aligned16(sg.g_part)))
            return RRL_EINVAL;
        a.seg[k] = sg;
        a.vec[k] = aligned16(sg.p) && aligned16(sg.g) && aligned16(sg.m) && aligned16(sg.v) && aligned16(sg.target) &&
                   aligned16(sg.g2);
        if (sg.w2p) {        // the fragment-order copies ride on the vectorised pass: whole float4s of [256, 256] matrices
            if (!a.vec[k] || !aligned16(sg.w2p) || !aligned16(sg.target_w2p) || (sg.w2_off & 3) || sg.w2_off < 0 ||
                sg.w2_heads <= 0 || sg.w2_off + (long long)sg.w2_heads * 65536 > sg.n)
                return RRL_EINVAL;
        }
        // workgroups: 64 (the step ticket is one device-scope atomic per workgroup on a single word), up to 96 where that lets
        // every thread finish in its two float4 slots -- the twin critics' 134 658 parameters are 33 665 float4 = 2.05 per thread
        // of 64 workgroups, and the ~900 threads with a third slot added a whole dependent load -> store round trip to the launch
        const long long want = ((sg.n >> 2) + 2 * kBlock - 1) / (2 * kBlock);
        const int cap = int(std::min<long long>(std::max<long long>(64, want), 96));
        a.first_block[k + 1] = a.first_block[k] + (grid_for(sg.n) < cap ? grid_for(sg.n) : cap);
    }
    for (int k = n_seg; k < RRL_ADAM_MAX_SEGS; ++k) a.first_block[k + 1] = a.first_block[n_seg];
    return RRL_OK;
}

// ---- W2 in MFMA fragment order (rrl_stack_t.W2p): pure permutation, one float4 per thread ----
__global__ __launch_bounds__(kBlock) void w2_pack_kernel(int G, int H, const float* __restrict__ W2, float* __restrict__ W2p) {
    const int per_row = H / 4, J = H / 16;
    const long long n4 = (long long)G * H * per_row;
    for (long long e = blockIdx.x * (long long)kBlock + threadIdx.x; e < n4; e += (long long)gridDim.x * kBlock) {
        const int c4 = int(e % per_row), row = int((e / per_row) % H), g = int(e / ((long long)per_row * H));
        const long long f = (long long)g * H * per_row + ((long long)(row >> 4) * J + (c4 >> 2)) * 64 + (c4 & 3) * 16 + (row & 15);
        reinterpret_cast<float4*>(W2p)[f] = reinterpret_cast<const float4*>(W2)[e];
    }
}

int rrl_w2_pack(int G, int H, const float* W2, float* W2p, void* stream) {
    if (!W2 || !W2p || !aligned16(W2) || !aligned16(W2p)) return RRL_EINVAL;
    if (G <= 0 || H <= 0 || (H & 15)) return RRL_ERANGE;
    const long long n4 = (long long)G * H * (H / 4);
    const int grid = int(std::min<long long>((n4 + kBlock - 1) / kBlock, 1024));
    hipLaunchKernelGGL(w2_pack_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, G, H, W2, W2p);
    return check_launch();
}

int rrl_adam_step_multi(int n_seg, const rrl_adam_seg_t* segs, float lr, float beta1, float beta2, float eps,
                        void* stream) {
    AdamSegs a;
    const int rc = build_adam_segs(n_seg, segs, a);
    if (rc != RRL_OK) return rc;
    int most = 1;
    for (int k = 0; k < n_seg; ++k) most = a.first_block[k + 1] - a.first_block[k] > most ? a.first_block[k + 1] - a.first_block[k] : most;
    hipLaunchKernelGGL(adam_multi_kernel, dim3(most, n_seg), dim3(kBlock), 0, (hipStream_t)stream, a, n_seg,
                       lr, beta1, beta2, eps);
    return check_launch();
}

int rrl_adam_step_multi_packed(int S, const int* n_seg, const rrl_adam_seg_t* const* segs, const float* lr, float beta1,
                               float beta2, float eps, void* stream) {
    if (S <= 0 || S > rrl_pack::kMaxSeeds || !n_seg || !segs || !lr) return RRL_EINVAL;
    rrl_pack::Key key;
    key.pod(4);
    key.pod(S);
    key.pod(beta1); key.pod(beta2); key.pod(eps);
    for (int s = 0; s < S; ++s) {
        if (n_seg[s] <= 0 || n_seg[s] > RRL_ADAM_MAX_SEGS || !segs[s]) return RRL_EINVAL;
        key.pod(n_seg[s]);
        key.pod(lr[s]);
        key.add(segs[s], sizeof(rrl_adam_seg_t) * n_seg[s]);
    }
    // one seed: the packed launch IS the solo launch (argument block in the kernel arguments, no plan)
    if (S == 1) return rrl_adam_step_multi(n_seg[0], segs[0], lr[0], beta1, beta2, eps, stream);
    hipStream_t st = (hipStream_t)stream;
    rrl_pack::Plan* plan = rrl_pack::lookup(key);
    if (!plan) {
        std::vector<AdamPack> packs(S);
        rrl_pack::Idx ix;
        ix.S = S;
        ix.first[0] = 0;
        int segs_most = 1;
        for (int s = 0; s < S; ++s) {
            packs[s] = AdamPack{};
            const int rc = build_adam_segs(n_seg[s], segs[s], packs[s].a);
            if (rc != RRL_OK) return rc;
            packs[s].n_seg = n_seg[s];
            packs[s].lr = lr[s]; packs[s].b1 = beta1; packs[s].b2 = beta2; packs[s].eps = eps;
            int most = 1;                 // 2-D grid: a seed owns as many workgroups per segment row as its largest segment has
            for (int k = 0; k < n_seg[s]; ++k) most = std::max(most, packs[s].a.first_block[k + 1] - packs[s].a.first_block[k]);
            ix.first[s + 1] = ix.first[s] + most;
            segs_most = std::max(segs_most, n_seg[s]);
        }
        for (int s = S; s < rrl_pack::kMaxSeeds; ++s) ix.first[s + 1] = ix.first[S];
        plan = rrl_pack::store(key, packs.data(), sizeof(AdamPack) * S, st);
        if (!plan) return rrl_pack::store_error();
        plan->grid = rrl_pack::finish(ix);
        plan->ix = ix;
        plan->i0 = segs_most;
    }
    hipLaunchKernelGGL(adam_pack_kernel, dim3(plan->grid, plan->i0), dim3(kBlock), 0, st, (const AdamPack*)plan->dev, plan->ix);
    return check_launch();
}

int rrl_normal_fill(long long n_pairs, uint64_t seed, uint64_t counter, uint64_t* counter_dev, uint64_t counter_inc,
                    float* out, void* stream) {
    if (!out || n_pairs <= 0 || n_pairs >= (1LL << 32)) return RRL_EINVAL;
    hipLaunchKernelGGL(normal_fill_kernel, dim3(grid_for(n_pairs)), dim3(kBlock), 0, (hipStream_t)stream, n_pairs,
                       seed, counter, counter_dev, counter_inc, out);
    return check_launch();
}

int rrl_recovery_select(int N, const float* z, float eps_safe, const float* task_action, int ld_task,
                        const float* rec_action, float* real_action, uint8_t* recovery, float* task_out,
                        void* stream) {
    if (!z || !task_action || !rec_action || !real_action || !recovery || N <= 0) return RRL_EINVAL;
    hipLaunchKernelGGL(recovery_select_kernel, rows_grid(N), dim3(kBlock), 0, (hipStream_t)stream, N, z, eps_safe,
                       task_action, ld_task, rec_action, real_action, recovery, task_out);
    return check_launch();
}

}  // extern "C"
