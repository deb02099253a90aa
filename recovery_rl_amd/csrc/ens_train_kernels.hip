// ens_train_kernels.hip -- one optimiser step of the PETS dynamics ensemble (MPC.train, recovery_rl/MPC.py:266-292:
// bootstrap batch gather, PtModel forward config/navigation1.py:71-96, Gaussian NLL + logvar-bound + weight-decay
// loss :276-287 / :52-59, backward) as ONE kernel per step; Adam is rrl_adam_step_multi on the same buffers.
//
// The PyTorch step is ~45 launches of tiny kernels (batch 32 x 5 nets x 200-wide layers = 39 MFLOP), i.e. bound by
// launch latency (0.12 ms per step even when replayed from a hipGraph); the re-fit after every episode is what
// dominates the wall-clock of model-based recovery (experiment.py:464-480).  Here one workgroup (1024 threads) owns
// one ensemble member for the whole step: activations and their gradients live in LDS (128 KB), weights are read
// from L2 with coalesced accesses, every reduction has a fixed order (deterministic).
//
// Shapes are the reference's: 4 inputs (obs 2 + action 2), 3 hidden layers of 200 with swish, 4 outputs (mean 2,
// logvar 2), batch 32.  Weight layout [net][in][out] (torch.baddbmm(b, x, w)).
#include <hip/hip_runtime.h>

#include "rrl_host.hpp"

namespace {

using rrl_host::check_launch;

constexpr int kH = 200;          // hidden width
constexpr int kB = 32;           // batch rows per net
constexpr int kDin = 4, kDout = 4;
constexpr int kThreads = 1024;
constexpr int kLdsFloats = 5 * kB * kH + kB * (kDin + 2 + kDout + kDout) + 16;
constexpr int kLdsBytes = kLdsFloats * 4;      // 130 KB

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float softplus(float x) { return x > 20.f ? x : log1pf(expf(x)); }   // F.softplus
__device__ __forceinline__ float softplus_grad(float x) { return x > 20.f ? 1.f : sigm(x); }

// pre[r][j] = b[j] + sum_k in[r][k] W[k][j]; writes h = swish(pre) to LDS and swish'(pre) to the global scratch.
// thread -> column j = tid % 256 (< 200), row group tid / 256 (8 rows)
template <int K>
__device__ __forceinline__ void fwd_layer(const float* in, int ld_in, const float* __restrict__ W,
                                          const float* __restrict__ b, float* h_out, float* __restrict__ sp, int tid) {
    const int j = tid & 255, r0 = (tid >> 8) * 8;
    if (j >= kH) return;
    float acc[8];
    const float bj = b[j];
#pragma unroll
    for (int r = 0; r < 8; ++r) acc[r] = bj;
#pragma unroll 8
    for (int k = 0; k < K; ++k) {
        const float w = W[k * kH + j];
#pragma unroll
        for (int r = 0; r < 8; ++r) acc[r] = fmaf(in[(r0 + r) * ld_in + k], w, acc[r]);
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const float p = acc[r], s = sigm(p);
        h_out[(r0 + r) * kH + j] = p * s;
        sp[(r0 + r) * kH + j] = s * (1.f + p * (1.f - s));      // d swish / d pre
    }
}

// gW[k][j] = sum_r h_in[r][k] dpre[r][j] + decay W[k][j];  gb[j] = sum_r dpre[r][j]
// thread -> column j = tid % 256, k range = (tid / 256) * K/4 ...
template <int K>
__device__ __forceinline__ void grad_weights(const float* h_in, int ld_in, const float* dpre,
                                             const float* __restrict__ W, float decay, float* __restrict__ gW,
                                             float* __restrict__ gb, int tid) {
    const int j = tid & 255, kg = tid >> 8;
    if (j >= kH) return;
    float d[kB];
#pragma unroll
    for (int r = 0; r < kB; ++r) d[r] = dpre[r * kH + j];
    constexpr int per = (K + 3) / 4;
    for (int k = kg * per; k < min(K, (kg + 1) * per); ++k) {
        float acc = 0.f;
#pragma unroll
        for (int r = 0; r < kB; ++r) acc = fmaf(h_in[r * ld_in + k], d[r], acc);
        gW[k * kH + j] = acc + decay * W[k * kH + j];
    }
    if (kg == 0) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < kB; ++r) s += d[r];
        gb[j] = s;
    }
}

// dprev[r][k] = sp_prev[r][k] * sum_j dpre[r][j] W[k][j]     (W row k contiguous: float4 along j)
// thread -> k = tid % 256 (< 200), row group tid / 256 (8 rows)
__device__ __forceinline__ void grad_input(const float* dpre, const float* __restrict__ W, const float* __restrict__ sp_prev,
                                           float* dprev, int tid) {
    const int k = tid & 255, r0 = (tid >> 8) * 8;
    if (k >= kH) return;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float4* wrow = reinterpret_cast<const float4*>(W + k * kH);
#pragma unroll 5
    for (int jj = 0; jj < kH / 4; ++jj) {
        const float4 w = wrow[jj];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const float4 d = *reinterpret_cast<const float4*>(dpre + (r0 + r) * kH + 4 * jj);
            acc[r] = fmaf(d.x, w.x, fmaf(d.y, w.y, fmaf(d.z, w.z, fmaf(d.w, w.w, acc[r]))));
        }
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) dprev[(r0 + r) * kH + k] = acc[r] * sp_prev[(r0 + r) * kH + k];
}

__global__ __launch_bounds__(kThreads) void ens_train_grad_kernel(rrl_ens_t m, const float* __restrict__ train_in,
                                                                  const float* __restrict__ train_targ,
                                                                  const int64_t* __restrict__ idx,
                                                                  long long idx_stride, int nb,
                                                                  float* __restrict__ scratch,
                                                                  float* __restrict__ loss_out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* h0 = lds;                       // [32][200] swish outputs of the three hidden layers
    float* h1 = h0 + kB * kH;
    float* h2 = h1 + kB * kH;
    float* da = h2 + kB * kH;              // gradient w.r.t. pre-activations, ping
    float* db = da + kB * kH;              // pong
    float* xin = db + kB * kH;             // [32][4] standardised inputs
    float* yt = xin + kB * kDin;           // [32][2] targets
    float* out = yt + kB * 2;              // [32][4]
    float* dout = out + kB * kDout;        // [32][4]
    float* red = dout + kB * kDout;        // [16]

    const int e = blockIdx.x, tid = threadIdx.x;
    const float* W0 = m.w0 + (size_t)e * kDin * kH;
    const float* b0 = m.b0 + (size_t)e * kH;
    const float* W1 = m.w1 + (size_t)e * kH * kH;
    const float* b1 = m.b1 + (size_t)e * kH;
    const float* W2 = m.w2 + (size_t)e * kH * kH;
    const float* b2 = m.b2 + (size_t)e * kH;
    const float* W3 = m.w3 + (size_t)e * kH * kDout;
    const float* b3 = m.b3 + (size_t)e * kDout;
    float* sp = scratch + (size_t)e * 3 * kB * kH;      // swish' of the three hidden layers (L2-resident)

    // ---- bootstrap rows of this net, standardised (config/navigation1.py:72) ----
    // nb <= 32 rows are real (the last batch of an epoch is shorter); the others contribute zero gradient
    if (tid < kB * kDin) {
        const int r = tid / kDin, k = tid % kDin;
        const int64_t row = r < nb ? idx[(size_t)e * idx_stride + r] : 0;
        xin[tid] = r < nb ? (train_in[row * kDin + k] - m.mu[k]) / m.sigma[k] : 0.f;
        if (k < 2) yt[r * 2 + k] = r < nb ? train_targ[row * 2 + k] : 0.f;
    }
    __syncthreads();
    fwd_layer<kDin>(xin, kDin, W0, b0, h0, sp, tid);
    __syncthreads();
    fwd_layer<kH>(h0, kH, W1, b1, h1, sp + kB * kH, tid);
    __syncthreads();
    fwd_layer<kH>(h1, kH, W2, b2, h2, sp + 2 * kB * kH, tid);
    __syncthreads();
    // ---- output layer (4 wide) ----
    if (tid < kB * kDout) {
        const int r = tid / kDout, o = tid % kDout;
        float acc = b3[o];
        for (int k = 0; k < kH; ++k) acc = fmaf(h2[r * kH + k], W3[k * kDout + o], acc);
        out[tid] = acc;
    }
    __syncthreads();
    // ---- loss (MPC.py:276-287) and its gradient w.r.t. the outputs; one thread per (row, dim) ----
    if (tid < 64) {
        const int r = tid >> 1, k = tid & 1;
        const float mx = m.max_logvar[k], mn = m.min_logvar[k];
        const float mean = out[r * kDout + k], lv0 = out[r * kDout + 2 + k];
        const float a1 = mx - lv0, lv1 = mx - softplus(a1);
        const float a2 = lv1 - mn, lv2 = mn + softplus(a2);
        const float inv = expf(-lv2), diff = mean - yt[r * 2 + k];
        const float live = r < nb ? 1.f : 0.f;
        float tl = (diff * diff * inv + lv2) * live;
        const float scale = live / float(nb * 2);                // mean over the real rows and the two dims
        const float d_lv2 = (1.f - diff * diff * inv) * scale;
        const float s2 = softplus_grad(a2), d_lv1 = d_lv2 * s2;
        const float s1 = softplus_grad(a1);
        dout[r * kDout + k] = 2.f * diff * inv * scale;
        dout[r * kDout + 2 + k] = d_lv1 * s1;
        float d_min = d_lv2 * (1.f - s2), d_max = d_lv1 * (1.f - s1);
        // fixed-order reductions over the 32 rows (lanes with the same parity)
#pragma unroll
        for (int off = 2; off < 64; off <<= 1) {
            tl += __shfl_xor(tl, off);
            d_min += __shfl_xor(d_min, off);
            d_max += __shfl_xor(d_max, off);
        }
        if (tid < 2) {
            red[k] = tl / float(nb * 2);
            m.g_logvar_part[e * 4 + k] = d_max;
            m.g_logvar_part[e * 4 + 2 + k] = d_min;
        }
    }
    __syncthreads();
    if (tid == 0 && loss_out) loss_out[e] = red[0] + red[1];
    // ---- backward: output layer ----
    {
        // gW3[k][o], gb3[o]
        if (tid < kH * kDout) {
            const int k = tid / kDout, o = tid % kDout;
            float acc = 0.f;
#pragma unroll
            for (int r = 0; r < kB; ++r) acc = fmaf(h2[r * kH + k], dout[r * kDout + o], acc);
            m.g_w3[(size_t)e * kH * kDout + tid] = acc + 0.00075f * W3[tid];
        } else if (tid < kH * kDout + kDout) {
            const int o = tid - kH * kDout;
            float s = 0.f;
            for (int r = 0; r < kB; ++r) s += dout[r * kDout + o];
            m.g_b3[e * kDout + o] = s;
        }
        // dpre2[r][k] = sp2[r][k] * sum_o dout[r][o] W3[k][o]
        const int k = tid & 255, r0 = (tid >> 8) * 8;
        if (k < kH) {
            const float4 w = *reinterpret_cast<const float4*>(W3 + k * kDout);
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float4 d = *reinterpret_cast<const float4*>(dout + (r0 + r) * kDout);
                da[(r0 + r) * kH + k] = (d.x * w.x + d.y * w.y + d.z * w.z + d.w * w.w) *
                                        sp[2 * kB * kH + (r0 + r) * kH + k];
            }
        }
    }
    __syncthreads();
    // ---- hidden layer 2: grads of W2/b2 from (h1, da), then dpre1 -> db ----
    grad_weights<kH>(h1, kH, da, W2, 0.0005f, m.g_w2 + (size_t)e * kH * kH, m.g_b2 + (size_t)e * kH, tid);
    grad_input(da, W2, sp + kB * kH, db, tid);
    __syncthreads();
    // ---- hidden layer 1 ----
    grad_weights<kH>(h0, kH, db, W1, 0.0005f, m.g_w1 + (size_t)e * kH * kH, m.g_b1 + (size_t)e * kH, tid);
    grad_input(db, W1, sp, da, tid);
    __syncthreads();
    // ---- input layer ----
    grad_weights<kDin>(xin, kDin, da, W0, 0.00025f, m.g_w0 + (size_t)e * kDin * kH, m.g_b0 + (size_t)e * kH, tid);
}

// g_max_logvar[k] = 0.01 + sum_e part[e][k];  g_min_logvar[k] = -0.01 + sum_e part[e][2 + k]     (MPC.py:271)
__global__ void ens_logvar_grad_kernel(int n_nets, const float* __restrict__ part, float* __restrict__ g_max,
                                       float* __restrict__ g_min) {
    const int k = threadIdx.x;
    if (k >= 4) return;
    float s = 0.f;
    for (int e = 0; e < n_nets; ++e) s += part[e * 4 + k];
    if (k < 2) g_max[k] = 0.01f + s;
    else g_min[k - 2] = -0.01f + s;
}

}  // namespace

extern "C" {

int rrl_ens_train_supported(int d_in, int hidden, int d_out, int batch) {
    return d_in == kDin && hidden == kH && d_out == kDout && batch >= 1 && batch <= kB;
}

long long rrl_ens_scratch_floats(int n_nets) { return (long long)n_nets * 3 * kB * kH; }

int rrl_ens_train_grad(const rrl_ens_t* m, int batch, const float* train_in, const float* train_targ,
                       const int64_t* idx, long long idx_stride, float* scratch, float* loss_out, void* stream) {
    if (!m || !train_in || !train_targ || !idx || !scratch || m->n_nets <= 0) return RRL_EINVAL;
    if (!rrl_ens_train_supported(m->d_in, m->hidden, m->d_out, batch)) return RRL_ERANGE;
    if (!m->w0 || !m->b0 || !m->w1 || !m->b1 || !m->w2 || !m->b2 || !m->w3 || !m->b3 || !m->max_logvar ||
        !m->min_logvar || !m->mu || !m->sigma || !m->g_w0 || !m->g_b0 || !m->g_w1 || !m->g_b1 || !m->g_w2 ||
        !m->g_b2 || !m->g_w3 || !m->g_b3 || !m->g_max_logvar || !m->g_min_logvar || !m->g_logvar_part)
        return RRL_EINVAL;
    static bool lds_set = false;
    if (!lds_set) {
        if (hipFuncSetAttribute((const void*)ens_train_grad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                kLdsBytes) != hipSuccess) {
            rrl_host::last_hip_error = int(hipGetLastError());
            return RRL_ELAUNCH;
        }
        lds_set = true;
    }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(ens_train_grad_kernel, dim3(m->n_nets), dim3(kThreads), kLdsBytes, st, *m, train_in, train_targ,
                       idx, idx_stride, batch, scratch, loss_out);
    hipLaunchKernelGGL(ens_logvar_grad_kernel, dim3(1), dim3(64), 0, st, m->n_nets, m->g_logvar_part,
                       m->g_max_logvar, m->g_min_logvar);
    return check_launch();
}

}  // extern "C"
