// ens_train_kernels.hip -- one optimiser step of the PETS dynamics ensemble (MPC.train, recovery_rl/MPC.py:266-292:
// bootstrap batch gather, PtModel forward config/navigation1.py:71-96, Gaussian NLL + logvar-bound + weight-decay
// loss :276-287 / :52-59, backward) as ONE kernel per step; Adam is rrl_adam_step_multi on the same buffers.
//
// The PyTorch step is ~45 launches of tiny kernels (batch 32 x 5 nets x 200-wide layers = 39 MFLOP), i.e. bound by
// launch latency (0.12 ms per step even when replayed from a hipGraph); the re-fit after every episode is what
// dominates the wall-clock of model-based recovery (experiment.py:464-480).  Here two workgroups (1024 threads, 16
// batch rows each; the weight gradients of the halves are added by the Adam kernel) own one ensemble member for
// the whole step: activations and their gradients live in LDS (128 KB), the layer
// products run on the MFMA unit with weights read from L2, every reduction has a fixed order (deterministic).
//
// Shapes are the reference's: 4 inputs (obs 2 + action 2), 3 hidden layers of 200 with swish, 4 outputs (mean 2,
// logvar 2), batch 32.  Weight layout [net][in][out] (torch.baddbmm(b, x, w)).
#include <hip/hip_runtime.h>

#include "rrl_host.hpp"

namespace {

using rrl_host::check_launch;

constexpr int kH = 200;          // hidden width
constexpr int kB = 32;           // batch rows per net
constexpr int kR = 16;           // rows per workgroup: a member's batch is split over kHalves workgroups (CUs)
constexpr int kHalves = kB / kR;
constexpr int kDin = 4, kDout = 4;
constexpr int kThreads = 1024;
constexpr int kSmall = 3 * kH + kH * kDout + kDout;   // b0, b1, b2, W3, b3 staged in LDS with the first loads
constexpr int kLdsFloats = 5 * kR * kH + kR * (kDin + 2 + kDout + kDout) + 16 + kSmall;
constexpr int kLdsBytes = kLdsFloats * 4;      // 71 KB

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float softplus(float x) { return x > 20.f ? x : log1pf(expf(x)); }   // F.softplus
__device__ __forceinline__ float softplus_grad(float x) { return x > 20.f ? 1.f : sigm(x); }

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

constexpr int kColTiles = (kH + 15) / 16;     // 13 (200 = 12.5 x 16: the last tile is half masked)

// The three products of a layer are 16x16x4 f32 MFMA tiles (one ensemble member = one CU: an LDS-fed VALU
// formulation was LDS-issue bound, 134 us per step).  MFMA step s uses k = 4 s + lane / 16; C layout: row =
// 4 (lane / 16) + i, col = lane % 16.
//
// Work split of every product: wave w < 13 owns the 16-wide strip w of the 200-wide dimension for the workgroup's 16
// rows, so each weight element is requested once per workgroup and a layer costs ONE L2 round trip per wave (the
// 1024-thread workgroup caps a lane at 128 VGPRs: 50 dwords / 13 float4 of weights in flight fit, twice that spills).
//
// forward: pre[32 x 200] = in[32 x K] . W[K x 200] + b;  h = swish(pre) -> LDS, swish'(pre) -> global scratch
template <int K>
__device__ __forceinline__ void fwd_layer(const float* in, int ld_in, const float* __restrict__ W,
                                          const float* b /* LDS */, float* h_out, float* __restrict__ sp, int tid) {
    const int lane = tid & 63, wave = tid >> 6, lr = lane & 15, lq = lane >> 4;
    if (wave >= kColTiles) return;
    constexpr int S = K / 4;
    const int col = wave * 16 + lr;
    const bool cok = col < kH;
    const float* bcol = W + (size_t)lq * kH + (cok ? col : kH - 1);
    float bv[S];
#pragma unroll
    for (int s = 0; s < S; ++s) bv[s] = bcol[(size_t)4 * s * kH];
    const float* a0 = in + lr * ld_in + lq;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};      // two chains: 40-cycle MFMA latency
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const float w = cok ? bv[s] : 0.f;
        if (s & 1) acc1 = mfma(a0[4 * s], w, acc1);
        else acc0 = mfma(a0[4 * s], w, acc0);
    }
    acc0 += acc1;
    if (cok) {
        const float bj = b[col];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 4 * lq + i;
            const float p = acc0[i] + bj, sg = sigm(p);
            h_out[row * kH + col] = p * sg;
            sp[row * kH + col] = sg * (1.f + p * (1.f - sg));      // d swish / d pre
        }
    }
}

// gW[K x 200] = h_in^T[K x 32] . dpre[32 x 200];  gb[j] = sum_r dpre[r][j]
// (the weight-decay terms of the loss, config/navigation1.py:52-59, are added by the Adam kernel as
// weight_decay * W: the same gradient without re-reading the weights here)
// wave w: output columns 16 w .. 16 w + 15, all ceil(K / 16) row tiles; the dpre operand is read once
template <int K>
__device__ __forceinline__ void grad_weights(const float* h_in, int ld_in, const float* dpre,
                                             float* __restrict__ gW, float* __restrict__ gb, int tid) {
    const int lane = tid & 63, wave = tid >> 6, lr = lane & 15, lq = lane >> 4;
    if (wave < kColTiles) {
        const int n = wave * 16 + lr;
        const bool nok = n < kH;
        float bvv[kR / 4];
#pragma unroll
        for (int s = 0; s < kR / 4; ++s) bvv[s] = nok ? dpre[(4 * s + lq) * kH + n] : 0.f;   // contraction over rows
        constexpr int mtiles = (K + 15) / 16;
#pragma unroll 1
        for (int mt = 0; mt < mtiles; ++mt) {
            const int m = mt * 16 + lr;
            const bool mok = m < K;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < kR / 4; ++s) acc = mfma(mok ? h_in[(4 * s + lq) * ld_in + m] : 0.f, bvv[s], acc);
            if (nok) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = mt * 16 + 4 * lq + i;
                    if (row < K) gW[row * kH + n] = acc[i];
                }
            }
        }
    }
    if (tid >= kThreads - 256 && tid - (kThreads - 256) < kH) {       // the otherwise idle waves 12..15
        const int j = tid - (kThreads - 256);
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < kR; ++r) sum += dpre[r * kH + j];
        gb[j] = sum;
    }
}

// dprev[32 x 200] = (dpre[32 x 200] . W^T)[r][k] * sp_prev[r][k]        (contraction over the layer's outputs j)
// B[j][k] = W[k][j]: a lane owns row k of W, contiguous along j.  Reading it one dword per MFMA step touched every
// 128-byte line eight times from sixteen waves (L1 thrash: 45 us per layer); with the K order permuted inside chunks
// of 16 (step t of chunk c uses j = 16 c + 4 (lane / 16) + t on BOTH operands) a lane reads its row as 13 float4 and
// the activations as matching ds_read_b128.  prefetch() issues the loads; the caller runs the LDS-only
// weight-gradient tiles before compute() consumes them.
struct GradInput {
    static constexpr int chunks = (kH + 15) / 16;     // 13; the last one holds j = 192 .. 199 (lane groups 0, 1)
    f32x4 bv[chunks];
    float spv[4];
    bool kok;

    __device__ __forceinline__ void prefetch(const float* __restrict__ W, const float* __restrict__ sp_prev, int tid) {
        const int lane = tid & 63, wave = tid >> 6, lr = lane & 15, lq = lane >> 4;
        const int k = wave * 16 + lr;
        kok = wave < kColTiles && k < kH;
        const int kc = kok ? k : kH - 1;
        const float* brow = W + (size_t)kc * kH + 4 * lq;
#pragma unroll
        for (int c = 0; c < chunks; ++c)
            bv[c] = (16 * c + 4 * lq < kH) ? *reinterpret_cast<const f32x4*>(brow + 16 * c) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) spv[i] = sp_prev[(4 * lq + i) * kH + kc];
    }

    __device__ __forceinline__ void compute(const float* dpre, float* dprev, int tid) const {
        const int lane = tid & 63, wave = tid >> 6, lr = lane & 15, lq = lane >> 4;
        if (wave >= kColTiles) return;
        const int k = wave * 16 + lr;
        const float* a0 = dpre + lr * kH + 4 * lq;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < chunks; ++c) {
            const bool jok = 16 * c + 4 * lq < kH;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            const f32x4 av0 = jok ? *reinterpret_cast<const f32x4*>(a0 + 16 * c) : z;
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                const float w = kok ? bv[c][tt] : 0.f;
                if (tt & 1) acc1 = mfma(av0[tt], w, acc1);
                else acc0 = mfma(av0[tt], w, acc0);
            }
        }
        acc0 += acc1;
        if (kok) {
#pragma unroll
            for (int i = 0; i < 4; ++i) dprev[(4 * lq + i) * kH + k] = acc0[i] * spv[i];
        }
    }
};

__global__ __launch_bounds__(kThreads) void ens_train_grad_kernel(rrl_ens_t m, const float* __restrict__ train_in,
                                                                  const float* __restrict__ train_targ,
                                                                  const int64_t* __restrict__ idx,
                                                                  long long idx_stride, int nb,
                                                                  float* __restrict__ scratch,
                                                                  float* __restrict__ loss_out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* h0 = lds;                       // [16][200] swish outputs of the three hidden layers
    float* h1 = h0 + kR * kH;
    float* h2 = h1 + kR * kH;
    float* da = h2 + kR * kH;              // gradient w.r.t. pre-activations, ping
    float* db = da + kR * kH;              // pong
    float* xin = db + kR * kH;             // [16][4] standardised inputs
    float* yt = xin + kR * kDin;           // [16][2] targets
    float* out = yt + kR * 2;              // [16][4]
    float* dout = out + kR * kDout;        // [16][4]
    float* red = dout + kR * kDout;        // [16]
    float* bs = red + 16;                  // b0 | b1 | b2  [3][200]
    float* w3s = bs + 3 * kH;              // W3 [200][4]
    float* b3s = w3s + kH * kDout;         // [4]

    const int e = blockIdx.x / kHalves, half = blockIdx.x % kHalves, tid = threadIdx.x;
    const int part = e * kHalves + half;
#ifdef RRL_ENS_TIMING      // phase stamps (s_memtime) of workgroup 0 in the tail of the scratch buffer (profiles/)
    long long* stamps = reinterpret_cast<long long*>(scratch + (size_t)gridDim.x * 3 * kR * kH);
    int n_stamp = 0;
#define STAMP() do { if (blockIdx.x == 0 && tid == 0) stamps[n_stamp++] = __builtin_readcyclecounter(); } while (0)
#else
#define STAMP() do { } while (0)
#endif
    STAMP();
    const float* W0 = m.w0 + (size_t)e * kDin * kH;
    const float* b0 = m.b0 + (size_t)e * kH;
    const float* W1 = m.w1 + (size_t)e * kH * kH;
    const float* b1 = m.b1 + (size_t)e * kH;
    const float* W2 = m.w2 + (size_t)e * kH * kH;
    const float* b2 = m.b2 + (size_t)e * kH;
    const float* W3 = m.w3 + (size_t)e * kH * kDout;
    const float* b3 = m.b3 + (size_t)e * kDout;
    float* sp = scratch + (size_t)part * 3 * kR * kH;   // swish' of the three hidden layers (L2-resident)
    // gradient outputs of this half (the Adam kernel adds the halves)
    float* const gw0 = (half ? m.g2_w0 : m.g_w0) + (size_t)e * kDin * kH;
    float* const gb0 = (half ? m.g2_b0 : m.g_b0) + (size_t)e * kH;
    float* const gw1 = (half ? m.g2_w1 : m.g_w1) + (size_t)e * kH * kH;
    float* const gb1 = (half ? m.g2_b1 : m.g_b1) + (size_t)e * kH;
    float* const gw2 = (half ? m.g2_w2 : m.g_w2) + (size_t)e * kH * kH;
    float* const gb2 = (half ? m.g2_b2 : m.g_b2) + (size_t)e * kH;
    float* const gw3 = (half ? m.g2_w3 : m.g_w3) + (size_t)e * kH * kDout;
    float* const gb3 = (half ? m.g2_b3 : m.g_b3) + (size_t)e * kDout;

    // ---- bootstrap rows of this net, standardised (config/navigation1.py:72) ----
    // the small parameters ride in the same round trip as the batch gather
    for (int i = tid; i < kSmall; i += kThreads) {
        float v;
        if (i < kH) v = b0[i];
        else if (i < 2 * kH) v = b1[i - kH];
        else if (i < 3 * kH) v = b2[i - 2 * kH];
        else if (i < 3 * kH + kH * kDout) v = W3[i - 3 * kH];
        else v = b3[i - 3 * kH - kH * kDout];
        bs[i] = v;
    }
    // nb <= 32 rows are real (the last batch of an epoch is shorter); the others contribute zero gradient
    if (tid < kR * kDin) {
        const int rl = tid / kDin, k = tid % kDin, r = half * kR + rl;      // r = row of the member's batch
        const int64_t row = r < nb ? idx[(size_t)e * idx_stride + r] : 0;
        xin[tid] = r < nb ? (train_in[row * kDin + k] - m.mu[k]) / m.sigma[k] : 0.f;
        if (k < 2) yt[rl * 2 + k] = r < nb ? train_targ[row * 2 + k] : 0.f;
    }
    __syncthreads();
    STAMP();
    fwd_layer<kDin>(xin, kDin, W0, bs, h0, sp, tid);
    __syncthreads();
    STAMP();
    fwd_layer<kH>(h0, kH, W1, bs + kH, h1, sp + kR * kH, tid);
    __syncthreads();
    STAMP();
    fwd_layer<kH>(h1, kH, W2, bs + 2 * kH, h2, sp + 2 * kR * kH, tid);
    __syncthreads();
    STAMP();
    // ---- output layer (4 wide) ----
    if (tid < kR * kDout) {
        const int r = tid / kDout, o = tid % kDout;
        float acc = b3s[o];
#pragma unroll 8
        for (int k = 0; k < kH; ++k) acc = fmaf(h2[r * kH + k], w3s[k * kDout + o], acc);
        out[tid] = acc;
    }
    __syncthreads();
    STAMP();
    // ---- loss (MPC.py:276-287) and its gradient w.r.t. the outputs; one thread per (row, dim) ----
    if (tid < 2 * kR) {
        const int r = tid >> 1, k = tid & 1;
        const float mx = m.max_logvar[k], mn = m.min_logvar[k];
        const float mean = out[r * kDout + k], lv0 = out[r * kDout + 2 + k];
        const float a1 = mx - lv0, lv1 = mx - softplus(a1);
        const float a2 = lv1 - mn, lv2 = mn + softplus(a2);
        const float inv = expf(-lv2), diff = mean - yt[r * 2 + k];
        const float live = half * kR + r < nb ? 1.f : 0.f;
        float tl = (diff * diff * inv + lv2) * live;
        const float scale = live / float(nb * 2);                // mean over the real rows and the two dims
        const float d_lv2 = (1.f - diff * diff * inv) * scale;
        const float s2 = softplus_grad(a2), d_lv1 = d_lv2 * s2;
        const float s1 = softplus_grad(a1);
        dout[r * kDout + k] = 2.f * diff * inv * scale;
        dout[r * kDout + 2 + k] = d_lv1 * s1;
        float d_min = d_lv2 * (1.f - s2), d_max = d_lv1 * (1.f - s1);
        // fixed-order reductions over the 16 rows (lanes 0..31 with the same parity)
#pragma unroll
        for (int off = 2; off < 2 * kR; off <<= 1) {
            tl += __shfl_xor(tl, off);
            d_min += __shfl_xor(d_min, off);
            d_max += __shfl_xor(d_max, off);
        }
        if (tid < 2) {
            red[k] = tl / float(nb * 2);
            m.g_logvar_part[part * 4 + k] = d_max;
            m.g_logvar_part[part * 4 + 2 + k] = d_min;
        }
    }
    __syncthreads();
    STAMP();
    if (tid == 0) m.loss_part[part] = red[0] + red[1];
    // ---- backward: output layer ----
    {
        // gW3[k][o], gb3[o]
        if (tid < kH * kDout) {
            const int k = tid / kDout, o = tid % kDout;
            float acc = 0.f;
#pragma unroll
            for (int r = 0; r < kR; ++r) acc = fmaf(h2[r * kH + k], dout[r * kDout + o], acc);
            gw3[tid] = acc;
        } else if (tid < kH * kDout + kDout) {
            const int o = tid - kH * kDout;
            float s = 0.f;
            for (int r = 0; r < kR; ++r) s += dout[r * kDout + o];
            gb3[o] = s;
        }
        // dpre2[r][k] = sp2[r][k] * sum_o dout[r][o] W3[k][o]
        const int k = tid & 255, r0 = (tid >> 8) * 4;
        if (k < kH) {
            const float4 w = *reinterpret_cast<const float4*>(w3s + k * kDout);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float4 d = *reinterpret_cast<const float4*>(dout + (r0 + r) * kDout);
                da[(r0 + r) * kH + k] = (d.x * w.x + d.y * w.y + d.z * w.z + d.w * w.w) *
                                        sp[2 * kR * kH + (r0 + r) * kH + k];
            }
        }
    }
    __syncthreads();
    STAMP();
    // ---- hidden layer 2: grads of W2/b2 from (h1, da), then dpre1 -> db ----
    {
        GradInput gi;
        gi.prefetch(W2, sp + kR * kH, tid);           // the L2 round trip hides behind the weight-gradient tiles
        grad_weights<kH>(h1, kH, da, gw2, gb2, tid);
        gi.compute(da, db, tid);
    }
    __syncthreads();
    STAMP();
    // ---- hidden layer 1 ----
    {
        GradInput gi;
        gi.prefetch(W1, sp, tid);
        grad_weights<kH>(h0, kH, db, gw1, gb1, tid);
        gi.compute(db, da, tid);
    }
    __syncthreads();
    STAMP();
    // ---- input layer ----
    grad_weights<kDin>(xin, kDin, da, gw0, gb0, tid);
    __syncthreads();
    STAMP();
#undef STAMP
}

// g_max_logvar[k] = 0.01 + sum_parts part[.][k];  g_min_logvar[k] = -0.01 + sum_parts part[.][2 + k]  (MPC.py:271);
// loss_out[e] = the member's NLL = sum of its halves
__global__ void ens_logvar_grad_kernel(int n_nets, const float* __restrict__ part, const float* __restrict__ loss_part,
                                       float* __restrict__ g_max, float* __restrict__ g_min,
                                       float* __restrict__ loss_out) {
    const int k = threadIdx.x;
    if (k < 4) {
        float s = 0.f;
        for (int p = 0; p < n_nets * kHalves; ++p) s += part[p * 4 + k];
        if (k < 2) g_max[k] = 0.01f + s;
        else g_min[k - 2] = -0.01f + s;
    } else if (loss_out && k - 4 < n_nets) {
        const int e = k - 4;
        float s = 0.f;
        for (int h = 0; h < kHalves; ++h) s += loss_part[e * kHalves + h];
        loss_out[e] = s;
    }
}

}  // namespace

extern "C" {

int rrl_ens_train_supported(int d_in, int hidden, int d_out, int batch) {
    return d_in == kDin && hidden == kH && d_out == kDout && batch >= 1 && batch <= kB;
}

long long rrl_ens_scratch_floats(int n_nets) { return (long long)n_nets * kHalves * 3 * kR * kH + 64; }   // + stamps

int rrl_ens_train_grad(const rrl_ens_t* m, int batch, const float* train_in, const float* train_targ,
                       const int64_t* idx, long long idx_stride, float* scratch, float* loss_out, void* stream) {
    if (!m || !train_in || !train_targ || !idx || !scratch || m->n_nets <= 0) return RRL_EINVAL;
    if (!rrl_ens_train_supported(m->d_in, m->hidden, m->d_out, batch)) return RRL_ERANGE;
    if (!m->w0 || !m->b0 || !m->w1 || !m->b1 || !m->w2 || !m->b2 || !m->w3 || !m->b3 || !m->max_logvar ||
        !m->min_logvar || !m->mu || !m->sigma || !m->g_w0 || !m->g_b0 || !m->g_w1 || !m->g_b1 || !m->g_w2 ||
        !m->g_b2 || !m->g_w3 || !m->g_b3 || !m->g2_w0 || !m->g2_b0 || !m->g2_w1 || !m->g2_b1 || !m->g2_w2 || !m->g2_b2 ||
        !m->g2_w3 || !m->g2_b3 || !m->g_max_logvar || !m->g_min_logvar || !m->g_logvar_part || !m->loss_part ||
        m->n_nets > 60)
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
    hipLaunchKernelGGL(ens_train_grad_kernel, dim3(m->n_nets * kHalves), dim3(kThreads), kLdsBytes, st, *m, train_in, train_targ,
                       idx, idx_stride, batch, scratch, loss_out);
    hipLaunchKernelGGL(ens_logvar_grad_kernel, dim3(1), dim3(64), 0, st, m->n_nets, m->g_logvar_part, m->loss_part,
                       m->g_max_logvar, m->g_min_logvar, loss_out);
    return check_launch();
}

// One epoch of MPC.train's batch loop (MPC.py:266-292) issued from C: ceil(n_rows / batch) x {gradient kernel,
// logvar reduction, Adam}.  Launching from a C loop costs ~4 us per launch on the host instead of ~20 us through
// Python, which keeps the 35-us optimiser step GPU-bound.
int rrl_ens_train_epoch(const rrl_ens_t* m, int n_seg, const rrl_adam_seg_t* segs, float lr, float beta1, float beta2,
                        float eps, const float* train_in, const float* train_targ, const int64_t* idx,
                        long long idx_stride, long long n_rows, int batch, float* scratch, float* loss_out,
                        void* stream) {
    if (!idx || n_rows <= 0 || batch <= 0 || !segs) return RRL_EINVAL;
    for (long long lo = 0; lo < n_rows; lo += batch) {
        const int nb = int(n_rows - lo < batch ? n_rows - lo : batch);
        int rc = rrl_ens_train_grad(m, nb, train_in, train_targ, idx + lo, idx_stride, scratch, loss_out, stream);
        if (rc != RRL_OK) return rc;
        rc = rrl_adam_step_multi(n_seg, segs, lr, beta1, beta2, eps, stream);
        if (rc != RRL_OK) return rc;
    }
    return RRL_OK;
}

}  // extern "C"
