// ens_train_big_kernels.hip -- one optimiser step of the PETS dynamics ensemble at LARGE batch (the lock-step loop's
// online re-fit: batch 32 x num_envs rows per member, 131 072 at 4096 envs; MPC.train, recovery_rl/MPC.py:250-298, with
// PtModel.forward config/navigation1.py:71-96 and the loss :276-287) on the f32 matrix pipe.  Same mathematics, same
// ABI struct (rrl_ens_t) and the same Adam launch as the batch-32 kernel of ens_train_kernels.hip; that one keeps a
// member's whole batch on one CU pair and is launch-latency work, this one is 320 GFLOP per step and MFMA-bound.
//
//   A  ens_big_fwd_bwd_kernel : workgroup (member e, part p) walks 64-row tiles p, p + P, ... of the member's bootstrap
//                               batch: gather + standardise, three hidden layers forward (swish), output layer, Gaussian
//                               NLL with the softplus log-variance bounds, backward down to dpre0 -- activations live in
//                               LDS ([64][204] f32 x 3), every 64 x 200 x 200 product is v_mfma_f32_16x16x4_f32 (exact f32);
//                               what the weight-gradient pass needs (h0, h1, dpre1, dpre2) and what the backward re-reads
//                               (swish'(pre0), swish'(pre1)) goes to HBM, 4.8 KB written + 1.6 KB read per row; the narrow
//                               gradients (W0 [4 x 200], W3 [200 x 4], biases, log-variance bounds, loss) are accumulated
//                               in registers over the workgroup's tiles and leave as ONE partial per workgroup;
//   B  ens_big_wgrad_kernel   : gW1 = h0^T dpre1, gW2 = h1^T dpre2 as split-K TN products: workgroup (layer, member, chunk q)
//                               streams its rows straight from HBM into MFMA operands (no LDS: both operands are
//                               k-contiguous in memory) and keeps the 13 x 13 output tiles in registers (16 waves in a
//                               4 x 4 arrangement placed so that every SIMD issues 42-43 of the 169 tiles);
//   C  ens_big_reduce_kernel  : partials -> gradient tensors in a FIXED order (deterministic), logvar-bound gradients
//                               (+-0.01, MPC.py:271) and the per-member loss.
// Weight-decay terms (config/navigation1.py:52-59) are added by rrl_adam_step_multi (weight_decay of the segment), as for
// the batch-32 kernel.
#include <hip/hip_runtime.h>

#include "rrl_host.hpp"

namespace {

using rrl_host::check_launch;

constexpr int kH = 200, kDin = 4, kDout = 4;
constexpr int kR = 64;               // rows per tile
constexpr int kS = 204;              // LDS row stride (floats): row r starts at bank 12 r mod 32 -> conflict-free b128 reads
constexpr int kThreadsA = 512;       // 8 waves
constexpr int kBuf = kR * kS;
constexpr int kLdsFloatsA = 3 * kBuf + kR * (kDin + 2 + kDout + kDout) + 16;
constexpr int kLdsBytesA = kLdsFloatsA * 4;          // 160 320 B (of 163 840)
constexpr int kPartA = 2304;         // floats per workgroup partial: gW0 800 | gW3 800 | gb0 200 | gb1 200 | gb2 200 | gb3 4 | lv 4 | loss 1
constexpr int kOffW0 = 0, kOffW3 = 800, kOffB0 = 1600, kOffB1 = 1800, kOffB2 = 2000, kOffB3 = 2200, kOffLv = 2204,
              kOffLoss = 2208;
constexpr int kThreadsB = 1024;      // 16 waves
constexpr int kMaxP = 64, kMaxQ = 32;

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// Identity the optimiser cannot see through: address arithmetic derived from opaque(lane) is redone per phase instead of
// being hoisted out of the tile loop and kept live (hoisted, the per-element addresses of the epilogues cost > 200 VGPRs)
__device__ __forceinline__ int opaque(int x) {
    asm volatile("" : "+v"(x));
    return x;
}
__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float softplus(float x) { return x > 20.f ? x : log1pf(expf(x)); }   // F.softplus
__device__ __forceinline__ float softplus_grad(float x) { return x > 20.f ? 1.f : sigm(x); }

// A wave's share of a [64 x 200] = [64 x 200] . [200 x 200] product: row tiles {2 rh, 2 rh + 1} x column tiles {cs, cs + 4,
// cs + 8} (rh = wave & 1, cs = wave >> 1), plus tile (row tile `wave`, column tile 12) on waves 0..3 -- 7 or 6 of the 52
// tiles per wave, the same count on every SIMD.
struct Acc {
    f32x4 main[2][3];
    f32x4 extra;
};

// C = in . W (TRANS = false, W[k][n] row-major: the forward) or C = in . W^T (TRANS = true, W[n][k]: the backward's
// input gradient).  K order permuted inside chunks of 16 (MFMA step t of chunk j uses k = 16 j + 4 (lane / 16) + t on both
// operands), so the LDS operand is one ds_read_b128 per row tile and chunk, and the TRANS weight operand one 16-byte load.
template <bool TRANS>
__device__ __forceinline__ void wave_gemm(const float* in, const float* __restrict__ W, int wave, int lane, Acc& acc) {
    const int lr = lane & 15, lq = lane >> 4;
    const int rh = wave & 1, cs = wave >> 1;
    const bool has_extra = wave < 4;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c) acc.main[a][c] = z;
    acc.extra = z;
    const float* a0p = in + (16 * (2 * rh) + lr) * kS + 4 * lq;
    const float* a1p = a0p + 16 * kS;
    const float* axp = in + (16 * (wave & 3) + lr) * kS + 4 * lq;
    const int nx = 192 + lr;
    const bool nxok = has_extra && nx < kH;
    auto load_b = [&](int j, f32x4 (&b)[3], f32x4& bx) {
        const int kb = 16 * j + 4 * lq;
        const bool kok = kb < kH;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int n = 16 * (cs + 4 * c) + lr;
            if (TRANS) {
                b[c] = kok ? *reinterpret_cast<const f32x4*>(W + (size_t)n * kH + kb) : z;
            } else {
                f32x4 v = z;
                if (kok) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[t] = W[(size_t)(kb + t) * kH + n];
                }
                b[c] = v;
            }
        }
        bx = z;
        if (nxok && kok) {
            if (TRANS) {
                bx = *reinterpret_cast<const f32x4*>(W + (size_t)nx * kH + kb);
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) bx[t] = W[(size_t)(kb + t) * kH + nx];
            }
        }
    };
    f32x4 b[3], bx, bn[3], bxn;
    load_b(0, b, bx);
    constexpr int chunks = (kH + 15) / 16;     // 13; the last one holds k = 192 .. 199 (lane groups 0 and 1)
    // not unrolled: unrolled, the scheduler hoists every chunk's weight loads to the top (13 x 16 dwords: spills)
#pragma unroll 1
    for (int j = 0; j < chunks; ++j) {
        if (j + 1 < chunks) load_b(j + 1, bn, bxn);          // the next chunk's weights are in flight under this chunk's MFMAs
        const bool kok = 16 * j + 4 * lq < kH;
        const f32x4 a0 = kok ? *reinterpret_cast<const f32x4*>(a0p + 16 * j) : z;
        const f32x4 a1 = kok ? *reinterpret_cast<const f32x4*>(a1p + 16 * j) : z;
        f32x4 ax = z;
        if (has_extra && kok) ax = *reinterpret_cast<const f32x4*>(axp + 16 * j);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                acc.main[0][c] = mfma(a0[t], b[c][t], acc.main[0][c]);
                acc.main[1][c] = mfma(a1[t], b[c][t], acc.main[1][c]);
            }
            if (has_extra) acc.extra = mfma(ax[t], bx[t], acc.extra);
        }
        if (j + 1 < chunks) {
#pragma unroll
            for (int c = 0; c < 3; ++c) b[c] = bn[c];
            bx = bxn;
        }
    }
}

// f(row in tile, column, value) for every element of the wave's tiles (C layout: row = 4 (lane / 16) + i, col = lane % 16)
template <class F>
__device__ __forceinline__ void for_each_elem(const Acc& acc, int wave, int lane, F f) {
    const int lr = lane & 15, lq = lane >> 4;
    const int rh = wave & 1, cs = wave >> 1;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) f(16 * (2 * rh + a) + 4 * lq + i, 16 * (cs + 4 * c) + lr, acc.main[a][c][i]);
    if (wave < 4 && 192 + lr < kH) {
#pragma unroll
        for (int i = 0; i < 4; ++i) f(16 * wave + 4 * lq + i, 192 + lr, acc.extra[i]);
    }
}

struct BigArgs {
    rrl_ens_t m;
    const float* train_in;
    const float* train_targ;
    const int64_t* idx;
    long long idx_stride;
    long long nb;          // real rows per member
    long long rows_pad;    // nb rounded up to a multiple of 64
    int P;                 // workgroups per member (kernel A)
    int Q;                 // row chunks per (layer, member) (kernel B)
    float *h0, *h1, *sp0, *sp1, *d1, *d2;      // [E][rows_pad][200]
    float* partA;          // [E][P][kPartA]
    float* partB;          // [2][E][Q][200*200]
};

__global__ __launch_bounds__(kThreadsA) void ens_big_fwd_bwd_kernel(BigArgs g) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* bufA = lds;
    float* bufB = bufA + kBuf;
    float* bufC = bufB + kBuf;
    float* xin = bufC + kBuf;              // [64][4] standardised inputs
    float* yt = xin + kR * kDin;           // [64][2] targets
    float* outb = yt + kR * 2;             // [64][4] network outputs
    float* doutb = outb + kR * kDout;      // [64][4] loss gradient w.r.t. the outputs
    float* redb = doutb + kR * kDout;      // [2 waves][2 dims][4] loss / logvar-bound sums of the tile
    const rrl_ens_t& m = g.m;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int e = blockIdx.x / g.P, p = blockIdx.x % g.P;
    const float* W0 = m.w0 + (size_t)e * kDin * kH;
    const float* b0 = m.b0 + (size_t)e * kH;
    const float* W1 = m.w1 + (size_t)e * kH * kH;
    const float* b1 = m.b1 + (size_t)e * kH;
    const float* W2 = m.w2 + (size_t)e * kH * kH;
    const float* b2 = m.b2 + (size_t)e * kH;
    const float* W3 = m.w3 + (size_t)e * kH * kDout;
    const float* b3 = m.b3 + (size_t)e * kDout;
    const size_t mem_off = (size_t)e * g.rows_pad * kH;
    float* const h0g = g.h0 + mem_off;
    float* const h1g = g.h1 + mem_off;
    float* const sp0g = g.sp0 + mem_off;
    float* const sp1g = g.sp1 + mem_off;
    float* const d1g = g.d1 + mem_off;
    float* const d2g = g.d2 + mem_off;

    // per-thread running sums over this workgroup's tiles (fixed order: tile by tile, row by row)
    float gw0_acc[2] = {0.f, 0.f}, gw3_acc[2] = {0.f, 0.f};
    float gb_acc[3] = {0.f, 0.f, 0.f};      // gb0, gb1, gb2 of column tid (tid < 200)
    float gb3_acc = 0.f;                    // gb3[tid] (tid < 4)
    float lv_acc = 0.f, loss_acc = 0.f;     // threads 0..3: d max_logvar[0..1], d min_logvar[0..1]; thread 0: loss

    const long long n_tiles = g.rows_pad / kR;
    const int tid_k = tid, lane_k = lane;
    for (long long tile = p; tile < n_tiles; tile += g.P) {
        const long long row0 = tile * kR;
        int tid = opaque(tid_k), lane = opaque(lane_k);
        // ---- bootstrap rows of this member, standardised (config/navigation1.py:72) ----
        if (tid < kR * kDin) {
            const int rl = tid / kDin, k = tid % kDin;
            const long long r = row0 + rl;
            const bool live = r < g.nb;
            const int64_t row = live ? g.idx[(size_t)e * g.idx_stride + r] : 0;
            xin[tid] = live ? (g.train_in[row * kDin + k] - m.mu[k]) / m.sigma[k] : 0.f;
            if (k < 2) yt[rl * 2 + k] = live ? g.train_targ[row * 2 + k] : 0.f;
        }
        __syncthreads();
        // ---- layer 0 (K = 4: one MFMA step per tile) -> h0 in A ----
        {
            const int lr = lane & 15, lq = lane >> 4;
            const int rh = wave & 1, cs = wave >> 1;
            Acc acc;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const float av = xin[(16 * (2 * rh + a) + lr) * kDin + lq];
#pragma unroll
                for (int c = 0; c < 3; ++c) acc.main[a][c] = mfma(av, W0[lq * kH + 16 * (cs + 4 * c) + lr], z);
            }
            acc.extra = z;
            if (wave < 4) {
                const int nx = 192 + lr;
                acc.extra = mfma(xin[(16 * wave + lr) * kDin + lq], nx < kH ? W0[lq * kH + nx] : 0.f, z);
            }
            for_each_elem(acc, wave, lane, [&](int r, int n, float v) {
                const float pre = v + b0[n], sg = sigm(pre), h = pre * sg;
                bufA[r * kS + n] = h;
                h0g[(size_t)(row0 + r) * kH + n] = h;
                sp0g[(size_t)(row0 + r) * kH + n] = sg * (1.f + pre * (1.f - sg));      // d swish / d pre
            });
        }
        __syncthreads();
        tid = opaque(tid_k); lane = opaque(lane_k);
        // ---- layer 1: h0 (A) -> h1 (B) ----
        {
            Acc acc;
            wave_gemm<false>(bufA, W1, wave, lane, acc);
            for_each_elem(acc, wave, lane, [&](int r, int n, float v) {
                const float pre = v + b1[n], sg = sigm(pre), h = pre * sg;
                bufB[r * kS + n] = h;
                h1g[(size_t)(row0 + r) * kH + n] = h;
                sp1g[(size_t)(row0 + r) * kH + n] = sg * (1.f + pre * (1.f - sg));
            });
        }
        __syncthreads();
        tid = opaque(tid_k); lane = opaque(lane_k);
        // ---- layer 2: h1 (B) -> h2 (A), swish' (C) ----
        {
            Acc acc;
            wave_gemm<false>(bufB, W2, wave, lane, acc);
            for_each_elem(acc, wave, lane, [&](int r, int n, float v) {
                const float pre = v + b2[n], sg = sigm(pre);
                bufA[r * kS + n] = pre * sg;
                bufC[r * kS + n] = sg * (1.f + pre * (1.f - sg));
            });
        }
        __syncthreads();
        tid = opaque(tid_k); lane = opaque(lane_k);
        // ---- output layer (4 wide): two threads per output, 100 terms each, fixed order ----
        {
            const int o_idx = tid >> 1, half = tid & 1;          // o_idx = r * 4 + o
            const int r = o_idx >> 2, o = o_idx & 3;
            float acc = 0.f;
            const float* hr = bufA + r * kS + half * (kH / 2);
            const float* wr = W3 + (size_t)half * (kH / 2) * kDout + o;
#pragma unroll 10
            for (int k = 0; k < kH / 2; ++k) acc = fmaf(hr[k], wr[k * kDout], acc);
            const float other = __shfl_xor(acc, 1, 64);
            if (half == 0) outb[o_idx] = (acc + other) + b3[o];
        }
        __syncthreads();
        tid = opaque(tid_k); lane = opaque(lane_k);
        // ---- loss (MPC.py:276-287) and its gradient w.r.t. the outputs; one thread per (row, dim) ----
        if (tid < 2 * kR) {
            const int r = tid >> 1, k = tid & 1;
            const float mx = m.max_logvar[k], mn = m.min_logvar[k];
            const float mean = outb[r * kDout + k], lv0 = outb[r * kDout + 2 + k];
            const float a1 = mx - lv0, lv1 = mx - softplus(a1);
            const float a2 = lv1 - mn, lv2 = mn + softplus(a2);
            const float inv = expf(-lv2), diff = mean - yt[r * 2 + k];
            const float live = row0 + r < g.nb ? 1.f : 0.f;
            float tl = (diff * diff * inv + lv2) * live;
            const float scale = live / float(g.nb * 2);              // mean over the real rows and the two dims
            const float d_lv2 = (1.f - diff * diff * inv) * scale;
            const float s2 = softplus_grad(a2), d_lv1 = d_lv2 * s2;
            const float s1 = softplus_grad(a1);
            doutb[r * kDout + k] = 2.f * diff * inv * scale;
            doutb[r * kDout + 2 + k] = d_lv1 * s1;
            float d_min = d_lv2 * (1.f - s2), d_max = d_lv1 * (1.f - s1);
            // fixed-order reductions over the 64 rows: lanes of equal parity inside each wave, then the two waves
#pragma unroll
            for (int off = 2; off < 64; off <<= 1) {
                tl += __shfl_xor(tl, off, 64);
                d_min += __shfl_xor(d_min, off, 64);
                d_max += __shfl_xor(d_max, off, 64);
            }
            if ((tid & 63) < 2) {          // lanes 0, 1 of waves 0 and 1 hold their wave's sums for dim k
                float* rw = redb + (tid >> 6) * 8 + k * 4;
                rw[0] = tl;
                rw[1] = d_max;
                rw[2] = d_min;
            }
        }
        __syncthreads();
        if (tid < 4) {
            // tid = 0, 1: d max_logvar[k]; tid = 2, 3: d min_logvar[k]; wave 0's rows first, then wave 1's
            const int k = tid & 1, which = tid < 2 ? 1 : 2;
            lv_acc += redb[k * 4 + which] + redb[8 + k * 4 + which];
            if (tid == 0) loss_acc += ((redb[0] + redb[8]) + (redb[4] + redb[12])) / float(g.nb * 2);
        }
        tid = opaque(tid_k); lane = opaque(lane_k);
        // ---- backward: output layer.  gW3 += h2^T dout, gb3 += colsum(dout), dpre2 = (dout W3^T) * swish'(pre2) in C ----
        {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int i = tid + q * kThreadsA;
                if (i < kH * kDout) {
                    const int k = i / kDout, o = i % kDout;
                    float s = 0.f;
#pragma unroll 8
                    for (int r = 0; r < kR; ++r) s = fmaf(bufA[r * kS + k], doutb[r * kDout + o], s);
                    gw3_acc[q] += s;
                }
            }
            if (tid < kDout) {
                float s = 0.f;
                for (int r = 0; r < kR; ++r) s += doutb[r * kDout + tid];
                gb3_acc += s;
            }
            for (int i = tid; i < kR * kH; i += kThreadsA) {
                const int r = i / kH, k = i % kH;
                const float4 d = *reinterpret_cast<const float4*>(doutb + r * kDout);
                const float4 w = *reinterpret_cast<const float4*>(W3 + (size_t)k * kDout);
                const float v = (d.x * w.x + d.y * w.y + d.z * w.z + d.w * w.w) * bufC[r * kS + k];
                bufC[r * kS + k] = v;
                d2g[(size_t)(row0 + r) * kH + k] = v;
            }
        }
        __syncthreads();
        if (tid < kH) {
            float s = 0.f;
#pragma unroll 8
            for (int r = 0; r < kR; ++r) s += bufC[r * kS + tid];
            gb_acc[2] += s;
        }
        tid = opaque(tid_k); lane = opaque(lane_k);
        // ---- hidden layer 2: dpre1 = (dpre2 W2^T) * swish'(pre1) -> A ----
        {
            Acc acc;
            wave_gemm<true>(bufC, W2, wave, lane, acc);
            for_each_elem(acc, wave, lane, [&](int r, int n, float v) {
                const float d = v * sp1g[(size_t)(row0 + r) * kH + n];
                bufA[r * kS + n] = d;
                d1g[(size_t)(row0 + r) * kH + n] = d;
            });
        }
        __syncthreads();
        if (tid < kH) {
            float s = 0.f;
#pragma unroll 8
            for (int r = 0; r < kR; ++r) s += bufA[r * kS + tid];
            gb_acc[1] += s;
        }
        tid = opaque(tid_k); lane = opaque(lane_k);
        // ---- hidden layer 1: dpre0 = (dpre1 W1^T) * swish'(pre0) -> B ----
        {
            Acc acc;
            wave_gemm<true>(bufA, W1, wave, lane, acc);
            for_each_elem(acc, wave, lane, [&](int r, int n, float v) {
                bufB[r * kS + n] = v * sp0g[(size_t)(row0 + r) * kH + n];
            });
        }
        __syncthreads();
        tid = opaque(tid_k); lane = opaque(lane_k);
        // ---- input layer: gW0 += x^T dpre0, gb0 += colsum(dpre0) ----
        if (tid < kH) {
            float s = 0.f;
#pragma unroll 8
            for (int r = 0; r < kR; ++r) s += bufB[r * kS + tid];
            gb_acc[0] += s;
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int i = tid + q * kThreadsA;
            if (i < kDin * kH) {
                const int k = i / kH, n = i % kH;
                float s = 0.f;
#pragma unroll 8
                for (int r = 0; r < kR; ++r) s = fmaf(xin[r * kDin + k], bufB[r * kS + n], s);
                gw0_acc[q] += s;
            }
        }
        __syncthreads();       // the next tile overwrites xin / the buffers
    }
    float* part = g.partA + ((size_t)e * g.P + p) * kPartA;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int i = tid + q * kThreadsA;
        if (i < kDin * kH) part[kOffW0 + i] = gw0_acc[q];
        if (i < kH * kDout) part[kOffW3 + i] = gw3_acc[q];
    }
    if (tid < kH) {
        part[kOffB0 + tid] = gb_acc[0];
        part[kOffB1 + tid] = gb_acc[1];
        part[kOffB2 + tid] = gb_acc[2];
    }
    if (tid < kDout) {
        part[kOffB3 + tid] = gb3_acc;
        part[kOffLv + tid] = lv_acc;
    }
    if (tid == 0) part[kOffLoss] = loss_acc;
}

// gW[k][n] = sum over the chunk's rows of h[r][k] * d[r][n]: 13 x 13 output tiles over 16 waves.  Wave w sits at (a, b)
// of a 4 x 4 arrangement with b chosen so that w % 4 = (a + b) % 4: the row-tile sets {a, a + 4, a + 8, 12 if a == 0} and
// the column-tile sets of the four waves of a SIMD then add up to 43 / 42 / 42 / 42 tiles.  Rows arrive in groups of 32
// through LDS (coalesced 16-byte loads of the next group are in flight under the current group's MFMAs; row stride 208
// floats: the two 16-lane rows a half-wave reads land on disjoint banks).
constexpr int kGB = 32;              // rows per LDS group
constexpr int kSB = 208;             // LDS row stride
constexpr int kLdsBytesB = 2 * 2 * kGB * kSB * 4;     // 106 496 B
constexpr int kVecB = (2 * kGB * kH / 4 + kThreadsB - 1) / kThreadsB;    // float4 per thread and group: 4 (3200 / 1024)

__global__ __launch_bounds__(kThreadsB) void ens_big_wgrad_kernel(BigArgs g) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, lq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int a = wave >> 2, b = ((wave & 3) - a) & 3;
    const int q = blockIdx.x % g.Q;
    const int e = (blockIdx.x / g.Q) % g.m.n_nets;
    const int layer = blockIdx.x / (g.Q * g.m.n_nets);       // 0: gW1 = h0^T dpre1;  1: gW2 = h1^T dpre2
    const size_t mem_off = (size_t)e * g.rows_pad * kH;
    const float* __restrict__ hsrc = (layer ? g.h1 : g.h0) + mem_off;
    const float* __restrict__ dsrc = (layer ? g.d2 : g.d1) + mem_off;
    const long long groups = g.rows_pad / kGB;
    const long long g_lo = groups * q / g.Q, g_hi = groups * (q + 1) / g.Q;
    const int na = a == 0 ? 4 : 3, nbt = b == 0 ? 4 : 3;
    int mcol[4], ncol[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        mcol[i] = 16 * (a + 4 * i) + lr;      // i = 3 only for a == 0 (tile 12: columns 200..207 are the zeroed pad)
        ncol[i] = 16 * (b + 4 * i) + lr;
        if (i >= na) mcol[i] = 0;
        if (i >= nbt) ncol[i] = 0;
    }
    // pad columns 200..207 of both operands stay zero: tile 12 multiplies them
    for (int i = tid; i < 2 * 2 * kGB * 8; i += kThreadsB) lds[(i >> 3) * kSB + kH + (i & 7)] = 0.f;
    f32x4 acc[4][4];
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = z;
    // element v of a group: operand (v / 1600), row (v % 1600) / 50, float4 (v % 50)
    f32x4 pre[kVecB];
    auto fetch = [&](long long grp) {
#pragma unroll
        for (int u = 0; u < kVecB; ++u) {
            const int v = tid + u * kThreadsB;
            if (v < 2 * kGB * kH / 4) {
                const int op = v / (kGB * kH / 4), rem = v % (kGB * kH / 4);
                const float* src = (op ? dsrc : hsrc) + (size_t)grp * kGB * kH;
                pre[u] = *reinterpret_cast<const f32x4*>(src + (size_t)rem * 4);
            }
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int u = 0; u < kVecB; ++u) {
            const int v = tid + u * kThreadsB;
            if (v < 2 * kGB * kH / 4) {
                const int op = v / (kGB * kH / 4), rem = v % (kGB * kH / 4);
                const int row = rem / (kH / 4), c4 = rem % (kH / 4);
                *reinterpret_cast<f32x4*>(lds + ((buf * 2 + op) * kGB + row) * kSB + 4 * c4) = pre[u];
            }
        }
    };
    if (g_lo < g_hi) {
        fetch(g_lo);
        stash(0);
    }
    __syncthreads();
    int buf = 0;
    for (long long grp = g_lo; grp < g_hi; ++grp) {
        if (grp + 1 < g_hi) fetch(grp + 1);
        const float* hs = lds + (buf * 2 + 0) * kGB * kSB + lq * kSB;
        const float* ds = lds + (buf * 2 + 1) * kGB * kSB + lq * kSB;
#pragma unroll 2
        for (int s = 0; s < kGB / 4; ++s) {
            float av[4], bv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                av[i] = hs[4 * s * kSB + mcol[i]];
                bv[i] = ds[4 * s * kSB + ncol[i]];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (i < 3 || a == 0)
                        if (j < 3 || b == 0) acc[i][j] = mfma(av[i], bv[j], acc[i][j]);
        }
        if (grp + 1 < g_hi) stash(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    float* out = g.partB + (((size_t)layer * g.m.n_nets + e) * g.Q + q) * (size_t)(kH * kH);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (i >= na || j >= nbt) continue;
            const int n = 16 * (b + 4 * j) + lr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = 16 * (a + 4 * i) + 4 * lq + r;
                if (k < kH && n < kH) out[(size_t)k * kH + n] = acc[i][j][r];
            }
        }
}

// partials -> gradients, fixed order
__global__ __launch_bounds__(256) void ens_big_reduce_kernel(BigArgs g, float* __restrict__ loss_out) {
    const rrl_ens_t& m = g.m;
    const int E = m.n_nets;
    const long long n_big = 2LL * E * kH * kH;
    const long long n_small = (long long)E * 2204;          // gW0 | gW3 | gb0 | gb1 | gb2 | gb3 per member
    const long long total = n_big + n_small + 4 + E;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        if (i < n_big) {
            const int layer = int(i / ((long long)E * kH * kH));
            const long long rest = i % ((long long)E * kH * kH);
            const int e = int(rest / (kH * kH));
            const int el = int(rest % (kH * kH));
            const float* src = g.partB + (((size_t)layer * E + e) * g.Q) * (size_t)(kH * kH) + el;
            float s = 0.f;
            for (int q = 0; q < g.Q; ++q) s += src[(size_t)q * kH * kH];
            (layer ? m.g_w2 : m.g_w1)[(size_t)e * kH * kH + el] = s;
        } else if (i < n_big + n_small) {
            const long long j = i - n_big;
            const int e = int(j / 2204), off = int(j % 2204);
            const float* src = g.partA + (size_t)e * g.P * kPartA + off;
            float s = 0.f;
            for (int p = 0; p < g.P; ++p) s += src[(size_t)p * kPartA];
            if (off < kOffW3) m.g_w0[(size_t)e * 800 + off] = s;
            else if (off < kOffB0) m.g_w3[(size_t)e * 800 + (off - kOffW3)] = s;
            else if (off < kOffB1) m.g_b0[(size_t)e * kH + (off - kOffB0)] = s;
            else if (off < kOffB2) m.g_b1[(size_t)e * kH + (off - kOffB1)] = s;
            else if (off < kOffB3) m.g_b2[(size_t)e * kH + (off - kOffB2)] = s;
            else m.g_b3[(size_t)e * kDout + (off - kOffB3)] = s;
        } else if (i < n_big + n_small + 4) {
            // g_max_logvar[k] = 0.01 + sum, g_min_logvar[k] = -0.01 + sum over all members and parts (MPC.py:271)
            const int k = int(i - n_big - n_small);
            float s = 0.f;
            for (int ep = 0; ep < E * g.P; ++ep) s += g.partA[(size_t)ep * kPartA + kOffLv + k];
            if (k < 2) m.g_max_logvar[k] = 0.01f + s;
            else m.g_min_logvar[k - 2] = -0.01f + s;
        } else if (loss_out) {
            const int e = int(i - n_big - n_small - 4);
            float s = 0.f;
            for (int p = 0; p < g.P; ++p) s += g.partA[((size_t)e * g.P + p) * kPartA + kOffLoss];
            loss_out[e] = s;
        }
    }
}

inline long long rows_padded(long long batch) { return (batch + kR - 1) / kR * kR; }
inline int parts_A(long long rows_pad, int E) {
    long long tiles = rows_pad / kR;
    long long want = 256 / (E > 0 ? E : 1);          // ~ one workgroup per CU over all members
    if (want < 1) want = 1;
    long long P = tiles < want ? tiles : want;
    return int(P > kMaxP ? kMaxP : P);
}
inline int parts_B(long long rows_pad, int E) {
    long long groups = rows_pad / kGB;
    long long want = 256 / (2 * (E > 0 ? E : 1));
    if (want < 1) want = 1;
    long long Q = groups < want ? groups : want;
    return int(Q > kMaxQ ? kMaxQ : Q);
}

}  // namespace

extern "C" {

int rrl_ens_train_big_supported(int d_in, int hidden, int d_out) {
    return d_in == kDin && hidden == kH && d_out == kDout;
}

long long rrl_ens_big_scratch_floats(int n_nets, long long batch) {
    if (n_nets <= 0 || batch <= 0) return 0;
    const long long rp = rows_padded(batch);
    return 6LL * n_nets * rp * kH + (long long)n_nets * parts_A(rp, n_nets) * kPartA +
           2LL * n_nets * parts_B(rp, n_nets) * kH * kH;
}

int rrl_ens_train_grad_big(const rrl_ens_t* m, long long batch, const float* train_in, const float* train_targ,
                           const int64_t* idx, long long idx_stride, float* scratch, float* loss_out, void* stream) {
    if (!m || !train_in || !train_targ || !idx || !scratch || m->n_nets <= 0 || m->n_nets > 60 || batch <= 0)
        return RRL_EINVAL;
    if (!rrl_ens_train_big_supported(m->d_in, m->hidden, m->d_out)) return RRL_ERANGE;
    if (!m->w0 || !m->b0 || !m->w1 || !m->b1 || !m->w2 || !m->b2 || !m->w3 || !m->b3 || !m->max_logvar ||
        !m->min_logvar || !m->mu || !m->sigma || !m->g_w0 || !m->g_b0 || !m->g_w1 || !m->g_b1 || !m->g_w2 ||
        !m->g_b2 || !m->g_w3 || !m->g_b3 || !m->g_max_logvar || !m->g_min_logvar)
        return RRL_EINVAL;
    static bool lds_set = false;
    if (!lds_set) {
        if (hipFuncSetAttribute((const void*)ens_big_fwd_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                kLdsBytesA) != hipSuccess ||
            hipFuncSetAttribute((const void*)ens_big_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                kLdsBytesB) != hipSuccess) {
            rrl_host::last_hip_error = int(hipGetLastError());
            return RRL_ELAUNCH;
        }
        lds_set = true;
    }
    BigArgs g;
    g.m = *m;
    g.train_in = train_in;
    g.train_targ = train_targ;
    g.idx = idx;
    g.idx_stride = idx_stride;
    g.nb = batch;
    g.rows_pad = rows_padded(batch);
    const int E = m->n_nets;
    g.P = parts_A(g.rows_pad, E);
    g.Q = parts_B(g.rows_pad, E);
    const size_t act = (size_t)E * g.rows_pad * kH;
    g.h0 = scratch;
    g.h1 = g.h0 + act;
    g.sp0 = g.h1 + act;
    g.sp1 = g.sp0 + act;
    g.d1 = g.sp1 + act;
    g.d2 = g.d1 + act;
    g.partA = g.d2 + act;
    g.partB = g.partA + (size_t)E * g.P * kPartA;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(ens_big_fwd_bwd_kernel, dim3(E * g.P), dim3(kThreadsA), kLdsBytesA, st, g);
    hipLaunchKernelGGL(ens_big_wgrad_kernel, dim3(2 * E * g.Q), dim3(kThreadsB), kLdsBytesB, st, g);
    hipLaunchKernelGGL(ens_big_reduce_kernel, dim3(512), dim3(256), 0, st, g, loss_out);
    return check_launch();
}

// One epoch of MPC.train's batch loop (MPC.py:266-292) at large batch, issued from C.
int rrl_ens_train_epoch_big(const rrl_ens_t* m, int n_seg, const rrl_adam_seg_t* segs, float lr, float beta1, float beta2,
                            float eps, const float* train_in, const float* train_targ, const int64_t* idx,
                            long long idx_stride, long long n_rows, long long batch, float* scratch, float* loss_out,
                            void* stream) {
    if (!idx || n_rows <= 0 || batch <= 0 || !segs) return RRL_EINVAL;
    for (long long lo = 0; lo < n_rows; lo += batch) {
        const long long nb = n_rows - lo < batch ? n_rows - lo : batch;
        int rc = rrl_ens_train_grad_big(m, nb, train_in, train_targ, idx + lo, idx_stride, scratch, loss_out, stream);
        if (rc != RRL_OK) return rc;
        rc = rrl_adam_step_multi(n_seg, segs, lr, beta1, beta2, eps, stream);
        if (rc != RRL_OK) return rc;
    }
    return RRL_OK;
}

}  // extern "C"
