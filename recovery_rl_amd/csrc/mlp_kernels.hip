// mlp_kernels.hip -- f32 MFMA building blocks for the SAC / Q_risk multilayer perceptrons on
// gfx950 (MI355X).
//
// The MLPs of the hot path are tiny (2 -> 256 -> 256 -> {1,2,4}; batch 256 for updates, 4096 rows
// for acting).  Vendor GEMMs pick 256x256 macro-tiles for them (one workgroup, 40-60 us per
// layer); here every 16x16 output tile is one wavefront running v_mfma_f32_16x16x4_f32 (exact
// f32, bitwise an fmaf chain; 32-cycle issue, two accumulators cover its 40-cycle latency), so a
// 256x256x256 layer is 256 independent waves (x heads), each 64 MFMAs deep, K staged through LDS
// in 128-wide panels with the next panel's global loads in flight under the MFMAs.  One kernel, three operand layouts, fused prologue/epilogue:
//
//   mode NT :  C[g] = A[g] . B[g]^T (+ bias[g]) (relu)          forward   Y = X W^T + b
//   mode NN :  C[g] = A[g] . B[g]   (* [S[g] > 0])              backward  dX = dY W   (relu mask of
//                                                                          the producing layer)
//   mode TN :  C[g] = A[g]^T . B[g] ; colsum[g] = sum_k A[g]    backward  dW = dY^T X ; db = sum dY
//
// All matrices row-major with explicit leading dimensions and per-head strides (g = blockIdx.z),
// arbitrary M, N, K (edges are zero-filled / bounds-checked).
#include <hip/hip_runtime.h>

#include <stdlib.h>

#include <algorithm>

#include "mlp_common.hpp"

namespace {

using rrl_host::check_launch;
using rrl::row16_sum;

constexpr int kTile = 16;    // output tile edge: one wavefront per 16x16 tile (v_mfma_f32_16x16x4_f32)
constexpr int kPanel = 128;  // K elements per panel
constexpr int kLd = 20;      // LDS tile[k][20]: 16 columns + 4 pad, rows 16-byte aligned
constexpr int kVec = 8;      // float4 per lane per operand per panel

struct GemmArgs {
    const float* A;
    const float* B;
    float* C;
    const float* bias;    // NT: [N] per head, nullable
    const float* mask;    // NN: saved activation, same shape as C, nullable
    float* colsum;        // TN: [M] per head, nullable (written by the blockIdx.x == 0 tiles)
    int M, N, K;
    int lda, ldb, ldc, ldmask;
    long long sA, sB, sC, sBias, sMask, sColsum;  // per-head strides (elements)
    int relu;
    int accumulate;       // C += result
    // NN tiles of a stack backward can finish the FIRST layer's backward as well (rrl_mlp_input_backward), from the
    // 16 x 16 tile of dh1 they hold, instead of a dependent launch that re-reads dh1:
    //   first_part[by][g*H*din + col*din + d] = sum over the tile's 16 rows of dh1[row][col] x[row][d]   (dW1 partial)
    //   first_part[by][G*H*din + g*H + col]   = sum over the tile's 16 rows of dh1[row][col]             (db1 partial)
    //   dx_part[bx][g][row][d]                = sum over the tile's 16 cols of dh1[row][col] W1[col][d]   (dx partial)
    // The consumers (Adam; the policy-head backward) add the 16 row-tile / column-tile partials in a fixed order.
    const float* x;       // [M, din] rows ldx apart, shared by the heads; null = no first-layer work
    const float* W1;      // [G, N, din]
    float* first_part;    // nullable
    float* dx_part;       // nullable
    long long first_stride;
    int ldx, din, G;
    int skip_c;           // do not write the C tile itself (dh1): nothing reads it once the first layer is done here
    int dx_fold;          // dx_part holds sums over groups of four consecutive column tiles, folded by the workgroup that holds them
};

// pointers of an argument block that was copied out of device memory (packed launches): see rrl_pack::to_global
__device__ __forceinline__ void globalize(GemmArgs& a) {
    rrl_pack::to_global_all(a.A, a.B, a.C, a.bias, a.mask, a.colsum, a.x, a.W1, a.first_part, a.dx_part);
}

template <int VEC>
struct FragT {
    float4 v[VEC];
};
using Frag = FragT<kVec>;

// Loads never branch: the FAST instantiation (every tile full, leading dimensions multiples of 4,
// 16-byte aligned bases) issues plain float4 loads; the generic one clamps indices into range and
// zeroes by predicate, so in both cases all loads of a panel are in flight together.
template <bool FAST>
__device__ __forceinline__ float4 load4(const float* __restrict__ base, long long row_off, int k, int K,
                                        bool row_ok) {
    if (FAST) return *reinterpret_cast<const float4*>(base + row_off + k);
    float v[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int kk = min(k + t, K - 1);
        const float x = base[row_off + (kk < 0 ? 0 : kk)];
        v[t] = (row_ok && k + t < K) ? x : 0.f;
    }
    return make_float4(v[0], v[1], v[2], v[3]);
}

// The K order inside a panel is permuted (the sum over k does not care): MFMA step s = 4 j + t of
// lane group q = lane >> 4 consumes k = 16 j + 4 q + t.  An operand whose k index is contiguous in
// memory (rows = M or N index) is then exactly element t of the lane's j-th float4 of its own row
// i = lane & 15 -- it feeds the MFMA straight from registers, no LDS.
template <bool FAST, int VEC>
__device__ __forceinline__ void load_direct(FragT<VEC>& f, const float* __restrict__ src, int ld, int row0,
                                            int rows, int k0, int K, int lane) {
    const int gr = row0 + (lane & 15);
    const bool ok = gr < rows;
    const long long off = (long long)(ok ? gr : rows - 1) * ld;
#pragma unroll
    for (int j = 0; j < VEC; ++j) f.v[j] = load4<FAST>(src, off, k0 + 16 * j + 4 * (lane >> 4), K, ok);
}

// An operand stored [k][col] (col contiguous) is staged through LDS: coalesced float4 loads along
// the columns, one ds_write_b128 each, read back as tile[k][i].
template <bool FAST, int VEC>
__device__ __forceinline__ void load_staged(FragT<VEC>& f, const float* __restrict__ src, int ld, int col0,
                                            int cols, int k0, int K, int lane) {
    const int c = col0 + (lane & 3) * 4;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const int k = k0 + (lane >> 2) + 16 * j;
        const bool ok = k < K;
        f.v[j] = load4<FAST>(src, (long long)(ok ? k : K - 1) * ld, c, cols, ok);
    }
}
template <int VEC>
__device__ __forceinline__ void store_staged(const FragT<VEC>& f, float* tile, int lane) {
#pragma unroll
    for (int j = 0; j < VEC; ++j)
        *reinterpret_cast<float4*>(tile + ((lane >> 2) + 16 * j) * kLd + (lane & 3) * 4) = f.v[j];
}

__device__ __forceinline__ float elem(const float4& q, int t) {
    return t == 0 ? q.x : (t == 1 ? q.y : (t == 2 ? q.z : q.w));
}

// this wave's LDS operations have completed (a wave that works on an LDS region of its own needs no workgroup barrier)
__device__ __forceinline__ void wave_lds_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// PANEL = K elements per panel (a multiple of 16).  The K chunks are consumed in the same ascending order with the same
// accumulator for every PANEL, so the result does not depend on it; a smaller panel = smaller LDS tiles = more tiles in
// flight per CU (the packed launches, which have more tiles than LDS for them) at the price of a wave-level sync per panel.
// DEEP (the solo launches: a handful of tiles per CU, every dependent round trip to memory is ~0.7 us of a ~7 us kernel): TWO
// panels in flight instead of one -- with K = 2 PANEL every operand of the tile is requested before the first wait -- and the
// epilogue's operands (bias, relu mask, the first layer's x and W1 rows) requested with them instead of behind the K loop.
// Same MFMA steps on the same operands in the same order: the same bits.
// GEN (the fused head + hidden backward, backward_pair_kernel): the left operand dh2 of both products is not read but derived
// where it is consumed -- a.A points at the saved activation h2 (same shape), and dh2[b][h] = h2[b][h] > 0 ? dOut[b] W3[h] : 0
// with dOut[b] in LDS (`dsh`, one output per row: the critic-loss kinds) and the head's W3 row `w3` -- the value
// head_bwd_loss_body would have stored, bit for bit (fmaf(go, w, 0)).  WSYNC: several tiles share a workgroup (one wave each,
// LDS regions of their own): wave-level syncs instead of workgroup barriers.
// `behind_requests()` runs once the tile's operands have been requested and before the first of them is looked at (GEN: the
// workgroup evaluates dOut there, its own round trip under the operands').
struct NothingBehindRequests {
    __device__ __forceinline__ void operator()() const {}
};
// DOUT (GEN): outputs of the head whose dh2 is derived -- dh2[b][h] = h2[b][h] > 0 ? sum_o dOut[b][o] W3[o][h] : 0 as the
// fmaf chain over o = 0 .. DOUT - 1 starting from 0 (head_bwd_loss_body's order); dsh is [B][DOUT], w3 [DOUT][H].  One output:
// the critic-loss kinds, W3 fragments from registers; 2 / 4 outputs (the stochastic / tanh-Gaussian policy heads): the k-
// contiguous W3 elements of the NN product come from an LDS copy `w3s` ([DOUT][H], staged by behind_requests()).
// SHARE (round 6; GEN, four tiles per 256-thread workgroup that consume the SAME left operand -- the four column tiles of one
// row tile in both planes of backward_pair_kernel): every wave requests and derives a QUARTER of each dh2 panel (16 of its
// 64 k) and hands it to the others through LDS (`Ash`: two buffers `ash_stride` floats apart, one workgroup barrier per
// panel) instead of all four requesting and deriving all of it.  These tiles are bound by the number of memory requests a
// CU can issue, not by a latency chain (profiles/round6_pair_allk_ab.txt): 8 -> 5 (TN) and 12 -> 5 (NN) operand requests
// per wave and panel.  Same values, same MFMA steps in the same order.
template <int MODE, bool FAST, int PANEL = kPanel, bool DEEP = false, bool GEN = false, bool WSYNC = false,
          class BEHIND = NothingBehindRequests, int DOUT = 1, bool SHARE = false>  // MODE: 0 NT, 1 NN, 2 TN
__device__ __forceinline__ void gemm16_tile(const GemmArgs& a, float* As, float* Bs, int bx, int by, int g,
                                            const float* dsh = nullptr, const float* w3 = nullptr,
                                            BEHIND behind_requests = BEHIND(), const float* w3s = nullptr,
                                            bool keep_sx = false, float* sx_out = nullptr, float* Ash = nullptr,
                                            int ash_stride = 0) {
    static_assert(!SHARE || (GEN && FAST && WSYNC && MODE != 0 && PANEL == 64), "SHARE: the paired backward's tiles");
    const int lane = threadIdx.x & 63;
    const int wave = SHARE ? int(threadIdx.x >> 6) : 0;
    const int m0 = by * kTile, n0 = bx * kTile;
    const float* A = a.A + g * a.sA;
    constexpr int VEC = PANEL / 16;
    using Frag = FragT<VEC>;
    const float* B = a.B + g * a.sB;
    float* C = a.C + g * a.sC;
    constexpr bool kStageA = MODE == 2, kStageB = MODE != 0;

    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    float asum = 0.f;  // TN: running sum of my A operands (for colsum)
    Frag fa, fb, na, nb;
    Frag fw, nw;          // GEN, NN, one output: the W3 elements that go with the k-contiguous dh2 fragments
    float4 w3c[DOUT];     // GEN, TN: W3 of the lane's four dh2 columns, per output
#pragma unroll
    for (int o = 0; o < DOUT; ++o) w3c[o] = make_float4(0.f, 0.f, 0.f, 0.f);
    const auto sync = [&]() {
        if constexpr (WSYNC) wave_lds_sync();
        else __syncthreads();
    };

    auto load_into = [&](Frag& ra, Frag& rb, Frag& rw, int k0) {
        if constexpr (SHARE) {       // this wave's quarter of the left operand: chunk j = wave of load_staged / load_direct
            if (kStageA) ra.v[0] = load4<FAST>(A, (long long)(k0 + (lane >> 2) + 16 * wave) * a.lda, m0 + (lane & 3) * 4, a.M, true);
            else ra.v[0] = load4<FAST>(A, (long long)(m0 + (lane & 15)) * a.lda, k0 + 16 * wave + 4 * (lane >> 4), a.K, true);
        } else {
            if (kStageA) load_staged<FAST>(ra, A, a.lda, m0, a.M, k0, a.K, lane);
            else load_direct<FAST>(ra, A, a.lda, m0, a.M, k0, a.K, lane);
        }
        if (kStageB) load_staged<FAST>(rb, B, a.ldb, n0, a.N, k0, a.K, lane);
        else load_direct<FAST>(rb, B, a.ldb, n0, a.N, k0, a.K, lane);
        if constexpr (GEN && MODE == 1 && DOUT == 1) {
            if constexpr (SHARE) {
                rw.v[0] = *reinterpret_cast<const float4*>(w3 + k0 + 16 * wave + 4 * (lane >> 4));
            } else {
#pragma unroll
                for (int j = 0; j < VEC; ++j) rw.v[j] = *reinterpret_cast<const float4*>(w3 + k0 + 16 * j + 4 * (lane >> 4));
            }
        }
    };
    auto load = [&](int k0) { load_into(fa, fb, fw, k0); };
    if constexpr (GEN && MODE == 2) {
#pragma unroll
        for (int o = 0; o < DOUT; ++o) w3c[o] = *reinterpret_cast<const float4*>(w3 + (long long)o * a.lda + m0 + (lane & 3) * 4);
    }
    // dh2 from (h2 fragment, dOut, W3): the formula of head_bwd_loss_body -- d = fmaf(go[o], w[o], d) over the outputs
    const auto gen4 = [](const float4& h, const float (&go)[DOUT], const float4 (&w)[DOUT]) {
        float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int o = 0; o < DOUT; ++o) {
            d.x = fmaf(go[o], w[o].x, d.x);
            d.y = fmaf(go[o], w[o].y, d.y);
            d.z = fmaf(go[o], w[o].z, d.z);
            d.w = fmaf(go[o], w[o].w, d.w);
        }
        return make_float4(h.x > 0.f ? d.x : 0.f, h.y > 0.f ? d.y : 0.f, h.z > 0.f ? d.z : 0.f, h.w > 0.f ? d.w : 0.f);
    };

    const int np = (a.K + PANEL - 1) / PANEL;
    const int i = lane & 15, q = lane >> 4;
    load(0);
    if (DEEP && np > 1) load_into(na, nb, nw, PANEL);
    // DEEP: the epilogue's operands, requested now
    const int ecol = n0 + (lane & 15);
    float pre_bias = 0.f, pre_mask[4] = {1.f, 1.f, 1.f, 1.f}, pre_x = 0.f, pre_w = 0.f;
    if constexpr (DEEP) {
        if (MODE == 0 && a.bias && ecol < a.N) pre_bias = a.bias[g * a.sBias + ecol];
        if (MODE == 1 && a.mask && ecol < a.N) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + 4 * (lane >> 4) + r;
                if (row < a.M) pre_mask[r] = a.mask[g * a.sMask + (long long)row * a.ldmask + ecol];
            }
        }
        if (MODE == 1 && a.x) {
            const int rr = lane & 15, dd = lane >> 4;
            pre_x = dd < a.din ? a.x[(long long)(m0 + rr) * a.ldx + dd] : 0.f;
            pre_w = dd < a.din ? a.W1[((long long)g * a.N + n0 + rr) * a.din + dd] : 0.f;
        }
    }
    behind_requests();
    for (int p = 0; p < np; ++p) {
        Frag ca = fa, cb = fb;           // operands of this panel (registers)
        float* Ab = SHARE ? Ash + (p & 1) * ash_stride : nullptr;          // SHARE: this panel's dh2, all four quarters
        if constexpr (SHARE) {
            float go[DOUT];
            float4 w[DOUT];
            if constexpr (MODE == 2) {
#pragma unroll
                for (int o = 0; o < DOUT; ++o) go[o] = dsh[(p * PANEL + (lane >> 2) + 16 * wave) * DOUT + o];
                *reinterpret_cast<float4*>(Ab + ((lane >> 2) + 16 * wave) * kLd + (lane & 3) * 4) = gen4(fa.v[0], go, w3c);
            } else {
#pragma unroll
                for (int o = 0; o < DOUT; ++o) {
                    go[o] = dsh[(m0 + (lane & 15)) * DOUT + o];
                    if constexpr (DOUT == 1) w[o] = fw.v[0];
                    else w[o] = *reinterpret_cast<const float4*>(w3s + o * a.K + p * PANEL + 16 * wave + 4 * (lane >> 4));
                }
                *reinterpret_cast<float4*>(Ab + (wave * 64 + lane) * 4) = gen4(fa.v[0], go, w);
            }
            if (p) sync();               // (wave level) my previous panel's reads of Bs are done
            store_staged(cb, Bs, lane);
            __syncthreads();             // every quarter of this panel's dh2 is in Ab (its last readers were two panels ago)
        }
        if constexpr (GEN && !SHARE) {
            if constexpr (MODE == 2) {           // staged [k = batch row][4 dh2 columns]
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    float go[DOUT];
#pragma unroll
                    for (int o = 0; o < DOUT; ++o) go[o] = dsh[(p * PANEL + (lane >> 2) + 16 * j) * DOUT + o];
                    ca.v[j] = gen4(fa.v[j], go, w3c);
                }
            } else {                             // direct: row m0 + i, four consecutive hidden columns per fragment
                float go[DOUT];
#pragma unroll
                for (int o = 0; o < DOUT; ++o) go[o] = dsh[(m0 + (lane & 15)) * DOUT + o];
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    float4 w[DOUT];
                    if constexpr (DOUT == 1) {
                        w[0] = fw.v[j];
                    } else {
#pragma unroll
                        for (int o = 0; o < DOUT; ++o)
                            w[o] = *reinterpret_cast<const float4*>(w3s + o * a.K + p * PANEL + 16 * j + 4 * (lane >> 4));
                    }
                    ca.v[j] = gen4(fa.v[j], go, w);
                }
            }
        }
        if (!SHARE && (kStageA || kStageB)) {
            if (p) sync();               // the previous panel's LDS reads are done
            if (kStageA) store_staged(ca, As, lane);
            if (kStageB) store_staged(cb, Bs, lane);
            sync();
        }
        float4 a4[VEC];                  // SHARE, NN: the lane's k-contiguous fragments of dh2, read back from Ab
        if constexpr (SHARE && MODE == 1) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) a4[j] = *reinterpret_cast<const float4*>(Ab + (j * 64 + lane) * 4);
        }
        if constexpr (DEEP) {
            if (p + 1 < np) { fa = na; fb = nb; fw = nw; }              // the panel behind this one is on its way already
            if (p + 2 < np) load_into(na, nb, nw, (p + 2) * PANEL);
        } else {
            if (p + 1 < np) load((p + 1) * PANEL);   // next panel's global loads fly under the MFMAs
        }
        const int klen = min(PANEL, a.K - p * PANEL);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            if (16 * j < klen) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int k = 16 * j + 4 * q + t;
                    const float av = SHARE ? (MODE == 2 ? Ab[k * kLd + i] : elem(a4[j], t))
                                           : (kStageA ? As[k * kLd + i] : elem(ca.v[j], t));
                    const float bv = kStageB ? Bs[k * kLd + i] : elem(cb.v[j], t);
                    if (MODE == 2) asum += av;
                    if (t & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc1, 0, 0, 0);
                    else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc0, 0, 0, 0);
                }
            }
        }
    }
    const f32x4 acc = acc0 + acc1;

    // epilogue: lane holds C[row][col], col = lane & 15, row = 4 (lane >> 4) + r
    const int col = n0 + (lane & 15);
    const float bias = DEEP ? pre_bias : ((MODE == 0 && a.bias && col < a.N) ? a.bias[g * a.sBias + col] : 0.f);
    float vout[4] = {0.f, 0.f, 0.f, 0.f};
    const bool store_c = MODE != 1 || !a.skip_c;
    if (col < a.N) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + 4 * (lane >> 4) + r;
            if (row < a.M) {
                float v = acc[r] + bias;
                if (a.relu) v = v > 0.f ? v : 0.f;
                if (MODE == 1 && a.mask) {
                    const float s = DEEP ? pre_mask[r] : a.mask[g * a.sMask + (long long)row * a.ldmask + col];
                    v = s > 0.f ? v : 0.f;
                }
                vout[r] = v;
                if (store_c) {
                    float* dst = C + (long long)row * a.ldc + col;
                    *dst = a.accumulate ? (*dst + v) : v;
                }
            }
        }
    }
    if constexpr (MODE == 1) {
        if (a.x) {      // first-layer backward from this tile (FAST geometry: full tiles); a WG is one wavefront
            // the tile and the 16 rows of x / W1 it meets, through LDS (the K loop is done with Bs)
            float* T = Bs;                    // [16][17] tile, then xs [16][4] at 272, ws [16][4] at 336
            const int rr = lane & 15, dd = lane >> 4;
            const float xv = DEEP ? pre_x : (dd < a.din ? a.x[(long long)(m0 + rr) * a.ldx + dd] : 0.f);
            const float wv = DEEP ? pre_w : (dd < a.din ? a.W1[((long long)g * a.N + n0 + rr) * a.din + dd] : 0.f);
            sync();
#pragma unroll
            for (int r = 0; r < 4; ++r) T[(4 * (lane >> 4) + r) * 17 + (lane & 15)] = vout[r];
            T[272 + rr * 4 + dd] = xv;
            T[336 + rr * 4 + dd] = wv;
            sync();
            float sw = 0.f, sb = 0.f, sx = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const float tc = T[k * 17 + rr];                     // dh1[row k][col rr]
                sw = fmaf(tc, T[272 + k * 4 + dd], sw);              // x[row k][d]
                sb += tc;
                sx = fmaf(T[rr * 17 + k], T[336 + k * 4 + dd], sx);  // dh1[row rr][col k] W1[col k][d]
            }
            if (a.first_part && dd < a.din)
                a.first_part[by * a.first_stride + ((long long)g * a.N + n0 + rr) * a.din + dd] = sw;
            if (a.first_part && dd == 0)
                a.first_part[by * a.first_stride + (long long)a.G * a.N * a.din + (long long)g * a.N + n0 + rr] = sb;
            if (keep_sx) *sx_out = sx;         // the caller folds the dx partials of its four tiles (dx_fold)
            else if (a.dx_part && dd < a.din)
                a.dx_part[(((long long)bx * a.G + g) * a.M + m0 + rr) * a.din + dd] = sx;
        }
    }
    if (MODE == 2 && a.colsum && bx == 0) {
        float tot = asum + __shfl_xor(asum, 16);
        tot += __shfl_xor(tot, 32);
        const int row = m0 + (lane & 15);
        if (lane < 16 && row < a.M) a.colsum[g * a.sColsum + row] = tot;
    }
}

template <int MODE, bool FAST>
__global__ __launch_bounds__(64) void gemm16_kernel(GemmArgs a) {
    __shared__ __attribute__((aligned(16))) float As[MODE == 2 ? kPanel * kLd : 4];
    __shared__ __attribute__((aligned(16))) float Bs[MODE != 0 ? kPanel * kLd : 4];
    gemm16_tile<MODE, FAST>(a, As, Bs, blockIdx.x, blockIdx.y, blockIdx.z);
}

// The two H x H products of a stack backward share their left operand dh2 and do not depend on each other:
//   dW2[g] = dh2[g]^T h1[g] (+ column sums = db2)   (TN)        dh1[g] = (dh2[g] W2[g]) * [h1[g] > 0]   (NN)
// One launch: the first `tn_tiles` workgroups (per head) take the TN tiles, the rest the NN tiles.
template <bool FAST>
__global__ __launch_bounds__(64) void gemm16_pair_kernel(GemmArgs tn, GemmArgs nn, int tn_tiles_x, int tn_tiles,
                                                         int nn_tiles_x) {
    __shared__ __attribute__((aligned(16))) float As[kPanel * kLd];
    __shared__ __attribute__((aligned(16))) float Bs[kPanel * kLd];
    const int b = blockIdx.x, g = blockIdx.y;
    if (b < tn_tiles) {
        gemm16_tile<2, FAST>(tn, As, Bs, b % tn_tiles_x, b / tn_tiles_x, g);
    } else {
        const int c = b - tn_tiles;
        gemm16_tile<1, FAST>(nn, As, Bs, c % nn_tiles_x, c / nn_tiles_x, g);
    }
}

// Several independent stack backwards (e.g. the critic's backward for its own loss and its backward for the policy
// loss, sac.py:233-239) share one launch: a flat grid over (problem, head, tile).  Problems with dW2 == null
// contribute only their NN tiles (input gradient).
struct HiddenGroup {
    GemmArgs tn[kMaxGroup], nn[kMaxGroup];
    int tn_tiles_x[kMaxGroup], tn_tiles[kMaxGroup], nn_tiles_x[kMaxGroup], per_head[kMaxGroup], fast[kMaxGroup];
    int first[kMaxGroup + 1];
    int n;
};

// Solo launch: grid (tiles per product x heads, members, 2) -- blockIdx.y IS the member and blockIdx.z the product (0: input
// gradient NN, 1: weight gradient TN), so the ONE argument block the workgroup needs sits at a kernel-argument address known
// at wave start (mlp_common.hpp).  (The flat grid needs first[] to find the member, the member's tile counts to find the
// product, the product's argument block, and only then the operands: four dependent round trips before the first MFMA.)
// Members without a weight gradient leave their z = 1 workgroups empty: they are dispatched last and exit after that batch.
struct HiddenJob {
    GemmArgs ga;
    int tiles, tiles_x, fast, G;      // tiles per head (0: nothing to do), tiles per tile row, FAST geometry, heads
};
struct HiddenJobs {
    HiddenJob job[kMaxGroup][2];      // [member][0: NN, 1: TN]
};
// the member's job, wherever the jobs live (kernel arguments of the solo launch, the plan's device copy of a packed one)
template <int PANEL, bool DEEP>
__device__ __forceinline__ void hidden_jobs_body(const HiddenJobs& hj, int x, float* As, float* Bs) {
    const bool is_tn = blockIdx.z != 0;
    HiddenJob j = hj.job[blockIdx.y][blockIdx.z];
    GemmArgs& ga = j.ga;
    globalize(ga);
    arrive_together(ga.M, ga.N, ga.K, ga.lda, ga.ldb, ga.ldc, ga.ldmask, ga.sA, ga.sB, ga.sC, ga.sBias, ga.sMask, ga.sColsum,
                    ga.relu, ga.accumulate, ga.first_stride, ga.ldx, ga.din, ga.G, ga.skip_c, j.tiles, j.tiles_x, j.fast, j.G);
    if (x >= j.tiles * j.G) return;
    const int g = x / j.tiles, b = x - g * j.tiles;
    if (is_tn) {
        if (j.fast) gemm16_tile<2, true, PANEL, DEEP>(ga, As, Bs, b % j.tiles_x, b / j.tiles_x, g);
        else gemm16_tile<2, false, PANEL, DEEP>(ga, As, Bs, b % j.tiles_x, b / j.tiles_x, g);
    } else {
        if (j.fast) gemm16_tile<1, true, PANEL, DEEP>(ga, As, Bs, b % j.tiles_x, b / j.tiles_x, g);
        else gemm16_tile<1, false, PANEL, DEEP>(ga, As, Bs, b % j.tiles_x, b / j.tiles_x, g);
    }
}
__global__ __launch_bounds__(64) void gemm16_group_kernel(HiddenJobs hj) {
    __shared__ __attribute__((aligned(16))) float As[kPanel * kLd];
    __shared__ __attribute__((aligned(16))) float Bs[kPanel * kLd];
    hidden_jobs_body<kPanel, true>(hj, blockIdx.x, As, Bs);
}

// Packed launch (pack.hpp): grid (tiles of the seeds' largest jobs under the XCD-aware placement, members, {NN, TN}) -- member
// and product out of the grid as in the solo launch, the seed from blockIdx.x by arithmetic: the job's block arrives in one
// batch of scalar loads from the plan.  PANEL = 128: 20 KB of LDS per single-wave workgroup, i.e. 8 tiles in flight per CU
// -- enough for one seed (1 000-1 500 tiles per launch); PANEL = 64 (every member FAST, i.e. K a multiple of 128): 16 tiles
// per CU.  Two K panels in flight (DEEP) as in the solo launch: one wave per workgroup has the registers for it.
template <int PANEL>
__global__ __launch_bounds__(64) void gemm16_pack_kernel(const HiddenJobs* __restrict__ groups, rrl_pack::Idx ix) {
    __shared__ __attribute__((aligned(16))) float As[PANEL * kLd];
    __shared__ __attribute__((aligned(16))) float Bs[PANEL * kLd];
    RRL_PACK_LOCATE(ix, groups, s, local);
    hidden_jobs_body<PANEL, true>(groups[s], local, As, Bs);
}

// ---- block form of the hidden-layer backward for the packed launches -------------------------------------------------
// With many seeds in one launch the 16 x 16 tiles above are bound by operand traffic, not by latency: every tile pulls its
// own 16 x K slices of both operands through its CU's vector memory path, in half-used 128-byte lines.  Here a
// four-wave workgroup owns a (32 WM) x (32 WN) block of the output, stages each K panel of the [k][col] operands ONCE for
// all its waves (whole 128-byte lines, 1 / (2 WM) resp. 1 / (2 WN) of the tile form's loads per output element) in
// double-buffered LDS, and every wave keeps WM x WN tiles in registers.  Per output element nothing changes: the same
// v_mfma_f32_16x16x4_f32 steps on the same operands in the same order (k = 16 j + 4 q + t ascending in j, t; even t into one
// accumulator, odd t into the other), the same epilogue -- a packed seed still equals its solo run bit for bit.
// FAST geometry only (full blocks, K a multiple of the panel, aligned bases); anything else keeps the tile kernel.
constexpr int kBlkPanel = 32;    // K elements per staged panel

template <int W>   // W = operand width in columns (32 or 64): LDS rows of W + 4 floats (conflict-free fragment reads)
__device__ __forceinline__ void blk_load(FragT<kBlkPanel * W / 1024>& f, const float* __restrict__ src, int ld, int col0,
                                         int k0, int tid) {
    constexpr int LPR = W / 4, RPP = 256 / LPR;
#pragma unroll
    for (int jj = 0; jj < kBlkPanel / RPP; ++jj)
    {   // native vector load / store: a float4 struct copy between address spaces stays a memcpy through scratch
        const f32x4 t = *reinterpret_cast<const f32x4*>(src + (long long)(k0 + tid / LPR + RPP * jj) * ld + col0 + 4 * (tid % LPR));
        f.v[jj] = make_float4(t[0], t[1], t[2], t[3]);
    }
}
template <int W>
__device__ __forceinline__ void blk_store(const FragT<kBlkPanel * W / 1024>& f, float* buf, int tid) {
    constexpr int LPR = W / 4, RPP = 256 / LPR;
#pragma unroll
    for (int jj = 0; jj < kBlkPanel / RPP; ++jj)
        *reinterpret_cast<f32x4*>(buf + (tid / LPR + RPP * jj) * (W + 4) + 4 * (tid % LPR)) =
            f32x4{f.v[jj].x, f.v[jj].y, f.v[jj].z, f.v[jj].w};
}

template <int WM, int WN>
struct BlkLds {
    static constexpr int BM = 32 * WM, BN = 32 * WN;
    static constexpr int kA = kBlkPanel * (BM + 4), kB = kBlkPanel * (BN + 4);
    static constexpr int kFloats = 2 * kA + 2 * kB > 4 * 400 ? 2 * kA + 2 * kB : 4 * 400;
};

// DOUT > 0 (the paired launch of the packed backward, backward_pair_block_pack_kernel): the left operand dh2 of both products
// is derived where it goes to LDS / to the MFMA, exactly as in gemm16_tile<GEN> -- a.A points at the saved activation h2,
// dsh [B][DOUT] is the head's dOut and w3s [DOUT][H] its W3, both in LDS (filled by behind_requests(), which runs once the
// first panel's operands are requested).  Same values as the head backward stores, same MFMA steps: the same bits.
template <int MODE, int WM, int WN, int DOUT = 0, class BEHIND = NothingBehindRequests>   // MODE: 1 NN (A direct, B staged), 2 TN (both staged)
__device__ __forceinline__ void gemm_block(const GemmArgs& a, float* lds, int bx, int by, int g, const float* dsh = nullptr,
                                           const float* w3s = nullptr, BEHIND behind_requests = BEHIND()) {
    using L = BlkLds<WM, WN>;
    constexpr int D = DOUT > 0 ? DOUT : 1;
    constexpr int BM = L::BM, BN = L::BN, LDA = BM + 4, LDB = BN + 4, VEC = kBlkPanel / 16;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = w >> 1, wn = w & 1;
    const int i = lane & 15, q = lane >> 4;
    const int m0 = by * BM, n0 = bx * BN;             // block origin
    const int tm0 = wm * WM, tn0 = wn * WN;           // my first row / column tile inside the block
    const float* A = a.A + g * a.sA;
    const float* B = a.B + g * a.sB;
    float* C = a.C + g * a.sC;
    float* As = lds;                                  // [2][kBlkPanel][LDA]   (TN)
    float* Bs = lds + 2 * L::kA;                      // [2][kBlkPanel][LDB]

    f32x4 acc0[WM][WN], acc1[WM][WN];
#pragma unroll
    for (int x = 0; x < WM; ++x)
#pragma unroll
        for (int y = 0; y < WN; ++y) acc0[x][y] = acc1[x][y] = f32x4{0.f, 0.f, 0.f, 0.f};
    float asum[WM];
#pragma unroll
    for (int x = 0; x < WM; ++x) asum[x] = 0.f;
    const bool want_sum = MODE == 2 && a.colsum && bx == 0 && wn == 0;   // the tiles of output column 0 own the column sums

    FragT<kBlkPanel * BM / 1024> ra;                                    // staged operands on their way to LDS
    FragT<kBlkPanel * BN / 1024> rb;
    FragT<VEC> fa[WM];                                                  // NN: my rows of A (next panel)
    auto load = [&](int k0) __attribute__((always_inline)) {
        if (MODE == 2) blk_load<BM>(ra, A, a.lda, m0, k0, tid);
        else {
#pragma unroll
            for (int x = 0; x < WM; ++x) load_direct<true>(fa[x], A, a.lda, m0 + (tm0 + x) * kTile, a.M, k0, a.K, lane);
        }
        blk_load<BN>(rb, B, a.ldb, n0, k0, tid);
    };
    // dh2 from (h2 fragment, dOut, W3): d = fmaf(go[o], w[o], d) over the outputs (head_bwd_loss_body's order)
    const auto gen4 = [](const float4& h, const float (&go)[D], const float4 (&w)[D]) {
        float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int o = 0; o < D; ++o) {
            d.x = fmaf(go[o], w[o].x, d.x);
            d.y = fmaf(go[o], w[o].y, d.y);
            d.z = fmaf(go[o], w[o].z, d.z);
            d.w = fmaf(go[o], w[o].w, d.w);
        }
        return make_float4(h.x > 0.f ? d.x : 0.f, h.y > 0.f ? d.y : 0.f, h.z > 0.f ? d.z : 0.f, h.w > 0.f ? d.w : 0.f);
    };
    auto stage = [&](int buf, int k0) __attribute__((always_inline)) {
        if (MODE == 2) {
            if constexpr (DOUT > 0) {         // staged [k = batch row][4 dh2 columns per thread]
                constexpr int LPR = BM / 4, RPP = 256 / LPR;
                float4 w[D];
#pragma unroll
                for (int o = 0; o < D; ++o) w[o] = *reinterpret_cast<const float4*>(w3s + o * a.lda + m0 + 4 * (tid % LPR));
#pragma unroll
                for (int jj = 0; jj < kBlkPanel / RPP; ++jj) {
                    float go[D];
#pragma unroll
                    for (int o = 0; o < D; ++o) go[o] = dsh[(k0 + tid / LPR + RPP * jj) * D + o];
                    ra.v[jj] = gen4(ra.v[jj], go, w);
                }
            }
            blk_store<BM>(ra, As + buf * L::kA, tid);
        }
        blk_store<BN>(rb, Bs + buf * L::kB, tid);
    };

    // panel p: registers -> LDS buffer p & 1 (its last readers, panel p - 2, are behind the barrier of panel p - 1), panel
    // p + 1's global loads issued, ONE barrier, MFMAs -- the loads fly under them
    const int np = a.K / kBlkPanel;
    load(0);
    behind_requests();
    for (int p = 0; p < np; ++p) {
        FragT<VEC> ca[WM];                    // NN: this panel's rows of A
#pragma unroll
        for (int x = 0; x < WM; ++x) ca[x] = fa[x];
        if constexpr (DOUT > 0 && MODE == 1) {       // direct: row (tm0 + x) * 16 + i, four consecutive hidden columns per fragment
#pragma unroll
            for (int x = 0; x < WM; ++x) {
                float go[D];
#pragma unroll
                for (int o = 0; o < D; ++o) go[o] = dsh[(m0 + (tm0 + x) * kTile + i) * D + o];
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    float4 w[D];
#pragma unroll
                    for (int o = 0; o < D; ++o)
                        w[o] = *reinterpret_cast<const float4*>(w3s + o * a.K + p * kBlkPanel + 16 * j + 4 * q);
                    ca[x].v[j] = gen4(ca[x].v[j], go, w);
                }
            }
        }
        stage(p & 1, p * kBlkPanel);
        if (p + 1 < np) load((p + 1) * kBlkPanel);
        __syncthreads();
        const float* Ap = As + (p & 1) * L::kA;
        const float* Bp = Bs + (p & 1) * L::kB;
        // all fragments of the panel first (ds_reads in flight together), then its MFMAs back to back
        float av[VEC][4][WM], bv[VEC][4][WN];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int k = 16 * j + 4 * q + t;
#pragma unroll
                for (int x = 0; x < WM; ++x) av[j][t][x] = MODE == 2 ? Ap[k * LDA + (tm0 + x) * kTile + i] : elem(ca[x].v[j], t);
#pragma unroll
                for (int y = 0; y < WN; ++y) bv[j][t][y] = Bp[k * LDB + (tn0 + y) * kTile + i];
            }
        }
        __builtin_amdgcn_sched_barrier(0);     // (the scheduler would sink every read to just before its use again)
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (MODE == 2 && want_sum) {
#pragma unroll
                    for (int x = 0; x < WM; ++x) asum[x] += av[j][t][x];
                }
#pragma unroll
                for (int x = 0; x < WM; ++x)
#pragma unroll
                    for (int y = 0; y < WN; ++y) {
                        if (t & 1) acc1[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][t][x], bv[j][t][y], acc1[x][y], 0, 0, 0);
                        else acc0[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][t][x], bv[j][t][y], acc0[x][y], 0, 0, 0);
                    }
            }
        }
    }
    __syncthreads();                       // every wave is done with the staged panels: the epilogue re-uses the LDS

    // epilogue, tile by tile, exactly gemm16_tile's: lane holds C[row][col], col = lane & 15, row = 4 (lane >> 4) + r
    float* T = lds + w * 400;             // this wave's scratch for the first-layer work (the staged panels are done with)
#pragma unroll
    for (int x = 0; x < WM; ++x) {
        float sxv[WN];                    // dx partials of my column tiles (dx_fold)
#pragma unroll
        for (int y = 0; y < WN; ++y) {
            sxv[y] = 0.f;
            const f32x4 acc = acc0[x][y] + acc1[x][y];
            const int tm = m0 + (tm0 + x) * kTile, tn = n0 + (tn0 + y) * kTile;
            const int col = tn + i;
            const float bias = 0.f;
            float vout[4];
            const bool store_c = MODE != 1 || !a.skip_c;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = tm + 4 * q + r;
                float v = acc[r] + bias;
                if (a.relu) v = v > 0.f ? v : 0.f;
                if (MODE == 1 && a.mask) {
                    const float s = a.mask[g * a.sMask + (long long)row * a.ldmask + col];
                    v = s > 0.f ? v : 0.f;
                }
                vout[r] = v;
                if (store_c) {
                    float* dst = C + (long long)row * a.ldc + col;
                    *dst = a.accumulate ? (*dst + v) : v;
                }
            }
            if constexpr (MODE == 1) {
                if (a.x) {
                    const int rr = lane & 15, dd = lane >> 4;
                    const float xv = dd < a.din ? a.x[(long long)(tm + rr) * a.ldx + dd] : 0.f;
                    const float wv = dd < a.din ? a.W1[((long long)g * a.N + tn + rr) * a.din + dd] : 0.f;
                    wave_lds_sync();
#pragma unroll
                    for (int r = 0; r < 4; ++r) T[(4 * q + r) * 17 + i] = vout[r];
                    T[272 + rr * 4 + dd] = xv;
                    T[336 + rr * 4 + dd] = wv;
                    wave_lds_sync();
                    float sw = 0.f, sb = 0.f, sx = 0.f;
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const float tc = T[k * 17 + rr];
                        sw = fmaf(tc, T[272 + k * 4 + dd], sw);
                        sb += tc;
                        sx = fmaf(T[rr * 17 + k], T[336 + k * 4 + dd], sx);
                    }
                    const int tby = tm / kTile, tbx = tn / kTile;
                    if (a.first_part && dd < a.din)
                        a.first_part[tby * a.first_stride + ((long long)g * a.N + tn + rr) * a.din + dd] = sw;
                    if (a.first_part && dd == 0)
                        a.first_part[tby * a.first_stride + (long long)a.G * a.N * a.din + (long long)g * a.N + tn + rr] = sb;
                    if (a.dx_fold) sxv[y] = sx;
                    else if (a.dx_part && dd < a.din)
                        a.dx_part[(((long long)tbx * a.G + g) * a.M + tm + rr) * a.din + dd] = sx;
                }
            }
        }
        if constexpr (MODE == 1 && WN == 2) {
            if (a.x && a.dx_part && a.dx_fold) {          // workgroup-uniform
                // the block's 64 columns are one group of four column tiles: ((p0 + p1) + p2) + p3 -- the left wave of the
                // row hands p0 + p1 over through LDS, the right wave adds its two and stores (gemm16's paired launch: the same)
                float* F = lds + 4 * 400 + (wm * WM + x) * 64;
                if (wn == 0) F[lane] = sxv[0] + sxv[1];
                __syncthreads();
                if (wn == 1) {
                    float gs = F[lane];
                    gs += sxv[0];
                    gs += sxv[1];
                    const int rr = lane & 15, dd = lane >> 4;
                    const int tm = m0 + (tm0 + x) * kTile;
                    if (dd < a.din) a.dx_part[(((long long)bx * a.G + g) * a.M + tm + rr) * a.din + dd] = gs;
                }
            }
        }
        if (MODE == 2 && want_sum) {
            float tot = asum[x] + __shfl_xor(asum[x], 16);
            tot += __shfl_xor(tot, 32);
            if (lane < 16) a.colsum[g * a.sColsum + m0 + (tm0 + x) * kTile + lane] = tot;
        }
    }
}

// HiddenJobs with their tile counts in BLOCK units (hidden_blocks): grid (blocks, members, {NN, TN}) as above
template <int WM, int WN>
__global__ __launch_bounds__(256) void gemm_block_pack_kernel(const HiddenJobs* __restrict__ groups, rrl_pack::Idx ix) {
    __shared__ __attribute__((aligned(16))) float lds[BlkLds<WM, WN>::kFloats];
    RRL_PACK_LOCATE(ix, groups, s, local);
    const bool is_tn = blockIdx.z != 0;
    HiddenJob j = groups[s].job[blockIdx.y][blockIdx.z];
    GemmArgs& ga = j.ga;
    globalize(ga);
    arrive_together(ga.M, ga.N, ga.K, ga.lda, ga.ldb, ga.ldc, ga.ldmask, ga.sA, ga.sB, ga.sC, ga.sBias, ga.sMask, ga.sColsum,
                    ga.relu, ga.accumulate, ga.first_stride, ga.ldx, ga.din, ga.G, ga.skip_c, j.tiles, j.tiles_x, j.G);
    if (local >= j.tiles * j.G) return;
    const int g = local / j.tiles, b = local - g * j.tiles;
    if (is_tn) gemm_block<2, WM, WN>(ga, lds, b % j.tiles_x, b / j.tiles_x, g);
    else gemm_block<1, WM, WN>(ga, lds, b % j.tiles_x, b / j.tiles_x, g);
}

// ---- thin-dimension pieces of the stack backward (dout <= 4, din <= 4) ------------------------------
// They are far from GEMM-shaped (one side is 1..4 wide), so each gets a dedicated streaming kernel
// instead of a padded MFMA tile.

// head backward: given dOut [G,B,dout] (gradient w.r.t. the last linear layer's output),
//   dW3[g][o][h] = sum_b dOut[g][b][o] h2[g][b][h]      db3[g][o] = sum_b dOut[g][b][o]
//   dh2[g][b][h] = [h2 > 0] sum_o dOut[g][b][o] W3[g][o][h]
// grid (H / 16, G); 256 threads = 16 hidden columns x 16 batch slices.  The batch loop has a fixed,
// fully unrolled trip count (predicated), so all of a thread's loads are in flight together; the
// slices are summed in a fixed order (deterministic).
constexpr int kCols = 16, kSlices = 16, kUnroll = 16;

// dOut is either read from memory (KIND = kPlainDOut: la.out = dOut [G,B,dout]) or computed in place from a loss
// description (rrl_loss_t): the formulas of update_kernels.hip (sac/qrisk *_grad, gauss/stoch_head_bwd),
// evaluated per (g, b, o).
constexpr int kPlainDOut = -1;
namespace loss {


__device__ __forceinline__ float sigm(float z) { return 1.f / (1.f + expf(-z)); }


// The policy-head kinds, split into "request everything" and "evaluate": a thread requests the operands of ALL its elements
// (two critic heads x 16 column-tile partials of dL/d action, the head's partial sums, noise, scale) before it adds anything.
// (d_action_sum above adds head 0's partials before it asks for head 1's, and the element loop asked for element 2's operands
// after element 1's tanh / exp: four to five dependent round trips in a kernel whose critic-loss twin has one -- 7.7 / 9.1 us
// against 4.6.)  Same additions in the same order: the same bits.
// NP = partials of a critic head held per element: 4 (the folded layout of H <= 256, or a plain tensor) or 16 (tile partials)
template <int NP>
struct HeadIn {
    float da[2][NP];      // dL/d action partials of critic heads 0 and 1 (heads beyond two: added by d_action_fold)
    float m[4], r[4];     // partial sums of the head's outputs: (mean | raw log-std) or (raw mean | unused)
    float e, sc, x2;      // noise, scale, (stochastic head) v2[j]
};
// dL/d action[b][j] = sum over the critic heads that consumed the action and, when the critic's first-layer backward came
// out of the hidden-layer tiles (rrl_first_layer_t), over their partials.  Order (every consumer, every producer layout):
// head by head; inside a head the partials one after the other (da_group <= 1: they are final -- a plain tensor, or the
// sums a folding producer stored), or (da_group = 4: column-TILE partials) every four consecutive ones first as
// ((p0 + p1) + p2) + p3 and the group sums one after the other -- which is what a folding producer stores, so 16 tile
// partials and 4 folded ones give the same bits.  Up to four final partials per head are requested as 8 loads (NP = 4),
// anything else as 32 (NP = 16): the caller picks the instantiation once per kernel (needs_np16).
__device__ __forceinline__ bool needs_np16(const rrl_loss_t& a) { return a.da_parts > 4 || a.da_group == 4; }
template <int NP>
__device__ __forceinline__ void d_action_load(const rrl_loss_t& a, int b, int j, float (&da)[2][NP]) {
    const float* p = a.d_action + (long long)b * a.ld + j;
    const int parts = a.da_parts > 1 ? a.da_parts : 1;
    const long long h1 = (1 < a.n_heads ? 1 : 0) * a.head_stride;
#pragma unroll
    for (int t = 0; t < NP; ++t) {
        const long long off = (t < parts ? t : 0) * a.da_part_stride;
        da[0][t] = p[off];
        da[1][t] = p[h1 + off];
    }
}
// the sum of one head's partials v[0 .. parts) on top of `da`, in the order above
template <int NP>
__device__ __forceinline__ float d_action_head(float da, const float (&v)[NP], int parts, int group) {
    if constexpr (NP == 16) {
        if (group == 4) {
#pragma unroll
            for (int t0 = 0; t0 < 16; t0 += 4) {
                float gs = v[t0];
#pragma unroll
                for (int t = 1; t < 4; ++t) gs = t0 + t < parts ? gs + v[t0 + t] : gs;
                da = t0 < parts ? da + gs : da;
            }
            return da;
        }
    }
#pragma unroll
    for (int t = 0; t < NP; ++t) da = t < parts ? da + v[t] : da;
    return da;
}
template <int NP>
__device__ __forceinline__ float d_action_fold(const rrl_loss_t& a, int b, int j, const float (&v)[2][NP]) {
    const int parts = a.da_parts > 1 ? a.da_parts : 1;
    float da = 0.f;
    da = d_action_head<NP>(da, v[0], parts, a.da_group);
    if (1 < a.n_heads) da = d_action_head<NP>(da, v[1], parts, a.da_group);
    if (a.n_heads > 2) {          // more than twin critics: the remaining heads the same way, one after the other
        const float* p = a.d_action + (long long)b * a.ld + j;
        for (int hd = 2; hd < a.n_heads; ++hd) {
            float w[NP];
#pragma unroll
            for (int t = 0; t < NP; ++t) w[t] = p[hd * a.head_stride + (t < parts ? t : 0) * a.da_part_stride];
            da = d_action_head<NP>(da, w, parts, a.da_group);
        }
    }
    return da;
}
__device__ __forceinline__ void psum_load(const float* p, long long idx, int np, long long ps, float (&v)[4]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = p[(np > k ? k * ps : 0) + idx];
}
__device__ __forceinline__ float psum_fold(const float (&v)[4], int np) {
    float x = v[0];
#pragma unroll
    for (int k = 1; k < 4; ++k) x = np > k ? x + v[k] : x;
    return x;
}
template <int KIND, int NP>
__device__ __forceinline__ void head_in_load(const rrl_loss_t& a, int b, int j, HeadIn<NP>& in) {
    d_action_load<NP>(a, b, j, in.da);
    if constexpr (KIND == RRL_LOSS_GAUSS_HEAD) {
        psum_load(a.out, 4 * b + j, a.n_part, a.part_stride, in.m);
        psum_load(a.out, 4 * b + 2 + j, a.n_part, a.part_stride, in.r);
        in.e = a.v0[2 * b + j];
        in.sc = a.v1[j];
        in.x2 = 0.f;
    } else {
        psum_load(a.out, 2 * b + j, a.n_part, a.part_stride, in.m);
        in.e = a.v0[2 * b + j];
        in.sc = a.v1[j];
        in.x2 = a.v2[j];
    }
}
// tanh-Gaussian head: d mean -> dx, d raw log-std -> ds (dout_at<RRL_LOSS_GAUSS_HEAD> for o = j and o = 2 + j)
template <int NP>
__device__ __forceinline__ void gauss_head_eval(const rrl_loss_t& a, int b, int j, const HeadIn<NP>& in, float& dx, float& ds) {
    const float da = d_action_fold<NP>(a, b, j, in.da);
    const float mean = psum_fold(in.m, a.n_part), raw = psum_fold(in.r, a.n_part);
    const float ls = fminf(fmaxf(raw, kLogSigMin), kLogSigMax);
    const float sd = expf(ls), e = in.e, sc = in.sc;
    const float y = tanhf(mean + sd * e);
    const float one_m = 1.f - y * y;
    dx = da * sc * one_m + a.f0 * (2.f * sc * y * one_m) / (sc * one_m + kEps);
    const bool inside = (raw >= kLogSigMin) & (raw <= kLogSigMax);
    ds = inside ? (dx * sd * e - a.f0) : 0.f;
}
// stochastic head (dout_at<RRL_LOSS_STOCH_HEAD>): returns dOut, `term` = the element's contribution to dlog_std[j]
template <int NP>
__device__ __forceinline__ float stoch_head_eval(const rrl_loss_t& a, int b, int j, const HeadIn<NP>& in, float& term) {
    const float t = tanhf(psum_fold(in.m, a.n_part));
    const float da = d_action_fold<NP>(a, b, j, in.da);
    const float sd = expf(fmaxf(in.sc, a.f0));
    term = (in.sc >= a.f0) ? da * sd * in.e : 0.f;
    return da * in.x2 * (1.f - t * t);
}

// dOut[g][b][o]; `term` = this element's contribution to loss[g] (critics), loss[0] (policies, g == 0 only)
// or dlog_std[o] (stochastic head)
template <int KIND>
__device__ __forceinline__ float dout_at(const rrl_loss_t& a, int B, int g, int b, int o, float& term) {
    const int np = a.n_part;
    const long long ps = a.part_stride;
    term = 0.f;
    if constexpr (KIND == RRL_LOSS_SAC_CRITIC) {
        float y = a.v1[b] + a.v2[b] * a.f0 *
                                (fminf(psum(a.out_t, b, np, ps), psum(a.out_t, B + b, np, ps)) - a.alpha[0] * a.v0[b]);
        if (a.v3) y -= a.v3[b];
        const float e = psum(a.out, (long long)g * B + b, np, ps) - y;
        term = e * e;
        return 2.f * e / B;
    } else if constexpr (KIND == RRL_LOSS_SAC_POLICY) {
        const float q0 = psum(a.out, b, np, ps), q1 = psum(a.out, B + b, np, ps);
        const float w0 = q0 < q1 ? 1.f : (q0 == q1 ? 0.5f : 0.f);
        if (g == 0) term = a.alpha[0] * a.v0[b] - fminf(q0, q1);
        return g == 0 ? -w0 / B : -(1.f - w0) / B;
    } else if constexpr (KIND == RRL_LOSS_QRISK_CRITIC) {
        const float y = a.v0[b] + a.v1[b] * a.f0 *
                                      fmaxf(sigm(psum(a.out_t, b, np, ps)), sigm(psum(a.out_t, B + b, np, ps)));
        const float q = sigm(psum(a.out, (long long)g * B + b, np, ps));
        const float e = q - y;
        term = e * e;
        return 2.f * e / B * q * (1.f - q);
    } else if constexpr (KIND == RRL_LOSS_QRISK_POLICY) {
        const float q0 = sigm(psum(a.out, b, np, ps)), q1 = sigm(psum(a.out, B + b, np, ps));
        const float w0 = q0 > q1 ? 1.f : (q0 == q1 ? 0.5f : 0.f);
        if (g == 0) term = fmaxf(q0, q1);
        return g == 0 ? w0 / B * q0 * (1.f - q0) : (1.f - w0) / B * q1 * (1.f - q1);
    } else {
        static_assert(KIND == RRL_LOSS_QRISK_POLICY, "the policy-head kinds: head_in_load + gauss_head_eval / stoch_head_eval");
        return 0.f;
    }
}

}  // namespace loss

struct HeadBwdArgs {
    rrl_loss_t la;
    int B, H, dout, need_w;
    const float* h2;
    const float* W3;
    float* dW3;
    float* db3;
    float* dh2;
};

template <int KIND>
constexpr int kind_dout() {
    return (KIND >= RRL_LOSS_SAC_CRITIC && KIND <= RRL_LOSS_QRISK_POLICY) ? 1
           : KIND == RRL_LOSS_GAUSS_HEAD ? 4 : KIND == RRL_LOSS_STOCH_HEAD ? 2 : 0;   // 0: run-time (plain dOut)
}

// dOut [B][dout] of head g into LDS, by a 256-thread workgroup, from the loss description (the formulas of update_kernels.hip:
// sac / qrisk *_grad, gauss / stoch_head_bwd); lsum = the thread's partial sums of the loss terms (critics: loss[g]; policies:
// loss[0], g == 0 only) or of dlog_std[0..1] (stochastic head) -- the per-thread partial sums of the stand-alone kernels, so
// every bit of the reduced values is theirs.  Used by the head backward and by the tiles of the paired launches.
// (rows [b0, b0 + nb) of the batch: all of it for the head backward and the weight-gradient tiles, the tile's own rows for the
// input-gradient tiles of the paired launches)
template <int KIND, int NP>
__device__ __forceinline__ void eval_policy_rows(const rrl_loss_t& la, int b0, int nb, float* dsh, float (&lsum)[2]) {
    constexpr int DOUT = kind_dout<KIND>();
    if constexpr (KIND == RRL_LOSS_GAUSS_HEAD) {
        // one thread per (row, action dim), two of them per pass: the mean and log-std gradients share tanh/exp
        const int e_end = 2 * (b0 + nb);
        for (int e0 = 2 * b0 + threadIdx.x; e0 < e_end; e0 += 512) {
            const bool two = e0 + 256 < e_end;
            const int e1 = two ? e0 + 256 : e0;
            loss::HeadIn<NP> in0, in1;
            loss::head_in_load<KIND>(la, e0 >> 1, e0 & 1, in0);
            loss::head_in_load<KIND>(la, e1 >> 1, e1 & 1, in1);
            float dx, ds;
            loss::gauss_head_eval(la, e0 >> 1, e0 & 1, in0, dx, ds);
            dsh[4 * (e0 >> 1) + (e0 & 1)] = dx;
            dsh[4 * (e0 >> 1) + 2 + (e0 & 1)] = ds;
            if (two) {
                loss::gauss_head_eval(la, e1 >> 1, e1 & 1, in1, dx, ds);
                dsh[4 * (e1 >> 1) + (e1 & 1)] = dx;
                dsh[4 * (e1 >> 1) + 2 + (e1 & 1)] = ds;
            }
        }
    } else {
        // one thread per batch row, both action dims: per-thread partial sums of dlog_std as in the stand-alone kernel
        for (int b = b0 + threadIdx.x; b < b0 + nb; b += 256) {
            loss::HeadIn<NP> in0, in1;
            loss::head_in_load<KIND>(la, b, 0, in0);
            loss::head_in_load<KIND>(la, b, 1, in1);
            float term;
            dsh[b * DOUT + 0] = loss::stoch_head_eval(la, b, 0, in0, term);
            lsum[0] += term;
            dsh[b * DOUT + 1] = loss::stoch_head_eval(la, b, 1, in1, term);
            lsum[1] += term;
        }
    }
}
// dOut [B][dout] of head g into LDS, by a 256-thread workgroup, from the loss description (the formulas of update_kernels.hip:
// sac / qrisk *_grad, gauss / stoch_head_bwd); lsum = the thread's partial sums of the loss terms (critics: loss[g]; policies:
// loss[0], g == 0 only) or of dlog_std[0..1] (stochastic head) -- the per-thread partial sums of the stand-alone kernels, so
// every bit of the reduced values is theirs.  Used by the head backward and by the tiles of the paired launches.
template <int KIND>
__device__ __forceinline__ void eval_dout_rows(const rrl_loss_t& la, int B, int g, float* dsh, float (&lsum)[2], int b0 = 0,
                                               int nb = -1) {
    if (nb < 0) nb = B;
    if constexpr (KIND == RRL_LOSS_GAUSS_HEAD || KIND == RRL_LOSS_STOCH_HEAD) {
        if (loss::needs_np16(la)) eval_policy_rows<KIND, 16>(la, b0, nb, dsh, lsum);      // (workgroup-uniform)
        else eval_policy_rows<KIND, 4>(la, b0, nb, dsh, lsum);
    } else {
        // one thread per batch row (one output): the per-thread partial sums of the loss terms are then the ones of the
        // stand-alone kernels (update_kernels.hip), and so is every bit of the reduced value
        for (int b = b0 + threadIdx.x; b < b0 + nb; b += 256) {
            float term;
            dsh[b] = loss::dout_at<KIND>(la, B, g, b, 0, term);
            lsum[0] += term;
        }
    }
}

template <int KIND>
__device__ __forceinline__ void head_bwd_loss_body(const HeadBwdArgs& hb, int bx, int g, float (*red)[4][kCols],
                                                   float* dsh) {
    const rrl_loss_t& la = hb.la;
    constexpr int DOUT = kind_dout<KIND>();
    const int B = hb.B, H = hb.H, dout = DOUT ? DOUT : hb.dout, need_w = hb.need_w;
    const float* __restrict__ h2 = hb.h2 + (long long)g * B * H;
    const float* __restrict__ W3 = hb.W3 + (long long)g * dout * H;
    float* __restrict__ dW3 = hb.dW3;
    float* __restrict__ db3 = hb.db3;
    float* __restrict__ dh2 = hb.dh2 ? hb.dh2 + (long long)g * B * H : nullptr;     // null: not wanted
    const int hc = threadIdx.x & (kCols - 1), slice = threadIdx.x / kCols;
    const int h = bx * kCols + hc;
    const bool hok = h < H;
    const int hh = hok ? h : H - 1;
    // operands of the batch loop that do not depend on dOut: requested before the loss formulas are evaluated
    float w[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) w[o] = o < dout ? W3[(long long)(o < dout ? o : 0) * H + hh] : 0.f;
    float a0[kUnroll];
#pragma unroll
    for (int it = 0; it < kUnroll; ++it) a0[it] = h2[(long long)min(slice + kSlices * it, B - 1) * H + hh];
    float lsum[2] = {0.f, 0.f};
    if constexpr (KIND == kPlainDOut) {
        const float* dO = la.out + (long long)g * B * dout;
        for (int e = threadIdx.x; e < B * dout; e += 256) dsh[e] = dO[e];
    } else {
        eval_dout_rows<KIND>(la, B, g, dsh, lsum);
    }
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    for (int b0 = 0; b0 < B; b0 += kSlices * kUnroll) {
        float a[kUnroll];
        if (b0 == 0) {
#pragma unroll
            for (int it = 0; it < kUnroll; ++it) a[it] = a0[it];
        } else {
#pragma unroll
            for (int it = 0; it < kUnroll; ++it) a[it] = h2[(long long)min(b0 + slice + kSlices * it, B - 1) * H + hh];
        }
        const bool whole = (b0 + kSlices * kUnroll <= B) & hok;     // uniform for hok-uniform column blocks
        if (whole) {                                                  // no per-row bounds checks, no store predicates
#pragma unroll
            for (int it = 0; it < kUnroll; ++it) {
                const int b = b0 + slice + kSlices * it;
                float d = 0.f;
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    if (o < (DOUT ? DOUT : 4)) {
                        const float go = (DOUT || o < dout) ? dsh[b * dout + (o < dout ? o : 0)] : 0.f;
                        d = fmaf(go, w[o], d);
                        acc[o] = fmaf(go, a[it], acc[o]);
                    }
                }
                if (hb.dh2) dh2[(long long)b * H + h] = a[it] > 0.f ? d : 0.f;
            }
        } else {
#pragma unroll
            for (int it = 0; it < kUnroll; ++it) {
                const int b = b0 + slice + kSlices * it;
                if (b < B) {
                    float d = 0.f;
#pragma unroll
                    for (int o = 0; o < 4; ++o) {
                        const float go = o < dout ? dsh[b * dout + o] : 0.f;
                        d = fmaf(go, w[o], d);
                        acc[o] = fmaf(go, a[it], acc[o]);
                    }
                    if (hok && hb.dh2) dh2[(long long)b * H + h] = a[it] > 0.f ? d : 0.f;
                }
            }
        }
    }
    if (need_w) {
#pragma unroll
        for (int o = 0; o < 4; ++o) red[slice][o][hc] = acc[o];
    }
    // bias gradient (column sums of dOut) and the loss scalars / dlog_std: workgroup bx == 0 reduces up to 4 + 2 values
    // over its 256 threads -- DPP sums inside the 16-lane rows, the 16 row sums through LDS, ONE barrier (the serial
    // 256-term bias loop and the 8-step barrier tree of the loss were ~1.5 us of this kernel's critical path)
    constexpr bool per_head = KIND == RRL_LOSS_SAC_CRITIC || KIND == RRL_LOSS_QRISK_CRITIC;
    const bool want_loss = KIND != kPlainDOut && KIND != RRL_LOSS_GAUSS_HEAD && la.loss && (per_head || g == 0);
    float* tail = dsh + 1024 * 4 - 6 * 16;            // dsh holds B * dout <= 4096 floats only when B = 1024, dout = 4:
    const bool tail_free = B * dout <= 1024 * 4 - 6 * 16;   // then the scalars take the slow path below
    float part[6] = {0.f, 0.f, 0.f, 0.f, lsum[0], lsum[1]};
    if (bx == 0 && tail_free) {
        for (int e = threadIdx.x; e < B; e += 256) {
#pragma unroll
            for (int o = 0; o < 4; ++o)
                if (o < dout) part[o] += dsh[e * dout + o];
        }
    }
    __syncthreads();                                   // red[] complete (need_w); dsh reads above done before tail writes
    if (need_w && slice < dout && hok) {
        float sum = 0.f;
#pragma unroll
        for (int sl = 0; sl < kSlices; ++sl) sum += red[sl][slice][hc];
        dW3[((long long)g * dout + slice) * H + h] = sum;
    }
    if (bx != 0 || (!need_w && !want_loss)) return;
    if (tail_free) {
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const float rsum = row16_sum(part[k]);
            if ((threadIdx.x & 15) == 0) tail[k * 16 + (threadIdx.x >> 4)] = rsum;
        }
        __syncthreads();
        if (threadIdx.x < 6) {
            float tot = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) tot += tail[threadIdx.x * 16 + r];
            const int k = threadIdx.x;
            if (k < 4) {
                if (need_w && k < dout) db3[g * dout + k] = tot;
            } else if (want_loss) {
                if constexpr (KIND == RRL_LOSS_STOCH_HEAD) la.loss[k - 4] = tot;
                else if (k == 4) la.loss[per_head ? g : 0] = tot / B;
            }
        }
        return;
    }
    // B * dout too large for the LDS tail (B = 1024 with four outputs): serial sums by single threads
    if (need_w && threadIdx.x >= 128 && threadIdx.x < 128 + (unsigned)dout) {
        const int o = threadIdx.x - 128;
        float sum = 0.f;
        for (int b = 0; b < B; ++b) sum += dsh[b * dout + o];
        db3[g * dout + o] = sum;
    }
    if (!want_loss) return;
    __syncthreads();
    float* r0 = &red[0][0][0];          // 1024 floats: two arrays of 256
    r0[threadIdx.x] = lsum[0];
    r0[256 + threadIdx.x] = lsum[1];
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            r0[threadIdx.x] += r0[threadIdx.x + off];
            r0[256 + threadIdx.x] += r0[256 + threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if constexpr (KIND == RRL_LOSS_STOCH_HEAD) {
            la.loss[0] = r0[0];
            la.loss[1] = r0[256];
        } else {
            la.loss[per_head ? g : 0] = r0[0] / B;
        }
    }
}

template <int KIND>
__global__ __launch_bounds__(256) void head_bwd_loss_kernel(HeadBwdArgs hb) {
    __shared__ float red[kSlices][4][kCols];
    __shared__ float dsh[1024 * 4];
    head_bwd_loss_body<KIND>(hb, blockIdx.x, blockIdx.y, red, dsh);
}

__device__ __forceinline__ void head_bwd_dispatch(const HeadBwdArgs& hb, int bx, int g, float (*red)[4][kCols],
                                                  float* dsh) {
    switch (hb.la.kind) {
        case RRL_LOSS_SAC_CRITIC: head_bwd_loss_body<RRL_LOSS_SAC_CRITIC>(hb, bx, g, red, dsh); break;
        case RRL_LOSS_SAC_POLICY: head_bwd_loss_body<RRL_LOSS_SAC_POLICY>(hb, bx, g, red, dsh); break;
        case RRL_LOSS_QRISK_CRITIC: head_bwd_loss_body<RRL_LOSS_QRISK_CRITIC>(hb, bx, g, red, dsh); break;
        case RRL_LOSS_QRISK_POLICY: head_bwd_loss_body<RRL_LOSS_QRISK_POLICY>(hb, bx, g, red, dsh); break;
        case RRL_LOSS_GAUSS_HEAD: head_bwd_loss_body<RRL_LOSS_GAUSS_HEAD>(hb, bx, g, red, dsh); break;
        case RRL_LOSS_STOCH_HEAD: head_bwd_loss_body<RRL_LOSS_STOCH_HEAD>(hb, bx, g, red, dsh); break;
        default: head_bwd_loss_body<kPlainDOut>(hb, bx, g, red, dsh); break;
    }
}

// independent head backwards (e.g. critic loss on (s,a) and policy loss on (s,pi)) in one launch: flat grid over
// (problem, head, column block)
__device__ __forceinline__ void globalize(HeadBwdArgs& hb) {
    rrl_loss_t& l = hb.la;
    rrl_pack::to_global_all(l.out, l.out_t, l.v0, l.v1, l.v2, l.v3, l.alpha, l.d_action, l.loss, hb.h2, hb.W3, hb.dW3, hb.db3,
                            hb.dh2);
}
struct HeadBwdGroup {
    HeadBwdArgs p[kMaxGroup];
    int G[kMaxGroup], blocks_x[kMaxGroup];
    int first[kMaxGroup + 1];
    int n;
};

// the member (blockIdx.y) of a group launch, wherever the group lives (kernel arguments of the solo launch, the plan's device
// copy of a packed one): its block arrives in one batch of scalar loads (mlp_common.hpp)
__device__ __forceinline__ void head_bwd_member_body(const HeadBwdGroup& hg, int local, float (*red)[4][kCols], float* dsh) {
    const int k = blockIdx.y;
    HeadBwdArgs hb = hg.p[k];
    const int blocks_x = hg.blocks_x[k], G = hg.G[k];
    globalize(hb);
    const rrl_loss_t& l = hb.la;
    arrive_together(hb.B, hb.H, hb.dout, hb.need_w, l.kind, l.n_part, l.part_stride, l.f0, l.ld, l.n_heads, l.head_stride,
                    l.da_parts, l.da_part_stride, l.da_group, blocks_x, G);
    if (local >= blocks_x * G) return;                  // (an unused member slot has blocks_x = 0)
    head_bwd_dispatch(hb, local % blocks_x, local / blocks_x, red, dsh);
}

// solo launch: grid (column blocks x heads of the largest member, members): the member comes out of the grid (mlp_common.hpp)
__global__ __launch_bounds__(256) void head_bwd_group_kernel(HeadBwdGroup hg) {
    __shared__ float red[kSlices][4][kCols];
    __shared__ float dsh[1024 * 4];
    head_bwd_member_body(hg, blockIdx.x, red, dsh);
}

// packed launch: the same grid per seed under the XCD-aware placement (pack.hpp, rrl_pack::locate_grid)
__global__ __launch_bounds__(256) void head_bwd_pack_kernel(const HeadBwdGroup* __restrict__ groups, rrl_pack::Idx ix) {
    __shared__ float red[kSlices][4][kCols];
    __shared__ float dsh[1024 * 4];
    RRL_PACK_LOCATE(ix, groups, s, local);
    head_bwd_member_body(groups[s], local, red, dsh);
}

// ---- head backward + hidden backward of the critic-loss kinds in ONE launch ----------------------------------------------
// The two stages are dependent through dh2 [G,B,H] only, and with one output per row dh2[b][h] = h2[b][h] > 0 ? dOut[b] W3[h] : 0
// is cheaper to derive where the hidden backward consumes it than to write and read back: the tiles evaluate dOut[b] for their
// head themselves (the loss formulas: a handful of loads and a sigmoid per row) and gemm16_tile<GEN> builds the operand from
// the saved activation h2.  The head-backward workgroups (dW3, db3, the loss scalars) ride in the same grid and no longer
// write dh2.  One launch instead of two dependent ones (~5.5 us each time).  Every value is the one the two-launch path
// produces: same dOut formulas, fmaf(go, w, 0) for dh2, the same MFMA steps in the same order (the K order does not depend on
// the panel width).
// grid (x, member, 3): z = 0 input-gradient tiles (NN), z = 1 weight-gradient tiles (TN) -- four tiles per 256-thread
// workgroup, one wave each, 64-wide K panels (10 KB of LDS per tile) -- z = 2 the head backward's column blocks.
constexpr int kPairPanel = 64;
constexpr int kPairTileFloats = 2 * kPairPanel * kLd;
struct PairJobs {
    HiddenJob job[kMaxGroup][2];      // [member][0: NN, 1: TN]; ga.A = the saved activation h2
    HeadBwdArgs head[kMaxGroup];      // dh2 = null
    int blocks_x[kMaxGroup];
};
// DOUT = outputs of the members' heads (every member of a paired launch has the same): 1 the four critic-loss kinds, 4 the
// tanh-Gaussian policy head, 2 the stochastic policy head
// rows [b0, b0 + nb) of the head's dOut (a weight-gradient workgroup sums over the whole batch; an input-gradient one needs
// the rows of its own tiles only)
template <int DOUT>
__device__ __forceinline__ void pair_eval_dout(const rrl_loss_t& la, int B, int g, float* dsh, int b0, int nb) {
    float unused[2] = {0.f, 0.f};
    if constexpr (DOUT == 4) {
        eval_dout_rows<RRL_LOSS_GAUSS_HEAD>(la, B, g, dsh, unused, b0, nb);
    } else if constexpr (DOUT == 2) {
        eval_dout_rows<RRL_LOSS_STOCH_HEAD>(la, B, g, dsh, unused, b0, nb);
    } else {
        switch (la.kind) {
            case RRL_LOSS_SAC_CRITIC: eval_dout_rows<RRL_LOSS_SAC_CRITIC>(la, B, g, dsh, unused, b0, nb); break;
            case RRL_LOSS_SAC_POLICY: eval_dout_rows<RRL_LOSS_SAC_POLICY>(la, B, g, dsh, unused, b0, nb); break;
            case RRL_LOSS_QRISK_CRITIC: eval_dout_rows<RRL_LOSS_QRISK_CRITIC>(la, B, g, dsh, unused, b0, nb); break;
            default: eval_dout_rows<RRL_LOSS_QRISK_POLICY>(la, B, g, dsh, unused, b0, nb); break;
        }
    }
}
constexpr int kPairDsh = 1024;                         // floats of the dOut tile in LDS: B * DOUT <= 1024
template <int DOUT>
constexpr int pair_smem_floats() {
    return 4 * kPairTileFloats + kPairDsh + (DOUT > 1 ? DOUT * 256 : 0);      // tiles | dOut | W3 copy (policy heads, H <= 256)
}

template <int DOUT>
__device__ __forceinline__ void backward_pair_body(const PairJobs& pj, int x, float* smem) {
    const int k = blockIdx.y, z = blockIdx.z;
    HeadBwdArgs hb = pj.head[k];
    const int blocks_x = pj.blocks_x[k];
    globalize(hb);
    const rrl_loss_t& l = hb.la;
    if (z == 2) {
        const int G = pj.job[k][0].G;
        arrive_together(hb.B, hb.H, hb.dout, hb.need_w, l.kind, l.n_part, l.part_stride, l.f0, l.ld, l.n_heads, l.head_stride,
                        l.da_parts, l.da_part_stride, l.da_group, blocks_x, G);
        const int local = x;
        if (local >= blocks_x * G) return;
        // (only the kinds of this DOUT: the other kinds' code would double the kernel)
        float (*red)[4][kCols] = reinterpret_cast<float (*)[4][kCols]>(smem);
        const int bx = local % blocks_x, g = local / blocks_x;
        if constexpr (DOUT == 4) {
            head_bwd_loss_body<RRL_LOSS_GAUSS_HEAD>(hb, bx, g, red, smem + 1024);
        } else if constexpr (DOUT == 2) {
            head_bwd_loss_body<RRL_LOSS_STOCH_HEAD>(hb, bx, g, red, smem + 1024);
        } else {
            switch (l.kind) {
                case RRL_LOSS_SAC_CRITIC: head_bwd_loss_body<RRL_LOSS_SAC_CRITIC>(hb, bx, g, red, smem + 1024); break;
                case RRL_LOSS_SAC_POLICY: head_bwd_loss_body<RRL_LOSS_SAC_POLICY>(hb, bx, g, red, smem + 1024); break;
                case RRL_LOSS_QRISK_CRITIC: head_bwd_loss_body<RRL_LOSS_QRISK_CRITIC>(hb, bx, g, red, smem + 1024); break;
                default: head_bwd_loss_body<RRL_LOSS_QRISK_POLICY>(hb, bx, g, red, smem + 1024); break;
            }
        }
        return;
    }
    // (the head's block and the job's block requested in ONE batch -- ~100 scalars -- measured slower: 11.4 us against 10.5
    // for the critic-loss launch; the compiler splits such a batch where it runs out of scalar registers)
    HiddenJob j = pj.job[k][z];
    GemmArgs& ga = j.ga;
    globalize(ga);
    arrive_together(ga.M, ga.N, ga.K, ga.lda, ga.ldb, ga.ldc, ga.ldmask, ga.sA, ga.sB, ga.sC, ga.sMask, ga.sColsum,
                    ga.relu, ga.accumulate, ga.first_stride, ga.ldx, ga.din, ga.G, ga.skip_c, ga.dx_fold, j.tiles, j.tiles_x, j.G,
                    hb.B, hb.H, l.kind, l.n_part, l.part_stride, l.f0, l.ld, l.n_heads, l.head_stride, l.da_parts, l.da_part_stride,
                    l.da_group);
    const int t0 = 4 * x;
    if (t0 >= j.tiles * j.G) return;
    const int g = t0 / j.tiles;                    // tiles % 4 == 0 (host-checked): the four tiles serve one head
    float* dsh = smem + 4 * kPairTileFloats;
    float* w3s = dsh + kPairDsh;
    const float* w3 = hb.W3 + (long long)g * DOUT * hb.H;
    // dOut of the head, evaluated by the whole workgroup BEHIND the four tiles' operand requests (all four waves hold a tile:
    // every wave passes the barrier once); the policy heads also copy their W3 [DOUT][H] to LDS for the NN tiles
    const int wave = threadIdx.x >> 6, t = t0 + wave - g * j.tiles;
    const int row_tile = (t0 - g * j.tiles) / j.tiles_x;      // of the workgroup's four tiles (tiles_x % 4 == 0: one tile row)
    const auto eval_dout = [&]() {
        if constexpr (DOUT > 1) {
            for (int e = threadIdx.x; e < DOUT * hb.H; e += 256) w3s[e] = w3[e];
        }
        if (z == 1) pair_eval_dout<DOUT>(l, hb.B, g, dsh, 0, hb.B);                 // dW2 sums over the batch
        else pair_eval_dout<DOUT>(l, hb.B, g, dsh, row_tile * kTile, kTile);          // dh1: the tiles' own 16 rows
        __syncthreads();
    };
    float* As = smem + wave * kPairTileFloats;
    float* Bs = As + kPairPanel * kLd;
    using Behind = decltype(eval_dout);
    if (z == 1)
        gemm16_tile<2, true, kPairPanel, true, true, true, Behind, DOUT, true>(ga, As, Bs, t % j.tiles_x, t / j.tiles_x, g, dsh,
                                                                               w3, eval_dout, w3s, false, nullptr, smem,
                                                                               kPairTileFloats);
    else {    // (128-wide panels for these tiles -- they stage one operand only, 10 KB either way -- measured: 10.4 -> 13.8 us)
        const bool fold = ga.dx_part && ga.dx_fold;          // workgroup-uniform
        float sx = 0.f;
        gemm16_tile<1, true, kPairPanel, true, true, true, Behind, DOUT, true>(ga, As, Bs, t % j.tiles_x, t / j.tiles_x, g, dsh,
                                                                               w3, eval_dout, w3s, fold, &sx, smem, kPairTileFloats);
        if (fold) {
            // the workgroup's four tiles are four consecutive column tiles of one row tile (tiles_x % 4 == 0): their dx
            // partials as ONE sum ((p0 + p1) + p2) + p3 -- the policy-head backward then reads H / 64 partials per critic
            // head instead of H / 16 (da_group of rrl_loss_t: the same bits either way)
            __syncthreads();                       // every wave is done with its tile (and with dsh)
            dsh[threadIdx.x] = sx;
            __syncthreads();
            if (wave == 0) {
                const int lane = threadIdx.x, rr = lane & 15, dd = lane >> 4;
                float gs = dsh[lane];
                gs += dsh[64 + lane];
                gs += dsh[128 + lane];
                gs += dsh[192 + lane];
                const int bx = t % j.tiles_x, by = t / j.tiles_x;
                if (dd < ga.din)
                    ga.dx_part[(((long long)(bx >> 2) * ga.G + g) * ga.M + by * kTile + rr) * ga.din + dd] = gs;
            }
        }
    }
}
template <int DOUT>
__global__ __launch_bounds__(256) void backward_pair_kernel(PairJobs pj) {
    __shared__ __attribute__((aligned(16))) float smem[pair_smem_floats<DOUT>()];      // tiles | dOut | W3; or red | dsh of the head path
    backward_pair_body<DOUT>(pj, blockIdx.x, smem);
}
// the same launch for S seeds (pack.hpp): grid (x under the XCD-aware placement, members, 3) -- 22 -> 17 launches per packed
// iteration, as in the solo graph
template <int DOUT>
__global__ __launch_bounds__(256) void backward_pair_pack_kernel(const PairJobs* __restrict__ groups, rrl_pack::Idx ix) {
    __shared__ __attribute__((aligned(16))) float smem[pair_smem_floats<DOUT>()];
    RRL_PACK_LOCATE(ix, groups, s, local);
    backward_pair_body<DOUT>(groups[s], local, smem);
}

// ---- block form of the paired launch for the packed backward (from pack_block(S) seeds on) ------------------------------
// backward_pair_body with gemm_block<.., DOUT> in the place of the four tiles: a four-wave workgroup owns a 32 x 64 block of dh1
// (z = 0) or dW2 (z = 1) and derives dh2 as it stages / consumes the saved activation; the head backward's column blocks
// (z = 2: dW3, db3, the loss scalars) ride in the same grid.  22 -> 17 launches per packed iteration at every seed count.
template <int DOUT>
constexpr int pair_block_smem_floats() {
    return BlkLds<1, 2>::kFloats + kPairDsh + DOUT * 256;       // staged panels | dOut | W3 copy
}
template <int DOUT>
__global__ __launch_bounds__(256) void backward_pair_block_pack_kernel(const PairJobs* __restrict__ groups, rrl_pack::Idx ix) {
    __shared__ __attribute__((aligned(16))) float smem[pair_block_smem_floats<DOUT>() > 5120 ? pair_block_smem_floats<DOUT>() : 5120];
    RRL_PACK_LOCATE(ix, groups, s, local);
    const PairJobs& pj = groups[s];
    const int k = blockIdx.y, z = blockIdx.z;
    HeadBwdArgs hb = pj.head[k];
    const int blocks_x = pj.blocks_x[k];
    globalize(hb);
    const rrl_loss_t& l = hb.la;
    if (z == 2) {
        const int G = pj.job[k][0].G;
        arrive_together(hb.B, hb.H, hb.dout, hb.need_w, l.kind, l.n_part, l.part_stride, l.f0, l.ld, l.n_heads, l.head_stride,
                        l.da_parts, l.da_part_stride, l.da_group, blocks_x, G);
        if (local >= blocks_x * G) return;
        float (*red)[4][kCols] = reinterpret_cast<float (*)[4][kCols]>(smem);
        const int bx = local % blocks_x, g = local / blocks_x;
        if constexpr (DOUT == 4) {
            head_bwd_loss_body<RRL_LOSS_GAUSS_HEAD>(hb, bx, g, red, smem + 1024);
        } else if constexpr (DOUT == 2) {
            head_bwd_loss_body<RRL_LOSS_STOCH_HEAD>(hb, bx, g, red, smem + 1024);
        } else {
            switch (l.kind) {
                case RRL_LOSS_SAC_CRITIC: head_bwd_loss_body<RRL_LOSS_SAC_CRITIC>(hb, bx, g, red, smem + 1024); break;
                case RRL_LOSS_SAC_POLICY: head_bwd_loss_body<RRL_LOSS_SAC_POLICY>(hb, bx, g, red, smem + 1024); break;
                case RRL_LOSS_QRISK_CRITIC: head_bwd_loss_body<RRL_LOSS_QRISK_CRITIC>(hb, bx, g, red, smem + 1024); break;
                default: head_bwd_loss_body<RRL_LOSS_QRISK_POLICY>(hb, bx, g, red, smem + 1024); break;
            }
        }
        return;
    }
    HiddenJob j = pj.job[k][z];                    // tile counts in BLOCK units
    GemmArgs& ga = j.ga;
    globalize(ga);
    arrive_together(ga.M, ga.N, ga.K, ga.lda, ga.ldb, ga.ldc, ga.ldmask, ga.sA, ga.sB, ga.sC, ga.sMask, ga.sColsum,
                    ga.relu, ga.accumulate, ga.first_stride, ga.ldx, ga.din, ga.G, ga.skip_c, ga.dx_fold, j.tiles, j.tiles_x, j.G,
                    hb.B, hb.H, l.kind, l.n_part, l.part_stride, l.f0, l.ld, l.n_heads, l.head_stride, l.da_parts, l.da_part_stride,
                    l.da_group);
    if (local >= j.tiles * j.G) return;
    const int g = local / j.tiles, b = local - g * j.tiles;
    float* dsh = smem + BlkLds<1, 2>::kFloats;
    float* w3s = dsh + kPairDsh;
    const float* w3 = hb.W3 + (long long)g * DOUT * hb.H;
    const auto eval_dout = [&]() {
        for (int e = threadIdx.x; e < DOUT * hb.H; e += 256) w3s[e] = w3[e];
        if (z == 1) pair_eval_dout<DOUT>(l, hb.B, g, dsh, 0, hb.B);                                   // dW2 sums over the batch
        else pair_eval_dout<DOUT>(l, hb.B, g, dsh, (b / j.tiles_x) * BlkLds<1, 2>::BM, BlkLds<1, 2>::BM);   // dh1: the block's own 32 rows
        __syncthreads();
    };
    using Behind = decltype(eval_dout);
    if (z == 1) gemm_block<2, 1, 2, DOUT, Behind>(ga, smem, b % j.tiles_x, b / j.tiles_x, g, dsh, w3s, eval_dout);
    else gemm_block<1, 1, 2, DOUT, Behind>(ga, smem, b % j.tiles_x, b / j.tiles_x, g, dsh, w3s, eval_dout);
}

// input-layer backward: dh1 [G,B,H] (already masked by relu'), x [B,din] shared by the heads
//   dW1[g][h][d] = sum_b dh1[g][b][h] x[b][d]     db1[g][h] = sum_b dh1[g][b][h]          (need_w)
//   dx[g][b][d]  = sum_h dh1[g][b][h] W1[g][h][d]                                         (need_x)
// grid (H / 16 + B / 4, G) x 256 threads: the first H/16 blocks do the weight gradients (16 columns x 16
// batch slices, as above), the remaining ones the input gradients (one wavefront per batch row).
struct InputBwdArgs {
    int B, H, din, ldx, need_w, need_x;
    const float* dh1;
    const float* x;
    const float* W1;
    float* dW1;
    float* db1;
    float* dx;
};

__device__ __forceinline__ void input_bwd_body(const InputBwdArgs& ib, int bx, int g, float (*red)[5][kCols]) {
    const int B = ib.B, H = ib.H, din = ib.din, ldx = ib.ldx, need_w = ib.need_w, need_x = ib.need_x;
    const float* __restrict__ dh1 = ib.dh1;
    const float* __restrict__ x = ib.x;
    const float* __restrict__ W1 = ib.W1;
    float* __restrict__ dW1 = ib.dW1;
    float* __restrict__ db1 = ib.db1;
    float* __restrict__ dx = ib.dx;
    const int wblocks = need_w ? (H + kCols - 1) / kCols : 0;
    if (bx < wblocks) {
        const int hc = threadIdx.x & (kCols - 1), slice = threadIdx.x / kCols, h = bx * kCols + hc;
        const bool hok = h < H;
        const int hh = hok ? h : H - 1;
        float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        for (int b0 = 0; b0 < B; b0 += kSlices * kUnroll) {
            float d[kUnroll], xv[kUnroll][4];
#pragma unroll
            for (int it = 0; it < kUnroll; ++it) {
                const int b = min(b0 + slice + kSlices * it, B - 1);
                d[it] = dh1[((long long)g * B + b) * H + hh];
#pragma unroll
                for (int k = 0; k < 4; ++k) xv[it][k] = k < din ? x[(long long)b * ldx + k] : 0.f;
            }
#pragma unroll
            for (int it = 0; it < kUnroll; ++it) {
                if (b0 + slice + kSlices * it < B) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[k] = fmaf(d[it], xv[it][k], acc[k]);
                    acc[4] += d[it];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 5; ++k) red[slice][k][hc] = acc[k];
        __syncthreads();
        if (slice < 5 && hok && (slice == 4 || slice < din)) {
            float sum = 0.f;
#pragma unroll
            for (int sl = 0; sl < kSlices; ++sl) sum += red[sl][slice][hc];
            if (slice == 4) db1[(long long)g * H + h] = sum;
            else dW1[((long long)g * H + h) * din + slice] = sum;
        }
        return;
    }
    if (!need_x) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = (bx - wblocks) * 4 + wave;
    if (b >= B) return;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int h0 = 0; h0 < H; h0 += 256) {
        float d[4], wv[4][4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int h = min(h0 + lane + 64 * it, H - 1);
            d[it] = dh1[((long long)g * B + b) * H + h];
#pragma unroll
            for (int k = 0; k < 4; ++k) wv[it][k] = k < din ? W1[((long long)g * H + h) * din + k] : 0.f;
        }
#pragma unroll
        for (int it = 0; it < 4; ++it)
            if (h0 + lane + 64 * it < H)
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] = fmaf(d[it], wv[it][k], acc[k]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float v = acc[k];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if (lane == 0 && k < din) dx[((long long)g * B + b) * din + k] = v;
    }
}

__global__ __launch_bounds__(256) void input_bwd_kernel(InputBwdArgs ib) {
    __shared__ float red[kSlices][5][kCols];
    input_bwd_body(ib, blockIdx.x, blockIdx.y, red);
}

struct InputBwdGroup {
    InputBwdArgs p[kMaxGroup];
    int G[kMaxGroup], blocks_x[kMaxGroup];
    int first[kMaxGroup + 1];
    int n;
};

__global__ __launch_bounds__(256) void input_bwd_group_kernel(InputBwdGroup ig) {
    __shared__ float red[kSlices][5][kCols];
    int k = 0;
    while (k + 1 < ig.n && (int)blockIdx.x >= ig.first[k + 1]) ++k;
    const int local = blockIdx.x - ig.first[k];
    input_bwd_body(ig.p[k], local % ig.blocks_x[k], local / ig.blocks_x[k], red);
}

}  // namespace

extern "C" {

int rrl_gemm_f32(int mode, int G, int M, int N, int K, const float* A, int lda, long long sA,
                 const float* B, int ldb, long long sB, float* C, int ldc, long long sC,
                 const float* bias, long long sBias, int relu, const float* mask, int ldmask,
                 long long sMask, float* colsum, long long sColsum, int accumulate, void* stream) {
    if (mode < 0 || mode > 2 || !A || !B || !C) return RRL_EINVAL;
    if (G <= 0 || M <= 0 || N <= 0 || K <= 0 || G > 65535) return RRL_ERANGE;
    GemmArgs a{A, B, C, bias, mask, colsum, M, N, K, lda, ldb, ldc, ldmask,
               sA, sB, sC, sBias, sMask, sColsum, relu, accumulate};
    const dim3 grid((N + kTile - 1) / kTile, (M + kTile - 1) / kTile, G), block(64);
    hipStream_t st = (hipStream_t)stream;
    auto aligned = [](const void* p, int ld, long long stride) {
        return (reinterpret_cast<uintptr_t>(p) & 15) == 0 && (ld % 4) == 0 && (stride % 4) == 0;
    };
    const bool fast = (M % kTile) == 0 && (N % kTile) == 0 && (K % kPanel) == 0 && aligned(A, lda, sA) &&
                      aligned(B, ldb, sB);
    if (fast) {
        if (mode == 0) hipLaunchKernelGGL((gemm16_kernel<0, true>), grid, block, 0, st, a);
        else if (mode == 1) hipLaunchKernelGGL((gemm16_kernel<1, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((gemm16_kernel<2, true>), grid, block, 0, st, a);
    } else {
        if (mode == 0) hipLaunchKernelGGL((gemm16_kernel<0, false>), grid, block, 0, st, a);
        else if (mode == 1) hipLaunchKernelGGL((gemm16_kernel<1, false>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((gemm16_kernel<2, false>), grid, block, 0, st, a);
    }
    return check_launch();
}

// GemmArgs of one stack's hidden-layer backward (TN: dW2 + db2; NN: dh1)
static bool hidden_args(int G, int B, int H, const float* dh2, const float* h1, const float* W2, float* dW2, float* db2,
                        float* dh1, GemmArgs& tn, GemmArgs& nn, const rrl_first_layer_t* fl = nullptr) {
    const long long sAct = (long long)B * H, sW = (long long)H * H;
    // TN: dW2 [H,H] = dh2^T [H,B] . h1 [B,H], column sums of dh2 -> db2        (A = dh2, K = B)
    tn = GemmArgs{dh2, h1, dW2, nullptr, nullptr, db2, H, H, B, H, H, H, 0, sAct, sAct, sW, 0, 0, (long long)H, 0, 0};
    // NN: dh1 [B,H] = dh2 [B,H] . W2 [H,H], masked by h1 > 0                     (K = H)
    nn = GemmArgs{dh2, W2, dh1, nullptr, h1, nullptr, B, H, H, H, H, H, H, sAct, sW, sAct, 0, sAct, 0, 0, 0};
    if (fl && fl->x) {
        nn.x = fl->x; nn.W1 = fl->W1; nn.first_part = fl->first_part; nn.dx_part = fl->dx_part;
        nn.first_stride = fl->first_stride; nn.ldx = fl->ldx; nn.din = fl->din; nn.G = G;
        nn.skip_c = dh1 == nullptr;
        nn.dx_fold = fl->dx_part ? fl->dx_fold : 0;
    }
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    return (H % kTile) == 0 && (B % kTile) == 0 && (H % kPanel) == 0 && (B % kPanel) == 0 && al(dh2) && al(h1) && al(W2);
}

int rrl_mlp_hidden_backward(int G, int B, int H, const float* dh2, const float* h1, const float* W2, float* dW2,
                            float* db2, float* dh1, void* stream) {
    if (!dh2 || !h1 || !W2 || !dW2 || !db2 || !dh1) return RRL_EINVAL;
    if (G <= 0 || G > 65535 || B <= 0 || H <= 0) return RRL_ERANGE;
    GemmArgs tn, nn;
    const bool fast = hidden_args(G, B, H, dh2, h1, W2, dW2, db2, dh1, tn, nn);
    const int tx = (H + kTile - 1) / kTile, ty = tx, nx = tx, ny = (B + kTile - 1) / kTile;
    const dim3 grid(tx * ty + nx * ny, G), block(64);
    if (fast) hipLaunchKernelGGL(gemm16_pair_kernel<true>, grid, block, 0, (hipStream_t)stream, tn, nn, tx, tx * ty, nx);
    else hipLaunchKernelGGL(gemm16_pair_kernel<false>, grid, block, 0, (hipStream_t)stream, tn, nn, tx, tx * ty, nx);
    return check_launch();
}

static int build_hidden_group(int n, const rrl_hidden_bwd_t* ps, HiddenGroup& hg) {
    if (!ps || n <= 0 || n > kMaxGroup) return RRL_EINVAL;
    hg = HiddenGroup{};
    hg.n = n;
    hg.first[0] = 0;
    for (int k = 0; k < n; ++k) {
        const rrl_hidden_bwd_t& p = ps[k];
        if (!p.dh2 || !p.h1 || !p.W2 || ((p.dW2 == nullptr) != (p.db2 == nullptr))) return RRL_EINVAL;
        if (p.G <= 0 || p.G > 65535 || p.B <= 0 || p.H <= 0) return RRL_ERANGE;
        const bool first = p.first.x != nullptr;
        if (first && (!p.first.W1 || p.first.din <= 0 || p.first.din > 4 || (!p.first.first_part && !p.first.dx_part)))
            return RRL_EINVAL;
        if (!p.dh1 && !first) return RRL_EINVAL;
        hg.fast[k] = hidden_args(p.G, p.B, p.H, p.dh2, p.h1, p.W2, p.dW2, p.db2, p.dh1, hg.tn[k], hg.nn[k], &p.first);
        if (first && !hg.fast[k]) return RRL_ERANGE;          // the fused first layer needs full, aligned tiles
        const int tx = (p.H + kTile - 1) / kTile, ny = (p.B + kTile - 1) / kTile;
        hg.tn_tiles_x[k] = tx;
        hg.tn_tiles[k] = p.dW2 ? tx * tx : 0;           // no weight gradient wanted: input gradient tiles only
        hg.nn_tiles_x[k] = tx;
        hg.per_head[k] = hg.tn_tiles[k] + tx * ny;
        hg.first[k + 1] = hg.first[k] + hg.per_head[k] * p.G;
    }
    for (int k = n; k < kMaxGroup; ++k) hg.first[k + 1] = hg.first[n];
    return RRL_OK;
}

// some member asks for folded dx partials (rrl_first_layer_t.dx_fold): the paired launches and the 32 x 64 block form fold
// inside a workgroup, a one-tile workgroup cannot
static bool wants_fold(int n, const rrl_hidden_bwd_t* ps) {
    for (int k = 0; k < n; ++k)
        if (ps[k].first.x && ps[k].first.dx_part && ps[k].first.dx_fold) return true;
    return false;
}

// the jobs of a group launch (grid (x, member, {NN, TN})) from the tile / block counts of a HiddenGroup; returns the
// workgroups of the largest job
static int hidden_jobs(int n, const rrl_hidden_bwd_t* ps, const HiddenGroup& hg, HiddenJobs& hj) {
    hj = HiddenJobs{};
    int most = 1;
    for (int k = 0; k < n; ++k) {
        const int nn_tiles = hg.per_head[k] - hg.tn_tiles[k];
        hj.job[k][0] = HiddenJob{hg.nn[k], nn_tiles, hg.nn_tiles_x[k], hg.fast[k], ps[k].G};
        hj.job[k][1] = HiddenJob{hg.tn[k], hg.tn_tiles[k], hg.tn_tiles_x[k], hg.fast[k], ps[k].G};
        most = std::max(most, std::max(hg.tn_tiles[k], nn_tiles) * ps[k].G);
    }
    return most;
}

int rrl_mlp_hidden_backward_multi(int n, const rrl_hidden_bwd_t* ps, void* stream) {
    HiddenGroup hg;
    const int rc = build_hidden_group(n, ps, hg);
    if (rc != RRL_OK) return rc;
    if (wants_fold(n, ps)) return RRL_ERANGE;          // one tile per workgroup: nothing to fold with
    HiddenJobs hj;
    const int most = hidden_jobs(n, ps, hg, hj);
    hipLaunchKernelGGL(gemm16_group_kernel, dim3(most, n, 2), dim3(64), 0, (hipStream_t)stream, hj);
    return check_launch();
}

// Packed launches are throughput-bound from a few seeds on (more tiles / workgroups than the chip holds at once), where the
// solo kernels' shapes -- chosen for the latency of ONE seed -- are not the best ones.  Seeds from which the packed launch
// switches shape (per output element the arithmetic is the same either way; the constants below are the measured choices):
//   hidden-layer backward: 64-wide K panels from 2 seeds on, 32-wide from 3 (10 / 5 KB of LDS per tile instead of 20: 16 / 32
//                          tiles in flight per CU instead of 8; one seed has ~6 tiles per CU and launch, four have 24)
//   B <= 1024 forwards   : 2 row tiles per workgroup from 3 seeds on (half the workgroups, each W2 fragment used twice)
// Measured (profiles/packed_ab.sh, ms per packed iteration at 16 updates per step, S = 2 / 3 / 4 / 8): solo shapes
// 3.12 / 3.99 / 4.06 / 5.77, these 3.02 / 3.66 / 3.72 / 5.13; each alone and the other panel widths in profiles/README.md.
static int pack_panel(int S) { return S >= 3 ? 32 : (S >= 2 ? 64 : kPanel); }

// Block form of the packed hidden-layer backward (gemm_block_pack_kernel<1, 2>: 32 x 64 blocks) from 3 seeds on; 64 x 64 and
// 32 x 32 blocks measured as well (profiles/patches/README.md)
static int pack_block(int S) { return S >= 3 ? 12 : 0; }
// seeds up to which the packed head + hidden backward is ONE launch in the tile form (as the solo graph), and up to which in
// the block form; beyond, the head launch + the block form of the hidden backward: with many seeds the launches are throughput-
// bound, the separate head launch costs little and the dh2 generation in every block is what shows (measured per packed
// iteration, 16 updates: S = 4 paired 3.04 ms against 3.11, S = 8 4.30 = 4.27, S = 16 7.12 against 6.67:
// profiles/round5_packed/; with one update per iteration the paired block form wins up to 8 seeds: 7 / 8 seeds 0.456-0.461 /
// 0.448-0.454 ms against 0.466-0.485 / 0.459-0.461).  (RRL_PACK_PAIR_MAX_SEEDS / RRL_PACK_PAIR_BLOCK_MAX_SEEDS: A/B runs)
static int env_int(const char* name, int fallback) {
    const char* e = getenv(name);
    return e ? atoi(e) : fallback;
}
static int pack_pair_max_seeds() {
    static const int v = env_int("RRL_PACK_PAIR_MAX_SEEDS", 2);
    return v;
}
static int pack_pair_block_max_seeds() {
    static const int v = env_int("RRL_PACK_PAIR_BLOCK_MAX_SEEDS", 8);
    return v;
}
// tile counts of a HiddenGroup -> block counts; false: some member has no whole number of full, aligned blocks
static bool hidden_blocks(int n, const rrl_hidden_bwd_t* ps, HiddenGroup& hg, int wm, int wn) {
    const int bm = 32 * wm, bn = 32 * wn;
    for (int k = 0; k < n; ++k) {
        const rrl_hidden_bwd_t& p = ps[k];
        if (!hg.fast[k] || p.H % bm || p.H % bn || p.B % bm || p.H % kBlkPanel || p.B % kBlkPanel) return false;
        const int tx = p.H / bn;
        hg.tn_tiles_x[k] = tx;
        hg.tn_tiles[k] = p.dW2 ? tx * (p.H / bm) : 0;
        hg.nn_tiles_x[k] = tx;
        hg.per_head[k] = hg.tn_tiles[k] + tx * (p.B / bm);
        hg.first[k + 1] = hg.first[k] + hg.per_head[k] * p.G;
    }
    for (int k = n; k < kMaxGroup; ++k) hg.first[k + 1] = hg.first[n];
    return true;
}

// plan of a packed hidden-layer backward: per seed the jobs (tile form, or block form from pack_block(S) seeds on), the 2-D
// placement; i1 = kernel shape (-12: 32 x 64 blocks; 128 / 64 / 32: K-panel width of the tile form), i0 = members
static int hidden_pack_plan(const rrl_pack::Key& key, int S, const int* n, const rrl_hidden_bwd_t* const* members, hipStream_t st,
                            rrl_pack::Plan*& plan) {
    std::vector<HiddenJobs> jobs(S);
    int most[rrl_pack::kMaxSeeds], members_most = 1;
    int shape = pack_block(S);
    for (int pass = 0; pass < 2; ++pass) {        // block form for every seed, or (second pass) the tile form for all
        bool ok = true;
        for (int s = 0; s < S && ok; ++s) {
            HiddenGroup hg;
            const int rc = build_hidden_group(n[s], members[s], hg);
            if (rc != RRL_OK) return rc;
            if (shape) ok = hidden_blocks(n[s], members[s], hg, shape / 10, shape % 10);
            if (ok) most[s] = hidden_jobs(n[s], members[s], hg, jobs[s]);
            members_most = std::max(members_most, n[s]);
        }
        if (ok) break;
        shape = 0;                                 // some member has no whole number of full, aligned blocks
    }
    if (shape != 12)                               // the tile form, one tile per workgroup, cannot fold dx partials
        for (int s = 0; s < S; ++s)
            if (wants_fold(n[s], members[s])) return RRL_ERANGE;
    plan = rrl_pack::store(key, jobs.data(), sizeof(HiddenJobs) * S, st);
    if (!plan) return rrl_pack::store_error();
    rrl_pack::Idx ix;
    plan->grid = finish_members(ix, S, most);
    plan->ix = ix;
    plan->i0 = members_most;
    plan->i1 = shape ? -shape : pack_panel(S);
    return RRL_OK;
}
static void launch_hidden_pack(const rrl_pack::Plan* plan, hipStream_t st) {
    const dim3 grid(plan->grid, plan->i0, 2);
    const HiddenJobs* dev = (const HiddenJobs*)plan->dev;
    if (plan->i1 == -12) hipLaunchKernelGGL((gemm_block_pack_kernel<1, 2>), grid, dim3(256), 0, st, dev, plan->ix);
    else if (plan->i1 == 64) hipLaunchKernelGGL(gemm16_pack_kernel<64>, grid, dim3(64), 0, st, dev, plan->ix);
    else if (plan->i1 == 32) hipLaunchKernelGGL(gemm16_pack_kernel<32>, grid, dim3(64), 0, st, dev, plan->ix);
    else hipLaunchKernelGGL(gemm16_pack_kernel<kPanel>, grid, dim3(64), 0, st, dev, plan->ix);
}

int rrl_mlp_hidden_backward_multi_packed(int S, const int* n, const rrl_hidden_bwd_t* const* members, void* stream) {
    rrl_pack::Key key;
    if (!pack_key(1, S, n, members, key)) return RRL_EINVAL;
    // one seed: the packed launch IS the solo launch (argument block in the kernel arguments, no plan)
    if (S == 1) return rrl_mlp_hidden_backward_multi(n[0], members[0], stream);
    hipStream_t st = (hipStream_t)stream;
    rrl_pack::Plan* plan = rrl_pack::lookup(key);
    if (!plan) {
        const int rc = hidden_pack_plan(key, S, n, members, st, plan);
        if (rc != RRL_OK) return rc;
    }
    launch_hidden_pack(plan, st);
    return check_launch();
}

int rrl_mlp_head_backward(int G, int B, int H, int dout, const float* dOut, const float* h2, const float* W3,
                          float* dW3, float* db3, float* dh2, void* stream) {
    if (!dOut || !h2 || !W3 || !dh2) return RRL_EINVAL;
    if (G <= 0 || B <= 0 || B > 1024 || H <= 0 || dout <= 0 || dout > 4) return RRL_ERANGE;
    HeadBwdArgs hb{};
    hb.la.kind = kPlainDOut;
    hb.la.out = dOut;
    hb.B = B; hb.H = H; hb.dout = dout; hb.need_w = dW3 != nullptr && db3 != nullptr;
    hb.h2 = h2; hb.W3 = W3; hb.dW3 = dW3; hb.db3 = db3; hb.dh2 = dh2;
    hipLaunchKernelGGL((head_bwd_loss_kernel<kPlainDOut>), dim3((H + kCols - 1) / kCols, G), dim3(256), 0,
                       (hipStream_t)stream, hb);
    return check_launch();
}

static int head_loss_args(const rrl_loss_t* la, int G, int B, int H, int dout, const float* h2, const float* W3,
                          float* dW3, float* db3, float* dh2, HeadBwdArgs& hb) {
    if (!la || !la->out || !h2 || !W3) return RRL_EINVAL;          // dh2 == NULL: weight gradients and loss scalars only
    if (G <= 0 || B <= 0 || B > 1024 || H <= 0 || dout <= 0 || dout > 4) return RRL_ERANGE;
    if (la->kind != kPlainDOut) {
        if (la->kind < 0 || la->kind > RRL_LOSS_STOCH_HEAD || la->n_part <= 0 || la->n_part > 4) return RRL_ERANGE;
        const int heads = la->kind <= RRL_LOSS_QRISK_POLICY ? 2 : 1;
        const int width = la->kind <= RRL_LOSS_QRISK_POLICY ? 1 : (la->kind == RRL_LOSS_GAUSS_HEAD ? 4 : 2);
        if (G != heads || dout != width) return RRL_EINVAL;
        if (la->da_parts < 0 || la->da_parts > 16) return RRL_ERANGE;
        if ((la->da_group != 0 && la->da_group != 1 && la->da_group != 4) || (la->da_group == 4 && (la->da_parts & 3)))
            return RRL_ERANGE;
    }
    hb.la = *la;
    hb.B = B; hb.H = H; hb.dout = dout; hb.need_w = dW3 != nullptr && db3 != nullptr;
    hb.h2 = h2; hb.W3 = W3; hb.dW3 = dW3; hb.db3 = db3; hb.dh2 = dh2;
    return RRL_OK;
}

int rrl_mlp_head_backward_loss(const rrl_loss_t* la, int G, int B, int H, int dout, const float* h2,
                               const float* W3, float* dW3, float* db3, float* dh2, void* stream) {
    HeadBwdArgs hb{};
    if (la && la->kind == kPlainDOut) return RRL_EINVAL;
    const int rc = head_loss_args(la, G, B, H, dout, h2, W3, dW3, db3, dh2, hb);
    if (rc != RRL_OK) return rc;
    const dim3 grid((H + kCols - 1) / kCols, G), block(256);
    hipStream_t st = (hipStream_t)stream;
#define RRL_LAUNCH_LOSS(K)                                                            \
    case K:                                                                           \
        hipLaunchKernelGGL((head_bwd_loss_kernel<K>), grid, block, 0, st, hb);        \
        break;
    switch (la->kind) {
        RRL_LAUNCH_LOSS(RRL_LOSS_SAC_CRITIC)
        RRL_LAUNCH_LOSS(RRL_LOSS_SAC_POLICY)
        RRL_LAUNCH_LOSS(RRL_LOSS_QRISK_CRITIC)
        RRL_LAUNCH_LOSS(RRL_LOSS_QRISK_POLICY)
        RRL_LAUNCH_LOSS(RRL_LOSS_GAUSS_HEAD)
        RRL_LAUNCH_LOSS(RRL_LOSS_STOCH_HEAD)
        default:
            return RRL_EINVAL;
    }
#undef RRL_LAUNCH_LOSS
    return check_launch();
}

static int build_head_group(int n, const rrl_head_bwd_t* ps, HeadBwdGroup& hg) {
    if (!ps || n <= 0 || n > kMaxGroup) return RRL_EINVAL;
    hg = HeadBwdGroup{};
    hg.n = n;
    hg.first[0] = 0;
    for (int k = 0; k < n; ++k) {
        const rrl_head_bwd_t& p = ps[k];
        const int rc = head_loss_args(&p.loss, p.G, p.B, p.H, p.dout, p.h2, p.W3, p.dW3, p.db3, p.dh2, hg.p[k]);
        if (rc != RRL_OK) return rc;
        hg.G[k] = p.G;
        hg.blocks_x[k] = (p.H + kCols - 1) / kCols;
        hg.first[k + 1] = hg.first[k] + hg.blocks_x[k] * p.G;
    }
    for (int k = n; k < kMaxGroup; ++k) hg.first[k + 1] = hg.first[n];
    return RRL_OK;
}

// one member with a loss description: its own kernel (head_bwd_loss_kernel<kind>: the same body without the other kinds' code
// and the group's argument blocks around it); several members, or a plain dOut tensor: the group kernel
static void launch_head_group(const HeadBwdGroup& hg, int n, hipStream_t st) {
    if (n == 1) {
        const HeadBwdArgs& hb = hg.p[0];
        const dim3 grid(hg.blocks_x[0], hg.G[0]), block(256);
        switch (hb.la.kind) {
            case RRL_LOSS_SAC_CRITIC: hipLaunchKernelGGL((head_bwd_loss_kernel<RRL_LOSS_SAC_CRITIC>), grid, block, 0, st, hb); return;
            case RRL_LOSS_SAC_POLICY: hipLaunchKernelGGL((head_bwd_loss_kernel<RRL_LOSS_SAC_POLICY>), grid, block, 0, st, hb); return;
            case RRL_LOSS_QRISK_CRITIC: hipLaunchKernelGGL((head_bwd_loss_kernel<RRL_LOSS_QRISK_CRITIC>), grid, block, 0, st, hb); return;
            case RRL_LOSS_QRISK_POLICY: hipLaunchKernelGGL((head_bwd_loss_kernel<RRL_LOSS_QRISK_POLICY>), grid, block, 0, st, hb); return;
            case RRL_LOSS_GAUSS_HEAD: hipLaunchKernelGGL((head_bwd_loss_kernel<RRL_LOSS_GAUSS_HEAD>), grid, block, 0, st, hb); return;
            case RRL_LOSS_STOCH_HEAD: hipLaunchKernelGGL((head_bwd_loss_kernel<RRL_LOSS_STOCH_HEAD>), grid, block, 0, st, hb); return;
            default: break;
        }
    }
    hipLaunchKernelGGL(head_bwd_group_kernel, dim3(largest_member(hg, n), n), dim3(256), 0, st, hg);
}

int rrl_mlp_head_backward_multi(int n, const rrl_head_bwd_t* ps, void* stream) {
    HeadBwdGroup hg;
    const int rc = build_head_group(n, ps, hg);
    if (rc != RRL_OK) return rc;
    launch_head_group(hg, n, (hipStream_t)stream);
    return check_launch();
}

// head backward + hidden backward of n stacks: ONE launch (backward_pair_kernel) when every member is a critic-loss kind
// with one output, full aligned tiles and dh2 as the only link between its two stages; otherwise the two launches of
// rrl_mlp_head_backward_multi + rrl_mlp_hidden_backward_multi.  Same results either way.
// PairJobs of n stacks (backward_pair_kernel); pair = false: some member does not qualify for the paired form (pj is then
// unspecified).  most = workgroups of the largest job (grid.x)
static bool hidden_blocks(int n, const rrl_hidden_bwd_t* ps, HiddenGroup& hg, int wm, int wn);
// blocks = true: the jobs' tile counts in units of 32 x 64 blocks (backward_pair_block_pack_kernel)
static int build_pair_jobs(int n, const rrl_head_bwd_t* heads, const rrl_hidden_bwd_t* hidden, PairJobs& pj, int& most, bool& pair,
                           int& dout, HeadBwdGroup* hg_out = nullptr, HiddenGroup* hd_out = nullptr, bool blocks = false) {
    HeadBwdGroup hg;
    int rc = build_head_group(n, heads, hg);
    if (rc != RRL_OK) return rc;
    HiddenGroup hd;
    rc = build_hidden_group(n, hidden, hd);
    if (rc != RRL_OK) return rc;
    if (hg_out) *hg_out = hg;
    if (hd_out) *hd_out = hd;
    pair = true;
    dout = 0;
    for (int k = 0; k < n && pair; ++k) {
        const rrl_head_bwd_t& h = heads[k];
        const rrl_hidden_bwd_t& d = hidden[k];
        const int kind = h.loss.kind;
        const int nn_tiles = hd.per_head[k] - hd.tn_tiles[k];
        // every member of one launch has the same number of outputs (the kernel is compiled per DOUT): the critic-loss kinds
        // (1), the tanh-Gaussian head (4) or the stochastic head (2)
        const int my = (kind >= RRL_LOSS_SAC_CRITIC && kind <= RRL_LOSS_QRISK_POLICY) ? 1
                       : kind == RRL_LOSS_GAUSS_HEAD ? 4 : kind == RRL_LOSS_STOCH_HEAD ? 2 : 0;
        pair = my != 0 && (dout == 0 || dout == my) && h.dout == my && h.dh2 && h.dh2 == d.dh2 &&
               h.G == d.G && h.B == d.B && h.H == d.H && h.B * my <= kPairDsh && h.H <= 256 && hd.fast[k] &&
               (h.H % kPairPanel) == 0 && (h.B % kPairPanel) == 0 && (nn_tiles % 4) == 0 && (hd.tn_tiles[k] % 4) == 0 &&
               (reinterpret_cast<uintptr_t>(h.h2) & 15) == 0 && (reinterpret_cast<uintptr_t>(h.W3) & 15) == 0;
        dout = my;
    }
    if (pair && blocks) pair = hidden_blocks(n, hidden, hd, 1, 2);          // whole 32 x 64 blocks, or no block form
    if (!pair) return RRL_OK;
    pj = PairJobs{};
    most = 1;
    const int per_wg = blocks ? 1 : 4;             // tile form: four tiles per workgroup; block form: counts are workgroups
    for (int k = 0; k < n; ++k) {
        const int nn_tiles = hd.per_head[k] - hd.tn_tiles[k];
        pj.job[k][0] = HiddenJob{hd.nn[k], nn_tiles, hd.nn_tiles_x[k], 1, hidden[k].G};
        pj.job[k][1] = HiddenJob{hd.tn[k], hd.tn_tiles[k], hd.tn_tiles_x[k], 1, hidden[k].G};
        pj.job[k][0].ga.A = pj.job[k][1].ga.A = heads[k].h2;          // dh2 is derived from h2 in the tiles
        pj.head[k] = hg.p[k];
        pj.head[k].dh2 = nullptr;
        pj.blocks_x[k] = hg.blocks_x[k];
        most = std::max(most, std::max(std::max(hd.tn_tiles[k], nn_tiles) * hidden[k].G / per_wg, hg.blocks_x[k] * heads[k].G));
    }
    return RRL_OK;
}

// head backward + hidden backward of n stacks: ONE launch (backward_pair_kernel) when every member is a critic-loss kind
// with one output, full aligned tiles and dh2 as the only link between its two stages; otherwise the two launches of
// rrl_mlp_head_backward_multi + rrl_mlp_hidden_backward_multi.  Same results either way.
int rrl_mlp_backward_pair_multi(int n, const rrl_head_bwd_t* heads, const rrl_hidden_bwd_t* hidden, void* stream) {
    PairJobs pj;
    HeadBwdGroup hg;
    HiddenGroup hd;
    int most = 1, dout = 0;
    bool pair = false;
    const int rc = build_pair_jobs(n, heads, hidden, pj, most, pair, dout, &hg, &hd);
    if (rc != RRL_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (!pair) {
        if (wants_fold(n, hidden)) return RRL_ERANGE;
        launch_head_group(hg, n, st);
        HiddenJobs hj;
        const int m = hidden_jobs(n, hidden, hd, hj);
        hipLaunchKernelGGL(gemm16_group_kernel, dim3(m, n, 2), dim3(64), 0, st, hj);
        return check_launch();
    }
    if (dout == 4) hipLaunchKernelGGL(backward_pair_kernel<4>, dim3(most, n, 3), dim3(256), 0, st, pj);
    else if (dout == 2) hipLaunchKernelGGL(backward_pair_kernel<2>, dim3(most, n, 3), dim3(256), 0, st, pj);
    else hipLaunchKernelGGL(backward_pair_kernel<1>, dim3(most, n, 3), dim3(256), 0, st, pj);
    return check_launch();
}

static int head_pack_plan(const rrl_pack::Key& key, int S, const int* n, const rrl_head_bwd_t* const* members, hipStream_t st,
                          rrl_pack::Plan*& plan) {
    std::vector<HeadBwdGroup> groups(S);
    int most[rrl_pack::kMaxSeeds], members_most = 1;
    for (int s = 0; s < S; ++s) {
        const int rc = build_head_group(n[s], members[s], groups[s]);
        if (rc != RRL_OK) return rc;
        most[s] = largest_member(groups[s], n[s]);
        members_most = std::max(members_most, n[s]);
    }
    plan = rrl_pack::store(key, groups.data(), sizeof(HeadBwdGroup) * S, st);
    if (!plan) return rrl_pack::store_error();
    rrl_pack::Idx ix;
    plan->grid = finish_members(ix, S, most);
    plan->ix = ix;
    plan->i0 = members_most;
    return RRL_OK;
}

int rrl_mlp_head_backward_multi_packed(int S, const int* n, const rrl_head_bwd_t* const* members, void* stream) {
    rrl_pack::Key key;
    if (!pack_key(3, S, n, members, key)) return RRL_EINVAL;
    // one seed: the packed launch IS the solo launch (argument block in the kernel arguments, no plan)
    if (S == 1) return rrl_mlp_head_backward_multi(n[0], members[0], stream);
    hipStream_t st = (hipStream_t)stream;
    rrl_pack::Plan* plan = rrl_pack::lookup(key);
    if (!plan) {
        const int rc = head_pack_plan(key, S, n, members, st, plan);
        if (rc != RRL_OK) return rc;
    }
    hipLaunchKernelGGL(head_bwd_pack_kernel, dim3(plan->grid, plan->i0), dim3(256), 0, st, (const HeadBwdGroup*)plan->dev, plan->ix);
    return check_launch();
}

// rrl_mlp_backward_pair_multi for S seeds: ONE launch (backward_pair_pack_kernel) when every member of every seed qualifies
// for the paired form, the two packed launches otherwise -- the same decision, per seed the same results as the solo entry.
int rrl_mlp_backward_pair_multi_packed(int S, const int* n, const rrl_head_bwd_t* const* heads,
                                       const rrl_hidden_bwd_t* const* hidden, void* stream) {
    rrl_pack::Key key, key_head, key_hidden;
    if (!pack_key(3, S, n, heads, key_head) || !pack_key(1, S, n, hidden, key_hidden)) return RRL_EINVAL;
    // one seed: the packed launch IS the solo launch (argument block in the kernel arguments, no plan)
    if (S == 1) return rrl_mlp_backward_pair_multi(n[0], heads[0], hidden[0], stream);
    key.pod(7);
    key.add(key_head.bytes.data(), key_head.bytes.size());
    key.add(key_hidden.bytes.data(), key_hidden.bytes.size());
    hipStream_t st = (hipStream_t)stream;
    rrl_pack::Plan* plan = rrl_pack::lookup(key);
    if (!plan) {
        std::vector<PairJobs> jobs(S);
        int most[rrl_pack::kMaxSeeds], members_most = 1;
        // up to pack_pair_max_seeds() seeds the paired launch keeps the solo kernel's latency shape (one wave per 16 x 16 tile,
        // four tiles per workgroup); beyond, the block form (32 x 64 blocks per four-wave workgroup: measured, S = 4, the
        // tile-form pair 23.6 us against 8.3 + 9.2 us for the head launch + the block form: profiles/round5_packed/)
        const bool blocks = S > pack_pair_max_seeds();
        bool pair = S <= pack_pair_block_max_seeds();
        int dout = 0;
        for (int s = 0; s < S && pair; ++s) {
            int m = 1, my = 0;
            const int rc = build_pair_jobs(n[s], heads[s], hidden[s], jobs[s], m, pair, my, nullptr, nullptr, blocks);
            if (rc != RRL_OK) return rc;
            pair = pair && (dout == 0 || dout == my);
            dout = my;
            most[s] = m;
            members_most = std::max(members_most, n[s]);
        }
        if (pair) {
            plan = rrl_pack::store(key, jobs.data(), sizeof(PairJobs) * S, st);
            if (!plan) return rrl_pack::store_error();
            rrl_pack::Idx ix;
            plan->grid = finish_members(ix, S, most);
            plan->ix = ix;
            plan->i0 = members_most;
            plan->i1 = dout + (blocks ? 16 : 0);
        } else {
            // not pairable: remember that (an empty plan: no device copy needed) and issue the two packed launches
            PairJobs none{};
            plan = rrl_pack::store(key, &none, sizeof none, st);
            if (!plan) return rrl_pack::store_error();
            plan->i1 = 0;
        }
    }
    if (plan->i1 == 0) {
        const int rc = rrl_mlp_head_backward_multi_packed(S, n, heads, stream);
        return rc != RRL_OK ? rc : rrl_mlp_hidden_backward_multi_packed(S, n, hidden, stream);
    }
    const dim3 grid(plan->grid, plan->i0, 3);
    const PairJobs* dev = (const PairJobs*)plan->dev;
    if (plan->i1 == 16 + 4) hipLaunchKernelGGL(backward_pair_block_pack_kernel<4>, grid, dim3(256), 0, st, dev, plan->ix);
    else if (plan->i1 == 16 + 2) hipLaunchKernelGGL(backward_pair_block_pack_kernel<2>, grid, dim3(256), 0, st, dev, plan->ix);
    else if (plan->i1 == 16 + 1) hipLaunchKernelGGL(backward_pair_block_pack_kernel<1>, grid, dim3(256), 0, st, dev, plan->ix);
    else if (plan->i1 == 4) hipLaunchKernelGGL(backward_pair_pack_kernel<4>, grid, dim3(256), 0, st, dev, plan->ix);
    else if (plan->i1 == 2) hipLaunchKernelGGL(backward_pair_pack_kernel<2>, grid, dim3(256), 0, st, dev, plan->ix);
    else hipLaunchKernelGGL(backward_pair_pack_kernel<1>, grid, dim3(256), 0, st, dev, plan->ix);
    return check_launch();
}

static int input_args(int G, int B, int H, int din, const float* dh1, const float* x, int ldx, const float* W1,
                      float* dW1, float* db1, float* dx, InputBwdArgs& ib, int& blocks) {
    if (!dh1 || !x || !W1) return RRL_EINVAL;
    if (G <= 0 || B <= 0 || H <= 0 || din <= 0 || din > 4) return RRL_ERANGE;
    const int need_w = dW1 != nullptr && db1 != nullptr, need_x = dx != nullptr;
    ib = InputBwdArgs{B, H, din, ldx, need_w, need_x, dh1, x, W1, dW1, db1, dx};
    blocks = (need_w ? (H + kCols - 1) / kCols : 0) + (need_x ? (B + 3) / 4 : 0);
    return RRL_OK;
}

int rrl_mlp_input_backward(int G, int B, int H, int din, const float* dh1, const float* x, int ldx,
                           const float* W1, float* dW1, float* db1, float* dx, void* stream) {
    InputBwdArgs ib;
    int blocks;
    const int rc = input_args(G, B, H, din, dh1, x, ldx, W1, dW1, db1, dx, ib, blocks);
    if (rc != RRL_OK) return rc;
    if (blocks == 0) return RRL_OK;
    hipLaunchKernelGGL(input_bwd_kernel, dim3(blocks, G), dim3(256), 0, (hipStream_t)stream, ib);
    return check_launch();
}

int rrl_mlp_input_backward_multi(int n, const rrl_input_bwd_t* ps, void* stream) {
    if (!ps || n <= 0 || n > kMaxGroup) return RRL_EINVAL;
    InputBwdGroup ig{};
    ig.first[0] = 0;
    int m = 0;
    for (int k = 0; k < n; ++k) {
        const rrl_input_bwd_t& p = ps[k];
        int blocks;
        const int rc = input_args(p.G, p.B, p.H, p.din, p.dh1, p.x, p.ldx, p.W1, p.dW1, p.db1, p.dx, ig.p[m], blocks);
        if (rc != RRL_OK) return rc;
        if (blocks == 0) continue;
        ig.G[m] = p.G;
        ig.blocks_x[m] = blocks;
        ig.first[m + 1] = ig.first[m] + blocks * p.G;
        ++m;
    }
    if (m == 0) return RRL_OK;
    ig.n = m;
    for (int k = m; k < kMaxGroup; ++k) ig.first[k + 1] = ig.first[m];
    hipLaunchKernelGGL(input_bwd_group_kernel, dim3(ig.first[m]), dim3(256), 0, (hipStream_t)stream, ig);
    return check_launch();
}

}  // extern "C"
