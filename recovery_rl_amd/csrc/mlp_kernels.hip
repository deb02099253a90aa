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

#include <cstdlib>

#include "rrl_device.hpp"
#include "pack.hpp"
#include "rrl_host.hpp"

namespace {

using rrl_host::check_launch;

typedef float f32x4 __attribute__((ext_vector_type(4)));

// -DRRL_NT_STORES (profiles/nt_store_probe.py builds a second library with it): the intermediates one stage hands to the
// next (saved activations, dh2, weight gradients) are stored with the non-temporal hint
#ifdef RRL_NT_STORES
#define RRL_HANDOVER_STORE(ptr, v) __builtin_nontemporal_store((v), (ptr))
#else
#define RRL_HANDOVER_STORE(ptr, v) (*(ptr) = (v))
#endif

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
    // GEN tiles (hidden_head_group_kernel): the left operand dh2 is not read but generated from what it is made of,
    //   dh2[b][j] = [h2[b][j] > 0] * dOut[b] * W3[j]        (one-output heads: the critic-type losses)
    // with dOut[b] in LDS (evaluated per workgroup from the loss description) -- the head-backward launch that used to write
    // dh2 (and its round trip through memory) is gone; same products, same bits.
    const float* gen_h2;  // [G, B, H], the layout of dh2
    const float* gen_w3;  // [G, H]
};

// pointers of an argument block that was copied out of device memory (packed launches): see rrl_pack::to_global
__device__ __forceinline__ void globalize(GemmArgs& a) {
    rrl_pack::to_global_all(a.A, a.B, a.C, a.bias, a.mask, a.colsum, a.x, a.W1, a.first_part, a.dx_part, a.gen_h2, a.gen_w3);
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

// ---- whole-line loads of a k-contiguous operand, restaged into fragment order by lane permutation (opt-in) ------------
// The fragment-order load below has every 16-lane pass of a global_load_dwordx4 touch 16 different 128-byte lines and use 16
// bytes of each.  The whole-line form: register 2 p + c of lane l = row 8 c + (l >> 3), floats 32 p + 4 (l & 7) .. + 3 (8
// lanes per line, 2 lines per pass); `restage_panels` then moves the values to where the MFMAs expect them: fragment 2 p + jj
// of lane (i, q) = row i, floats 32 p + 16 jj + 4 q .. + 3 = register (i >> 3) of lane 8 (i & 7) + 4 jj + q.  A ds_bpermute
// moves ONE register per source lane and both rows 8 c + r live in the same source lanes, so pass A serves fragment 0 of the
// rows below 8 and fragment 1 of the rows from 8 on (source lanes with l & 4 == 0 send register 0, the others register 1:
// every source lane is asked exactly once), pass B the two other quarters, and the receiving lane sorts A / B into fragment
// 0 / 1 by its own row.  Eight permutes and sixteen selects per 32-float panel, no LDS memory; the same values land in the
// same registers as with fragment-order loads, so nothing downstream changes (lane arithmetic emulated in
// tests/test_w2_permute_cpu.py, hardware check tests/test_w2_permute_gpu.py).
template <int NP>
__device__ __forceinline__ void restage_panels(float4* v, int lane) {
    const int i = lane & 15, q = lane >> 4;
    const bool low_src = (lane & 4) == 0, low_row = (i & 8) == 0;
    const int addr_a = 4 * (8 * (i & 7) + (low_row ? 0 : 4) + q);
    const int addr_b = 4 * (8 * (i & 7) + (low_row ? 4 : 0) + q);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const float r0[4] = {v[2 * p].x, v[2 * p].y, v[2 * p].z, v[2 * p].w};
        const float r1[4] = {v[2 * p + 1].x, v[2 * p + 1].y, v[2 * p + 1].z, v[2 * p + 1].w};
        float f0[4], f1[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int send_a = __float_as_int(low_src ? r0[c] : r1[c]);
            const int send_b = __float_as_int(low_src ? r1[c] : r0[c]);
            const float got_a = __int_as_float(__builtin_amdgcn_ds_bpermute(addr_a, send_a));
            const float got_b = __int_as_float(__builtin_amdgcn_ds_bpermute(addr_b, send_b));
            f0[c] = low_row ? got_a : got_b;
            f1[c] = low_row ? got_b : got_a;
        }
        v[2 * p] = make_float4(f0[0], f0[1], f0[2], f0[3]);
        v[2 * p + 1] = make_float4(f1[0], f1[1], f1[2], f1[3]);
        __builtin_amdgcn_sched_barrier(0);      // one panel's permutes in flight at a time: eight temporaries, not 8 NP
    }
}
#ifndef RRL_COALESCE_DIRECT
#define RRL_COALESCE_DIRECT 0     /* opt-in: the k-contiguous operands of the 16 x 16 GEMM tiles (hidden-layer backward: dh2) */
#endif
template <bool FAST, int VEC>
constexpr bool direct_coalesced() { return RRL_COALESCE_DIRECT != 0 && FAST && VEC % 2 == 0; }

// The K order inside a panel is permuted (the sum over k does not care): MFMA step s = 4 j + t of
// lane group q = lane >> 4 consumes k = 16 j + 4 q + t.  An operand whose k index is contiguous in
// memory (rows = M or N index) is then exactly element t of the lane's j-th float4 of its own row
// i = lane & 15 -- it feeds the MFMA straight from registers, no LDS.
template <bool FAST, int VEC>
__device__ __forceinline__ void load_direct(FragT<VEC>& f, const float* __restrict__ src, int ld, int row0,
                                            int rows, int k0, int K, int lane) {
    if constexpr (direct_coalesced<FAST, VEC>()) {      // FAST: every tile full, K a multiple of the panel
        const float* base = src + (long long)(row0 + (lane >> 3)) * ld + k0 + 4 * (lane & 7);
#pragma unroll
        for (int pj = 0; pj < VEC; ++pj)
            f.v[pj] = *reinterpret_cast<const float4*>(base + (long long)(8 * (pj & 1)) * ld + 32 * (pj >> 1));
        return;
    }
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

struct NoPrologue {
    __device__ __forceinline__ void operator()() const {}
};

// GEN: `prologue` fills dsh (LDS) and runs AFTER the first panel's loads have been issued, so its memory round trip and
// theirs overlap
// One wave per tile: with WL the tile's LDS region belongs to its wave alone (several tiles per workgroup, hidden_head_*
// kernels), so "barrier" = this wave's LDS operations have completed; otherwise the workgroup is the wave (64 threads) and
// __syncthreads is the same thing
template <bool WL>
__device__ __forceinline__ void tile_sync() {
    if constexpr (WL) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    } else {
        __syncthreads();
    }
}
// GEN tiles keep dOut[r], r < 512, in the pad columns 16..19 of the As tile (rows of kLd = 20 floats, 16 used)
__device__ __forceinline__ int dsh_index(int r) { return (r >> 2) * kLd + (r & 3); }

// PANEL = K elements per panel (a multiple of 16).  The K chunks are consumed in the same ascending order with the same
// accumulator for every PANEL, so the result does not depend on it; a smaller panel = smaller LDS tiles = more tiles in
// flight per CU (the packed launches, which have more tiles than LDS for them) at the price of a wave-level sync per panel.
template <int MODE, bool FAST, bool GEN = false, class Prologue = NoPrologue, bool WL = false, int PANEL = kPanel>  // MODE: 0 NT, 1 NN, 2 TN
__device__ __forceinline__ void gemm16_tile(const GemmArgs& a, float* As, float* Bs, int bx, int by, int g,
                                            const float* dsh = nullptr, Prologue prologue = Prologue()) {
    static_assert(!GEN || (FAST && MODE != 0), "generated operands: full aligned NN / TN tiles");
    const int lane = threadIdx.x & 63;
    const int m0 = by * kTile, n0 = bx * kTile;
    const float* A = GEN ? a.gen_h2 + g * a.sA : a.A + g * a.sA;
    // GEN: W3 of this head; TN tiles use 4 fixed columns of it, NN tiles the k positions of every panel
    const float* W3 = GEN ? a.gen_w3 + (long long)g * (MODE == 2 ? a.M : a.K) : nullptr;
    float4 w3c = make_float4(0.f, 0.f, 0.f, 0.f);
    float drow = 0.f;
    if constexpr (GEN && MODE == 2) w3c = *reinterpret_cast<const float4*>(W3 + m0 + (lane & 3) * 4);
    constexpr int VEC = PANEL / 16;
    using Frag = FragT<VEC>;
    Frag fw;
    const float* B = a.B + g * a.sB;
    float* C = a.C + g * a.sC;
    constexpr bool kStageA = MODE == 2, kStageB = MODE != 0;

    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    float asum = 0.f;  // TN: running sum of my A operands (for colsum)
    Frag fa, fb;

    auto load = [&](int k0) {
        if (kStageA) load_staged<FAST>(fa, A, a.lda, m0, a.M, k0, a.K, lane);
        else load_direct<FAST>(fa, A, a.lda, m0, a.M, k0, a.K, lane);
        if (kStageB) load_staged<FAST>(fb, B, a.ldb, n0, a.N, k0, a.K, lane);
        else load_direct<FAST>(fb, B, a.ldb, n0, a.N, k0, a.K, lane);
        if constexpr (GEN && MODE == 1) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) fw.v[j] = *reinterpret_cast<const float4*>(W3 + k0 + 16 * j + 4 * (lane >> 4));
        }
    };
    auto gen4 = [](float4 h, float d, float4 w) {       // the head backward's  a > 0 ? fmaf(dOut, W3, 0) : 0
        return make_float4(h.x > 0.f ? d * w.x : 0.f, h.y > 0.f ? d * w.y : 0.f, h.z > 0.f ? d * w.z : 0.f,
                           h.w > 0.f ? d * w.w : 0.f);
    };


    const int np = (a.K + PANEL - 1) / PANEL;
    const int i = lane & 15, q = lane >> 4;
    load(0);
    if constexpr (GEN) {
        prologue();
        tile_sync<WL>();
        if constexpr (MODE == 1) drow = dsh[dsh_index(m0 + (lane & 15))];
    }
    for (int p = 0; p < np; ++p) {
        if constexpr (direct_coalesced<FAST, VEC>()) {      // whole-line loads: into fragment order first
            if (!kStageA) restage_panels<VEC / 2>(fa.v, lane);
            if (!kStageB) restage_panels<VEC / 2>(fb.v, lane);
        }
        Frag ca = fa, cb = fb;           // operands of this panel (registers)
        if constexpr (GEN && MODE == 1) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) ca.v[j] = gen4(fa.v[j], drow, fw.v[j]);
        }
        if constexpr (GEN && MODE == 2) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) fa.v[j] = gen4(fa.v[j], dsh[dsh_index(p * PANEL + (lane >> 2) + 16 * j)], w3c);
        }
        if (kStageA || kStageB) {
            if (p) tile_sync<WL>();      // the previous panel's LDS reads are done
            if (kStageA) store_staged(fa, As, lane);
            if (kStageB) store_staged(fb, Bs, lane);
            tile_sync<WL>();
        }
        if (p + 1 < np) load((p + 1) * PANEL);   // next panel's global loads fly under the MFMAs
        const int klen = min(PANEL, a.K - p * PANEL);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            if (16 * j < klen) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int k = 16 * j + 4 * q + t;
                    const float av = kStageA ? As[k * kLd + i] : elem(ca.v[j], t);
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
    const float bias = (MODE == 0 && a.bias && col < a.N) ? a.bias[g * a.sBias + col] : 0.f;
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
                    const float s = a.mask[g * a.sMask + (long long)row * a.ldmask + col];
                    v = s > 0.f ? v : 0.f;
                }
                vout[r] = v;
                if (store_c) {
                    float* dst = C + (long long)row * a.ldc + col;
                    RRL_HANDOVER_STORE(dst, a.accumulate ? (*dst + v) : v);
                }
            }
        }
    }
    if constexpr (MODE == 1) {
        if (a.x) {      // first-layer backward from this tile (FAST geometry: full tiles); a WG is one wavefront
            // the tile and the 16 rows of x / W1 it meets, through LDS (the K loop is done with Bs)
            float* T = Bs;                    // [16][17] tile, then xs [16][4] at 272, ws [16][4] at 336
            const int rr = lane & 15, dd = lane >> 4;
            const float xv = dd < a.din ? a.x[(long long)(m0 + rr) * a.ldx + dd] : 0.f;
            const float wv = dd < a.din ? a.W1[((long long)g * a.N + n0 + rr) * a.din + dd] : 0.f;
            tile_sync<WL>();
#pragma unroll
            for (int r = 0; r < 4; ++r) T[(4 * (lane >> 4) + r) * 17 + (lane & 15)] = vout[r];
            T[272 + rr * 4 + dd] = xv;
            T[336 + rr * 4 + dd] = wv;
            tile_sync<WL>();
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
            if (a.dx_part && dd < a.din)
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
constexpr int kMaxGroup = 4;
struct HiddenGroup {
    GemmArgs tn[kMaxGroup], nn[kMaxGroup];
    int tn_tiles_x[kMaxGroup], tn_tiles[kMaxGroup], nn_tiles_x[kMaxGroup], per_head[kMaxGroup], fast[kMaxGroup];
    int first[kMaxGroup + 1];
    int n;
};

template <int PANEL = kPanel>
__device__ __forceinline__ void gemm16_group_body(const HiddenGroup& hg, int block, float* As, float* Bs);

__global__ __launch_bounds__(64) void gemm16_group_kernel(HiddenGroup hg) {
    __shared__ __attribute__((aligned(16))) float As[kPanel * kLd];
    __shared__ __attribute__((aligned(16))) float Bs[kPanel * kLd];
    gemm16_group_body(hg, blockIdx.x, As, Bs);
}

// PANEL = 128: 20 KB of LDS per single-wave workgroup, i.e. 8 tiles in flight per CU -- enough for one seed (1 000-1 500 tiles
// per launch), three rounds for four seeds.  PANEL = 64 (every member FAST, i.e. K a multiple of 128): 16 tiles per CU.
template <int PANEL>
__global__ __launch_bounds__(64) void gemm16_pack_kernel(const HiddenGroup* __restrict__ groups, rrl_pack::Idx ix) {
    __shared__ __attribute__((aligned(16))) float As[PANEL * kLd];
    __shared__ __attribute__((aligned(16))) float Bs[PANEL * kLd];
    int s, local;
    if (!rrl_pack::locate(ix, blockIdx.x, s, local)) return;
    // the one member this workgroup serves, not the whole 1.8 KB group, is what it copies out of device memory
    gemm16_group_body<PANEL>(groups[s], local, As, Bs);
}

template <int PANEL>
__device__ __forceinline__ void gemm16_group_body(const HiddenGroup& hg, int block, float* As, float* Bs) {
    int k = 0;
    while (k + 1 < hg.n && block >= hg.first[k + 1]) ++k;
    const int local = block - hg.first[k];
    const int g = local / hg.per_head[k], b = local - g * hg.per_head[k];
    // the one problem this workgroup serves is copied out of the group (kernel arguments, or device memory for the packed
    // launch): its fields are then wave-uniform registers whatever the group's home
    if (b < hg.tn_tiles[k]) {
        GemmArgs ga = hg.tn[k];
        globalize(ga);
        if (hg.fast[k]) gemm16_tile<2, true, false, NoPrologue, false, PANEL>(ga, As, Bs, b % hg.tn_tiles_x[k], b / hg.tn_tiles_x[k], g);
        else gemm16_tile<2, false, false, NoPrologue, false, PANEL>(ga, As, Bs, b % hg.tn_tiles_x[k], b / hg.tn_tiles_x[k], g);
    } else {
        const int c = b - hg.tn_tiles[k];
        GemmArgs ga = hg.nn[k];
        globalize(ga);
        if (hg.fast[k]) gemm16_tile<1, true, false, NoPrologue, false, PANEL>(ga, As, Bs, c % hg.nn_tiles_x[k], c / hg.nn_tiles_x[k], g);
        else gemm16_tile<1, false, false, NoPrologue, false, PANEL>(ga, As, Bs, c % hg.nn_tiles_x[k], c / hg.nn_tiles_x[k], g);
    }
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

template <int MODE, int WM, int WN>   // MODE: 1 NN (A direct, B staged), 2 TN (both staged)
__device__ __forceinline__ void gemm_block(const GemmArgs& a, float* lds, int bx, int by, int g) {
    using L = BlkLds<WM, WN>;
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
    auto stage = [&](int buf) __attribute__((always_inline)) {
        if (MODE == 2) blk_store<BM>(ra, As + buf * L::kA, tid);
        blk_store<BN>(rb, Bs + buf * L::kB, tid);
    };

    // panel p: registers -> LDS buffer p & 1 (its last readers, panel p - 2, are behind the barrier of panel p - 1), panel
    // p + 1's global loads issued, ONE barrier, MFMAs -- the loads fly under them
    const int np = a.K / kBlkPanel;
    load(0);
    for (int p = 0; p < np; ++p) {
        FragT<VEC> ca[WM];                    // NN: this panel's rows of A
        if constexpr (MODE != 2 && direct_coalesced<true, VEC>()) {      // whole-line loads: into fragment order first
#pragma unroll
            for (int x = 0; x < WM; ++x) restage_panels<VEC / 2>(fa[x].v, lane);
        }
#pragma unroll
        for (int x = 0; x < WM; ++x) ca[x] = fa[x];
        stage(p & 1);
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
#pragma unroll
        for (int y = 0; y < WN; ++y) {
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
                    RRL_HANDOVER_STORE(dst, a.accumulate ? (*dst + v) : v);
                }
            }
            if constexpr (MODE == 1) {
                if (a.x) {
                    const int rr = lane & 15, dd = lane >> 4;
                    const float xv = dd < a.din ? a.x[(long long)(tm + rr) * a.ldx + dd] : 0.f;
                    const float wv = dd < a.din ? a.W1[((long long)g * a.N + tn + rr) * a.din + dd] : 0.f;
                    tile_sync<true>();
#pragma unroll
                    for (int r = 0; r < 4; ++r) T[(4 * q + r) * 17 + i] = vout[r];
                    T[272 + rr * 4 + dd] = xv;
                    T[336 + rr * 4 + dd] = wv;
                    tile_sync<true>();
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
                    if (a.dx_part && dd < a.din)
                        a.dx_part[(((long long)tbx * a.G + g) * a.M + tm + rr) * a.din + dd] = sx;
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

// HiddenGroup with its tile counts in BLOCK units (build_hidden_blocks)
template <int WM, int WN>
__global__ __launch_bounds__(256) void gemm_block_pack_kernel(const HiddenGroup* __restrict__ groups, rrl_pack::Idx ix) {
    __shared__ __attribute__((aligned(16))) float lds[BlkLds<WM, WN>::kFloats];
    int s, block;
    if (!rrl_pack::locate(ix, blockIdx.x, s, block)) return;
    const HiddenGroup& hg = groups[s];
    int k = 0;
    while (k + 1 < hg.n && block >= hg.first[k + 1]) ++k;
    const int local = block - hg.first[k];
    const int g = local / hg.per_head[k], b = local - g * hg.per_head[k];
    if (b < hg.tn_tiles[k]) {
        GemmArgs ga = hg.tn[k];
        globalize(ga);
        gemm_block<2, WM, WN>(ga, lds, b % hg.tn_tiles_x[k], b / hg.tn_tiles_x[k], g);
    } else {
        const int c = b - hg.tn_tiles[k];
        GemmArgs ga = hg.nn[k];
        globalize(ga);
        gemm_block<1, WM, WN>(ga, lds, c % hg.nn_tiles_x[k], c / hg.nn_tiles_x[k], g);
    }
}

// ---- fused forward of a whole 2-hidden-layer stack -----------------------------------------------
//   out[g] = W3[g] relu(W2[g] relu(W1[g] x + b1[g]) + b2[g]) + b3[g]      x [M, din] shared by the heads
// One workgroup (16 waves) per 16 rows and head: layer 1 on the VALU (din <= 4), layer 2 on MFMA with
// the 16 x H activation tile in LDS shared by all waves (wave w owns output columns 16w..16w+15 and
// streams its 16 rows of W2 straight from L2 into MFMA operands), layer 3 by 16-lane dot products.
// No intermediate activation touches HBM unless the caller asks for h1 / h2 (needed by backward).
namespace loss {

constexpr float kLogSigMax = 2.f, kLogSigMin = -20.f, kEps = 1e-6f;   // model.py:14-16

// value of a stack output given as np <= 4 partial sums ps floats apart: ((p0 + p1) + p2) + p3, the order of the
// stand-alone sum kernel.  All loads are issued together (a run-time loop over np chained one memory round trip per
// part: twelve of them in a row set the 11 us of the critic-loss head backward).
__device__ __forceinline__ float psum(const float* p, long long idx, int np, long long ps) {
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

}  // namespace loss

struct StackArgs {
    const float* x;       // [M, din]
    const float* W1; const float* b1;   // [G,H,din], [G,H]
    const float* W2; const float* b2;   // [G,H,H],   [G,H]
    const float* W3; const float* b3;   // [G,dout,H],[G,dout]
    float* h1; float* h2;               // [G,M,H] or null
    float* out;                          // [G,M,dout]
    int M, H, din, dout, ldx;
    // optional: columns 2..3 of x are not read but computed -- the action a policy head (rrl_gauss_head_fwd /
    // rrl_stoch_head_fwd) yields for the same row -- so the head needs no launch of its own between the policy stack and
    // the critic stack that consumes its action (sac.py:192-218, qrisk.py:119-152, experiment.py:546-577)
    rrl_policy_head_t in_head;
    int use_in_head;
};

constexpr int kStackRows = 16;
constexpr int kStackMaxH = 256;

// Sum over each 16-lane row of a wave, result in every lane, in the order of the xor butterfly 8, 4, 2, 1 (bit-identical
// to `v += __shfl_xor(v, 8); ... 4; 2; 1`): after step k the row's values repeat with period 16 / 2^k, so the partner
// lane^m holds the same value as lane + m (mod 16) and a DPP row rotation delivers it -- one VALU instruction with a DPP
// operand per step instead of a ds_bpermute round trip through the LDS crossbar (~120 cycles each, four dependent
// ones per output row: 7 000 of the 25 000 cycles of a 64-row forward tile).
using rrl::row16_sum;

// -DRRL_FWD_TIMING (profiles/mlp_fwd_timing.sh builds a second library with it): wave 0 of every workgroup of the
// split forward stamps s_memtime at its phase boundaries
#ifdef RRL_FWD_TIMING
__device__ unsigned long long rrl_fwd_stamps[8 * 8192];
#define RRL_STAMP(k)                                                                              \
    do {                                                                                          \
        const unsigned flat_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);    \
        if (threadIdx.x == 0 && flat_ < 8192) {                                                   \
            rrl_fwd_stamps[8 * flat_ + (k)] = __builtin_readcyclecounter();                       \
            /* the cycle counters are not comparable across (or even within) XCDs; the 100 MHz real-time counter is global */ \
            if ((k) == 0) rrl_fwd_stamps[8 * flat_ + 6] = __builtin_amdgcn_s_memrealtime();       \
            if ((k) == 5) rrl_fwd_stamps[8 * flat_ + 7] = __builtin_amdgcn_s_memrealtime();       \
        }                                                                                         \
    } while (0)
#else
#define RRL_STAMP(k)
#endif

// R = row tiles (of 16 rows) per workgroup: they share the wave's W2 registers, so a big batch re-reads
// W2 from L2 M / (16 R) times instead of M / 16 (the re-streaming is what bounds M = 4096).
template <int R>
__device__ __forceinline__ void mlp3_fwd_body(const StackArgs& a, int bx, int g, float* h1s, float* h2s) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = bx * (R * kStackRows);
    const int H = a.H, ldh = H + 20;
    const float* W1 = a.W1 + (long long)g * H * a.din;
    const float* b1 = a.b1 + (long long)g * H;
    const float* W2 = a.W2 + (long long)g * H * H;
    const float* b2 = a.b2 + (long long)g * H;
    const float* W3 = a.W3 + (long long)g * a.dout * H;
    const float* b3 = a.b3 + (long long)g * a.dout;
    const int i = lane & 15, q = lane >> 4;
    const bool has_tile = wave * 16 < H;          // wave w owns hidden columns [16 w, 16 w + 16)
    const int n0 = has_tile ? wave * 16 : 0;

    // ---- every global read of the kernel is issued up front, branch-free, in the order of first use ---
    // layer 1 as ONE MFMA step (K = din <= 4): A = x[row i][d = q], B = W1[n0 + i][d = q]
    float xa[R];
#pragma unroll
    for (int t = 0; t < R; ++t) {
        const int xrow = min(m0 + 16 * t + i, a.M - 1);
        xa[t] = (q < a.din) ? a.x[(long long)xrow * a.ldx + q] : 0.f;
    }
    const float w1b = (q < a.din) ? W1[(n0 + i) * a.din + q] : 0.f;
    const float bias1 = b1[n0 + i];
    float4 wv[kStackMaxH / 16];                    // my 16 rows of W2: MFMA B operands of layer 2
    {
        const float* wrow = W2 + (long long)(n0 + i) * H + 4 * q;
#pragma unroll
        for (int j = 0; j < kStackMaxH / 16; ++j) wv[j] = *reinterpret_cast<const float4*>(wrow + min(16 * j, H - 16));
    }
    const float bias2 = b2[n0 + i];
    // layer 3 operands: wave w -> rows w, w + 16, ...; 16-lane group o = output index, 16 strided k per lane
    const int o3 = min(q, a.dout - 1);
    float w3v[kStackMaxH / 16];
#pragma unroll
    for (int it = 0; it < kStackMaxH / 16; ++it) w3v[it] = W3[o3 * H + min(i + 16 * it, H - 1)];
    const float bias3 = b3[o3];

    // ---- layer 1 ------------------------------------------------------------------------------------
    if (has_tile) {
#pragma unroll
        for (int t = 0; t < R; ++t) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[t], w1b, acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rr = 16 * t + 4 * q + r;
                float v = acc[r] + bias1;
                v = v > 0.f ? v : 0.f;
                h1s[rr * ldh + n0 + i] = v;
                if (a.h1 && m0 + rr < a.M) a.h1[((long long)g * a.M + m0 + rr) * H + n0 + i] = v;
            }
        }
    }
    __syncthreads();
    // ---- layer 2: R x 16 x H tile of h1 in LDS is the A operand of every wave; K order as in gemm16 ----
    if (has_tile) {
        f32x4 acc0[R], acc1[R];
#pragma unroll
        for (int t = 0; t < R; ++t) {
            acc0[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            acc1[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < kStackMaxH / 16; ++j) {
            if (16 * j < H) {
#pragma unroll
                for (int t = 0; t < R; ++t) {
                    const float4 av = *reinterpret_cast<const float4*>(h1s + (16 * t + i) * ldh + 4 * q + 16 * j);
                    acc0[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, wv[j].x, acc0[t], 0, 0, 0);
                    acc1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, wv[j].y, acc1[t], 0, 0, 0);
                    acc0[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, wv[j].z, acc0[t], 0, 0, 0);
                    acc1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, wv[j].w, acc1[t], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < R; ++t) {
            const f32x4 acc = acc0[t] + acc1[t];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rr = 16 * t + 4 * q + r;
                float v = acc[r] + bias2;
                v = v > 0.f ? v : 0.f;
                h2s[rr * ldh + n0 + i] = v;
                if (a.h2 && m0 + rr < a.M) a.h2[((long long)g * a.M + m0 + rr) * H + n0 + i] = v;
            }
        }
    }
    __syncthreads();
    // ---- layer 3: 16-lane dot products -----------------------------------------------------------------
#pragma unroll
    for (int t = 0; t < R; ++t) {
        const int r = 16 * t + wave;
        float v = 0.f;
#pragma unroll
        for (int it = 0; it < kStackMaxH / 16; ++it)
            if (i + 16 * it < H) v = fmaf(h2s[r * ldh + i + 16 * it], w3v[it], v);
        v = row16_sum(v);
        if (i == 0 && q < a.dout && m0 + r < a.M) a.out[((long long)g * a.M + m0 + r) * a.dout + q] = v + bias3;
    }
}

template <int R>
__global__ __launch_bounds__(1024) void mlp3_fwd_kernel(StackArgs a) {
    // row stride H + 20 floats: the 16 rows of a ds_read_b128 lane group land on distinct 16-byte slots
    __shared__ __attribute__((aligned(16))) float h1s[R * kStackRows * (kStackMaxH + 20)];
    __shared__ __attribute__((aligned(16))) float h2s[R * kStackRows * (kStackMaxH + 20)];
    mlp3_fwd_body<R>(a, blockIdx.x, blockIdx.y, h1s, h2s);
}

// Several independent stacks (different networks and / or different inputs) in one launch: flat grid over
// (stack, head, row tile).  The acting pass evaluates the task policy and the recovery policy on the same
// observations (experiment.py:546-577): neither depends on the other.
__device__ __forceinline__ void globalize(StackArgs& a) {
    rrl_pack::to_global_all(a.x, a.W1, a.b1, a.W2, a.b2, a.W3, a.b3, a.h1, a.h2, a.out);
    rrl_pack::globalize(a.in_head);
}
struct StackGroup {
    StackArgs a[kMaxGroup];
    float* partial[kMaxGroup];      // split variant only
    int G[kMaxGroup], tiles[kMaxGroup];
    int big[kMaxGroup];             // mixed split launch: member k runs kBigR row tiles per workgroup (else 1)
    int first[kMaxGroup + 1];
    int n;
};

template <int R>
__global__ __launch_bounds__(1024) void mlp3_fwd_group_kernel(StackGroup sg) {
    __shared__ __attribute__((aligned(16))) float h1s[R * kStackRows * (kStackMaxH + 20)];
    __shared__ __attribute__((aligned(16))) float h2s[R * kStackRows * (kStackMaxH + 20)];
    int k = 0;
    while (k + 1 < sg.n && (int)blockIdx.x >= sg.first[k + 1]) ++k;
    const int local = blockIdx.x - sg.first[k];
    mlp3_fwd_body<R>(sg.a[k], local % sg.tiles[k], local / sg.tiles[k], h1s, h2s);
}

// ---- small-batch variant of the fused stack forward: hidden-2 columns split over S = 4 workgroups -----
// With B = 256 rows the kernel above has only 16 workgroups (x heads) and each must pull all of W2
// (256 KB, ~600 wave-level loads) through ONE compute unit, which is what bounds it (~13 us).  Here each
// (16-row tile, head) is served by 4 workgroups of 4 waves; each recomputes the cheap layer 1 for all
// columns, owns 64 hidden-2 columns (64 KB of W2) and emits a PARTIAL last-layer sum; a tiny second kernel
// adds the four partials in a fixed order (deterministic).
constexpr int kSplit = 4;   // measured: 8 column groups are slower (0.335 vs 0.320 ms per iteration)
#ifndef RRL_FWD_ABLATE
#define RRL_FWD_ABLATE 0      /* TIMING ABLATIONS, results are WRONG by design (profiles/round3_w2perm/check_6_ablations.txt): bit 0 = W2
                                 fragments not loaded, bit 1 = layer-2 MFMAs skipped, bit 2 = layer 3 skipped */
#endif
#ifndef RRL_FWD_SETPRIO
#define RRL_FWD_SETPRIO 0     /* experiment: s_setprio level of a wave during layer 2 (1..3), 0 = off */
#endif
#ifndef RRL_COALESCE_W2
#define RRL_COALESCE_W2 0     /* opt-in (with -DRRL_SPLIT_PAD=4): built and measured at the end of round 3, not the default */
#endif
constexpr bool kCoalesceW2 = RRL_COALESCE_W2 != 0;   // multi-row-tile forwards: whole-line W2 loads restaged into fragment order
// HOW the whole-line loads are restaged: 1 = through per-wave LDS strips (measured, DESIGN 11: the strips or the registers
// cost a workgroup per CU); 2 = by ds_bpermute_b32 (the LDS crossbar without LDS memory: no footprint, eight temporaries per
// panel), R > 1 only; 3 = the same for the single-row-tile forwards of the update batches as well
constexpr bool kPermuteW2 = RRL_COALESCE_W2 >= 2;
constexpr bool kPermuteW2All = RRL_COALESCE_W2 >= 3;
// pad floats per row of the h1 tile (a knob of the LDS-footprint experiments: 4 keeps rows 16-byte aligned and as
// conflict-free as 20; the footprint that matters is the one that puts FOUR workgroups of the 4096-row forward on a CU:
// 35.3 KB does, 37.4 KB leaves two)
#ifndef RRL_SPLIT_PAD
#define RRL_SPLIT_PAD 20
#endif
constexpr int kSplitPad = RRL_SPLIT_PAD;

// R = row tiles (of 16 rows) per workgroup.  R = 1 for the small update batches (latency-bound: as many workgroups as
// possible).  Large batches (the acting pass, 4096 rows) are bound by re-streaming W2 from L2 once per row tile (64 MB
// per network and forward at R = 1: 14-15 us); with R > 1 a wave keeps its W2 fragments for R row tiles and the stream
// drops R-fold.  Measured at 4096 rows (profiles/mlp_fwd_probe.py; one head / two heads): plain tiling 15.4 / 20.7 us,
// R = 1 split 13.9 / 24.2, R = 4 10.4 / 16.0, R = 2 9.6 / 15.4 (more workgroups in flight per CU).  Per output element the
// arithmetic (MFMA order, partial-sum order) is the same for every R.
// HC = the hidden width as a compile-time constant (256, the reference's --hidden_size default) or 0 = read it from
// the arguments.  With HC fixed every loop below is straight-line code: no per-chunk bounds branches between the LDS
// reads and the MFMAs (the run-time version waited for each ds_read right before its four MFMAs: 3 500 cycles for the
// 2 048 cycles of MFMA issue of one 16-row tile), and the row-bounds checks are hoisted into one uniform branch.
// ZW = column splits per workgroup.  ZW = 2 (opt-in -DRRL_FWD_WIDE=1, stand-alone kernel only so far): an EIGHT-wave workgroup
// evaluates splits ZW z and ZW z + 1 of its rows -- waves 0..3 the first, 4..7 the second -- on ONE h1 tile that the eight
// waves compute together (each layer-1 column tile once instead of once per split).  The W2 stream of a launch is (row
// blocks) x |W2| whatever the split (measured: 24.4 / 17.9 us per 4096-row forward at 16 / 32 rows per workgroup, ~1 us per
// 10 MB of L2 reads, profiles/round3_w2perm/): twice the rows per workgroup at the same LDS per wave and the same 16 waves
// per CU halves it.  Per output element nothing changes (same MFMA steps, same partial sums, same order).
template <int R, int HC, int ZW = 1>
__device__ __forceinline__ void mlp3_fwd_split_body(const StackArgs& a, float* partial, int bx, int g, int z, int G,
                                                    float* h1s, float* h2s) {
    RRL_STAMP(0);
    static_assert(ZW == 1 || (HC != 0 && R > 1 && (!kCoalesceW2 || kPermuteW2)), "wide workgroups: H = 256, multi-row tiles");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int w4 = ZW == 1 ? wave : (wave & 3);          // my place among the four waves of my column split
    const int zsub = ZW == 1 ? 0 : (wave >> 2);
    const int zb = z;                                    // the workgroup's z index (ZW splits each)
    z = ZW * z + zsub;                                   // my column split
    constexpr int kW = 4 * ZW;                           // waves per workgroup
    const int m0 = bx * (R * kStackRows);
    const int H = HC ? HC : a.H, ldh = H + kSplitPad, HS = H / kSplit, ld2 = HS + 1;
    constexpr int kJ = HC ? HC / 16 : kStackMaxH / 16;             // K chunks of layer 2
    constexpr int kU = HC ? HC / (16 * kW) : kStackMaxH / (16 * kW);   // layer-1 column tiles per wave
    constexpr int kT3 = HC ? HC / (16 * kSplit) : kStackMaxH / (16 * kSplit);
    const int colbase = z * HS;
    const int M = a.M, din = a.din, dout = a.dout;
    const float* W1 = a.W1 + (long long)g * H * din;
    const float* b1 = a.b1 + (long long)g * H;
    const float* W2 = a.W2 + (long long)g * H * H;
    const float* b2 = a.b2 + (long long)g * H;
    const float* W3 = a.W3 + (long long)g * dout * H;
    const float* b3 = a.b3 + (long long)g * dout;
    float* const h1g = (a.h1 && zb == 0) ? a.h1 + ((long long)g * M + m0) * H : nullptr;
    float* const h2g = a.h2 ? a.h2 + ((long long)g * M + m0) * H : nullptr;
    const int i = lane & 15, q = lane >> 4;
    const int ntiles1 = H / 16;                    // layer-1 column tiles, 4 waves take them round-robin
    const bool has_tile2 = HC ? true : w4 * 16 < HS;   // my layer-2 tile inside this group's columns
    const int n2 = colbase + (has_tile2 ? w4 * 16 : 0);
    const bool full = m0 + R * kStackRows <= M;    // uniform: every row of the workgroup's tiles exists

    // ---- all global reads up front, branch-free ---------------------------------------------------------
    float xa[R];
#pragma unroll
    for (int t = 0; t < R; ++t) {
        const int xrow = min(m0 + 16 * t + i, M - 1);
        const float xv = a.x[(long long)xrow * a.ldx + min(q, din - 1)];
        xa[t] = (q < din) ? xv : 0.f;
    }
    if (a.use_in_head) {
        // lanes q = 0, 1 carry the observation, lanes q = 2, 3 the action dimension j = q - 2 of the policy head for
        // their row.  Wave 0 evaluates it for the workgroup's rows (the transcendental chain costs ~1 000 cycles per row
        // tile: done by every wave of four workgroups per CU it was +3.7 us on the 4096-row forward) and hands the
        // values to the other waves through LDS (the h2 tile's space: nothing lives there yet); workgroup (z, g) = (0, 0)
        // stores action and log-probability for the consumers downstream.  Same formulas, same bits as the kernels of
        // update_kernels.hip.
        const rrl_policy_head_t& hd = a.in_head;
        const int j = q & 1;
        const bool writer = z == 0 && g == 0;
        float* xs = h2s;                                 // [R * 16][4]
        if (wave == 0) {
#pragma unroll
            for (int t = 0; t < R; ++t) {
                const int row = min(m0 + 16 * t + i, M - 1);
                const bool row_ok = m0 + 16 * t + i < M;
                float val, lp_term = 0.f;
                const float e = hd.eps ? hd.eps[2 * row + j] : 0.f;
                const float sc = hd.scale[j], bi = hd.bias[j];
                if (hd.kind == RRL_HEAD_GAUSS) {
                    const float mean = loss::psum(hd.head, 4 * row + j, hd.n_part, hd.part_stride);
                    const float ls = fminf(fmaxf(loss::psum(hd.head, 4 * row + 2 + j, hd.n_part, hd.part_stride),
                                                 loss::kLogSigMin), loss::kLogSigMax);
                    const float y = tanhf(mean + expf(ls) * e);
                    val = y * sc + bi;
                    lp_term = -0.5f * e * e - ls - 0.918938533204672742f - logf(sc * (1.f - y * y) + loss::kEps);
                } else {
                    const float mean = tanhf(loss::psum(hd.head, 2 * row + j, hd.n_part, hd.part_stride)) * sc + bi;
                    val = mean + expf(fmaxf(hd.log_std[j], hd.min_log_std)) * e;
                }
                const float other = __shfl_xor(lp_term, 16);           // lane (i, 2) <-> lane (i, 3)
                float xv = xa[t];
                if (q >= 2) {
                    xv = val;
                    if (writer && row_ok) {
                        if (hd.action) hd.action[(long long)row * hd.ld_action + j] = val;
                        if (hd.logp && q == 2) hd.logp[row] = lp_term + other;
                    }
                } else if (hd.obs_in) {
                    xv = hd.obs_in[2 * row + q];
                    if (writer && row_ok && hd.obs_out) hd.obs_out[(long long)row * hd.ld_action + q] = xv;
                }
                xs[(16 * t + i) * 4 + q] = xv;
            }
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < R; ++t) xa[t] = xs[(16 * t + i) * 4 + q];
        __syncthreads();                                 // h2s is reused by layer 2 (and aliases h1s for R > 1)
    }
    float w1b[kU], bias1[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
        const int t = min(wave + kW * u, ntiles1 - 1);
        const float wv1 = W1[(t * 16 + i) * din + min(q, din - 1)];
        w1b[u] = (q < din) ? wv1 : 0.f;
        bias1[u] = b1[t * 16 + i];
    }
    // My 16 rows of W2 as MFMA B operands: lane (i, q) holds W2[n2 + i][16 j + 4 q .. + 3] in wv[j].
    // Loading them in that layout has every 16-lane pass of a global_load_dwordx4 touch 16 different 128-byte lines and use
    // 16 bytes of each: with several workgroups per CU streaming W2 at once (the 4096-row acting forwards, the packed
    // forwards) a CU pulls ~15 bytes per cycle and the loads are what the kernel waits for (profiles/
    // round3_fwd_timing_4096_2.txt: 17 000 of 28 700 cycles).  kCoalesced (the multi-row-tile kernels at H = 256): the wave
    // loads 8 rows x one whole line per instruction (lane l: row l >> 3, 16 bytes at 4 (l & 7)) and restages each
    // instruction's 1 KB through its own LDS strip into fragment order after layer 1 -- the same values in the same
    // registers, so nothing downstream changes.
    constexpr bool kCoalesced = kCoalesceW2 && (R > 1 || kPermuteW2All) && HC == 256;
    float4 wv[kJ];
    if constexpr (kCoalesced) {
        const float* wbase = W2 + (long long)(n2 + (lane >> 3)) * H + 4 * (lane & 7);
#pragma unroll
        for (int pj = 0; pj < kJ; ++pj)      // wv[2 p + c] for now: rows 8 c .. 8 c + 7, floats 32 p .. 32 p + 31
            wv[pj] = *reinterpret_cast<const float4*>(wbase + (long long)(8 * (pj & 1)) * H + 32 * (pj >> 1));
    } else if constexpr ((RRL_FWD_ABLATE & 1) != 0) {
#pragma unroll
        for (int j = 0; j < kJ; ++j) wv[j] = make_float4(float(lane + j), float(lane - j), float(j), 1.f);
    } else {
        const float* wrow = W2 + (long long)(n2 + i) * H + 4 * q;
#pragma unroll
        for (int j = 0; j < kJ; ++j) wv[j] = *reinterpret_cast<const float4*>(wrow + min(16 * j, H - 16));
    }
    const float bias2 = b2[n2 + i];
    const int o3 = min(q, dout - 1);
    float w3v[kT3];
#pragma unroll
    for (int it = 0; it < kT3; ++it) w3v[it] = W3[o3 * H + colbase + min(i + 16 * it, HS - 1)];
    const float b3v = b3[o3];
    const float bias3 = (z == 0) ? b3v : 0.f;

    RRL_STAMP(1);
    // ---- layer 1 (all H columns; one MFMA step per 16-column tile and row tile) ---------------------------
#pragma unroll
    for (int u = 0; u < kU; ++u) {
        const int t = wave + kW * u;
        if (HC || t < ntiles1) {
#pragma unroll
            for (int rt = 0; rt < R; ++rt) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[rt], w1b[u], acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rr = 16 * rt + 4 * q + r;
                    float v = acc[r] + bias1[u];
                    v = v > 0.f ? v : 0.f;
                    h1s[rr * ldh + t * 16 + i] = v;
                }
                if (h1g) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int rr = 16 * rt + 4 * q + r;
                        float v = acc[r] + bias1[u];
                        v = v > 0.f ? v : 0.f;
                        if (full || m0 + rr < M) RRL_HANDOVER_STORE(&h1g[(long long)rr * H + t * 16 + i], v);
                    }
                }
            }
        }
    }
    if constexpr (kCoalesced && kPermuteW2) {
        // Restage into fragment order after layer 1 (the loads had layer 1 to arrive under) by lane permutation: no LDS memory
        // involved, the footprint -- and with it the four workgroups per CU -- stays what it is (`restage_panels`).
        restage_panels<kJ / 2>(wv, lane);
    } else if constexpr (kCoalesced) {
        // Restage into fragment order after layer 1 (the loads had layer 1 to arrive under).  512-byte strip per wave behind
        // the h1 tile: with kSplitPad = 4 the workgroup's LDS is the 35.3 KB it was before (tile 33.3 KB + 4 x 512 B), the
        // footprint that puts FOUR workgroups on a CU -- 37 - 39 KB leave two and the second round eats the gain
        // (profiles/round3_fwd_timing_coalesce_rt.txt).  A load instruction's 1 KB goes through the strip in two halves: in
        // stage (p, c, h) the lanes of half h store their 4 rows x 128 bytes to the strip, the other half stores to the pad
        // slot of an h1 row instead (16 bytes nobody reads: no exec masking, no branches), and the lanes whose row lies in
        // that group of four (i >> 2 == 2 c + h) take their two fragments of panel p from the strip.  A wave's LDS
        // instructions execute in order: write -> read -> next write need no waits of their own.
        // NOT the default yet: this form compiles to 220 VGPRs (two waves per SIMD -- the occupancy the footprint was meant to
        // keep); the whole-instruction form with 1 KB strips compiled to 128 and was measured.
        static_assert(!kCoalesceW2 || kPermuteW2 || kSplitPad >= 4, "the pad slot of a row takes one float4");
        float* stg = h1s + R * kStackRows * (kStackMaxH + kSplitPad) + wave * 128;
        float* dump = h1s + (lane & 31) * ldh + H;             // pad columns of row lane & 31 (R >= 2: 32 rows exist)
        float* w0 = (lane < 32) ? stg + 4 * (lane & 31) : dump;   // where my float4 goes in a stage of half 0 / half 1
        float* w1 = (lane < 32) ? dump : stg + 4 * (lane & 31);
        const int grp = i >> 2;
        const int rd = (i & 3) * 32 + 4 * q;
#pragma unroll
        for (int p = 0; p < kJ / 2; ++p) {
            const float4 r0 = wv[2 * p], r1 = wv[2 * p + 1];
            const f32x4 v0 = {r0.x, r0.y, r0.z, r0.w}, v1 = {r1.x, r1.y, r1.z, r1.w};
            *reinterpret_cast<f32x4*>(w0) = v0;
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(stg + rd), a1 = *reinterpret_cast<const f32x4*>(stg + rd + 16);
            *reinterpret_cast<f32x4*>(w1) = v0;
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(stg + rd), b1 = *reinterpret_cast<const f32x4*>(stg + rd + 16);
            *reinterpret_cast<f32x4*>(w0) = v1;
            const f32x4 c0 = *reinterpret_cast<const f32x4*>(stg + rd), c1 = *reinterpret_cast<const f32x4*>(stg + rd + 16);
            *reinterpret_cast<f32x4*>(w1) = v1;
            const f32x4 d0 = *reinterpret_cast<const f32x4*>(stg + rd), d1 = *reinterpret_cast<const f32x4*>(stg + rd + 16);
            const f32x4 f0 = grp == 0 ? a0 : (grp == 1 ? b0 : (grp == 2 ? c0 : d0));
            const f32x4 f1 = grp == 0 ? a1 : (grp == 1 ? b1 : (grp == 2 ? c1 : d1));
            wv[2 * p] = make_float4(f0[0], f0[1], f0[2], f0[3]);
            wv[2 * p + 1] = make_float4(f1[0], f1[1], f1[2], f1[3]);
            __builtin_amdgcn_sched_barrier(0);      // one panel's eight reads in flight, not every panel's (216 VGPRs)
        }
    }
    __syncthreads();
    RRL_STAMP(2);
    // ---- layer 2: my 16 columns, R row tiles sharing the W2 fragments -----------------------------------------
    float* const h2z = ZW == 1 ? h2s : h2s + zsub * (R * kStackRows * ld2);      // my split's h2 tile
    if (has_tile2) {
        f32x4 acc0[R], acc1[R];
#pragma unroll
        for (int rt = 0; rt < R; ++rt) {
            acc0[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
            acc1[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#if RRL_FWD_SETPRIO
        __builtin_amdgcn_s_setprio(RRL_FWD_SETPRIO);     // experiment: waves in their MFMA phase issue before waves in a prologue / epilogue
#endif
#pragma unroll
        for (int j = 0; j < kJ; ++j) {
            if (HC || 16 * j < H) {
#pragma unroll
                for (int rt = 0; rt < R; ++rt) {
                    const float4 av = *reinterpret_cast<const float4*>(h1s + (16 * rt + i) * ldh + 4 * q + 16 * j);
#if (RRL_FWD_ABLATE & 2)
                    asm volatile("" ::"v"(av.x), "v"(av.y), "v"(av.z), "v"(av.w), "v"(wv[j].x), "v"(wv[j].y), "v"(wv[j].z), "v"(wv[j].w));
#else
                    acc0[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, wv[j].x, acc0[rt], 0, 0, 0);
                    acc1[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, wv[j].y, acc1[rt], 0, 0, 0);
                    acc0[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, wv[j].z, acc0[rt], 0, 0, 0);
                    acc1[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, wv[j].w, acc1[rt], 0, 0, 0);
#endif
                }
            }
        }
#if RRL_FWD_SETPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        // R > 1: the h2 tile reuses the h1 tile's LDS (one 70 KB tile per workgroup instead of 87 KB: two workgroups per
        // CU), so every wave must be done reading h1 first.  (All four waves own a layer-2 tile here: HC fixes H = 256.)
        if (R > 1) __syncthreads();
#pragma unroll
        for (int rt = 0; rt < R; ++rt) {
            const f32x4 acc = acc0[rt] + acc1[rt];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rr = 16 * rt + 4 * q + r;
                float v = acc[r] + bias2;
                v = v > 0.f ? v : 0.f;
                h2z[rr * ld2 + w4 * 16 + i] = v;
            }
            if (h2g) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rr = 16 * rt + 4 * q + r;
                    float v = acc[r] + bias2;
                    v = v > 0.f ? v : 0.f;
                    if (full || m0 + rr < M) RRL_HANDOVER_STORE(&h2g[(long long)rr * H + n2 + i], v);
                }
            }
        }
    }
    RRL_STAMP(3);
    __syncthreads();
    RRL_STAMP(4);
    // ---- layer 3 partial over my HS columns: wave w -> rows 4 w .. 4 w + 3 of every row tile -----------------
    float res[R][4];
#pragma unroll
    for (int rt = 0; rt < R; ++rt) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int r = 16 * rt + w4 * 4 + rr;
            float v = 0.f;
#pragma unroll
            for (int it = 0; it < ((RRL_FWD_ABLATE & 4) ? 0 : kT3); ++it) {
                const float hv = h2z[r * ld2 + min(i + 16 * it, HS - 1)];
                if (HC || i + 16 * it < HS) v = fmaf(hv, w3v[it], v);
            }
            res[rt][rr] = v;
        }
    }
#pragma unroll
    for (int rt = 0; rt < R; ++rt) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const float v = row16_sum(res[rt][rr]);
            const int r = 16 * rt + w4 * 4 + rr;
            if (i == 0 && q < dout && (full || m0 + r < M))
                partial[(((long long)z * G + g) * M + m0 + r) * dout + q] = v + bias3;
        }
    }
    RRL_STAMP(5);
}

#ifndef RRL_BIG_R
#define RRL_BIG_R 2
#endif
constexpr int kBigR = RRL_BIG_R;     // row tiles per workgroup for batches above kSplitSmallM rows
#ifndef RRL_PACK_R
#define RRL_PACK_R 4
#endif
#ifndef RRL_PACK_MIN_SEEDS
#define RRL_PACK_MIN_SEEDS 99     /* measured: 4 row tiles per workgroup are not faster than 2 at S = 4, 8 (0.401 vs 0.389 ms, 0.676 vs 0.649) */
#endif
constexpr int kPackR = RRL_PACK_R;   // ... of the packed launch from kPackMinSeeds seeds on
constexpr int kPackMinSeeds = RRL_PACK_MIN_SEEDS;
constexpr int kSplitSmallM = 1024;
constexpr size_t split_lds_floats(int R) {
    return size_t(R) * kStackRows * (kStackMaxH + kSplitPad) +
           (R > 1 ? (kCoalesceW2 && !kPermuteW2 ? 4 * 128 : 0)                    // R > 1: h2 aliases h1; W2 restaging strips
                  : size_t(R) * kStackRows * (kStackMaxH / kSplit + 1));
}

// Which workgroup evaluates which (row block, head, column split) -- an experiment on the stand-alone kernel (opt-in
// -DRRL_FWD_XCD_MAP=1 / 2).  All workgroups of a 4096-row forward start in ONE round, four per CU, and in launch order
// the four on a CU hold four DIFFERENT (head, split) slices of W2: every CU pulls 4 x 64 KB through its L1 at the same time.
// Workgroups go to the XCDs round-robin (id % 8); inside an XCD (local index m = id / 8) the map gives the same slice to the
// workgroups that share a CU, so that one L1 fill can serve four of them, under either assumption about the dispatcher:
//   1: CUs are dealt workgroups round-robin (m, m + 32, m + 64, m + 96 share a CU)     2: a CU is filled first (4 c .. 4 c + 3)
// A bijection of the grid: every (row block, head, split) is still evaluated exactly once, by the same code.
#ifndef RRL_FWD_XCD_MAP
#define RRL_FWD_XCD_MAP 0
#endif
__device__ __forceinline__ void split_block_of(int& bx, int& g, int& z) {
    bx = blockIdx.x, g = blockIdx.y, z = blockIdx.z;
    if constexpr (RRL_FWD_XCD_MAP != 0) {
        const int nx = gridDim.x, G = gridDim.y, S = G * kSplit;
        const int rows_per = nx / 8, q = 32 / S;                       // row blocks per XCD and slice; CUs per slice (map 1)
        const bool ok = nx % 8 == 0 && S <= 32 && 32 % S == 0 && rows_per % 4 == 0 && (RRL_FWD_XCD_MAP != 1 || rows_per % q == 0);
        if (!ok) return;
        const int id = blockIdx.x + nx * (blockIdx.y + G * blockIdx.z);
        const int k = id % 8, m = id / 8;
        int s, r;
        if (RRL_FWD_XCD_MAP == 1) {
            s = (m % 32) / q;
            r = q * (m / 32) + m % q;
        } else {
            s = (m / 4) % S;
            r = 4 * (m / (4 * S)) + m % 4;
        }
        bx = 8 * r + k, g = s % G, z = s / G;
    }
}

#ifndef RRL_FWD_WIDE
#define RRL_FWD_WIDE 0
#endif
#ifndef RRL_WIDE_R
#define RRL_WIDE_R 4
#endif
#if RRL_FWD_WIDE
constexpr bool kFwdWide = true;
constexpr int kWideR = RRL_WIDE_R;
// eight waves, two column splits, kWideR row tiles on one h1 tile (see mlp3_fwd_split_body); H = 256 only
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void mlp3_fwd_split_wide_kernel(StackArgs a, float* partial) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    mlp3_fwd_split_body<kWideR, 256, 2>(a, partial, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.y, lds, lds);
}
#endif

template <int R>
__global__ __launch_bounds__(256) void mlp3_fwd_split_kernel(StackArgs a, float* partial) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* h2s = R > 1 ? lds : lds + R * kStackRows * (kStackMaxH + kSplitPad);
    int bx, g, z;
    split_block_of(bx, g, z);
    if (a.H == 256) mlp3_fwd_split_body<R, 256>(a, partial, bx, g, z, gridDim.y, lds, h2s);
    else mlp3_fwd_split_body<R, 0>(a, partial, bx, g, z, gridDim.y, lds, h2s);
}

// flat grid over (stack, column split, head, row tile)
template <int R>
__device__ __forceinline__ void mlp3_fwd_split_group_body(const StackGroup& sg, int block, float* lds) {
    int k = 0;
    while (k + 1 < sg.n && block >= sg.first[k + 1]) ++k;
    const int local = block - sg.first[k];
    const int bx = local % sg.tiles[k], rest = local / sg.tiles[k];
    float* h2s = R > 1 ? lds : lds + R * kStackRows * (kStackMaxH + kSplitPad);
    StackArgs a = sg.a[k];                   // this workgroup's member, copied out of the group (see gemm16_group_body)
    globalize(a);
    float* partial = sg.partial[k];
    rrl_pack::to_global(partial);
    const int G = sg.G[k];
    if (a.H == 256) mlp3_fwd_split_body<R, 256>(a, partial, bx, rest % G, rest / G, G, lds, h2s);
    else mlp3_fwd_split_body<R, 0>(a, partial, bx, rest % G, rest / G, G, lds, h2s);
}

template <int R>
__global__ __launch_bounds__(256) void mlp3_fwd_split_group_kernel(StackGroup sg) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    mlp3_fwd_split_group_body<R>(sg, blockIdx.x, lds);
}

// Members of BOTH kinds in one launch -- small batches (R = 1) next to large ones (R = kBigR row tiles per workgroup): the
// 4096-row forwards of the acting pass ride in the launches of the Q_risk update's 256-row forwards that become ready at
// the same points of the iteration.  Every member runs the body of its own kind: same arithmetic, same bits.
__device__ __forceinline__ void mlp3_fwd_split_mixed_body(const StackGroup& sg, int block, float* lds) {
    int k = 0;
    while (k + 1 < sg.n && block >= sg.first[k + 1]) ++k;
    const int local = block - sg.first[k];
    const int bx = local % sg.tiles[k], rest = local / sg.tiles[k];
    StackArgs a = sg.a[k];
    globalize(a);
    float* partial = sg.partial[k];
    rrl_pack::to_global(partial);
    const int G = sg.G[k];
    if (sg.big[k]) {
        if (a.H == 256) mlp3_fwd_split_body<kBigR, 256>(a, partial, bx, rest % G, rest / G, G, lds, lds);
        else mlp3_fwd_split_body<kBigR, 0>(a, partial, bx, rest % G, rest / G, G, lds, lds);
    } else {
        float* h2s = lds + kStackRows * (kStackMaxH + kSplitPad);
        if (a.H == 256) mlp3_fwd_split_body<1, 256>(a, partial, bx, rest % G, rest / G, G, lds, h2s);
        else mlp3_fwd_split_body<1, 0>(a, partial, bx, rest % G, rest / G, G, lds, h2s);
    }
}

__global__ __launch_bounds__(256) void mlp3_fwd_split_mixed_kernel(StackGroup sg) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    mlp3_fwd_split_mixed_body(sg, blockIdx.x, lds);
}

__global__ __launch_bounds__(256) void mlp3_fwd_split_mixed_pack_kernel(const StackGroup* __restrict__ groups,
                                                                        rrl_pack::Idx ix) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    int s, local;
    if (!rrl_pack::locate(ix, blockIdx.x, s, local)) return;
    mlp3_fwd_split_mixed_body(groups[s], local, lds);
}

// the same launch for S seeds (pack.hpp): seed s runs its group on workgroups [first[s], first[s + 1])
template <int R>
__global__ __launch_bounds__(256) void mlp3_fwd_split_pack_kernel(const StackGroup* __restrict__ groups, rrl_pack::Idx ix) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    int s, local;
    if (!rrl_pack::locate(ix, blockIdx.x, s, local)) return;
    mlp3_fwd_split_group_body<R>(groups[s], local, lds);
}

__global__ void sum_partials_kernel(int n, const float* __restrict__ partial, float* __restrict__ out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    float v = partial[e];
#pragma unroll
    for (int z = 1; z < kSplit; ++z) v += partial[(long long)z * n + e];
    out[e] = v;
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


// dL/d action[b][j]: over the critic heads that consumed the action and, when the critic's first-layer backward came out
// of the hidden-layer tiles (rrl_first_layer_t), over their column-tile partials -- up to 2 x 16 loads, all issued
// before the first add, summed head by head, tile by tile.
__device__ __forceinline__ float d_action_sum(const rrl_loss_t& a, int b, int j) {
    const float* p = a.d_action + (long long)b * a.ld + j;
    const int parts = a.da_parts > 1 ? a.da_parts : 1;
    float da = 0.f;
    for (int hd = 0; hd < a.n_heads; ++hd) {
        float v[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) v[t] = p[hd * a.head_stride + (t < parts ? t : 0) * a.da_part_stride];
#pragma unroll
        for (int t = 0; t < 16; ++t) da = t < parts ? da + v[t] : da;
    }
    return da;
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
    } else if constexpr (KIND == RRL_LOSS_GAUSS_HEAD) {
        const int j = o & 1;
        float da = 0.f;
        da = d_action_sum(a, b, j);
        const float mean = psum(a.out, 4 * b + j, np, ps);
        const float raw = psum(a.out, 4 * b + 2 + j, np, ps);
        const float ls = fminf(fmaxf(raw, kLogSigMin), kLogSigMax);
        const float sd = expf(ls), e = a.v0[2 * b + j], sc = a.v1[j];
        const float y = tanhf(mean + sd * e);
        const float one_m = 1.f - y * y;
        const float dx = da * sc * one_m + a.f0 * (2.f * sc * y * one_m) / (sc * one_m + kEps);
        if (o < 2) return dx;
        const bool inside = (raw >= kLogSigMin) & (raw <= kLogSigMax);
        return inside ? (dx * sd * e - a.f0) : 0.f;
    } else {
        const int j = o;
        const float t = tanhf(psum(a.out, 2 * b + j, np, ps));
        float da = 0.f;
        da = d_action_sum(a, b, j);
        const float sd = expf(fmaxf(a.v1[j], a.f0));
        term = (a.v1[j] >= a.f0) ? da * sd * a.v0[2 * b + j] : 0.f;
        return da * a.v2[j] * (1.f - t * t);
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
    float* __restrict__ dh2 = hb.dh2 ? hb.dh2 + (long long)g * B * H : nullptr;     // null: the hidden-layer tiles generate it
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
    } else if constexpr (KIND == RRL_LOSS_GAUSS_HEAD) {
        // one thread per (row, action dim): the mean and log-std gradients share tanh/exp
        for (int e = threadIdx.x; e < B * 2; e += 256) {
            const int b = e >> 1, j = e & 1;
            float term;
            const float dx = loss::dout_at<KIND>(la, B, g, b, j, term);
            const float raw = loss::psum(la.out, 4 * b + 2 + j, la.n_part, la.part_stride);
            const bool inside = (raw >= loss::kLogSigMin) & (raw <= loss::kLogSigMax);
            const float sd = expf(fminf(fmaxf(raw, loss::kLogSigMin), loss::kLogSigMax));
            dsh[4 * b + j] = dx;
            dsh[4 * b + 2 + j] = inside ? (dx * sd * la.v0[2 * b + j] - la.f0) : 0.f;
        }
    } else {
        // one thread per batch row (all its outputs): the per-thread partial sums of the loss terms are then the ones of
        // the stand-alone kernels (update_kernels.hip), and so is every bit of the reduced value
        for (int b = threadIdx.x; b < B; b += 256) {
#pragma unroll
            for (int o = 0; o < (DOUT ? DOUT : 1); ++o) {
                float term;
                dsh[b * dout + o] = loss::dout_at<KIND>(la, B, g, b, o, term);
                if (KIND == RRL_LOSS_STOCH_HEAD && o == 1) lsum[1] += term; else lsum[0] += term;
            }
        }
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
                if (hb.dh2) RRL_HANDOVER_STORE(&dh2[(long long)b * H + h], a[it] > 0.f ? d : 0.f);
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
                    if (hok && hb.dh2) RRL_HANDOVER_STORE(&dh2[(long long)b * H + h], a[it] > 0.f ? d : 0.f);
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

__device__ __forceinline__ void head_bwd_group_body(const HeadBwdGroup& hg, int block, float (*red)[4][kCols], float* dsh) {
    int k = 0;
    while (k + 1 < hg.n && block >= hg.first[k + 1]) ++k;
    const int local = block - hg.first[k];
    HeadBwdArgs hb = hg.p[k];                // this workgroup's member, copied out of the group (see gemm16_group_body)
    globalize(hb);
    head_bwd_dispatch(hb, local % hg.blocks_x[k], local / hg.blocks_x[k], red, dsh);
}

__global__ __launch_bounds__(256) void head_bwd_group_kernel(HeadBwdGroup hg) {
    __shared__ float red[kSlices][4][kCols];
    __shared__ float dsh[1024 * 4];
    head_bwd_group_body(hg, blockIdx.x, red, dsh);
}

__global__ __launch_bounds__(256) void head_bwd_pack_kernel(const HeadBwdGroup* __restrict__ groups, rrl_pack::Idx ix) {
    __shared__ float red[kSlices][4][kCols];
    __shared__ float dsh[1024 * 4];
    int s, local;
    if (!rrl_pack::locate(ix, blockIdx.x, s, local)) return;
    head_bwd_group_body(groups[s], local, red, dsh);
}

// ---- head backward INSIDE the hidden-layer launch (one-output heads: the four critic-type losses) -------------------
// A stack backward was head launch (dW3, db3, loss scalars, dh2 -> memory) -> hidden launch (reads dh2).  Here one launch
// does both for every fused member: its first blocks run the head body WITHOUT the dh2 store, the 16 x 16 tiles of the two
// H x H products generate their dh2 operand from (h2, W3, dOut) -- dOut[b] evaluated per workgroup from the loss description
// into LDS (B values: a few loads per row, all in flight together).  256-thread workgroups: a tile workgroup runs on its
// first wave, the other three leave at once (the tile code is written for one wave per 16 x 16 tile).
struct FusedHiddenGroup {
    HiddenGroup hg;
    HeadBwdArgs head[kMaxGroup];
    int fused[kMaxGroup];
    int head_blocks[kMaxGroup];          // blocks_x * G of the head body (0: member not fused)
    int blocks_x[kMaxGroup];
    int G[kMaxGroup];
    int first[kMaxGroup + 1];            // block ranges incl. the head blocks
};

// dOut[b] of the rows a tile contracts over, into LDS: TN tiles need all B <= 256 rows (four per lane, every row's loads
// issued before the first formula is evaluated: ONE memory round trip), NN tiles their own 16
template <int KIND>
__device__ __forceinline__ void fill_dout(const rrl_loss_t& la, int B, int g, float* dsh, bool tn, int m0) {
    float term;
    const int lane = threadIdx.x & 63;
    if (tn) {
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = lane + 64 * i;
            v[i] = loss::dout_at<KIND>(la, B, g, r < B ? r : B - 1, 0, term);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (lane + 64 * i < B) dsh[dsh_index(lane + 64 * i)] = v[i];
    } else if (lane < kTile) {
        dsh[dsh_index(m0 + lane)] = loss::dout_at<KIND>(la, B, g, m0 + lane, 0, term);
    }
}

__device__ __forceinline__ void fused_hidden_body(const FusedHiddenGroup& fg, int block, float* lds) {
    int k = 0;
    while (k + 1 < fg.hg.n && block >= fg.first[k + 1]) ++k;
    int local = block - fg.first[k];
    if (local < fg.head_blocks[k]) {                   // head body: dW3, db3, loss scalars (no dh2 store)
        const HeadBwdArgs hb = fg.head[k];
        head_bwd_dispatch(hb, local % fg.blocks_x[k], local / fg.blocks_x[k],
                          reinterpret_cast<float(*)[4][kCols]>(lds), lds + kSlices * 4 * kCols);
        return;
    }
    // tiles: FOUR per workgroup, one per wave, each wave on its own As | Bs region and never waiting for another
    const int wave = threadIdx.x >> 6;
    float* As = lds + wave * (2 * kPanel * kLd);
    float* Bs = As + kPanel * kLd;
    float* dsh = As + 16;                              // pad columns of the As tile (dsh_index)
    const HiddenGroup& hg = fg.hg;
    const int tile = (local - fg.head_blocks[k]) * 4 + wave;
    if (tile >= hg.per_head[k] * fg.G[k]) return;
    const int g = tile / hg.per_head[k], b = tile - g * hg.per_head[k];
    const bool tn = b < hg.tn_tiles[k];
    const int c = tn ? b : b - hg.tn_tiles[k];
    const int tx = tn ? hg.tn_tiles_x[k] : hg.nn_tiles_x[k];
    const GemmArgs ga = tn ? hg.tn[k] : hg.nn[k];     // (per WAVE here: not passed through rrl_pack::to_global, whose "s" wants scalars)
    if (!fg.fused[k]) {                                // (members of a fused launch have full aligned tiles)
        if (tn) gemm16_tile<2, true, false, NoPrologue, true>(ga, As, Bs, c % tx, c / tx, g);
        else gemm16_tile<1, true, false, NoPrologue, true>(ga, As, Bs, c % tx, c / tx, g);
        return;
    }
    // dOut of the rows this tile contracts over (TN: all B rows; NN: its own 16)
    const rrl_loss_t la = fg.head[k].la;
    const int B = fg.head[k].B;
    const int m0 = (c / tx) * kTile;
    auto prologue = [&]() {
        switch (la.kind) {
            case RRL_LOSS_SAC_CRITIC: fill_dout<RRL_LOSS_SAC_CRITIC>(la, B, g, dsh, tn, m0); break;
            case RRL_LOSS_SAC_POLICY: fill_dout<RRL_LOSS_SAC_POLICY>(la, B, g, dsh, tn, m0); break;
            case RRL_LOSS_QRISK_CRITIC: fill_dout<RRL_LOSS_QRISK_CRITIC>(la, B, g, dsh, tn, m0); break;
            default: fill_dout<RRL_LOSS_QRISK_POLICY>(la, B, g, dsh, tn, m0); break;
        }
    };
    if (tn) gemm16_tile<2, true, true, decltype(prologue), true>(ga, As, Bs, c % tx, c / tx, g, dsh, prologue);
    else gemm16_tile<1, true, true, decltype(prologue), true>(ga, As, Bs, c % tx, c / tx, g, dsh, prologue);
}

// LDS: four tile regions (As | Bs) of 20 KB, one per wave = 80 KB: two workgroups = eight tiles per CU, what the 64-thread
// tile kernel has too (a 256-thread workgroup whose tile ran on one wave held four wave slots at 196 VGPRs: two workgroups =
// two tiles per CU, three rounds per launch).  A head workgroup lays its buffers over the first region.
constexpr int kFusedLdsFloats = 4 * 2 * kPanel * kLd;
static_assert(kSlices * 4 * kCols + 1024 * 4 <= 2 * kPanel * kLd, "head buffers fit a tile region");

__global__ __launch_bounds__(256) void hidden_head_group_kernel(FusedHiddenGroup fg) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    fused_hidden_body(fg, blockIdx.x, lds);
}

__global__ __launch_bounds__(256) void hidden_head_pack_kernel(const FusedHiddenGroup* __restrict__ groups,
                                                               rrl_pack::Idx ix) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    int s, local;
    if (!rrl_pack::locate(ix, blockIdx.x, s, local)) return;
    fused_hidden_body(groups[s], local, lds);
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

// ---- packed launches: the same group launch for S seeds side by side (pack.hpp) ----
template <class Member>
static bool pack_key(int site, int S, const int* n, const Member* const* members, rrl_pack::Key& key) {
    if (S <= 0 || S > rrl_pack::kMaxSeeds || !n || !members) return false;
    key.pod(site);
    key.pod(S);
    for (int s = 0; s < S; ++s) {
        if (n[s] <= 0 || n[s] > kMaxGroup || !members[s]) return false;
        key.pod(n[s]);
        key.add(members[s], sizeof(Member) * n[s]);
    }
    return true;
}

template <class Group, class Member, class Build>
static int build_pack(int S, const int* n, const Member* const* members, std::vector<Group>& groups, rrl_pack::Idx& ix,
                      Build build) {
    groups.resize(S);
    ix.S = S;
    ix.first[0] = 0;
    for (int s = 0; s < S; ++s) {
        const int rc = build(n[s], members[s], groups[s]);
        if (rc != RRL_OK) return rc;
        ix.first[s + 1] = ix.first[s] + groups[s].first[n[s]];
    }
    for (int s = S; s < rrl_pack::kMaxSeeds; ++s) ix.first[s + 1] = ix.first[S];
    return RRL_OK;
}

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
        if ((!p.dh2 && !p.fuse_head) || !p.h1 || !p.W2 || ((p.dW2 == nullptr) != (p.db2 == nullptr))) return RRL_EINVAL;
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

static int head_loss_args(const rrl_loss_t* la, int G, int B, int H, int dout, const float* h2, const float* W3,
                          float* dW3, float* db3, float* dh2, HeadBwdArgs& hb);
static bool grant_lds(const void* kernel, size_t bytes);

static int build_fused_hidden_group(int n, const rrl_hidden_bwd_t* ps, FusedHiddenGroup& fg) {
    const int rc = build_hidden_group(n, ps, fg.hg);
    if (rc != RRL_OK) return rc;
    fg.first[0] = 0;
    for (int k = 0; k < n; ++k) {
        const rrl_hidden_bwd_t& p = ps[k];
        fg.fused[k] = p.fuse_head != 0;
        fg.head_blocks[k] = 0;
        fg.blocks_x[k] = 1;
        fg.head[k] = HeadBwdArgs{};
        if (p.fuse_head) {
            const rrl_head_bwd_t& h = p.head;
            if (h.loss.kind < RRL_LOSS_SAC_CRITIC || h.loss.kind > RRL_LOSS_QRISK_POLICY || h.dout != 1 || h.G != p.G ||
                h.B != p.B || h.H != p.H || !fg.hg.fast[k] || p.B % kPanel || p.H % kPanel || p.B > 256)
                return RRL_EINVAL;
            const int r2 = head_loss_args(&h.loss, h.G, h.B, h.H, h.dout, h.h2, h.W3, h.dW3, h.db3, nullptr, fg.head[k]);
            if (r2 != RRL_OK) return r2;
            fg.blocks_x[k] = (p.H + kCols - 1) / kCols;
            fg.head_blocks[k] = fg.blocks_x[k] * p.G;
            fg.hg.tn[k].gen_h2 = fg.hg.nn[k].gen_h2 = h.h2;
            fg.hg.tn[k].gen_w3 = fg.hg.nn[k].gen_w3 = h.W3;
        }
        if (!fg.hg.fast[k]) return RRL_EINVAL;               // every member of a fused launch: full aligned tiles
        fg.G[k] = p.G;
        fg.first[k + 1] = fg.first[k] + fg.head_blocks[k] + (fg.hg.per_head[k] * p.G + 3) / 4;
    }
    for (int k = n; k < kMaxGroup; ++k) {
        fg.first[k + 1] = fg.first[n];
        fg.G[k] = 1;
        fg.fused[k] = fg.head_blocks[k] = 0;
        fg.blocks_x[k] = 1;
        fg.head[k] = HeadBwdArgs{};
    }
    return RRL_OK;
}

static bool any_fused(int n, const rrl_hidden_bwd_t* ps) {
    for (int k = 0; ps && k < n && k < kMaxGroup; ++k)
        if (ps[k].fuse_head) return true;
    return false;
}

int rrl_mlp_hidden_backward_multi(int n, const rrl_hidden_bwd_t* ps, void* stream) {
    if (any_fused(n, ps)) {
        FusedHiddenGroup fg;
        const int rc = build_fused_hidden_group(n, ps, fg);
        if (rc != RRL_OK) return rc;
        static const bool ok = grant_lds((const void*)hidden_head_group_kernel, kFusedLdsFloats * 4);
        if (!ok) return RRL_ERANGE;
        hipLaunchKernelGGL(hidden_head_group_kernel, dim3(fg.first[n]), dim3(256), kFusedLdsFloats * 4, (hipStream_t)stream, fg);
        return check_launch();
    }
    HiddenGroup hg;
    const int rc = build_hidden_group(n, ps, hg);
    if (rc != RRL_OK) return rc;
    hipLaunchKernelGGL(gemm16_group_kernel, dim3(hg.first[n]), dim3(64), 0, (hipStream_t)stream, hg);
    return check_launch();
}

// Packed launches are throughput-bound from a few seeds on (more tiles / workgroups than the chip holds at once), where the
// solo kernels' shapes -- chosen for the latency of ONE seed -- are not the best ones.  Seeds from which the packed launch
// switches shape (per output element the arithmetic is the same either way; RRL_PACK_* override the measured defaults):
//   hidden-layer backward: 64-wide K panels from 2 seeds on, 32-wide from 3 (10 / 5 KB of LDS per tile instead of 20: 16 / 32
//                          tiles in flight per CU instead of 8; one seed has ~6 tiles per CU and launch, four have 24)
//   B <= 1024 forwards   : 2 row tiles per workgroup from 3 seeds on (half the workgroups, each W2 fragment used twice)
// Measured (profiles/packed_ab.sh, ms per packed iteration at 16 updates per step, S = 2 / 3 / 4 / 8): solo shapes
// 3.12 / 3.99 / 4.06 / 5.77, these 3.02 / 3.66 / 3.72 / 5.13; each alone and the other panel widths in profiles/README.md.
static int pack_threshold(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
static int pack_panel(int S) {
    static const int min64 = pack_threshold("RRL_PACK_PANEL64_MIN_SEEDS", 2), min32 = pack_threshold("RRL_PACK_PANEL32_MIN_SEEDS", 3);
    return S >= min32 ? 32 : (S >= min64 ? 64 : kPanel);
}
static int pack_small_r2_min_seeds() {
    static const int v = pack_threshold("RRL_PACK_SMALL_R2_MIN_SEEDS", 3);
    return v;
}

// Block form of the packed hidden-layer backward (gemm_block_pack_kernel): RRL_PACK_BLOCK = "WM WN" as two digits (22: 64 x 64
// blocks, 12: 32 x 64, 11: 32 x 32), 0 = the tile kernel; RRL_PACK_BLOCK_MIN_SEEDS = seeds from which it is used.
static int pack_block(int S) {
    static const int shape = pack_threshold("RRL_PACK_BLOCK", 12), min_seeds = pack_threshold("RRL_PACK_BLOCK_MIN_SEEDS", 3);
    return S >= min_seeds ? shape : 0;
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

int rrl_mlp_hidden_backward_multi_packed(int S, const int* n, const rrl_hidden_bwd_t* const* members, void* stream) {
    rrl_pack::Key key;
    if (!pack_key(1, S, n, members, key)) return RRL_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    rrl_pack::Plan* plan = rrl_pack::lookup(key);
    if (!plan) {
        bool fused = false;
        for (int s = 0; s < S; ++s) fused = fused || any_fused(n[s], members[s]);
        rrl_pack::Idx ix;
        int shape = 0;
        if (fused) {
            std::vector<FusedHiddenGroup> groups;
            const int rc = build_pack<FusedHiddenGroup>(S, n, members, groups, ix, build_fused_hidden_group);
            if (rc != RRL_OK) return rc;
            static const bool ok = grant_lds((const void*)hidden_head_pack_kernel, kFusedLdsFloats * 4);
            if (!ok) return RRL_ERANGE;
            plan = rrl_pack::store(key, groups.data(), sizeof(FusedHiddenGroup) * S, st);
        } else {
            std::vector<HiddenGroup> groups;
            int rc = build_pack<HiddenGroup>(S, n, members, groups, ix, build_hidden_group);
            if (rc != RRL_OK) return rc;
            shape = pack_block(S);
            if (shape) {       // every seed's members in whole blocks, or the launch keeps the tile kernel
                const int wm = shape / 10, wn = shape % 10;
                bool ok = (wm == 1 || wm == 2) && (wn == 1 || wn == 2) && wm <= wn;
                std::vector<HiddenGroup> blocks = groups;
                for (int s = 0; ok && s < S; ++s) ok = hidden_blocks(n[s], members[s], blocks[s], wm, wn);
                if (ok) {
                    groups.swap(blocks);
                    for (int s = 0; s < S; ++s) ix.first[s + 1] = ix.first[s] + groups[s].first[n[s]];
                    for (int s = S; s < rrl_pack::kMaxSeeds; ++s) ix.first[s + 1] = ix.first[S];
                } else {
                    shape = 0;
                }
            }
            plan = rrl_pack::store(key, groups.data(), sizeof(HiddenGroup) * S, st);
        }
        if (!plan) return RRL_ELAUNCH;
        plan->grid = rrl_pack::finish(ix);
        plan->ix = ix;
        plan->i0 = fused;
        plan->i1 = shape ? -shape : pack_panel(S);
    }
    if (plan->i0)
        hipLaunchKernelGGL(hidden_head_pack_kernel, dim3(plan->grid), dim3(256), kFusedLdsFloats * 4, st, (const FusedHiddenGroup*)plan->dev,
                           plan->ix);
    else if (plan->i1 == -22)
        hipLaunchKernelGGL((gemm_block_pack_kernel<2, 2>), dim3(plan->grid), dim3(256), 0, st, (const HiddenGroup*)plan->dev, plan->ix);
    else if (plan->i1 == -12)
        hipLaunchKernelGGL((gemm_block_pack_kernel<1, 2>), dim3(plan->grid), dim3(256), 0, st, (const HiddenGroup*)plan->dev, plan->ix);
    else if (plan->i1 == -11)
        hipLaunchKernelGGL((gemm_block_pack_kernel<1, 1>), dim3(plan->grid), dim3(256), 0, st, (const HiddenGroup*)plan->dev, plan->ix);
    else if (plan->i1 == 64)
        hipLaunchKernelGGL(gemm16_pack_kernel<64>, dim3(plan->grid), dim3(64), 0, st, (const HiddenGroup*)plan->dev, plan->ix);
    else if (plan->i1 == 32)
        hipLaunchKernelGGL(gemm16_pack_kernel<32>, dim3(plan->grid), dim3(64), 0, st, (const HiddenGroup*)plan->dev, plan->ix);
    else
        hipLaunchKernelGGL(gemm16_pack_kernel<kPanel>, dim3(plan->grid), dim3(64), 0, st, (const HiddenGroup*)plan->dev, plan->ix);
    return check_launch();
}

static int split_max_rows() {
    // RRL_SPLIT_MAX_M: tuning knob for profiles/mlp_fwd_probe.py (largest batch that takes the column-split forward)
    static const int v = [] {
        const char* e = getenv("RRL_SPLIT_MAX_M");
        return e ? atoi(e) : (1 << 30);
    }();
    return v;
}

int rrl_mlp3_is_split(int M, int H) {
    return (M <= split_max_rows() && (H % (16 * kSplit)) == 0 && H <= kStackMaxH) ? kSplit : 0;
}

// tiles of R >= 4 need more than the default 64 KB of LDS per workgroup (gfx950 has 160 KB per CU): opt in once
static bool grant_lds(const void* kernel, size_t bytes) {
    if (bytes <= 64 * 1024) return true;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, int(bytes)) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return true;
}

static int stack_check(int G, int M, int H, int din, int dout, const float* x, const float* W1, const float* b1,
                       const float* W2, const float* b2, const float* W3, const float* b3, const float* out) {
    if (!x || !W1 || !b1 || !W2 || !b2 || !W3 || !b3 || !out) return RRL_EINVAL;
    if (G <= 0 || G > 65535 || M <= 0 || din <= 0 || din > 4 || dout <= 0 || dout > 4) return RRL_ERANGE;
    if (H <= 0 || H > kStackMaxH || (H % 16) != 0) return RRL_ERANGE;
    return RRL_OK;
}

int rrl_mlp3_forward(int G, int M, int H, int din, int dout, const float* x, int ldx, const float* W1,
                     const float* b1, const float* W2, const float* b2, const float* W3, const float* b3,
                     float* h1, float* h2, float* out, float* scratch, int finalize, void* stream) {
    const int rc = stack_check(G, M, H, din, dout, x, W1, b1, W2, b2, W3, b3, out);
    if (rc != RRL_OK) return rc;
    StackArgs a{x, W1, b1, W2, b2, W3, b3, h1, h2, out, M, H, din, dout, ldx, rrl_policy_head_t{}, 0};
    if (scratch && rrl_mlp3_is_split(M, H)) {
        // 4 workgroups (column groups) per row tile + fixed-order sum of their partial last-layer outputs
        if (M <= kSplitSmallM || H != 256) {
            hipLaunchKernelGGL(mlp3_fwd_split_kernel<1>, dim3((M + kStackRows - 1) / kStackRows, G, kSplit), dim3(256),
                               split_lds_floats(1) * 4, (hipStream_t)stream, a, scratch);
#if RRL_FWD_WIDE
        } else if (kFwdWide) {
            static const bool ok = grant_lds((const void*)mlp3_fwd_split_wide_kernel, split_lds_floats(kWideR) * 4);
            if (!ok) return RRL_ERANGE;
            const int rows = kWideR * kStackRows;
            hipLaunchKernelGGL(mlp3_fwd_split_wide_kernel, dim3((M + rows - 1) / rows, G, kSplit / 2), dim3(512),
                               split_lds_floats(kWideR) * 4, (hipStream_t)stream, a, scratch);
#endif
        } else {
            static const bool ok = grant_lds((const void*)mlp3_fwd_split_kernel<kBigR>, split_lds_floats(kBigR) * 4);
            if (!ok) return RRL_ERANGE;
            const int rows = kBigR * kStackRows;
            hipLaunchKernelGGL(mlp3_fwd_split_kernel<kBigR>, dim3((M + rows - 1) / rows, G, kSplit), dim3(256),
                               split_lds_floats(kBigR) * 4, (hipStream_t)stream, a, scratch);
        }
        if (finalize) {
            const int n = G * M * dout;
            hipLaunchKernelGGL(sum_partials_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n,
                               scratch, out);
        }
        return check_launch();
    }
    // more than one workgroup per CU (256 CUs): two row tiles per workgroup halve the W2 re-streaming
    if ((long long)((M + kStackRows - 1) / kStackRows) * G > 256)
        hipLaunchKernelGGL((mlp3_fwd_kernel<2>), dim3((M + 2 * kStackRows - 1) / (2 * kStackRows), G), dim3(1024), 0,
                           (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL((mlp3_fwd_kernel<1>), dim3((M + kStackRows - 1) / kStackRows, G), dim3(1024), 0,
                           (hipStream_t)stream, a);
    return check_launch();
}

// Every stack of the group takes the path rrl_mlp3_forward would take for it on its own (so the results are the
// stand-alone launches', bit for bit); the group must be homogeneous: all split (scratch given, partial sums left in
// scratch = finalize 0) or all on the same non-split tiling.
#ifdef RRL_FWD_TIMING
int rrl_debug_fwd_stamps(unsigned long long* host, int n_blocks) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(rrl_fwd_stamps), sizeof(unsigned long long) * 8 * n_blocks) == hipSuccess
               ? RRL_OK : RRL_ELAUNCH;
}
#endif

static int build_stack_group(int n, const rrl_stack_t* st, StackGroup& sg, int& path, int big_r = kBigR, int small_r = 1) {
    if (!st || n <= 0 || n > kMaxGroup) return RRL_EINVAL;
    sg = StackGroup{};
    sg.n = n;
    sg.first[0] = 0;
    path = -1;   // 0 split (small batch), 3 split (R = kBigR row tiles), 4 split, both kinds; 1 plain R = 1, 2 plain R = 2
    for (int k = 0; k < n; ++k) {
        const rrl_stack_t& p = st[k];
        const int rc = stack_check(p.G, p.M, p.H, p.din, p.dout, p.x, p.W1, p.b1, p.W2, p.b2, p.W3, p.b3, p.out);
        if (rc != RRL_OK) return rc;
        sg.a[k] = StackArgs{p.x, p.W1, p.b1, p.W2, p.b2, p.W3, p.b3, p.h1, p.h2, p.out, p.M, p.H, p.din, p.dout, p.ldx,
                            p.in_head, p.use_in_head};
        if (p.use_in_head) {
            const rrl_policy_head_t& h = p.in_head;
            if (p.din != 4 || !h.head || !h.scale || !h.bias || h.n_part <= 0 || h.n_part > 4 ||
                (h.kind == RRL_HEAD_GAUSS ? !h.eps : (h.kind != RRL_HEAD_STOCH || !h.log_std)) || (h.obs_out && !h.action))
                return RRL_EINVAL;
            if (!(p.scratch && rrl_mlp3_is_split(p.M, p.H))) return RRL_EINVAL;   // the column-split kernels only
        }
        sg.partial[k] = p.scratch;
        sg.G[k] = p.G;
        int my;
        const long long tiles16 = (p.M + kStackRows - 1) / kStackRows;
        if (p.scratch && rrl_mlp3_is_split(p.M, p.H)) {
            my = (p.M <= kSplitSmallM || p.H != 256) ? 0 : 3;      // the multi-row tiles are built for H = 256
            if (my == 0 && small_r > 1 && p.H != 256) return RRL_EINVAL;
            const int rows = (my == 0 ? small_r : big_r) * kStackRows;
            sg.tiles[k] = (p.M + rows - 1) / rows;
            sg.big[k] = my == 3;
            sg.first[k + 1] = sg.first[k] + sg.tiles[k] * p.G * kSplit;
        } else if (tiles16 * p.G > 256) {
            my = 2;
            sg.tiles[k] = (p.M + 2 * kStackRows - 1) / (2 * kStackRows);
            sg.first[k + 1] = sg.first[k] + sg.tiles[k] * p.G;
        } else {
            my = 1;
            sg.tiles[k] = int(tiles16);
            sg.first[k + 1] = sg.first[k] + sg.tiles[k] * p.G;
        }
        const bool split_mix = (my == 0 || my == 3) && (path == 0 || path == 3 || path == 4);
        if (path >= 0 && my != path && !split_mix) return RRL_EINVAL;
        path = (path >= 0 && my != path) ? 4 : my;
    }
    if (path == 4 && big_r != kBigR && small_r != big_r) return RRL_EINVAL;      // the mixed kernel is built for kBigR
    for (int k = n; k < kMaxGroup; ++k) sg.first[k + 1] = sg.first[n];
    return RRL_OK;
}

int rrl_mlp3_forward_multi(int n, const rrl_stack_t* st, void* stream) {
    StackGroup sg;
    int path;
    const int rc = build_stack_group(n, st, sg, path);
    if (rc != RRL_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (path == 0) {
        hipLaunchKernelGGL(mlp3_fwd_split_group_kernel<1>, dim3(sg.first[n]), dim3(256), split_lds_floats(1) * 4, s, sg);
    } else if (path == 3) {
        static const bool ok = grant_lds((const void*)mlp3_fwd_split_group_kernel<kBigR>, split_lds_floats(kBigR) * 4);
        if (!ok) return RRL_ERANGE;
        hipLaunchKernelGGL(mlp3_fwd_split_group_kernel<kBigR>, dim3(sg.first[n]), dim3(256),
                           split_lds_floats(kBigR) * 4, s, sg);
    } else if (path == 4) {
        hipLaunchKernelGGL(mlp3_fwd_split_mixed_kernel, dim3(sg.first[n]), dim3(256), split_lds_floats(kBigR) * 4, s, sg);
    } else if (path == 1) hipLaunchKernelGGL((mlp3_fwd_group_kernel<1>), dim3(sg.first[n]), dim3(1024), 0, s, sg);
    else hipLaunchKernelGGL((mlp3_fwd_group_kernel<2>), dim3(sg.first[n]), dim3(1024), 0, s, sg);
    return check_launch();
}

// the column-split kernels only (what the steady-state iteration launches at H = 256); every seed on the same path
int rrl_mlp3_forward_multi_packed(int S, const int* n, const rrl_stack_t* const* members, void* stream) {
    rrl_pack::Key key;
    if (!pack_key(2, S, n, members, key)) return RRL_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    rrl_pack::Plan* plan = rrl_pack::lookup(key);
    if (!plan) {
        std::vector<StackGroup> groups;
        rrl_pack::Idx ix;
        int path = -1;
        // row tiles per workgroup of the large-batch kernel: with several seeds in the launch there are workgroups to spare,
        // so each keeps its W2 fragments for kPackR = 4 row tiles (half the weight stream of the solo kernel's 2; per output
        // element the arithmetic is the same for every R)
        static const int r4_min = pack_threshold("RRL_PACK_R4_MIN_SEEDS", kPackMinSeeds);
        const int big_r = S >= r4_min ? kPackR : kBigR;
        // small batches (the updates' B = 256 forwards): kBigR row tiles per workgroup from pack_small_r2_min_seeds() seeds
        // on, when every member has the hidden width the multi-row tiles are built for
        bool all256 = true;
        for (int s = 0; s < S; ++s)
            for (int k = 0; k < n[s]; ++k) all256 = all256 && members[s] && members[s][k].H == 256;
        const int small_r = (S >= pack_small_r2_min_seeds() && all256) ? big_r : 1;
        const int rc = build_pack<StackGroup>(S, n, members, groups, ix, [&](int nk, const rrl_stack_t* m, StackGroup& g) {
            int my;
            const int r = build_stack_group(nk, m, g, my, big_r, small_r);
            if (r != RRL_OK) return r;
            if ((my != 0 && my != 3 && my != 4) || (path >= 0 && my != path)) return int(RRL_EINVAL);
            path = my;
            return int(RRL_OK);
        });
        if (rc != RRL_OK) return rc;
        if (path == 3) {
            static const bool ok = grant_lds((const void*)mlp3_fwd_split_pack_kernel<kBigR>, split_lds_floats(kBigR) * 4) &&
                                   grant_lds((const void*)mlp3_fwd_split_pack_kernel<kPackR>, split_lds_floats(kPackR) * 4);
            if (!ok) return RRL_ERANGE;
        }
        plan = rrl_pack::store(key, groups.data(), sizeof(StackGroup) * S, st);
        if (!plan) return RRL_ELAUNCH;
        plan->grid = rrl_pack::finish(ix);
        plan->ix = ix;
        // small members on multi-row tiles run the large-batch kernel (path 3); a mix of small and large members (path 4)
        // then has ONE tile shape as well
        if (small_r > 1 && (path == 0 || path == 4)) {
            static const bool ok = grant_lds((const void*)mlp3_fwd_split_pack_kernel<kBigR>, split_lds_floats(kBigR) * 4) &&
                                   grant_lds((const void*)mlp3_fwd_split_pack_kernel<kPackR>, split_lds_floats(kPackR) * 4);
            if (!ok) return RRL_ERANGE;
            path = 3;
        }
        plan->i0 = path;
        plan->i1 = big_r;
    }
    if (plan->i0 == 0)
        hipLaunchKernelGGL(mlp3_fwd_split_pack_kernel<1>, dim3(plan->grid), dim3(256), split_lds_floats(1) * 4, st,
                           (const StackGroup*)plan->dev, plan->ix);
    else if (plan->i0 == 4)
        hipLaunchKernelGGL(mlp3_fwd_split_mixed_pack_kernel, dim3(plan->grid), dim3(256), split_lds_floats(kBigR) * 4, st,
                           (const StackGroup*)plan->dev, plan->ix);
    else if (plan->i1 == kPackR)
        hipLaunchKernelGGL(mlp3_fwd_split_pack_kernel<kPackR>, dim3(plan->grid), dim3(256),
                           split_lds_floats(kPackR) * 4, st, (const StackGroup*)plan->dev, plan->ix);
    else
        hipLaunchKernelGGL(mlp3_fwd_split_pack_kernel<kBigR>, dim3(plan->grid), dim3(256),
                           split_lds_floats(kBigR) * 4, st, (const StackGroup*)plan->dev, plan->ix);
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
    if (!la || !la->out || !h2 || !W3) return RRL_EINVAL;          // dh2 == NULL: the hidden-layer tiles generate it
    if (G <= 0 || B <= 0 || B > 1024 || H <= 0 || dout <= 0 || dout > 4) return RRL_ERANGE;
    if (la->kind != kPlainDOut) {
        if (la->kind < 0 || la->kind > RRL_LOSS_STOCH_HEAD || la->n_part <= 0 || la->n_part > 4) return RRL_ERANGE;
        const int heads = la->kind <= RRL_LOSS_QRISK_POLICY ? 2 : 1;
        const int width = la->kind <= RRL_LOSS_QRISK_POLICY ? 1 : (la->kind == RRL_LOSS_GAUSS_HEAD ? 4 : 2);
        if (G != heads || dout != width) return RRL_EINVAL;
        if (la->da_parts < 0 || la->da_parts > 16) return RRL_ERANGE;
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

int rrl_mlp_head_backward_multi(int n, const rrl_head_bwd_t* ps, void* stream) {
    HeadBwdGroup hg;
    const int rc = build_head_group(n, ps, hg);
    if (rc != RRL_OK) return rc;
    hipLaunchKernelGGL(head_bwd_group_kernel, dim3(hg.first[n]), dim3(256), 0, (hipStream_t)stream, hg);
    return check_launch();
}

int rrl_mlp_head_backward_multi_packed(int S, const int* n, const rrl_head_bwd_t* const* members, void* stream) {
    rrl_pack::Key key;
    if (!pack_key(3, S, n, members, key)) return RRL_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    rrl_pack::Plan* plan = rrl_pack::lookup(key);
    if (!plan) {
        std::vector<HeadBwdGroup> groups;
        rrl_pack::Idx ix;
        const int rc = build_pack<HeadBwdGroup>(S, n, members, groups, ix, build_head_group);
        if (rc != RRL_OK) return rc;
        plan = rrl_pack::store(key, groups.data(), sizeof(HeadBwdGroup) * S, st);
        if (!plan) return RRL_ELAUNCH;
        plan->grid = rrl_pack::finish(ix);
        plan->ix = ix;
    }
    hipLaunchKernelGGL(head_bwd_pack_kernel, dim3(plan->grid), dim3(256), 0, st, (const HeadBwdGroup*)plan->dev, plan->ix);
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
