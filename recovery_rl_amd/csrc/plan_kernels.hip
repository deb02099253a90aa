// plan_kernels.hip -- the candidate-evaluation inner loop of the model-based recovery controller
// (MPC._compile_cost, recovery_rl/MPC.py:374-416, with _predict_next_obs :421-439, the PETS ensemble
// forward config/navigation1.py:71-96 and QRiskWrapper.get_value recovery_rl/qrisk.py:184-196) as ONE
// MFMA kernel for gfx950.
//
// For every (env m, candidate c, particle p) row the reference runs plan_hor steps of
//     cost += max(sigmoid Q1, sigmoid Q2)(obs, ac_t);  obs += mean_e(obs, ac_t) + N(0,1) * sqrt(var_e(obs, ac_t))
// through two 256-wide MLPs and one 200-wide ensemble member, materialising every [rows, 256] activation
// in HBM between layers (26 GB per layer at 4096 envs).  Here a workgroup owns 64 rows for the whole
// rollout: activations live in LDS, weights stream from L2 as pre-packed MFMA fragments, and only the
// per-candidate cost leaves the chip.  Arithmetic is v_mfma_f32_16x16x4_f32 (exact f32 fma chains), so
// results differ from the PyTorch path only by summation order.
//
// Only the work the costs depend on is done (round 6; 12.90 of the 17.22 GFLOP per env and CEM iteration the literal
// loop spends): at t = 0 every particle of a candidate sees the same (cur_obs, ac_0), so plan_first_step_kernel evaluates
// Q_risk once per candidate and each member once per (candidate, member); the prediction of the last step feeds a
// cur_obs nothing reads, so plan_cost_kernel ends with a Q_risk-only step.  Rows of an MFMA tile are independent and both
// kernels run the same phase code: the costs are bit-identical to the literal rollout's.
//
// Row order inside a tile: 16 consecutive (m, c) groups x the 4 particles that are bound to ensemble member
// e (TS-infinity: member = particle / (npart / nets), MPC.py:441-455); tile id = (group block, e).
#include "rrl_device.hpp"
#include "rrl_host.hpp"
#include "plan_layout.hpp"

using namespace rrl_host;
using namespace rrl_plan;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// waves per SIMD the register allocator targets: 2 = one workgroup per CU (<= 256 VGPRs), 4 = two (<= 128)
// (the timing ablations of rounds 2-4 -- no weight loads / no LDS reads / no stores / one product / no tail -- are
// profiles/patches/plan_ablate.patch, not product code)
#ifndef RRL_PLAN_WAVES_PER_EU
#define RRL_PLAN_WAVES_PER_EU 4
#endif

constexpr int kRows = 64;                 // rows per workgroup
constexpr int kWaves = 8;
constexpr int kThreads = kWaves * 64;
constexpr int kActStride = kHQ + 4;       // +4 floats: row r starts at bank 4r, ds_read_b128 conflict-free
// F16X3 mode keeps the activations as two f16 planes (hi, lo) of [64][kHalfStride] instead: 132 words per row, so row r
// starts at bank 4r again and the 8-byte fragment reads of a half-wave (16 rows x 2 k-groups) hit 64 different banks
constexpr int kHalfStride = kHQ + 8;
constexpr int kActFloats = kRows * kHalfStride;       // 2 planes x 64 x 264 x 2 B = 66 KB (f32 layout: 65 KB)
static_assert(kActFloats >= kRows * kActStride, "the f32 activation tile must fit the same region");
constexpr int kLdsFloats = kActFloats + 3 * kRows * 4 + 2 * kWaves * kRows + 4 * kRows * 4;
constexpr int kLdsBytes = kLdsFloats * 4;  // 78.5 KB: two workgroups per CU

__device__ __forceinline__ f32x4 mfma(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// v = hi + lo with hi = f16(v) and lo = f16(v - hi): v - hi is exact in f32 (13 significant bits at most), so for
// |v| >= 2^-3 hi + lo carries 22 bits of v and three products hi*hi + hi*lo + lo*hi carry ~2^-21 of the f32 product (the
// lo*lo term, 2^-22 of it, is dropped).  lo is stored UNSCALED: for |v| below ~0.06 it falls into the f16 subnormal range
// (spacing 2^-24), where hi + lo keeps an ABSOLUTE error <= 2^-25 = 3e-8 instead of a relative 2^-22 (14-17 bits of a
// small value) -- an absolute bound that is below the f32 rounding of the O(1) sums these values enter, and what the
// 2e-5 agreement with the f32 kernel (tests/test_plan_gpu.py) rests on; the CDNA4 f16 MFMA does not flush subnormal
// inputs.  Activations beyond the f16 range saturate (NaN stays NaN and becomes the reference's 1e6 cost).
__device__ __forceinline__ void split16(float v, _Float16& hi, _Float16& lo) {
    v = v > 65504.f ? 65504.f : v;      // activations are relu / swish outputs: bounded below (NaN compares false: kept)
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
}

// Identity the optimiser cannot see through: address arithmetic derived from opaque(lane) is redone per phase
// instead of being hoisted out of the rollout loop and kept live (that hoisting costs > 100 VGPRs here).
__device__ __forceinline__ int opaque(int x) {
    asm volatile("" : "+v"(x));
    return x;
}

__device__ __forceinline__ float reluf(float x) { return x < 0.f ? 0.f : x; }   // NaN stays NaN (F.relu)
// v_rcp_f32 (1 ulp) instead of the IEEE division sequence (~10 instructions): swish runs once per hidden activation of
// the ensemble, 40 k times per workgroup and rollout step
__device__ __forceinline__ float sigmoidf(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float swishf(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float softplusf(float x) { return x > 20.0f ? x : log1pf(expf(x)); }

// acc[r][c] += act[rt[r] tile, :K] * W[:, ct[c] tile] over J chunks of 16 k.  A fragments come from LDS
// (row-major, k contiguous), B fragments from the packed weight stream ([ct][j][lane] float4, one coalesced
// 1 KB load per fragment).  Chunk step t uses k = 16 j + 4 (lane / 16) + t on both operands.  The loads of
// Tile (r, c) of a wave's block is computed when c < 3 or r == XR: columns 0..2 for every row tile, the 4th
// column (only the ensemble has one: its 13th column tile) for ONE row tile.
template <int XR>
__device__ __forceinline__ constexpr bool tile_on(int r, int c) {
    return c < 3 || r == XR;
}

template <int NCV, int NC>
__device__ __forceinline__ void load_b(f32x4 (&b)[NC], const float* __restrict__ wpk, const int (&ct)[NC], int J,
                                       int j, int lane) {
#pragma unroll
    for (int c = 0; c < NCV; ++c) {
        b[c] = *reinterpret_cast<const f32x4*>(wpk + ((size_t)(ct[c] * J + j) * 64 + lane) * 4);
    }
}

template <int NCV, int XR, int MR, int NC>
__device__ __forceinline__ void mma_chunk(f32x4 (&acc)[MR][NC], const float* act, const int (&rt)[MR],
                                          const f32x4 (&b)[NC], int j, int lane) {
    f32x4 a[MR];
#pragma unroll
    for (int r = 0; r < MR; ++r) {
        a[r] = *reinterpret_cast<const f32x4*>(act + (rt[r] * 16 + (lane & 15)) * kActStride + 16 * j +
                                               (lane >> 4) * 4);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < MR; ++r)
#pragma unroll
            for (int c = 0; c < NCV; ++c)
                if (tile_on<XR>(r, c)) acc[r][c] = mfma(a[r][t], b[c][t], acc[r][c]);
}

// F16X3 with the K = 32 shape (v_mfma_f32_16x16x32_f16: lane l holds k = 8 (l / 16) + 0..7 of a 32-wide block on both
// operands; the same 16 cycles as the K = 16 shape for twice the k).  A: one 16-byte read per plane; B: the two float4
// slots of block jp hold {hi[8]} and {lo[8]} of this lane's column (pack_layer_f16x3_k32_kernel).
// Fragment sets in flight: the next block's hi set is requested before this block's MFMAs; the lo set has ONE buffer --
// the hi*lo products go first, and the next block's lo set is requested into the same registers as soon as they are
// issued (3 float4 sets live instead of 4: what fits next to 32 accumulator registers when a wave owns four column tiles).
template <int NCV, int NB, int XR = -1, int MR, int NC>
__device__ __forceinline__ void layer_mma_k32(f32x4 (&acc)[MR][NC], const float* act, const int (&rt)[MR],
                                              const float* __restrict__ wpk, const int (&ct)[NC], int lane) {
    constexpr int J = 2 * NB;          // float4 slots per column tile
    const _Float16* ah = reinterpret_cast<const _Float16*>(act);
    const _Float16* al = ah + kRows * kHalfStride;
    f32x4 h0[NC], h1[NC], l[NC];
    const auto block = [&](const f32x4 (&bh)[NC], int jp, bool more) {
        f16x8 ahi[MR], alo[MR];
#pragma unroll
        for (int r = 0; r < MR; ++r) {
            const int off = (rt[r] * 16 + (lane & 15)) * kHalfStride + 32 * jp + (lane >> 4) * 8;
            ahi[r] = *reinterpret_cast<const f16x8*>(ah + off);
            alo[r] = *reinterpret_cast<const f16x8*>(al + off);
        }
#pragma unroll
        for (int r = 0; r < MR; ++r)
#pragma unroll
            for (int c = 0; c < NCV; ++c)
                if (tile_on<XR>(r, c))
                    acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi[r], __builtin_bit_cast(f16x8, l[c]), acc[r][c], 0, 0, 0);
        if (more) load_b<NCV>(l, wpk, ct, J, 2 * jp + 3, lane);
#pragma unroll
        for (int prod = 0; prod < 2; ++prod)
#pragma unroll
            for (int r = 0; r < MR; ++r)
#pragma unroll
                for (int c = 0; c < NCV; ++c)
                    if (tile_on<XR>(r, c))
                        acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(prod ? alo[r] : ahi[r],
                                                                           __builtin_bit_cast(f16x8, bh[c]), acc[r][c], 0, 0, 0);
    };
    load_b<NCV>(h0, wpk, ct, J, 0, lane);
    load_b<NCV>(l, wpk, ct, J, 1, lane);
    int jp = 0;
#pragma unroll 1
    for (; jp + 2 < NB; jp += 2) {
        load_b<NCV>(h1, wpk, ct, J, 2 * jp + 2, lane);
        __builtin_amdgcn_sched_barrier(0);
        block(h0, jp, true);
        load_b<NCV>(h0, wpk, ct, J, 2 * jp + 4, lane);
        __builtin_amdgcn_sched_barrier(0);
        block(h1, jp + 1, true);
    }
    if constexpr (NB % 2 == 1) {
        block(h0, NB - 1, false);
    } else {
        load_b<NCV>(h1, wpk, ct, J, 2 * NB - 2, lane);
        block(h0, NB - 2, true);
        block(h1, NB - 1, false);
    }
}

// The weight fragments of chunk j + 1 are requested before the 4 MR NC MFMAs of chunk j (register double
// buffer): their L2 latency hides behind ~1000 cycles of matrix work.  LDS fragments are read per chunk.
template <int NCV, int J, int XR = -1, int MR, int NC>
__device__ __forceinline__ void layer_mma(f32x4 (&acc)[MR][NC], const float* act, const int (&rt)[MR],
                                          const float* __restrict__ wpk, const int (&ct)[NC], int lane) {
    f32x4 b0[NC], b1[NC];
    load_b<NCV>(b0, wpk, ct, J, 0, lane);
    int j = 0;
    // steady state without conditionals: the compiler's vmcnt tracking then waits for the OLDER fragment set
    // only, leaving the prefetch of the next chunk in flight during the 4 MR NCV MFMAs
#pragma unroll 1
    for (; j + 2 < J; j += 2) {
        load_b<NCV>(b1, wpk, ct, J, j + 1, lane);
        __builtin_amdgcn_sched_barrier(0);      // keep the prefetch ahead of the MFMAs it hides behind
        mma_chunk<NCV, XR>(acc, act, rt, b0, j, lane);
        load_b<NCV>(b0, wpk, ct, J, j + 2, lane);
        __builtin_amdgcn_sched_barrier(0);
        mma_chunk<NCV, XR>(acc, act, rt, b1, j + 1, lane);
    }
    if constexpr (J % 2 == 0) {
        load_b<NCV>(b1, wpk, ct, J, J - 1, lane);
        mma_chunk<NCV, XR>(acc, act, rt, b0, J - 2, lane);
        mma_chunk<NCV, XR>(acc, act, rt, b1, J - 1, lane);
    } else {
        mma_chunk<NCV, XR>(acc, act, rt, b0, J - 1, lane);
    }
}

// first layers: K = 4 inputs = ONE mfma per tile.  x is [64][4] in LDS, w1 packed [ct][lane].
template <int NCV, int XR = -1, int MR, int NC>
__device__ __forceinline__ void input_mma(f32x4 (&acc)[MR][NC], const float* x, const int (&rt)[MR],
                                          const float* __restrict__ w1, const int (&ct)[NC], int lane) {
#pragma unroll
    for (int r = 0; r < MR; ++r) {
        const float a = x[(rt[r] * 16 + (lane & 15)) * 4 + (lane >> 4)];
#pragma unroll
        for (int c = 0; c < NCV; ++c)
            if (tile_on<XR>(r, c)) acc[r][c] = mfma(a, w1[ct[c] * 64 + lane], acc[r][c]);
    }
}

template <int MR, int NC>
__device__ __forceinline__ void zero(f32x4 (&acc)[MR][NC]) {
#pragma unroll
    for (int r = 0; r < MR; ++r)
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// act[row][col] = f(acc + bias[col]); C layout: row = 16 rt + 4 (lane / 16) + i, col = 16 ct + lane % 16
template <bool F16X3, bool SWISH, int NCV, int XR = -1, int MR, int NC>
__device__ __forceinline__ void store_act(const f32x4 (&acc)[MR][NC], float* act, const int (&rt)[MR],
                                          const float (&bias)[NC], const int (&ct)[NC], int lane) {
    _Float16* ah = reinterpret_cast<_Float16*>(act);
    _Float16* al = ah + kRows * kHalfStride;
#pragma unroll
    for (int c = 0; c < NCV; ++c) {
        const int col = ct[c] * 16 + (lane & 15);
        const float bv = bias[c];
#pragma unroll
        for (int r = 0; r < MR; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (!tile_on<XR>(r, c)) continue;
                const float v = acc[r][c][i] + bv;
                const float f = SWISH ? swishf(v) : reluf(v);
                const int row = rt[r] * 16 + 4 * (lane >> 4) + i;
                if constexpr (F16X3) {
                    _Float16 hi, lo;
                    split16(f, hi, lo);
                    ah[row * kHalfStride + col] = hi;
                    al[row * kHalfStride + col] = lo;
                } else {
                    act[row * kActStride + col] = f;
                }
            }
    }
}

__device__ __forceinline__ float reduce16(float v) {   // sum over the 16 lanes that share lane / 16
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    return v;
}

// LDS carve-up shared by the two kernels
struct PlanLds {
    float* act;                        // [64][260] f32 activations, or two f16 planes
    float* xs;                         // [64][4] raw (obs, ac)
    float* xn;                         // [64][4] standardised ensemble input
    float (*qpart)[kWaves][kRows];     // [2 heads][8 waves][64 rows] partial last-layer sums of Q_risk
    float (*epart)[kRows][4];          // [4 column strips][64 rows][4 outputs] partial last-layer sums of the member
    float* rowstate;                   // [64][4] = {obs x, obs y, cost, -}
};
__device__ __forceinline__ PlanLds carve(float* lds) {
    PlanLds L;
    L.act = lds;
    L.xs = L.act + kActFloats;
    L.xn = L.xs + kRows * 4;
    L.qpart = reinterpret_cast<float(*)[kWaves][kRows]>(L.xn + kRows * 4);
    L.epart = reinterpret_cast<float(*)[kRows][4]>(L.xn + kRows * 4 + 2 * kWaves * kRows);
    L.rowstate = L.xn + kRows * 4 + 2 * kWaves * kRows + 4 * kRows * 4;
    return L;
}

// ---- Q_risk twin heads on the 64 rows of xs: 4 -> HQ relu -> HQ relu -> 1; leaves qpart[h][wave][row] ----
// (pre-activation of the output = b3 + sum over the 8 waves, added by the caller in wave order).  Ends with a barrier.
template <bool F16X3>
__device__ __forceinline__ void q_phase(const PlanLds& L, const float* __restrict__ pk, int wave, int lane) {
    const int ln = opaque(lane);
    const int q_rt[4] = {0, 1, 2, 3};
    const int q_ct[2] = {2 * wave, 2 * wave + 1};
    float* act = L.act;
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
        const float* __restrict__ w = pk + h * kQSize;
        // epilogue constants first: their latency hides behind the matrix work
        float b1v[2], b2v[2], w3v[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int col = q_ct[c] * 16 + (ln & 15);
            b1v[c] = w[kQB1 + col];
            b2v[c] = w[kQB2 + col];
            w3v[c] = w[kQW3 + col];
        }
        f32x4 acc[4][2];
        zero(acc);
        input_mma<2>(acc, L.xs, q_rt, w + kQW1, q_ct, opaque(lane));
        store_act<F16X3, false, 2>(acc, act, q_rt, b1v, q_ct, opaque(lane));
        __syncthreads();
        zero(acc);
        if constexpr (F16X3) layer_mma_k32<2, kQTiles / 2>(acc, act, q_rt, w + kQW2, q_ct, opaque(lane));
        else layer_mma<2, kQTiles>(acc, act, q_rt, w + kQW2, q_ct, opaque(lane));
        // last layer folded in: q[row] = sum_col relu(h2 + b2) w3[col]
        float s[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) s[r][i] = 0.f;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float bv = b2v[c], w3 = w3v[c];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) s[r][i] += reluf(acc[r][c][i] + bv) * w3;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float v = reduce16(s[r][i]);
                if ((ln & 15) == 0) L.qpart[h][wave][r * 16 + 4 * (ln >> 4) + i] = v;
            }
        __syncthreads();       // act is free again; qpart[h] complete
    }
}

// max(sigmoid Q1, sigmoid Q2) of row `row` from the partial sums q_phase left (qrisk.py:184-196; torch.max: NaN propagates)
__device__ __forceinline__ float q_value(const PlanLds& L, const float* __restrict__ pk, int row) {
    float q[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        float v = pk[h * kQSize + kQB3];
#pragma unroll
        for (int w = 0; w < kWaves; ++w) v += L.qpart[h][w][row];
        q[h] = sigmoidf(v);
    }
    return (q[0] > q[1] || q[0] != q[0]) ? q[0] : q[1];
}

// ---- ensemble member (weights at epk) on the 64 rows of xn: 4 -> 200 swish -> 200 swish -> 200 swish -> 4; leaves
// epart[strip][row][o] (output o = b3[o] + the four strips, added by the caller in a fixed order).  Ends with a barrier.
// 4 row tiles x 13 column tiles: every wave owns 2 row tiles x 3 column tiles (cs, cs+4, cs+8); the 13th column's four
// tiles go one each to waves 0..3 (row tile xr of their own pair), so every SIMD carries the same number of MFMAs.
template <bool F16X3>
__device__ __forceinline__ void e_phase(const PlanLds& L, const float* __restrict__ epk, int wave, int lane, int tid) {
    const int ln = opaque(lane);
    const int rh = wave & 1, cs = wave >> 1;
    const int e_rt[2] = {2 * rh, 2 * rh + 1};
    const int e_ct[4] = {cs, cs + 4, cs + 8, wave < 4 ? 12 : cs};   // the 13th column: waves 0..3, one row tile each
    const int xr = wave < 4 ? (wave >> 1) : -1;                     // which of the wave's two row tiles (0/1), or none
    float* act = L.act;
    // Wave-uniform dispatch on the extra tile: E_STAGE(f, args) = f<NCV, ..., XR>(args)
#define E_STAGE(CALL4X0, CALL4X1, CALL3)  \
    if (xr == 0) {                        \
        CALL4X0;                          \
    } else if (xr == 1) {                 \
        CALL4X1;                          \
    } else {                              \
        CALL3;                            \
    }
    // column of this lane in column tile c, re-derived at every use (not kept live across the phases)
    auto ecol = [&](int c) { return e_ct[c] * 16 + (opaque(lane) & 15); };
    // per-column constants are requested one phase ahead of their use, never all live at once
    float eb[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) eb[c] = epk[kEB0 + ecol(c)];
    if constexpr (F16X3) {
        // the ensemble's last 32-wide k block covers columns 192..223, its stores only 0..207: the rest
        // multiplies zero weights, which needs them FINITE -- the Q_risk phase leaves its own activations
        // there, and a NaN / inf among them (a diverged safety critic) would leak into this member's
        // prediction as NaN x 0; re-zeroed before every ensemble phase (nobody reads `act` right now: the
        // previous phase ended with a barrier, the layer-0 stores below touch columns < 208 only)
        _Float16* planes = reinterpret_cast<_Float16*>(act);
        for (int i = opaque(tid); i < 2 * kRows * 16; i += kThreads)
            planes[(i >> 10) * (kRows * kHalfStride) + ((i >> 4) & (kRows - 1)) * kHalfStride + kHEPad + (i & 15)] =
                (_Float16)0.f;
    }
    f32x4 acc[2][4];
    zero(acc);
    E_STAGE((input_mma<4, 0>(acc, L.xn, e_rt, epk + kEW0, e_ct, opaque(lane)),
             store_act<F16X3, true, 4, 0>(acc, act, e_rt, eb, e_ct, opaque(lane))),
            (input_mma<4, 1>(acc, L.xn, e_rt, epk + kEW0, e_ct, opaque(lane)),
             store_act<F16X3, true, 4, 1>(acc, act, e_rt, eb, e_ct, opaque(lane))),
            (input_mma<3>(acc, L.xn, e_rt, epk + kEW0, e_ct, opaque(lane)),
             store_act<F16X3, true, 3>(acc, act, e_rt, eb, e_ct, opaque(lane))))
#pragma unroll
    for (int c = 0; c < 4; ++c) eb[c] = epk[kEB1 + ecol(c)];
    __syncthreads();
    zero(acc);
    if constexpr (F16X3) {
        E_STAGE((layer_mma_k32<4, kEBlocks32, 0>(acc, act, e_rt, epk + kEW1, e_ct, opaque(lane))),
                (layer_mma_k32<4, kEBlocks32, 1>(acc, act, e_rt, epk + kEW1, e_ct, opaque(lane))),
                (layer_mma_k32<3, kEBlocks32>(acc, act, e_rt, epk + kEW1, e_ct, opaque(lane))))
    } else {
        E_STAGE((layer_mma<4, kETiles, 0>(acc, act, e_rt, epk + kEW1, e_ct, opaque(lane))),
                (layer_mma<4, kETiles, 1>(acc, act, e_rt, epk + kEW1, e_ct, opaque(lane))),
                (layer_mma<3, kETiles>(acc, act, e_rt, epk + kEW1, e_ct, opaque(lane))))
    }
    __syncthreads();       // every wave has finished reading layer-1 input
    E_STAGE((store_act<F16X3, true, 4, 0>(acc, act, e_rt, eb, e_ct, opaque(lane))),
            (store_act<F16X3, true, 4, 1>(acc, act, e_rt, eb, e_ct, opaque(lane))),
            (store_act<F16X3, true, 3>(acc, act, e_rt, eb, e_ct, opaque(lane))))
#pragma unroll
    for (int c = 0; c < 4; ++c) eb[c] = epk[kEB2 + ecol(c)];
    __syncthreads();
    zero(acc);
    if constexpr (F16X3) {
        E_STAGE((layer_mma_k32<4, kEBlocks32, 0>(acc, act, e_rt, epk + kEW2, e_ct, opaque(lane))),
                (layer_mma_k32<4, kEBlocks32, 1>(acc, act, e_rt, epk + kEW2, e_ct, opaque(lane))),
                (layer_mma_k32<3, kEBlocks32>(acc, act, e_rt, epk + kEW2, e_ct, opaque(lane))))
    } else {
        E_STAGE((layer_mma<4, kETiles, 0>(acc, act, e_rt, epk + kEW2, e_ct, opaque(lane))),
                (layer_mma<4, kETiles, 1>(acc, act, e_rt, epk + kEW2, e_ct, opaque(lane))),
                (layer_mma<3, kETiles>(acc, act, e_rt, epk + kEW2, e_ct, opaque(lane))))
    }
#undef E_STAGE
    // last layer (200 -> 4) folded in: out[row][o] = sum_col swish(h3 + b2) W3[col][o]
    f32x4 ew3[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) ew3[c] = *reinterpret_cast<const f32x4*>(epk + kEW3 + ecol(c) * 4);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        float s[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int o = 0; o < 4; ++o) s[i][o] = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c == 3 && xr != r) continue;          // acc[r][3] is zero there, but swish(b) is not
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float v = swishf(acc[r][c][i] + eb[c]);
#pragma unroll
                for (int o = 0; o < 4; ++o) s[i][o] += v * ew3[c][o];
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                const float v = reduce16(s[i][o]);
                if ((ln & 15) == 0) L.epart[cs][e_rt[r] * 16 + 4 * (ln >> 4) + i][o] = v;
            }
    }
    __syncthreads();
}

// The member's predictive distribution for row `row` from the partial sums e_phase left: {mean dx, mean dy, sd x, sd y}
// (config/navigation1.py:90-96: logvar soft-clamped between min_logvar and max_logvar; sd = sqrt(exp(logvar)))
__device__ __forceinline__ f32x4 e_value(const PlanLds& L, const float* __restrict__ epk, const float* __restrict__ g,
                                         int row) {
    float out[4];
#pragma unroll
    for (int o = 0; o < 4; ++o)
        out[o] = epk[kEB3 + o] +
                 ((L.epart[0][row][o] + L.epart[1][row][o]) + (L.epart[2][row][o] + L.epart[3][row][o]));
    f32x4 d = {out[0], out[1], 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        float lv = out[2 + k];
        lv = g[8 + k] - softplusf(g[8 + k] - lv);
        lv = g[10 + k] + softplusf(lv - g[10 + k]);
        d[2 + k] = sqrtf(expf(lv));
    }
    return d;
}

__device__ __forceinline__ void write_inputs(const PlanLds& L, const float* __restrict__ g, int row, f32x4 x) {
    *reinterpret_cast<f32x4*>(L.xs + row * 4) = x;
    f32x4 n;
#pragma unroll
    for (int k = 0; k < 4; ++k) n[k] = (x[k] - g[k]) / g[4 + k];
    *reinterpret_cast<f32x4*>(L.xn + row * 4) = n;
}

// First step of the rollout, once per DISTINCT row.  At t = 0 the npart particles of a candidate share
// (cur_obs, ac_0) (MPC.py:393-402: cur_obs expanded over nopt * npart rows, ac_seqs tiled over the particles), so
// Q_risk(cur_obs, ac_0) is one value per candidate and the member's predictive distribution one per (candidate, member);
// only the noise draw differs between the particles.  A workgroup takes 64 consecutive candidates ("groups") through ONE
// network: blockIdx = (group block, net), net < n_nets = ensemble member, net == n_nets = the twin Q_risk.  MFMA rows are
// independent and the phases are the rollout kernel's own code, so every value equals, bit for bit, what each of the
// particle rows would have computed -- given ONE thing: the member's last layer adds its 13th column tile into strip 0
// for even row tiles and into strip 1 for odd ones (e_phase), so a row's sum order depends on the parity of its row tile.
// In the rollout kernel candidate j of a 16-candidate block sits in row tile j / 4; here candidate j of the 64-candidate
// block goes to row (first_step_row) whose row tile has the same parity, bit 2 of j.
//   q0[group] = max(sigmoid Q1, sigmoid Q2)(cur_obs, ac_0);  e0[group][e] = {mean dx, mean dy, sd x, sd y}
__device__ __forceinline__ int first_step_row(int j) {       // bits j5 j4 j3 j2 j1 j0 -> row j5 j2 j4 j3 j1 j0
    return (j & 3) | (((j >> 3) & 3) << 2) | (((j >> 2) & 1) << 4) | ((j >> 5) << 5);
}

template <bool F16X3>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(RRL_PLAN_WAVES_PER_EU, RRL_PLAN_WAVES_PER_EU)))
void plan_first_step_kernel(
    const float* __restrict__ pk, int n_nets, long long n_groups, int pop, int plan_hor,
    const float* __restrict__ cur_obs, const float* __restrict__ ac_seqs, float* __restrict__ q0,
    float* __restrict__ e0, const int32_t* __restrict__ m_dev) {
    if (m_dev) n_groups = (long long)m_dev[0] * pop;
    const long long tile = blockIdx.x;
    const int net = int(tile % (n_nets + 1));
    const long long gblock = tile / (n_nets + 1);
    if (gblock * kRows >= n_groups) return;           // device-counted planning set: workgroups past the live rows leave
    if (net < n_nets && plan_hor == 1) return;        // nothing reads the prediction of the last step (MPC.py:406-412)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const PlanLds L = carve(lds);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* g = pk + glob_off(n_nets);                // mu[4], sigma[4], max_logvar[2], min_logvar[2]
    const bool owner = tid < kRows;
    const long long group = gblock * kRows + tid;
    const int row = first_step_row(tid & (kRows - 1));
    if (owner) {
        f32x4 x = {0.f, 0.f, 0.f, 0.f};
        if (group < n_groups) {
            const long long m = group / pop;
            x[0] = cur_obs[m * 2];
            x[1] = cur_obs[m * 2 + 1];
            x[2] = ac_seqs[group * (plan_hor * 2)];
            x[3] = ac_seqs[group * (plan_hor * 2) + 1];
        }
        write_inputs(L, g, row, x);
    }
    __syncthreads();
    if (net == n_nets) {
        q_phase<F16X3>(L, pk, wave, lane);
        if (owner && group < n_groups) q0[group] = q_value(L, pk, row);
    } else {
        const float* epk = pk + e_off(net);
        e_phase<F16X3>(L, epk, wave, lane, tid);
        if (owner && group < n_groups)
            *reinterpret_cast<f32x4*>(e0 + (group * n_nets + net) * 4) = e_value(L, epk, g, row);
    }
}

// Steps 1 .. plan_hor - 1 for every particle row, starting from the first step's values (plan_first_step_kernel) and this
// row's own noise draw.  The ensemble phase of the LAST step is not run: its prediction would only become a cur_obs that
// nothing reads (MPC.py:406-412).
template <bool F16X3>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(RRL_PLAN_WAVES_PER_EU, RRL_PLAN_WAVES_PER_EU)))
void plan_cost_kernel(
    const float* __restrict__ pk, int n_nets, int npart, long long n_groups, int pop, int plan_hor,
    const float* __restrict__ cur_obs, const float* __restrict__ ac_seqs, const float* __restrict__ noise,
    uint64_t seed, uint64_t counter, const uint64_t* __restrict__ counter_dev, const float* __restrict__ q0,
    const float* __restrict__ e0, float* __restrict__ partial, const int32_t* __restrict__ m_dev) {
    if (m_dev) {
        // the number of planning problems was decided on the device (rrl_cem_begin): the grid covers the launch bound,
        // workgroups past the live tiles leave before they touch anything
        n_groups = (long long)m_dev[0] * pop;
        if ((long long)blockIdx.x >= ((n_groups + 15) / 16) * n_nets) return;
    }
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const PlanLds L = carve(lds);
    float* rowstate = L.rowstate;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform: tile indices live in SGPRs
    const long long tile = blockIdx.x;
    const int e = int(tile % n_nets);
    const long long gblock = tile / n_nets;
    const int ppn = npart / n_nets;                       // particles per net (4)
    const uint64_t ctr = rrl::effective_counter(counter, counter_dev);
    const long long nrows = n_groups * npart;

    // the particle noise of step t for one row (explicit array, or Philox: stream RRL_STREAM_PLAN, row, tick * 16 + t)
    const auto draw = [&](int t, long long row_global, float& z0, float& z1) {
        if (noise) {
            z0 = noise[((long long)t * nrows + row_global) * 2];
            z1 = noise[((long long)t * nrows + row_global) * 2 + 1];
        } else {
            double d0, d1;
            rrl::normal_at(seed, uint32_t(row_global), rrl::kStreamPlan, ctr * 16 + uint64_t(t), d0, d1);
            z0 = float(d0);
            z1 = float(d1);
        }
    };

    // one thread per row advances the row's rollout state, which lives in LDS between the steps (registers that stay
    // live across the matrix phases are what the 128-VGPR budget of two workgroups per CU is short of)
    const bool owner = tid < kRows;
    const float* epk = pk + e_off(e);
    const float* g = pk + glob_off(n_nets);                // mu[4], sigma[4], max_logvar[2], min_logvar[2]
    if (owner) {
        // step 0: cost = Q_risk(cur_obs, ac_0); obs_1 = cur_obs + (mean + z sd) (obs_postproc: obs + prediction, :131-133)
        const long long group = gblock * 16 + (tid >> 2);
        f32x4 st = {0.f, 0.f, 0.f, 0.f};
        if (group < n_groups) {
            const long long m = group / pop;
            st[0] = cur_obs[m * 2];
            st[1] = cur_obs[m * 2 + 1];
            st[2] = q0[group];
            if (plan_hor > 1) {
                const f32x4 d = *reinterpret_cast<const f32x4*>(e0 + (group * n_nets + e) * 4);
                float z0, z1;
                draw(0, group * npart + e * ppn + (tid & 3), z0, z1);
                st[0] = st[0] + (d[0] + z0 * d[2]);
                st[1] = st[1] + (d[1] + z1 * d[3]);
            }
        }
        *reinterpret_cast<f32x4*>(rowstate + tid * 4) = st;
    }

    const int order = __builtin_amdgcn_readfirstlane(int(blockIdx.x >> 8) & 1);

    for (int t = 1; t < plan_hor; ++t) {
        if (owner) {
            const int otid = opaque(tid);        // keeps the 64-bit row addresses out of the loop-invariant (spilled) set
            const long long group = gblock * 16 + (otid >> 2);
            float ax = 0.f, ay = 0.f;
            if (group < n_groups) {
                ax = ac_seqs[group * (plan_hor * 2) + 2 * t];
                ay = ac_seqs[group * (plan_hor * 2) + 2 * t + 1];
            }
            write_inputs(L, g, tid, f32x4{rowstate[tid * 4], rowstate[tid * 4 + 1], ax, ay});
        }
        __syncthreads();
        const bool predict = t + 1 < plan_hor;

        // The two networks read the same (obs, ac) and are independent, so their order is free.  Workgroups that
        // share a CU (dispatch slots alternate every 256 workgroups: 8 XCDs x 32 CUs) run them in opposite order,
        // which keeps one of them in a matrix phase while the other is in an epilogue or at a barrier.
#pragma unroll 1
        for (int phase = 0; phase < 2; ++phase) {
            if ((phase ^ order) == 0) q_phase<F16X3>(L, pk, wave, lane);
            else if (predict) e_phase<F16X3>(L, epk, wave, lane, tid);
        }

        // ---- per-row tail: cost, predictive distribution, next observation ----
        if (owner) {
            const int otid = opaque(tid);
            const long long group = gblock * 16 + (otid >> 2);
            f32x4 st = *reinterpret_cast<f32x4*>(rowstate + tid * 4);
            st[2] += q_value(L, pk, tid);
            if (predict) {
                const f32x4 d = e_value(L, epk, g, tid);
                float z0 = 0.f, z1 = 0.f;
                if (group < n_groups) draw(t, group * npart + e * ppn + (otid & 3), z0, z1);
                st[0] = st[0] + (d[0] + z0 * d[2]);
                st[1] = st[1] + (d[1] + z1 * d[3]);
            }
            *reinterpret_cast<f32x4*>(rowstate + tid * 4) = st;
        }
        // xs / xn are rewritten by the owners only after every wave passed the barriers above
    }

    // sum of the member's particles per (m, c) group; NaN -> 1e6 per particle (MPC.py:415)
    if (owner) {
        const long long group = gblock * 16 + (tid >> 2);
        const float cost = rowstate[tid * 4 + 2];
        float c = (cost != cost) ? 1e6f : cost;
        c += __shfl_xor(c, 1, 64);
        c += __shfl_xor(c, 2, 64);
        if (group < n_groups && (tid & 3) == 0) partial[group * n_nets + e] = c;
    }
}

// costs[g] = (sum over members of partial[g][e]) / npart
__global__ __launch_bounds__(kBlock) void plan_finish_kernel(long long n_groups, int n_nets, int npart,
                                                             const float* __restrict__ partial,
                                                             float* __restrict__ costs, uint64_t* counter_dev,
                                                             uint64_t counter_inc, const int32_t* __restrict__ m_dev,
                                                             int pop) {
    if (m_dev) n_groups = (long long)m_dev[0] * pop;
    for (long long g = blockIdx.x * (long long)blockDim.x + threadIdx.x; g < n_groups;
         g += (long long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int e = 0; e < n_nets; ++e) s += partial[g * n_nets + e];
        costs[g] = s / float(npart);
    }
    // an EMPTY device-counted planning set leaves the tick alone, as the host-count path (MPC.act returns before it plans)
    rrl::advance_counter(counter_dev, (m_dev && n_groups == 0) ? 0 : counter_inc);
}

// ---- weight packing -----------------------------------------------------------------------------
// fragment stream for one [N x K] layer: out[((ct * J + j) * 64 + lane) * 4 + t] = W[n = 16 ct + lane % 16]
// [k = 16 j + 4 (lane / 16) + t], zero outside N x K.  W element (n, k) is at src[n * sn + k * sk].
__global__ __launch_bounds__(kBlock) void pack_layer_kernel(const float* __restrict__ src, long long sn, long long sk,
                                                            int N, int K, int tiles, int J, float* __restrict__ dst) {
    const long long total = (long long)tiles * J * 256;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int t = int(i & 3), lane = int((i >> 2) & 63);
        const long long f = i >> 8;
        const int j = int(f % J), ct = int(f / J);
        const int n = 16 * ct + (lane & 15), k = 16 * j + 4 * (lane >> 4) + t;
        dst[i] = (n < N && k < K) ? src[n * sn + k * sk] : 0.f;
    }
}

// fragment stream of the F16X3 kernel (K = 32 MFMA shape): slot (ct, 2 jp, lane) = hi[8], slot (ct, 2 jp + 1, lane) = lo[8]
// of W[n = 16 ct + lane % 16][k = 32 jp + 8 (lane / 16) + 0..7].  J (16-wide chunks) must be even.
__global__ __launch_bounds__(kBlock) void pack_layer_f16x3_k32_kernel(const float* __restrict__ src, long long sn,
                                                                      long long sk, int N, int K, int tiles, int J,
                                                                      float* __restrict__ dst) {
    const long long total = (long long)tiles * (J / 2) * 64;
    for (long long f = blockIdx.x * (long long)blockDim.x + threadIdx.x; f < total;
         f += (long long)gridDim.x * blockDim.x) {
        const int lane = int(f & 63);
        const long long blk = f >> 6;
        const int jp = int(blk % (J / 2)), ct = int(blk / (J / 2));
        const int n = 16 * ct + (lane & 15);
        _Float16* hi = reinterpret_cast<_Float16*>(dst + (((long long)ct * J + 2 * jp) * 64 + lane) * 4);
        _Float16* lo = reinterpret_cast<_Float16*>(dst + (((long long)ct * J + 2 * jp + 1) * 64 + lane) * 4);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int k = 32 * jp + 8 * (lane >> 4) + t;
            float w = (n < N && k < K) ? src[n * sn + k * sk] : 0.f;
            w = w > 65504.f ? 65504.f : (w < -65504.f ? -65504.f : w);
            const _Float16 h = (_Float16)w;
            hi[t] = h;
            lo[t] = (_Float16)(w - (float)h);
        }
    }
}

// first layers (K = 4): out[ct * 64 + lane] = W[n = 16 ct + lane % 16][k = lane / 16]
__global__ __launch_bounds__(kBlock) void pack_input_kernel(const float* __restrict__ src, long long sn, long long sk,
                                                            int N, int tiles, float* __restrict__ dst) {
    const int total = tiles * 64;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int lane = i & 63, ct = i >> 6;
        const int n = 16 * ct + (lane & 15), k = lane >> 4;
        dst[i] = n < N ? src[n * sn + k * sk] : 0.f;
    }
}

// vectors (bias, last-layer weights) copied with zero padding; src element i at src[i * stride]
__global__ __launch_bounds__(kBlock) void pack_vector_kernel(const float* __restrict__ src, long long stride, int n,
                                                             int n_pad, float* __restrict__ dst) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_pad; i += gridDim.x * blockDim.x)
        dst[i] = i < n ? src[i * stride] : 0.f;
}

// last ensemble layer [HE x 4] as [col][o] with zero rows up to the padded width
__global__ __launch_bounds__(kBlock) void pack_head_kernel(const float* __restrict__ src, int he,
                                                           float* __restrict__ dst) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < kHEPad * 4; i += gridDim.x * blockDim.x)
        dst[i] = (i >> 2) < he ? src[i] : 0.f;          // lin3_w[e] is [HE][4] row-major already
}

}  // namespace

extern "C" {

long long rrl_plan_pack_floats(int hq, int he, int n_nets) {
    if (hq != kHQ || he != kHE || n_nets <= 0) return RRL_EINVAL;
    return (long long)packed_floats(n_nets);
}

int rrl_plan_supported(int hq, int he, int n_nets, int npart, int d_obs, int d_act) {
    return hq == kHQ && he == kHE && n_nets > 0 && npart % n_nets == 0 && npart / n_nets == 4 && d_obs == 2 &&
           d_act == 2;
}

static int plan_pack_impl(const rrl_plan_weights_t* w, float* packed, void* stream_, bool f16x3) {
    if (!w || !packed || !rrl_plan_supported(w->hq, w->he, w->n_nets, 4 * w->n_nets, 2, 2)) return RRL_EINVAL;
    hipStream_t st = (hipStream_t)stream_;
    const dim3 b(kBlock);
    // hidden layers: f32 fragments of J 16-wide chunks, or (f16x3) {hi, lo} fragments of J32 32-wide blocks
    const auto pack_layer = [&](const float* src, long long sn, long long sk, int N, int K, int tiles, int J, int J32,
                                float* dst) {
        if (f16x3)
            hipLaunchKernelGGL(pack_layer_f16x3_k32_kernel, dim3(64), b, 0, st, src, sn, sk, N, K, tiles, 2 * J32, dst);
        else
            hipLaunchKernelGGL(pack_layer_kernel, dim3(64), b, 0, st, src, sn, sk, N, K, tiles, J, dst);
    };
    for (int h = 0; h < 2; ++h) {
        float* d = packed + q_off(h);
        hipLaunchKernelGGL(pack_input_kernel, dim3(4), b, 0, st, w->q_w1 + (size_t)h * kHQ * 4, 4LL, 1LL, kHQ,
                           kQTiles, d + kQW1);
        hipLaunchKernelGGL(pack_vector_kernel, dim3(1), b, 0, st, w->q_b1 + h * kHQ, 1LL, kHQ, kHQ, d + kQB1);
        pack_layer(w->q_w2 + (size_t)h * kHQ * kHQ, (long long)kHQ, 1LL, kHQ, kHQ, kQTiles, kQTiles, kQTiles / 2, d + kQW2);
        hipLaunchKernelGGL(pack_vector_kernel, dim3(1), b, 0, st, w->q_b2 + h * kHQ, 1LL, kHQ, kHQ, d + kQB2);
        hipLaunchKernelGGL(pack_vector_kernel, dim3(1), b, 0, st, w->q_w3 + h * kHQ, 1LL, kHQ, kHQ, d + kQW3);
        hipLaunchKernelGGL(pack_vector_kernel, dim3(1), b, 0, st, w->q_b3 + h, 1LL, 1, 4, d + kQB3);
    }
    for (int e = 0; e < w->n_nets; ++e) {
        float* d = packed + e_off(e);
        // ensemble weights are [in][out] (torch.baddbmm(b, x, w)): element (n = out, k = in) at w[k * out_dim + n]
        hipLaunchKernelGGL(pack_input_kernel, dim3(4), b, 0, st, w->e_w0 + (size_t)e * 4 * kHE, 1LL, (long long)kHE,
                           kHE, kETiles, d + kEW0);
        hipLaunchKernelGGL(pack_vector_kernel, dim3(1), b, 0, st, w->e_b0 + e * kHE, 1LL, kHE, kHEPad, d + kEB0);
        pack_layer(w->e_w1 + (size_t)e * kHE * kHE, 1LL, (long long)kHE, kHE, kHE, kETiles, kETiles, kEBlocks32, d + kEW1);
        hipLaunchKernelGGL(pack_vector_kernel, dim3(1), b, 0, st, w->e_b1 + e * kHE, 1LL, kHE, kHEPad, d + kEB1);
        pack_layer(w->e_w2 + (size_t)e * kHE * kHE, 1LL, (long long)kHE, kHE, kHE, kETiles, kETiles, kEBlocks32, d + kEW2);
        hipLaunchKernelGGL(pack_vector_kernel, dim3(1), b, 0, st, w->e_b2 + e * kHE, 1LL, kHE, kHEPad, d + kEB2);
        hipLaunchKernelGGL(pack_head_kernel, dim3(4), b, 0, st, w->e_w3 + (size_t)e * kHE * 4, kHE, d + kEW3);
        hipLaunchKernelGGL(pack_vector_kernel, dim3(1), b, 0, st, w->e_b3 + e * 4, 1LL, 4, 4, d + kEB3);
    }
    float* gl = packed + glob_off(w->n_nets);
    hipLaunchKernelGGL(pack_vector_kernel, dim3(1), b, 0, st, w->inputs_mu, 1LL, 4, 4, gl);
    hipLaunchKernelGGL(pack_vector_kernel, dim3(1), b, 0, st, w->inputs_sigma, 1LL, 4, 4, gl + 4);
    hipLaunchKernelGGL(pack_vector_kernel, dim3(1), b, 0, st, w->max_logvar, 1LL, 2, 2, gl + 8);
    hipLaunchKernelGGL(pack_vector_kernel, dim3(1), b, 0, st, w->min_logvar, 1LL, 2, 2, gl + 10);
    return check_launch();
}

int rrl_plan_pack(const rrl_plan_weights_t* w, float* packed, void* stream) {
    return plan_pack_impl(w, packed, stream, false);
}

int rrl_plan_pack_f16x3(const rrl_plan_weights_t* w, float* packed, void* stream) {
    return plan_pack_impl(w, packed, stream, true);
}

long long rrl_plan_scratch_floats(int n_nets, long long M, int pop) {
    if (n_nets <= 0 || M <= 0 || pop <= 0) return RRL_EINVAL;
    return M * pop * (5LL * n_nets + 1);
}

static int plan_cost_impl(bool f16x3, const float* packed, int hq, int he, int n_nets, int npart, long long M, int pop,
                          int plan_hor, const float* cur_obs, const float* ac_seqs, const float* noise, uint64_t seed,
                          uint64_t counter, uint64_t* counter_dev, uint64_t counter_inc, float* scratch, float* costs,
                          void* stream_, const int32_t* m_dev = nullptr) {
    if (!packed || !cur_obs || !ac_seqs || !scratch || !costs || M <= 0 || pop <= 0 || plan_hor <= 0 ||
        plan_hor > 16 || !rrl_plan_supported(hq, he, n_nets, npart, 2, 2))
        return RRL_EINVAL;
    const long long n_groups = M * pop;
    if (n_groups * npart >= (1LL << 32)) return RRL_EINVAL;      // Philox row index is 32 bits
    const long long tiles = ((n_groups + 15) / 16) * n_nets;
    const long long first_tiles = ((n_groups + kRows - 1) / kRows) * (n_nets + 1);
    if (tiles >= (1LL << 31)) return RRL_EINVAL;
    // scratch (rrl_plan_scratch_floats): e0 [n_groups][n_nets][4] | partial [n_groups][n_nets] | q0 [n_groups]
    float* e0 = scratch;
    float* partial = e0 + n_groups * n_nets * 4;
    float* q0 = partial + n_groups * n_nets;
    hipStream_t st = (hipStream_t)stream_;
    static bool lds_set = false;
    if (!lds_set) {       // > 64 KB of LDS has to be granted explicitly
        const void* kernels[4] = {(const void*)plan_cost_kernel<false>, (const void*)plan_cost_kernel<true>,
                                  (const void*)plan_first_step_kernel<false>, (const void*)plan_first_step_kernel<true>};
        for (const void* k : kernels)
            if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes) != hipSuccess) {
                last_hip_error = int(hipGetLastError());
                return RRL_ELAUNCH;
            }
        lds_set = true;
    }
    if (f16x3) {
        hipLaunchKernelGGL(plan_first_step_kernel<true>, dim3((unsigned)first_tiles), dim3(kThreads), kLdsBytes, st, packed,
                           n_nets, n_groups, pop, plan_hor, cur_obs, ac_seqs, q0, e0, m_dev);
        hipLaunchKernelGGL(plan_cost_kernel<true>, dim3((unsigned)tiles), dim3(kThreads), kLdsBytes, st, packed, n_nets,
                           npart, n_groups, pop, plan_hor, cur_obs, ac_seqs, noise, seed, counter, counter_dev, q0, e0,
                           partial, m_dev);
    } else {
        hipLaunchKernelGGL(plan_first_step_kernel<false>, dim3((unsigned)first_tiles), dim3(kThreads), kLdsBytes, st, packed,
                           n_nets, n_groups, pop, plan_hor, cur_obs, ac_seqs, q0, e0, m_dev);
        hipLaunchKernelGGL(plan_cost_kernel<false>, dim3((unsigned)tiles), dim3(kThreads), kLdsBytes, st, packed, n_nets,
                           npart, n_groups, pop, plan_hor, cur_obs, ac_seqs, noise, seed, counter, counter_dev, q0, e0,
                           partial, m_dev);
    }
    hipLaunchKernelGGL(plan_finish_kernel, dim3(grid_for(n_groups)), dim3(kBlock), 0, st, n_groups, n_nets, npart,
                       partial, costs, counter_dev, counter_inc, m_dev, pop);
    return check_launch();
}

int rrl_plan_cost(const float* packed, int hq, int he, int n_nets, int npart, long long M, int pop, int plan_hor,
                  const float* cur_obs, const float* ac_seqs, const float* noise, uint64_t seed, uint64_t counter,
                  uint64_t* counter_dev, uint64_t counter_inc, float* scratch, float* costs, void* stream) {
    return plan_cost_impl(false, packed, hq, he, n_nets, npart, M, pop, plan_hor, cur_obs, ac_seqs, noise, seed, counter,
                          counter_dev, counter_inc, scratch, costs, stream);
}

int rrl_plan_cost_f16x3(const float* packed, int hq, int he, int n_nets, int npart, long long M, int pop, int plan_hor,
                        const float* cur_obs, const float* ac_seqs, const float* noise, uint64_t seed, uint64_t counter,
                        uint64_t* counter_dev, uint64_t counter_inc, float* scratch, float* costs, void* stream) {
    return plan_cost_impl(true, packed, hq, he, n_nets, npart, M, pop, plan_hor, cur_obs, ac_seqs, noise, seed, counter,
                          counter_dev, counter_inc, scratch, costs, stream);
}

int rrl_plan_cost_n(int f16x3, const float* packed, int hq, int he, int n_nets, int npart, const int32_t* m_dev,
                    long long m_max, int pop, int plan_hor, const float* cur_obs, const float* ac_seqs, const float* noise,
                    uint64_t seed, uint64_t counter, uint64_t* counter_dev, uint64_t counter_inc, float* scratch,
                    float* costs, void* stream) {
    if (!m_dev) return RRL_EINVAL;
    return plan_cost_impl(f16x3 != 0, packed, hq, he, n_nets, npart, m_max, pop, plan_hor, cur_obs, ac_seqs, noise, seed,
                          counter, counter_dev, counter_inc, scratch, costs, stream, m_dev);
}

}  // extern "C"
