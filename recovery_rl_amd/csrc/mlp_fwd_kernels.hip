// mlp_fwd_kernels.hip -- fused forward of a whole 2-hidden-layer stack for gfx950 (MI355X): the SAC / Q_risk networks'
// forward passes of the updates (B = 256 rows) and of the acting pass (one row per env).
//
//   out[g] = W3[g] relu(W2[g] relu(W1[g] x + b1[g]) + b2[g]) + b3[g]      x [M, din] shared by the heads
// One workgroup (16 waves) per 16 rows and head: layer 1 on the VALU (din <= 4), layer 2 on MFMA with
// the 16 x H activation tile in LDS shared by all waves (wave w owns output columns 16w..16w+15 and
// streams its 16 rows of W2 straight from L2 into MFMA operands), layer 3 by 16-lane dot products.
// No intermediate activation touches HBM unless the caller asks for h1 / h2 (needed by backward).
#include "mlp_common.hpp"

namespace {

using rrl_host::check_launch;


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
    const float* W2p;     // optional: W2 in fragment order (rrl_stack_t.W2p)
};

constexpr int kStackRows = 16;
constexpr int kStackMaxH = 256;

// Sum over each 16-lane row of a wave, result in every lane, in the order of the xor butterfly 8, 4, 2, 1 (bit-identical
// to `v += __shfl_xor(v, 8); ... 4; 2; 1`): after step k the row's values repeat with period 16 / 2^k, so the partner
// lane^m holds the same value as lane + m (mod 16) and a DPP row rotation delivers it -- one VALU instruction with a DPP
// operand per step instead of a ds_bpermute round trip through the LDS crossbar (~120 cycles each, four dependent
// ones per output row: 7 000 of the 25 000 cycles of a 64-row forward tile).
using rrl::row16_sum;

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
    rrl_pack::to_global_all(a.x, a.W1, a.b1, a.W2, a.b2, a.W3, a.b3, a.h1, a.h2, a.out, a.W2p);
    rrl_pack::globalize(a.in_head);
}
struct StackGroup {
    StackArgs a[kMaxGroup];
    float* partial[kMaxGroup];      // split variant only
    int G[kMaxGroup], tiles[kMaxGroup];
    int nb[kMaxGroup];              // row blocks per workgroup (1 except in the packed loop form)
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
constexpr int kSplitPad = 20;   // pad floats per row of the h1 tile: rows 16-byte aligned, ds_read_b128 conflict-free

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
// LOOP (packed launches with many workgroups per CU slot, rrl_mlp3_forward_multi_packed): the workgroup keeps its weights and
// walks `nb` consecutive row blocks of its (stack, head, column group) -- what a workgroup costs besides its MFMAs (argument
// block, ~150 address instructions, 64 KB of W2 from L2, its launch) is paid once per nb blocks.  On gfx950 VALU work does not
// run under v_mfma_f32_16x16x4_f32 -- neither a partner wave's nor the wave's own (profiles/mfma_valu_overlap.hip,
// mfma_valu_inwave.hip) -- so in the throughput regime a SIMD's time is the SUM of its waves' MFMA and VALU cycles and every
// instruction removed counts.  Per output element the arithmetic is the one-block form's.
template <int R, int HC, bool LOOP = false>
__device__ __forceinline__ void mlp3_fwd_split_body(const StackArgs& a, float* partial, int bx, int g, int z, int G,
                                                    float* h1s, float* h2s, int nb = 1, int tiles = 1, float* xs_own = nullptr) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (wave-uniform, and the compiler may know it)
    constexpr int kW = 4;                                // waves per workgroup
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
    const int ntiles1 = H / 16;                    // layer-1 column tiles, 4 waves take them round-robin
    const bool has_tile2 = HC ? true : wave * 16 < HS;   // my layer-2 tile inside this group's columns
    const int n2 = colbase + (has_tile2 ? wave * 16 : 0);

    float w1b[kU], bias1[kU];
    float4 wv[kJ];                  // my 16 rows of W2 as MFMA B operands: lane (i, q) holds W2[n2 + i][16 j + 4 q .. + 3] in wv[j]
    float bias2, w3v[kT3], bias3;
    const auto load_weights = [&]() {
        const int i = lane & 15, q = lane >> 4, o3 = min(q, dout - 1);
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int t = min(wave + kW * u, ntiles1 - 1);
            const float wv1 = W1[(t * 16 + i) * din + min(q, din - 1)];
            w1b[u] = (q < din) ? wv1 : 0.f;
            bias1[u] = b1[t * 16 + i];
        }
        if (HC == 256) {
            // from the fragment-order copy when there is one: the wave's 16 K chunks are 16 consecutive KB and every load
            // instruction covers whole 128-byte lines (row-major, lane (i, q) reads 16 bytes of row n2 + i: an instruction touches
            // 16 lines and uses half of each -- 16.3 -> 13.8 us for the 4096-row forward, 5.7 -> 4.75 for the 256-row one).  ONE
            // set of loads for both layouts: uniform base and chunk step, per-lane offset (two arms of a branch that both define
            // the 64 weight registers made the kernel spill 20 of them)
            const bool frag = a.W2p != nullptr;
            const char* wbase = frag ? reinterpret_cast<const char*>(a.W2p + (long long)g * H * H) + (n2 / 16) * (kJ * 1024)
                                     : reinterpret_cast<const char*>(W2 + (long long)n2 * H);
            rrl_pack::to_global(wbase);
            const unsigned voff = frag ? unsigned(lane) * 16u : unsigned(i * H + 4 * q) * 4u;
            const unsigned stepb = frag ? 1024u : 64u;
#pragma unroll
            for (int j = 0; j < kJ; ++j) wv[j] = *reinterpret_cast<const float4*>(wbase + j * stepb + voff);
        } else {
            const float* wrow = W2 + (long long)(n2 + i) * H + 4 * q;
#pragma unroll
            for (int j = 0; j < kJ; ++j) wv[j] = *reinterpret_cast<const float4*>(wrow + min(16 * j, H - 16));
        }
        bias2 = b2[n2 + i];
#pragma unroll
        for (int it = 0; it < kT3; ++it) w3v[it] = W3[o3 * H + colbase + min(i + 16 * it, HS - 1)];
        const float b3v = b3[o3];
        bias3 = (z == 0) ? b3v : 0.f;
    };
    if (LOOP) load_weights();
    const int blk0 = LOOP ? bx * nb : bx, blk1 = LOOP ? min(blk0 + nb, tiles) : bx + 1;
    // raw input of a block, lane (i, q): row 16 t + i, column min(q, din - 1) (lanes q >= din are zeroed behind the input head)
    const auto load_x = [&](int blk, float (&x)[R]) {
        const int i = lane & 15, q = lane >> 4, m0 = blk * (R * kStackRows);
        const float* const xblk = a.x + (long long)m0 * a.ldx;
#pragma unroll
        for (int t = 0; t < R; ++t) x[t] = xblk[unsigned(min(16 * t + i, M - 1 - m0) * a.ldx + min(q, din - 1))];
    };
    // operands of the input head for a block, lane = (row of the block, action dimension j): noise, the <= 4 partial sums of
    // mean and log-std (or the mean's only), the observation; optional ones through a selected address (the row's own x element
    // when absent) instead of under a branch: a load under a branch is waited for where the branch ends
    struct HeadOps { float e, mp[4], rp[4], ob; };
    const int head_wave = (bx + g + z) & (kW - 1);
    const auto load_head = [&](int blk, HeadOps& o) {
        const rrl_policy_head_t& hd = a.in_head;
        const int m0 = blk * (R * kStackRows), j = lane & 1;
        const int rb = min(min(lane >> 1, R * kStackRows - 1), M - 1 - m0);    // row relative to the block, clamped to the batch
        const unsigned row = unsigned(m0 + rb), xoff = unsigned(rb * a.ldx);
        const float* const xblk = a.x + (long long)m0 * a.ldx;
        const float* eb = hd.eps ? hd.eps : xblk;
        const float* obb = hd.obs_in ? hd.obs_in : xblk;
        rrl_pack::to_global_all(eb, obb);               // (a select of two global pointers is not global to the compiler)
        o.e = eb[hd.eps ? 2 * row + j : xoff];
        const bool gauss = hd.kind == RRL_HEAD_GAUSS;
        const unsigned im = gauss ? 4 * row + j : 2 * row + j, ir = gauss ? im + 2 : im;
#pragma unroll
        for (int k = 0; k < 4; ++k) {      // (min, not a select to part 0: the compiler turns that into branches around the loads)
            const float* const hk = hd.head + min(k, hd.n_part - 1) * hd.part_stride;
            o.mp[k] = hk[im];
            o.rp[k] = hk[ir];
        }
        o.ob = obb[hd.obs_in ? 2 * row + j : xoff + j];
    };
    // per-lane bases, computed once: every LDS access and global store of a block is base + a compile-time offset
    const int li = lane & 15, lq = lane >> 4;
    float* const h1w = h1s + (4 * lq) * ldh + wave * 16 + li;           // layer 1 writes row 4 q + r, column 16 (wave + 4 u) + i
    const float* const h1r = h1s + li * ldh + 4 * lq;                   // layer 2 reads row i, columns 16 j + 4 q ..
    float* const h2w = h2s + (4 * lq) * ld2 + wave * 16 + li;           // layer 2 writes row 4 q + r, column 16 wave + i
    const float* const h3r = h2s + (wave * 4) * ld2 + li;               // layer 3 reads row 4 wave + rr, columns i + 16 it
    const unsigned g1off = unsigned((4 * lq) * H + wave * 16 + li);     // the same elements of the h1 / h2 arrays in memory
    const unsigned g2off = unsigned((4 * lq) * H + n2 + li);
    float xn[R];                              // LOOP: the next block's input and head operands, requested a block ahead
    HeadOps hn;
    if (LOOP) {
        load_x(blk0, xn);
        if (a.use_in_head && wave == head_wave) load_head(blk0, hn);
    }
    for (int blk = blk0; blk < blk1; ++blk) {
    int m0 = blk * (R * kStackRows);
    // (opaque per block: otherwise the loop optimiser keeps one induction pointer per global store of the body -- 40 of them --
    // and the kernel spills 63 registers to stay at four waves per SIMD)
    if (LOOP) asm volatile("" : "+s"(m0));
    // (the lane's coordinates likewise: with them loop-invariant, every one of the body's ~40 store offsets is hoisted as a
    // zero-extended 64-bit pair and 75 registers spill; recomputed per block they are a dozen instructions)
    int lane_b = lane;
    if (LOOP) asm volatile("" : "+v"(lane_b));
    const int i = lane_b & 15, q = lane_b >> 4;
    // addresses: a uniform base per block (scalar registers) + a small per-lane offset (32 bits), so that the loop form keeps a
    // handful of lane offsets across blocks instead of a 64-bit pointer per access
    float* const h1g = (a.h1 && z == 0) ? a.h1 + ((long long)g * M + m0) * H : nullptr;
    float* const h2g = a.h2 ? a.h2 + ((long long)g * M + m0) * H : nullptr;
    float* const pout = partial + (((long long)z * G + g) * M + m0) * dout;
    const bool full = m0 + R * kStackRows <= M;    // uniform: every row of the workgroup's tiles exists
    const int last = M - 1 - m0;                   // last existing row, relative to the block

    // ---- all global reads up front, branch-free ---------------------------------------------------------
    float xa[R];
    if (LOOP) {
#pragma unroll
        for (int t = 0; t < R; ++t) xa[t] = xn[t];
        load_x(min(blk + 1, blk1 - 1), xn);
    } else load_x(blk, xa);
    if (a.use_in_head) {
        // Columns 0, 1 of the input are the observation, columns 2, 3 the action the policy head yields for the row.  ONE wave
        // evaluates it for the workgroup's R * 16 rows, a lane per (row, action dimension j) -- 32 rows fill the wave in one
        // pass -- and hands the values to the others through LDS (the h2 tile's space: nothing lives there yet); the wave
        // rotates with the workgroup's index: wave w of every workgroup sits on SIMD w, and with the head always on wave 0 one
        // SIMD of each CU carried the whole transcendental chain of the four (eight) workgroups that re-evaluate a row's head
        // (56 of 242 us of the packed 16-seed launch, profiles/round5_fwd_packed/).  Workgroup (z, g) = (0, 0) stores
        // action and log-probability for the consumers downstream.  Same formulas, same bits as the kernels of
        // update_kernels.hip.
        static_assert(R * kStackRows <= 32, "the input head is ONE pass of one wave, a lane per (row, action dimension): 32 rows");
        const rrl_policy_head_t& hd = a.in_head;
        const bool writer = z == 0 && g == 0;
        float* xs = LOOP ? xs_own : h2s;                 // [R * 16][4]
        if (wave == head_wave) {
            const int rl = min(lane_b >> 1, R * kStackRows - 1), j = lane_b & 1;
            const bool lane_ok = (lane_b >> 1) < R * kStackRows;
            const unsigned row = unsigned(m0 + min(rl, last));
            const bool row_ok = lane_ok && rl <= last;
            const bool gauss = hd.kind == RRL_HEAD_GAUSS;
            const float sc = hd.scale[j], bi = hd.bias[j];
            const float* lb = hd.log_std ? hd.log_std : hd.scale;
            rrl_pack::to_global(lb);
            const float lstd_ = lb[j];
            HeadOps ho;
            if (LOOP) {
                ho = hn;
                load_head(min(blk + 1, blk1 - 1), hn);
            } else load_head(blk, ho);
            const float e_ = ho.e, ob = ho.ob;
            const float (&mp)[4] = ho.mp;
            const float (&rp)[4] = ho.rp;
            const auto fold = [&](const float (&v)[4]) {
                float x = v[0];
#pragma unroll
                for (int k = 1; k < 4; ++k) x = hd.n_part > k ? x + v[k] : x;
                return x;
            };
            float val, lp_term = 0.f;
            const float e = hd.eps ? e_ : 0.f;
            if (gauss) {
                const float mean = fold(mp);
                const float ls = fminf(fmaxf(fold(rp), loss::kLogSigMin), loss::kLogSigMax);
                const float y = tanhf(mean + expf(ls) * e);
                val = y * sc + bi;
                lp_term = -0.5f * e * e - ls - 0.918938533204672742f - logf(sc * (1.f - y * y) + loss::kEps);
            } else {
                const float mean = tanhf(fold(mp)) * sc + bi;
                val = mean + expf(fmaxf(lstd_, hd.min_log_std)) * e;
            }
            const float other = __shfl_xor(lp_term, 1);           // lane (row, 0) <-> lane (row, 1)
            if (writer && row_ok) {
                if (hd.action) hd.action[(long long)row * hd.ld_action + j] = val;
                if (hd.logp && j == 0) hd.logp[row] = lp_term + other;
                if (hd.obs_in && hd.obs_out) hd.obs_out[(long long)row * hd.ld_action + j] = ob;
            }
            if (lane_ok) {
                xs[rl * 4 + j] = ob;
                xs[rl * 4 + 2 + j] = val;
            }
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < R; ++t) xa[t] = xs[(16 * t + i) * 4 + q];
        if (!LOOP) __syncthreads();                      // h2s is reused by layer 2 (and aliases h1s for R > 1)
    } else {
#pragma unroll
        for (int t = 0; t < R; ++t) xa[t] = (q < din) ? xa[t] : 0.f;
    }
    if (!LOOP) load_weights();

    // ---- layer 1 (all H columns; one MFMA step per 16-column tile and row tile) ---------------------------
#pragma unroll
    for (int u = 0; u < kU; ++u) {
        const int t = wave + kW * u;
        if (HC || t < ntiles1) {
#pragma unroll
            for (int rt = 0; rt < R; ++rt) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[rt], w1b[u], acc, 0, 0, 0);
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = acc[r] + bias1[u];
                    v[r] = v[r] > 0.f ? v[r] : 0.f;
                    h1w[(16 * rt + r) * ldh + 16 * kW * u] = v[r];
                }
                if (h1g) {
                    float* const dst = h1g + g1off + 16 * rt * H + 16 * kW * u;
                    if (full) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) dst[r * H] = v[r];
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (16 * rt + 4 * q + r <= last) dst[r * H] = v[r];
                    }
                }
            }
        }
    }
    __syncthreads();
    // ---- layer 2: my 16 columns, R row tiles sharing the W2 fragments -----------------------------------------
    if (has_tile2) {
        f32x4 acc0[R], acc1[R];
#pragma unroll
        for (int rt = 0; rt < R; ++rt) {
            acc0[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
            acc1[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < kJ; ++j) {
            if (HC || 16 * j < H) {
#pragma unroll
                for (int rt = 0; rt < R; ++rt) {
                    const float4 av = *reinterpret_cast<const float4*>(h1r + 16 * rt * ldh + 16 * j);
                    acc0[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, wv[j].x, acc0[rt], 0, 0, 0);
                    acc1[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, wv[j].y, acc1[rt], 0, 0, 0);
                    acc0[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, wv[j].z, acc0[rt], 0, 0, 0);
                    acc1[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, wv[j].w, acc1[rt], 0, 0, 0);
                }
            }
        }
        // R > 1: the h2 tile reuses the h1 tile's LDS (one 70 KB tile per workgroup instead of 87 KB: two workgroups per
        // CU), so every wave must be done reading h1 first.  (All four waves own a layer-2 tile here: HC fixes H = 256.)
        if (R > 1 && !LOOP) __syncthreads();
#pragma unroll
        for (int rt = 0; rt < R; ++rt) {
            const f32x4 acc = acc0[rt] + acc1[rt];
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = acc[r] + bias2;
                v[r] = v[r] > 0.f ? v[r] : 0.f;
                h2w[(16 * rt + r) * ld2] = v[r];
            }
            if (h2g) {
                float* const dst = h2g + g2off + 16 * rt * H;
                if (full) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) dst[r * H] = v[r];
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (16 * rt + 4 * q + r <= last) dst[r * H] = v[r];
                }
            }
        }
    }
    __syncthreads();
    // ---- layer 3 partial over my HS columns: wave w -> rows 4 w .. 4 w + 3 of every row tile -----------------
    float res[R][4];
#pragma unroll
    for (int rt = 0; rt < R; ++rt) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int r = 16 * rt + wave * 4 + rr;
            float v = 0.f;
#pragma unroll
            for (int it = 0; it < kT3; ++it) {
                const float hv = HC ? h3r[(16 * rt + rr) * ld2 + 16 * it] : h2s[r * ld2 + min(i + 16 * it, HS - 1)];
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
            const int r = 16 * rt + wave * 4 + rr;
            if (i == 0 && q < dout && (full || r <= last)) pout[unsigned(r * dout + q)] = v + bias3;
        }
    }
    // LOOP: the three tiles (h1, h2, the head's values) have LDS of their own, so a block needs two barriers (three with an input
    // head) instead of four (six): the next block's layer 1 writes h1 behind this block's second barrier (all of layer 2 has
    // read it), its layer 2 writes h2 behind the next first barrier (every wave's layer 3 of this block is before that in
    // program order), and the head's values are re-written behind a barrier every reader has passed
    }
}

constexpr int kBigR = 2;     // row tiles per workgroup for batches above kSplitSmallM rows (measured: 4 is slower, 10.4 vs 9.6 us)
constexpr int kSplitSmallM = 1024;
constexpr size_t loop_lds_floats(int R) {          // the loop form: h1, h2 and the input head's values side by side
    return size_t(R) * kStackRows * (kStackMaxH + kSplitPad) + size_t(R) * kStackRows * (kStackMaxH / kSplit + 1) + size_t(R) * kStackRows * 4;
}
constexpr size_t split_lds_floats(int R) {
    return size_t(R) * kStackRows * (kStackMaxH + kSplitPad) +
           (R > 1 ? 0 : size_t(R) * kStackRows * (kStackMaxH / kSplit + 1));      // R > 1: the h2 tile aliases the h1 tile
}

template <int R>
__global__ __launch_bounds__(256, 4) void mlp3_fwd_split_kernel(StackArgs a, float* partial) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* h2s = R > 1 ? lds : lds + R * kStackRows * (kStackMaxH + kSplitPad);
    if (a.H == 256) mlp3_fwd_split_body<R, 256>(a, partial, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.y, lds, h2s);
    else mlp3_fwd_split_body<R, 0>(a, partial, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.y, lds, h2s);
}

// Solo launch: grid (row tiles x heads x column splits of the largest member, members) -- see mlp_common.hpp.
// (Requesting the weights before the input head is evaluated instead of behind its barriers was tried twice -- as written, and
// raw with the selects behind the head, for the 256-row forwards only: 4096-row forward 18.5 -> 23.8 us / 256-row ones 7.0 ->
// 8.3 us.  The 64 KB of W2 per workgroup are a throughput term; ahead of the head's few operands they delay the one wave
// whose tanh / exp / log chain everybody waits for.  Not kept.)
template <int R>
__global__ __launch_bounds__(256, 4) void mlp3_fwd_split_group_kernel(StackGroup sg) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int k = blockIdx.y;
    StackArgs a = sg.a[k];
    float* partial = sg.partial[k];
    const int G = sg.G[k], tiles = sg.tiles[k];
    globalize(a);
    rrl_pack::to_global(partial);
    arrive_together(a.M, a.H, a.din, a.dout, a.ldx, a.use_in_head, a.in_head.kind, a.in_head.n_part, a.in_head.part_stride,
                    a.in_head.ld_action, a.in_head.min_log_std, G, tiles);
    const int local = blockIdx.x;
    if (local >= tiles * G * kSplit) return;
    const int bx = local % tiles, rest = local / tiles;
    float* h2s = R > 1 ? lds : lds + R * kStackRows * (kStackMaxH + kSplitPad);
    if (a.H == 256) mlp3_fwd_split_body<R, 256>(a, partial, bx, rest % G, rest / G, G, lds, h2s);
    else mlp3_fwd_split_body<R, 0>(a, partial, bx, rest % G, rest / G, G, lds, h2s);
}

// Members of very different sizes (round 6: a 4096-row acting forward riding with an update's 256-row forwards) on a FLAT
// grid of exactly first[n] workgroups, in member order, every member on the tiles of its stand-alone launch (one row tile
// per workgroup up to kSplitSmallM rows, kBigR above).  On the (largest member, members) grid above the small members'
// surplus workgroups -- 960 of 1024 -- each took a 35 KB LDS slot for the ~1.5 us their argument batch needs before they can
// leave: 21.4 us for a launch whose large member alone takes 16.2.  Measured forms (profiles/round6_rider_forms.txt; the
// three forwards of the acting pass + the two update launches they ride in, 42.9 us as five launches): small members on
// two-row tiles, small first 41.0 / large first 41.9; small members on one-row tiles, small first 39.0 / LARGE FIRST 37.6
// (the twin Q_risk at 4096 rows fills the chip's 1024 resident workgroups exactly: what comes behind it starts as its
// workgroups retire).  The price of the flat grid is the member search in front of the argument batch (one more dependent
// scalar round trip, ~0.5 us of 10-18).
__global__ __launch_bounds__(256, 4) void mlp3_fwd_split_flat_group_kernel(StackGroup sg) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    arrive_together(sg.first[1], sg.first[2], sg.first[3]);
    const int x = blockIdx.x;
    const int k = (x >= sg.first[1]) + (x >= sg.first[2]) + (x >= sg.first[3]);     // first[k + 1] = first[n] past the last member
    StackArgs a = sg.a[k];
    float* partial = sg.partial[k];
    const int G = sg.G[k], tiles = sg.tiles[k];
    globalize(a);
    rrl_pack::to_global(partial);
    arrive_together(a.M, a.H, a.din, a.dout, a.ldx, a.use_in_head, a.in_head.kind, a.in_head.n_part, a.in_head.part_stride,
                    a.in_head.ld_action, a.in_head.min_log_std, G, tiles);
    const int local = x - sg.first[k];
    const int bx = local % tiles, rest = local / tiles;
    if (a.M <= kSplitSmallM)
        mlp3_fwd_split_body<1, 256>(a, partial, bx, rest % G, rest / G, G, lds, lds + kStackRows * (kStackMaxH + kSplitPad));
    else
        mlp3_fwd_split_body<kBigR, 256>(a, partial, bx, rest % G, rest / G, G, lds, lds);
}

// the same launch for S seeds (pack.hpp): grid (workgroups of the seeds' largest members under the XCD-aware placement,
// members) -- blockIdx.y IS the member and the seed follows from blockIdx.x by arithmetic, so the member's argument block
// sits at an address known at wave start: one batch of scalar loads from the plan's device copy, as in the solo launch
template <int R>
__global__ __launch_bounds__(256, 4) void mlp3_fwd_split_pack_kernel(const StackGroup* __restrict__ groups, rrl_pack::Idx ix) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    RRL_PACK_LOCATE(ix, groups, s, local);
    const StackGroup& sg = groups[s];
    const int k = blockIdx.y;
    StackArgs a = sg.a[k];
    float* partial = sg.partial[k];
    const int G = sg.G[k], tiles = sg.tiles[k];
    globalize(a);
    rrl_pack::to_global(partial);
    arrive_together(a.M, a.H, a.din, a.dout, a.ldx, a.use_in_head, a.in_head.kind, a.in_head.n_part, a.in_head.part_stride,
                    a.in_head.ld_action, a.in_head.min_log_std, G, tiles);
    if (local >= tiles * G * kSplit) return;               // (an unused member slot of this seed has tiles = 0)
    const int bx = local % tiles, rest = local / tiles;
    float* h2s = R > 1 ? lds : lds + R * kStackRows * (kStackMaxH + kSplitPad);
    if (a.H == 256) mlp3_fwd_split_body<R, 256>(a, partial, bx, rest % G, rest / G, G, lds, h2s);
    else mlp3_fwd_split_body<R, 0>(a, partial, bx, rest % G, rest / G, G, lds, h2s);
}

#ifndef RRL_FWD_LOOP_WAVES
#define RRL_FWD_LOOP_WAVES 3        // resident workgroups per CU of the loop form (one wave of each per SIMD)
#endif
// ... and with workgroups that keep their weights for nb row blocks (mlp3_fwd_split_body<.., LOOP>): launched when the seeds
// together have more row blocks than the chip holds workgroups at once
template <int R>
__global__ __launch_bounds__(256, RRL_FWD_LOOP_WAVES) void mlp3_fwd_split_pack_loop_kernel(const StackGroup* __restrict__ groups, rrl_pack::Idx ix) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    RRL_PACK_LOCATE(ix, groups, s, local);
    const StackGroup& sg = groups[s];
    const int k = blockIdx.y;
    StackArgs a = sg.a[k];
    float* partial = sg.partial[k];
    const int G = sg.G[k], tiles = sg.tiles[k], nb = sg.nb[k];
    globalize(a);
    rrl_pack::to_global(partial);
    arrive_together(a.M, a.H, a.din, a.dout, a.ldx, a.use_in_head, a.in_head.kind, a.in_head.n_part, a.in_head.part_stride,
                    a.in_head.ld_action, a.in_head.min_log_std, G, tiles, nb);
    const int wt = (tiles + nb - 1) / nb;                  // workgroups per (head, column group)
    if (local >= wt * G * kSplit) return;
    const int bx = local % wt, rest = local / wt;
    float* h2s = lds + R * kStackRows * (kStackMaxH + kSplitPad);
    float* xs = h2s + R * kStackRows * (kStackMaxH / kSplit + 1);
    mlp3_fwd_split_body<R, 256, true>(a, partial, bx, rest % G, rest / G, G, lds, h2s, nb, tiles, xs);
}

__global__ void sum_partials_kernel(int n, const float* __restrict__ partial, float* __restrict__ out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    float v = partial[e];
#pragma unroll
    for (int z = 1; z < kSplit; ++z) v += partial[(long long)z * n + e];
    out[e] = v;
}

constexpr int kPackSmallR2MinSeeds = 3;   // packed launches: 2 row tiles per workgroup for the B <= 1024 forwards from 3 seeds on
constexpr int kResidentWorkgroups = 256 * RRL_FWD_LOOP_WAVES;  // packed loop form: 256 CUs x resident workgroups of the multi-row forward
constexpr int kLoopMaxBlocks = 32, kLoopMinBlocks = 8;

static int env_int_fwd(const char* name, int fallback) {
    const char* v = getenv(name);
    return v && *v ? atoi(v) : fallback;
}
// RRL_PACK_FWD_LOOP=0: one row block per workgroup in every packed forward (the form of rounds 3-4; A/B switch of bench and tests)
static bool pack_fwd_loop() {
    static const bool on = env_int_fwd("RRL_PACK_FWD_LOOP", 1) != 0;
    return on;
}

}  // namespace

extern "C" {

int rrl_mlp3_is_split(int M, int H) {
    (void)M;
    return ((H % (16 * kSplit)) == 0 && H <= kStackMaxH) ? kSplit : 0;
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
    StackArgs a{x, W1, b1, W2, b2, W3, b3, h1, h2, out, M, H, din, dout, ldx, rrl_policy_head_t{}, 0, nullptr};
    if (scratch && rrl_mlp3_is_split(M, H)) {
        // 4 workgroups (column groups) per row tile + fixed-order sum of their partial last-layer outputs
        if (M <= kSplitSmallM || H != 256) {
            hipLaunchKernelGGL(mlp3_fwd_split_kernel<1>, dim3((M + kStackRows - 1) / kStackRows, G, kSplit), dim3(256),
                               split_lds_floats(1) * 4, (hipStream_t)stream, a, scratch);
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
// scratch = finalize 0) or all on the same non-split tiling.  One exception (round 6): small-batch members of hidden width
// 256 may share a launch with a member on the multi-row tiles (an acting-pass forward riding with an update's forwards,
// fast_update.FastUpdater.qrisk_update_grouped): path 5, the flat grid on which every member keeps the tiles of its
// stand-alone launch (mlp3_fwd_split_flat_group_kernel); in a packed launch (flat = false) the small members take the multi-row
// tiles too (small_r = big_r, as from three seeds on) -- the rows of a tile are independent, the results are the one-row-tile
// launch's bit for bit.
static int build_stack_group(int n, const rrl_stack_t* st, StackGroup& sg, int& path, int big_r = kBigR, int small_r = 1,
                             int loop_nb = 1, bool flat = true) {
    if (!st || n <= 0 || n > kMaxGroup) return RRL_EINVAL;
    sg = StackGroup{};
    sg.n = n;
    sg.first[0] = 0;
    path = -1;   // 0 split (small batch), 3 split (R = kBigR row tiles); 1 plain R = 1, 2 plain R = 2
    bool any_big = false, all_split256 = true;
    for (int k = 0; k < n; ++k) {
        const bool split = st[k].scratch && rrl_mlp3_is_split(st[k].M, st[k].H);
        all_split256 = all_split256 && split && st[k].H == 256;
        any_big = any_big || (split && st[k].H == 256 && st[k].M > kSplitSmallM);
    }
    bool any_small = false;
    for (int k = 0; k < n; ++k) any_small = any_small || st[k].M <= kSplitSmallM;
    const bool mixed = any_big && any_small && all_split256;
    if (mixed && !flat) small_r = big_r;              // packed launches: every member on the multi-row tiles
    for (int k = 0; k < n; ++k) {
        const rrl_stack_t& p = st[k];
        const int rc = stack_check(p.G, p.M, p.H, p.din, p.dout, p.x, p.W1, p.b1, p.W2, p.b2, p.W3, p.b3, p.out);
        if (rc != RRL_OK) return rc;
        sg.a[k] = StackArgs{p.x, p.W1, p.b1, p.W2, p.b2, p.W3, p.b3, p.h1, p.h2, p.out, p.M, p.H, p.din, p.dout, p.ldx,
                            p.in_head, p.use_in_head, p.H == 256 ? p.W2p : nullptr};
        if (sg.a[k].W2p && (reinterpret_cast<uintptr_t>(p.W2p) & 15)) return RRL_EINVAL;
        if (p.use_in_head) {
            const rrl_policy_head_t& h = p.in_head;
            if (p.din != 4 || !h.head || !h.scale || !h.bias || h.n_part <= 0 || h.n_part > 4 ||
                (h.kind == RRL_HEAD_GAUSS ? !h.eps : (h.kind != RRL_HEAD_STOCH || !h.log_std)) || (h.obs_out && !h.action))
                return RRL_EINVAL;
            if (!(p.scratch && rrl_mlp3_is_split(p.M, p.H))) return RRL_EINVAL;   // the column-split kernels only
        }
        sg.partial[k] = p.scratch;
        sg.G[k] = p.G;
        sg.nb[k] = 1;
        int my;
        const long long tiles16 = (p.M + kStackRows - 1) / kStackRows;
        if (p.scratch && rrl_mlp3_is_split(p.M, p.H)) {
            my = (p.M <= kSplitSmallM || p.H != 256) ? 0 : 3;      // the multi-row tiles are built for H = 256
            if (my == 0 && small_r > 1 && p.H != 256) return RRL_EINVAL;
            const int rows = (my == 0 ? small_r : big_r) * kStackRows;
            if (mixed) my = 5;       // 5 = path 3 on the flat grid (rrl_mlp3_forward_multi; the packed builder reads it as 3)
            sg.tiles[k] = (p.M + rows - 1) / rows;
            sg.nb[k] = loop_nb < sg.tiles[k] ? loop_nb : sg.tiles[k];
            sg.first[k + 1] = sg.first[k] + (sg.tiles[k] + sg.nb[k] - 1) / sg.nb[k] * p.G * kSplit;
        } else if (tiles16 * p.G > 256) {
            my = 2;
            sg.tiles[k] = (p.M + 2 * kStackRows - 1) / (2 * kStackRows);
            sg.first[k + 1] = sg.first[k] + sg.tiles[k] * p.G;
        } else {
            my = 1;
            sg.tiles[k] = int(tiles16);
            sg.first[k + 1] = sg.first[k] + sg.tiles[k] * p.G;
        }
        if (path >= 0 && my != path) return RRL_EINVAL;
        path = my;
    }
    for (int k = n; k < kMaxGroup; ++k) sg.first[k + 1] = sg.first[n];
    return RRL_OK;
}

int rrl_mlp3_forward_multi(int n, const rrl_stack_t* st, void* stream) {
    StackGroup sg;
    int path;
    const int rc = build_stack_group(n, st, sg, path);
    if (rc != RRL_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int most = largest_member(sg, n);       // split paths: grid (workgroups of the largest member, members)
    if (path == 0) {
        hipLaunchKernelGGL(mlp3_fwd_split_group_kernel<1>, dim3(most, n), dim3(256), split_lds_floats(1) * 4, s, sg);
    } else if (path == 3) {
        static const bool ok = grant_lds((const void*)mlp3_fwd_split_group_kernel<kBigR>, split_lds_floats(kBigR) * 4);
        if (!ok) return RRL_ERANGE;
        hipLaunchKernelGGL(mlp3_fwd_split_group_kernel<kBigR>, dim3(most, n), dim3(256),
                           split_lds_floats(kBigR) * 4, s, sg);
    } else if (path == 5) {
        constexpr size_t lds_floats = split_lds_floats(kBigR) > split_lds_floats(1) ? split_lds_floats(kBigR) : split_lds_floats(1);
        static const bool ok = grant_lds((const void*)mlp3_fwd_split_flat_group_kernel, lds_floats * 4);
        if (!ok) return RRL_ERANGE;
        hipLaunchKernelGGL(mlp3_fwd_split_flat_group_kernel, dim3(sg.first[n]), dim3(256), lds_floats * 4, s, sg);
    } else if (path == 1) hipLaunchKernelGGL((mlp3_fwd_group_kernel<1>), dim3(sg.first[n]), dim3(1024), 0, s, sg);
    else hipLaunchKernelGGL((mlp3_fwd_group_kernel<2>), dim3(sg.first[n]), dim3(1024), 0, s, sg);
    return check_launch();
}

// the column-split kernels only (what the steady-state iteration launches at H = 256); every seed on the same path
int rrl_mlp3_forward_multi_packed(int S, const int* n, const rrl_stack_t* const* members, void* stream) {
    rrl_pack::Key key;
    if (!pack_key(2, S, n, members, key)) return RRL_EINVAL;
    // one seed: the packed launch IS the solo launch (argument block in the kernel arguments, no plan)
    if (S == 1) {
        StackGroup sg;
        int path;
        const int rc = build_stack_group(n[0], members[0], sg, path);
        if (rc != RRL_OK) return rc;
        if (path != 0 && path != 3 && path != 5) return RRL_EINVAL;          // the packed entry covers the column-split path only
        return rrl_mlp3_forward_multi(n[0], members[0], stream);
    }
    hipStream_t st = (hipStream_t)stream;
    rrl_pack::Plan* plan = rrl_pack::lookup(key);
    if (!plan) {
        std::vector<StackGroup> groups;
        rrl_pack::Idx ix;
        int path = -1;
        const int big_r = kBigR;      // (4 row tiles per workgroup measured at S = 4, 8: not faster, profiles/patches/README.md)
        // small batches (the updates' B = 256 forwards): kBigR row tiles per workgroup from kPackSmallR2MinSeeds seeds
        // on, when every member has the hidden width the multi-row tiles are built for
        bool all256 = true;
        for (int s = 0; s < S; ++s)
            for (int k = 0; k < n[s]; ++k) all256 = all256 && members[s] && members[s][k].H == 256;
        const int small_r = (S >= kPackSmallR2MinSeeds && all256) ? big_r : 1;
        int loop_nb = 1;
        const auto build_all = [&]() {
            path = -1;
            return build_pack<StackGroup>(S, n, members, groups, ix, [&](int nk, const rrl_stack_t* m, StackGroup& g) {
                int my;
                const int r = build_stack_group(nk, m, g, my, big_r, small_r, loop_nb, false);
                if (r != RRL_OK) return r;
                if (my == 5) my = 3;          // mixed members: multi-row tiles for all of them (the placement is the pack's)
                if ((my != 0 && my != 3) || (path >= 0 && my != path)) return int(RRL_EINVAL);
                path = my;
                return int(RRL_OK);
            });
        };
        int rc = build_all();
        if (rc != RRL_OK) return rc;
        // more row blocks than workgroups fit on the chip at once (4 per CU: 35 KB of LDS, 128 registers): workgroups that keep
        // their weights for nb consecutive row blocks, nb the smallest count with which all of them are resident together
        // (the acting pass's 4096-row forwards; the updates' 256-row forwards have 8 row blocks per column group and gain nothing)
        if (path == 3 && all256 && pack_fwd_loop()) {
            // pinned seeds (pack.hpp) have a share of the chip each: every seed's workgroups must fit its share
            const int share = rrl_pack::seed_share(S);
            const auto fits = [&]() {
                long long all = 0;
                for (int s = 0; s < S; ++s) {
                    all += groups[s].first[n[s]];
                    if (share > 1 && (long long)groups[s].first[n[s]] * share > kResidentWorkgroups) return false;
                }
                return all <= kResidentWorkgroups;
            };
            while (!fits() && loop_nb < kLoopMaxBlocks) {
                ++loop_nb;
                rc = build_all();
                if (rc != RRL_OK) return rc;
            }
            // few blocks per workgroup: the uneven last round of the one-block form costs less than three (instead of four)
            // workgroups per CU and the coarser grain do (2 / 4 seeds: 63.0 / 109.8 us for the acting pass's two launches against
            // 57.8 / 106.9; 8 / 16 seeds: 189.6 / 365.0 against 199.5 / 386.5 -- profiles/round5_fwd_packed/)
            if (loop_nb < kLoopMinBlocks) {
                loop_nb = 1;
                rc = build_all();
                if (rc != RRL_OK) return rc;
            }
        }
        if (path == 3 || (small_r > 1 && path == 0)) {
            static const bool ok = grant_lds((const void*)mlp3_fwd_split_pack_kernel<kBigR>, split_lds_floats(kBigR) * 4);
            static const bool ok2 = grant_lds((const void*)mlp3_fwd_split_pack_loop_kernel<kBigR>, loop_lds_floats(kBigR) * 4);
            if (!ok || !ok2) return RRL_ERANGE;
            path = loop_nb > 1 ? 4 : 3;   // small members on multi-row tiles run the large-batch kernel; 4 = its loop form
        }
        // 2-D grid: a seed owns as many workgroups per member row as its largest member has
        int most[rrl_pack::kMaxSeeds], members_most = 1;
        for (int s = 0; s < S; ++s) {
            most[s] = largest_member(groups[s], n[s]);
            members_most = n[s] > members_most ? n[s] : members_most;
        }
        plan = rrl_pack::store(key, groups.data(), sizeof(StackGroup) * S, st);
        if (!plan) return rrl_pack::store_error();
        plan->grid = finish_members(ix, S, most);
        plan->ix = ix;
        plan->i0 = path;
        plan->i1 = members_most;
    }
    if (plan->i0 == 0)
        hipLaunchKernelGGL(mlp3_fwd_split_pack_kernel<1>, dim3(plan->grid, plan->i1), dim3(256), split_lds_floats(1) * 4, st,
                           (const StackGroup*)plan->dev, plan->ix);
    else if (plan->i0 == 4)
        hipLaunchKernelGGL(mlp3_fwd_split_pack_loop_kernel<kBigR>, dim3(plan->grid, plan->i1), dim3(256),
                           loop_lds_floats(kBigR) * 4, st, (const StackGroup*)plan->dev, plan->ix);
    else
        hipLaunchKernelGGL(mlp3_fwd_split_pack_kernel<kBigR>, dim3(plan->grid, plan->i1), dim3(256),
                           split_lds_floats(kBigR) * 4, st, (const StackGroup*)plan->dev, plan->ix);
    return check_launch();
}

}  // extern "C"
