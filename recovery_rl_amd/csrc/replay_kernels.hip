// replay_kernels.hip -- device-resident replay buffers for gfx950 (MI355X).
//
// Replaces recovery_rl/replay_memory.py (python list of tuples + random.sample + np.stack +
// five host->device copies per batch) with a structure-of-arrays ring in HBM:
//   push           : n rows x 32 B, coalesced (lane i -> slot pos+i), one launch
//   sample+gather  : ONE workgroup draws B distinct slots (Philox + LDS all-pairs dedupe) and
//                    gathers the rows into five contiguous batch tensors -- 16 KB moved for
//                    B=256; latency-bound by design (DESIGN.md "replay")
//   stratified     : per-64-slot positive counts maintained by push; the sampler scans the
//                    count table in LDS and touches 64 rewards per drawn row instead of the
//                    reference's O(capacity) argwhere per call (replay_memory.py:58-66)
#include <hip/hip_runtime.h>

#include "replay_device.hpp"
#include "rrl_device.hpp"
#include "pack.hpp"
#include "rrl_host.hpp"

namespace {

using rrl_host::check_launch;
using rrl_host::grid_for;
using rrl_host::kBlock;

constexpr int kTile = 1024;   // rows per workgroup in the masked push
using rrl_replay::advance_ring;
using rrl_replay::kChunk;

struct Rows {
    const float2* s;
    const float2* a;
    const float* r;
    const float2* s2;
    const float* m;
};

__device__ __forceinline__ void store_row(const rrl_replay_t& rb, int64_t slot, int64_t size,
                                          const Rows& in, int64_t i) {
    rrl_replay::store_values(rb, slot, rrl_replay::was_positive(rb, slot, size), in.s[i], in.a[i], in.r[i], in.s2[i], in.m[i]);
}

__global__ __launch_bounds__(kBlock) void push_kernel(rrl_replay_t rb, int64_t n, Rows in) {
    const int64_t pos = rb.state[0], size = rb.state[1];
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride)
        store_row(rb, rrl_replay::ring_slot(rb, pos, i), size, in, i);
    advance_ring(rb, pos, size, n);
}

// masked push, pass 1: valid rows per 1024-row tile
__global__ __launch_bounds__(kBlock) void mask_count_kernel(const uint8_t* valid, int64_t n,
                                                            int32_t* tile_cnt) {
    __shared__ int32_t wave_cnt[kBlock / 64];
    const int64_t base = int64_t(blockIdx.x) * kTile;
    int32_t c = 0;
    for (int k = 0; k < kTile / kBlock; ++k) {
        const int64_t i = base + k * kBlock + threadIdx.x;
        c += (i < n && valid[i]) ? 1 : 0;
    }
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
    if ((threadIdx.x & 63) == 0) wave_cnt[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        int32_t tot = 0;
        for (int w = 0; w < kBlock / 64; ++w) tot += wave_cnt[w];
        tile_cnt[blockIdx.x] = tot;
    }
}

// masked push, pass 2: rows keep their order; tile offsets come from the pass-1 counts
__global__ __launch_bounds__(kBlock) void push_masked_kernel(rrl_replay_t rb, int64_t n, Rows in,
                                                             const uint8_t* valid,
                                                             const int32_t* tile_cnt) {
    __shared__ int64_t red[kBlock];
    __shared__ int32_t wave_off[kBlock / 64];
    const int64_t pos = rb.state[0], size = rb.state[1];
    // exclusive offset of this tile and the grand total
    int64_t before = 0, total = 0;
    for (int j = threadIdx.x; j < int(gridDim.x); j += kBlock) {
        const int32_t c = tile_cnt[j];
        total += c;
        if (j < int(blockIdx.x)) before += c;
    }
    red[threadIdx.x] = before;
    __syncthreads();
    for (int off = kBlock / 2; off > 0; off >>= 1) {
        if (int(threadIdx.x) < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    before = red[0];
    __syncthreads();
    red[threadIdx.x] = total;
    __syncthreads();
    for (int off = kBlock / 2; off > 0; off >>= 1) {
        if (int(threadIdx.x) < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    total = red[0];
    __syncthreads();
    int64_t run = before;
    const int64_t base = int64_t(blockIdx.x) * kTile;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = 0; k < kTile / kBlock; ++k) {
        const int64_t i = base + k * kBlock + threadIdx.x;
        const bool v = (i < n) && valid[i];
        const unsigned long long bal = __ballot(v);
        const int rank_in_wave = __popcll(bal & ((1ULL << lane) - 1ULL));
        if (lane == 0) wave_off[wave] = __popcll(bal);
        __syncthreads();
        int32_t woff = 0, tile_tot = 0;
        for (int w = 0; w < kBlock / 64; ++w) {
            const int32_t c = wave_off[w];
            if (w < wave) woff += c;
            tile_tot += c;
        }
        if (v) store_row(rb, rrl_replay::ring_slot(rb, pos, run + woff + rank_in_wave), size, in, i);
        run += tile_tot;
        __syncthreads();
    }
    advance_ring(rb, pos, size, total);
}

// ---- sampling -------------------------------------------------------------------------------
struct BatchOut {
    float2* s;
    float2* a;
    float* r;
    float2* s2;
    float* m;
    int64_t* idx;
    float4* xu;    // optional [B,4] rows (s, a): the critics' input, written by the gather itself
    float4* x2u;   // optional [B,4] rows (s', *, *): columns 2..3 are left for the policy head kernel
    float4* xpu;   // optional [B,4] rows (s,  *, *)
};

// B distinct draws per group from [0, population): each slot draws independently; a slot loses
// a round when an accepted slot, or a lower-numbered pending slot of its group, holds the same
// value, and redraws with the round number bumped (uniform over ordered subsets by symmetry).
// cand / acc live in LDS.  Returns false if the round cap is hit.
// Implementation: per round an LDS hash table maps value -> lowest claiming tag (accepted lanes claim
// with tag 0, pending lane i with tag i + 1) through 64-bit atomicMin on (value << 32 | tag); the table
// content that matters (minimum tag per value) does not depend on insertion order, so the outcome is the
// same as the sequential all-pairs rule of the CPU checker.  O(B) work per round instead of O(B^2).
// `group` (0/1) keeps the two populations of the stratified sampler apart.  key[i] = value | accepted << 31.
__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// `whole`: this lane's class is taken whole (as many rows requested as it has -- what the clamped stratified draw does to
// a starved class): its lanes get the ranks 0, 1, ... in lane order instead of drawing.  (Drawing n distinct values out
// of n by rejection needs O(n) rounds: 50 us for 69 positives in the Maze loop.)
__device__ __forceinline__ bool draw_distinct(int i, int B, int group, uint64_t population, uint64_t seed,
                                              uint32_t stream, uint64_t ctr, int row, uint32_t* key,
                                              unsigned long long* table, int table_mask, bool whole = false) {
    const unsigned long long kEmpty = ~0ULL;
    const bool active = i < B;
    bool mine = !active;  // inactive lanes count as settled
    uint32_t v = 0;
    if (active && whole) {
        mine = true;
        v = uint32_t(row);
        key[i] = v | 0x80000000u;
    }
    for (uint32_t round = 0; round <= 4096; ++round) {
        for (int e = threadIdx.x; e <= table_mask; e += blockDim.x) table[e] = kEmpty;
        if (active && !mine) {
            const rrl::Bits128 b = rrl::philox_at(seed, uint32_t(row), stream, (ctr << 12) | round);
            v = uint32_t(__umul64hi(b.lo, population));
        }
        __syncthreads();
        const uint32_t hv = v | (uint32_t(group) << 31);            // value tagged with its population
        if (active) {                                                // claim: accepted lanes with tag 0
            const unsigned long long pack = ((unsigned long long)hv << 32) | (mine ? 0u : uint32_t(i + 1));
            uint32_t h = hash32(hv) & table_mask;
            for (;;) {
                unsigned long long cur = table[h];
                if (cur == kEmpty) {
                    cur = atomicCAS(&table[h], kEmpty, pack);
                    if (cur == kEmpty) break;
                }
                if (uint32_t(cur >> 32) == hv) {
                    atomicMin(&table[h], pack);
                    break;
                }
                h = (h + 1) & table_mask;
            }
        }
        __syncthreads();
        if (active && !mine) {
            uint32_t h = hash32(hv) & table_mask;
            while (uint32_t(table[h] >> 32) != hv) h = (h + 1) & table_mask;
            if (uint32_t(table[h]) == uint32_t(i + 1)) {             // lowest claimant of this value: accepted
                mine = true;
                key[i] = v | 0x80000000u;
            }
        }
        if (__syncthreads_count(!mine) == 0) return true;
    }
    return false;
}

__device__ __forceinline__ void gather_row(const rrl_replay_t& rb, int64_t slot, int i,
                                           const BatchOut& out) {
    const float2 s = ((const float2*)rb.s)[slot], a = ((const float2*)rb.a)[slot];
    const float2 s2 = ((const float2*)rb.s2)[slot];
    out.s[i] = s;
    out.a[i] = a;
    out.r[i] = rb.r[slot];
    out.s2[i] = s2;
    out.m[i] = rb.m[slot];
    if (out.idx) out.idx[i] = slot;
    if (out.xu) out.xu[i] = make_float4(s.x, s.y, a.x, a.y);
    if (out.x2u) ((float2*)out.x2u)[2 * i] = s2;
    if (out.xpu) ((float2*)out.xpu)[2 * i] = s;
}

__device__ __forceinline__ void sample_gather_body(const rrl_replay_t& rb, int B, uint64_t seed, uint64_t counter,
                                                   uint64_t* counter_dev, uint64_t counter_inc, int table_mask,
                                                   const BatchOut& out, char* smem) {
    unsigned long long* table = (unsigned long long*)smem;      // [table_mask + 1]
    uint32_t* key = (uint32_t*)(table + table_mask + 1);
    const int64_t size = rb.state[1];
    const uint64_t ctr = rrl::effective_counter(counter, counter_dev);      // requested with the size, not behind its test
    if (int64_t(B) > size) {  // random.sample would raise ValueError
        if (threadIdx.x == 0) rb.state[3] = 1;
        return;
    }
    rrl::advance_counter_single(counter_dev, counter_inc, counter, ctr);        // one workgroup per draw
    const int i = threadIdx.x;
    if (!draw_distinct(i, B, 0, uint64_t(size), seed, rrl::kStreamSample, ctr, i, key, table, table_mask)) {
        if (threadIdx.x == 0) rb.state[3] = 2;
        return;
    }
    if (i < B) gather_row(rb, int64_t(key[i] & 0x7fffffffu), i, out);
}

__global__ __launch_bounds__(1024) void sample_gather_kernel(rrl_replay_t rb, int B, uint64_t seed,
                                                             uint64_t counter,
                                                             uint64_t* counter_dev,
                                                             uint64_t counter_inc, int table_mask,
                                                             BatchOut out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    sample_gather_body(rb, B, seed, counter, counter_dev, counter_inc, table_mask, out, smem);
}

// Stratified: lanes [0,n_pos) draw ranks among positives, lanes [n_pos,B) among negatives.
__device__ __forceinline__ void creplay_sample_gather_body(const rrl_replay_t& rb, int n_pos, int n_neg, int n_chunks,
                                                           uint64_t seed, uint64_t counter, uint64_t* counter_dev,
                                                           uint64_t counter_inc, int table_mask, const BatchOut& out,
                                                           char* smem) {
    const int B = n_pos + n_neg;
    unsigned long long* table = (unsigned long long*)smem;      // [table_mask + 1]
    uint32_t* key = (uint32_t*)(table + table_mask + 1);
    int32_t* sup = (int32_t*)(key + ((B + 3) & ~3));      // [n_super + 1] exclusive positive counts per super-chunk
    const int64_t size = rb.state[1];
    const int tid = threadIdx.x;
    // Second count level (one entry per 1024 slots, <= 2048 of them) -> exclusive scan in LDS.  (The first version
    // copied and scanned the whole first level -- 15 625 entries at 1e6 slots, 62 KB -- in this one workgroup: 27 us.)
    const int n_super = int(rrl_replay::count_supers(rb.cap));
    const int32_t* sup_cnt = rb.pos_cnt + rrl_replay::super_base(rb.cap);
    // wave 0: lane l owns entries 32 l .. 32 l + 31 (cap <= 2^21: at most 2048 entries), wave prefix by shuffles
    if (tid < 64) {
        constexpr int kOwn = 32;
        int32_t v[kOwn], run = 0;
        const bool vec = (reinterpret_cast<uintptr_t>(sup_cnt) & 15) == 0;
#pragma unroll
        for (int q = 0; q < kOwn / 4; ++q) {             // eight independent 16-byte loads per lane, in flight together
            const int c = kOwn * tid + 4 * q;
            if (vec && c + 3 < n_super) {
                const int4 t4 = *reinterpret_cast<const int4*>(sup_cnt + c);
                v[4 * q] = t4.x; v[4 * q + 1] = t4.y; v[4 * q + 2] = t4.z; v[4 * q + 3] = t4.w;
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) v[4 * q + u] = c + u < n_super ? sup_cnt[c + u] : 0;
            }
        }
#pragma unroll
        for (int u = 0; u < kOwn; ++u) run += v[u];
        int32_t incl = run;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int32_t up = __shfl_up(incl, off, 64);
            if (tid >= off) incl += up;
        }
        int32_t acc = incl - run;                      // exclusive prefix of this lane's first entry
#pragma unroll
        for (int u = 0; u < kOwn; ++u) {
            const int c = kOwn * tid + u;
            if (c <= n_super) sup[c] = acc;
            acc += v[u];
        }
        if (kOwn * tid + kOwn == n_super) sup[n_super] = acc;     // total, when n_super is a multiple of kOwn
    }
    __syncthreads();
    const int64_t total_pos = sup[n_super];
    const int64_t total_neg = size - total_pos;
    if (int64_t(n_pos) > total_pos || int64_t(n_neg) > total_neg) {
        const bool feasible = int64_t(B) <= size && (rb.flags & RRL_REPLAY_CLAMP_STRATIFIED);
        if (!feasible) {
            if (tid == 0) rb.state[3] = 1;
            return;
        }
        // every row of the short class, the rest of the batch from the other one (uniform for all threads)
        if (int64_t(n_pos) > total_pos) n_pos = int(total_pos);
        else n_pos = B - int(total_neg);
        n_neg = B - n_pos;
    }
    const uint64_t ctr = rrl::effective_counter(counter, counter_dev);
    rrl::advance_counter_single(counter_dev, counter_inc, counter, ctr);
    const bool is_pos = tid < n_pos;
    const uint64_t population = uint64_t(is_pos ? total_pos : total_neg);
    const uint32_t stream = is_pos ? rrl::kStreamSample : rrl::kStreamSampleNeg;
    // lanes of the negative group are numbered from 0 within their group, like a separate call
    const int gi = is_pos ? tid : tid - n_pos;
    const bool class_whole = population == uint64_t(is_pos ? n_pos : n_neg);
    if (!draw_distinct(tid, B, is_pos ? 0 : 1, population, seed, stream, ctr, gi, key, table, table_mask, class_whole)) {
        if (tid == 0) rb.state[3] = 2;
        return;
    }
    if (tid >= B) return;
    // rank -> slot: binary search the super-chunk in LDS, walk its 64 first-level counts (16 independent 16-byte loads),
    // then scan the chunk's 64 rewards
    const int64_t k = int64_t(key[tid] & 0x7fffffffu);
    auto before_super = [&](int sc) -> int64_t {  // rows of my class in super-chunks [0, sc)
        const int64_t filled = min(size, int64_t(sc) * rrl_replay::kSuper);
        return is_pos ? int64_t(sup[sc]) : filled - int64_t(sup[sc]);
    };
    int sa = 0, sb = n_super;  // invariant: before_super(sa) <= k < before_super(sb)
    while (sb - sa > 1) {
        const int mid = (sa + sb) >> 1;
        if (before_super(mid) <= k) sa = mid; else sb = mid;
    }
    constexpr int kPer = rrl_replay::kSuper / kChunk;     // 16 chunks per super-chunk
    const int c_first = sa * kPer;
    int4 cv[kPer / 4];
    {
        const int4* src = reinterpret_cast<const int4*>(rb.pos_cnt + c_first);   // c_first % 4 == 0, table 16-byte aligned
        const bool all = c_first + kPer <= n_chunks && (reinterpret_cast<uintptr_t>(rb.pos_cnt) & 15) == 0;
#pragma unroll
        for (int q = 0; q < kPer / 4; ++q) {
            if (all) {
                cv[q] = src[q];
            } else {
                int t4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) t4[u] = c_first + 4 * q + u < n_chunks ? rb.pos_cnt[c_first + 4 * q + u] : 0;
                cv[q] = make_int4(t4[0], t4[1], t4[2], t4[3]);
            }
        }
    }
    // From here on everything is relative to the super-chunk / chunk and fits 32 bits (the 64-bit version of these two
    // 64-step walks was 20 of the kernel's 26 us: ~1300 emulated-int64 instructions per lane on a single CU).
    int32_t rem = int32_t(k - before_super(sa));                          // rank inside the super-chunk, < 1024
    const int64_t sup_lo = int64_t(c_first) * kChunk;
    const int32_t filled_sup = int32_t(min(int64_t(rrl_replay::kSuper), max(int64_t(0), size - sup_lo)));
    int32_t a_rel = -1;
#pragma unroll
    for (int q = 0; q < kPer / 4; ++q) {
        const int e4[4] = {cv[q].x, cv[q].y, cv[q].z, cv[q].w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int cu = 4 * q + u;
            const int32_t filled = min(max(filled_sup - cu * kChunk, 0), kChunk);      // filled slots of this chunk
            const int32_t mine = is_pos ? e4[u] : filled - e4[u];
            const bool take = (a_rel < 0) & (rem < mine);
            a_rel = take ? cu : a_rel;
            rem = a_rel < 0 ? rem - mine : rem;
        }
    }
    if (a_rel < 0) {          // the two count levels disagree
        rb.state[3] = 3;
        return;
    }
    const int a = c_first + a_rel;
    const int64_t c0 = int64_t(a) * kChunk;
    // the chunk's bit mask instead of its 64 rewards (256 B and a 64-step compare chain per row: 6 of the kernel's 19 us):
    // the rem-th set bit of the class's candidates, by popcounts of halves
    const int32_t filled_c = int32_t(min(int64_t(kChunk), max(int64_t(0), size - c0)));
    const unsigned long long pos_bits = rrl_replay::chunk_masks(rb)[a];
    const unsigned long long filled_bits = filled_c >= kChunk ? ~0ULL : ((1ULL << filled_c) - 1ULL);
    unsigned long long cand = is_pos ? (pos_bits & filled_bits) : (filled_bits & ~pos_bits);
    int32_t slot_rel = -1;
    if (rem < __popcll(cand)) {
        int32_t at = 0;
#pragma unroll
        for (int w = 32; w >= 1; w >>= 1) {
            const unsigned long long low = cand & ((1ULL << w) - 1ULL);
            const int32_t c = __popcll(low);
            const bool upper = rem >= c;
            rem -= upper ? c : 0;
            cand = upper ? (cand >> w) : low;
            at += upper ? w : 0;
        }
        slot_rel = at;
    }
    const int64_t slot = slot_rel < 0 ? int64_t(-1) : c0 + slot_rel;
    if (slot < 0) {  // count table out of sync with the rows: flag, never read out of bounds
        rb.state[3] = 3;
        return;
    }
    gather_row(rb, slot, tid, out);
}

__global__ __launch_bounds__(1024) void creplay_sample_gather_kernel(rrl_replay_t rb, int n_pos,
                                                                     int n_neg, int n_chunks,
                                                                     uint64_t seed, uint64_t counter,
                                                                     uint64_t* counter_dev,
                                                                     uint64_t counter_inc, int table_mask,
                                                                     BatchOut out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    creplay_sample_gather_body(rb, n_pos, n_neg, n_chunks, seed, counter, counter_dev, counter_inc, table_mask, out,
                               smem);
}

// Demonstration-share draw (the lock-step loop's rule for the safety critic's batch, DESIGN "replay"): lanes [0, n_demo)
// draw distinct rows of the pinned range [0, pinned) -- the offline constraint demonstrations, experiment.py:278-286 --
// lanes [n_demo, B) distinct rows of the online range [pinned, size).  A range with too few rows gives all it has and
// the other one fills the batch.  Ranks ARE slots here (demo rank k = slot k, online rank k = slot pinned + k).
__device__ __forceinline__ void split_sample_gather_body(const rrl_replay_t& rb, int n_demo, int n_online, uint64_t seed,
                                                         uint64_t counter, uint64_t* counter_dev, uint64_t counter_inc,
                                                         int table_mask, const BatchOut& out, char* smem) {
    const int B = n_demo + n_online;
    unsigned long long* table = (unsigned long long*)smem;      // [table_mask + 1]
    uint32_t* key = (uint32_t*)(table + table_mask + 1);
    const int64_t size = rb.state[1];
    const int tid = threadIdx.x;
    if (int64_t(B) > size) {  // random.sample would raise ValueError
        if (tid == 0) rb.state[3] = 1;
        return;
    }
    const int64_t demo_total = min(rb.pinned, size), online_total = size - demo_total;
    if (int64_t(n_online) > online_total) { n_online = int(online_total); n_demo = B - n_online; }
    else if (int64_t(n_demo) > demo_total) { n_demo = int(demo_total); n_online = B - n_demo; }
    const uint64_t ctr = rrl::effective_counter(counter, counter_dev);
    rrl::advance_counter_single(counter_dev, counter_inc, counter, ctr);
    const bool is_demo = tid < n_demo;
    const uint64_t population = uint64_t(is_demo ? demo_total : online_total);
    const uint32_t stream = is_demo ? rrl::kStreamSample : rrl::kStreamSampleNeg;
    const int gi = is_demo ? tid : tid - n_demo;              // numbered from 0 within the group, like a separate call
    const bool whole = population == uint64_t(is_demo ? n_demo : n_online);
    if (!draw_distinct(tid, B, is_demo ? 0 : 1, population, seed, stream, ctr, gi, key, table, table_mask, whole)) {
        if (tid == 0) rb.state[3] = 2;
        return;
    }
    if (tid >= B) return;
    const int64_t k = int64_t(key[tid] & 0x7fffffffu);
    gather_row(rb, is_demo ? k : demo_total + k, tid, out);
}

__global__ __launch_bounds__(1024) void split_sample_gather_kernel(rrl_replay_t rb, int n_demo, int n_online,
                                                                   uint64_t seed, uint64_t counter,
                                                                   uint64_t* counter_dev, uint64_t counter_inc,
                                                                   int table_mask, BatchOut out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    split_sample_gather_body(rb, n_demo, n_online, seed, counter, counter_dev, counter_inc, table_mask, out, smem);
}

// The two draws of one lock-step iteration (task buffer for the SAC update, safety buffer for the Q_risk update:
// experiment.py:397-416) and the iteration's policy noise do not depend on each other: one launch, workgroup 0 and 1
// are the samplers (exactly the stand-alone kernels' code), the remaining workgroups fill the noise buffer.
struct DrawArgs {
    rrl_replay_t rb;
    int mode;            // 0: none, 1: uniform (sample_gather), 2: stratified (creplay_sample_gather), 3: demo share (split_sample_gather)
    int B, n_pos, n_neg, n_chunks, table_mask;
    uint64_t seed, counter;
    uint64_t* counter_dev;
    uint64_t counter_inc;
    BatchOut out;
};
struct NoiseArgs {
    long long n_pairs;
    uint64_t seed, counter;
    uint64_t* counter_dev;
    uint64_t counter_inc;
    float* out;
    int blocks;
};

__device__ __forceinline__ void draw_body(const DrawArgs& d, char* smem) {
    if (d.mode == 1)
        sample_gather_body(d.rb, d.B, d.seed, d.counter, d.counter_dev, d.counter_inc, d.table_mask, d.out, smem);
    else if (d.mode == 2)
        creplay_sample_gather_body(d.rb, d.n_pos, d.n_neg, d.n_chunks, d.seed, d.counter, d.counter_dev, d.counter_inc,
                                   d.table_mask, d.out, smem);
    else if (d.mode == 3)
        split_sample_gather_body(d.rb, d.n_pos, d.n_neg, d.seed, d.counter, d.counter_dev, d.counter_inc, d.table_mask,
                                 d.out, smem);
}

__device__ __forceinline__ void sample_group_body(const DrawArgs& a, const DrawArgs& b, const NoiseArgs& nz, int block,
                                                  char* smem) {
    if (block == 0) { draw_body(a, smem); return; }
    if (block == 1) { draw_body(b, smem); return; }
    // N(0,1) pairs of Philox stream RRL_STREAM_NOISE (rrl_normal_fill)
    const uint64_t ctr = rrl::effective_counter(nz.counter, nz.counter_dev);
    const long long stride = (long long)nz.blocks * blockDim.x;
    for (long long i = (long long)(block - 2) * blockDim.x + threadIdx.x; i < nz.n_pairs; i += stride) {
        double z0, z1;
        rrl::normal_at(nz.seed, uint32_t(i), rrl::kStreamNoise, ctr, z0, z1);
        reinterpret_cast<float2*>(nz.out)[i] = make_float2(float(z0), float(z1));
    }
    rrl::advance_counter_blocks(nz.counter_dev, nz.counter_inc, unsigned(nz.blocks));
}

__global__ __launch_bounds__(1024) void sample_group_kernel(DrawArgs a, DrawArgs b, NoiseArgs nz) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    sample_group_body(a, b, nz, blockIdx.x, smem);
}

// the same launch for S seeds (pack.hpp)
struct SamplePack {
    DrawArgs a, b;
    NoiseArgs nz;
};
__global__ __launch_bounds__(1024) void sample_pack_kernel(const SamplePack* __restrict__ packs, rrl_pack::Idx ix) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int s, block;
    if (!rrl_pack::locate(ix, blockIdx.x, s, block)) return;
    // a sampler workgroup copies its own draw, a noise workgroup the noise block, out of device memory
    // (pointers that come out of device memory are passed through the global address space: rrl_pack::to_global)
    auto glob = [](DrawArgs& d) __attribute__((always_inline)) {
        rrl_pack::globalize(d.rb);
        rrl_pack::to_global_all(d.counter_dev, d.out.s, d.out.a, d.out.r, d.out.s2, d.out.m, d.out.idx, d.out.xu, d.out.x2u, d.out.xpu);
    };
    if (block == 0) { DrawArgs d = packs[s].a; glob(d); draw_body(d, smem); return; }
    if (block == 1) { DrawArgs d = packs[s].b; glob(d); draw_body(d, smem); return; }
    NoiseArgs nz = packs[s].nz;
    rrl_pack::to_global_all(nz.counter_dev, nz.out);
    sample_group_body(packs[s].a, packs[s].b, nz, block, smem);
}

inline bool valid_rb(const rrl_replay_t* rb) {
    return rb && rb->s && rb->a && rb->r && rb->s2 && rb->m && rb->state && rb->cap > 0;
}

}  // namespace

template <class K>
static bool grant_sample_lds(K kernel, size_t lds, size_t& granted) {
    if (lds > granted) {
        if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        granted = lds;
    }
    return true;
}

extern "C" {

int rrl_replay_push(const rrl_replay_t* rb, int64_t n, const float* s, const float* a,
                    const float* r, const float* s2, const float* m, const uint8_t* valid,
                    int32_t* scratch, void* stream) {
    if (!valid_rb(rb) || !s || !a || !r || !s2 || !m || n < 0) return RRL_EINVAL;
    if (rb->pinned < 0 || rb->pinned >= rb->cap || n > rb->cap - rb->pinned) return RRL_ERANGE;
    if (n == 0) return RRL_OK;
    const Rows in{(const float2*)s, (const float2*)a, r, (const float2*)s2, m};
    hipStream_t st = (hipStream_t)stream;
    if (!valid) {
        hipLaunchKernelGGL(push_kernel, dim3(grid_for(n)), dim3(kBlock), 0, st, *rb, n, in);
        return check_launch();
    }
    if (!scratch) return RRL_EINVAL;
    const int64_t tiles = (n + kTile - 1) / kTile;
    if (tiles > 65535 * 16) return RRL_ERANGE;
    hipLaunchKernelGGL(mask_count_kernel, dim3((unsigned)tiles), dim3(kBlock), 0, st, valid, n, scratch);
    hipLaunchKernelGGL(push_masked_kernel, dim3((unsigned)tiles), dim3(kBlock), 0, st, *rb, n, in,
                       valid, (const int32_t*)scratch);
    return check_launch();
}

// launch parameters of one draw (threads, dynamic LDS) shared by the stand-alone and the grouped entry points
static int draw_setup(const rrl_draw_t& d, DrawArgs& a, int& threads, size_t& lds) {
    const rrl_replay_t* rb = d.rb;
    if (!valid_rb(rb) || !d.s || !d.a || !d.r || !d.s2 || !d.m) return RRL_EINVAL;
    a.rb = *rb;
    a.seed = d.seed; a.counter = d.counter; a.counter_dev = d.counter_dev; a.counter_inc = d.counter_inc;
    a.out = BatchOut{(float2*)d.s, (float2*)d.a, d.r, (float2*)d.s2, d.m, d.idx_out, (float4*)d.xu, (float4*)d.x2u,
                     (float4*)d.xpu};
    const int B = d.n_pos + d.n_neg;
    if (d.n_pos < 0 || d.n_neg < 0 || B <= 0 || B > 1024) return RRL_ERANGE;
    a.B = B; a.n_pos = d.n_pos; a.n_neg = d.n_neg;
    int table_size = 64;
    while (table_size < 4 * B) table_size <<= 1;
    a.table_mask = table_size - 1;
    threads = ((B + 63) / 64) * 64;
    if (d.stratified == RRL_DRAW_UNIFORM || d.stratified == RRL_DRAW_DEMO_SHARE) {
        if (rb->cap >= (int64_t(1) << 31)) return RRL_ERANGE;
        if (d.stratified == RRL_DRAW_DEMO_SHARE && (rb->pinned < 0 || rb->pinned >= rb->cap)) return RRL_ERANGE;
        a.mode = d.stratified == RRL_DRAW_UNIFORM ? 1 : 3;
        a.n_chunks = 0;
        lds = size_t(table_size) * 8 + size_t(B) * 4 + 16;
        return RRL_OK;
    }
    if (d.stratified != RRL_DRAW_STRATIFIED) return RRL_EINVAL;
    if (!rb->pos_cnt) return RRL_EINVAL;
    if (rb->cap > (int64_t(1) << 21)) return RRL_ERANGE;
    a.mode = 2;
    a.n_chunks = int((rb->cap + kChunk - 1) / kChunk);
    if (threads < 64) threads = 64;
    lds = size_t(table_size) * 8 + size_t((B + 3) & ~3) * 4 + size_t(rrl_replay::count_supers(rb->cap) + 2) * 4 + 16;
    return RRL_OK;
}

static int build_sample(const rrl_draw_t* first, const rrl_draw_t* second, long long noise_pairs, uint64_t noise_seed,
                        uint64_t noise_counter, uint64_t* noise_counter_dev, uint64_t noise_counter_inc, float* noise_out,
                        DrawArgs& a, DrawArgs& b, NoiseArgs& nz, int& threads, size_t& lds) {
    if (!first) return RRL_EINVAL;
    if (noise_pairs < 0 || noise_pairs >= (1LL << 32) || (noise_pairs > 0 && !noise_out)) return RRL_EINVAL;
    a = DrawArgs{};
    b = DrawArgs{};
    int ta = 0, tb = 0;
    size_t la = 0, lb = 0;
    int rc = draw_setup(*first, a, ta, la);
    if (rc != RRL_OK) return rc;
    if (second) {
        rc = draw_setup(*second, b, tb, lb);
        if (rc != RRL_OK) return rc;
    }
    // every member's results are independent of the workgroup size (integer prefix sums, per-index Philox draws), so
    // the launch takes the largest thread count a member would use on its own
    threads = ta > tb ? ta : tb;
    if (noise_pairs > 0 && threads < 256) threads = 256;
    lds = la > lb ? la : lb;
    nz = NoiseArgs{noise_pairs, noise_seed, noise_counter, noise_counter_dev, noise_counter_inc, noise_out, 0};
    return RRL_OK;
}

static void noise_blocks(NoiseArgs& nz, int threads) {
    if (nz.n_pairs > 0) {
        long long nb = (nz.n_pairs + threads - 1) / threads;
        nz.blocks = int(nb < 1 ? 1 : (nb > 1024 ? 1024 : nb));
    }
}

int rrl_sample_multi(const rrl_draw_t* first, const rrl_draw_t* second, long long noise_pairs, uint64_t noise_seed,
                     uint64_t noise_counter, uint64_t* noise_counter_dev, uint64_t noise_counter_inc, float* noise_out,
                     void* stream) {
    DrawArgs a, b;
    NoiseArgs nz;
    int threads;
    size_t lds;
    const int rc = build_sample(first, second, noise_pairs, noise_seed, noise_counter, noise_counter_dev, noise_counter_inc,
                                noise_out, a, b, nz, threads, lds);
    if (rc != RRL_OK) return rc;
    static size_t granted = 64 * 1024;
    if (!grant_sample_lds(sample_group_kernel, lds, granted)) return RRL_ERANGE;
    noise_blocks(nz, threads);
    hipLaunchKernelGGL(sample_group_kernel, dim3(2 + nz.blocks), dim3(threads), lds, (hipStream_t)stream, a, b, nz);
    return check_launch();
}

static void key_draw(rrl_pack::Key& key, const rrl_draw_t* d) {
    key.pod(d != nullptr);
    if (d) {
        key.pod(*d);
        if (d->rb) key.pod(*d->rb);          // capacity / pinned rows / flags belong to the launch
    }
}

int rrl_sample_multi_packed(int S, const rrl_sample_args_t* args, void* stream) {
    if (S <= 0 || S > rrl_pack::kMaxSeeds || !args) return RRL_EINVAL;
    // one seed: the packed launch IS the solo launch (argument block in the kernel arguments, no plan)
    if (S == 1)
        return rrl_sample_multi(args[0].first, args[0].second, args[0].noise_pairs, args[0].noise_seed, args[0].noise_counter,
                                args[0].noise_counter_dev, args[0].noise_counter_inc, args[0].noise_out, stream);
    rrl_pack::Key key;
    key.pod(5);
    key.pod(S);
    for (int s = 0; s < S; ++s) {
        key_draw(key, args[s].first);
        key_draw(key, args[s].second);
        key.pod(args[s].noise_pairs); key.pod(args[s].noise_seed); key.pod(args[s].noise_counter);
        key.pod(args[s].noise_counter_dev); key.pod(args[s].noise_counter_inc); key.pod(args[s].noise_out);
    }
    hipStream_t st = (hipStream_t)stream;
    rrl_pack::Plan* plan = rrl_pack::lookup(key);
    if (!plan) {
        std::vector<SamplePack> packs(S);
        int threads = 0;
        size_t lds = 0;
        for (int s = 0; s < S; ++s) {
            const rrl_sample_args_t& g = args[s];
            int t;
            size_t l;
            const int rc = build_sample(g.first, g.second, g.noise_pairs, g.noise_seed, g.noise_counter, g.noise_counter_dev,
                                        g.noise_counter_inc, g.noise_out, packs[s].a, packs[s].b, packs[s].nz, t, l);
            if (rc != RRL_OK) return rc;
            threads = t > threads ? t : threads;
            lds = l > lds ? l : lds;
        }
        rrl_pack::Idx ix;
        ix.S = S;
        ix.first[0] = 0;
        for (int s = 0; s < S; ++s) {
            noise_blocks(packs[s].nz, threads);
            ix.first[s + 1] = ix.first[s] + 2 + packs[s].nz.blocks;
        }
        for (int s = S; s < rrl_pack::kMaxSeeds; ++s) ix.first[s + 1] = ix.first[S];
        static size_t granted = 64 * 1024;
        if (!grant_sample_lds(sample_pack_kernel, lds, granted)) return RRL_ERANGE;
        plan = rrl_pack::store(key, packs.data(), sizeof(SamplePack) * S, st);
        if (!plan) return rrl_pack::store_error();
        plan->grid = rrl_pack::finish(ix);
        plan->ix = ix;
        plan->i0 = threads;
        plan->z0 = lds;
    }
    hipLaunchKernelGGL(sample_pack_kernel, dim3(plan->grid), dim3(plan->i0), plan->z0, st,
                       (const SamplePack*)plan->dev, plan->ix);
    return check_launch();
}

int rrl_replay_sample_gather(const rrl_replay_t* rb, int32_t B, uint64_t seed, uint64_t counter,
                             uint64_t* counter_dev, uint64_t counter_inc, float* s, float* a, float* r, float* s2,
                             float* m, int64_t* idx_out, float* xu, float* x2u, float* xpu, void* stream) {
    if (!valid_rb(rb) || !s || !a || !r || !s2 || !m) return RRL_EINVAL;
    if (B <= 0 || B > 1024 || rb->cap >= (int64_t(1) << 31)) return RRL_ERANGE;
    const BatchOut out{(float2*)s, (float2*)a, r, (float2*)s2, m, idx_out, (float4*)xu, (float4*)x2u, (float4*)xpu};
    const int threads = ((B + 63) / 64) * 64;
    int table_size = 64;
    while (table_size < 4 * B) table_size <<= 1;
    const size_t lds = size_t(table_size) * 8 + size_t(B) * 4 + 16;
    hipLaunchKernelGGL(sample_gather_kernel, dim3(1), dim3(threads), lds, (hipStream_t)stream, *rb,
                       B, seed, counter, counter_dev, counter_inc, table_size - 1, out);
    return check_launch();
}

int rrl_replay_sample_gather_split(const rrl_replay_t* rb, int32_t n_demo, int32_t n_online, uint64_t seed,
                                   uint64_t counter, uint64_t* counter_dev, uint64_t counter_inc, float* s, float* a,
                                   float* r, float* s2, float* m, int64_t* idx_out, float* xu, float* x2u, float* xpu,
                                   void* stream) {
    if (!valid_rb(rb) || !s || !a || !r || !s2 || !m) return RRL_EINVAL;
    const int B = n_demo + n_online;
    if (n_demo < 0 || n_online < 0 || B <= 0 || B > 1024 || rb->cap >= (int64_t(1) << 31)) return RRL_ERANGE;
    if (rb->pinned < 0 || rb->pinned >= rb->cap) return RRL_ERANGE;
    const BatchOut out{(float2*)s, (float2*)a, r, (float2*)s2, m, idx_out, (float4*)xu, (float4*)x2u, (float4*)xpu};
    const int threads = ((B + 63) / 64) * 64;
    int table_size = 64;
    while (table_size < 4 * B) table_size <<= 1;
    const size_t lds = size_t(table_size) * 8 + size_t(B) * 4 + 16;
    hipLaunchKernelGGL(split_sample_gather_kernel, dim3(1), dim3(threads), lds, (hipStream_t)stream, *rb, n_demo,
                       n_online, seed, counter, counter_dev, counter_inc, table_size - 1, out);
    return check_launch();
}

int rrl_creplay_sample_gather(const rrl_replay_t* rb, int32_t n_pos, int32_t n_neg, uint64_t seed,
                              uint64_t counter, uint64_t* counter_dev, uint64_t counter_inc, float* s, float* a,
                              float* r, float* s2, float* m, int64_t* idx_out, float* xu, float* x2u, float* xpu,
                              void* stream) {
    if (!valid_rb(rb) || !rb->pos_cnt || !s || !a || !r || !s2 || !m) return RRL_EINVAL;
    const int B = n_pos + n_neg;
    if (n_pos < 0 || n_neg < 0 || B <= 0 || B > 1024) return RRL_ERANGE;
    if (rb->cap > (int64_t(1) << 21)) return RRL_ERANGE;
    const int n_chunks = int((rb->cap + kChunk - 1) / kChunk);
    const int threads = ((B + 63) / 64) * 64;
    int table_size = 64;
    while (table_size < 4 * B) table_size <<= 1;
    const size_t lds = size_t(table_size) * 8 + size_t((B + 3) & ~3) * 4 + size_t(rrl_replay::count_supers(rb->cap) + 2) * 4 + 16;
    static size_t granted = 64 * 1024;   // gfx950 has 160 KiB of LDS per CU; opt in (once per size) above the default
    if (lds > granted) {
        if (hipFuncSetAttribute((const void*)creplay_sample_gather_kernel,
                                hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)) != hipSuccess) {
            (void)hipGetLastError();
            return RRL_ERANGE;
        }
        granted = lds;
    }
    const BatchOut out{(float2*)s, (float2*)a, r, (float2*)s2, m, idx_out, (float4*)xu, (float4*)x2u, (float4*)xpu};
    hipLaunchKernelGGL(creplay_sample_gather_kernel, dim3(1), dim3(threads), lds,
                       (hipStream_t)stream, *rb, n_pos, n_neg, n_chunks, seed, counter, counter_dev,
                       counter_inc, table_size - 1, out);
    return check_launch();
}

}  // extern "C"
