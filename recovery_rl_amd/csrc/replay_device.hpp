// replay_device.hpp -- device-side pieces of the replay ring shared by the push kernels and the
// fused env-step + push kernel.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rrl_hip.h"

namespace rrl_replay {

constexpr int kChunk = 64;    // slots per positive-count chunk
constexpr int kSuper = 1024;  // slots per second-level count (16 chunks)

// pos_cnt layout: [n_chunks] per-chunk counts, padded to a multiple of 4, then [n_super] per-1024-slot counts, padded to
// a multiple of 2, then [n_chunks] 64-bit masks (bit b of mask c = slot 64 c + b holds a positive row)
__host__ __device__ __forceinline__ int64_t count_chunks(int64_t cap) { return (cap + kChunk - 1) / kChunk; }
__host__ __device__ __forceinline__ int64_t count_supers(int64_t cap) { return (cap + kSuper - 1) / kSuper; }
__host__ __device__ __forceinline__ int64_t super_base(int64_t cap) { return (count_chunks(cap) + 3) & ~int64_t(3); }
__host__ __device__ __forceinline__ int64_t mask_base(int64_t cap) {
    return (super_base(cap) + count_supers(cap) + 1) & ~int64_t(1);
}
__device__ __forceinline__ unsigned long long* chunk_masks(const rrl_replay_t& rb) {
    return reinterpret_cast<unsigned long long*>(rb.pos_cnt + mask_base(rb.cap));
}

// Both count levels += delta (delta in {-1, 0, +1}) and the chunk masks for the active lanes of a wave: ONE set of atomics per distinct
// chunk for the first two distinct chunks (a wave pushes 64 consecutive slots: at most two chunks), lane by lane for
// anything beyond that (a ring wrap).  4096 envs pushing into one super-chunk would otherwise queue hundreds of atomics
// on one address (+4 us on the 11 us step kernel).  Call from converged code: every active lane of the wave.
// `block_acc` (optional, LDS int[2]): the caller sums the second level per workgroup and pass instead -- the wave's net
// goes to block_acc[super-chunk != s0], s0 = the super-chunk of the workgroup's first slot of the pass (256 consecutive
// slots touch at most two), and the caller adds the two sums to memory after a barrier.
__device__ __forceinline__ void wave_count_add(int32_t* chunk_cnt, int32_t* super_cnt, unsigned long long* masks,
                                               int chunk, int bit, int delta, int* block_acc = nullptr, int s0 = 0) {
    bool pending = delta != 0;
    unsigned long long act = __ballot(pending);
    if (!act) return;
    const int lane = threadIdx.x & 63;
    constexpr int kPer = kSuper / kChunk;
#pragma unroll
    for (int round = 0; round < 2 && act; ++round) {
        const int leader = __ffsll(act) - 1;
        const int c0 = __shfl(chunk, leader, 64), bit0 = __shfl(bit, leader, 64);
        const bool same = pending & (chunk == c0);
        const unsigned long long up = __ballot(same & (delta > 0)), down = __ballot(same & (delta < 0));
        const int net = __popcll(up) - __popcll(down);
        // the chunk's mask: when the group's lanes hold consecutive slots (they do unless the push was masked) the two
        // ballots ARE the bits to set and to clear, shifted from lane numbers to slot numbers
        const bool in_line = __ballot(same & (bit - bit0 != lane - leader)) == 0;
        if (lane == leader) {
            if (net != 0) {
                atomicAdd(&chunk_cnt[c0], net);
                if (block_acc) atomicAdd(&block_acc[(c0 / kPer) != s0], net);
                else atomicAdd(&super_cnt[c0 / kPer], net);
            }
            if (in_line) {
                if (up) atomicOr(&masks[c0], (up >> leader) << bit0);
                if (down) atomicAnd(&masks[c0], ~((down >> leader) << bit0));
            }
        }
        if (!in_line && same) {
            if (delta > 0) atomicOr(&masks[chunk], 1ULL << bit);
            else atomicAnd(&masks[chunk], ~(1ULL << bit));
        }
        pending = pending & !same;
        act &= ~(up | down);
    }
    if (pending) {
        atomicAdd(&chunk_cnt[chunk], delta);
        if (block_acc) atomicAdd(&block_acc[(chunk / kPer) != s0], delta);
        else atomicAdd(&super_cnt[chunk / kPer], delta);
        if (delta > 0) atomicOr(&masks[chunk], 1ULL << bit);
        else atomicAnd(&masks[chunk], ~(1ULL << bit));
    }
}

// 1 when `slot` holds a positive row now (only buffers that keep positive counts look): the load a push has to wait for
// before it can update the counts -- callers request it early
__device__ __forceinline__ int was_positive(const rrl_replay_t& rb, int64_t slot, int64_t size) {
    return (rb.pos_cnt && slot < size) ? int(rb.r[slot] != 0.0f) : 0;
}
// The same in two halves for callers that want the row's request in flight while they do something else: `safe` = any
// readable 4 bytes -- the row is read through a selected address instead of under a branch (a load under a branch is waited
// for where the branch ends) -- and the value is looked at by was_positive_of only.
struct WasRow {
    float r;
    bool look;
};
template <bool BRANCH_FREE = true>
__device__ __forceinline__ WasRow was_positive_request(const rrl_replay_t& rb, int64_t slot, int64_t size, const void* safe) {
    WasRow w;
    w.look = rb.pos_cnt && slot < size;
    if constexpr (BRANCH_FREE) w.r = *(w.look ? rb.r + slot : reinterpret_cast<const float*>(safe));
    else w.r = w.look ? rb.r[slot] : 0.f;
    return w;
}
__device__ __forceinline__ int was_positive_of(const WasRow& w) { return w.look ? int(w.r != 0.0f) : 0; }

// write one row into `slot` (`was` = was_positive(rb, slot, size)); keeps both levels of positive counts exact (pos_idx,
// replay_memory.py:50).  Call from converged code (all active lanes of the wave reach it together).
__device__ __forceinline__ void store_values(const rrl_replay_t& rb, int64_t slot, int was, float2 s,
                                             float2 a, float r, float2 s2, float m, int* block_acc = nullptr,
                                             int s0 = 0) {
    if (rb.pos_cnt) {
        const int delta = int(r != 0.0f) - was;
        wave_count_add(rb.pos_cnt, rb.pos_cnt + super_base(rb.cap), chunk_masks(rb), int(slot / kChunk),
                       int(slot % kChunk), delta, block_acc, s0);
    }
    ((float2*)rb.s)[slot] = s;
    ((float2*)rb.a)[slot] = a;
    rb.r[slot] = r;
    ((float2*)rb.s2)[slot] = s2;
    rb.m[slot] = m;
}

// slot of the i-th row pushed when the ring's cursor is at `pos` (i <= cap - pinned): past the last slot the ring continues
// at slot `pinned` (rows [0, pinned) are never overwritten; pinned = 0: (pos + i) % cap)
__device__ __forceinline__ int64_t ring_slot(const rrl_replay_t& rb, int64_t pos, int64_t i) {
    const int64_t s = pos + i;
    return s < rb.cap ? s : s - rb.cap + rb.pinned;
}

// {position, size} after `pushed` more rows (called by the one thread that won the launch's ticket)
__device__ __forceinline__ void set_ring(const rrl_replay_t& rb, int64_t pos, int64_t size, int64_t pushed) {
    rb.state[0] = ring_slot(rb, pos, pushed);
    const int64_t ns = size + pushed;
    rb.state[1] = ns > rb.cap ? rb.cap : ns;
}

// Last workgroup to finish advances {position, size}; every workgroup has read them before it takes
// its ticket, so no workgroup can observe the new values.  Call with all threads of the block.
__device__ __forceinline__ void advance_ring(const rrl_replay_t& rb, int64_t pos, int64_t size,
                                             int64_t pushed) {
    __syncthreads();
    if (threadIdx.x == 0) {
        // no fence: the ticket only orders this workgroup's READS of {position, size} (already consumed)
        // before the last arriver's write; the write itself is published by the kernel boundary
        const unsigned long long ticket = atomicAdd((unsigned long long*)&rb.state[2], 1ULL);
        if (ticket == gridDim.x - 1) {
            rb.state[0] = ring_slot(rb, pos, pushed);
            const int64_t ns = size + pushed;
            rb.state[1] = ns > rb.cap ? rb.cap : ns;
            rb.state[2] = 0;
        }
    }
}

}  // namespace rrl_replay
