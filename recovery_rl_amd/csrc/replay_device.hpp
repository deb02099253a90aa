// replay_device.hpp -- device-side pieces of the replay ring shared by the push kernels and the
// fused env-step + push kernel.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rrl_hip.h"

namespace rrl_replay {

constexpr int kChunk = 64;    // slots per positive-count chunk
constexpr int kSuper = 4096;  // slots per second-level count (64 chunks)

// pos_cnt layout: [n_chunks] per-chunk counts, padded to a multiple of 4, then [n_super] per-4096-slot counts
__host__ __device__ __forceinline__ int64_t count_chunks(int64_t cap) { return (cap + kChunk - 1) / kChunk; }
__host__ __device__ __forceinline__ int64_t count_supers(int64_t cap) { return (cap + kSuper - 1) / kSuper; }
__host__ __device__ __forceinline__ int64_t super_base(int64_t cap) { return (count_chunks(cap) + 3) & ~int64_t(3); }

// write one row into `slot`; keeps the per-chunk positive counts exact (pos_idx, replay_memory.py:50)
__device__ __forceinline__ void store_values(const rrl_replay_t& rb, int64_t slot, int64_t size, float2 s,
                                             float2 a, float r, float2 s2, float m) {
    if (rb.pos_cnt) {
        const int was = (slot < size) ? int(rb.r[slot] != 0.0f) : 0;
        const int delta = int(r != 0.0f) - was;
        if (delta) {
            atomicAdd(&rb.pos_cnt[slot / kChunk], delta);
            atomicAdd(&rb.pos_cnt[super_base(rb.cap) + slot / kSuper], delta);
        }
    }
    ((float2*)rb.s)[slot] = s;
    ((float2*)rb.a)[slot] = a;
    rb.r[slot] = r;
    ((float2*)rb.s2)[slot] = s2;
    rb.m[slot] = m;
}

// {position, size} after `pushed` more rows (called by the one thread that won the launch's ticket)
__device__ __forceinline__ void set_ring(const rrl_replay_t& rb, int64_t pos, int64_t size, int64_t pushed) {
    rb.state[0] = (pos + pushed) % rb.cap;
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
            rb.state[0] = (pos + pushed) % rb.cap;
            const int64_t ns = size + pushed;
            rb.state[1] = ns > rb.cap ? rb.cap : ns;
            rb.state[2] = 0;
        }
    }
}

}  // namespace rrl_replay
