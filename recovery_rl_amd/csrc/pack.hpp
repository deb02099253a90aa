// pack.hpp -- S independent learners ("seeds": own envs, replay rings, networks, Philox keys) sharing every launch of the
// lock-step iteration.  One launch of the iteration occupies <= 64 of the 256 CUs and mostly waits on memory round trips,
// and the command processor, not the CUs, limits how many such launches per second a device takes (S streams replaying S
// graphs overlap to 1.6x at most, profiles/seed_pack_probe.py), so the reference's unit of parallelism -- the seed loop,
// scripts/navigation1.sh:4-8 -- is packed INSIDE the launches: launch k of the packed iteration is launch k of every seed,
// side by side on disjoint workgroups, each seed running exactly its stand-alone code on its own argument block.
//
// Argument blocks of S seeds exceed the 4 KB kernel-argument limit, so they live in device memory: `upload` keeps a
// content-addressed cache (the blocks of the steady-state iteration never change: after the first iteration every launch is
// a hit and nothing is copied -- in particular nothing inside a captured hipGraph); the kernel gets one pointer plus the
// per-seed block ranges (Idx, by value).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <unordered_map>
#include <vector>

namespace rrl_pack {

constexpr int kMaxSeeds = 16;

struct Idx {
    int first[kMaxSeeds + 1];     // workgroups [first[s], first[s + 1]) belong to seed s
    int S;
};

__device__ __forceinline__ int seed_of(const Idx& ix, int block) {
    int s = 0;
    while (s + 1 < ix.S && block >= ix.first[s + 1]) ++s;
    return s;
}

struct Entry {
    void* dev;
    std::vector<char> host;
};

// device copy of `bytes` bytes at `host` (stream-ordered copy on a miss); nullptr on allocation failure
inline const void* upload(const void* host, size_t bytes, hipStream_t st) {
    static std::unordered_map<uint64_t, std::vector<Entry>> cache;
    static size_t entries = 0;
    uint64_t h = 1469598103934665603ULL;                    // FNV-1a over the bytes
    const unsigned char* p = static_cast<const unsigned char*>(host);
    for (size_t i = 0; i < bytes; ++i) h = (h ^ p[i]) * 1099511628211ULL;
    auto& bucket = cache[h];
    for (const Entry& e : bucket)
        if (e.host.size() == bytes && memcmp(e.host.data(), host, bytes) == 0) return e.dev;
    if (entries >= 4096) return nullptr;                    // argument blocks that change every call: not this mechanism
    Entry e;
    e.host.assign(reinterpret_cast<const char*>(host), reinterpret_cast<const char*>(host) + bytes);
    if (hipMalloc(&e.dev, bytes) != hipSuccess) return nullptr;
    bucket.push_back(std::move(e));
    ++entries;
    const Entry& kept = bucket.back();                      // the staging copy outlives the asynchronous transfer
    if (hipMemcpyAsync(kept.dev, kept.host.data(), bytes, hipMemcpyHostToDevice, st) != hipSuccess) return nullptr;
    return kept.dev;
}

}  // namespace rrl_pack
