// pack.hpp -- S independent learners ("seeds": own envs, replay rings, networks, Philox keys) sharing every launch of the
// lock-step iteration.  One launch of the iteration occupies <= 64 of the 256 CUs and mostly waits on memory round trips,
// and the command processor, not the CUs, limits how many such launches per second a device takes (S streams replaying S
// graphs overlap to 1.6x at most, profiles/seed_pack_probe.py), so the reference's unit of parallelism -- the seed loop,
// scripts/navigation1.sh:4-8 -- is packed INSIDE the launches: launch k of the packed iteration is launch k of every seed,
// side by side on disjoint workgroups, each seed running exactly its stand-alone code on its own argument block.
//
// Argument blocks of S seeds exceed the 4 KB kernel-argument limit, so they live in device memory, one `Plan` per distinct
// input (the inputs of the steady-state iteration never change: after the first iteration every launch is a hit and nothing
// is built, allocated or copied -- in particular nothing inside a captured hipGraph); the kernel gets one pointer plus the
// per-seed block ranges (Idx, by value).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <cstdlib>
#include <unordered_map>
#include <vector>

#include "rrl_hip.h"

namespace rrl_pack {

constexpr int kMaxSeeds = 16;

struct Idx {
    int first[kMaxSeeds + 1];     // seed s owns first[s + 1] - first[s] workgroups (linear mapping: [first[s], first[s + 1]))
    int S;
    // XCD-aware mapping (sp > 0): the dispatcher hands workgroup b to XCD b % 8, and every XCD has its own L2.  Seed s is
    // pinned to the XCDs {s, s + sp, s + 2 sp, ...} (sp = S rounded up to a power of two, p = 8 / sp of them): its weights,
    // activations and replay rows then live in p L2s instead of all eight, and no L2 holds more than one seed's working
    // set (linear mapping: every L2 caches every seed's weights -- 4 seeds x ~1.6 MB against 4 MB per L2).
    // S > 8: r = ceil(S / 8) seeds share an XCD (sp = 8, p = 1): the workgroups XCD x receives serve its seeds x, x + 8, ...
    // in turn
    int sp, p, r;
};

__device__ __forceinline__ int seed_of(const Idx& ix, int block) {
    int s = 0;
    while (s + 1 < ix.S && block >= ix.first[s + 1]) ++s;
    return s;
}

// (seed, workgroup index inside the seed's own grid) of physical workgroup b; false: b serves nobody (XCD-aware grids are
// padded to whole rounds of eight)
__device__ __forceinline__ bool locate(const Idx& ix, int b, int& s, int& local) {
    if (ix.sp == 0) {
        s = seed_of(ix, b);
        local = b - ix.first[s];
        return true;
    }
    const int x = b & 7;
    if (ix.r > 1) {
        const int j = b >> 3;
        s = x + 8 * (j % ix.r);
        local = j / ix.r;
    } else {
        s = x % ix.sp;
        local = (b >> 3) * ix.p + x / ix.sp;
    }
    return s < ix.S && local < ix.first[s + 1] - ix.first[s];
}

// The same for a launch whose MEMBER (stack / backward problem / Adam segment of the seed's group) is blockIdx.y -- the
// round-4 structure of the solo group launches, brought to the packed ones in round 5: first[] then holds the workgroups of
// every seed's LARGEST member, and under the XCD-aware placement the seed and the index inside its grid follow from b by
// arithmetic alone.  Nothing is read before the member's argument block -- whose address is then known at wave start, so
// that ONE batch of scalar loads fetches it from the plan's device copy -- and the block's own workgroup count bounds `local`
// (a surplus workgroup exits behind that batch).  The linear placement (S > 8 with an uneven last round) keeps the walk.
__device__ __forceinline__ bool locate_grid(const Idx& ix, int b, int& s, int& local) {
    if (ix.sp == 0) return locate(ix, b, s, local);
    const int x = b & 7;
    if (ix.r > 1) {
        const int j = b >> 3;
        s = x + 8 * (j % ix.r);
        local = j / ix.r;
    } else {
        s = x % ix.sp;
        local = (b >> 3) * ix.p + x / ix.sp;
    }
    return s < ix.S;
}

// A packed kernel copies its argument block out of DEVICE memory, and a pointer loaded from memory is a generic ("flat")
// pointer to the compiler: every access through it becomes a flat_load / flat_store, which counts against BOTH the vector
// memory and the LDS counters and may complete out of order -- each LDS wait then drains every global load in flight
// (s_waitcnt vmcnt(0) lgkmcnt(0)), i.e. the overlap of the next panel's loads with LDS-fed MFMAs that the solo kernels
// (pointers in kernel arguments = known global) rely on is gone.  Passing the copied pointers through the global address
// space tells the compiler what they are; the bits of the pointer (and of every result) do not change.  (The empty asm keeps
// the optimiser from folding the cast pair away before address spaces are inferred; "s": an argument block is per seed,
// i.e. wave-uniform, and stays in scalar registers as the base of global_load v, v_offset, s[base].)
template <class T>
__device__ __forceinline__ void to_global(T*& p) {
    auto g = (__attribute__((address_space(1))) T*)p;
    asm volatile("" : "+s"(g));
    p = (T*)g;
}
template <class... P>
__device__ __forceinline__ void to_global_all(P&... p) {
    (to_global(p), ...);
}

__device__ __forceinline__ void globalize(rrl_replay_t& rb) {
    to_global_all(rb.s, rb.a, rb.r, rb.s2, rb.m, rb.state, rb.pos_cnt);
}
__device__ __forceinline__ void globalize(rrl_policy_head_t& h) {
    to_global_all(h.head, h.eps, h.scale, h.bias, h.action, h.logp, h.mean_out, h.obs_in, h.obs_out, h.log_std);
}

// host: the mapping S seeds get -- false: linear (the seeds share all eight XCDs); true: pinned, seed s on p = 8 / sp XCDs of
// its own (S <= 8) or r seeds taking turns on every XCD (S > 8, sp = 8, p = 1)
inline bool pinned_mapping(int S, int& sp, int& p, int& r) {
    sp = p = 0;
    r = 1;
    if (S > 8) {
        // r seeds per XCD; XCDs with fewer seeds idle while the others finish, so only when (almost) every XCD has r:
        // measured at 16 updates per step, S = 16: 7.94 ms against 9.10 linear; S = 12: 7.82 against 7.39
        if (8 * ((S + 7) / 8) - S > 2) return false;
        sp = 8;
        p = 1;
        r = (S + 7) / 8;
        return true;
    }
    sp = 1;
    while (sp < S) sp <<= 1;
    // S = 5, 6: one XCD per seed would leave three / two XCDs without work -- the linear mapping keeps all eight busy and loses
    // only the L2 locality (0.366 / 0.400 ms per packed iteration against 0.493 / 0.469 pinned; S = 3 on two XCDs each: 0.317
    // pinned against 0.344 linear; S = 7: equal).  RRL_PACK_PINNED=1: always pinned (A/B switch of profiles/).
    static const bool always = [] {
        const char* v = getenv("RRL_PACK_PINNED");
        return v && *v && atoi(v) != 0;
    }();
    if (!always && sp == 8 && S <= 6) {
        sp = 0;
        return false;
    }
    p = 8 / sp;
    return true;
}
// ... and the share of the chip one seed's workgroups can occupy at a time under it, as a divisor (1 = the seeds share the chip)
inline int seed_share(int S) {
    int sp, p, r;
    return pinned_mapping(S, sp, p, r) ? sp * r : 1;
}

// host: choose the mapping for `ix` (first[] and S filled in) and return the grid size
inline int finish(Idx& ix) {
    int most = 0;
    for (int s = 0; s < ix.S; ++s) most = ix.first[s + 1] - ix.first[s] > most ? ix.first[s + 1] - ix.first[s] : most;
    if (!pinned_mapping(ix.S, ix.sp, ix.p, ix.r)) return ix.first[ix.S];
    if (ix.S > 8) return 8 * ix.r * most;
    return 8 * ((most + ix.p - 1) / ix.p);
}

// A packed launch = (device copy of the S argument blocks, block ranges, launch parameters), built once per distinct
// INPUT and looked up by the bytes of that input (the caller's descriptor arrays, which the Python side builds in zeroed
// ctypes memory: deterministic, unlike the padding bytes of structs assembled here).  A hit costs one hash of a few KB.
struct Key {
    std::vector<char> bytes;
    void add(const void* p, size_t n) {
        const char* c = static_cast<const char*>(p);
        bytes.insert(bytes.end(), c, c + n);
    }
    template <class T>
    void pod(const T& v) { add(&v, sizeof v); }
    uint64_t hash() const {
        uint64_t h = 1469598103934665603ULL;                    // FNV-1a
        for (char c : bytes) h = (h ^ (unsigned char)c) * 1099511628211ULL;
        return h;
    }
};

struct Plan {
    void* dev = nullptr;          // S argument blocks in device memory
    Idx ix{};
    int grid = 0;                 // workgroups of the launch (finish(ix))
    int i0 = 0, i1 = 0;           // kernel-specific launch parameters (path, threads, ...)
    size_t z0 = 0;                // ... dynamic LDS bytes
    std::vector<char> key, host;  // the input it was built from; staging copy of the blocks (outlives the async transfer)
};

inline std::unordered_map<uint64_t, std::vector<Plan*>>& plans() {
    static std::unordered_map<uint64_t, std::vector<Plan*>> m;
    return m;
}

inline std::mutex& plan_mutex() {
    static std::mutex m;
    return m;
}
// why the last store() of this thread returned nullptr (RRL_ELAUNCH: allocation / copy failed)
inline int& store_error() {
    static thread_local int e = RRL_ELAUNCH;
    return e;
}

inline Plan* lookup(const Key& k) {
    std::lock_guard<std::mutex> lock(plan_mutex());
    auto it = plans().find(k.hash());
    if (it == plans().end()) return nullptr;
    for (Plan* p : it->second)
        if (p->key.size() == k.bytes.size() && memcmp(p->key.data(), k.bytes.data(), k.bytes.size()) == 0) return p;
    return nullptr;
}

// new plan for `k`: `blocks` (bytes) copied to device memory, stream-ordered.  nullptr: allocation failed, or more distinct
// inputs than this mechanism is meant for (argument blocks that change on every call)
inline size_t& plan_count() {
    static size_t n = 0;
    return n;
}

inline Plan* store(const Key& k, const void* blocks, size_t bytes, hipStream_t st) {
    std::lock_guard<std::mutex> lock(plan_mutex());
    store_error() = RRL_ELAUNCH;
    // a miss while the stream is CAPTURING would put a hipMalloc + a pageable copy into the graph and invalidate it with an
    // opaque error: say what happened instead (the caller's warm-up iterations did not launch this argument block)
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
        store_error() = RRL_ECAPTURE;
        return nullptr;
    }
    if (plan_count() >= 8192) {          // argument blocks that change on every call are not what this cache is for
        store_error() = RRL_EPLANS;
        return nullptr;
    }
    Plan* p = new Plan();
    p->key = k.bytes;
    p->host.assign(static_cast<const char*>(blocks), static_cast<const char*>(blocks) + bytes);
    if (hipMalloc(&p->dev, bytes) != hipSuccess ||
        hipMemcpyAsync(p->dev, p->host.data(), bytes, hipMemcpyHostToDevice, st) != hipSuccess) {
        (void)hipGetLastError();
        delete p;
        return nullptr;
    }
    plans()[k.hash()].push_back(p);
    ++plan_count();
    return p;
}

// free every plan (device copies included).  Only when no captured graph that launches a packed kernel is alive: the graphs
// hold the plans' device pointers as kernel arguments.
inline int clear() {
    std::lock_guard<std::mutex> lock(plan_mutex());
    int n = 0;
    for (auto& kv : plans())
        for (Plan* p : kv.second) {
            (void)hipFree(p->dev);
            delete p;
            ++n;
        }
    plans().clear();
    plan_count() = 0;
    return n;
}

}  // namespace rrl_pack
