// mlp_common.hpp -- pieces shared by the MLP kernel files (mlp_kernels.hip: GEMM tiles and the stack backward;
// mlp_fwd_kernels.hip: the fused stack forward).
#pragma once

#include <hip/hip_runtime.h>

#include <vector>

#include "rrl_device.hpp"
#include "pack.hpp"
#include "rrl_host.hpp"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kMaxGroup = 4;     // members of a grouped launch

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

}  // namespace

// ---- solo group launches: the member comes out of the grid -------------------------------------------------------------
// A solo group kernel is launched on grid (workgroups of the largest member, members[, ...]): blockIdx.y IS the member, so
// its argument block sits at a kernel-argument address known at wave start and ONE batch of scalar loads fetches it.  A flat
// grid (the packed form keeps it: its blocks live in device memory anyway) has to walk first[] to find the member before it
// can ask for the member's block: one more dependent round trip to memory, ~0.5 us of a ~7 us kernel (profiles/
// round4_levels.txt).  Members with fewer workgroups than the largest leave the surplus empty (exit after that batch).
// arrive_together: every listed scalar is in a register at this point, i.e. the compiler requests all of them up front and
// waits once, instead of one s_waitcnt per first use with further scalar loads issued behind it.
template <class T>
__device__ __forceinline__ void in_sgpr(const T& v) {
    asm volatile("" ::"s"(v));
}
template <class... T>
__device__ __forceinline__ void arrive_together(const T&... v) {
    (in_sgpr(v), ...);
}
template <class Group>
static int largest_member(const Group& g, int n) {
    int most = 1;
    for (int k = 0; k < n; ++k) most = g.first[k + 1] - g.first[k] > most ? g.first[k + 1] - g.first[k] : most;
    return most;
}

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

// 2-D packed launch (rrl_pack::locate_grid): seed s owns `most[s]` workgroups per member row; returns grid.x
static int finish_members(rrl_pack::Idx& ix, int S, const int* most) {
    ix.S = S;
    ix.first[0] = 0;
    for (int s = 0; s < S; ++s) ix.first[s + 1] = ix.first[s] + (most[s] > 0 ? most[s] : 1);
    for (int s = S; s < rrl_pack::kMaxSeeds; ++s) ix.first[s + 1] = ix.first[S];
    return rrl_pack::finish(ix);
}
// the placement's four scalars in one batch of kernel-argument loads, then (seed, index inside the seed's member row).
// (`plan` is named for the reader; pinning the pointer as well -- "s"(address) -- makes the compiler copy it out of a vector
// register in some kernels and fails: "illegal VGPR to SGPR copy")
#define RRL_PACK_LOCATE(ix, plan, s, local)                                                   \
    int s, local;                                                                             \
    arrive_together((ix).sp, (ix).p, (ix).r, (ix).S);                                         \
    if (!rrl_pack::locate_grid((ix), blockIdx.x, s, local)) return

// tiles of R >= 4 need more than the default 64 KB of LDS per workgroup (gfx950 has 160 KB per CU): opt in once
static bool grant_lds(const void* kernel, size_t bytes) {
    if (bytes <= 64 * 1024) return true;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, int(bytes)) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return true;
}
