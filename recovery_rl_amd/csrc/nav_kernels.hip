// nav_kernels.hip -- batched Navigation1 / Navigation2 env kernels for gfx950 (MI355X).
//
// Structure-of-arrays state in HBM, every access unit-stride across the 64-lane wavefront; one, two or four
// consecutive envs per lane depending on the launch size (latency regime / one resident round / streaming).
// General layout (rrl_nav_step: the reference's arrays): 28 B read + 44 B written per env-step; compact layout
// (rrl_nav_step_compact: u16 status words, one observation array): 26 B + 30 B.  All arithmetic in registers
// (Philox + Box-Muller in f64); LDS only for the per-wave list of finished rows whose start states are drawn
// once per wave and pass.  Grid: <= 2048 workgroups of 256 threads, grid-stride.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "replay_device.hpp"
#include "rrl_device.hpp"
#include "rrl_host.hpp"
#include "step_push.hpp"

#pragma clang fp contract(off)

namespace {

using rrl_host::check_launch;
using rrl_host::grid_for;
using rrl_host::kBlock;

template <int KIND, bool EXT_NOISE>
__global__ __launch_bounds__(kBlock) void nav_step_kernel(StepArgs a) {
    const uint64_t ctr = rrl::effective_counter(a.counter, a.counter_dev);
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < a.n; i += stride) {
        const double2 p = a.pos[i];
        const float2 act = a.action[i];
        int32_t ti = a.t[i];
        double ex, ey;
        if constexpr (EXT_NOISE) {
            const double2 e = a.noise[i];
            ex = e.x;
            ey = e.y;
        } else {
            rrl::normal_at(a.seed, uint32_t(i), rrl::kStreamStep, ctr, ex, ey);
        }
        double nx, ny, cost;
        rrl::nav_transition<KIND>(p.x, p.y, double(act.x), double(act.y), ex, ey, nx, ny, cost);
        const bool cons = rrl::in_obstacle<KIND>(nx, ny);
        const bool succ = cost > -4.0;
        const bool dn = succ | cons;
        ti += 1;
        const bool epd = dn | (ti == a.horizon);
        a.next_obs[i] = make_float2(float(nx), float(ny));
        a.reward[i] = float(cost);
        a.done[i] = uint8_t(dn);
        a.constraint[i] = uint8_t(cons);
        a.success[i] = uint8_t(succ);
        if (a.ep_done) a.ep_done[i] = uint8_t(epd);
        if (a.auto_reset && epd) {
            double z0, z1;
            rrl::normal_at(a.seed, uint32_t(i), rrl::kStreamReset, ctr, z0, z1);
            nx = -50.0 + z0;
            ny = 0.0 + z1;
            ti = 0;
        }
        a.pos[i] = make_double2(nx, ny);
        a.t[i] = ti;
        if (a.obs) a.obs[i] = make_float2(float(nx), float(ny));
    }
    rrl::advance_counter(a.counter_dev, a.counter_inc);
}

// Resets of the bandwidth-regime kernels.  With a ~1.2 % termination rate per step, one to three of the 256 envs a wave
// steps per pass finish, and an inline reset sends the whole wave through the Philox + Box-Muller chain again for each k
// that has a finished lane (2.2 extra trips per pass).  Measured at 2^24 envs, all variants interleaved in one process
// (profiles/nav_step_probe.py; compact layout without reset_obs / general layout; no resets at all: 233 / 261 us):
//   inline per lane                                            268 / 276 us
//   once per wave and pass, handed back through LDS (below)     243 / 262 us   <- kept
//   deferred to the end of the wave, pos patched by scattered stores       267 / 353 us
//   once per workgroup and pass (two barriers)                  ~340 / 357 us
//   once per wave and TWO passes, the first parked in LDS (nav_step_compact_hold_kernel, two envs per thread)   222 / 249 us
// Scattered small stores are what the deferred variant pays for: ~200 k of them per launch cost 35-50 us next to the
// streaming traffic (profiles/sparse_write_probe.hip: anything below a whole 64-byte granule is a read-modify-write).
// The reset draws of the rows a wave finished in this pass, evaluated once per wave and pass.  The finished
// rows of the wave's V x 64 envs are listed in LDS (slots from ballots), the first `total` lanes evaluate normal_at() for
// them in ONE trip (instead of one trip per k with any finished lane: 2.2 trips per pass at a 1.2 % termination rate), and
// the owners read the pair back and store it with their dense stores -- no scattered store anywhere.
template <int V>
__device__ __forceinline__ void wave_reset_draws(uint64_t seed, uint64_t ctr, const uint32_t (&row)[V],
                                                 const bool (&fin)[V], double (&z0)[V], double (&z1)[V], uint32_t* rows,
                                                 double2* draws) {
    const int lane = threadIdx.x & 63;
    const uint64_t below = (1ULL << lane) - 1ULL;
    int slot[V], total = 0;
#pragma unroll
    for (int k = 0; k < V; ++k) {
        const uint64_t bal = __ballot(fin[k]);
        slot[k] = total + __popcll(bal & below);
        total += __popcll(bal);
    }
    if (total == 0) return;
    if (total > 64) {
#pragma unroll
        for (int k = 0; k < V; ++k)
            if (fin[k]) rrl::normal_at(seed, row[k], rrl::kStreamReset, ctr, z0[k], z1[k]);
        return;
    }
#pragma unroll
    for (int k = 0; k < V; ++k)
        if (fin[k]) rows[slot[k]] = row[k];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane < total) {
        double x, y;
        rrl::normal_at(seed, rows[lane], rrl::kStreamReset, ctr, x, y);
        draws[lane] = make_double2(x, y);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < V; ++k)
        if (fin[k]) {
            const double2 d = draws[slot[k]];
            z0[k] = d.x;
            z1[k] = d.y;
        }
    __builtin_amdgcn_wave_barrier();   // the lists are reused by the wave's next pass
}

// the same for V consecutive rows from i0
template <int V>
__device__ __forceinline__ void wave_reset_draws(uint64_t seed, uint64_t ctr, int64_t i0, const bool (&fin)[V],
                                                 double (&z0)[V], double (&z1)[V], uint32_t* rows, double2* draws) {
    uint32_t row[V];
#pragma unroll
    for (int k = 0; k < V; ++k) row[k] = uint32_t(i0 + k);
    wave_reset_draws<V>(seed, ctr, row, fin, z0, z1, rows, draws);
}

// Mid-range variant (2^19 <= n < 2^22, n % 4 == 0; measured: 34.9 -> 27.1 us at 2^20, slower below 2^18): one thread steps FOUR consecutive envs
// (a single round of 4 waves per SIMD covers 2^20 envs; beyond that the two-env kernel below wins).  All loads of the
// four envs are issued before the first dependent f64 operation (4x the bytes in flight per thread), every
// access is a 16-byte vector (the four u8 masks of the four envs become one 32-bit store per array), and a wave
// touches 4 KB of contiguous positions.  The per-env arithmetic is the scalar kernel's, call for call, so the
// results are bit-identical.
template <int KIND, bool EXT_NOISE>
__global__ __launch_bounds__(kBlock) void nav_step4_kernel(StepArgs a) {
    __shared__ uint32_t wave_rows[kBlock / 64][64];
    __shared__ double2 wave_draws[kBlock / 64][64];
    const uint64_t ctr = rrl::effective_counter(a.counter, a.counter_dev);
    const int64_t n4 = a.n >> 2, stride = int64_t(gridDim.x) * kBlock;
    const int64_t n_pass = (n4 + stride - 1) / stride;     // uniform trip count: the reset lists need whole waves
    for (int64_t pass = 0; pass < n_pass; ++pass) {
        int64_t q = pass * stride + int64_t(blockIdx.x) * kBlock + threadIdx.x;
        const bool live = q < n4;
        if (!live) q = n4 - 1;                             // idle lanes shadow the last quad and store nothing
        const int64_t i0 = q << 2;
        double2 p[4], e[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) p[k] = a.pos[i0 + k];
        const float4 a01 = reinterpret_cast<const float4*>(a.action)[2 * q];
        const float4 a23 = reinterpret_cast<const float4*>(a.action)[2 * q + 1];
        const int4 tv = reinterpret_cast<const int4*>(a.t)[q];
        if constexpr (EXT_NOISE) {
#pragma unroll
            for (int k = 0; k < 4; ++k) e[k] = a.noise[i0 + k];
        }
        const float ax[4] = {a01.x, a01.z, a23.x, a23.z}, ay[4] = {a01.y, a01.w, a23.y, a23.w};
        int32_t ti[4] = {tv.x, tv.y, tv.z, tv.w};
        float2 nobs[4], obs[4];
        float rew[4];
        uint32_t dn4 = 0, cons4 = 0, succ4 = 0, epd4 = 0;
        bool fin[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            double ex, ey;
            if constexpr (EXT_NOISE) {
                ex = e[k].x;
                ey = e[k].y;
            } else {
                rrl::normal_at(a.seed, uint32_t(i0 + k), rrl::kStreamStep, ctr, ex, ey);
            }
            double nx, ny, cost;
            rrl::nav_transition<KIND>(p[k].x, p[k].y, double(ax[k]), double(ay[k]), ex, ey, nx, ny, cost);
            const bool cons = rrl::in_obstacle<KIND>(nx, ny);
            const bool succ = cost > -4.0;
            const bool dn = succ | cons;
            ti[k] += 1;
            const bool epd = dn | (ti[k] == a.horizon);
            nobs[k] = make_float2(float(nx), float(ny));
            rew[k] = float(cost);
            dn4 |= uint32_t(dn) << (8 * k);
            cons4 |= uint32_t(cons) << (8 * k);
            succ4 |= uint32_t(succ) << (8 * k);
            epd4 |= uint32_t(epd) << (8 * k);
            fin[k] = live & epd & (a.auto_reset != 0);
            p[k] = make_double2(nx, ny);
        }
        double w0[4], w1[4];
        wave_reset_draws<4>(a.seed, ctr, i0, fin, w0, w1, wave_rows[threadIdx.x >> 6], wave_draws[threadIdx.x >> 6]);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (fin[k]) {
                ti[k] = 0;
                p[k] = make_double2(-50.0 + w0[k], 0.0 + w1[k]);     // START_STATE + randn(2), navigation1.py:92
            }
            obs[k] = make_float2(float(p[k].x), float(p[k].y));
        }
        if (!live) continue;
#pragma unroll
        for (int k = 0; k < 4; ++k) a.pos[i0 + k] = p[k];
        reinterpret_cast<float4*>(a.next_obs)[2 * q] = make_float4(nobs[0].x, nobs[0].y, nobs[1].x, nobs[1].y);
        reinterpret_cast<float4*>(a.next_obs)[2 * q + 1] = make_float4(nobs[2].x, nobs[2].y, nobs[3].x, nobs[3].y);
        if (a.obs) {
            reinterpret_cast<float4*>(a.obs)[2 * q] = make_float4(obs[0].x, obs[0].y, obs[1].x, obs[1].y);
            reinterpret_cast<float4*>(a.obs)[2 * q + 1] = make_float4(obs[2].x, obs[2].y, obs[3].x, obs[3].y);
        }
        reinterpret_cast<float4*>(a.reward)[q] = make_float4(rew[0], rew[1], rew[2], rew[3]);
        reinterpret_cast<uint32_t*>(a.done)[q] = dn4;
        reinterpret_cast<uint32_t*>(a.constraint)[q] = cons4;
        reinterpret_cast<uint32_t*>(a.success)[q] = succ4;
        if (a.ep_done) reinterpret_cast<uint32_t*>(a.ep_done)[q] = epd4;
        reinterpret_cast<int4*>(a.t)[q] = make_int4(ti[0], ti[1], ti[2], ti[3]);
    }
    rrl::advance_counter(a.counter_dev, a.counter_inc);
}

// Bandwidth-regime variant (n >= 2^22, n even): one thread steps TWO consecutive envs.  All loads of both envs are issued before
// the first dependent f64 operation, the accesses are 16-byte vectors where the layout allows (action, both observation arrays, the
// two positions; 8 bytes for reward and t, 2 for each mask array).  Two rather than four envs per thread: 81 VGPRs instead of 115
// (six waves per SIMD instead of four) and no SGPR-spill traffic.  The per-env arithmetic is the scalar kernel's, call for call, so
// the results are bit-identical.
// The two-env kernel with HOLD passes sharing one reset draw (see nav_step_compact_hold_kernel below for the scheme): a
// pass stores next_obs, reward and the four flags at once and parks positions, step counts and finished flags in LDS;
// after HOLD passes one trip through normal_at() serves every row the wave finished in them, and pos / t / obs go out.
struct HeldStep {
    double2 p[2][kBlock];
    int32_t t[2][kBlock];
    uint32_t fin[kBlock];
};

template <int KIND, bool EXT_NOISE, int HOLD>
__global__ __launch_bounds__(kBlock) void nav_step2_hold_kernel(StepArgs a) {
    __shared__ HeldStep held[HOLD];
    __shared__ uint32_t wave_rows[kBlock / 64][64];
    __shared__ double2 wave_draws[kBlock / 64][64];
    const int tid = threadIdx.x;
    const uint64_t ctr = rrl::effective_counter(a.counter, a.counter_dev);
    const int64_t n2 = a.n >> 1, stride = int64_t(gridDim.x) * kBlock;
    const int64_t n_pass = (n2 + stride - 1) / stride;     // uniform trip count: the reset lists need whole waves
    const int64_t q_first = int64_t(blockIdx.x) * kBlock + tid;
    for (int64_t pass0 = 0; pass0 < n_pass; pass0 += HOLD) {
        const int n_here = int(n_pass - pass0 < HOLD ? n_pass - pass0 : HOLD);      // wave-uniform
#pragma unroll 1
        for (int h = 0; h < n_here; ++h) {
            int64_t q = (pass0 + h) * stride + q_first;
            const bool live = q < n2;
            if (!live) q = n2 - 1;                         // idle lanes shadow the last pair and store nothing
            const int64_t i0 = q << 1;
            double2 p[2], e[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) p[k] = a.pos[i0 + k];
            const float4 a01 = reinterpret_cast<const float4*>(a.action)[q];
            const int2 tv = reinterpret_cast<const int2*>(a.t)[q];
            if constexpr (EXT_NOISE) {
#pragma unroll
                for (int k = 0; k < 2; ++k) e[k] = a.noise[i0 + k];
            }
            const float ax[2] = {a01.x, a01.z}, ay[2] = {a01.y, a01.w};
            int32_t ti[2] = {tv.x, tv.y};
            float2 nobs[2];
            float rew[2];
            uint32_t dn2 = 0, cons2 = 0, succ2 = 0, epd2 = 0, finbits = 0;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                double ex, ey;
                if constexpr (EXT_NOISE) {
                    ex = e[k].x;
                    ey = e[k].y;
                } else {
                    rrl::normal_at(a.seed, uint32_t(i0 + k), rrl::kStreamStep, ctr, ex, ey);
                }
                double nx, ny, cost;
                rrl::nav_transition<KIND>(p[k].x, p[k].y, double(ax[k]), double(ay[k]), ex, ey, nx, ny, cost);
                const bool cons = rrl::in_obstacle<KIND>(nx, ny);
                const bool succ = cost > -4.0;
                const bool dn = succ | cons;
                ti[k] += 1;
                const bool epd = dn | (ti[k] == a.horizon);
                nobs[k] = make_float2(float(nx), float(ny));
                rew[k] = float(cost);
                dn2 |= uint32_t(dn) << (8 * k);
                cons2 |= uint32_t(cons) << (8 * k);
                succ2 |= uint32_t(succ) << (8 * k);
                epd2 |= uint32_t(epd) << (8 * k);
                const bool f = live & epd & (a.auto_reset != 0);
                finbits |= uint32_t(f) << k;
                held[h].p[k][tid] = make_double2(nx, ny);
                held[h].t[k][tid] = f ? 0 : ti[k];
            }
            held[h].fin[tid] = finbits;
            if (live) {
                reinterpret_cast<float4*>(a.next_obs)[q] = make_float4(nobs[0].x, nobs[0].y, nobs[1].x, nobs[1].y);
                reinterpret_cast<float2*>(a.reward)[q] = make_float2(rew[0], rew[1]);
                reinterpret_cast<uint16_t*>(a.done)[q] = uint16_t(dn2);
                reinterpret_cast<uint16_t*>(a.constraint)[q] = uint16_t(cons2);
                reinterpret_cast<uint16_t*>(a.success)[q] = uint16_t(succ2);
                if (a.ep_done) reinterpret_cast<uint16_t*>(a.ep_done)[q] = uint16_t(epd2);
            }
        }
        bool fin[HOLD * 2];
        uint32_t row[HOLD * 2];
#pragma unroll
        for (int h = 0; h < HOLD; ++h) {
            const uint32_t fb = h < n_here ? held[h].fin[tid] : 0u;
            int64_t q = (pass0 + (h < n_here ? h : 0)) * stride + q_first;
            if (q >= n2) q = n2 - 1;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                fin[h * 2 + k] = (fb >> k) & 1u;
                row[h * 2 + k] = uint32_t(2 * q + k);
            }
        }
        double w0[HOLD * 2], w1[HOLD * 2];
        wave_reset_draws<HOLD * 2>(a.seed, ctr, row, fin, w0, w1, wave_rows[tid >> 6], wave_draws[tid >> 6]);
#pragma unroll
        for (int h = 0; h < HOLD; ++h) {
            if (h >= n_here) continue;
            const int64_t q = (pass0 + h) * stride + q_first;
            if (q >= n2) continue;
            const int64_t i0 = q << 1;
            double2 p[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                p[k] = held[h].p[k][tid];
                if (fin[h * 2 + k]) p[k] = make_double2(-50.0 + w0[h * 2 + k], 0.0 + w1[h * 2 + k]);   // navigation1.py:92
                a.pos[i0 + k] = p[k];
            }
            if (a.obs)
                reinterpret_cast<float4*>(a.obs)[q] = make_float4(float(p[0].x), float(p[0].y), float(p[1].x), float(p[1].y));
            reinterpret_cast<int2*>(a.t)[q] = make_int2(held[h].t[0][tid], held[h].t[1][tid]);
        }
    }
    rrl::advance_counter(a.counter_dev, a.counter_inc);
}

// ---- compact form (rrl_nav_step_compact): 56 B moved per env-step instead of 72 ----
// The general entry keeps the reference's separate arrays (four u8 masks, an i32 step count, two f32 observations).
// Here the step count and the four flags share one u16 status word per env (read for the count, rewritten), and the
// post-reset observation -- equal to next_obs wherever the episode goes on -- is written only for the rows whose
// episode ended.  Four envs per thread, every access a 16-byte vector (8 bytes for the status words); a last partial
// quad goes through scalar accesses.  Per-env arithmetic = the general kernel's, call for call.
struct CompactArgs {
    int64_t n;
    double2* pos;
    const float2* action;
    const double2* noise;
    uint64_t seed, counter;
    uint64_t* counter_dev;
    uint64_t counter_inc;
    float2* next_obs;
    float2* reset_obs;
    float* reward;
    uint16_t* status;
    int32_t horizon, auto_reset;
};

template <int V, bool EXT_NOISE>
struct CompactIn {
    double2 p[V], e[V];
    float ax[V], ay[V];
    uint32_t st[V];
};

// inputs of group q (V consecutive envs from q * V; `have` of them exist): 16-byte vectors when V == 4 and the group is whole
template <int V, bool EXT_NOISE>
__device__ __forceinline__ void compact_load(const CompactArgs& a, int64_t q, int have, CompactIn<V, EXT_NOISE>& in) {
    const int64_t i0 = q * V;
    if (V == 4 && have == 4) {
#pragma unroll
        for (int k = 0; k < V; ++k) in.p[k] = a.pos[i0 + k];
        const float4 a01 = reinterpret_cast<const float4*>(a.action)[2 * q];
        const float4 a23 = reinterpret_cast<const float4*>(a.action)[2 * q + 1];
        const uint2 sv = reinterpret_cast<const uint2*>(a.status)[q];
        if constexpr (EXT_NOISE) {
#pragma unroll
            for (int k = 0; k < V; ++k) in.e[k] = a.noise[i0 + k];
        }
        const float fx[4] = {a01.x, a01.z, a23.x, a23.z}, fy[4] = {a01.y, a01.w, a23.y, a23.w};
        const uint32_t sw[4] = {sv.x & 0xffffu, sv.x >> 16, sv.y & 0xffffu, sv.y >> 16};
#pragma unroll
        for (int k = 0; k < V; ++k) {
            in.ax[k] = fx[k];
            in.ay[k] = fy[k];
            in.st[k] = sw[k];
        }
    } else if (V == 2 && have == 2) {
        // two envs per thread: one 16-byte action load, one 4-byte status load
#pragma unroll
        for (int k = 0; k < V; ++k) in.p[k] = a.pos[i0 + k];
        const float4 a01 = reinterpret_cast<const float4*>(a.action)[q];
        const uint32_t sv = reinterpret_cast<const uint32_t*>(a.status)[q];
        if constexpr (EXT_NOISE) {
#pragma unroll
            for (int k = 0; k < V; ++k) in.e[k] = a.noise[i0 + k];
        }
        in.ax[0] = a01.x; in.ay[0] = a01.y; in.ax[1 % V] = a01.z; in.ay[1 % V] = a01.w;
        in.st[0] = sv & 0xffffu; in.st[1 % V] = sv >> 16;
    } else {
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const int64_t i = i0 + (k < have ? k : 0);
            in.p[k] = a.pos[i];
            const float2 ak = a.action[i];
            in.ax[k] = ak.x;
            in.ay[k] = ak.y;
            in.st[k] = a.status[i];
            if constexpr (EXT_NOISE) in.e[k] = a.noise[i];
        }
    }
}

// V envs per thread: 4 in the bandwidth regime, 1 when the launch is latency-bound (a short dependent chain per thread
// matters more than wide accesses).  (Requesting the next pass's inputs before the f64 chain of the current one --
// software pipelining at 125 VGPRs -- measured 257.5 vs 258.0 us at 2^24 envs: not kept.)
template <int KIND, bool EXT_NOISE, int V>
__global__ __launch_bounds__(kBlock) void nav_step_compact_kernel(CompactArgs a) {
    __shared__ uint32_t wave_rows[V > 1 ? kBlock / 64 : 1][64];
    __shared__ double2 wave_draws[V > 1 ? kBlock / 64 : 1][64];
    const uint64_t ctr = rrl::effective_counter(a.counter, a.counter_dev);
    const int64_t nq = (a.n + V - 1) / V, stride = int64_t(gridDim.x) * kBlock;
    const int64_t n_pass = (nq + stride - 1) / stride;     // uniform trip count: the reset lists need whole waves
    const int64_t q_first = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    const auto group_of = [&](int64_t pass) {              // idle lanes shadow the last group and store nothing
        const int64_t q = pass * stride + q_first;
        return q < nq ? q : nq - 1;
    };
    const auto have_of = [&](int64_t q) { return int(a.n - q * V < V ? a.n - q * V : V); };
    for (int64_t pass = 0; pass < n_pass; ++pass) {
        const bool live = pass * stride + q_first < nq;
        const int64_t q = group_of(pass);
        const int64_t i0 = q * V;
        const int have = have_of(q);
        CompactIn<V, EXT_NOISE> in;
        compact_load<V, EXT_NOISE>(a, q, have, in);
        double2 p[V];
        float2 nobs[V];
        float rew[V];
        uint32_t st[V];
        bool fin[V];
#pragma unroll
        for (int k = 0; k < V; ++k) {
            double ex, ey;
            if constexpr (EXT_NOISE) {
                ex = in.e[k].x;
                ey = in.e[k].y;
            } else {
                rrl::normal_at(a.seed, uint32_t(i0 + k), rrl::kStreamStep, ctr, ex, ey);
            }
            double nx, ny, cost;
            rrl::nav_transition<KIND>(in.p[k].x, in.p[k].y, double(in.ax[k]), double(in.ay[k]), ex, ey, nx, ny, cost);
            const bool cons = rrl::in_obstacle<KIND>(nx, ny);
            const bool succ = cost > -4.0;
            const bool dn = succ | cons;
            uint32_t ti = (in.st[k] & RRL_STATUS_STEPS) + 1u;
            const bool epd = dn | (int32_t(ti) == a.horizon);
            if (ti > RRL_STATUS_STEPS) ti = RRL_STATUS_STEPS;
            nobs[k] = make_float2(float(nx), float(ny));
            rew[k] = float(cost);
            fin[k] = live & (k < have) & epd & (a.auto_reset != 0);
            st[k] = (fin[k] ? 0u : ti) | (dn ? RRL_STATUS_DONE : 0u) | (cons ? RRL_STATUS_CONSTRAINT : 0u) |
                    (succ ? RRL_STATUS_SUCCESS : 0u) | (epd ? RRL_STATUS_EP_DONE : 0u);
            p[k] = make_double2(nx, ny);
        }
        double w0[V], w1[V];
        if constexpr (V > 1) {
            wave_reset_draws<V>(a.seed, ctr, i0, fin, w0, w1, wave_rows[threadIdx.x >> 6], wave_draws[threadIdx.x >> 6]);
        }
#pragma unroll
        for (int k = 0; k < V; ++k)
            if (fin[k]) {
                if constexpr (V == 1) rrl::normal_at(a.seed, uint32_t(i0 + k), rrl::kStreamReset, ctr, w0[k], w1[k]);
                p[k] = make_double2(-50.0 + w0[k], 0.0 + w1[k]);     // START_STATE + randn(2), navigation1.py:92
                if (a.reset_obs) a.reset_obs[i0 + k] = make_float2(float(p[k].x), float(p[k].y));
            }
        if (!live) continue;
        if (V == 4 && have == 4) {
#pragma unroll
            for (int k = 0; k < V; ++k) a.pos[i0 + k] = p[k];
            reinterpret_cast<float4*>(a.next_obs)[2 * q] = make_float4(nobs[0].x, nobs[0].y, nobs[1 % V].x, nobs[1 % V].y);
            reinterpret_cast<float4*>(a.next_obs)[2 * q + 1] =
                make_float4(nobs[2 % V].x, nobs[2 % V].y, nobs[3 % V].x, nobs[3 % V].y);
            reinterpret_cast<float4*>(a.reward)[q] = make_float4(rew[0], rew[1 % V], rew[2 % V], rew[3 % V]);
            reinterpret_cast<uint2*>(a.status)[q] = make_uint2(st[0] | (st[1 % V] << 16), st[2 % V] | (st[3 % V] << 16));
        } else if (V == 2 && have == 2) {
#pragma unroll
            for (int k = 0; k < V; ++k) a.pos[i0 + k] = p[k];
            reinterpret_cast<float4*>(a.next_obs)[q] = make_float4(nobs[0].x, nobs[0].y, nobs[1 % V].x, nobs[1 % V].y);
            reinterpret_cast<float2*>(a.reward)[q] = make_float2(rew[0], rew[1 % V]);
            reinterpret_cast<uint32_t*>(a.status)[q] = st[0] | (st[1 % V] << 16);
        } else {
#pragma unroll
            for (int k = 0; k < V; ++k)
                if (k < have) {
                    a.pos[i0 + k] = p[k];
                    a.next_obs[i0 + k] = nobs[k];
                    a.reward[i0 + k] = rew[k];
                    a.status[i0 + k] = uint16_t(st[k]);
                }
        }
    }
    rrl::advance_counter(a.counter_dev, a.counter_inc);
}

// Streaming variant of the compact step (n >= 2^22): two envs per thread, and HOLD passes share ONE reset draw.  A pass
// stores what a reset does not touch (next_obs, reward) at once and parks its positions, status words and finished flags
// in LDS (40 B per thread and pass); after HOLD passes the wave draws the start states of all the rows it finished in
// them in one trip through normal_at(), applies them to the parked positions and stores positions and status words
// densely.  At a 1.2 % termination rate that is ~1 trip per 64 x 2 x HOLD envs instead of 0.79 per 128.  The passes of a
// group run in a loop that is NOT unrolled: the register footprint stays that of one pass (holding them in registers
// let the compiler interleave them: 118 VGPRs, no gain).
struct HeldPass {
    double2 p[2][kBlock];
    uint32_t st[kBlock];       // both status words
    uint32_t fin[kBlock];      // bit k: row k finished and takes a reset
};

template <int KIND, bool EXT_NOISE, int HOLD>
__global__ __launch_bounds__(kBlock) void nav_step_compact_hold_kernel(CompactArgs a) {
    constexpr int V = 2;
    __shared__ HeldPass held[HOLD];
    __shared__ uint32_t wave_rows[kBlock / 64][64];
    __shared__ double2 wave_draws[kBlock / 64][64];
    const int tid = threadIdx.x;
    const uint64_t ctr = rrl::effective_counter(a.counter, a.counter_dev);
    const int64_t nq = (a.n + V - 1) / V, stride = int64_t(gridDim.x) * kBlock;
    const int64_t n_pass = (nq + stride - 1) / stride;     // uniform trip count: the reset lists need whole waves
    const int64_t q_first = int64_t(blockIdx.x) * kBlock + tid;
    const auto group_of = [&](int64_t pass) {              // idle lanes shadow the last group and store nothing
        const int64_t q = pass * stride + q_first;
        return q < nq ? q : nq - 1;
    };
    const auto have_of = [&](int64_t q) { return int(a.n - q * V < V ? a.n - q * V : V); };
    for (int64_t pass0 = 0; pass0 < n_pass; pass0 += HOLD) {
        const int n_here = int(n_pass - pass0 < HOLD ? n_pass - pass0 : HOLD);      // wave-uniform
#pragma unroll 1
        for (int h = 0; h < n_here; ++h) {
            const int64_t pass = pass0 + h;
            const bool live = pass * stride + q_first < nq;
            const int64_t q = group_of(pass);
            const int64_t i0 = q * V;
            const int have = have_of(q);
            CompactIn<V, EXT_NOISE> in;
            compact_load<V, EXT_NOISE>(a, q, have, in);
            float2 nobs[V];
            float rew[V];
            uint32_t st[V], finbits = 0;
#pragma unroll
            for (int k = 0; k < V; ++k) {
                double ex, ey;
                if constexpr (EXT_NOISE) {
                    ex = in.e[k].x;
                    ey = in.e[k].y;
                } else {
                    rrl::normal_at(a.seed, uint32_t(i0 + k), rrl::kStreamStep, ctr, ex, ey);
                }
                double nx, ny, cost;
                rrl::nav_transition<KIND>(in.p[k].x, in.p[k].y, double(in.ax[k]), double(in.ay[k]), ex, ey, nx, ny, cost);
                const bool cons = rrl::in_obstacle<KIND>(nx, ny);
                const bool succ = cost > -4.0;
                const bool dn = succ | cons;
                uint32_t ti = (in.st[k] & RRL_STATUS_STEPS) + 1u;
                const bool epd = dn | (int32_t(ti) == a.horizon);
                if (ti > RRL_STATUS_STEPS) ti = RRL_STATUS_STEPS;
                nobs[k] = make_float2(float(nx), float(ny));
                rew[k] = float(cost);
                const bool f = live & (k < have) & epd & (a.auto_reset != 0);
                finbits |= uint32_t(f) << k;
                st[k] = (f ? 0u : ti) | (dn ? RRL_STATUS_DONE : 0u) | (cons ? RRL_STATUS_CONSTRAINT : 0u) |
                        (succ ? RRL_STATUS_SUCCESS : 0u) | (epd ? RRL_STATUS_EP_DONE : 0u);
                held[h].p[k][tid] = make_double2(nx, ny);
            }
            held[h].st[tid] = st[0] | (st[1] << 16);
            held[h].fin[tid] = finbits;
            if (live) {
                if (have == 2) {
                    reinterpret_cast<float4*>(a.next_obs)[q] = make_float4(nobs[0].x, nobs[0].y, nobs[1].x, nobs[1].y);
                    reinterpret_cast<float2*>(a.reward)[q] = make_float2(rew[0], rew[1]);
                } else {
                    a.next_obs[i0] = nobs[0];
                    a.reward[i0] = rew[0];
                }
            }
        }
        // one reset draw for the rows this wave finished in the group's passes (own LDS slots: no barrier needed)
        bool fin[HOLD * V];
        uint32_t row[HOLD * V];
#pragma unroll
        for (int h = 0; h < HOLD; ++h) {
            const uint32_t fb = h < n_here ? held[h].fin[tid] : 0u;
            const int64_t q = group_of(h < n_here ? pass0 + h : pass0);
#pragma unroll
            for (int k = 0; k < V; ++k) {
                fin[h * V + k] = (fb >> k) & 1u;
                row[h * V + k] = uint32_t(q * V + k);
            }
        }
        double w0[HOLD * V], w1[HOLD * V];
        wave_reset_draws<HOLD * V>(a.seed, ctr, row, fin, w0, w1, wave_rows[tid >> 6], wave_draws[tid >> 6]);
#pragma unroll
        for (int h = 0; h < HOLD; ++h) {
            if (h >= n_here) continue;
            const int64_t pass = pass0 + h;
            if (!(pass * stride + q_first < nq)) continue;
            const int64_t q = group_of(pass), i0 = q * V;
            const int have = have_of(q);
            double2 p[V];
#pragma unroll
            for (int k = 0; k < V; ++k) {
                p[k] = held[h].p[k][tid];
                if (fin[h * V + k]) {
                    p[k] = make_double2(-50.0 + w0[h * V + k], 0.0 + w1[h * V + k]);     // START_STATE + randn(2), navigation1.py:92
                    if (a.reset_obs) a.reset_obs[i0 + k] = make_float2(float(p[k].x), float(p[k].y));
                }
            }
            const uint32_t stw = held[h].st[tid];
            if (have == 2) {
                a.pos[i0] = p[0];
                a.pos[i0 + 1] = p[1];
                reinterpret_cast<uint32_t*>(a.status)[q] = stw;
            } else {
                a.pos[i0] = p[0];
                a.status[i0] = uint16_t(stw & 0xffffu);
            }
        }
    }
    rrl::advance_counter(a.counter_dev, a.counter_inc);
}

__global__ __launch_bounds__(kBlock) void nav_reset_kernel(int64_t n, double2* pos, float2* obs,
                                                           int32_t* t, const uint8_t* mask,
                                                           const double2* noise, uint64_t seed,
                                                           uint64_t counter,
                                                           const uint64_t* counter_dev) {
    const uint64_t ctr = rrl::effective_counter(counter, counter_dev);
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride) {
        if (mask && !mask[i]) continue;
        double z0, z1;
        if (noise) {
            z0 = noise[i].x;
            z1 = noise[i].y;
        } else {
            rrl::normal_at(seed, uint32_t(i), rrl::kStreamReset, ctr, z0, z1);
        }
        const double x = -50.0 + z0, y = 0.0 + z1;  // START_STATE + randn(2), navigation1.py:92
        pos[i] = make_double2(x, y);
        if (t) t[i] = 0;
        if (obs) obs[i] = make_float2(float(x), float(y));
    }
}

// T open-loop steps per env with the state in registers; per step only the action (8 B) is
// read and the requested outputs are written.
template <int KIND>
__global__ __launch_bounds__(kBlock) void nav_rollout_kernel(int64_t n, int32_t T, double2* pos,
                                                             const float2* actions, uint64_t seed,
                                                             uint64_t counter,
                                                             const uint64_t* counter_dev,
                                                             float2* obs_seq, float* reward_seq,
                                                             uint8_t* constraint_seq,
                                                             uint8_t* done_seq) {
    const uint64_t ctr = rrl::effective_counter(counter, counter_dev);
    const int64_t stride = int64_t(gridDim.x) * kBlock;
    for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride) {
        double2 p = pos[i];
        for (int32_t k = 0; k < T; ++k) {
            const int64_t o = int64_t(k) * n + i;
            const float2 act = actions[o];
            double ex, ey, nx, ny, cost;
            rrl::normal_at(seed, uint32_t(i), rrl::kStreamStep, ctr + uint64_t(k), ex, ey);
            rrl::nav_transition<KIND>(p.x, p.y, double(act.x), double(act.y), ex, ey, nx, ny, cost);
            const bool cons = rrl::in_obstacle<KIND>(nx, ny);
            if (obs_seq) obs_seq[o] = make_float2(float(nx), float(ny));
            if (reward_seq) reward_seq[o] = float(cost);
            if (constraint_seq) constraint_seq[o] = uint8_t(cons);
            if (done_seq) done_seq[o] = uint8_t((cost > -4.0) | cons);
            p = make_double2(nx, ny);
        }
        pos[i] = p;
    }
}


// ---- fused step + push: the navigation transition plugged into step_push_kernel (step_push.hpp) ----
template <int KIND>
struct NavEnv {
    static __device__ __forceinline__ rrl_step::Outcome step(const StepArgs& a, int64_t i, uint64_t ctr, double2 p,
                                                             float2 act, int32_t t_after) {
        double ex, ey, nx, ny, cost;
        rrl::normal_at(a.seed, uint32_t(i), rrl::kStreamStep, ctr, ex, ey);
        rrl::nav_transition<KIND>(p.x, p.y, double(act.x), double(act.y), ex, ey, nx, ny, cost);
        rrl_step::Outcome o;
        o.x = nx;
        o.y = ny;
        o.reward = float(cost);
        o.constraint = rrl::in_obstacle<KIND>(nx, ny);
        o.success = cost > -4.0;
        o.done = o.success | o.constraint;
        return o;
    }
    static __device__ __forceinline__ void reset(const StepArgs& a, int64_t i, uint64_t ctr, double& x, double& y) {
        double z0, z1;
        rrl::normal_at(a.seed, uint32_t(i), rrl::kStreamReset, ctr, z0, z1);
        x = -50.0 + z0;
        y = 0.0 + z1;
    }
};

// ---- offline constraint data (env/navigation1.py:133-164, env/navigation2.py:133-243) ----
struct OfflinePlan {
    int64_t n_roll, n0, n1;
};

inline OfflinePlan offline_plan(int env_kind, int64_t num) {
    OfflinePlan p{0, 0, 0};
    if (env_kind == RRL_ENV_NAV1) {
        p.n_roll = num / 10;
    } else {
        p.n0 = num / 10 / 3;
        p.n1 = num / 10 / 4;
        p.n_roll = p.n0 + 4 * p.n1;
    }
    return p;
}

__device__ __forceinline__ void offline_uniform2(uint64_t seed, uint32_t row, uint64_t k,
                                                 uint32_t hi, double& u0, double& u1) {
    const rrl::Bits128 b = rrl::philox_at(seed, row, rrl::kStreamOffline, k | (uint64_t(hi) << 32));
    u0 = rrl::unit_open(b.lo);
    u1 = rrl::unit_open(b.hi);
}

// Simulates rollout `i`; when WRITE, stores its rows starting at row `base`. Returns its length.
template <int KIND, bool WRITE>
__device__ __forceinline__ int offline_rollout(int64_t i, int64_t n0, int64_t n1, uint64_t seed,
                                               int64_t base, float2* s, float2* a, float* c,
                                               float2* s2, float* m) {
    const uint32_t row = uint32_t(i);
    double u0, u1, v0, v1, x, y;
    offline_uniform2(seed, row, 0, 0, u0, u1);
    offline_uniform2(seed, row, 1, 0, v0, v1);
    int phase = 0;
    if constexpr (KIND == 0) {
        x = -80.0 + 130.0 * u1;
        y = (u0 < 0.5) ? (-5.0 + 3.0 * v0) : (2.0 + 3.0 * v0);
    } else {
        phase = (i < n0) ? 0 : 1 + int((i - n0) / (n1 > 0 ? n1 : 1));
        if (phase == 0) {
            x = -40.0 + 50.0 * u0;
            y = -25.0 + 50.0 * u1;
            for (uint32_t r = 0; rrl::in_obstacle<KIND>(x, y); ++r) {
                double q0, q1;
                offline_uniform2(seed, row, 0, 1 + r, q0, q1);
                x = -40.0 + 50.0 * q0;
                y = -25.0 + 50.0 * q1;
            }
        } else if (phase == 1) {
            x = -35.0 + 5.0 * u0;
            y = -12.0 + 24.0 * u1;
        } else if (phase == 2) {
            x = -20.0 + 5.0 * u0;
            y = -12.0 + 24.0 * u1;
        } else if (phase == 3) {
            x = -30.0 + 10.0 * u0;
            y = 10.0 + 5.0 * u1;
        } else {
            x = -30.0 + 10.0 * u0;
            y = -15.0 + 5.0 * u1;
        }
    }
    int len = 0;
    for (int j = 0; j < 10; ++j) {
        double z0, z1, q0, q1, e0, e1;
        rrl::normal_at(seed, row, rrl::kStreamOffline, uint64_t(2 + 3 * j), z0, z1);
        offline_uniform2(seed, row, uint64_t(3 + 3 * j), 0, q0, q1);
        rrl::normal_at(seed, row, rrl::kStreamOffline, uint64_t(4 + 3 * j), e0, e1);
        double ax = rrl::clamp_unit(z0), ay = rrl::clamp_unit(z1);
        if (phase == 1) ax = 0.5 + 0.5 * q0;
        else if (phase == 2) ax = -1.0 + 0.5 * q0;
        else if (phase == 3) ay = -1.0 + 0.5 * q0;
        else if (phase == 4) ay = 0.5 + 0.5 * q0;
        const float axf = float(ax), ayf = float(ay);
        double nx, ny, cost;   // the transition takes the float64 action, as the reference (navigation1.py:149-150)
        rrl::nav_transition<KIND>(x, y, ax, ay, e0, e1, nx, ny, cost);
        const bool cons = rrl::in_obstacle<KIND>(nx, ny);
        if constexpr (WRITE) {
            const int64_t w = base + len;
            s[w] = make_float2(float(x), float(y));
            a[w] = make_float2(axf, ayf);
            c[w] = cons ? 1.0f : 0.0f;
            s2[w] = make_float2(float(nx), float(ny));
            m[w] = cons ? 0.0f : 1.0f;
        }
        ++len;
        x = nx;
        y = ny;
        if (cons) break;
    }
    return len;
}

template <int KIND>
__global__ __launch_bounds__(kBlock) void offline_count_kernel(int64_t n_roll, int64_t n0,
                                                               int64_t n1, uint64_t seed,
                                                               int32_t* lens) {
    const int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    if (i < n_roll)
        lens[i] = offline_rollout<KIND, false>(i, n0, n1, seed, 0, nullptr, nullptr, nullptr,
                                               nullptr, nullptr);
}

// single workgroup: in-place exclusive scan of lens[0..n), total -> lens[n] and *count_dev
__global__ __launch_bounds__(1024) void exclusive_scan_kernel(int32_t* lens, int64_t n,
                                                              int64_t* count_dev) {
    __shared__ int32_t part[1024];
    __shared__ int32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < n; base += 1024) {
        const int64_t i = base + threadIdx.x;
        const int32_t v = (i < n) ? lens[i] : 0;
        part[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const int32_t add = (int(threadIdx.x) >= off) ? part[threadIdx.x - off] : 0;
            __syncthreads();
            part[threadIdx.x] += add;
            __syncthreads();
        }
        const int32_t incl = part[threadIdx.x];
        const int32_t c0 = carry;
        if (i < n) lens[i] = c0 + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = c0 + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        lens[n] = carry;
        if (count_dev) *count_dev = carry;
    }
}

template <int KIND>
__global__ __launch_bounds__(kBlock) void offline_write_kernel(int64_t n_roll, int64_t n0,
                                                               int64_t n1, uint64_t seed,
                                                               const int32_t* offs, int64_t capacity,
                                                               float2* s, float2* a, float* c,
                                                               float2* s2, float* m) {
    const int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= n_roll) return;
    const int64_t base = offs[i];
    if (int64_t(offs[i + 1]) > capacity) return;  // caller sized the arrays too small: drop
    offline_rollout<KIND, true>(i, n0, n1, seed, base, s, a, c, s2, m);
}

__global__ void counter_add_kernel(uint64_t* ctr, uint64_t inc) { *ctr += inc; }

}  // namespace

thread_local int rrl_host::last_hip_error = 0;

extern "C" {

int rrl_pack_clear(void) { return rrl_pack::clear(); }

int rrl_abi_version(void) { return 4; }   // 2: pos_cnt carries a second count level (RRL_POS_CNT_LEN); 3: rrl_replay_t.pinned; 4: RRL_DRAW_DEMO_SHARE

int rrl_last_hip_error(void) { return rrl_host::last_hip_error; }

int rrl_counter_add(uint64_t* ctr, uint64_t inc, void* stream) {
    if (!ctr) return RRL_EINVAL;
    hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, ctr, inc);
    return check_launch();
}

int rrl_nav_step(int env_kind, int64_t n, double* pos, const float* action, const double* noise,
                 uint64_t seed, uint64_t counter, uint64_t* counter_dev, uint64_t counter_inc,
                 float* next_obs,
                 float* obs, float* reward, uint8_t* done, uint8_t* constraint, uint8_t* success,
                 uint8_t* ep_done, int32_t* t, int32_t horizon, int auto_reset, void* stream) {
    if (env_kind != RRL_ENV_NAV1 && env_kind != RRL_ENV_NAV2) return RRL_EINVAL;
    if (n < 0 || n > 0xffffffffLL) return RRL_ERANGE;
    if (!pos || !action || !next_obs || !reward || !done || !constraint || !success || !t)
        return RRL_EINVAL;
    if (n == 0) return RRL_OK;
    StepArgs a{n, (double2*)pos, (const float2*)action, (const double2*)noise, seed, counter,
               counter_dev, counter_inc, (float2*)next_obs, (float2*)obs, reward, done, constraint, success,
               ep_done, t, horizon, auto_reset};
    const dim3 block(kBlock);
    hipStream_t st = (hipStream_t)stream;
    // bandwidth regime: two envs per thread, vector accesses (needs the alignment torch gives whole tensors)
    auto al = [](const void* p, uintptr_t m) { return (reinterpret_cast<uintptr_t>(p) & (m - 1)) == 0; };
    const bool vec2 = n >= (1 << 22) && (n & 1) == 0 && al(action, 16) && al(next_obs, 16) && al(obs, 16) &&
                      al(reward, 8) && al(t, 8) && al(done, 2) && al(constraint, 2) && al(success, 2) && al(ep_done, 2);
    if (vec2) {
        const dim3 grid(grid_for(n >> 1));
        // two passes share one reset draw (same scheme as the compact kernel; the plain two-env kernel: 236 vs 225 us at 2^24)
        if (env_kind == RRL_ENV_NAV1) {
            if (noise) hipLaunchKernelGGL((nav_step2_hold_kernel<0, true, 2>), grid, block, 0, st, a);
            else hipLaunchKernelGGL((nav_step2_hold_kernel<0, false, 2>), grid, block, 0, st, a);
        } else {
            if (noise) hipLaunchKernelGGL((nav_step2_hold_kernel<1, true, 2>), grid, block, 0, st, a);
            else hipLaunchKernelGGL((nav_step2_hold_kernel<1, false, 2>), grid, block, 0, st, a);
        }
        return check_launch();
    }
    // bandwidth regime: four envs per thread, 16-byte accesses (needs the vector alignment torch gives whole tensors)
    const bool vec4 = n >= (1 << 19) && (n & 3) == 0 && al(action, 16) && al(next_obs, 16) && al(obs, 16) &&
                      al(reward, 16) && al(t, 16) && al(done, 4) && al(constraint, 4) && al(success, 4) &&
                      al(ep_done, 4);
    if (vec4) {
        // 2048 workgroups (8 per CU) although only 5 are resident at ~90 VGPRs: measured 262 vs 272 us for one full round
        const dim3 grid(grid_for(n >> 2));
        if (env_kind == RRL_ENV_NAV1) {
            if (noise) hipLaunchKernelGGL((nav_step4_kernel<0, true>), grid, block, 0, st, a);
            else hipLaunchKernelGGL((nav_step4_kernel<0, false>), grid, block, 0, st, a);
        } else {
            if (noise) hipLaunchKernelGGL((nav_step4_kernel<1, true>), grid, block, 0, st, a);
            else hipLaunchKernelGGL((nav_step4_kernel<1, false>), grid, block, 0, st, a);
        }
        return check_launch();
    }
    const dim3 grid(grid_for(n));
    if (env_kind == RRL_ENV_NAV1) {
        if (noise) hipLaunchKernelGGL((nav_step_kernel<0, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((nav_step_kernel<0, false>), grid, block, 0, st, a);
    } else {
        if (noise) hipLaunchKernelGGL((nav_step_kernel<1, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((nav_step_kernel<1, false>), grid, block, 0, st, a);
    }
    return check_launch();
}

int rrl_nav_step_compact(int env_kind, int64_t n, double* pos, const float* action, const double* noise,
                         uint64_t seed, uint64_t counter, uint64_t* counter_dev, uint64_t counter_inc,
                         float* next_obs, float* reset_obs, float* reward, uint16_t* status, int32_t horizon,
                         int auto_reset, void* stream) {
    if (env_kind != RRL_ENV_NAV1 && env_kind != RRL_ENV_NAV2) return RRL_EINVAL;
    if (n < 0 || n > 0xffffffffLL || horizon < 1 || horizon > int32_t(RRL_STATUS_STEPS)) return RRL_ERANGE;
    if (!pos || !action || !next_obs || !reward || !status) return RRL_EINVAL;
    auto al = [](const void* p, uintptr_t m) { return (reinterpret_cast<uintptr_t>(p) & (m - 1)) == 0; };
    if (!al(pos, 16) || !al(action, 16) || !al(noise, 16) || !al(next_obs, 16) || !al(reset_obs, 8) ||
        !al(reward, 16) || !al(status, 8))
        return RRL_EINVAL;
    if (n == 0) return RRL_OK;
    CompactArgs a{n, (double2*)pos, (const float2*)action, (const double2*)noise, seed, counter, counter_dev,
                  counter_inc, (float2*)next_obs, (float2*)reset_obs, reward, status, horizon, auto_reset};
    const dim3 block(kBlock);
    hipStream_t st = (hipStream_t)stream;
    const auto go = [&](auto kind, auto ext) {
        constexpr int K = decltype(kind)::value;
        constexpr bool E = decltype(ext)::value;
        // envs per thread, measured at 2^24 envs in one process (profiles/nav_step_probe.py): without resets
        // 1 and 2 run at 192 us, 4 at 220 us (115 VGPRs, 236 SGPR-spill reads per pass); with resets 241 / 231 / 236 us --
        // the once-per-wave reset draw batches 64 V envs, and at V = 1 every finished lane costs its wave a second chain.
        // Below 2^22 envs four per thread: 2^20 envs are then ONE round of 4 waves per SIMD (24.5 us; two per thread 34 us).
        const int v = n < (1 << 18) ? 1 : (n < (1 << 22) ? 4 : 2);
        if (v == 1) {                 // also the latency regime: a short dependent chain per thread, resets inline
            hipLaunchKernelGGL((nav_step_compact_kernel<K, E, 1>), dim3(grid_for(n)), block, 0, st, a);
        } else if (v == 2) {
            // passes that share one reset draw (parked in LDS): 1 -> 236 us, 2 -> 225 us, 3 -> 228 us, 4 -> 240 us at 2^24
            // envs (the larger groups cost registers in the apply phase)
            const dim3 grid2(grid_for((n + 1) >> 1));
            hipLaunchKernelGGL((nav_step_compact_hold_kernel<K, E, 2>), grid2, block, 0, st, a);
        } else {
            hipLaunchKernelGGL((nav_step_compact_kernel<K, E, 4>), dim3(grid_for((n + 3) >> 2)), block, 0, st, a);
        }
    };
    using std::integral_constant;
    if (env_kind == RRL_ENV_NAV1) {
        if (noise) go(integral_constant<int, 0>{}, integral_constant<bool, true>{});
        else go(integral_constant<int, 0>{}, integral_constant<bool, false>{});
    } else {
        if (noise) go(integral_constant<int, 1>{}, integral_constant<bool, true>{});
        else go(integral_constant<int, 1>{}, integral_constant<bool, false>{});
    }
    return check_launch();
}

int rrl_nav_reset(int env_kind, int64_t n, double* pos, float* obs, int32_t* t,
                  const uint8_t* mask, const double* noise, uint64_t seed, uint64_t counter,
                  const uint64_t* counter_dev, void* stream) {
    if (env_kind != RRL_ENV_NAV1 && env_kind != RRL_ENV_NAV2) return RRL_EINVAL;
    if (n < 0 || n > 0xffffffffLL) return RRL_ERANGE;
    if (!pos) return RRL_EINVAL;
    if (n == 0) return RRL_OK;
    hipLaunchKernelGGL(nav_reset_kernel, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream,
                       n, (double2*)pos, (float2*)obs, t, mask, (const double2*)noise, seed,
                       counter, counter_dev);
    return check_launch();
}

int rrl_nav_rollout(int env_kind, int64_t n, int32_t T, double* pos, const float* actions,
                    uint64_t seed, uint64_t counter, const uint64_t* counter_dev, float* obs_seq,
                    float* reward_seq, uint8_t* constraint_seq, uint8_t* done_seq, void* stream) {
    if (env_kind != RRL_ENV_NAV1 && env_kind != RRL_ENV_NAV2) return RRL_EINVAL;
    if (n < 0 || n > 0xffffffffLL || T < 0) return RRL_ERANGE;
    if (!pos || !actions) return RRL_EINVAL;
    if (n == 0 || T == 0) return RRL_OK;
    const dim3 grid(grid_for(n)), block(kBlock);
    hipStream_t st = (hipStream_t)stream;
    if (env_kind == RRL_ENV_NAV1)
        hipLaunchKernelGGL((nav_rollout_kernel<0>), grid, block, 0, st, n, T, (double2*)pos,
                           (const float2*)actions, seed, counter, counter_dev, (float2*)obs_seq,
                           reward_seq, constraint_seq, done_seq);
    else
        hipLaunchKernelGGL((nav_rollout_kernel<1>), grid, block, 0, st, n, T, (double2*)pos,
                           (const float2*)actions, seed, counter, counter_dev, (float2*)obs_seq,
                           reward_seq, constraint_seq, done_seq);
    return check_launch();
}

int64_t rrl_nav_offline_rollouts(int env_kind, int64_t num_transitions) {
    if (env_kind != RRL_ENV_NAV1 && env_kind != RRL_ENV_NAV2) return RRL_EINVAL;
    if (num_transitions < 0) return RRL_EINVAL;
    return offline_plan(env_kind, num_transitions).n_roll;
}

int rrl_nav_offline(int env_kind, int64_t num_transitions, uint64_t seed, float* s, float* a,
                    float* c, float* s2, float* m, int64_t capacity, int64_t* count_dev,
                    int32_t* scratch, void* stream) {
    if (env_kind != RRL_ENV_NAV1 && env_kind != RRL_ENV_NAV2) return RRL_EINVAL;
    if (num_transitions < 0 || !s || !a || !c || !s2 || !m || !scratch) return RRL_EINVAL;
    const OfflinePlan p = offline_plan(env_kind, num_transitions);
    if (p.n_roll > 0x7fffffffLL / 10) return RRL_ERANGE;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)((p.n_roll + kBlock - 1) / kBlock > 0 ? (p.n_roll + kBlock - 1) / kBlock : 1)),
        block(kBlock);
    if (env_kind == RRL_ENV_NAV1)
        hipLaunchKernelGGL((offline_count_kernel<0>), grid, block, 0, st, p.n_roll, p.n0, p.n1, seed, scratch);
    else
        hipLaunchKernelGGL((offline_count_kernel<1>), grid, block, 0, st, p.n_roll, p.n0, p.n1, seed, scratch);
    hipLaunchKernelGGL(exclusive_scan_kernel, dim3(1), dim3(1024), 0, st, scratch, p.n_roll, count_dev);
    if (env_kind == RRL_ENV_NAV1)
        hipLaunchKernelGGL((offline_write_kernel<0>), grid, block, 0, st, p.n_roll, p.n0, p.n1, seed,
                           scratch, capacity, (float2*)s, (float2*)a, c, (float2*)s2, m);
    else
        hipLaunchKernelGGL((offline_write_kernel<1>), grid, block, 0, st, p.n_roll, p.n0, p.n1, seed,
                           scratch, capacity, (float2*)s, (float2*)a, c, (float2*)s2, m);
    return check_launch();
}

static int nav_step_push_launch(int env_kind, const rrl_step::StepPushArgs& p, int64_t n, void* stream) {
    if (env_kind == RRL_ENV_NAV1) rrl_step::launch<NavEnv<0>>(p, n, (hipStream_t)stream);
    else rrl_step::launch<NavEnv<1>>(p, n, (hipStream_t)stream);
    return check_launch();
}

int rrl_nav_step_push(int env_kind, int64_t n, double* pos, int32_t* t, float* obs,
                      const float* task_action, const float* real_action, const uint8_t* recovery,
                      uint64_t seed, uint64_t counter, uint64_t* counter_dev, uint64_t counter_inc,
                      int32_t horizon, int auto_reset, float reward_penalty, int push_real_action,
                      const rrl_replay_t* memory, const rrl_replay_t* recovery_memory, float* next_obs,
                      float* reward, uint8_t* done, uint8_t* constraint, uint8_t* success, uint8_t* ep_done,
                      uint64_t* stats, double* reward_sums, float* ep_reward, void* stream) {
    if (env_kind != RRL_ENV_NAV1 && env_kind != RRL_ENV_NAV2) return RRL_EINVAL;
    rrl_step::StepPushArgs p;
    const int rc = rrl_step::fill_args(p, n, pos, t, obs, task_action, 2, real_action, recovery, nullptr, seed, counter,
                                       counter_dev, counter_inc, horizon, auto_reset, reward_penalty, push_real_action,
                                       memory, recovery_memory, next_obs, reward, done, constraint, success, ep_done,
                                       stats, reward_sums, ep_reward);
    if (rc != RRL_OK || n == 0) return rc;
    return nav_step_push_launch(env_kind, p, n, stream);
}

int rrl_nav_step_push_select(int env_kind, int64_t n, double* pos, int32_t* t, float* obs, const float* task_action,
                             int ld_task, const float* z, int z_n_part, long long z_part_stride,
                             float eps_safe, const float* rec_action, const rrl_policy_head_t* rec_head,
                             float* real_action,
                             uint8_t* recovery, uint64_t seed, uint64_t counter, uint64_t* counter_dev,
                             uint64_t counter_inc, int32_t horizon, int auto_reset, float reward_penalty,
                             int push_real_action, const rrl_replay_t* memory, const rrl_replay_t* recovery_memory,
                             float* next_obs, float* reward, uint8_t* done, uint8_t* constraint, uint8_t* success,
                             uint8_t* ep_done, uint64_t* stats, double* reward_sums, float* ep_reward, void* stream) {
    if (env_kind != RRL_ENV_NAV1 && env_kind != RRL_ENV_NAV2) return RRL_EINVAL;
    rrl_step::StepPushArgs p;
    const rrl_step::SelectIn sel{z, z_n_part, z_part_stride, eps_safe, rec_action, rec_head, real_action, recovery};
    const int rc = rrl_step::fill_args(p, n, pos, t, obs, task_action, ld_task, nullptr, nullptr, &sel, seed, counter,
                                       counter_dev, counter_inc, horizon, auto_reset, reward_penalty, push_real_action,
                                       memory, recovery_memory, next_obs, reward, done, constraint, success, ep_done,
                                       stats, reward_sums, ep_reward);
    if (rc != RRL_OK || n == 0) return rc;
    return nav_step_push_launch(env_kind, p, n, stream);
}

int rrl_nav_step_push_packed(int S, int env_kind, const rrl_step_push_t* a, void* stream) {
    if ((env_kind != RRL_ENV_NAV1 && env_kind != RRL_ENV_NAV2) || S <= 0 || S > rrl_pack::kMaxSeeds || !a) return RRL_EINVAL;
    // one seed: the packed launch IS the solo launch (argument block in the kernel arguments, no plan)
    if (S == 1) return rrl_nav_step_push_x(env_kind, &a[0], stream);
    rrl_pack::Key key;
    key.pod(6);
    key.pod(S);
    key.pod(env_kind);
    for (int s = 0; s < S; ++s) {
        key.pod(a[s]);
        if (a[s].memory) key.pod(*a[s].memory);
        if (a[s].recovery_memory) key.pod(*a[s].recovery_memory);
        if (a[s].sel_rec_head) key.pod(*a[s].sel_rec_head);
    }
    hipStream_t st = (hipStream_t)stream;
    rrl_pack::Plan* plan = rrl_pack::lookup(key);
    if (!plan) {
        std::vector<rrl_step::StepPushArgs> ps(S);
        rrl_pack::Idx ix;
        ix.S = S;
        ix.first[0] = 0;
        const int regime = rrl_step::regime_of(a[0].n);
        for (int s = 0; s < S; ++s) {
            const int rc = rrl_step::fill_args(ps[s], &a[s]);
            if (rc != RRL_OK) return rc;
            if (a[s].n <= 0 || rrl_step::regime_of(a[s].n) != regime) return RRL_EINVAL;
            ix.first[s + 1] = ix.first[s] + rrl_step::grid_cover(a[s].n);
        }
        for (int s = S; s < rrl_pack::kMaxSeeds; ++s) ix.first[s + 1] = ix.first[S];
        plan = rrl_pack::store(key, ps.data(), sizeof(rrl_step::StepPushArgs) * S, st);
        if (!plan) return rrl_pack::store_error();
        plan->grid = rrl_pack::finish(ix);
        plan->ix = ix;
        plan->i0 = regime;
    }
    const auto* dev = (const rrl_step::StepPushArgs*)plan->dev;
    if (env_kind == RRL_ENV_NAV1) rrl_step::launch_pack<NavEnv<0>>(dev, plan->ix, plan->grid, plan->i0, st);
    else rrl_step::launch_pack<NavEnv<1>>(dev, plan->ix, plan->grid, plan->i0, st);
    return check_launch();
}

int rrl_nav_step_push_x(int env_kind, const rrl_step_push_t* a, void* stream) {
    if (env_kind != RRL_ENV_NAV1 && env_kind != RRL_ENV_NAV2) return RRL_EINVAL;
    rrl_step::StepPushArgs p;
    const int rc = rrl_step::fill_args(p, a);
    if (rc != RRL_OK || a->n == 0) return rc;
    return nav_step_push_launch(env_kind, p, a->n, stream);
}

}  // extern "C"
