// rrl_device.hpp -- device-side primitives shared by the gfx950 kernels:
// Philox4x32-10, bit-reproducible uniform / normal generation, obstacle tables.
//
// "Deterministic math" contract: every floating-point result below is produced
// by IEEE-754 double add / mul / div / sqrt / fma only, in a fixed order, with FMA
// contraction disabled, so that the values are identical on the GPU and in the CPU checker.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace rrl {

constexpr uint32_t kStreamStep = 0, kStreamReset = 1, kStreamOffline = 2, kStreamSample = 3,
                   kStreamSampleNeg = 4, kStreamCem = 5, kStreamAction = 6, kStreamPlan = 7, kStreamNoise = 8;

struct Bits128 {
    uint64_t lo, hi;
};

// Philox4x32-10, counter (c0..c3), key (k0,k1).
__device__ __forceinline__ Bits128 philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                          uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32x32->64 product per multiplier (v_mad_u64_u32) instead of a mul_hi + mul_lo pair: 20 quarter-rate
        // multiplies per call instead of 40, same bits
        const uint64_t p0 = uint64_t(0xD2511F53u) * c0, p1 = uint64_t(0xCD9E8D57u) * c2;
        const uint32_t h0 = uint32_t(p0 >> 32), l0 = uint32_t(p0);
        const uint32_t h1 = uint32_t(p1 >> 32), l1 = uint32_t(p1);
        c0 = h1 ^ c1 ^ k0;
        c1 = l1;
        c2 = h0 ^ c3 ^ k1;
        c3 = l0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return Bits128{(uint64_t(c1) << 32) | c0, (uint64_t(c3) << 32) | c2};
}

__device__ __forceinline__ Bits128 philox_at(uint64_t seed, uint32_t row, uint32_t stream,
                                             uint64_t counter) {
    return philox(row, stream, uint32_t(counter), uint32_t(counter >> 32), uint32_t(seed),
                  uint32_t(seed >> 32));
}

// (k + 1/2) / 2^52 for the top 52 bits k: exact in double, strictly inside (0,1).
// Built without an integer -> double conversion: the bit pattern 0x3ff0... | k IS 1 + k 2^-52, and subtracting 1 - 2^-53
// (both operands within a factor of two: the difference is exact) leaves (2 k + 1) 2^-53 -- the value the specification's
// (double(k) + 0.5) * 2^-52 has, in two instructions instead of six.
__device__ __forceinline__ double unit_open(uint64_t bits) {
    return __longlong_as_double((long long)(0x3ff0000000000000ULL | (bits >> 12))) - 0x1.fffffffffffffp-1;
}
// (f + 1/2) / 2^49 for f < 2^49, the same way: 1 + f 2^-49 minus 1 - 2^-50
__device__ __forceinline__ double unit_open49(uint64_t f) {
    return __longlong_as_double((long long)(0x3ff0000000000000ULL | (f << 3))) - 0x1.ffffffffffff8p-1;
}

template <int N>
__device__ __forceinline__ double horner(const double (&coef)[N], double z) {
    double acc = coef[0];
#pragma unroll
    for (int i = 1; i < N; ++i) acc = acc * z + coef[i];
    return acc;
}

// natural log on (0,1): split exponent, fold mantissa into (sqrt(1/2), sqrt(2)], atanh series.
__device__ __forceinline__ double log_unit(double u) {
    constexpr double kAtanh[11] = {1.0 / 21.0, 1.0 / 19.0, 1.0 / 17.0, 1.0 / 15.0,
                                   1.0 / 13.0, 1.0 / 11.0, 1.0 / 9.0,  1.0 / 7.0,
                                   1.0 / 5.0,  1.0 / 3.0,  1.0};
    const uint64_t raw = (uint64_t)__double_as_longlong(u);
    int expo = int((raw >> 52) & 0x7ffu) - 1023;
    double mant =
        __longlong_as_double((long long)((raw & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL));
    if (mant > 1.4142135623730951) {
        mant = mant * 0.5;
        expo += 1;
    }
    const double s = (mant - 1.0) / (mant + 1.0);
    const double p = horner(kAtanh, s * s);
    return double(expo) * 0.6931471805599453 + 2.0 * s * p;
}

// sin / cos on (0, pi/4] by Taylor polynomials in x^2.
__device__ __forceinline__ void sincos_octant(double x, double& sn, double& cs) {
    constexpr double kSin[9] = {-1.0 / 355687428096000.0, 1.0 / 1307674368000.0,
                                -1.0 / 6227020800.0,      1.0 / 39916800.0,
                                -1.0 / 362880.0,          1.0 / 5040.0,
                                -1.0 / 120.0,             -1.0 / 6.0,
                                1.0};
    constexpr double kCos[9] = {1.0 / 20922789888000.0, -1.0 / 87178291200.0,
                                1.0 / 479001600.0,      -1.0 / 3628800.0,
                                1.0 / 40320.0,          -1.0 / 720.0,
                                1.0 / 24.0,             -0.5,
                                1.0};
    const double z = x * x;
    sn = x * horner(kSin, z);
    cs = horner(kCos, z);
}

// Two independent N(0,1) draws from 128 random bits (Box-Muller, integer octant reduction).
__device__ __forceinline__ void normal_pair(Bits128 b, double& z0, double& z1) {
    const double radius = sqrt(-2.0 * log_unit(unit_open(b.lo)));
    const uint64_t k = b.hi >> 12;
    const uint32_t octant = uint32_t(k >> 49);
    uint64_t frac = k & ((1ULL << 49) - 1);
    if (octant & 1u) frac = ((1ULL << 49) - 1) - frac;
    const double phi = unit_open49(frac) * 0.7853981633974483;     // = (double(frac) + 0.5) * 2^-49 * (pi / 4)
    double sn, cs;
    sincos_octant(phi, sn, cs);
    const double c = (octant & 1u) ? sn : cs;
    const double s = (octant & 1u) ? cs : sn;
    const double rc = radius * c, rs = radius * s;
    switch (octant >> 1) {
        case 0: z0 = rc; z1 = rs; break;
        case 1: z0 = -rs; z1 = rc; break;
        case 2: z0 = -rc; z1 = -rs; break;
        default: z0 = rs; z1 = -rc; break;
    }
}

__device__ __forceinline__ void normal_at(uint64_t seed, uint32_t row, uint32_t stream,
                                          uint64_t counter, double& z0, double& z1) {
    normal_pair(philox_at(seed, row, stream, counter), z0, z1);
}

// ---- obstacles: closed axis-aligned boxes (env/obstacle.py:13-15,44-45) ----
template <int KIND>
__device__ __forceinline__ bool in_obstacle(double x, double y) {
    if constexpr (KIND == 0) {  // env/navigation1.py:41-42
        const bool xr = (-100.0 <= x) & (x <= 150.0);
        const bool top = xr & (5.0 <= y) & (y <= 10.0);
        const bool bot = xr & (-10.0 <= y) & (y <= -5.0);
        const bool left = (-100.0 <= x) & (x <= -80.0) & (-10.0 <= y) & (y <= 10.0);
        return top | bot | left;
    } else {  // env/navigation2.py:41
        return (-30.0 <= x) & (x <= -20.0) & (-7.5 <= y) & (y <= 7.5);
    }
}

__device__ __forceinline__ double clamp_unit(double v) {
    return v < -1.0 ? -1.0 : (v > 1.0 ? 1.0 : v);
}

// One transition of Navigation1/2 (env/navigation1.py:71-89,99-110).
template <int KIND>
__device__ __forceinline__ void nav_transition(double x, double y, double ax, double ay, double ex,
                                               double ey, double& nx, double& ny, double& cost) {
    ax = clamp_unit(ax);
    ay = clamp_unit(ay);
    if (in_obstacle<KIND>(x, y)) {  // stuck inside an obstacle, no noise applied (:100-102)
        nx = x;
        ny = y;
    } else {
        nx = (x + ax) + 0.05 * ex;
        ny = (y + ay) + 0.05 * ey;
    }
    // -||s|| of the OLD state; the reference's norm is sqrt(ddot) with an FMA-accumulated dot
    cost = -sqrt(fma(y, y, x * x));
}

template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_move<0x128>(v);    // row_ror:8
    v += dpp_move<0x124>(v);    // row_ror:4
    v += dpp_move<0x122>(v);    // row_ror:2
    v += dpp_move<0x121>(v);    // row_ror:1
    return v;
}

__device__ __forceinline__ uint64_t effective_counter(uint64_t counter, const uint64_t* dev) {
    return dev ? counter + *dev : counter;
}

// tick += inc by the last workgroup to finish (dev = {tick, ticket}); every workgroup read the
// tick before taking its ticket, so none can observe the new value.
// `blocks` = number of workgroups that call this for `dev` (the whole grid, or the part of a grouped launch that
// serves this counter)
__device__ __forceinline__ void advance_counter_blocks(uint64_t* dev, uint64_t inc, unsigned blocks) {
    if (!dev || !inc) return;
    __syncthreads();
    if (threadIdx.x == 0) {
        // no fence: the ticket only orders this workgroup's READ of dev[0] (already consumed) before the
        // last arriver's write; the write itself is published by the kernel boundary
        const unsigned long long ticket = atomicAdd((unsigned long long*)&dev[1], 1ULL);
        if (ticket == blocks - 1) {
            dev[0] += inc;
            dev[1] = 0;
        }
    }
}

// ... and when ONE workgroup consumes the tick (the samplers): no ticket, no second read -- thread 0 stores what it read + inc
// (`ctr` = effective_counter(counter, dev) as read by this workgroup).  The ticket's round trip and the re-read of dev[0]
// were ~1.5 us in the middle of a 7 us kernel.
__device__ __forceinline__ void advance_counter_single(uint64_t* dev, uint64_t inc, uint64_t counter, uint64_t ctr) {
    if (!dev || !inc) return;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // this wave's read of the tick has returned ...
    __syncthreads();                                                  // ... and so has every other wave's
    if (threadIdx.x == 0) dev[0] = ctr - counter + inc;
}

__device__ __forceinline__ void advance_counter(uint64_t* dev, uint64_t inc) {
    advance_counter_blocks(dev, inc, gridDim.x);
}

}  // namespace rrl
