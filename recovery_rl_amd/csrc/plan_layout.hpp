// plan_layout.hpp -- layout of the packed weight stream consumed by plan_cost_kernel (plan_kernels.hip).
// All offsets in floats; every block starts 16-byte aligned.
#pragma once

namespace rrl_plan {

constexpr int kHQ = 256, kQTiles = kHQ / 16;          // Q_risk hidden width (hidden_size default, arg_utils)
constexpr int kHE = 200, kHEPad = 208, kETiles = 13;  // PETS ensemble hidden width (config/navigation1.py:29-41)
constexpr int kEBlocks32 = 7;                         // 32-wide k blocks of the f16x3 kernel (200 -> 224)
constexpr int kESlots = 2 * kEBlocks32;               // float4 slots per column tile: 13 chunks (f32) or 7 x {hi, lo}

// one Q_risk head
constexpr int kQW1 = 0;                                // [16 ct][64 lanes]
constexpr int kQB1 = kQW1 + kQTiles * 64;
constexpr int kQW2 = kQB1 + kHQ;                       // [16 ct][16 j][64 lanes][4]
constexpr int kQB2 = kQW2 + kQTiles * kQTiles * 256;
constexpr int kQW3 = kQB2 + kHQ;
constexpr int kQB3 = kQW3 + kHQ;
constexpr int kQSize = kQB3 + 4;

// one ensemble member
constexpr int kEW0 = 0;                                // [13 ct][64 lanes]
constexpr int kEB0 = kEW0 + kETiles * 64;
constexpr int kEW1 = kEB0 + kHEPad;                    // [13 ct][13 j][64 lanes][4]  (f16x3: [13 ct][7 blocks][hi, lo])
constexpr int kEB1 = kEW1 + kETiles * kESlots * 256;
constexpr int kEW2 = kEB1 + kHEPad;
constexpr int kEB2 = kEW2 + kETiles * kESlots * 256;
constexpr int kEW3 = kEB2 + kHEPad;                    // [208 col][4 out]
constexpr int kEB3 = kEW3 + kHEPad * 4;
constexpr int kESize = kEB3 + 4;

static_assert(kQW2 % 4 == 0 && kQSize % 4 == 0 && kEW1 % 4 == 0 && kEW2 % 4 == 0 && kEW3 % 4 == 0 &&
              kESize % 4 == 0, "fragment blocks must stay float4-aligned");

__host__ __device__ constexpr long long q_off(int h) { return (long long)h * kQSize; }
__host__ __device__ constexpr long long e_off(int e) { return 2LL * kQSize + (long long)e * kESize; }
__host__ __device__ constexpr long long glob_off(int n_nets) { return e_off(n_nets); }   // mu, sigma, max/min logvar
__host__ __device__ constexpr long long packed_floats(int n_nets) { return glob_off(n_nets) + 16; }

}  // namespace rrl_plan
