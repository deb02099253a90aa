// Sustained rate of the f16 MFMA shapes a split-f16 (hi + lo, three products) planner would use, next to the f32 MFMA the
// planner uses now: 8 independent accumulators per wave, 4 waves per SIMD, no memory traffic.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_f16_rate profiles/mfma_f16_rate.hip && /tmp/mfma_f16_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int SHAPE>
__global__ __launch_bounds__(512) void spin(float* out, int iters) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-4f + 1.f;
    const h4 a4 = {(_Float16)a, (_Float16)b, (_Float16)a, (_Float16)b}, b4 = {(_Float16)b, (_Float16)a, (_Float16)b, (_Float16)a};
    const h8 a8 = {a4[0], a4[1], a4[2], a4[3], a4[0], a4[1], a4[2], a4[3]};
    const h8 b8 = {b4[0], b4[1], b4[2], b4[3], b4[0], b4[1], b4[2], b4[3]};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if constexpr (SHAPE == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
                if constexpr (SHAPE == 1) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[i], 0, 0, 0);
                if constexpr (SHAPE == 2) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[i], 0, 0, 0);
            }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456f) out[0] = s;
}

template <int SHAPE>
void run(const char* name, double flop_per_mfma, float* out, int iters) {
    const int grid = 256 * 2;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(spin<SHAPE>, dim3(grid), dim3(512), 0, 0, out, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double n_mfma = double(grid) * 8 * double(iters) * 32;
    // cycles per MFMA per SIMD at 4 waves per SIMD: time * clock / (MFMAs per SIMD)
    printf("%-28s %.2f ms  %.1f TFLOP/s  (%.1f ns per MFMA per SIMD)\n", name, best, n_mfma * flop_per_mfma / best / 1e9,
           best * 1e6 / (n_mfma / 1024.0));
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    float* out;
    (void)hipMalloc(&out, 4);
    run<0>("v_mfma_f32_16x16x4_f32", 2048.0, out, iters);
    run<1>("v_mfma_f32_16x16x16_f16", 8192.0, out, iters);
    run<2>("v_mfma_f32_16x16x32_f16", 16384.0, out, iters);
    return 0;
}
