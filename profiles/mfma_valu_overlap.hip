// Do a wave's VALU / LDS instructions run UNDER another wave's v_mfma_f32_16x16x4_f32 stream on the same SIMD?
// One workgroup of 8 waves per CU (two waves per SIMD: wave w and wave w + 4): waves 0-3 issue NM independent f32 MFMAs,
// waves 4-7 issue NV plain f32 VALU ops (v_fma_f32, or v_add+v_max pairs) and / or ND ds_read_b128.  Timed: the matrix
// stream alone, the partner alone, both together.  together ~ max(alone) => they overlap; ~ sum => they serialise.
// hipcc --offload-arch=gfx950 -O3 -o profiles/_ab_mfma_valu_overlap profiles/mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>   // partner: 0 v_fma chain x8 independent, 1 ds_read_b128, 2 v_mfma (second matrix stream), 3 v_exp (transcendental)
__global__ __launch_bounds__(512) void duo(float* out, int nm, int nv) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    const int wave = threadIdx.x >> 6;
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    float s = 0.f;
    if (wave < 4) {
        f32x4 acc[4];
        for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-4f + 1.f;
        for (int it = 0; it < nm; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else if (MODE == 0) {
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
        const float m = 1.0001f, c = 0.001f;
        for (int it = 0; it < nv; ++it) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = fmaf(v[i], m, c);
        }
        for (int i = 0; i < 8; ++i) s += v[i];
    } else if (MODE == 1) {
        f32x4 v[8];
        for (int i = 0; i < 8; ++i) v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        const f32x4* p = reinterpret_cast<const f32x4*>(lds) + (threadIdx.x & 63);
        for (int it = 0; it < nv; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                f32x4 t = p[64 * i];
                asm volatile("" : "+v"(t));
                v[i] = t;
            }
        }
        for (int i = 0; i < 8; ++i) s += v[i][0];
    } else if (MODE == 2) {
        f32x4 acc[4];
        for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-4f + 1.f;
        for (int it = 0; it < nv; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) s += acc[i][0];
    } else {
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
        for (int it = 0; it < nv; ++it) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = __expf(v[i]) * 0.5f;
        }
        for (int i = 0; i < 8; ++i) s += v[i];
    }
    if (s == 123.456f) out[0] = s;
}

template <int MODE>
static float run(float* out, int nm, int nv) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(duo<MODE>, dim3(256), dim3(512), 0, 0, out, nm, nv);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    return best * 1000.f;
}

int main() {
    float* out;
    hipMalloc(&out, 4);
    const int nm = 4000;                       // 4000 x 32 MFMAs = 128 000 x 32 cycles = 4.1 M cycles ~ 1.7 ms
    const float tm = run<0>(out, nm, 0);
    printf("matrix stream alone (128 000 v_mfma_f32_16x16x4_f32 per wave): %.1f us = %.1f cycles per MFMA at 2.4 GHz\n", tm, tm * 2400.f / (nm * 32.f));
    const char* names[4] = {"v_fma_f32 (32 per step)", "ds_read_b128 (8 per step)", "second v_mfma stream (32 per step)", "v_exp_f32 + v_mul (32 per step)"};
    const int nvs[4] = {12000, 12000, 2000, 6000};
    for (int mode = 0; mode < 4; ++mode) {
        const int nv = nvs[mode];
        float ta, tb;
        if (mode == 0) ta = run<0>(out, 0, nv), tb = run<0>(out, nm, nv);
        else if (mode == 1) ta = run<1>(out, 0, nv), tb = run<1>(out, nm, nv);
        else if (mode == 2) ta = run<2>(out, 0, nv), tb = run<2>(out, nm, nv);
        else ta = run<3>(out, 0, nv), tb = run<3>(out, nm, nv);
        printf("partner %-36s alone %8.1f us   together %8.1f us   (max %8.1f, sum %8.1f)\n", names[mode], ta, tb, ta > tm ? ta : tm, ta + tm);
    }
    return 0;
}
