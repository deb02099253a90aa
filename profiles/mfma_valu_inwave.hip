// Does a wave's OWN VALU / LDS work hide in the gaps of its v_mfma_f32_16x16x4_f32 stream?  (profiles/mfma_valu_overlap.hip: a
// PARTNER wave's VALU work does not run under it.)  One wave per SIMD (256 workgroups of 4 waves); every step issues one MFMA
// (four accumulators in rotation) followed by K filler instructions, in program order (asm volatile keeps the order):
//   mode 0: K x v_fma_f32 on independent registers        mode 1: K x ds_read_b32        mode 2: K x (v_add, v_max, ds_write_b32)/3
// Time per step against K: flat up to K0 => K0 fillers per gap are free.
// hipcc --offload-arch=gfx950 -O3 -o profiles/_ab_mfma_valu_inwave profiles/mfma_valu_inwave.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int K, bool MFMA>
__global__ __launch_bounds__(256) void solo(float* out, int n) {
    __shared__ float lds[4096];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-4f + 1.f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
    const float m = 1.0001f, c = 0.001f;
    const unsigned addr = (threadIdx.x & 63) * 4;
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (MFMA) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[s & 3]) : "v"(a), "v"(b));
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int r = (s * K + k) & 7;
                if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[r]) : "v"(m), "v"(c));
                else if (MODE == 1) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v[r]) : "v"(addr), "n"(256 * ((s * K + k) & 7)));
                else {
                    const int ph = (s * K + k) % 3;
                    if (ph == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[r]) : "v"(c));
                    else if (ph == 1) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[r]) : "v"(c));
                    else asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(addr), "v"(v[r]), "n"(256 * ((s * K + k) & 7)));
                }
            }
        }
        if (MODE != 0) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    asm volatile("s_nop 15\n s_nop 15");
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 123.456f) out[0] = s;
}

template <int MODE, int K, bool MFMA>
static float run(float* out, int n) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((solo<MODE, K, MFMA>), dim3(256), dim3(256), 0, 0, out, n);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    return best * 1000.f;
}

template <int MODE, int K>
static void line(float* out, int n, const char* what) {
    const float both = run<MODE, K, true>(out, n), alone = K ? run<MODE, K, false>(out, n) : 0.f;
    printf("%-34s K = %d  per MFMA step: with the MFMA %6.1f cycles, fillers alone %6.1f cycles\n", what, K, both * 2400.f / (n * 8.f),
           alone * 2400.f / (n * 8.f));
}

int main() {
    float* out;
    hipMalloc(&out, 4);
    const int n = 20000;
    line<0, 0>(out, n, "v_fma_f32");
    line<0, 1>(out, n, "v_fma_f32"); line<0, 2>(out, n, "v_fma_f32"); line<0, 3>(out, n, "v_fma_f32"); line<0, 4>(out, n, "v_fma_f32");
    line<0, 6>(out, n, "v_fma_f32"); line<0, 8>(out, n, "v_fma_f32"); line<0, 12>(out, n, "v_fma_f32");
    line<1, 1>(out, n, "ds_read_b32"); line<1, 2>(out, n, "ds_read_b32"); line<1, 4>(out, n, "ds_read_b32"); line<1, 8>(out, n, "ds_read_b32");
    line<2, 3>(out, n, "v_add / v_max / ds_write_b32"); line<2, 6>(out, n, "v_add / v_max / ds_write_b32");
    line<2, 9>(out, n, "v_add / v_max / ds_write_b32"); line<2, 12>(out, n, "v_add / v_max / ds_write_b32");
    return 0;
}
