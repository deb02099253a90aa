// Does cycling through many distinct kernels (cold instruction cache) explain why 2-us kernels take 4.5 us inside
// the lock-step iteration?  16 distinct kernels with ~2-4 KB of straight-line code each: replay ONE kernel 320
// times vs the 16 in rotation, both as dependent launches in a hipGraph.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int ID>
__global__ __launch_bounds__(256) void k(const float* src, float* dst, int n_wg) {
    const int wg = blockIdx.x, nb = (wg + 1) % n_wg;
    float v = src[nb * 256 + threadIdx.x];
    float a = v * 0.5f + float(ID), b = v - 1.f, c = v + 2.f, d = v * 3.f;
#pragma unroll
    for (int i = 0; i < 512; ++i) {     // 2048 FMAs = 16 KB of code per kernel, unique constants
        a = fmaf(a, 1.0001f + ID * 1e-6f, 0.001f * i);
        b = fmaf(b, 0.9999f, a * 1e-9f);
        c = fmaf(c, 1.0002f, 0.002f);
        d = fmaf(d, 0.9998f, c * 1e-9f);
    }
    dst[wg * 256 + threadIdx.x] = v + 1.f + (a + b + c + d) * 1e-30f;
}

typedef void (*kern_t)(const float*, float*, int);

template <int... I>
void fill(kern_t* t, std::integer_sequence<int, I...>) {
    ((t[I] = k<I>), ...);
}

int main() {
    const int n_wg = 16, launches = 320;
    kern_t table[16];
    fill(table, std::make_integer_sequence<int, 16>{});
    float* buf;
    hipMalloc(&buf, sizeof(float) * 2 * n_wg * 256);
    hipMemset(buf, 0, sizeof(float) * 2 * n_wg * 256);
    hipStream_t st;
    hipStreamCreate(&st);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int distinct = 1; distinct <= 16; distinct *= 4) {
        hipGraph_t g;
        hipGraphExec_t ge;
        hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
        for (int r = 0; r < launches; ++r)
            hipLaunchKernelGGL(table[r % distinct], dim3(n_wg), dim3(256), 0, st, buf + (size_t)(r & 1) * n_wg * 256,
                               buf + (size_t)((r + 1) & 1) * n_wg * 256, n_wg);
        hipStreamEndCapture(st, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, st);
        hipStreamSynchronize(st);
        float ms;
        hipEventRecord(e0, st);
        hipGraphLaunch(ge, st);
        hipEventRecord(e1, st);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("%2d distinct kernels in rotation: %.3f us per launch\n", distinct, ms * 1e3 / launches);
    }
    return 0;
}
