// What a scattered small store costs on MI355X: one store of W bytes (aligned to W) every STRIDE bytes over a 1 GiB buffer
// that a dense kernel has just written (so no line is resident), for W = 8 .. 128.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/sparse_write_probe profiles/sparse_write_probe.hip && /tmp/sparse_write_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void dense_fill(float4* p, size_t n16) {
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n16; i += size_t(gridDim.x) * blockDim.x)
        p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

template <int W>
__global__ void sparse_store(char* base, size_t count, size_t stride, uint32_t salt) {
    const size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
    if (i >= count) return;
    // a pseudo-random granule inside the i-th stride window, aligned to W
    const uint32_t h = (uint32_t(i) * 2654435761u + salt) >> 8;
    char* p = base + i * stride + size_t(h % (stride / W)) * W;
    if constexpr (W == 8) *reinterpret_cast<float2*>(p) = make_float2(5.f, 6.f);
    if constexpr (W == 16) *reinterpret_cast<float4*>(p) = make_float4(5.f, 6.f, 7.f, 8.f);
    if constexpr (W >= 32) {
#pragma unroll
        for (int k = 0; k < W / 16; ++k) reinterpret_cast<float4*>(p)[k] = make_float4(5.f, 6.f, 7.f, 8.f);
    }
}

template <int W>
float run(char* buf, size_t bytes, size_t stride) {
    const size_t count = bytes / stride;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        dense_fill<<<2048, 256>>>((float4*)buf, bytes / 16);
        hipEventRecord(e0);
        sparse_store<W><<<(count + 255) / 256, 256>>>(buf, count, stride, 17u * rep);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best * 1e3f;
}

int main() {
    const size_t bytes = size_t(1) << 30;
    char* buf;
    hipMalloc(&buf, bytes);
    for (size_t stride : {size_t(1600), size_t(3200), size_t(6400)}) {
        const size_t count = bytes / stride;
        printf("stride %zu B (%zu stores over 1 GiB):", stride, count);
        printf("  8B %.1f us", run<8>(buf, bytes, stride));
        printf("  16B %.1f us", run<16>(buf, bytes, stride));
        printf("  32B %.1f us", run<32>(buf, bytes, stride));
        printf("  64B %.1f us", run<64>(buf, bytes, stride));
        printf("  128B %.1f us\n", run<128>(buf, bytes, stride));
    }
    // the same number of stores, dense (what they would cost as part of a stream)
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    dense_fill<<<2048, 256>>>((float4*)buf, bytes / 16);
    hipEventRecord(e0);
    dense_fill<<<2048, 256>>>((float4*)buf, bytes / 16);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("dense 1 GiB fill: %.1f us (%.2f TB/s)\n", ms * 1e3f, double(bytes) / (ms * 1e-3) / 1e12);
    return 0;
}
