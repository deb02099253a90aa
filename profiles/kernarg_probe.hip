// Kernel-boundary cost vs size of the by-value argument block: two do-nothing kernels, one with a 16-byte and one with a
// 2 KB argument block (both read one word of it), launched on the caller's stream (captured into a graph by the caller).
//   hipcc --offload-arch=gfx950 -O3 -fPIC -shared -o profiles/_ab_kernarg.so profiles/kernarg_probe.hip
#include <hip/hip_runtime.h>
struct Big { unsigned long long w[256]; };          // 2 KB
struct Mid { unsigned long long w[64]; };           // 512 B
__global__ void nop_small(unsigned long long* out, unsigned long long v) { if (v == 12345 && threadIdx.x == 999) out[0] = v; }
__global__ void nop_mid(unsigned long long* out, Mid b) { if (b.w[63] == 12345 && threadIdx.x == 999) out[0] = b.w[1]; }
__global__ void nop_big(unsigned long long* out, Big b) { if (b.w[255] == 12345 && threadIdx.x == 999) out[0] = b.w[1]; }
// the 2 KB block read from device memory instead (one pointer as the argument)
__global__ void nop_ptr(unsigned long long* out, const Big* b) { if (b->w[255] == 12345 && threadIdx.x == 999) out[0] = b->w[1]; }
extern "C" {
int probe_small(void* out, void* stream) { hipLaunchKernelGGL(nop_small, dim3(16), dim3(256), 0, (hipStream_t)stream, (unsigned long long*)out, 1ULL); return 0; }
int probe_mid(void* out, void* stream) { Mid b{}; hipLaunchKernelGGL(nop_mid, dim3(16), dim3(256), 0, (hipStream_t)stream, (unsigned long long*)out, b); return 0; }
int probe_big(void* out, void* stream) { Big b{}; hipLaunchKernelGGL(nop_big, dim3(16), dim3(256), 0, (hipStream_t)stream, (unsigned long long*)out, b); return 0; }
int probe_ptr(void* out, const void* blob, void* stream) { hipLaunchKernelGGL(nop_ptr, dim3(16), dim3(256), 0, (hipStream_t)stream, (unsigned long long*)out, (const Big*)blob); return 0; }
}
