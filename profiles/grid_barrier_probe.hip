// How cheap is a software grid barrier + producer/consumer hand-off compared with a kernel boundary?
// Workgroups are pinned to ONE XCD (dispatch is round-robin over the 8 XCDs, so blockIdx % 8 == 0 lands on XCD 0
// and the others exit at once): they share one L2, which is then the coherence point.
// Each round: every workgroup reads the value its neighbour wrote in the previous round, adds 1, writes it,
// then all meet at a barrier (atomic counter + spin).  Prints us per round for both scopes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <bool ONE_XCD>
__global__ __launch_bounds__(256) void rounds_kernel(float* buf, unsigned* bar, int n_wg, int rounds) {
    int wg = blockIdx.x;
    if (ONE_XCD) {
        if (wg % 8 != 0) return;
        wg /= 8;
    }
    if (wg >= n_wg) return;
    const int tid = threadIdx.x;
    for (int r = 0; r < rounds; ++r) {
        const float* src = buf + (size_t)(r & 1) * n_wg * 256;
        float* dst = buf + (size_t)((r + 1) & 1) * n_wg * 256;
        const int nb = (wg + 1) % n_wg;
        const float v = __builtin_nontemporal_load(src + nb * 256 + tid);      // bypass stale L1 lines
        __builtin_nontemporal_store(v + 1.f, dst + wg * 256 + tid);
        __threadfence();                      // release: my writes are visible device-wide
        __syncthreads();
        if (tid == 0) {
            const unsigned target = (unsigned)n_wg * (r + 1);
            atomicAdd(bar, 1u);
            while (__hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {}
        }
        __syncthreads();
    }
}

// Round 3: the same rounds with the XCD's L2 as the ONLY coherence point: workgroups of one XCD, data written with plain
// (write-through) stores and read with L1-bypassing loads, release fence at WORKGROUP scope (no L2 write-back), barrier
// counter bumped and polled with WORKGROUP-scope atomics (executed in the XCD's L2).  Outside the HIP memory model
// (workgroup scope does not cover other workgroups) -- it works because the participants share one L2; this is the
// cheapest barrier the hardware can give a persistent single-XCD kernel.
__global__ __launch_bounds__(256) void rounds_xcd_local_kernel(float* buf, unsigned* bar, int n_wg, int rounds,
                                                               unsigned* xcc_seen) {
    int wg = blockIdx.x;
    if (wg % 8 != 0) return;
    wg /= 8;
    if (wg >= n_wg) return;
    const int tid = threadIdx.x;
    if (tid == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc_seen[wg] = xcc & 0xf;
    }
    for (int r = 0; r < rounds; ++r) {
        const float* src = buf + (size_t)(r & 1) * n_wg * 256;
        float* dst = buf + (size_t)((r + 1) & 1) * n_wg * 256;
        const int nb = (wg + 1) % n_wg;
        const float v = __hip_atomic_load(src + nb * 256 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // L2, not L1
        dst[wg * 256 + tid] = v + 1.f;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        if (tid == 0) {
            const unsigned target = (unsigned)n_wg * (r + 1);
            __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            // polled with an L1-bypassing load (agent scope = served by the XCD's L2); a workgroup-scope `fetch_add(bar, 0)`
            // is folded into a plain workgroup-scope load by the compiler, which the CU's L1 answers forever (it hung)
            while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {}
        }
        __syncthreads();
    }
}

__global__ void step_kernel(const float* src, float* dst, int n_wg) {
    const int wg = blockIdx.x, nb = (wg + 1) % n_wg;
    dst[wg * 256 + threadIdx.x] = src[nb * 256 + threadIdx.x] + 1.f;
}

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int n_wg = argc > 1 ? atoi(argv[1]) : 32, rounds = 2000;
    float* buf;
    unsigned* bar;
    hipMalloc(&buf, sizeof(float) * 2 * n_wg * 256);
    hipMalloc(&bar, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms;
    for (int one = 1; one >= 0; --one) {
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(buf, 0, sizeof(float) * 2 * n_wg * 256);
            hipMemset(bar, 0, 4);
            hipEventRecord(e0);
            if (one) hipLaunchKernelGGL(rounds_kernel<true>, dim3(n_wg * 8), dim3(256), 0, 0, buf, bar, n_wg, rounds);
            else hipLaunchKernelGGL(rounds_kernel<false>, dim3(n_wg), dim3(256), 0, 0, buf, bar, n_wg, rounds);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        float h;
        hipMemcpy(&h, buf, 4, hipMemcpyDeviceToHost);
        printf("%s: %d workgroups, %.3f us per round (check %.0f == %d)\n", one ? "one XCD " : "all XCDs", n_wg,
               ms * 1e3 / rounds, h, rounds);
    }
    {
        unsigned* seen;
        hipMalloc(&seen, 4 * n_wg);
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(buf, 0, sizeof(float) * 2 * n_wg * 256);
            hipMemset(bar, 0, 4);
            hipEventRecord(e0);
            hipLaunchKernelGGL(rounds_xcd_local_kernel, dim3(n_wg * 8), dim3(256), 0, 0, buf, bar, n_wg, rounds, seen);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        float h;
        hipMemcpy(&h, buf, 4, hipMemcpyDeviceToHost);
        unsigned hs[64];
        hipMemcpy(hs, seen, 4 * (n_wg < 64 ? n_wg : 64), hipMemcpyDeviceToHost);
        unsigned lo = 99, hi = 0;
        for (int i = 0; i < n_wg && i < 64; ++i) { lo = hs[i] < lo ? hs[i] : lo; hi = hs[i] > hi ? hs[i] : hi; }
        printf("one XCD, L2-local (workgroup-scope atomics, no L2 write-back): %d workgroups, %.3f us per round "
               "(check %.0f == %d; XCC_ID of the participants %u..%u)\n", n_wg, ms * 1e3 / rounds, h, rounds, lo, hi);
    }
    // kernel boundaries: the same dependent rounds as separate launches captured in a graph
    hipStream_t st;
    hipStreamCreate(&st);
    hipGraph_t g;
    hipGraphExec_t ge;
    const int k = 200;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int r = 0; r < k; ++r)
        hipLaunchKernelGGL(step_kernel, dim3(n_wg), dim3(256), 0, st, buf + (size_t)(r & 1) * n_wg * 256,
                           buf + (size_t)((r + 1) & 1) * n_wg * 256, n_wg);
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, st);
    hipStreamSynchronize(st);
    hipEventRecord(e0, st);
    hipGraphLaunch(ge, st);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("kernel boundary (hipGraph): %.3f us per round\n", ms * 1e3 / k);
    return 0;
}
