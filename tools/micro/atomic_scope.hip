// Microbenchmark: float atomic add rate by memory scope (agent = memory-side; workgroup/wavefront = XCD L2?)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
template <int SCOPE>
__global__ void k(float *buf, uint32_t mask, int iters, int per_xcd_shift)
{
    uint32_t xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7;
    float *base = buf + ((size_t)xcc << per_xcd_shift);   // private region per XCD (per_xcd_shift=0 -> shared)
    if (per_xcd_shift == 0) base = buf;
    uint32_t s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    for (int i = 0; i < iters; ++i) {
        s = s * 1664525u + 1013904223u;
        uint32_t idx = (s >> 8) & mask;
        idx &= ~1u;
        if (SCOPE == 0) {
            __hip_atomic_fetch_add(base + idx, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(base + idx + 1, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (SCOPE == 1) {
            __hip_atomic_fetch_add(base + idx, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(base + idx + 1, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            __hip_atomic_fetch_add(base + idx, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __hip_atomic_fetch_add(base + idx + 1, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
    }
}
int main()
{
    const size_t total = (size_t)8 << 26;   // 8 x 64M floats = 2 GiB
    float *buf; hipMalloc(&buf, total * sizeof(float)); hipMemset(buf, 0, total * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 2048, threads = 256, iters = 256;
    for (int region_log2 = 20; region_log2 <= 26; region_log2 += 2) {           // floats: 4 MB, 16 MB, 64 MB, 256 MB
        for (int priv = 0; priv < 2; ++priv)
            for (int scope = 0; scope < 3; ++scope) {
                uint32_t mask = (1u << region_log2) - 1;
                int shift = priv ? 26 : 0;
                for (int rep = 0; rep < 2; ++rep) {
                    hipEventRecord(e0);
                    if (scope == 0) hipLaunchKernelGGL(k<0>, blocks, threads, 0, 0, buf, mask, iters, shift);
                    else if (scope == 1) hipLaunchKernelGGL(k<1>, blocks, threads, 0, 0, buf, mask, iters, shift);
                    else hipLaunchKernelGGL(k<2>, blocks, threads, 0, 0, buf, mask, iters, shift);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                }
                float ms; hipEventElapsedTime(&ms, e0, e1);
                double pairs = (double)blocks * threads * iters;
                printf("region %4d MB  private_per_xcd=%d scope=%s : %7.3f ms  %6.1f G pair-requests/s\n",
                       (int)((4ull << region_log2) >> 20), priv, scope == 0 ? "agent    " : scope == 1 ? "workgroup" : "wavefront", ms,
                       pairs / ms / 1e6);
            }
    }
    // correctness probe: workgroup-scope adds into PRIVATE per-XCD regions must sum to the number of adds
    hipMemset(buf, 0, total * sizeof(float));
    hipLaunchKernelGGL(k<1>, blocks, threads, 0, 0, buf, (1u << 20) - 1, iters, 26);
    hipDeviceSynchronize();
    float *h = (float *)malloc((size_t)8 << 22);
    double sum = 0;
    for (int x = 0; x < 8; ++x) {
        hipMemcpy(h, buf + ((size_t)x << 26), (size_t)4 << 20, hipMemcpyDeviceToHost);
        for (size_t i = 0; i < (1u << 20); ++i) sum += h[i];
    }
    printf("private workgroup-scope: sum %.0f expected %.0f\n", sum, 2.0 * blocks * threads * iters);
    return 0;
}
