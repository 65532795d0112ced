// Microbenchmark: float atomic add rate (random 8-byte pairs in a 64 MB region) by allocation type.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k(float *buf, uint32_t mask, int iters)
{
    uint32_t s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    for (int i = 0; i < iters; ++i) {
        s = s * 1664525u + 1013904223u;
        uint32_t idx = ((s >> 8) & mask) & ~1u;
        atomicAdd(buf + idx, 1.0f);
        atomicAdd(buf + idx + 1, 1.0f);
    }
}
int main()
{
    const size_t n = (size_t)1 << 24;   // 16M floats = 64 MB
    const char *names[] = {"hipMalloc (coarse-grained)", "hipExtMallocWithFlags(FineGrained)", "hipExtMallocWithFlags(Uncached)", "hipMallocManaged"};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int kind = 0; kind < 4; ++kind) {
        float *buf = nullptr; hipError_t rc;
        if (kind == 0) rc = hipMalloc(&buf, n * 4);
        else if (kind == 1) rc = hipExtMallocWithFlags((void **)&buf, n * 4, hipDeviceMallocFinegrained);
        else if (kind == 2) rc = hipExtMallocWithFlags((void **)&buf, n * 4, hipDeviceMallocUncached);
        else rc = hipMallocManaged(&buf, n * 4);
        if (rc != hipSuccess) { printf("%-40s allocation failed: %s\n", names[kind], hipGetErrorString(rc)); (void)hipGetLastError(); continue; }
        hipMemset(buf, 0, n * 4);
        const int blocks = 2048, threads = 256, iters = 128;
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, blocks, threads, 0, 0, buf, (uint32_t)(n - 1), iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        printf("%-40s %7.3f ms  %6.1f G pair-requests/s\n", names[kind], ms, (double)blocks * threads * iters / ms / 1e6);
        hipFree(buf);
    }
    return 0;
}
