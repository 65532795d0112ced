// Microbenchmark: do INTEGER global atomics run at a different rate than float atomics on gfx950?  (float adds execute at the
// memory side at ~20 G requests/s; if u32 / u64 adds or max were served from the XCD's L2, a fixed-point accumulation of the
// hash-table gradient could be considered.)  Random 8-byte-aligned addresses in a 4 MB and a 64 MB region.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int OP>
__global__ void k(uint32_t *buf, uint32_t mask, int iters)
{
    uint32_t s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    for (int i = 0; i < iters; ++i) {
        s = s * 1664525u + 1013904223u;
        uint32_t idx = ((s >> 8) & mask) & ~1u;
        if (OP == 0) atomicAdd((float *)buf + idx, 1.0f);
        else if (OP == 1) atomicAdd(buf + idx, 1u);
        else if (OP == 2) atomicAdd((unsigned long long *)(buf + idx), 1ull);
        else if (OP == 3) atomicMax(buf + idx, s);
        else if (OP == 4) buf[idx] = s;                                   // plain store, for scale
        else if (OP == 5) __hip_atomic_fetch_add(buf + idx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}
int main()
{
    const char *names[] = {"f32 add", "u32 add", "u64 add", "u32 max", "plain u32 store", "u32 add (workgroup scope)"};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mb : {4, 64}) {
        const size_t n = (size_t)mb << 18;      // u32 count
        uint32_t *buf; hipMalloc(&buf, n * 4); hipMemset(buf, 0, n * 4);
        const int blocks = 2048, iters = 1024;
        for (int op = 0; op < 6; ++op) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                switch (op) {
                case 0: hipLaunchKernelGGL(k<0>, blocks, 256, 0, 0, buf, (uint32_t)(n - 1), iters); break;
                case 1: hipLaunchKernelGGL(k<1>, blocks, 256, 0, 0, buf, (uint32_t)(n - 1), iters); break;
                case 2: hipLaunchKernelGGL(k<2>, blocks, 256, 0, 0, buf, (uint32_t)(n - 1), iters); break;
                case 3: hipLaunchKernelGGL(k<3>, blocks, 256, 0, 0, buf, (uint32_t)(n - 1), iters); break;
                case 4: hipLaunchKernelGGL(k<4>, blocks, 256, 0, 0, buf, (uint32_t)(n - 1), iters); break;
                case 5: hipLaunchKernelGGL(k<5>, blocks, 256, 0, 0, buf, (uint32_t)(n - 1), iters); break;
                }
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            }
            printf("region %3d MB  %-28s %8.3f ms  %6.1f G lane-requests/s\n", mb, names[op], ms,
                   (double)blocks * 256 * iters / (ms * 1e-3) / 1e9);
        }
        hipFree(buf);
    }
    return 0;
}
