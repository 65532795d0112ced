// Microbenchmark: what does a float-atomic REQUEST cost as a function of how the 64 lanes of one instruction are grouped?
// Every wave-instruction adds 64 floats; the lanes form G groups of 64/G contiguous floats (4*64/G bytes), each group at a
// random aligned address of a 64 MB region.  G = 64: one float per group (scattered lanes) ... G = 1: one 256-byte run.
// If the memory-side units charge per 32-byte sector, per 64-byte or per 128-byte line shows in where the rate stops growing.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int LOGG>
__global__ void k(float *buf, uint32_t n_floats, int iters)
{
    constexpr int G = 1 << LOGG, PER = 64 / G;            // lanes per group
    const int lane = threadIdx.x & 63;
    const int grp = lane / PER, in_grp = lane % PER;
    // one random stream per (wave, group): all lanes of a group compute the same address
    uint32_t s = ((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 64 + grp) * 2654435761u + 12345u;
    const uint32_t units = n_floats / PER;                 // aligned positions of a group
    for (int i = 0; i < iters; ++i) {
        s = s * 1664525u + 1013904223u;
        const uint32_t u = (s >> 4) % units;
        atomicAdd(buf + (size_t)u * PER + in_grp, 1.0f);
    }
}
int main()
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t n = (size_t)64 << 18;      // floats in 64 MB
    float *buf; hipMalloc(&buf, n * 4); hipMemset(buf, 0, n * 4);
    const int blocks = 2048, iters = 512;
    for (int logg = 6; logg >= 0; --logg) {
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            switch (logg) {
            case 6: hipLaunchKernelGGL(k<6>, blocks, 256, 0, 0, buf, (uint32_t)n, iters); break;
            case 5: hipLaunchKernelGGL(k<5>, blocks, 256, 0, 0, buf, (uint32_t)n, iters); break;
            case 4: hipLaunchKernelGGL(k<4>, blocks, 256, 0, 0, buf, (uint32_t)n, iters); break;
            case 3: hipLaunchKernelGGL(k<3>, blocks, 256, 0, 0, buf, (uint32_t)n, iters); break;
            case 2: hipLaunchKernelGGL(k<2>, blocks, 256, 0, 0, buf, (uint32_t)n, iters); break;
            case 1: hipLaunchKernelGGL(k<1>, blocks, 256, 0, 0, buf, (uint32_t)n, iters); break;
            case 0: hipLaunchKernelGGL(k<0>, blocks, 256, 0, 0, buf, (uint32_t)n, iters); break;
            }
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        }
        const double instr = (double)blocks * 4 * iters;
        const int G = 1 << logg;
        printf("groups/instr %2d  bytes/group %4d : %8.3f ms  %7.2f G lane-adds/s  %6.2f G groups/s  %6.1f ns per wave-instruction\n", G, 256 / G, ms,
               instr * 64 / (ms * 1e-3) / 1e9, instr * G / (ms * 1e-3) / 1e9, ms * 1e6 / instr * 2048 * 4 / 1.0 / (2048 * 4));
    }
    return 0;
}
