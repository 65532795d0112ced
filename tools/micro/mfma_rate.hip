// Microbenchmark: cycles per MFMA on one SIMD (one wave per SIMD, back-to-back on 4 independent accumulators and on 1):
// v_mfma_f32_16x16x32_bf16 (gfx950 form) against the legacy v_mfma_f32_16x16x16_bf16 (k = 16, two registers per operand).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
template <int OP, int NACC>
__global__ __launch_bounds__(256) void k(float *out, int iters, long long *cyc)
{
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float v = 1.0f + threadIdx.x * 1e-3f;
    bf16x8 a8, b8;
    for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)v; b8[i] = (__bf16)(v * 0.5f); }
    s16x4 a4 = {(short)0x3f80, (short)0x3f80, (short)0x3f00, (short)0x3f00}, b4 = a4;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (OP == 0) acc[u % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[u % NACC], 0, 0, 0);
            else acc[u % NACC] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[u % NACC], 0, 0, 0);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main()
{
    float *out; long long *cyc, h;
    hipMalloc(&out, 256 * 256 * sizeof(float)); hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4096;
    for (int cfg = 0; cfg < 4; ++cfg) {
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            switch (cfg) {
            case 0: hipLaunchKernelGGL((k<0, 4>), 256, 256, 0, 0, out, iters, cyc); break;
            case 1: hipLaunchKernelGGL((k<0, 1>), 256, 256, 0, 0, out, iters, cyc); break;
            case 2: hipLaunchKernelGGL((k<1, 4>), 256, 256, 0, 0, out, iters, cyc); break;
            case 3: hipLaunchKernelGGL((k<1, 1>), 256, 256, 0, 0, out, iters, cyc); break;
            }
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        }
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        const double n = (double)iters * 16;
        printf("%-28s %d accumulators: %.3f ms, %.1f ns per MFMA per SIMD, s_memtime ticks per MFMA %.2f\n",
               cfg < 2 ? "v_mfma_f32_16x16x32_bf16" : "v_mfma_f32_16x16x16_bf16", (cfg & 1) ? 1 : 4, ms, ms * 1e6 / n, (double)h / n);
    }
    return 0;
}
