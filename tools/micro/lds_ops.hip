// Microbenchmark: LDS operation rates on gfx950 with 64 distinct addresses per wave instruction (random slots in a
// 16 KB per-wave region): plain read / write / read-modify-write vs ds_add_f32 / ds_add_u32 / ds_cmpst_rtn_b32.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) float lds_f32;
typedef __attribute__((address_space(3))) uint64_t lds_u64;
template <int OP>
__global__ __launch_bounds__(256) void k(float *out, int iters, int active)
{
    __shared__ float buf[4][4096];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = lane; i < 4096; i += 64) buf[wave][i] = 0.f;
    lds_f32 *b = (lds_f32 *)&buf[wave][0];
    lds_u32 *bu = (lds_u32 *)&buf[wave][0];
    uint32_t s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.f;
    if (lane < active)
        for (int i = 0; i < iters; ++i) {
            s = s * 1664525u + 1013904223u;
            const uint32_t a = (s >> 10) & 4095;
            if (OP == 0) acc += b[a];
            else if (OP == 1) b[a] = acc + i;
            else if (OP == 2) { float t = b[a]; b[a] = t + 1.f; }
            else if (OP == 3) __hip_atomic_fetch_add(&b[a], 1.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            else if (OP == 4) __hip_atomic_fetch_add(&bu[a], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            else if (OP == 5) { uint32_t e = 0; __hip_atomic_compare_exchange_strong(&bu[a], &e, s, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); acc += e; }
            else if (OP == 6) acc += __hip_atomic_fetch_add(&b[a], 1.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            else if (OP == 7) { uint64_t e = 0; __hip_atomic_compare_exchange_strong((lds_u64 *)&bu[a & ~1u], &e, (uint64_t)s, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); acc += (float)(uint32_t)e; }
            else if (OP == 8) { acc += (float)(uint32_t)*(volatile lds_u64 *)&bu[a & ~1u]; }
            else if (OP == 9) { *(volatile lds_u64 *)&bu[a & ~1u] = (uint64_t)s; }
            else if (OP == 10) {   // float add by 32-bit CAS loop
                uint32_t cur = *(volatile lds_u32 *)&bu[a], prev;
                while (true) { prev = cur; __hip_atomic_compare_exchange_strong(&bu[a], &prev, __float_as_uint(__uint_as_float(cur) + 1.f), __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); if (prev == cur) break; cur = prev; }
            }
            else if (OP == 11) __hip_atomic_fetch_add(&bu[a], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (OP == 12) __hip_atomic_fetch_max(&bu[a], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = acc + buf[wave][lane];
}
int main()
{
    float *out; hipMalloc(&out, 4096 * 256 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char *names[] = {"ds_read_b32", "ds_write_b32", "read+write (plain RMW)", "ds_add_f32", "ds_add_u32", "ds_cmpst_rtn_b32", "ds_add_rtn_f32", "ds_cmpst_rtn_b64", "ds_read_b64", "ds_write_b64", "f32 add via b32 CAS loop", "ds_add_u32 (wg scope)", "ds_max_u32"};
    const int blocks = 2048, iters = 2048;   // 8 workgroups per CU
    for (int active = 64; active >= 8; active /= 4)
        for (int op = 0; op < 13; ++op) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                switch (op) {
                case 0: hipLaunchKernelGGL(k<0>, blocks, 256, 0, 0, out, iters, active); break;
                case 1: hipLaunchKernelGGL(k<1>, blocks, 256, 0, 0, out, iters, active); break;
                case 2: hipLaunchKernelGGL(k<2>, blocks, 256, 0, 0, out, iters, active); break;
                case 3: hipLaunchKernelGGL(k<3>, blocks, 256, 0, 0, out, iters, active); break;
                case 4: hipLaunchKernelGGL(k<4>, blocks, 256, 0, 0, out, iters, active); break;
                case 5: hipLaunchKernelGGL(k<5>, blocks, 256, 0, 0, out, iters, active); break;
                case 6: hipLaunchKernelGGL(k<6>, blocks, 256, 0, 0, out, iters, active); break;
                case 7: hipLaunchKernelGGL(k<7>, blocks, 256, 0, 0, out, iters, active); break;
                case 8: hipLaunchKernelGGL(k<8>, blocks, 256, 0, 0, out, iters, active); break;
                case 9: hipLaunchKernelGGL(k<9>, blocks, 256, 0, 0, out, iters, active); break;
                case 10: hipLaunchKernelGGL(k<10>, blocks, 256, 0, 0, out, iters, active); break;
                case 11: hipLaunchKernelGGL(k<11>, blocks, 256, 0, 0, out, iters, active); break;
                case 12: hipLaunchKernelGGL(k<12>, blocks, 256, 0, 0, out, iters, active); break;
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            // wave-instructions per CU: blocks*4*iters / 256 CUs ; cycles at 2.4 GHz
            const double instr_per_cu = (double)blocks * 4 * iters / 256.0;
            printf("active %2d lanes  %-24s %8.3f ms  -> %7.1f cycles per wave-instruction per CU (2.4 GHz)\n", active, names[op], ms,
                   ms * 1e-3 * 2.4e9 / instr_per_cu);
        }
    return 0;
}
