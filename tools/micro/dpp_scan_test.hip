#include <hip/hip_runtime.h>
#include <stdio.h>
#define LSE_ROW_SCAN16(v, mul, SHR)                                                                               \
    asm volatile("s_nop 1\n\t"                                                                                    \
                 "v_fmac_f32_dpp %0, %0, %16 " SHR "\n\tv_fmac_f32_dpp %1, %1, %16 " SHR "\n\t"                    \
                 "v_fmac_f32_dpp %2, %2, %16 " SHR "\n\tv_fmac_f32_dpp %3, %3, %16 " SHR "\n\t"                    \
                 "v_fmac_f32_dpp %4, %4, %16 " SHR "\n\tv_fmac_f32_dpp %5, %5, %16 " SHR "\n\t"                    \
                 "v_fmac_f32_dpp %6, %6, %16 " SHR "\n\tv_fmac_f32_dpp %7, %7, %16 " SHR "\n\t"                    \
                 "v_fmac_f32_dpp %8, %8, %16 " SHR "\n\tv_fmac_f32_dpp %9, %9, %16 " SHR "\n\t"                    \
                 "v_fmac_f32_dpp %10, %10, %16 " SHR "\n\tv_fmac_f32_dpp %11, %11, %16 " SHR "\n\t"                \
                 "v_fmac_f32_dpp %12, %12, %16 " SHR "\n\tv_fmac_f32_dpp %13, %13, %16 " SHR "\n\t"                \
                 "v_fmac_f32_dpp %14, %14, %16 " SHR "\n\tv_fmac_f32_dpp %15, %15, %16 " SHR                       \
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),   \
                   "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]) \
                 : "v"(mul))
template <int OFF> __device__ int row_shr_int(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x110 + OFF, 0xF, 0xF, true); }
template <int OFF> __device__ bool step(float (&v)[16], int row, int j)
{
    const int row_o = row_shr_int<OFF>(row);     // unconditionally: a DPP read of a lane that is masked out returns 0
    const bool take = (j >= OFF) & (row_o == row);
    if (__builtin_amdgcn_ballot_w64(take) == 0) return false;
    const float tf = take ? 1.f : 0.f;
    if (OFF == 1) LSE_ROW_SCAN16(v, tf, "row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1");
    if (OFF == 2) LSE_ROW_SCAN16(v, tf, "row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1");
    if (OFF == 4) LSE_ROW_SCAN16(v, tf, "row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1");
    if (OFF == 8) LSE_ROW_SCAN16(v, tf, "row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1");
    return true;
}
__global__ void k(const int *rows, const float *in, float *out, int *rown)
{
    const int lane = threadIdx.x, j = lane & 15;
    const int row = rows[j];
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = in[j] * (i + 1);
    if (step<1>(v, row, j) && step<2>(v, row, j) && step<4>(v, row, j)) step<8>(v, row, j);
    out[lane] = v[0];
    rown[lane] = __builtin_amdgcn_update_dpp(-1, row, 0x101, 0xF, 0xF, true);
}
int main()
{
    int h_rows[16] = {3, 10, 10, 25, 30, 30, 30, 30, 30, 30, 30, 30, 30, 30, 30, 30};
    float h_in[16] = {1, 2, 3, 4, 5, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int *rows, *rown; float *in, *out;
    hipMalloc(&rows, 64); hipMalloc(&in, 64); hipMalloc(&out, 256); hipMalloc(&rown, 256);
    hipMemcpy(rows, h_rows, 64, hipMemcpyHostToDevice); hipMemcpy(in, h_in, 64, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, 1, 64, 0, 0, rows, in, out, rown);
    float h_out[64]; int h_rn[64];
    hipMemcpy(h_out, out, 256, hipMemcpyDeviceToHost); hipMemcpy(h_rn, rown, 256, hipMemcpyDeviceToHost);
    for (int i = 0; i < 16; ++i) printf("j=%2d row=%2d v=%5.1f next_row=%d\n", i, h_rows[i], h_out[i], h_rn[i]);
    return 0;
}
