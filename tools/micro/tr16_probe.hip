#include <hip/hip_runtime.h>
#include <stdint.h>
typedef short v4i16 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4i16 lds_v4i16;
// image: [rows][cols] of u16, value = row*256+col; each lane supplies an address; result 4 u16 per lane
__global__ void k(const int *lane_off, uint16_t *out, int rows, int pitch_elems) {
  extern __shared__ uint16_t img[];
  for (int e = threadIdx.x; e < rows * pitch_elems; e += 64) img[e] = (uint16_t)((e / pitch_elems) * 256 + (e % pitch_elems));
  __syncthreads();
  lds_v4i16 *p = (lds_v4i16 *)(img + lane_off[threadIdx.x]);
  v4i16 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
  for (int i = 0; i < 4; ++i) out[threadIdx.x * 4 + i] = (uint16_t)v[i];
}
int main() {
  int h_off[64]; uint16_t h_out[256];
  const int pitch = 64;
  // group g (16 lanes): block rows 4g..4g+3, cols 16..31 ; lane 4q+p -> row 4g+q, cols 16+4p
  for (int l = 0; l < 64; ++l) { int g = l >> 4, q = (l & 15) >> 2, p = l & 3; h_off[l] = (4 * g + q) * pitch + 16 + 4 * p; }
  int *d_off; uint16_t *d_out;
  hipMalloc(&d_off, sizeof(h_off)); hipMalloc(&d_out, sizeof(h_out));
  hipMemcpy(d_off, h_off, sizeof(h_off), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 32 * pitch * 2, 0, d_off, d_out, 32, pitch);
  hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int i = 0; i < 4; ++i) printf(" (r%d,c%d)", h_out[l*4+i] >> 8, h_out[l*4+i] & 255); printf("\n"); }
  return 0;
}
