// Checks the cross-lane helpers used by net_resident_kernel.hip against __shfl_xor, and the fma_mix split.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ float row_sum8(float s) {
  s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0xB1, 0xF, 0xF, true));
  // v_permlane16_swap / v_permlane32_swap exchange the odd rows (upper half) of one register with the even rows (lower
  // half) of ANOTHER: copy, swap, add.  Written as asm: hipcc 7.2 mis-models the builtin's second result.
  float c;
  asm("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(s), "=&v"(c));
  s += c;
  asm("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(s), "=&v"(c));
  return s + c;
}
__global__ void k(float* out) {
  const int l = threadIdx.x;
  float x = (float)(l * l % 37) + 0.25f * l;
  float ref = x;
  ref += __shfl_xor(ref, 1);
  ref += __shfl_xor(ref, 16);
  ref += __shfl_xor(ref, 32);
  out[l] = row_sum8(x) - ref;
  float a = 1.2345678f * (l + 1), b = -0.000123456f * (l + 3);
  f16x2 h = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(a, b));
  float ra, rb;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(ra) : "v"(h), "v"(a));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rb) : "v"(h), "v"(b));
  out[64 + l] = ra - (a - (float)h[0]);
  out[128 + l] = rb - (b - (float)h[1]);
  out[192 + l] = ra;
}
int main() {
  float* d; (void)hipMalloc(&d, 256 * 4);
  k<<<1, 64>>>(d);
  float h[256]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  double e0 = 0, e1 = 0, e2 = 0;
  for (int i = 0; i < 64; ++i) { e0 = fmax(e0, fabs(h[i])); e1 = fmax(e1, fabs(h[64 + i])); e2 = fmax(e2, fabs(h[128 + i])); }
  printf("row_sum8 max diff %g; fma_mix lo diff %g, hi diff %g; sample ra %g\n", e0, e1, e2, h[192 + 5]);
  return 0;
}
