// Does v_mfma_f32_16x16x32_f16 honour f16 subnormal inputs, and how exact is its f32 accumulation of mixed magnitudes?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float a_val, float b_val, float c0, float* out) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.f; b[i] = (_Float16)0.f; }
  // one non-zero k element per lane group 0: A[i][k=0] = a_val, B[k=0][j] = b_val
  if ((threadIdx.x >> 4) == 0) { a[0] = (_Float16)a_val; b[0] = (_Float16)b_val; }
  f32x4 c = {c0, c0, c0, c0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  if (threadIdx.x == 0) { out[0] = c[0]; out[1] = (float)a[0]; out[2] = (float)b[0]; }
}
int main() {
  float* d; (void)hipMalloc(&d, 64);
  struct T { float a, b, c; } tests[] = {
      {1.0f, 1.0f, 0.f},
      {ldexpf(1.f, -20), 1024.f, 0.f},            // subnormal A
      {1024.f, ldexpf(1.f, -20), 0.f},            // subnormal B
      {ldexpf(1.f, -24), ldexpf(1.f, -24), 0.f},  // smallest subnormals: product 2^-48
      {ldexpf(1.f, -20), ldexpf(1.f, -20), 1.0f}, // tiny product into big accumulator
      {ldexpf(1.f, -12), 1.0f, 1.0f},             // 2^-12 product into 1.0: exact result 1.000244140625
      {ldexpf(1.f, -14) * 1.5f, ldexpf(1.f, -10), 1.0f},  // 1.5*2^-24 -> rounding of the accumulate
      {3.0f * ldexpf(1.f, -24), 1.0f, 0.f},       // subnormal with 2 bits
  };
  for (auto& t : tests) {
    k<<<1, 64>>>(t.a, t.b, t.c, d);
    float h[3]; (void)hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
    const double want = (double)h[1] * (double)h[2] + t.c;
    printf("a=%.6e (f16 %.6e) b=%.6e (f16 %.6e) c=%g -> got %.10e  exact %.10e  %s\n", t.a, h[1], t.b, h[2], t.c, h[0], want,
           (float)want == h[0] ? "OK" : "DIFF");
  }
  return 0;
}
