// Microbenchmark: GELU through a quadratic-interpolation table in LDS (512 x float4) against the polynomial + v_exp form the
// value-net kernel uses, per element pair incl. the f16x2 split.  One wave per SIMD, VALU + LDS only.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 splat2(float v) { return f32x2{v, v}; }
__device__ __forceinline__ f32x2 gelu_z(f32x2 z) {
  const f32x2 t = __builtin_elementwise_min(__builtin_elementwise_abs(z), splat2(4.0f));
  f32x2 r = splat2(2.635702834e-04f);
  r = fma2(r, t, splat2(-4.330650409e-03f));
  r = fma2(r, t, splat2(3.223223815e-02f));
  r = fma2(r, t, splat2(-1.509066050e-01f));
  r = fma2(r, t, splat2(-9.176831254e-01f));
  r = fma2(r, t, splat2(-1.627991484e+00f));
  r = fma2(r, t, splat2(-1.0f));
  const f32x2 e = f32x2{__builtin_amdgcn_exp2f(r[0]), __builtin_amdgcn_exp2f(r[1])};
  return fma2(t, e, __builtin_elementwise_min(-z, splat2(0.0f)));
}
__device__ __forceinline__ float gelu_lut1(float z, const f32x4* tab) {
  const float u = fminf(fabsf(z) * 128.0f, 511.96875f);
  const float fl = __builtin_floorf(u);
  const float f = u - fl;
  const f32x4 c = tab[(int)fl];
  return __builtin_fmaf(__builtin_fmaf(c[2], f, c[1]), f, c[0]) + fminf(-z, 0.0f);
}
__device__ __forceinline__ void split2(float a, float b, f16x2* hi, f16x2* lo) {
  const f16x2 h = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(a, b));
  float ra, rb;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(ra) : "v"(h), "v"(a));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rb) : "v"(h), "v"(b));
  *hi = h;
  *lo = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(ra, rb));
}
template <int V, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) k(int iters, float* out) {
  __shared__ f32x4 tab[512];
  for (int i = threadIdx.x; i < 512; i += WAVES * 64) tab[i] = f32x4{i * 1e-3f, 0.5f, -0.1f, 0.f};
  __syncthreads();
  f32x2 d[8];
  for (int i = 0; i < 8; ++i) d[i] = f32x2{threadIdx.x * 0.013f + i - 3.f, 1.0f - i * 0.37f - threadIdx.x * 0.007f};
  const f32x2 g = {0.999f, 1.001f}, o = {1e-3f, -1e-3f};
  unsigned acc = 0;
  float rs = 0.7f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const f32x2 a = g * splat2(rs);
      const f32x2 zz = fma2(d[i], a, o);
      f32x2 y;
      if (V == 0) y = gelu_z(zz);
      else y = f32x2{gelu_lut1(zz[0], tab), gelu_lut1(zz[1], tab)};
      f16x2 h, l;
      split2(y[0], y[1], &h, &l);
      acc += __builtin_bit_cast(unsigned, h) ^ __builtin_bit_cast(unsigned, l);
      d[i] = d[i] + splat2(1e-3f);
    }
    rs += 1e-6f;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (float)acc;
}
template <int V, int WAVES>
void run(const char* name) {
  float* out;
  (void)hipMalloc(&out, 256 * 512 * 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<V, WAVES><<<256, WAVES * 64>>>(10, out);
  (void)hipEventRecord(e0);
  k<V, WAVES><<<256, WAVES * 64>>>(4000, out);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-40s %d waves/SIMD  %8.1f us  = %6.1f ns per pair per wave\n", name, WAVES / 4, ms * 1e3f, ms * 1e6f / (4000.f * 8));
  (void)hipFree(out);
}
int main() {
  run<0, 4>("polynomial + v_exp (current)");
  run<1, 4>("LDS table, quadratic interpolation");
  run<0, 8>("polynomial + v_exp (current)");
  run<1, 8>("LDS table, quadratic interpolation");
  return 0;
}
