// Microbenchmark: cycle cost of the parts of the value-net epilogue per element pair (one wave per SIMD, VALU only).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 splat2(float v) { return f32x2{v, v}; }
template <int V>
__device__ __forceinline__ f32x2 gelu_z(f32x2 z) {
  const f32x2 t = __builtin_elementwise_min(__builtin_elementwise_abs(z), splat2(4.0f));
  f32x2 r = splat2(2.635702834e-04f);
  r = fma2(r, t, splat2(-4.330650409e-03f));
  r = fma2(r, t, splat2(3.223223815e-02f));
  r = fma2(r, t, splat2(-1.509066050e-01f));
  r = fma2(r, t, splat2(-9.176831254e-01f));
  r = fma2(r, t, splat2(-1.627991484e+00f));
  r = fma2(r, t, splat2(-1.0f));
  f32x2 e;
  if (V == 1) e = r;  // no exp
  else e = f32x2{__builtin_amdgcn_exp2f(r[0]), __builtin_amdgcn_exp2f(r[1])};
  return fma2(t, e, __builtin_elementwise_min(-z, splat2(0.0f)));
}
__device__ __forceinline__ void split2(float a, float b, f16x2* hi, f16x2* lo) {
  const f16x2 h = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(a, b));
  float ra, rb;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(ra) : "v"(h), "v"(a));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rb) : "v"(h), "v"(b));
  *hi = h;
  *lo = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(ra, rb));
}
// V: 0 full, 1 no exp, 2 no split, 3 only the 7-fma polynomial, 4 only 2 exps, 5 scalar (unpacked) polynomial + exp
template <int V>
__global__ void __launch_bounds__(256) k(int iters, float* out) {
  f32x2 d[8];
  for (int i = 0; i < 8; ++i) d[i] = f32x2{threadIdx.x * 0.01f + i - 3.f, 1.0f - i};
  const f32x2 g = {0.999f, 1.001f}, o = {1e-3f, -1e-3f};
  unsigned acc = 0;
  float rs = 0.7f, fs = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (V <= 2) {
        const f32x2 a = g * splat2(rs);
        const f32x2 y = gelu_z<V == 1 ? 1 : 0>(fma2(d[i], a, o));
        if (V != 2) {
          f16x2 h, l;
          split2(y[0], y[1], &h, &l);
          acc += __builtin_bit_cast(unsigned, h) ^ __builtin_bit_cast(unsigned, l);
        } else fs += y[0] + y[1];
      } else if (V == 3) {
        f32x2 r = d[i];
        for (int q = 0; q < 7; ++q) r = fma2(r, g, o);
        d[i] = r;
      } else if (V == 4) {
        d[i] = f32x2{__builtin_amdgcn_exp2f(d[i][0]), __builtin_amdgcn_exp2f(d[i][1])};
      } else {
        float r0 = d[i][0], r1 = d[i][1];
        for (int q = 0; q < 7; ++q) { r0 = __builtin_fmaf(r0, 0.999f, 1e-3f); r1 = __builtin_fmaf(r1, 1.001f, -1e-3f); }
        d[i] = f32x2{r0, r1};
      }
      if (V <= 2) d[i] = d[i] + splat2(1e-3f);
    }
    rs += 1e-6f;
  }
  for (int i = 0; i < 8; ++i) fs += d[i][0] + d[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = fs + (float)acc;
}
template <int V>
void run(const char* name) {
  float* out;
  (void)hipMalloc(&out, 256 * 256 * 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<V><<<256, 256>>>(10, out);
  (void)hipEventRecord(e0);
  k<V><<<256, 256>>>(4000, out);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-46s %8.1f us  = %6.1f ns per pair (one wave per SIMD)\n", name, ms * 1e3f, ms * 1e6f / (4000.f * 8));
  (void)hipFree(out);
}
int main() {
  run<0>("full: scale, fma, gelu_z, split2");
  run<1>("same without the two v_exp_f32");
  run<2>("same without split2");
  run<3>("7 v_pk_fma_f32 only");
  run<4>("2 v_exp_f32 only");
  run<5>("14 scalar v_fma_f32 (unpacked polynomial)");
  return 0;
}
