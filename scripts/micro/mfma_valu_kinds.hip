// Microbenchmark: which VALU instruction classes run concurrently with f16 MFMA on a gfx950 SIMD?
// Block = 512 threads: waves 0-3 (one per SIMD) issue MFMAs, waves 4-7 (one per SIMD) issue VALU ops of one kind.
// For each kind prints: MFMA waves alone, VALU waves alone, both together.  together ~ max => overlap; ~ sum => exclusive.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

template <int KIND, int MK>
__global__ void __launch_bounds__(512) k(int iters, int run_m, int run_v, float* out) {
  __shared__ float lds[4096];
  const int wave = threadIdx.x >> 6;
  const bool is_m = wave < 4;
  for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = i;
  __syncthreads();
  float s = 0;
  if (is_m) {
    if (run_m) {
      f16x8 a, b;
      for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
      f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
      if (MK == 0) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i & 3], 0, 0, 0);
        }
      } else {
        f32x4 c16[4];
        typedef float f32x16 __attribute__((ext_vector_type(16)));
        f32x16 big[2] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
          for (int i = 0; i < 4; ++i) big[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, big[i & 1], 0, 0, 0);
        }
        for (int i = 0; i < 16; ++i) s += big[0][i] + big[1][i];
      }
      for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    }
  } else if (run_v) {
    f32x2 v[8];
    for (int i = 0; i < 8; ++i) v[i] = f32x2{threadIdx.x * 0.01f + i, 1.0f + i};
    const f32x2 c1 = {0.999f, 1.001f}, c2 = {1e-3f, -1e-3f};
    unsigned u[8];
    for (int i = 0; i < 8; ++i) u[i] = threadIdx.x * 7 + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 24; ++i) {
        if (KIND == 0) v[i & 7] = __builtin_elementwise_fma(v[i & 7], c1, c2);                // v_pk_fma_f32
        if (KIND == 1) v[i & 7][0] = __builtin_fmaf(v[i & 7][0], 0.999f, 1e-3f);              // v_fma_f32
        if (KIND == 2) u[i & 7] = (u[i & 7] ^ 0x9e3779b9u) + (u[(i + 1) & 7] >> 3);           // integer VALU (2-3 ops)
        if (KIND == 3) v[i & 7][0] = __builtin_amdgcn_exp2f(v[i & 7][0]);                     // v_exp_f32
        if (KIND == 4) v[i & 7][0] = __builtin_bit_cast(float, __builtin_amdgcn_cvt_pkrtz(v[i & 7][0], v[i & 7][1]));
        if (KIND == 5) v[i & 7][0] += lds[(threadIdx.x * 4 + i * 64 + it) & 4095];           // ds_read_b32 + v_add
        if (KIND == 6) v[i & 7][0] = fminf(fabsf(v[i & 7][0]), 4.0f) + 1.0f;                  // v_min + v_add
        if (KIND == 7) v[i & 7] = v[i & 7] * c1;                                               // v_pk_mul_f32
      }
    }
    for (int i = 0; i < 8; ++i) s += v[i][0] + v[i][1] + (float)u[i];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND, int MK>
float time_one(int run_m, int run_v) {
  float* out;
  (void)hipMalloc(&out, 256 * 512 * 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<KIND, MK><<<256, 512>>>(10, run_m, run_v, out);
  (void)hipEventRecord(e0);
  k<KIND, MK><<<256, 512>>>(2000, run_m, run_v, out);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipFree(out);
  return ms * 1e3f;
}

template <int KIND, int MK>
void run(const char* name) {
  const float m = time_one<KIND, MK>(1, 0), v = time_one<KIND, MK>(0, 1), b = time_one<KIND, MK>(1, 1);
  printf("%-34s mfma %7.1f us   valu %7.1f us   both %7.1f us   (max %7.1f, sum %7.1f) -> overlap %.0f%%\n", name, m, v, b,
         m > v ? m : v, m + v, 100.f * (m + v - b) / (m < v ? m : v));
}

int main() {
  printf("--- MFMA = v_mfma_f32_16x16x32_f16 (8/iter)\n");
  run<0, 0>("v_pk_fma_f32 x24");
  run<1, 0>("v_fma_f32 x24");
  run<2, 0>("int xor/shift/add x24");
  run<3, 0>("v_exp_f32 x24");
  run<4, 0>("v_cvt_pkrtz x24");
  run<5, 0>("ds_read_b32 + v_add x24");
  run<6, 0>("v_min|abs| + v_add x24");
  run<7, 0>("v_pk_mul_f32 x24");
  printf("--- MFMA = v_mfma_f32_32x32x16_f16 (4/iter)\n");
  run<0, 1>("v_pk_fma_f32 x24");
  run<1, 1>("v_fma_f32 x24");
  run<2, 1>("int xor/shift/add x24");
  return 0;
}
