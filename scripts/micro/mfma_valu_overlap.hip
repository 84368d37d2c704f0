// Microbenchmark: can f16 MFMA and packed-f32 VALU work overlap (a) inside one wave's instruction stream,
// (b) between waves that share a SIMD?  Prints cycles per loop iteration for each mix.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// mode bit0: MFMA, bit1: VALU.  role: 0 = every wave does `mode`; 1 = even waves MFMA only, odd waves VALU only
template <int NM, int NV>
__global__ void __launch_bounds__(1024) k(int iters, int mode, int role, float* out, long long* cyc) {
  const int wave = threadIdx.x >> 6;
  bool do_m = mode & 1, do_v = mode & 2;
  if (role == 1) { do_m = ((wave >> 2) & 1) == 0; do_v = !do_m; }
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  f32x2 v[8];
  for (int i = 0; i < 8; ++i) v[i] = f32x2{threadIdx.x * 0.01f + i, 1.0f + i};
  const f32x2 c1 = {0.999f, 1.001f}, c2 = {1e-3f, -1e-3f};
  __syncthreads();
  const long long t0 = clock64();
  const long long w0 = wall_clock64();
  if (do_m && do_v) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int i = 0; i < NM; ++i) acc[(r * NM + i) & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[(r * NM + i) & 3], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NV; ++i) v[(r * NV + i) & 7] = __builtin_elementwise_fma(v[(r * NV + i) & 7], c1, c2);
      }
      // ask for the pattern 1 MFMA then NV/NM VALU
#pragma unroll
      for (int r = 0; r < 4 * NM; ++r) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, NV / NM, 0);
      }
    }
  } else if (do_m) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4 * NM; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i & 3], 0, 0, 0);
    }
  } else if (do_v) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4 * NV; ++i) v[i & 7] = __builtin_elementwise_fma(v[i & 7], c1, c2);
    }
  }
  const long long t1 = clock64();
  const long long w1 = wall_clock64();
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += v[i][0] + v[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = w1 - w0; }
}

template <int NM, int NV>
void run(const char* name, int block, int mode, int role) {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 16);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NM, NV><<<256, block>>>(10, mode, role, out, cyc);
  hipEventRecord(e0);
  k<NM, NV><<<256, block>>>(iters, mode, role, out, cyc);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[2]; hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
  printf("%-44s block=%4d  %8.1f us  clock64/iter=%7.1f  wall100MHz/iter=%6.2f  (per iter: %d MFMA, %d pkVALU per wave)\n", name, block, ms * 1e3,
         (double)h[0] / iters, (double)h[1] / iters, (mode & 1 || role) ? 4 * NM : 0, (mode & 2 || role) ? 4 * NV : 0);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int block : {256, 512, 1024}) {
    run<2, 6>("MFMA only (8/iter)", block, 1, 0);
    run<2, 6>("VALU only (24/iter)", block, 2, 0);
    run<2, 6>("in-wave interleave 8 MFMA + 24 VALU", block, 3, 0);
    run<2, 8>("in-wave interleave 8 MFMA + 32 VALU", block, 3, 0);
    run<2, 6>("waves 0-3 MFMA, 4-7 VALU (share SIMDs)", block, 0, 1);
  }
  return 0;
}
