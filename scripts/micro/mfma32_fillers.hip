// Microbenchmark (round 3): how much single-issue VALU work hides in the shadow of
// v_mfma_f32_32x32x16_f16 when ONE wave owns a SIMD and the stream is hand-placed -- the regime
// /opt/skills/guides/MI355X_MICROARCH.md measures (<= 5 fillers per 32-cycle MFMA) and the regime the round-1/2
// micro-benchmarks (two waves, 16x16x32, packed f32) did not cover.
//
//   part A  asm-pinned stream: 16 MFMAs per loop iteration, NF fillers of one kind after each MFMA
//           (kinds: scalar v_fma_f32, v_pk_fma_f32, v_exp_f32, v_cvt_pkrtz_f16_f32, v_fma_mix_f32, v_min_f32,
//            ds_read_b128, ds_write_b128), NACC = 2 or 4 independent accumulators, also with v_mfma_f32_16x16x32_f16
//   part B  compiler-scheduled REAL epilogue (scale/shift, GELU polynomial + v_exp, f16x2 split; scalar f32 only) for EL
//           elements per lane beside NM MFMAs, with and without sched_group_barrier interleaving
//   part C  two waves per SIMD: waves 0-3 MFMA only, waves 4-7 scalar-fma only (32x32x16)
//
// Prints shader cycles (s_memtime) per MFMA for each mix; solo MFMA = 32 cycles, solo VALU = its own issue time.
// Build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize scripts/micro/mfma32_fillers.hip -o /tmp/mfma32_fillers
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum Kind { kNone = 0, kFma, kPkFma, kExp, kCvt, kMix, kMin, kDsRead, kDsWrite, kMul };

#define MFMA32(ACC, A, B) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "v"(B))
#define MFMA16(ACC, A, B) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "v"(B))

template <int KIND>
__device__ __forceinline__ void filler(float (&v)[16], f32x2 (&p)[8], f32x4 (&l)[4], int i, unsigned lds_addr) {
  float& x = v[i & 15];
  if constexpr (KIND == kFma) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(v[(i + 5) & 15]), "v"(v[(i + 9) & 15]));
  if constexpr (KIND == kMul) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(v[(i + 5) & 15]));
  if constexpr (KIND == kPkFma)
    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i & 7]) : "v"(p[(i + 3) & 7]), "v"(p[(i + 5) & 7]));
  if constexpr (KIND == kExp) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
  if constexpr (KIND == kCvt) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(x) : "v"(v[(i + 5) & 15]));
  if constexpr (KIND == kMix)
    asm volatile("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(x) : "v"(v[(i + 5) & 15]));
  if constexpr (KIND == kMin) asm volatile("v_min_f32 %0, |%0|, %1" : "+v"(x) : "v"(v[(i + 5) & 15]));
  if constexpr (KIND == kDsRead) asm volatile("ds_read_b128 %0, %1" : "=v"(l[i & 3]) : "v"(lds_addr));
  if constexpr (KIND == kDsWrite) asm volatile("ds_write_b128 %0, %1" ::"v"(lds_addr), "v"(l[i & 3]));
}

// ---------------------------------------------------------------------------------------------------------- part A
// SHAPE 32: 32x32x16 (16-register accumulators); 16: 16x16x32 (4-register accumulators)
template <int SHAPE, int KIND, int NF, int NACC>
__global__ void __launch_bounds__(256, 1) pinned(int iters, float* out, long long* cyc) {
  __shared__ __align__(16) float pad[24 * 1024];  // 96 KB: one workgroup per CU
  pad[threadIdx.x] = threadIdx.x;
  __syncthreads();
  const unsigned lds_addr = (unsigned)(threadIdx.x * 16);
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = (_Float16)(threadIdx.x * 0.001f + i);
    b[i] = (_Float16)(i * 0.5f);
  }
  f32x16 acc[4];
  f32x4 acs[4];
  for (int i = 0; i < 4; ++i) {
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    acs[i] = f32x4{0, 0, 0, 0};
  }
  float v[16];
  f32x2 p[8];
  f32x4 l[4];
  for (int i = 0; i < 16; ++i) v[i] = 1.0f + 1e-3f * i + 1e-6f * threadIdx.x;
  for (int i = 0; i < 8; ++i) p[i] = f32x2{1.0f + 1e-3f * i, 1.0f - 1e-3f * i};
  for (int i = 0; i < 4; ++i) l[i] = f32x4{1, 2, 3, 4};
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      if constexpr (SHAPE == 32) MFMA32(acc[q % NACC], a, b);
      else MFMA16(acs[q % NACC], a, b);
#pragma unroll
      for (int f = 0; f < NF; ++f) filler<KIND>(v, p, l, q * NF + f, lds_addr);
    }
    if constexpr (KIND == kDsRead || KIND == kDsWrite) asm volatile("s_waitcnt lgkmcnt(0)");
  }
  const long long t1 = clock64();
  float s = 0;
  for (int i = 0; i < 4; ++i) {
    for (int r = 0; r < 16; ++r) s += acc[i][r];
    s += acs[i][0] + acs[i][3] + l[i][0] + l[i][3];
  }
  for (int i = 0; i < 16; ++i) s += v[i];
  for (int i = 0; i < 8; ++i) s += p[i][0] + p[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// ---------------------------------------------------------------------------------------------------------- part B
// The real epilogue of the value net per element, scalar f32 only (no v_pk_*): z = d * (g * rs) + o; GELU(z) by the
// degree-5 fit of net_resident_kernel.hip; f16x2 split with the remainder produced by v_fma_mixlo/hi_f16.
__device__ __forceinline__ float gelu1(float z) {
  const float t = __builtin_fminf(__builtin_fabsf(z), 4.0f);
  float r = 2.635702834e-04f;
  r = __builtin_fmaf(r, t, -4.330650409e-03f);
  r = __builtin_fmaf(r, t, 3.223223815e-02f);
  r = __builtin_fmaf(r, t, -1.509066050e-01f);
  r = __builtin_fmaf(r, t, -9.176831254e-01f);
  r = __builtin_fmaf(r, t, -1.627991484e+00f);
  r = __builtin_fmaf(r, t, -1.0f);
  const float e = __builtin_amdgcn_exp2f(r);
  return __builtin_fmaf(t, e, __builtin_fminf(-z, 0.0f));
}
__device__ __forceinline__ void split_pair(float a, float b, unsigned* hi, unsigned* lo) {
  const unsigned h = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a, b));
  unsigned l = 0;
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(h), "v"(a));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(h), "v"(b));
  *hi = h;
  *lo = l;
}

// MODE 0: MFMAs only; 1: epilogue only; 2: both, compiler's own order; 3: both, sched_group_barrier (1 MFMA, PER VALU)
template <int MODE, int NM, int EL, int PER>
__global__ void __launch_bounds__(256, 1) real_epi(int iters, float* out, long long* cyc, const float* __restrict__ par) {
  __shared__ __align__(16) float pad[24 * 1024];
  pad[threadIdx.x] = threadIdx.x;
  __syncthreads();
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = (_Float16)(threadIdx.x * 0.001f + i);
    b[i] = (_Float16)(i * 0.5f);
  }
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float d[EL], g[EL], o[EL];
  for (int i = 0; i < EL; ++i) {
    d[i] = par[i] + threadIdx.x * 0.01f - 1.f;
    g[i] = par[32 + i];
    o[i] = par[64 + i];
  }
  float rs = par[100], q = 0.f;
  unsigned sink = 0;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE != 1) {
#pragma unroll
      for (int m = 0; m < NM; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m & 3], 0, 0, 0);
    }
    if constexpr (MODE != 0) {
#pragma unroll
      for (int i = 0; i < EL; i += 2) {
        float y[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const float dd = __builtin_fmaf(d[i + e], 0.5f, o[i + e]);  // acc * inv_s + bias
          q = __builtin_fmaf(dd, dd, q);                               // LayerNorm sum of squares
          y[e] = gelu1(__builtin_fmaf(dd, g[i + e] * rs, o[i + e]));
        }
        unsigned h, l;
        split_pair(y[0], y[1], &h, &l);
        sink += h ^ l;
        d[i] += 1e-3f;
        d[i + 1] -= 1e-3f;
      }
      rs += 1e-6f;
    }
    if constexpr (MODE == 3) {
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, PER, 0);  // PER VALU
      }
    }
  }
  const long long t1 = clock64();
  float s = q + (float)sink;
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < EL; ++i) s += d[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// ---------------------------------------------------------------------------------------------------------- part C
template <int NV>
__global__ void __launch_bounds__(512, 1) two_waves(int iters, int role, float* out, long long* cyc) {
  __shared__ __align__(16) float pad[24 * 1024];
  pad[threadIdx.x] = threadIdx.x;
  __syncthreads();
  const int wave = threadIdx.x >> 6;
  const bool do_m = role == 0 ? true : (role == 1 ? false : wave < 4);
  const bool do_v = role == 0 ? false : (role == 1 ? true : wave >= 4);
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = (_Float16)(threadIdx.x * 0.001f + i);
    b[i] = (_Float16)(i * 0.5f);
  }
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float v[16];
  f32x2 p[8];
  f32x4 l[4];
  for (int i = 0; i < 16; ++i) v[i] = 1.0f + 1e-3f * i;
  for (int i = 0; i < 8; ++i) p[i] = f32x2{1, 1};
  for (int i = 0; i < 4; ++i) l[i] = f32x4{1, 2, 3, 4};
  const long long t0 = clock64();
  if (do_m) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int q = 0; q < 16; ++q) MFMA32(acc[q & 3], a, b);
    }
  }
  if (do_v) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int q = 0; q < 16 * NV; ++q) filler<kFma>(v, p, l, q, 0);
    }
  }
  const long long t1 = clock64();
  float s = 0;
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}


// ---------------------------------------------------------------------------------------------------------- part D
// two waves per SIMD, BOTH running the interleaved stream (1 MFMA + NF scalar fillers, NR ds_read_b128 per 3 MFMAs):
// the regime of an 8-wave workgroup whose waves each software-pipeline GEMM(tile r+1) with the epilogue of tile r.
// Ideal (matrix pipe saturated) = 2 x 16 x 32 = 1024 cycles per iteration.
template <int KIND, int NF, int NACC, int NR>
__global__ void __launch_bounds__(512, 1) both_waves(int iters, float* out, long long* cyc) {
  __shared__ __align__(16) float pad[24 * 1024];
  pad[threadIdx.x] = threadIdx.x;
  __syncthreads();
  const int wave = threadIdx.x >> 6;
  const unsigned lds_addr = (unsigned)(threadIdx.x * 16);
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = (_Float16)(threadIdx.x * 0.001f + i);
    b[i] = (_Float16)(i * 0.5f);
  }
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float v[16];
  f32x2 p[8];
  f32x4 l[4];
  for (int i = 0; i < 16; ++i) v[i] = 1.0f + 1e-3f * i;
  for (int i = 0; i < 8; ++i) p[i] = f32x2{1, 1};
  for (int i = 0; i < 4; ++i) l[i] = f32x4{1, 2, 3, 4};
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      if (NR > 0 && q % 3 == 0) {
#pragma unroll
        for (int r = 0; r < NR; ++r) filler<kDsRead>(v, p, l, q + r, lds_addr);
      }
      MFMA32(acc[q % NACC], a, b);
#pragma unroll
      for (int f = 0; f < NF; ++f) filler<KIND>(v, p, l, q * NF + f, lds_addr);
    }
    if (NR > 0) asm volatile("s_waitcnt lgkmcnt(0)");
  }
  const long long t1 = clock64();
  float s = 0;
  for (int i = 0; i < 4; ++i) {
    for (int r = 0; r < 16; ++r) s += acc[i][r];
    s += l[i][0] + l[i][3];
  }
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

// ---------------------------------------------------------------------------------------------------------- host
static float* g_out;
static long long* g_cyc;
static float* g_par;
constexpr int kIters = 4000;

template <typename F>
static void time_it(F&& launch, double* us, long long* cyc, int n_cyc = 1) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  launch(50);
  (void)hipEventRecord(e0);
  launch(kIters);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  *us = ms * 1e3;
  (void)hipMemcpy(cyc, g_cyc, sizeof(long long) * n_cyc, hipMemcpyDeviceToHost);
}

template <int SHAPE, int KIND, int NF, int NACC>
static void runA(const char* kind) {
  double us;
  long long c;
  time_it([&](int it) { pinned<SHAPE, KIND, NF, NACC><<<256, 256>>>(it, g_out, g_cyc); }, &us, &c);
  printf("A shape=%2d nacc=%d %-12s fillers/MFMA=%2d  cycles/MFMA=%6.1f  (%8.1f us)\n", SHAPE, NACC, kind, NF,
         (double)c / kIters / 16, us);
}
template <int SHAPE, int KIND, int NACC>
static void sweepA(const char* kind) {
  runA<SHAPE, KIND, 1, NACC>(kind);
  runA<SHAPE, KIND, 2, NACC>(kind);
  runA<SHAPE, KIND, 3, NACC>(kind);
  runA<SHAPE, KIND, 4, NACC>(kind);
  runA<SHAPE, KIND, 5, NACC>(kind);
  runA<SHAPE, KIND, 6, NACC>(kind);
  runA<SHAPE, KIND, 8, NACC>(kind);
  runA<SHAPE, KIND, 10, NACC>(kind);
}
template <int MODE, int NM, int EL, int PER>
static void runB(const char* what) {
  double us;
  long long c;
  time_it([&](int it) { real_epi<MODE, NM, EL, PER><<<256, 256>>>(it, g_out, g_cyc, g_par); }, &us, &c);
  printf("B %-44s NM=%2d EL=%2d per=%2d  cycles/iter=%7.1f  cycles/MFMA=%6.1f  (%8.1f us)\n", what, NM, EL, PER,
         (double)c / kIters, NM ? (double)c / kIters / NM : 0.0, us);
}
template <int NV>
static void runC(int role, const char* what) {
  double us;
  long long c[8];
  time_it([&](int it) { two_waves<NV><<<256, 512>>>(it, role, g_out, g_cyc); }, &us, c, 8);
  printf("C %-40s fma/MFMA=%2d  cycles/iter: wave0 %7.1f  wave4 %7.1f  (%8.1f us)\n", what, NV, (double)c[0] / kIters,
         (double)c[4] / kIters, us);
}

template <int KIND, int NF, int NACC, int NR>
static void runD(const char* kind) {
  double us;
  long long c[8];
  time_it([&](int it) { both_waves<KIND, NF, NACC, NR><<<256, 512>>>(it, g_out, g_cyc); }, &us, c, 8);
  printf("D %-10s nacc=%d ds_read/3MFMA=%d fillers/MFMA=%2d  cycles/iter: wave0 %7.1f wave4 %7.1f = %5.1f per MFMA of the SIMD (%8.1f us)\n",
         kind, NACC, NR, NF, (double)c[0] / kIters, (double)c[4] / kIters,
         (double)(c[0] > c[4] ? c[0] : c[4]) / kIters / 32, us);
}
template <int KIND, int NACC, int NR>
static void sweepD(const char* kind) {
  runD<KIND, 0, NACC, NR>(kind);
  runD<KIND, 3, NACC, NR>(kind);
  runD<KIND, 5, NACC, NR>(kind);
  runD<KIND, 6, NACC, NR>(kind);
  runD<KIND, 8, NACC, NR>(kind);
  runD<KIND, 10, NACC, NR>(kind);
  runD<KIND, 12, NACC, NR>(kind);
  runD<KIND, 14, NACC, NR>(kind);
}

int main(int argc, char** argv) {
  const bool only_d = argc > 1 && argv[1][0] == 'D';
  (void)hipMalloc(&g_out, 256 * 512 * 4);
  (void)hipMalloc(&g_cyc, 64);
  float hp[128];
  for (int i = 0; i < 128; ++i) hp[i] = 0.3f + 0.01f * i;
  (void)hipMalloc(&g_par, sizeof(hp));
  (void)hipMemcpy(g_par, hp, sizeof(hp), hipMemcpyHostToDevice);

  if (!only_d) {
  printf("# part A: pinned stream, one wave per SIMD, 16 MFMAs per iteration\n");
  runA<32, kNone, 0, 4>("none");
  runA<32, kNone, 0, 2>("none");
  runA<32, kNone, 0, 1>("none");
  runA<16, kNone, 0, 4>("none");
  runA<16, kNone, 0, 2>("none");
  sweepA<32, kFma, 4>("v_fma_f32");
  sweepA<32, kFma, 2>("v_fma_f32");
  sweepA<32, kPkFma, 4>("v_pk_fma_f32");
  sweepA<32, kExp, 4>("v_exp_f32");
  sweepA<32, kCvt, 4>("v_cvt_pkrtz");
  sweepA<32, kMix, 4>("v_fma_mix");
  sweepA<32, kMin, 4>("v_min_f32");
  sweepA<32, kMul, 4>("v_mul_f32");
  runA<32, kDsRead, 1, 4>("ds_read_b128");
  runA<32, kDsRead, 2, 4>("ds_read_b128");
  runA<32, kDsWrite, 1, 4>("ds_write_b128");
  runA<32, kDsWrite, 2, 4>("ds_write_b128");
  sweepA<16, kFma, 4>("v_fma_f32");
  sweepA<16, kPkFma, 4>("v_pk_fma_f32");

  printf("# part B: real epilogue (~16 scalar VALU per element), compiler-scheduled, one wave per SIMD\n");
  runB<0, 16, 8, 0>("MFMA only");
  runB<1, 16, 4, 0>("epilogue only");
  runB<1, 16, 6, 0>("epilogue only");
  runB<1, 16, 8, 0>("epilogue only");
  runB<1, 16, 10, 0>("epilogue only");
  runB<2, 16, 4, 0>("both, compiler order");
  runB<2, 16, 6, 0>("both, compiler order");
  runB<2, 16, 8, 0>("both, compiler order");
  runB<3, 16, 4, 4>("both, sched_group_barrier");
  runB<3, 16, 4, 5>("both, sched_group_barrier");
  runB<3, 16, 6, 5>("both, sched_group_barrier");
  runB<3, 16, 6, 6>("both, sched_group_barrier");
  runB<3, 16, 8, 5>("both, sched_group_barrier");
  runB<3, 16, 8, 8>("both, sched_group_barrier");
  runB<3, 16, 10, 5>("both, sched_group_barrier");
  runB<3, 16, 10, 10>("both, sched_group_barrier");

  printf("# part C: two waves per SIMD (512 threads): waves 0-3 MFMA 32x32x16, waves 4-7 scalar v_fma_f32\n");
  runC<5>(0, "all 8 waves MFMA only");
  runC<5>(1, "all 8 waves VALU only");
  runC<5>(2, "waves 0-3 MFMA | 4-7 VALU");
  runC<8>(1, "all 8 waves VALU only");
  runC<8>(2, "waves 0-3 MFMA | 4-7 VALU");
  }
  printf("# part D: two waves per SIMD, both interleaving 16 MFMAs (32x32x16) with scalar fillers; 32.0 = matrix pipe saturated\n");
  sweepD<kFma, 1, 0>("v_fma_f32");
  sweepD<kFma, 2, 0>("v_fma_f32");
  sweepD<kFma, 1, 2>("v_fma_f32");
  sweepD<kFma, 2, 2>("v_fma_f32");
  sweepD<kExp, 2, 2>("v_exp_f32");
  sweepD<kPkFma, 2, 2>("v_pk_fma");
  return 0;
}
