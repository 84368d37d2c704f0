// rebel_amd/csrc/net_pipe_kernel.hip -- value-net forward, SOFTWARE-PIPELINED over 32-row tiles (MlpDev::tile == 6).
//
// Net2 (cfvpy/models.py:64-94; one hidden layer of 256, liars_sp.yaml:28-33):
//   [rows, Q] f32 -> Linear -> LayerNorm -> GELU(erf) -> Linear -> LayerNorm -> GELU -> Linear -> [rows, H] f32
//
// Why this kernel exists.  The register-resident kernel (net_resident_kernel.hip, tile 5) runs its
// GEMM phases and its LayerNorm/GELU phases back to back: 6.1 k cycles of MFMA + 7.4 k cycles of VALU per 64-row group
// never overlap, and its epilogue is built on v_pk_fma_f32.  scripts/micro/mfma32_fillers.hip (output kept in
// profiles/r03_micro_mfma32_fillers*.txt) measured on MI355X what the round-1/2 micro-benchmarks missed:
//   * beside v_mfma_f32_32x32x16_f16 a wave issues 5 scalar f32 VALU instructions per MFMA for free (32.0 -> 33.5
//     cycles per MFMA); the 6th and later cost 4.7 cycles each with one wave per SIMD, 2.85 with two -- and the 5 are
//     per SIMD, not per wave (two waves with 5 fillers each: 32.7 cycles per MFMA of the SIMD; 6 each: 34; 8 each: 42);
//   * v_pk_fma_f32 is an anti-lever there: ONE per MFMA takes the MFMA from 32 to 50 cycles;
//   * v_mfma_f32_16x16x32_f16 hides only 2 fillers per 16.5-cycle MFMA; v_exp_f32 costs 8 cycles beyond the 3rd.
// So: 32x32x16 tiles, a scalar-only epilogue, and every wave interleaves the MFMAs of one row tile with the epilogue
// VALU work of its neighbours in the pipeline (sched_group_barrier pins "LDS reads, 1 MFMA, ~10 VALU").
//
// OUTCOME (MI355X, 589 824 rows).  Parity is the same as tile 5 (7e-7 max error), but the
// kernel is NOT faster: 8.4 k cycles per 32-row tile and CU against 9.4 k for tile 5, and 330 us per launch against 308
// (the denser instruction stream clocks lower).  Net2's epilogue needs ~10 non-MFMA instructions per MFMA (LayerNorm 3 +
// GELU 10 + split 1.5 per element, plus LDS traffic); the matrix pipe's shadow takes 5, the rest runs at the VALU rate:
// the kernel is VALU-issue bound whatever the interleaving, and what the overlap saves is the 5-per-MFMA share only.
// It therefore stays an opt-in variant (RBL_MLP_TILE=6) with the same parity tests as the default.
//
// Work split.  A persistent workgroup per CU loops over 32-row tiles.  Wave w owns NFT 32-feature tiles of both dense
// layers (NFT = 1: 8 waves, two per SIMD; NFT = 2: 4 waves, one per SIMD) and keeps its slice of the 256x256 hidden
// layer, hi and lo f16 halves, in registers for the whole launch (128 NFT registers).  Activations travel between the
// layers as MFMA B fragments in LDS: the D layout of the 32x32x16 MFMA (lane (n = row, h): features 8 (i / 4) + 4 h + i % 4,
// i = 0..15) IS two 16-byte B fragments of the next layer once that layer's k order is permuted on the host (pack_mlp,
// tile 6: k-step s, lane half h, position p <-> feature 16 s + 8 (p / 4) + 4 h + p % 4).
//
// Pipeline (t = tile index, two workgroup barriers per tile; R0 / R1 = the two accumulator sets of a wave):
//   half A(t):  MFMA  G1(t) k-steps 0..7 -> R[t&1], then G2(t-1) (output layer, this wave's k slice), then G0(t+1) -> R[~t&1]
//               VALU  N1(t-1) on R[~t&1]: LayerNorm scale, GELU, f16x2 split -> fragments (registers) -> G2
//               tail  V0(t+1): this wave's partial sum of squares of R[~t&1] -> LDS
//   half B(t):  MFMA  G1(t) k-steps 8..15
//               VALU  N0(t+1) on R[~t&1] -> hidden-layer B fragments of tile t+1 (LDS, double buffered); queries of tile
//                     t+2 -> f16x2 fragments (LDS); global prefetch of tile t+3's queries
//               tail  V1(t) -> LDS; one wave sums the output partials of tile t-1 and stores the rows
// Everything a stage needs from other waves crosses exactly one barrier.  LayerNorm: weights and biases are centred over
// the output features on the host, so only the variance is computed; layer 0's bias rides in a spare k column of the
// input (x[n_in] = 1), layer 1's bias is the accumulators' initial value; the 1/sqrt2 of GELU's argument, the -sqrt2 of
// its result and the power-of-two weight scales are folded into the LayerNorm constants / next layer's weights.
//
// Supported: n_layers == 2, n_hidden == 256, n_in + 1 <= 16 K0S (K0S <= 3), n_out <= 8 NO4 (NO4 <= 4); other shapes stay
// on tile 5 / 3.  Numerics: f16x2-split operands (v = hi + lo, three f16 products per multiply, f32 accumulate) as in
// the other variants; tests/test_net_parity.py holds every variant to 1e-5.
#include <stdexcept>
#include <type_traits>

#include "net_kernels.h"

namespace rbl {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

union Frag {
  f32x4 v;
  f16x8 h;
  u32x4 u;
};

#define RBL_MFMA32(A_, B_, C_) __builtin_amdgcn_mfma_f32_32x32x16_f16((A_), (B_), (C_), 0, 0, 0)
// scheduling pipeline of one chunk: NM_ x (ND_ LDS reads, 1 MFMA, NV_ VALU)
#define RBL_SGB(NM_, NV_, ND_)                                \
  _Pragma("unroll") for (int sg_ = 0; sg_ < (NM_); ++sg_) {  \
    __builtin_amdgcn_sched_group_barrier(0x100, (ND_), 0);    \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);        \
    __builtin_amdgcn_sched_group_barrier(0x002, (NV_), 0);    \
  }

// two f32 -> packed f16 pair hi (round to zero) and packed f16 pair lo = the remainders a - hi, straight out of the
// mixed-precision fma (3 instructions per pair; f16 subnormal lo parts are fine: the MFMA honours them)
__device__ __forceinline__ void split_pair(float a, float b, unsigned* hi, unsigned* lo) {
  const unsigned h = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a, b));
  unsigned l;
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l) : "v"(h), "v"(a));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(h), "v"(b));
  *hi = h;
  *lo = l;
}

// x + (the same register of the lane 32 away): copy, swap the upper half of one with the lower half of the other, add.
// Written as asm: hipcc 7.2 mis-models __builtin_amdgcn_permlane32_swap's second result.
__device__ __forceinline__ float add_other_half(float s) {
  float c;
  asm("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(s), "=&v"(c));
  return s + c;
}

// workgroup barrier that orders LDS traffic only (__syncthreads() would also drain the query prefetch and the stores)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int NFT, int K0S, int NO4>
struct PipeLds {
  static constexpr int NW = 8 / NFT;
  static constexpr int kX1 = 0;                            // [2 buffers][16 k-steps][hi, lo][64 lanes] x 16 B
  static constexpr int kX0 = kX1 + 2 * 16 * 2 * 1024;      // [K0S][hi, lo][64] x 16 B
  static constexpr int kW0 = kX0 + K0S * 2 * 1024;         // [8 feature tiles][K0S][hi, lo][64] x 16 B
  static constexpr int kWO = kW0 + 8 * K0S * 2 * 1024;     // [16 k-steps][hi, lo][64] x 16 B
  static constexpr int kS0 = kWO + 16 * 2 * 1024;          // [32 rows][NW] f32
  static constexpr int kS1 = kS0 + 32 * NW * 4;
  static constexpr int kP = kS1 + 32 * NW * 4;             // [NW][NO4][64] x 16 B
  static constexpr int kPR = kP + NW * NO4 * 1024;         // GS0, BT0, GS1, BT1, BI: 5 x [8][2][16] f32; b_out [32]
  static constexpr int kBytes = kPR + (5 * 256 + 32) * 4;
};

template <int NFT, int K0S, int NO4, bool LN>
__global__ void __launch_bounds__((8 / NFT) * 64, 1)
    mlp_pipe_kernel(const MlpDev m, const float* __restrict__ queries, int64_t rows, float* __restrict__ out, int n_tiles,
                    const long long* __restrict__ range) {
  using L = PipeLds<NFT, K0S, NO4>;
  constexpr int NW = L::NW, NT = NW * 64;
  if (range) {  // device-side row range (resident self-play: the host never learns the row counts of an epoch)
    const long long r0 = range[0], r1 = range[1];
    queries += r0 * m.n_in;
    out += r0 * m.n_out;
    rows = r1 - r0;
    n_tiles = (int)((rows + 31) / 32);
  }
  if ((int)blockIdx.x >= n_tiles) return;
  __shared__ __align__(16) unsigned char smem[L::kBytes];
  // All LDS traffic of the loop goes through byte offsets built from ONE per-thread value that is made opaque at the top
  // of every iteration: hipcc otherwise hoists each of the ~60 distinct (lane, wave)-dependent addresses out of the loop
  // into its own VGPR and then spills them (a reload from scratch costs a memory round trip: 10x on the whole kernel).
#define RBL_R128(OFF_) (*reinterpret_cast<const f32x4*>(smem + (OFF_)))
#define RBL_W128(OFF_) (*reinterpret_cast<f32x4*>(smem + (OFF_)))
#define RBL_R64(OFF_) (*reinterpret_cast<const f32x2*>(smem + (OFF_)))
#define RBL_F32(OFF_) (*reinterpret_cast<float*>(smem + (OFF_)))
#define RBL_U32(OFF_) (*reinterpret_cast<unsigned*>(smem + (OFF_)))

  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_in = m.n_in, n_out = m.n_out;
  const int G = gridDim.x;
  const int n = (n_tiles - (int)blockIdx.x + G - 1) / G;  // tiles of this workgroup: blockIdx.x + t G, t = 0..n-1

  // ------------------------------------------------------------------ resident weights + LDS images (once per launch)
  Frag w1h[NFT][16], w1l[NFT][16];
  {
    const int lane = tid & 63;
    const f32x4* w1 = reinterpret_cast<const f32x4*>(m.wh);
#pragma unroll
    for (int ft = 0; ft < NFT; ++ft)
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        w1h[ft][ks].v = w1[(((wave * NFT + ft) * 16 + ks) * 2 + 0) * 64 + lane];
        w1l[ft][ks].v = w1[(((wave * NFT + ft) * 16 + ks) * 2 + 1) * 64 + lane];
      }
    f32x4* dst = reinterpret_cast<f32x4*>(smem + L::kW0);
    const f32x4* src = reinterpret_cast<const f32x4*>(m.w0);
    for (int i = tid; i < 8 * K0S * 2 * 64; i += NT) dst[i] = src[i];
    dst = reinterpret_cast<f32x4*>(smem + L::kWO);
    src = reinterpret_cast<const f32x4*>(m.wo);
    for (int i = tid; i < 16 * 2 * 64; i += NT) dst[i] = src[i];
    float* pr = reinterpret_cast<float*>(smem + L::kPR);
    for (int i = tid; i < 256; i += NT) {
      pr[0 * 256 + i] = m.ln_w[i];        // GS0 (D-layout order)
      pr[1 * 256 + i] = m.ln_b[i];        // BT0
      pr[2 * 256 + i] = m.ln_w[256 + i];  // GS1
      pr[3 * 256 + i] = m.ln_b[256 + i];  // BT1
      pr[4 * 256 + i] = m.bias[i];        // BI: layer-1 bias x weight scale (accumulator initial values)
    }
    if (tid < 32) pr[5 * 256 + tid] = m.b_out[tid];
  }
  const float var_c0 = m.inv_scale[0] * m.inv_scale[0] * (1.0f / 256.0f);
  const float var_c1 = m.inv_scale[1] * m.inv_scale[1] * (1.0f / 256.0f);

  // ------------------------------------------------------------------ query staging: global -> registers -> f16x2 fragments
  // slot = (row r of the tile, k pair): thread `idx` of the tile-wide index space reads floats k, k + 1 of row r
  constexpr int KP = 8 * K0S;                   // k pairs per row
  constexpr int NSL = (32 * KP + NT - 1) / NT;  // slots per thread
  float qn[NSL][2];
  auto fetch_queries = [&](int t, int tid_) {
    const int64_t row0 = ((int64_t)blockIdx.x + (int64_t)t * G) * 32;
#pragma unroll
    for (int s = 0; s < NSL; ++s) {
      const int idx = tid_ + s * NT, r = idx / KP, k = 2 * (idx % KP);
      const int64_t row = row0 + r;
      const bool ok = t >= 0 && t < n && r < 32 && row < rows;
      const float* q = queries + (ok ? row : 0) * n_in;
#pragma unroll
      for (int e = 0; e < 2; ++e) qn[s][e] = (ok && k + e < n_in) ? q[k + e] : (k + e == n_in ? 1.0f : 0.0f);
    }
  };
  auto stage_queries = [&](int tid_) {
#pragma unroll
    for (int s = 0; s < NSL; ++s) {
      const int idx = tid_ + s * NT, r = idx / KP, k = 2 * (idx % KP);
      unsigned hi, lo;
      split_pair(qn[s][0], qn[s][1], &hi, &lo);
      if (NSL * NT == 32 * KP || r < 32) {
        const int ks = k >> 4, hh = (k >> 3) & 1, pp = k & 7;
        const unsigned off = L::kX0 + ((ks * 2) * 64 + hh * 32 + r) * 16 + pp * 2;
        RBL_U32(off) = hi;
        RBL_U32(off + 1024) = lo;
      }
    }
  };

  f32x16 R0[NFT], R1[NFT];
#ifdef RBL_PIPE_STAMPS  // developer build: phase stamps of iteration 10 of the first workgroups, waves 0 and 4
#define RBL_PSTAMP(K_)                                                                                      \
  do {                                                                                                      \
    if (m.dbg && t == 10 && blockIdx.x < 1024 && (threadIdx.x & 255) == 0)                                  \
      m.dbg[(size_t)blockIdx.x * 16 + (threadIdx.x >> 8) * 8 + (K_)] = (long long)clock64();               \
  } while (0)
#else
#define RBL_PSTAMP(K_) do { } while (0)
#endif

  // One pipeline iteration.  FULL: every stage is live (steady state, no branches inside the scheduling regions).
  // RG: accumulators of G1(t); RO: the other set = result of G1(t-1) (half A), then of G0(t+1).
  auto iteration = [&](auto full_tag, int t, f32x16 (&RG)[NFT], f32x16 (&RO)[NFT]) {
    constexpr bool FULL = decltype(full_tag)::value;
    const bool vG1 = FULL || (t >= 0 && t < n);          // G1(t), V1(t)
    const bool vN1 = FULL || (t >= 1 && t <= n);         // N1(t-1), G2(t-1), output rows of tile t-1
    const bool vG0 = FULL || (t + 1 >= 0 && t + 1 < n);  // G0(t+1), V0(t+1), N0(t+1)
    int tidv = threadIdx.x;
    asm volatile("" : "+v"(tidv));  // opaque: nothing derived from the lane index is hoisted out of the loop
    const unsigned lane = tidv & 63, l16 = lane * 16, half = lane >> 5, rown = lane & 31;
    const unsigned x1r = L::kX1 + (t & 1) * 32768 + l16, x1w = L::kX1 + ((t + 1) & 1) * 32768 + l16;
    // bases of the LDS regions beyond the 64 KB an instruction's immediate offset reaches; opaque, so that hipcc adds the
    // (compile-time) offsets in the instruction instead of materialising one address register per access
    unsigned pro = L::kPR + ((wave * NFT) * 2 + half) * 64;  // this lane's 16 per-feature constants (per feature tile: + 128)
    unsigned wl16 = L::kX0 + l16;                             // X0 / W0 / WO fragments of this lane
    asm volatile("" : "+v"(pro), "+v"(wl16));
    const unsigned wvo = wave * NFT;                          // first feature tile of the wave

    // 1 / sqrt(var + eps) of this lane's row from the NW partials S[row][wave]
    auto row_rstd = [&](unsigned s_off, float var_c) -> float {
      if constexpr (!LN) return 1.0f;
      float tt;
      if constexpr (NW == 8) {
        const f32x4 a = RBL_R128(s_off + rown * 32), b = RBL_R128(s_off + rown * 32 + 16);
        tt = ((a[0] + a[1]) + (a[2] + a[3])) + ((b[0] + b[1]) + (b[2] + b[3]));
      } else {
        const f32x4 a = RBL_R128(s_off + rown * 16);
        tt = (a[0] + a[1]) + (a[2] + a[3]);
      }
      const float var = __builtin_fmaf(tt, var_c, m.ln_eps);
      const float y0 = __builtin_amdgcn_rsqf(var);  // v_rsq_f32 (1 ulp) + one Newton step
      return y0 * __builtin_fmaf(-0.5f * var, y0 * y0, 1.5f);
    };
    // this wave's partial sum of squares over its NFT x 32 features, per row -> S[row][wave]
    auto var_partial = [&](const f32x16 (&acc)[NFT], unsigned s_off) {
      if constexpr (LN) {
        float q0 = 0.f, q1 = 0.f;
#pragma unroll
        for (int ft = 0; ft < NFT; ++ft)
#pragma unroll
          for (int i = 0; i < 16; i += 2) {
            q0 = __builtin_fmaf(acc[ft][i], acc[ft][i], q0);
            q1 = __builtin_fmaf(acc[ft][i + 1], acc[ft][i + 1], q1);
          }
        RBL_F32(s_off + (rown * NW + wave) * 4) = add_other_half(q0 + q1);
      }
    };
    // LayerNorm scale/shift + GELU + f16x2 split of elements 2 j, 2 j + 1 of an accumulator; gs_off / bt_off: LDS offsets of
    // the lane's constants for that feature tile
    auto norm_pair = [&](const f32x16& acc, int j, float rs, f32x2 g2, f32x2 b2, Frag& fh, Frag& fl) {
      // -GELU(sqrt2 z) / sqrt2 = t erfc(t) / 2 - max(z, 0), t = min(|z|, 4); t erfc(t) / 2 = t 2^(t R(t) - 1), R the
      // degree-5 minimax fit of net_resident_kernel.hip (scripts/fit_gelu.py: 1.5e-7 max error on GELU): 10 scalar VALU
      // instructions per element.  The two elements' chains are written side by side so that consecutive
      // instructions are independent.
      const float a0 = g2[0] * rs, a1 = g2[1] * rs;
      const float z0 = __builtin_fmaf(acc[2 * j], a0, b2[0]), z1 = __builtin_fmaf(acc[2 * j + 1], a1, b2[1]);
      const float t0 = __builtin_fminf(__builtin_fabsf(z0), 4.0f), t1 = __builtin_fminf(__builtin_fabsf(z1), 4.0f);
      float r0 = __builtin_fmaf(2.635702834e-04f, t0, -4.330650409e-03f), r1 = __builtin_fmaf(2.635702834e-04f, t1, -4.330650409e-03f);
      r0 = __builtin_fmaf(r0, t0, 3.223223815e-02f);
      r1 = __builtin_fmaf(r1, t1, 3.223223815e-02f);
      r0 = __builtin_fmaf(r0, t0, -1.509066050e-01f);
      r1 = __builtin_fmaf(r1, t1, -1.509066050e-01f);
      r0 = __builtin_fmaf(r0, t0, -9.176831254e-01f);
      r1 = __builtin_fmaf(r1, t1, -9.176831254e-01f);
      r0 = __builtin_fmaf(r0, t0, -1.627991484e+00f);
      r1 = __builtin_fmaf(r1, t1, -1.627991484e+00f);
      r0 = __builtin_fmaf(r0, t0, -1.0f);
      r1 = __builtin_fmaf(r1, t1, -1.0f);
      const float m0 = __builtin_fminf(-z0, 0.0f), m1 = __builtin_fminf(-z1, 0.0f);
      const float e0 = __builtin_amdgcn_exp2f(r0), e1 = __builtin_amdgcn_exp2f(r1);
      const float y0 = __builtin_fmaf(t0, e0, m0), y1 = __builtin_fmaf(t1, e1, m1);
      unsigned hi, lo;
      split_pair(y0, y1, &hi, &lo);
      fh.u[j % 4] = hi;
      fl.u[j % 4] = lo;
    };
    // B fragments of G1: the hi half of k-step ks + 1 is requested one chunk ahead (two register sets), the lo half at the
    // top of its own chunk (one set; its MFMA is the last of the chunk)
    Frag xh[2], xl;
    auto load_xh = [&](int ks) { xh[ks & 1].v = RBL_R128(x1r + (ks * 2 + 0) * 1024); };
    auto load_xl = [&](int ks) { xl.v = RBL_R128(x1r + (ks * 2 + 1) * 1024); };
    auto g1_step = [&](int ks) {
#pragma unroll
      for (int ft = 0; ft < NFT; ++ft) RG[ft] = RBL_MFMA32(w1l[ft][ks].h, xh[ks & 1].h, RG[ft]);
#pragma unroll
      for (int ft = 0; ft < NFT; ++ft) RG[ft] = RBL_MFMA32(w1h[ft][ks].h, xh[ks & 1].h, RG[ft]);
#pragma unroll
      for (int ft = 0; ft < NFT; ++ft) RG[ft] = RBL_MFMA32(w1h[ft][ks].h, xl.h, RG[ft]);
    };
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // ================================================================= half A
    RBL_PSTAMP(0);
    if (vG1) {
      // the accumulators of G1(t) start from the layer-1 bias image (D layout)
#pragma unroll
      for (int ft = 0; ft < NFT; ++ft)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 b4 = RBL_R128(pro + 4 * 1024 + ft * 128 + q * 16);
#pragma unroll
          for (int e = 0; e < 4; ++e) RG[ft][4 * q + e] = b4[e];
        }
      load_xh(0);
    }
    float rs1 = 1.0f;
    if (vN1) rs1 = row_rstd(L::kS1, var_c1);
    Frag fh[NFT][2], fl[NFT][2];
    // per-feature LayerNorm constants of a pair: requested one chunk ahead of their use (layer l: GS at pro + 2 l KB, BT + 1 KB)
    f32x2 gq[2][NFT], bq[2][NFT];
    auto load_gb = [&](int layer, int pi, int slot, int u) {
      gq[slot][u] = RBL_R64(pro + (2 * layer) * 1024 + (pi / 8) * 128 + 8 * (pi % 8));
      bq[slot][u] = RBL_R64(pro + (2 * layer + 1) * 1024 + (pi / 8) * 128 + 8 * (pi % 8));
    };
    if (vN1) {
#pragma unroll
      for (int u = 0; u < NFT; ++u) load_gb(1, u, 0, u);
    }
    __builtin_amdgcn_sched_barrier(0);
    // chunk ks: G1 k-step ks beside NFT element pairs of N1(t-1)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      if (vN1 && ks < 7) {
#pragma unroll
        for (int u = 0; u < NFT; ++u) load_gb(1, (ks + 1) * NFT + u, (ks + 1) & 1, u);
      }
      if (vG1) {
        load_xl(ks);
        load_xh(ks + 1);  // ks + 1 = 8 is the first k-step of half B
      }
      if (vN1) {
#pragma unroll
        for (int u = 0; u < NFT; ++u) {
          const int pi = ks * NFT + u, ft = pi / 8, j = pi % 8;
          norm_pair(RO[ft], j, rs1, gq[ks & 1][u], bq[ks & 1][u], fh[ft][j / 4], fl[ft][j / 4]);
        }
      }
      if (vG1) g1_step(ks);
      RBL_SGB(3 * NFT, 10, 2 * NFT)
      __builtin_amdgcn_sched_barrier(0);
    }
    RBL_PSTAMP(1);
    if (vN1) {  // G2(t-1): the output layer's k-steps of this wave into the set N1 has just released, then the partials
      f32x16 pacc = zero16;
#pragma unroll
      for (int fc = 0; fc < 2 * NFT; ++fc) {
        const unsigned wo = wl16 + (L::kWO - L::kX0) + ((2 * (wvo + fc / 2) + fc % 2) * 2) * 1024;
        Frag woh, wol;
        woh.v = RBL_R128(wo);
        wol.v = RBL_R128(wo + 1024);
        pacc = RBL_MFMA32(wol.h, fh[fc / 2][fc % 2].h, pacc);
        pacc = RBL_MFMA32(woh.h, fl[fc / 2][fc % 2].h, pacc);
        pacc = RBL_MFMA32(woh.h, fh[fc / 2][fc % 2].h, pacc);
      }
#pragma unroll
      for (int q = 0; q < NO4; ++q)
        RBL_W128(L::kP + (wave * NO4 + q) * 1024 + l16) = f32x4{pacc[4 * q], pacc[4 * q + 1], pacc[4 * q + 2], pacc[4 * q + 3]};
    }
    if (vG0) {  // G0(t+1): layer 0 of the next tile
#pragma unroll
      for (int ks = 0; ks < K0S; ++ks) {
        Frag xh, xl;
        xh.v = RBL_R128(wl16 + (ks * 2 + 0) * 1024);
        xl.v = RBL_R128(wl16 + (ks * 2 + 1) * 1024);
#pragma unroll
        for (int ft = 0; ft < NFT; ++ft) {
          const unsigned w0 = wl16 + (L::kW0 - L::kX0) + (((wvo + ft) * K0S + ks) * 2) * 1024;
          Frag wh, wl;
          wh.v = RBL_R128(w0);
          wl.v = RBL_R128(w0 + 1024);
          RO[ft] = RBL_MFMA32(wl.h, xh.h, ks == 0 ? zero16 : RO[ft]);
          RO[ft] = RBL_MFMA32(wh.h, xl.h, RO[ft]);
          RO[ft] = RBL_MFMA32(wh.h, xh.h, RO[ft]);
        }
      }
      var_partial(RO, L::kS0);
    }
    RBL_PSTAMP(2);
    lds_barrier();
    RBL_PSTAMP(3);

    // ================================================================= half B
    float rs0 = 1.0f;
    if (vG0) {
      rs0 = row_rstd(L::kS0, var_c0);
#pragma unroll
      for (int u = 0; u < NFT; ++u) load_gb(0, u, 0, u);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 8; ks < 16; ++ks) {
      if (vG0 && ks < 15) {
#pragma unroll
        for (int u = 0; u < NFT; ++u) load_gb(0, (ks - 7) * NFT + u, (ks + 1) & 1, u);
      }
      if (vG1) {
        load_xl(ks);
        if (ks < 15) load_xh(ks + 1);
      }
      if (vG0) {
#pragma unroll
        for (int u = 0; u < NFT; ++u) {
          const int pi = (ks - 8) * NFT + u, ft = pi / 8, j = pi % 8;
          norm_pair(RO[ft], j, rs0, gq[ks & 1][u], bq[ks & 1][u], fh[ft][j / 4], fl[ft][j / 4]);
          if (j % 4 == 3) {  // fragment pair complete: k-step 2 (wave NFT + ft) + j / 4 of the hidden layer, tile t + 1
            const unsigned xo = x1w + ((2 * (wvo + ft) + j / 4) * 2) * 1024;
            RBL_W128(xo) = fh[ft][j / 4].v;
            RBL_W128(xo + 1024) = fl[ft][j / 4].v;
          }
        }
      }
      if (ks == 15) {  // queries of tile t + 2 -> fragments, then request tile t + 3
        stage_queries(tidv);
        fetch_queries(t + 3, tidv);
      }
      if (vG1) g1_step(ks);
      RBL_SGB(3 * NFT, 10, 2 * NFT)
      __builtin_amdgcn_sched_barrier(0);
    }
    RBL_PSTAMP(4);
    if (vG1) var_partial(RG, L::kS1);
    if (vN1 && wave == ((t - 1) & (NW - 1))) {  // output rows of tile t - 1: sum the waves' k slices, bias, store
      const int64_t row = ((int64_t)blockIdx.x + (int64_t)(t - 1) * G) * 32 + rown;
#pragma unroll
      for (int q = 0; q < NO4; ++q) {
        f32x4 o = RBL_R128(L::kP + q * 1024 + l16);
#pragma unroll
        for (int w = 1; w < NW; ++w) o += RBL_R128(L::kP + (w * NO4 + q) * 1024 + l16);
        const unsigned col = 8 * q + 4 * half;
        const f32x4 r4 = o * m.inv_scale[2] + RBL_R128(L::kPR + 5 * 1024 + col * 4);
        if (row < rows) {
          float* og = out + row * n_out + col;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if ((int)col + e < n_out) og[e] = r4[e];
        }
      }
    }
    RBL_PSTAMP(5);
    lds_barrier();
    RBL_PSTAMP(6);
  };

  fetch_queries(0, tid);
  __syncthreads();  // LDS images visible
  // static issue priority for one of the two waves of a SIMD (experiment knob: MlpDev::stagger bit 2 = younger, bit 3 = older)
  if ((m.stagger & 4) && wave >= NW / 2) __builtin_amdgcn_s_setprio(1);
  if ((m.stagger & 8) && wave < NW / 2) __builtin_amdgcn_s_setprio(1);
  for (int t = -2; t <= n; ++t) {
    const bool full = t >= 1 && t <= n - 2;
    if (t & 1) {
      if (full) iteration(std::true_type{}, t, R1, R0);
      else iteration(std::false_type{}, t, R1, R0);
    } else {
      if (full) iteration(std::true_type{}, t, R0, R1);
      else iteration(std::false_type{}, t, R0, R1);
    }
  }
#undef RBL_R128
#undef RBL_W128
#undef RBL_R64
#undef RBL_F32
#undef RBL_U32
}

}  // namespace

bool mlp_pipe_supported(int n_layers, int n_in, int n_hidden, int n_out) {
  return n_layers == 2 && n_hidden == 256 && n_in >= 1 && n_in + 1 <= 48 && n_out >= 1 && n_out <= 32;
}

void launch_mlp_pipe(const MlpDev& m, const float* queries, int64_t rows, float* out, hipStream_t stream,
                     const long long* range) {
  static int n_cu[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) throw std::runtime_error("launch_mlp_pipe: no current device");
  if (dev < 64 && n_cu[dev] == 0) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    n_cu[dev] = v;
  }
  const int cus = dev < 64 ? n_cu[dev] : 256;
  const int n_tiles = (int)((rows + 31) / 32);
  const int grid = n_tiles < cus ? n_tiles : cus;
  const int k0s = m.l0_chunks, no4 = m.out_tiles, nft = (m.stagger & 3) == 2 ? 2 : 1;
#define RBL_PIPE(NFT_, K0S_, NO4_)                                                                                     \
  do {                                                                                                                 \
    if (m.use_ln)                                                                                                      \
      hipLaunchKernelGGL((mlp_pipe_kernel<NFT_, K0S_, NO4_, true>), dim3(grid), dim3((8 / NFT_) * 64), 0, stream, m,   \
                         queries, rows, out, n_tiles, range);                                                          \
    else                                                                                                               \
      hipLaunchKernelGGL((mlp_pipe_kernel<NFT_, K0S_, NO4_, false>), dim3(grid), dim3((8 / NFT_) * 64), 0, stream, m,  \
                         queries, rows, out, n_tiles, range);                                                          \
  } while (0)
  (void)nft;  // NFT = 2 (four waves, one per SIMD) compiles but loses to NFT = 1: a lone wave issues one VALU per 4.7 cycles
  if (k0s == 2 && no4 == 1) {
    RBL_PIPE(1, 2, 1);
  } else {
    throw std::runtime_error("launch_mlp_pipe: shape not instantiated");
  }
#undef RBL_PIPE
}

}  // namespace rbl
