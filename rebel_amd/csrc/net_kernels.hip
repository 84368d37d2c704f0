// rebel_amd/csrc/net_kernels.hip -- fused value-net forward on CDNA4 matrix cores (gfx950).
//
// The net is the reference's Net2 (cfvpy/models.py:64-94, conf/c02_selfplay/liars_sp.yaml:28-33):
//     n_layers x [Linear -> LayerNorm -> GELU(erf)] -> Linear,   [rows, Q] f32 -> [rows, H] f32.
// The reference evaluates it once per CFR iteration per data-gen thread on ~66 rows (ModelLocker::forward,
// csrc/liars_dice/rela/model_locker.h:85-95); here every pseudo-leaf of every lane goes through ONE launch.
//
// Design (not a GEMM-library call, not an LDS-tiled GEMM): the whole MLP of a 32-row batch tile runs inside one
// wavefront with activations resident in registers.  We compute the transposed problem  Y^T = W . X^T  with
// v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate, bitwise an fmaf chain):
//   * "A" operand = weights  W[out i][in k] : lane l holds A[i = l&31][k = l>>5]
//   * "B" operand = X^T                   : lane l holds B[k = l>>5][j = l&31]   (j = batch row)
//   * D tile                              : lane l holds, for batch row j = l&31, the 16 output features
//                                           i = (r&3) + 8*(r>>2) + 4*(l>>5), r = 0..15.
// The D layout of layer n is *already* a valid B layout for layer n+1 if the k-pairs of an MFMA step are chosen as
// (f_lo(r), f_lo(r)+4): lanes 0-31 hold feature f_lo(r), lanes 32-63 hold f_lo(r)+4 -- exactly B[k=l>>5].  So
// accumulator register r of feature tile kt feeds step (kt, r) of the next layer with no shuffle, no LDS round trip;
// the weights are pre-permuted on the host into that k order (pack_mlp) and stream from L2 as coalesced 16-byte loads.
// LayerNorm needs the 256 features of a row: they live in 2 lanes (l, l^32) x 128 registers -> one cross-half shuffle.
#include "net_kernels.h"

#include <cmath>
#include <stdexcept>
#include <type_traits>

namespace rbl {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Branch-free single-precision erf (both minimax branches evaluated, then selected): <= 1 ulp-class error like the
// device library's erff, but without its divergent control flow (the epilogue runs on 64 features x 64 lanes at once).
__device__ __forceinline__ float erf_nobranch(float a) {
  const float t = fabsf(a), s = a * a;
  // |a| > 0.927734375: erf = 1 - exp(poly)
  float r = fmaf(-1.72853470e-5f, t, 3.83197126e-4f);
  float u = fmaf(-3.88396438e-3f, t, 2.42546219e-2f);
  r = fmaf(r, s, u);
  r = fmaf(r, t, -1.06777877e-1f);
  r = fmaf(r, t, -6.34846687e-1f);
  r = fmaf(r, t, -1.28717512e-1f);
  r = fmaf(r, t, -t);
  r = 1.0f - __expf(r);
  r = copysignf(r, a);
  // |a| <= 0.927734375: erf = a + a * poly(a^2)
  float p = -5.96761703e-4f;
  p = fmaf(p, s, 4.99119423e-3f);
  p = fmaf(p, s, -2.67681349e-2f);
  p = fmaf(p, s, 1.12819925e-1f);
  p = fmaf(p, s, -3.76125336e-1f);
  p = fmaf(p, s, 1.28379166e-1f);
  p = fmaf(p, a, a);
  return t > 0.927734375f ? r : p;
}

__device__ __forceinline__ float gelu_erf(float x) {  // torch.nn.functional.gelu default (exact erf form)
  return 0.5f * x * (1.0f + erf_nobranch(x * 0.70710678118654752440f));
}


// The two waves that share a SIMD otherwise march in phase (fair arbitration keeps them aligned), so their MFMA phases
// collide and their LayerNorm/GELU (VALU) phases collide, leaving the matrix pipe idle ~40 % of the time.  A static
// priority on one of them (hardware wave-slot parity) makes the favoured wave own the matrix pipe during its MFMA phase
// while the other one fills the favoured wave's VALU phases -- the two fall into anti-phase.
__device__ __forceinline__ void stagger_priority() {
  const unsigned slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);  // HW_REG_HW_ID.wave_id
  if (slot & 1) __builtin_amdgcn_s_setprio(1);
}
// experiment: one-time start delay of ~half a tile period for half of the FIRST round of waves
__device__ __forceinline__ void stagger_sleep(bool first_round_odd, int n) {
  if (first_round_odd)
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(127);
}

// bias + LayerNorm + GELU on a register-resident [32 rows x 32*NT features] tile, in place.
template <int NT>
__device__ __forceinline__ void epilogue(f32x16 (&acc)[NT], const float* __restrict__ bias,
                                         const float* __restrict__ ln_w, const float* __restrict__ ln_b, int use_ln,
                                         float eps, int half) {
  constexpr float inv_n = 1.0f / (32 * NT);
#pragma unroll
  for (int it = 0; it < NT; ++it)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias + it * 32 + 8 * q + 4 * half);
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[it][4 * q + c] += b4[c];
    }
  if (use_ln) {
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < NT; ++it)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[it][r];
    s += __shfl_xor(s, 32);
    const float mean = s * inv_n;
    float vs = 0.f;
#pragma unroll
    for (int it = 0; it < NT; ++it)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float d = acc[it][r] - mean;
        vs += d * d;
      }
    vs += __shfl_xor(vs, 32);
    const float rstd = 1.0f / sqrtf(vs * inv_n + eps);
#pragma unroll
    for (int it = 0; it < NT; ++it)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 g4 = *reinterpret_cast<const f32x4*>(ln_w + it * 32 + 8 * q + 4 * half);
        const f32x4 o4 = *reinterpret_cast<const f32x4*>(ln_b + it * 32 + 8 * q + 4 * half);
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[it][4 * q + c] = (acc[it][4 * q + c] - mean) * rstd * g4[c] + o4[c];
      }
  }
#pragma unroll
  for (int it = 0; it < NT; ++it)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[it][r] = gelu_erf(acc[it][r]);
}

// One hidden->(32*OTILES) layer: out[it] += W-tile(it, kt) . x[kt]
template <int NT, int OTILES>
__device__ __forceinline__ void dense_from_regs(const f32x16 (&x)[NT], f32x16 (&out)[OTILES],
                                                const f32x4* __restrict__ wp, int lane) {
#pragma unroll
  for (int it = 0; it < OTILES; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r) out[it][r] = 0.f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const f32x4 a4 = wp[((it * NT + kt) * 4 + rg) * 64 + lane];
#pragma unroll
        for (int j = 0; j < 4; ++j)
          out[it] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j], x[kt][rg * 4 + j], out[it], 0, 0, 0);
      }
  }
}

template <int NT, int OT>
__global__ void __launch_bounds__(64) mlp_forward_kernel(const MlpDev m, const float* __restrict__ queries,
                                                         int64_t rows, float* __restrict__ out) {
  const int lane = threadIdx.x;
  const int j = lane & 31, half = lane >> 5;
  const int64_t row = (int64_t)blockIdx.x * 32 + j;
  const bool valid = row < rows;
  const float* qrow = queries + (valid ? row : 0) * m.n_in;

  f32x16 x[NT];
  // ---------------------------------------------------------------- layer 0: K = n_in straight from global
  {
#pragma unroll
    for (int it = 0; it < NT; ++it)
#pragma unroll
      for (int r = 0; r < 16; ++r) x[it][r] = 0.f;
    const f32x4* wp = reinterpret_cast<const f32x4*>(m.w0);
    const int sgn = m.k0_steps / 4;
    for (int sg = 0; sg < sgn; ++sg) {
      float b[4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int k = 8 * sg + 2 * jj + half;
        b[jj] = (valid && k < m.n_in) ? qrow[k] : 0.f;
      }
#pragma unroll
      for (int it = 0; it < NT; ++it) {
        const f32x4 a4 = wp[(it * sgn + sg) * 64 + lane];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) x[it] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[jj], b[jj], x[it], 0, 0, 0);
      }
    }
    epilogue<NT>(x, m.bias, m.ln_w, m.ln_b, m.use_ln, m.ln_eps, half);
  }
  // ---------------------------------------------------------------- hidden layers 1..n_layers-1, register to register
  for (int l = 1; l < m.n_layers; ++l) {
    f32x16 y[NT];
    dense_from_regs<NT, NT>(x, y, reinterpret_cast<const f32x4*>(m.wh) + (size_t)(l - 1) * NT * NT * 4 * 64, lane);
    epilogue<NT>(y, m.bias + l * 32 * NT, m.ln_w + l * 32 * NT, m.ln_b + l * 32 * NT, m.use_ln, m.ln_eps, half);
#pragma unroll
    for (int it = 0; it < NT; ++it) x[it] = y[it];
  }
  // ---------------------------------------------------------------- output layer
  {
    f32x16 o[OT];
    dense_from_regs<NT, OT>(x, o, reinterpret_cast<const f32x4*>(m.wo), lane);
    if (valid) {
      float* orow = out + row * m.n_out;
#pragma unroll
      for (int ot = 0; ot < OT; ++ot)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = ot * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (i < m.n_out) orow[i] = o[ot][r] + m.b_out[i];
        }
    }
  }
}


// ================================================================================================ 16x16x4 variant
// Same idea on v_mfma_f32_16x16x4_f32: a wavefront owns 16 batch rows; D tile: lane l holds, for row j = l&15, features
// 4*(l>>4) + r (r = 0..3) of a 16-feature tile, which is again exactly the B layout (k = l>>4) of step (tile, r) of the
// next layer.  Half the registers of the 32x32 form (64 + 64 accumulators for n_hidden = 256), so 2-3 waves fit per SIMD
// and one wave's LayerNorm/GELU epilogue (VALU) overlaps the other waves' MFMAs; the price is 2x the weight traffic
// from L2 per row, which the extra resident waves hide.
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 splat2(float v) { return f32x2{v, v}; }

// GELU(x) = 0.5 x (1 + erf(x / sqrt2)) on two elements at once (packed f32 math).  One branch-free formula:
//     erf(t) = 1 - 2^(t R(t)),  t = min(|x| / sqrt2, 4),  R = degree-7 minimax fit of log2(erfc(t)) / t on [0, 4]
// (weighted so that the error of erf itself is minimised: 1.6e-8 in exact arithmetic, 1.2e-7 evaluated in f32;
// erfc(4) = 1.5e-8, so clamping costs nothing), and GELU = x/2 + |x|/2 * erf(t).  Max |GELU error| over [-8, 8]:
// 6.8e-7 (that is 0.15 ulp-relative at x = 4.5); 14 VALU slots per element instead of ~24 for the usual two-branch
// erf.  The epilogue is what bounds the f16x2 kernel, so every slot counts.
__device__ __forceinline__ f32x2 gelu2(f32x2 x) {
  const f32x2 hx = x * splat2(0.5f);
  const f32x2 z = x * splat2(0.70710678118654752440f);
  const f32x2 t = __builtin_elementwise_min(__builtin_elementwise_abs(z), splat2(4.0f));
  f32x2 r = splat2(-4.535757872e-05f);
  r = fma2(r, t, splat2(4.454992795e-04f));
  r = fma2(r, t, splat2(-1.489414726e-03f));
  r = fma2(r, t, splat2(-7.746730062e-04f));
  r = fma2(r, t, splat2(2.825371816e-02f));
  r = fma2(r, t, splat2(-1.484816315e-01f));
  r = fma2(r, t, splat2(-9.184163899e-01f));
  r = fma2(r, t, splat2(-1.627908593e+00f));
  r = r * t;
  const f32x2 e = f32x2{__builtin_amdgcn_exp2f(r[0]), __builtin_amdgcn_exp2f(r[1])};
  return fma2(__builtin_elementwise_abs(hx), splat2(1.0f) - e, hx);
}

// bias + LayerNorm + GELU on a register-resident [16 rows x 16*NT features] tile, in place (packed f32 math).
template <int NT>
__device__ __forceinline__ void epilogue16(f32x4 (&acc)[NT], const float* __restrict__ bias,
                                           const float* __restrict__ ln_w, const float* __restrict__ ln_b, int use_ln,
                                           float eps, int g) {
  if (use_ln == 2) return;  // timing experiment only (RBL_MLP_DEBUG=1): no bias / LayerNorm / GELU
  constexpr float inv_n = 1.0f / (16 * NT);
  f32x2 s2 = splat2(0.f);
#pragma unroll
  for (int it = 0; it < NT; ++it) {
    acc[it] += *reinterpret_cast<const f32x4*>(bias + it * 16 + 4 * g);
    s2 += f32x2{acc[it][0], acc[it][1]} + f32x2{acc[it][2], acc[it][3]};
  }
  if (use_ln) {
    float s = s2[0] + s2[1];
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    const float mean = s * inv_n;
    f32x2 v2 = splat2(0.f);
#pragma unroll
    for (int it = 0; it < NT; ++it) {
      const f32x2 d0 = f32x2{acc[it][0], acc[it][1]} - splat2(mean), d1 = f32x2{acc[it][2], acc[it][3]} - splat2(mean);
      v2 = fma2(d0, d0, v2);
      v2 = fma2(d1, d1, v2);
    }
    float vs = v2[0] + v2[1];
    vs += __shfl_xor(vs, 16);
    vs += __shfl_xor(vs, 32);
    const float rstd = 1.0f / sqrtf(vs * inv_n + eps);
#pragma unroll
    for (int it = 0; it < NT; ++it) {
      const f32x4 g4 = *reinterpret_cast<const f32x4*>(ln_w + it * 16 + 4 * g);
      const f32x4 o4 = *reinterpret_cast<const f32x4*>(ln_b + it * 16 + 4 * g);
#pragma unroll
      for (int h = 0; h < 2; ++h) {  // (x - mean) * rstd * gamma + beta  ==  x * a + (beta - mean * a)
        const f32x2 a = f32x2{g4[2 * h], g4[2 * h + 1]} * splat2(rstd);
        const f32x2 b = fma2(splat2(-mean), a, f32x2{o4[2 * h], o4[2 * h + 1]});
        const f32x2 y = gelu2(fma2(f32x2{acc[it][2 * h], acc[it][2 * h + 1]}, a, b));
        acc[it][2 * h] = y[0];
        acc[it][2 * h + 1] = y[1];
      }
    }
  } else {
#pragma unroll
    for (int it = 0; it < NT; ++it)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const f32x2 y = gelu2(f32x2{acc[it][2 * h], acc[it][2 * h + 1]});
        acc[it][2 * h] = y[0];
        acc[it][2 * h + 1] = y[1];
      }
  }
}

// out[it] = W-tile(it, :) . x.  Two output tiles are in flight so that dependent MFMAs are 64 cycles apart (dependent
// latency of v_mfma_f32_16x16x4_f32 is 40, issue 32), and the weight fragments are software-pipelined through a
// register ring PF steps ahead: hipcc on its own places each global_load right in front of its first use (zero
// prefetch distance, one exposed L2 round trip per 8 MFMAs); the sched_barriers pin "8 MFMAs, then refill the slot
// they just freed" so ~(PF-1)*256 cycles of MFMA issue cover every load.
template <int NT, int OTILES>
__device__ __forceinline__ void dense16(const f32x4 (&x)[NT], f32x4 (&out)[OTILES], const f32x4* __restrict__ wp,
                                        int lane) {
  constexpr int PF = 4;                      // prefetch ring depth (steps)
  constexpr int OP = (OTILES + 1) / 2;       // output-tile pairs (the last one may be half empty)
  constexpr int T = OP * NT;                 // steps; step t = (pair t / NT, k-tile t % NT)
  f32x4 ra[PF], rb[PF];
  auto tile_b = [](int ip) { return 2 * ip + 1 < OTILES ? 2 * ip + 1 : 2 * ip; };  // odd OTILES: reuse tile a
#pragma unroll
  for (int t = 0; t < PF && t < T; ++t) {
    ra[t] = wp[((2 * (t / NT)) * NT + t % NT) * 64 + lane];
    rb[t] = wp[(tile_b(t / NT) * NT + t % NT) * 64 + lane];
  }
  f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int ip = t / NT, kt = t % NT, slot = t % PF;
    if (kt == 0) {
      a = f32x4{0.f, 0.f, 0.f, 0.f};
      b = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      a = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[slot][r], x[kt][r], a, 0, 0, 0);
      b = __builtin_amdgcn_mfma_f32_16x16x4f32(rb[slot][r], x[kt][r], b, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (t + PF < T) {
      const int u = t + PF;
      ra[slot] = wp[((2 * (u / NT)) * NT + u % NT) * 64 + lane];
      rb[slot] = wp[(tile_b(u / NT) * NT + u % NT) * 64 + lane];
    }
    if (kt == NT - 1) {
      out[2 * ip] = a;
      if (2 * ip + 1 < OTILES) out[2 * ip + 1] = b;
    }
  }
}

template <int NT, int OT>
__global__ void __launch_bounds__(64, 2) mlp16_forward_kernel(const MlpDev m, const float* __restrict__ queries,
                                                           int64_t rows, float* __restrict__ out) {
  if (m.stagger == 1) stagger_priority();
  if (m.stagger >= 2) {
    const unsigned slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);
    stagger_sleep(blockIdx.x < 2048 && (slot & 1), m.stagger);
  }
  const int lane = threadIdx.x;
  const int j = lane & 15, g = lane >> 4;
  const int64_t row = (int64_t)blockIdx.x * 16 + j;
  const bool valid = row < rows;
  const float* qrow = queries + (valid ? row : 0) * m.n_in;

  f32x4 x[NT];
  {  // layer 0: B operand straight from the query rows, k = 4*step + g
#pragma unroll
    for (int it = 0; it < NT; ++it) x[it] = f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4* wp = reinterpret_cast<const f32x4*>(m.w0);
    const int sgn = m.k0_steps / 4;
    for (int sg = 0; sg < sgn; ++sg) {
      float b[4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int k = 16 * sg + 4 * jj + g;
        b[jj] = (valid && k < m.n_in) ? qrow[k] : 0.f;
      }
#pragma unroll
      for (int it = 0; it < NT; ++it) {
        const f32x4 a4 = wp[(it * sgn + sg) * 64 + lane];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) x[it] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[jj], b[jj], x[it], 0, 0, 0);
      }
    }
    epilogue16<NT>(x, m.bias, m.ln_w, m.ln_b, m.use_ln, m.ln_eps, g);
  }
  for (int l = 1; l < m.n_layers; ++l) {
    f32x4 y[NT];
    dense16<NT, NT>(x, y, reinterpret_cast<const f32x4*>(m.wh) + (size_t)(l - 1) * NT * NT * 64, lane);
    epilogue16<NT>(y, m.bias + l * 16 * NT, m.ln_w + l * 16 * NT, m.ln_b + l * 16 * NT, m.use_ln, m.ln_eps, g);
#pragma unroll
    for (int it = 0; it < NT; ++it) x[it] = y[it];
  }
  {
    f32x4 o[OT];
    dense16<NT, OT>(x, o, reinterpret_cast<const f32x4*>(m.wo), lane);
    if (valid) {
      float* orow = out + row * m.n_out;
#pragma unroll
      for (int ot = 0; ot < OT; ++ot)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = ot * 16 + 4 * g + r;
          if (i < m.n_out) orow[i] = o[ot][r] + m.b_out[i];
        }
    }
  }
}


// ================================================================================================ LDS weight-tape variant
// n_hidden = 256 (the reference's net): 4 waves x 16 rows per workgroup share ONE copy of the weights.  The packed
// weights of the whole net form a linear "tape" that every wave consumes strictly in order (layer 0 by k-group, hidden
// layers by output-tile pair, output layer by tile pair), so the tape is streamed through a 2 x 32 KiB LDS ring with
// global_load_lds_dwordx4 (asynchronous, no VGPR staging), one 32 KiB chunk ahead of the MFMAs.  L2->CU weight traffic
// drops 4x versus every wave fetching its own fragments, and a load is in flight for a whole chunk (>= 4k cycles of
// MFMA issue) before it is needed.  Two workgroups per CU (2 waves/SIMD) run at different phases, so one's
// LayerNorm/GELU epilogue overlaps the other's MFMAs.
constexpr int kChunkF4 = 2048;  // float4 per chunk (32 KiB)

template <int OT>
__global__ void __launch_bounds__(256, 2) mlp_tape_forward_kernel(const MlpDev m, const float* __restrict__ queries,
                                                                  int64_t rows, float* __restrict__ out) {
  constexpr int NT = 16;
  __shared__ f32x4 ring[2 * kChunkF4];
  if (m.stagger == 1) stagger_priority();
  if (m.stagger >= 2) stagger_sleep(blockIdx.x >= 256 && blockIdx.x < 512, m.stagger);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int64_t row = ((int64_t)blockIdx.x * 4 + wave) * 16 + j;
  const bool valid = row < rows;
  const float* qrow = queries + (valid ? row : 0) * m.n_in;
  const f32x4* tape = reinterpret_cast<const f32x4*>(m.tape);
  const int nchunks = m.tape_chunks;

  // each wave moves 8 KiB of every chunk: 8 wave-instructions of 1 KiB (LDS destination = uniform base + lane*16)
  auto issue = [&](int c) {
    if (c < nchunks) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int off = (wave * 8 + i) * 64;
        __builtin_amdgcn_global_load_lds(tape + (size_t)c * kChunkF4 + off + lane, &ring[(c & 1) * kChunkF4 + off], 16, 0,
                                         0);
      }
    }
  };
  issue(0);
  issue(1);
  int c = 0;

  f32x4 x[NT];
#pragma unroll
  for (int it = 0; it < NT; ++it) x[it] = f32x4{0.f, 0.f, 0.f, 0.f};
  // ---------------------------------------------------------------- layer 0: two k-groups (32 inputs) per chunk
  for (int c0 = 0; c0 < m.l0_chunks; ++c0, ++c) {
    float b[2][4];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int k = 16 * (2 * c0 + s2) + 4 * jj + g;
        b[s2][jj] = (valid && k < m.n_in) ? qrow[k] : 0.f;
      }
    __syncthreads();  // chunk c landed (hipcc drains vmcnt before the barrier) and is visible to every wave
    const f32x4* buf = &ring[(c & 1) * kChunkF4];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int it = 0; it < NT; ++it) {
        const f32x4 a4 = buf[(s2 * NT + it) * 64 + lane];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) x[it] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[jj], b[s2][jj], x[it], 0, 0, 0);
      }
    __syncthreads();  // every wave is done reading the buffer before it is refilled
    issue(c + 2);
  }
  epilogue16<NT>(x, m.bias, m.ln_w, m.ln_b, m.use_ln, m.ln_eps, g);

  // one chunk = the 2 x 16 weight fragments of an output-tile pair; fragments ride a 2-step register ring from LDS
  auto pair_from_lds = [&](const f32x4* buf, f32x4& a, f32x4& bb) {
    constexpr int PF = 2;
    f32x4 ra[PF], rb[PF];
#pragma unroll
    for (int t = 0; t < PF; ++t) {
      ra[t] = buf[t * 64 + lane];
      rb[t] = buf[(NT + t) * 64 + lane];
    }
    a = f32x4{0.f, 0.f, 0.f, 0.f};
    bb = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      const int slot = kt % PF;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        a = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[slot][r], x[kt][r], a, 0, 0, 0);
        bb = __builtin_amdgcn_mfma_f32_16x16x4f32(rb[slot][r], x[kt][r], bb, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (kt + PF < NT) {
        ra[slot] = buf[(kt + PF) * 64 + lane];
        rb[slot] = buf[(NT + kt + PF) * 64 + lane];
      }
    }
  };

  // ---------------------------------------------------------------- hidden layers: 8 chunks each
  for (int l = 1; l < m.n_layers; ++l) {
    f32x4 y[NT];
#pragma unroll
    for (int ip = 0; ip < NT / 2; ++ip, ++c) {
      __syncthreads();
      pair_from_lds(&ring[(c & 1) * kChunkF4], y[2 * ip], y[2 * ip + 1]);
      __syncthreads();
      issue(c + 2);
    }
    epilogue16<NT>(y, m.bias + l * 16 * NT, m.ln_w + l * 16 * NT, m.ln_b + l * 16 * NT, m.use_ln, m.ln_eps, g);
#pragma unroll
    for (int it = 0; it < NT; ++it) x[it] = y[it];
  }
  // ---------------------------------------------------------------- output layer (tile pairs; a lone last tile is
  // followed by a zero tile on the tape)
  {
    f32x4 o[2 * ((OT + 1) / 2)];
#pragma unroll
    for (int op = 0; op < (OT + 1) / 2; ++op, ++c) {
      __syncthreads();
      pair_from_lds(&ring[(c & 1) * kChunkF4], o[2 * op], o[2 * op + 1]);
      __syncthreads();
      issue(c + 2);
    }
    if (valid) {
      float* orow = out + row * m.n_out;
#pragma unroll
      for (int ot = 0; ot < OT; ++ot)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = ot * 16 + 4 * g + r;
          if (i < m.n_out) orow[i] = o[ot][r] + m.b_out[i];
        }
    }
  }
}


// ================================================================================================ fp16 x 2-split tape variant
// f32-input MFMA executes on the SIMD's f32 FMA lanes (64 FLOP/clk/SIMD, the VALU rate), so with it the GEMM time and
// the LayerNorm/GELU VALU time ADD.  The f16 matrix pipe is 16x faster and separate.  To use it without giving up
// f32-class accuracy every operand is split into two f16 numbers, v = v_h + 2^-11 v_l (v_h = f16(v), v_l =
// f16((v - v_h) * 2^11): 22 significant bits, no subnormals thanks to the 2^11 rescale and a per-layer power-of-two
// weight scale S), and each product is evaluated as three MFMAs accumulated in f32:
//       acc1 += W_h x_h          acc2 += W_l x_h + W_h x_l          y = (acc1 + 2^-11 acc2) / S
// (f16 x f16 products are exact in f32; the dropped W_l x_l term is 2^-22 relative).  3/16 of the f32-MFMA cost.
// Measured max |error| vs float64 on O(1) outputs: ~1e-6 (tests/test_net_parity.py; bar 1e-5).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

union Frag16 {
  f32x4 v;
  f16x8 h;
};

// two f32 -> (hi, lo) f16 pairs.  RTZ on hi is fine: lo carries the remainder.
__device__ __forceinline__ void split2(float a, float b, f16x2* hi, f16x2* lo) {
  const f16x2 h = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(a, b));
  const float ra = (a - (float)h[0]) * 2048.0f, rb = (b - (float)h[1]) * 2048.0f;
  *hi = h;
  *lo = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(ra, rb));
}

__device__ __forceinline__ void split_tiles(const f32x4& t0, const f32x4& t1, f16x8* hi, f16x8* lo) {
  f16x2 h[4], l[4];
  split2(t0[0], t0[1], &h[0], &l[0]);
  split2(t0[2], t0[3], &h[1], &l[1]);
  split2(t1[0], t1[1], &h[2], &l[2]);
  split2(t1[2], t1[3], &h[3], &l[3]);
  *hi = f16x8{h[0][0], h[0][1], h[1][0], h[1][1], h[2][0], h[2][1], h[3][0], h[3][1]};
  *lo = f16x8{l[0][0], l[0][1], l[1][0], l[1][1], l[2][0], l[2][1], l[3][0], l[3][1]};
}

// Weight tape through LDS, f16x2 variant: 4-slot ring of 16 KiB chunks (one output tile = 8 k-steps x {h,l} fragments),
// filled by global_load_lds_dwordx4 three chunks ahead.  One raw s_barrier per chunk: at iteration c every wave first
// retires its own DMA pieces of chunk c+1 (counted s_waitcnt vmcnt, chunk c+2 stays in flight), the barrier then makes
// chunk c+1 visible to all waves AND proves that everybody is done with chunk c-1, whose slot is refilled with chunk c+3
// right away.  Because chunk c+1 is already visible while chunk c is being multiplied, the LDS->register fragment ring
// (PF k-steps ahead, pinned with sched_barrier) runs straight across chunk boundaries.
#ifndef RBL_NET_LDS_PAD_F4
#define RBL_NET_LDS_PAD_F4 0
#endif
constexpr int kSlotF4 = 1024;  // f32x4 (16-byte) units per 16 KiB ring slot
constexpr int kSlots = 4;

template <int OT>
__global__ void __launch_bounds__(256, 2) mlp_f16x2_forward_kernel(const MlpDev m, const float* __restrict__ queries,
                                                                   int64_t rows, float* __restrict__ out) {
  constexpr int NT = 16, KS = 8, PF = 3;
  __shared__ f32x4 ring[kSlots * kSlotF4 + RBL_NET_LDS_PAD_F4];
  if (m.stagger == 1) stagger_priority();
  if (m.stagger >= 2) stagger_sleep(blockIdx.x >= 256 && blockIdx.x < 512, m.stagger);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int64_t row = ((int64_t)blockIdx.x * 4 + wave) * 16 + j;
  const bool valid = row < rows;
  const float* qrow = queries + (valid ? row : 0) * m.n_in;
  const f32x4* tape = reinterpret_cast<const f32x4*>(m.tape);
  const int nchunks = m.tape_chunks;

  auto issue = [&](int c) {  // each wave moves 4 KiB of the chunk: 4 wave-instructions of 1 KiB
    if (c < nchunks) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int off = (wave * 4 + i) * 64;
        __builtin_amdgcn_global_load_lds(tape + (size_t)c * kSlotF4 + off + lane,
                                         &ring[(c & (kSlots - 1)) * kSlotF4 + off], 16, 0, 0);
      }
    }
  };
  // ring cursors: per-lane pointers to the slot of the current chunk and of the next one (updated once per chunk, so
  // every fragment address below is cursor + compile-time offset)
  int slot = 0;
  const f32x4* pcur = &ring[lane];
  const f32x4* pnxt = &ring[kSlotF4 + lane];
  bool first_turn = true;
  // start of iteration c: chunk c+1 landed everywhere, slot of chunk c-1 free -> refill it with chunk c+3
  auto turn = [&](int c) {
    if (c + 2 < nchunks)
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    issue(c + 3);
    if (!first_turn) {
      slot = (slot + 1) & (kSlots - 1);
      pcur = pnxt;
      pnxt = &ring[((slot + 1) & (kSlots - 1)) * kSlotF4 + lane];
    }
    first_turn = false;
  };
  auto frag = [&](const f32x4* cursor, int idx) -> f16x8 {
    Frag16 f;
    f.v = cursor[idx * 64];
    return f.h;
  };
  issue(0);
  issue(1);
  issue(2);
  int c = 0;
  constexpr float kLo = 1.0f / 2048.0f;

  f32x4 y[NT];
  {  // ---------------------------------------------------------------- layer 0: two chunks (8 tiles each) per 32 inputs
    f32x4 acc1[NT], acc2[NT];
#pragma unroll
    for (int it = 0; it < NT; ++it) {
      acc1[it] = f32x4{0.f, 0.f, 0.f, 0.f};
      acc2[it] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int ks = 0; ks < m.l0_chunks; ++ks) {
      float q8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = 32 * ks + 8 * g + e;
        q8[e] = (valid && k < m.n_in) ? qrow[k] : 0.f;
      }
      f16x2 h[4], l[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) split2(q8[2 * e], q8[2 * e + 1], &h[e], &l[e]);
      const f16x8 bh = f16x8{h[0][0], h[0][1], h[1][0], h[1][1], h[2][0], h[2][1], h[3][0], h[3][1]};
      const f16x8 bl = f16x8{l[0][0], l[0][1], l[1][0], l[1][1], l[2][0], l[2][1], l[3][0], l[3][1]};
#pragma unroll
      for (int half = 0; half < 2; ++half, ++c) {
        turn(c);
        f16x8 wh[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          wh[t] = frag(pcur, t * 2 + 0);
          const f16x8 wl = frag(pcur, t * 2 + 1);
          acc2[half * 8 + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, bh, acc2[half * 8 + t], 0, 0, 0);
          acc1[half * 8 + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[t], bh, acc1[half * 8 + t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < 8; ++t)
          acc2[half * 8 + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[t], bl, acc2[half * 8 + t], 0, 0, 0);
      }
    }
#pragma unroll
    for (int it = 0; it < NT; ++it) y[it] = (acc1[it] + acc2[it] * kLo) * m.inv_scale[0];
  }
  epilogue16<NT>(y, m.bias, m.ln_w, m.ln_b, m.use_ln, m.ln_eps, g);

  f16x8 xh[KS], xl[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) split_tiles(y[2 * ks], y[2 * ks + 1], &xh[ks], &xl[ks]);

  // TILES output tiles = TILES chunks, one long k-step stream: step t = (tile t / KS, k-step t % KS)
  auto dense = [&](auto tiles_tag, float inv_s, auto& dst) {
    constexpr int TILES = decltype(tiles_tag)::value;
    constexpr int T = TILES * KS;
    f16x8 rh[PF], rl[PF];
    turn(c);
#pragma unroll
    for (int t = 0; t < PF; ++t) {
      rh[t] = frag(pcur, t * 2 + 0);
      rl[t] = frag(pcur, t * 2 + 1);
    }
    f32x4 a1 = {0.f, 0.f, 0.f, 0.f}, a2 = a1, a3 = a1;
#pragma unroll
    for (int tile = 0; tile < TILES; ++tile) {
      if (tile > 0) turn(c + tile);  // chunk c+tile+1 becomes visible before we prefetch into it
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int t = tile * KS + ks, rs = t % PF;
        __builtin_amdgcn_sched_barrier(0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(rl[rs], xh[ks], a2, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(rh[rs], xh[ks], a1, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(rh[rs], xl[ks], a3, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (t + PF < T) {
          const int u = t + PF;
          const f32x4* cursor = (u / KS == tile) ? pcur : pnxt;  // chunk of step u: this one or the next (visible)
          rh[rs] = frag(cursor, (u % KS) * 2 + 0);
          rl[rs] = frag(cursor, (u % KS) * 2 + 1);
        }
      }
      dst[tile] = (a1 + (a2 + a3) * kLo) * inv_s;
      a1 = f32x4{0.f, 0.f, 0.f, 0.f};
      a2 = a1;
      a3 = a1;
    }
    c += TILES;
  };

  // ---------------------------------------------------------------- hidden layers: 16 chunks each
  for (int l = 1; l < m.n_layers; ++l) {
    dense(std::integral_constant<int, NT>{}, m.inv_scale[l], y);
    epilogue16<NT>(y, m.bias + l * 16 * NT, m.ln_w + l * 16 * NT, m.ln_b + l * 16 * NT, m.use_ln, m.ln_eps, g);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) split_tiles(y[2 * ks], y[2 * ks + 1], &xh[ks], &xl[ks]);
  }
  // ---------------------------------------------------------------- output layer: OT chunks
  {
    f32x4 o[OT];
    dense(std::integral_constant<int, OT>{}, m.inv_scale[m.n_layers], o);
    if (valid) {
      float* orow = out + row * m.n_out;
#pragma unroll
      for (int ot = 0; ot < OT; ++ot)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = ot * 16 + 4 * g + r;
          if (i < m.n_out) orow[i] = o[ot][r] + m.b_out[i];
        }
    }
  }
}


// ================================================================================================ feature-split variant
// The tape kernels above stream the WEIGHTS through LDS and synchronise the workgroup once per 16 KiB chunk (every ~24
// MFMAs per wave); PMC showed waves parked on those barriers / LDS waits 40 % of the time and every MFMA needing a fresh
// 1 KiB fragment from LDS.  This variant turns the sharing around:
//   * a workgroup = 8 waves = 64 batch rows (4 row tiles); wave w owns OUTPUT FEATURES [32w, 32w+32) of every layer.
//     Its weight fragments are needed by nobody else, so they go L2 -> registers directly (each weight byte is read
//     once per workgroup, as before) -- no LDS ring, no per-chunk barrier;
//   * activations are what the waves exchange, through LDS, once per layer: the GEMM writes its [64 x 32] slab of
//     pre-activations into a row-major f32 image; after a barrier LayerNorm + GELU run row-parallel (8 threads per
//     row, 32 features each), and the result is written back -- already f16x2-split and already in MFMA B-operand
//     order -- over the same bytes for the next layer.  3 barriers per layer instead of 16-19;
//   * each B fragment read from LDS now feeds 6 MFMAs (2 output tiles x 3 split products) and each weight fragment 4
//     row tiles, so LDS bytes per MFMA drop 4x; accumulators (64 VGPRs) are the only per-wave activation state, which
//     lets 4 waves share a SIMD (2 workgroups per CU) and hide each other's latencies.
constexpr int kFsYStride = 260;    // f32 per row of the pre-activation image (+4: conflict-free 16-byte column writes)

// WAVES = 8: 64 rows per workgroup, 32 features per wave, 2 workgroups per CU (66.5 KB LDS each)
// WAVES = 4: 32 rows per workgroup, 64 features per wave, 4 workgroups per CU (33 KB LDS each): more independent
//            workgroups to overlap one's LayerNorm/GELU phase with another's MFMA phase, 2x the L2 weight traffic
template <int OT, int WAVES>
__global__ void __launch_bounds__(WAVES * 64, 4) mlp_fsplit_forward_kernel(const MlpDev m,
                                                                          const float* __restrict__ queries,
                                                                          int64_t rows, float* __restrict__ out) {
  constexpr int KS = 8, RT = WAVES / 2, OTW = 16 / WAVES;  // row tiles per workgroup, output tiles per wave
  constexpr int kFsRows = RT * 16;
  constexpr int kFsLdsBytes = kFsRows * kFsYStride * 4;  // >= the f16x2 activation image (rows x 256 x 4 B)
  __shared__ __align__(16) unsigned char smem[kFsLdsBytes];
  float* Y = reinterpret_cast<float*>(smem);        // [64][260] f32
  f32x4* X = reinterpret_cast<f32x4*>(smem);        // [ks][part h,l][row tile][lane] 16-byte units (8 halves)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * kFsRows;
  constexpr float kLo = 1.0f / 2048.0f;
  if (m.stagger >= 2) stagger_sleep(blockIdx.x >= 256 && blockIdx.x < 512, m.stagger);
  long long* dbg = m.dbg && blockIdx.x < 1024 ? m.dbg + (size_t)blockIdx.x * 16 : nullptr;
  int dbg_k = 0;
#define RBL_NSTAMP()                                              \
  do {                                                            \
    if (dbg && tid == 0) dbg[dbg_k] = (long long)clock64();       \
    ++dbg_k;                                                      \
  } while (0)
  RBL_NSTAMP();

  // ---------------------------------------------------------------- stage the query rows as f16x2 B fragments
  if (tid < RT * 64) {
    const int rt = tid >> 6;
    const int64_t row = row0 + rt * 16 + j;
    const bool valid = row < rows;
    const float* qrow = queries + (valid ? row : 0) * m.n_in;
    for (int ks = 0; ks < m.l0_chunks; ++ks) {
      float q8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = 32 * ks + 8 * g + e;
        q8[e] = (valid && k < m.n_in) ? qrow[k] : 0.f;
      }
      f16x2 h[4], l[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) split2(q8[2 * e], q8[2 * e + 1], &h[e], &l[e]);
      Frag16 fh, fl;
      fh.h = f16x8{h[0][0], h[0][1], h[1][0], h[1][1], h[2][0], h[2][1], h[3][0], h[3][1]};
      fl.h = f16x8{l[0][0], l[0][1], l[1][0], l[1][1], l[2][0], l[2][1], l[3][0], l[3][1]};
      X[((ks * 2 + 0) * RT + rt) * 64 + lane] = fh.v;
      X[((ks * 2 + 1) * RT + rt) * 64 + lane] = fl.v;
    }
  }
  __syncthreads();

  // one dense layer for this wave's 2 output tiles x 4 row tiles; wg = the wave's fragments [ks][ot][part][lane]
  f32x4 acc1[OTW][RT], acc2[OTW][RT];
  auto gemm = [&](const f32x4* __restrict__ wg, int nks) {
#pragma unroll
    for (int ot = 0; ot < OTW; ++ot)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        acc1[ot][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc2[ot][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    constexpr bool kRing = OTW <= 2;  // 2 tiles/wave: double-buffer the k-step's weights; 4 tiles: registers are spent
    Frag16 wh[OTW], wl[OTW], nh[kRing ? OTW : 1], nl[kRing ? OTW : 1];
    if (kRing) {
#pragma unroll
      for (int ot = 0; ot < OTW; ++ot) {
        wh[ot].v = wg[(ot * 2 + 0) * 64 + lane];
        wl[ot].v = wg[(ot * 2 + 1) * 64 + lane];
      }
    }
    for (int ks = 0; ks < nks; ++ks) {
      if (kRing) {
        if (ks + 1 < nks) {  // next k-step's weights are in flight while this one is multiplied
#pragma unroll
          for (int ot = 0; ot < OTW; ++ot) {
            nh[kRing ? ot : 0].v = wg[(((ks + 1) * OTW + ot) * 2 + 0) * 64 + lane];
            nl[kRing ? ot : 0].v = wg[(((ks + 1) * OTW + ot) * 2 + 1) * 64 + lane];
          }
        }
      } else {
#pragma unroll
        for (int ot = 0; ot < OTW; ++ot) {
          wh[ot].v = wg[((ks * OTW + ot) * 2 + 0) * 64 + lane];
          wl[ot].v = wg[((ks * OTW + ot) * 2 + 1) * 64 + lane];
        }
      }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        Frag16 xh, xl;
        xh.v = X[((ks * 2 + 0) * RT + rt) * 64 + lane];
        xl.v = X[((ks * 2 + 1) * RT + rt) * 64 + lane];
#pragma unroll
        for (int ot = 0; ot < OTW; ++ot)
          acc2[ot][rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[ot].h, xh.h, acc2[ot][rt], 0, 0, 0);
#pragma unroll
        for (int ot = 0; ot < OTW; ++ot)
          acc1[ot][rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[ot].h, xh.h, acc1[ot][rt], 0, 0, 0);
#pragma unroll
        for (int ot = 0; ot < OTW; ++ot)
          acc2[ot][rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[ot].h, xl.h, acc2[ot][rt], 0, 0, 0);
      }
      if (kRing) {
#pragma unroll
        for (int ot = 0; ot < OTW; ++ot) {
          wh[ot] = nh[kRing ? ot : 0];
          wl[ot] = nl[kRing ? ot : 0];
        }
      }
    }
  };

  // pre-activations (+ bias) of this wave's 32 features -> row-major image; everybody must be done READING X first
  auto write_y = [&](float inv_s, const float* __restrict__ bias) {
    __syncthreads();
#pragma unroll
    for (int ot = 0; ot < OTW; ++ot) {
      const int f0 = 16 * OTW * wave + 16 * ot + 4 * g;
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias + f0);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const f32x4 v = (acc1[ot][rt] + acc2[ot][rt] * kLo) * inv_s + b4;
        *reinterpret_cast<f32x4*>(&Y[(rt * 16 + j) * kFsYStride + f0]) = v;
      }
    }
    __syncthreads();
  };

  // LayerNorm + GELU, row-parallel: thread (row = tid>>3, fg = tid&7) owns features {32 i + 4 fg + r}; the result is
  // written back over the same bytes as f16x2 B fragments in natural k order (k-step i, lane group fg>>1, half fg&1)
  auto epilogue_rows = [&](const float* __restrict__ ln_w, const float* __restrict__ ln_b) {
    // lane bits: [0] = fg bit 0, [1..3] = row bits 0..2, [4..5] = fg bits 1..2; wave = row bits 3..: the 16 lanes that one
    // ds_write_b64 services together are 8 rows x 2 halves of ONE 256-byte lane group -> 32 distinct banks (with
    // row = tid >> 3 they were 2 rows x 8 feature groups landing 4-way on the same banks; PMC: 42 % conflict cycles)
    const int fg = (lane & 1) | ((lane >> 4) << 1);
    const int row = wave * 8 + ((lane >> 1) & 7);
    f32x4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const f32x4*>(&Y[row * kFsYStride + 32 * i + 4 * fg]);
    __syncthreads();  // the image is about to be overwritten by the next layer's operands
    if (m.use_ln == 2) {
    } else if (m.use_ln) {
      f32x2 s2 = splat2(0.f);
#pragma unroll
      for (int i = 0; i < 8; ++i) s2 += f32x2{v[i][0], v[i][1]} + f32x2{v[i][2], v[i][3]};
      float s = s2[0] + s2[1];
      s += __shfl_xor(s, 1);
      s += __shfl_xor(s, 16);
      s += __shfl_xor(s, 32);
      const float mean = s * (1.0f / 256.0f);
      f32x2 q2 = splat2(0.f);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const f32x2 d0 = f32x2{v[i][0], v[i][1]} - splat2(mean), d1 = f32x2{v[i][2], v[i][3]} - splat2(mean);
        q2 = fma2(d0, d0, q2);
        q2 = fma2(d1, d1, q2);
      }
      float vs = q2[0] + q2[1];
      vs += __shfl_xor(vs, 1);
      vs += __shfl_xor(vs, 16);
      vs += __shfl_xor(vs, 32);
      const float rstd = 1.0f / sqrtf(vs * (1.0f / 256.0f) + m.ln_eps);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const f32x4 g4 = *reinterpret_cast<const f32x4*>(ln_w + 32 * i + 4 * fg);
        const f32x4 o4 = *reinterpret_cast<const f32x4*>(ln_b + 32 * i + 4 * fg);
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          const f32x2 a = f32x2{g4[2 * h2], g4[2 * h2 + 1]} * splat2(rstd);
          const f32x2 b = fma2(splat2(-mean), a, f32x2{o4[2 * h2], o4[2 * h2 + 1]});
          const f32x2 y = gelu2(fma2(f32x2{v[i][2 * h2], v[i][2 * h2 + 1]}, a, b));
          v[i][2 * h2] = y[0];
          v[i][2 * h2 + 1] = y[1];
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          const f32x2 y = gelu2(f32x2{v[i][2 * h2], v[i][2 * h2 + 1]});
          v[i][2 * h2] = y[0];
          v[i][2 * h2 + 1] = y[1];
        }
    }
    const int rt = row >> 4, lane2 = (fg >> 1) * 16 + (row & 15);
    unsigned long long* X8 = reinterpret_cast<unsigned long long*>(smem);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      f16x2 h0, l0, h1, l1;
      split2(v[i][0], v[i][1], &h0, &l0);
      split2(v[i][2], v[i][3], &h1, &l1);
      typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
      const f16x4 hh = f16x4{h0[0], h0[1], h1[0], h1[1]}, ll = f16x4{l0[0], l0[1], l1[0], l1[1]};
      X8[(((i * 2 + 0) * RT + rt) * 64 + lane2) * 2 + (fg & 1)] = __builtin_bit_cast(unsigned long long, hh);
      X8[(((i * 2 + 1) * RT + rt) * 64 + lane2) * 2 + (fg & 1)] = __builtin_bit_cast(unsigned long long, ll);
    }
    __syncthreads();
  };

  const f32x4* blob = reinterpret_cast<const f32x4*>(m.tape);
  RBL_NSTAMP();  // 1: queries staged
  // ---------------------------------------------------------------- layer 0
  gemm(blob + (size_t)wave * m.l0_chunks * OTW * 2 * 64, m.l0_chunks);
  RBL_NSTAMP();  // 2: L0 gemm
  write_y(m.inv_scale[0], m.bias);
  RBL_NSTAMP();  // 3: L0 y written
  epilogue_rows(m.ln_w, m.ln_b);
  RBL_NSTAMP();  // 4: L0 epilogue
  // ---------------------------------------------------------------- hidden layers
  for (int l = 1; l < m.n_layers; ++l) {
    const f32x4* wl = reinterpret_cast<const f32x4*>(m.wh) + ((size_t)(l - 1) * WAVES + wave) * KS * OTW * 2 * 64;
    gemm(wl, KS);
    RBL_NSTAMP();  // 5: hidden gemm
    write_y(m.inv_scale[l], m.bias + l * 256);
    RBL_NSTAMP();  // 6: hidden y written
    epilogue_rows(m.ln_w + l * 256, m.ln_b + l * 256);
    RBL_NSTAMP();  // 7: hidden epilogue
  }
  // ---------------------------------------------------------------- output layer: (tile, row tile) pairs over the waves
  for (int p = wave; p < OT * RT; p += WAVES) {
    const int ot = p / RT, rt = p % RT;
    const f32x4* wo = reinterpret_cast<const f32x4*>(m.wo) + (size_t)ot * KS * 2 * 64;
    f32x4 a1 = {0.f, 0.f, 0.f, 0.f}, a2 = a1, a3 = a1;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      Frag16 wh, wl, xh, xl;
      wh.v = wo[(ks * 2 + 0) * 64 + lane];
      wl.v = wo[(ks * 2 + 1) * 64 + lane];
      xh.v = X[((ks * 2 + 0) * RT + rt) * 64 + lane];
      xl.v = X[((ks * 2 + 1) * RT + rt) * 64 + lane];
      a2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl.h, xh.h, a2, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh.h, xh.h, a1, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh.h, xl.h, a3, 0, 0, 0);
    }
    const f32x4 o = (a1 + (a2 + a3) * kLo) * m.inv_scale[m.n_layers];
    const int64_t row = row0 + rt * 16 + j;
    if (row < rows) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = ot * 16 + 4 * g + r;
        if (i < m.n_out) out[row * m.n_out + i] = o[r] + m.b_out[i];
      }
    }
  }
  RBL_NSTAMP();  // 8: output layer
#undef RBL_NSTAMP
}

}  // namespace

bool mlp_supported(int n_layers, int n_in, int n_hidden, int n_out) {
  const bool hid_ok = n_hidden == 64 || n_hidden == 128 || n_hidden == 256;
  return n_layers >= 1 && n_in >= 1 && hid_ok && n_out >= 1 && n_out <= 64;
}

static MlpPacked pack_mlp16(int n_layers, int n_in, int n_hidden, int n_out, int use_ln, const float* const* w,
                            const float* const* b, const float* const* ln_w, const float* const* ln_b,
                            const float* w_out, const float* b_out) {
  MlpPacked p;
  p.tile = 16;
  const int NT = n_hidden / 16;
  const int k0 = (n_in + 15) / 16 * 16;  // 4 k per step, steps in groups of 4
  p.k0_steps = k0 / 4;
  p.out_tiles = (n_out + 15) / 16;
  const int sgn = p.k0_steps / 4;
  const size_t n_w0 = (size_t)NT * sgn * 64 * 4;
  const size_t n_wh = (size_t)(n_layers - 1) * NT * NT * 64 * 4;
  const size_t n_wo = (size_t)p.out_tiles * NT * 64 * 4;
  p.off_w0 = 0;
  p.off_wh = p.off_w0 + n_w0;
  p.off_wo = p.off_wh + n_wh;
  p.off_bias = p.off_wo + n_wo;
  p.off_lnw = p.off_bias + (size_t)n_layers * n_hidden;
  p.off_lnb = p.off_lnw + (size_t)n_layers * n_hidden;
  p.off_bout = p.off_lnb + (size_t)n_layers * n_hidden;
  p.blob.assign(p.off_bout + (size_t)p.out_tiles * 16, 0.f);
  float* w0 = p.blob.data() + p.off_w0;
  for (int it = 0; it < NT; ++it)
    for (int sg = 0; sg < sgn; ++sg)
      for (int lane = 0; lane < 64; ++lane)
        for (int jj = 0; jj < 4; ++jj) {
          const int k = 16 * sg + 4 * jj + (lane >> 4), i = it * 16 + (lane & 15);
          w0[(((size_t)it * sgn + sg) * 64 + lane) * 4 + jj] = k < n_in ? w[0][(size_t)i * n_in + k] : 0.f;
        }
  auto pack_hidden = [&](float* dst, const float* W, int out_tiles, int n_rows) {
    for (int it = 0; it < out_tiles; ++it)
      for (int kt = 0; kt < NT; ++kt)
        for (int lane = 0; lane < 64; ++lane)
          for (int r = 0; r < 4; ++r) {
            const int f = kt * 16 + 4 * (lane >> 4) + r;
            const int i = it * 16 + (lane & 15);
            dst[(((size_t)it * NT + kt) * 64 + lane) * 4 + r] = i < n_rows ? W[(size_t)i * n_hidden + f] : 0.f;
          }
  };
  for (int l = 1; l < n_layers; ++l)
    pack_hidden(p.blob.data() + p.off_wh + (size_t)(l - 1) * NT * NT * 64 * 4, w[l], NT, n_hidden);
  pack_hidden(p.blob.data() + p.off_wo, w_out, p.out_tiles, n_out);
  for (int l = 0; l < n_layers; ++l)
    for (int i = 0; i < n_hidden; ++i) {
      p.blob[p.off_bias + (size_t)l * n_hidden + i] = b[l][i];
      p.blob[p.off_lnw + (size_t)l * n_hidden + i] = use_ln ? ln_w[l][i] : 1.f;
      p.blob[p.off_lnb + (size_t)l * n_hidden + i] = use_ln ? ln_b[l][i] : 0.f;
    }
  for (int i = 0; i < n_out; ++i) p.blob[p.off_bout + i] = b_out[i];
  return p;
}

// tape layout (tile = 0): [layer 0: k-group pairs, [sg][it][lane][4], sg padded to even] [hidden layers: [it][kt][lane][4]]
// [output: tile pairs [ot][kt][lane][4], ot padded to even], i.e. exactly the order the kernel consumes 32 KiB chunks in.
static MlpPacked pack_mlp_tape(int n_layers, int n_in, int n_hidden, int n_out, int use_ln, const float* const* w,
                               const float* const* b, const float* const* ln_w, const float* const* ln_b,
                               const float* w_out, const float* b_out) {
  MlpPacked p;
  p.tile = 0;
  const int NT = 16;
  const int sgn = ((n_in + 15) / 16 + 1) / 2 * 2;  // k-groups of 16 inputs, padded to an even count
  p.k0_steps = sgn * 4;
  p.out_tiles = (n_out + 15) / 16;
  const int otp = (p.out_tiles + 1) / 2 * 2;
  const size_t n_w0 = (size_t)sgn * NT * 64 * 4;
  const size_t n_wh = (size_t)(n_layers - 1) * NT * NT * 64 * 4;
  const size_t n_wo = (size_t)otp * NT * 64 * 4;
  p.off_w0 = 0;
  p.off_wh = p.off_w0 + n_w0;
  p.off_wo = p.off_wh + n_wh;
  p.off_bias = p.off_wo + n_wo;
  p.off_lnw = p.off_bias + (size_t)n_layers * n_hidden;
  p.off_lnb = p.off_lnw + (size_t)n_layers * n_hidden;
  p.off_bout = p.off_lnb + (size_t)n_layers * n_hidden;
  p.blob.assign(p.off_bout + (size_t)otp * 16, 0.f);
  p.l0_chunks = sgn / 2;
  p.tape_chunks = (int)(p.off_bias / (2048 * 4));
  float* w0 = p.blob.data() + p.off_w0;
  for (int sg = 0; sg < sgn; ++sg)
    for (int it = 0; it < NT; ++it)
      for (int lane = 0; lane < 64; ++lane)
        for (int jj = 0; jj < 4; ++jj) {
          const int k = 16 * sg + 4 * jj + (lane >> 4), i = it * 16 + (lane & 15);
          w0[(((size_t)sg * NT + it) * 64 + lane) * 4 + jj] = k < n_in ? w[0][(size_t)i * n_in + k] : 0.f;
        }
  auto pack_hidden = [&](float* dst, const float* W, int out_tiles, int n_rows) {
    for (int it = 0; it < out_tiles; ++it)
      for (int kt = 0; kt < NT; ++kt)
        for (int lane = 0; lane < 64; ++lane)
          for (int r = 0; r < 4; ++r) {
            const int f = kt * 16 + 4 * (lane >> 4) + r;
            const int i = it * 16 + (lane & 15);
            dst[(((size_t)it * NT + kt) * 64 + lane) * 4 + r] = i < n_rows ? W[(size_t)i * n_hidden + f] : 0.f;
          }
  };
  for (int l = 1; l < n_layers; ++l)
    pack_hidden(p.blob.data() + p.off_wh + (size_t)(l - 1) * NT * NT * 64 * 4, w[l], NT, n_hidden);
  pack_hidden(p.blob.data() + p.off_wo, w_out, p.out_tiles, n_out);
  for (int l = 0; l < n_layers; ++l)
    for (int i = 0; i < n_hidden; ++i) {
      p.blob[p.off_bias + (size_t)l * n_hidden + i] = b[l][i];
      p.blob[p.off_lnw + (size_t)l * n_hidden + i] = use_ln ? ln_w[l][i] : 1.f;
      p.blob[p.off_lnb + (size_t)l * n_hidden + i] = use_ln ? ln_b[l][i] : 0.f;
    }
  for (int i = 0; i < n_out; ++i) p.blob[p.off_bout + i] = b_out[i];
  return p;
}

// f16 x 2-split tape (tile = 2): 16-byte units of 8 halves; fragment = 64 lanes x 16 B; chunk = 16 fragments (16 KiB).
//   layer 0 : per k-step of 32 inputs two chunks  [tile 0..7][part h,l]        k = 32 ks + 8 g + e
//   hidden  : per output tile one chunk           [k-step 0..7][part h,l]      feature = 16 (2 ks + (e>>2)) + 4 g + (e&3)
//   output  : as hidden
static MlpPacked pack_mlp_f16x2(int n_layers, int n_in, int n_hidden, int n_out, int use_ln, const float* const* w,
                                const float* const* b, const float* const* ln_w, const float* const* ln_b,
                                const float* w_out, const float* b_out) {
  MlpPacked p;
  p.tile = 2;
  const int NT = 16, KS = 8;
  const int ks0 = (n_in + 31) / 32;
  p.k0_steps = ks0;
  p.out_tiles = (n_out + 15) / 16;
  const size_t chunk_f = 1024 * 4;  // floats per 16 KiB chunk
  const size_t n_w0 = (size_t)ks0 * 2 * chunk_f;
  const size_t n_wh = (size_t)(n_layers - 1) * NT * chunk_f;
  const size_t n_wo = (size_t)p.out_tiles * chunk_f;
  p.off_w0 = 0;
  p.off_wh = n_w0;
  p.off_wo = p.off_wh + n_wh;
  p.off_bias = p.off_wo + n_wo;
  p.off_lnw = p.off_bias + (size_t)n_layers * n_hidden;
  p.off_lnb = p.off_lnw + (size_t)n_layers * n_hidden;
  p.off_bout = p.off_lnb + (size_t)n_layers * n_hidden;
  p.blob.assign(p.off_bout + (size_t)p.out_tiles * 16, 0.f);
  p.l0_chunks = ks0;
  p.tape_chunks = (int)(p.off_bias / chunk_f);
  _Float16* tape = reinterpret_cast<_Float16*>(p.blob.data());
  auto scale_of = [](const float* W, size_t n) {
    float mx = 0.f;
    for (size_t i = 0; i < n; ++i) mx = std::max(mx, std::fabs(W[i]));
    if (!(mx > 0.f) || !std::isfinite(mx)) return 1.0f;
    return std::ldexp(1.0f, (int)std::floor(std::log2(8192.0f / mx)));
  };
  // writes fragment `frag` (index in 64-lane x 8-half units from the tape start): rows i0..i0+15, k given by kfun(g, e)
  auto put = [&](size_t frag, const float* W, int ld, int n_rows, int n_cols, int i0, float S, int part, auto kfun) {
    for (int lane = 0; lane < 64; ++lane)
      for (int e = 0; e < 8; ++e) {
        const int i = i0 + (lane & 15), k = kfun(lane >> 4, e);
        const float v = (i < n_rows && k < n_cols) ? W[(size_t)i * ld + k] * S : 0.f;
        const _Float16 hi = (_Float16)v;
        const _Float16 lo = (_Float16)((v - (float)hi) * 2048.0f);
        tape[(frag * 64 + lane) * 8 + e] = part == 0 ? hi : lo;
      }
  };
  p.inv_scale.assign(n_layers + 1, 1.f);
  {
    const float S = scale_of(w[0], (size_t)n_hidden * n_in);
    p.inv_scale[0] = 1.0f / S;
    for (int ks = 0; ks < ks0; ++ks)
      for (int it = 0; it < NT; ++it)
        for (int part = 0; part < 2; ++part)
          put(((size_t)ks * 2 + it / 8) * 16 + (it % 8) * 2 + part, w[0], n_in, n_hidden, n_in, it * 16, S, part,
              [ks](int g, int e) { return 32 * ks + 8 * g + e; });
  }
  auto pack_dense = [&](size_t chunk0, const float* W, int n_rows, int n_tiles, float S) {
    for (int it = 0; it < n_tiles; ++it)
      for (int ks = 0; ks < KS; ++ks)
        for (int part = 0; part < 2; ++part)
          put((chunk0 + it) * 16 + ks * 2 + part, W, n_hidden, n_rows, n_hidden, it * 16, S, part,
              [ks](int g, int e) { return 16 * (2 * ks + (e >> 2)) + 4 * g + (e & 3); });
  };
  for (int l = 1; l < n_layers; ++l) {
    const float S = scale_of(w[l], (size_t)n_hidden * n_hidden);
    p.inv_scale[l] = 1.0f / S;
    pack_dense((size_t)ks0 * 2 + (size_t)(l - 1) * NT, w[l], n_hidden, NT, S);
  }
  {
    const float S = scale_of(w_out, (size_t)n_out * n_hidden);
    p.inv_scale[n_layers] = 1.0f / S;
    pack_dense((size_t)ks0 * 2 + (size_t)(n_layers - 1) * NT, w_out, n_out, p.out_tiles, S);
  }
  for (int l = 0; l < n_layers; ++l)
    for (int i = 0; i < n_hidden; ++i) {
      p.blob[p.off_bias + (size_t)l * n_hidden + i] = b[l][i];
      p.blob[p.off_lnw + (size_t)l * n_hidden + i] = use_ln ? ln_w[l][i] : 1.f;
      p.blob[p.off_lnb + (size_t)l * n_hidden + i] = use_ln ? ln_b[l][i] : 0.f;
    }
  for (int i = 0; i < n_out; ++i) p.blob[p.off_bout + i] = b_out[i];
  return p;
}

// feature-split layout (tile = 3): fragments of 64 lanes x 8 halves, natural k order k = 32 ks + 8 g + e.
//   layer 0 : [wave 8][k-step][tile-in-wave 2][part h,l]     (output features 32 w + 16 ot + (lane & 15))
//   hidden  : [layer][wave 8][k-step 8][tile-in-wave 2][part]
//   output  : [tile][k-step 8][part]
static MlpPacked pack_mlp_fsplit(int n_layers, int n_in, int n_hidden, int n_out, int use_ln, const float* const* w,
                                 const float* const* b, const float* const* ln_w, const float* const* ln_b,
                                 const float* w_out, const float* b_out, int waves, bool resident = false) {
  // resident (tile 5, net_resident_kernel.hip): lo halves are the plain remainders (no 2^11 pre-scale), the layers after
  // the first carry the sqrt2 that the kernel's GELU leaves out, and ln_b is stored divided by sqrt2
  MlpPacked p;
  p.tile = resident ? 5 : (waves == 4 ? 4 : 3);
  const float lo_scale = resident ? 1.0f : 2048.0f;
  const double post = resident ? -1.41421356237309504880 : 1.0, pre = resident ? 0.70710678118654752440 : 1.0;
  const int NW = waves == 4 ? 4 : 8, OTW = 16 / NW;
  const int KS = 8;
  const int ks0 = (n_in + 31) / 32;
  p.k0_steps = ks0;
  p.l0_chunks = ks0;
  p.out_tiles = (n_out + 15) / 16;
  const size_t frag_f = 64 * 4;  // floats per fragment (64 lanes x 16 B)
  const size_t n_w0 = (size_t)NW * ks0 * OTW * 2 * frag_f;
  const size_t n_wh = (size_t)(n_layers - 1) * NW * KS * OTW * 2 * frag_f;
  const size_t n_wo = (size_t)p.out_tiles * KS * 2 * frag_f;
  p.off_w0 = 0;
  p.off_wh = n_w0;
  p.off_wo = p.off_wh + n_wh;
  p.off_bias = p.off_wo + n_wo;
  p.off_lnw = p.off_bias + (size_t)n_layers * n_hidden;
  p.off_lnb = p.off_lnw + (size_t)n_layers * n_hidden;
  p.off_bout = p.off_lnb + (size_t)n_layers * n_hidden;
  p.blob.assign(p.off_bout + (size_t)p.out_tiles * 16, 0.f);
  p.tape_chunks = 0;
  _Float16* tape = reinterpret_cast<_Float16*>(p.blob.data());
  auto scale_of = [](const float* W, size_t n) {
    float mx = 0.f;
    for (size_t i = 0; i < n; ++i) mx = std::max(mx, std::fabs(W[i]));
    if (!(mx > 0.f) || !std::isfinite(mx)) return 1.0f;
    return std::ldexp(1.0f, (int)std::floor(std::log2(8192.0f / mx)));
  };
  // perm: the k order inside a 32-wide k-step is (tile, lane group, r) = the order in which the resident kernel's
  // in-register epilogue leaves the previous layer's activations (net_resident_kernel.hip: epilogue_regs)
  auto put = [&](size_t frag, const float* W, int ld, int n_rows, int n_cols, int i0, int ks, float S, int part,
                 double fold = 1.0, bool perm = false) {
    for (int lane = 0; lane < 64; ++lane)
      for (int e = 0; e < 8; ++e) {
        const int i = i0 + (lane & 15);
        const int k = perm ? 32 * ks + 16 * (e >> 2) + 4 * (lane >> 4) + (e & 3) : 32 * ks + 8 * (lane >> 4) + e;
        const float v = (i < n_rows && k < n_cols) ? (float)((double)W[(size_t)i * ld + k] * S * fold) : 0.f;
        const _Float16 hi = (_Float16)v;
        const _Float16 lo = (_Float16)((v - (float)hi) * lo_scale);
        tape[(frag * 64 + lane) * 8 + e] = part == 0 ? hi : lo;
      }
  };
  // resident + LayerNorm: centre each dense layer over its output features (W[f][k] -= mean_f W[f][k], b -= mean(b)); the
  // GEMM then produces y - mean(y) directly and the kernel's LayerNorm only has the variance left to compute
  std::vector<std::vector<float>> wc(n_layers), bc(n_layers);
  std::vector<const float*> wp(n_layers), bp(n_layers);
  for (int l = 0; l < n_layers; ++l) {
    const int K = l == 0 ? n_in : n_hidden;
    wp[l] = w[l];
    bp[l] = b[l];
    if (resident && use_ln) {
      wc[l].assign(w[l], w[l] + (size_t)n_hidden * K);
      bc[l].assign(b[l], b[l] + n_hidden);
      for (int k = 0; k < K; ++k) {
        double mu = 0;
        for (int f = 0; f < n_hidden; ++f) mu += w[l][(size_t)f * K + k];
        mu /= n_hidden;
        for (int f = 0; f < n_hidden; ++f) wc[l][(size_t)f * K + k] = (float)((double)w[l][(size_t)f * K + k] - mu);
      }
      double mb = 0;
      for (int f = 0; f < n_hidden; ++f) mb += b[l][f];
      mb /= n_hidden;
      for (int f = 0; f < n_hidden; ++f) bc[l][f] = (float)((double)b[l][f] - mb);
      wp[l] = wc[l].data();
      bp[l] = bc[l].data();
    }
  }
  p.inv_scale.assign(n_layers + 1, 1.f);
  {
    const float S = scale_of(wp[0], (size_t)n_hidden * n_in);
    p.inv_scale[0] = 1.0f / S;
    for (int wv = 0; wv < NW; ++wv)
      for (int ks = 0; ks < ks0; ++ks)
        for (int ot = 0; ot < OTW; ++ot)
          for (int part = 0; part < 2; ++part)
            put(((size_t)(wv * ks0 + ks) * OTW + ot) * 2 + part, wp[0], n_in, n_hidden, n_in, 16 * (OTW * wv + ot), ks, S,
                part);
  }
  const size_t frag_wh = p.off_wh / frag_f, frag_wo = p.off_wo / frag_f;
  for (int l = 1; l < n_layers; ++l) {
    const float S = scale_of(wp[l], (size_t)n_hidden * n_hidden);
    p.inv_scale[l] = 1.0f / S;
    for (int wv = 0; wv < NW; ++wv)
      for (int ks = 0; ks < KS; ++ks)
        for (int ot = 0; ot < OTW; ++ot)
          for (int part = 0; part < 2; ++part)
            put(frag_wh + ((((size_t)(l - 1) * NW + wv) * KS + ks) * OTW + ot) * 2 + part, wp[l], n_hidden, n_hidden,
                n_hidden, 16 * (OTW * wv + ot), ks, S, part, post, resident);
  }
  {
    const float S = scale_of(w_out, (size_t)n_out * n_hidden);
    p.inv_scale[n_layers] = 1.0f / S;
    for (int ot = 0; ot < p.out_tiles; ++ot)
      for (int ks = 0; ks < KS; ++ks)
        for (int part = 0; part < 2; ++part)
          put(frag_wo + ((size_t)ot * KS + ks) * 2 + part, w_out, n_hidden, n_out, n_hidden, 16 * ot, ks, S, part, post, resident);
  }
  for (int l = 0; l < n_layers; ++l)
    for (int i = 0; i < n_hidden; ++i) {
      p.blob[p.off_bias + (size_t)l * n_hidden + i] = bp[l][i];
      p.blob[p.off_lnw + (size_t)l * n_hidden + i] = use_ln ? ln_w[l][i] : 1.f;
      p.blob[p.off_lnb + (size_t)l * n_hidden + i] = use_ln ? (float)((double)ln_b[l][i] * pre) : 0.f;
    }
  for (int i = 0; i < n_out; ++i) p.blob[p.off_bout + i] = b_out[i];
  return p;
}

MlpPacked pack_mlp(int n_layers, int n_in, int n_hidden, int n_out, int use_ln, const float* const* w,
                   const float* const* b, const float* const* ln_w, const float* const* ln_b, const float* w_out,
                   const float* b_out, int tile) {
  if (!mlp_supported(n_layers, n_in, n_hidden, n_out))
    throw std::runtime_error("value net shape not supported by the MFMA forward (n_hidden in {64,128,256}, n_out <= 64)");
  if (tile == 5 && mlp_resident_supported(n_layers, n_in, n_hidden, n_out))
    return pack_mlp_fsplit(n_layers, n_in, n_hidden, n_out, use_ln, w, b, ln_w, ln_b, w_out, b_out, 8, true);
  if (tile == 5) tile = 3;
  if ((tile == 3 || tile == 4) && n_hidden == 256 && n_out <= 64 && n_layers <= 7 && n_in <= 128)
    return pack_mlp_fsplit(n_layers, n_in, n_hidden, n_out, use_ln, w, b, ln_w, ln_b, w_out, b_out, tile == 4 ? 4 : 8);
  if ((tile == 2 || tile == 3 || tile == 4) && n_hidden == 256 && n_out <= 64 && n_layers <= 7)
    return pack_mlp_f16x2(n_layers, n_in, n_hidden, n_out, use_ln, w, b, ln_w, ln_b, w_out, b_out);
  if ((tile == 0 || tile == 2 || tile == 3 || tile == 4) && n_hidden == 256 && n_out <= 64)
    return pack_mlp_tape(n_layers, n_in, n_hidden, n_out, use_ln, w, b, ln_w, ln_b, w_out, b_out);
  if (tile != 32) return pack_mlp16(n_layers, n_in, n_hidden, n_out, use_ln, w, b, ln_w, ln_b, w_out, b_out);
  MlpPacked p;
  p.tile = 32;
  const int NT = n_hidden / 32;
  const int k0 = (n_in + 7) / 8 * 8;  // k-pairs in groups of 4
  p.k0_steps = k0 / 2;
  p.out_tiles = (n_out + 31) / 32;
  const size_t n_w0 = (size_t)NT * (p.k0_steps / 4) * 64 * 4;
  const size_t n_wh = (size_t)(n_layers - 1) * NT * NT * 4 * 64 * 4;
  const size_t n_wo = (size_t)p.out_tiles * NT * 4 * 64 * 4;
  p.off_w0 = 0;
  p.off_wh = p.off_w0 + n_w0;
  p.off_wo = p.off_wh + n_wh;
  p.off_bias = p.off_wo + n_wo;
  p.off_lnw = p.off_bias + (size_t)n_layers * n_hidden;
  p.off_lnb = p.off_lnw + (size_t)n_layers * n_hidden;
  p.off_bout = p.off_lnb + (size_t)n_layers * n_hidden;
  p.blob.assign(p.off_bout + (size_t)p.out_tiles * 32, 0.f);
  float* w0 = p.blob.data() + p.off_w0;
  const int sgn = p.k0_steps / 4;
  for (int it = 0; it < NT; ++it)
    for (int sg = 0; sg < sgn; ++sg)
      for (int lane = 0; lane < 64; ++lane)
        for (int jj = 0; jj < 4; ++jj) {
          const int k = 8 * sg + 2 * jj + (lane >> 5), i = it * 32 + (lane & 31);
          w0[(((size_t)it * sgn + sg) * 64 + lane) * 4 + jj] = k < n_in ? w[0][(size_t)i * n_in + k] : 0.f;
        }
  auto pack_hidden = [&](float* dst, const float* W, int out_tiles, int n_rows) {
    for (int it = 0; it < out_tiles; ++it)
      for (int kt = 0; kt < NT; ++kt)
        for (int rg = 0; rg < 4; ++rg)
          for (int lane = 0; lane < 64; ++lane)
            for (int jj = 0; jj < 4; ++jj) {
              const int r = rg * 4 + jj;
              const int f = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
              const int i = it * 32 + (lane & 31);
              dst[((((size_t)it * NT + kt) * 4 + rg) * 64 + lane) * 4 + jj] =
                  i < n_rows ? W[(size_t)i * n_hidden + f] : 0.f;
            }
  };
  for (int l = 1; l < n_layers; ++l)
    pack_hidden(p.blob.data() + p.off_wh + (size_t)(l - 1) * NT * NT * 4 * 64 * 4, w[l], NT, n_hidden);
  pack_hidden(p.blob.data() + p.off_wo, w_out, p.out_tiles, n_out);
  for (int l = 0; l < n_layers; ++l)
    for (int i = 0; i < n_hidden; ++i) {
      p.blob[p.off_bias + (size_t)l * n_hidden + i] = b[l][i];
      p.blob[p.off_lnw + (size_t)l * n_hidden + i] = use_ln ? ln_w[l][i] : 1.f;
      p.blob[p.off_lnb + (size_t)l * n_hidden + i] = use_ln ? ln_b[l][i] : 0.f;
    }
  for (int i = 0; i < n_out; ++i) p.blob[p.off_bout + i] = b_out[i];
  return p;
}

static void launch_mlp16(const MlpDev& m, const float* queries, int64_t rows, float* out, hipStream_t stream) {
  const dim3 grid((unsigned)((rows + 15) / 16)), block(64);
  const int NT = m.n_hidden / 16;
#define RBL_LAUNCH16(NT_, OT_) \
  hipLaunchKernelGGL((mlp16_forward_kernel<NT_, OT_>), grid, block, 0, stream, m, queries, rows, out)
  if (NT == 16 && m.out_tiles == 1) RBL_LAUNCH16(16, 1);
  else if (NT == 16 && m.out_tiles == 2) RBL_LAUNCH16(16, 2);
  else if (NT == 16 && m.out_tiles == 3) RBL_LAUNCH16(16, 3);
  else if (NT == 16 && m.out_tiles == 4) RBL_LAUNCH16(16, 4);
  else if (NT == 8 && m.out_tiles <= 4) {
    if (m.out_tiles == 1) RBL_LAUNCH16(8, 1);
    else if (m.out_tiles == 2) RBL_LAUNCH16(8, 2);
    else if (m.out_tiles == 3) RBL_LAUNCH16(8, 3);
    else RBL_LAUNCH16(8, 4);
  } else if (NT == 4 && m.out_tiles <= 4) {
    if (m.out_tiles == 1) RBL_LAUNCH16(4, 1);
    else if (m.out_tiles == 2) RBL_LAUNCH16(4, 2);
    else if (m.out_tiles == 3) RBL_LAUNCH16(4, 3);
    else RBL_LAUNCH16(4, 4);
  } else throw std::runtime_error("launch_mlp_forward: unsupported shape");
#undef RBL_LAUNCH16
}

void launch_mlp_forward(const MlpDev& m, const float* queries, int64_t rows, float* out, hipStream_t stream) {
  if (rows <= 0) return;
  if (m.tile == 5) return launch_mlp_resident(m, queries, rows, out, stream);
  if (m.tile == 3 || m.tile == 4) {
#define RBL_FS(OT_, W_) \
  hipLaunchKernelGGL((mlp_fsplit_forward_kernel<OT_, W_>), dim3((unsigned)((rows + W_ * 8 - 1) / (W_ * 8))), dim3(W_ * 64), \
                     0, stream, m, queries, rows, out)
    if (m.tile == 3) {
      switch (m.out_tiles) {
        case 1: RBL_FS(1, 8); break;
        case 2: RBL_FS(2, 8); break;
        case 3: RBL_FS(3, 8); break;
        case 4: RBL_FS(4, 8); break;
        default: throw std::runtime_error("launch_mlp_forward: unsupported n_out");
      }
    } else {
      switch (m.out_tiles) {
        case 1: RBL_FS(1, 4); break;
        case 2: RBL_FS(2, 4); break;
        case 3: RBL_FS(3, 4); break;
        case 4: RBL_FS(4, 4); break;
        default: throw std::runtime_error("launch_mlp_forward: unsupported n_out");
      }
    }
#undef RBL_FS
    return;
  }
  if (m.tile == 2) {
    const dim3 grid((unsigned)((rows + 63) / 64)), block(256);
    switch (m.out_tiles) {
      case 1: hipLaunchKernelGGL(mlp_f16x2_forward_kernel<1>, grid, block, 0, stream, m, queries, rows, out); break;
      case 2: hipLaunchKernelGGL(mlp_f16x2_forward_kernel<2>, grid, block, 0, stream, m, queries, rows, out); break;
      case 3: hipLaunchKernelGGL(mlp_f16x2_forward_kernel<3>, grid, block, 0, stream, m, queries, rows, out); break;
      case 4: hipLaunchKernelGGL(mlp_f16x2_forward_kernel<4>, grid, block, 0, stream, m, queries, rows, out); break;
      default: throw std::runtime_error("launch_mlp_forward: unsupported n_out");
    }
    return;
  }
  if (m.tile == 0) {
    const dim3 grid((unsigned)((rows + 63) / 64)), block(256);
    switch (m.out_tiles) {
      case 1: hipLaunchKernelGGL(mlp_tape_forward_kernel<1>, grid, block, 0, stream, m, queries, rows, out); break;
      case 2: hipLaunchKernelGGL(mlp_tape_forward_kernel<2>, grid, block, 0, stream, m, queries, rows, out); break;
      case 3: hipLaunchKernelGGL(mlp_tape_forward_kernel<3>, grid, block, 0, stream, m, queries, rows, out); break;
      case 4: hipLaunchKernelGGL(mlp_tape_forward_kernel<4>, grid, block, 0, stream, m, queries, rows, out); break;
      default: throw std::runtime_error("launch_mlp_forward: unsupported n_out");
    }
    return;
  }
  if (m.tile == 16) return launch_mlp16(m, queries, rows, out, stream);
  const dim3 grid((unsigned)((rows + 31) / 32)), block(64);
  const int NT = m.n_hidden / 32;
#define RBL_LAUNCH(NT_, OT_) \
  hipLaunchKernelGGL((mlp_forward_kernel<NT_, OT_>), grid, block, 0, stream, m, queries, rows, out)
  if (NT == 8 && m.out_tiles == 1) RBL_LAUNCH(8, 1);
  else if (NT == 8 && m.out_tiles == 2) RBL_LAUNCH(8, 2);
  else if (NT == 4 && m.out_tiles == 1) RBL_LAUNCH(4, 1);
  else if (NT == 4 && m.out_tiles == 2) RBL_LAUNCH(4, 2);
  else if (NT == 2 && m.out_tiles == 1) RBL_LAUNCH(2, 1);
  else if (NT == 2 && m.out_tiles == 2) RBL_LAUNCH(2, 2);
  else throw std::runtime_error("launch_mlp_forward: unsupported shape");
#undef RBL_LAUNCH
}

}  // namespace rbl
