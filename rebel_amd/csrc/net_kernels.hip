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
template <int NT>
__device__ __forceinline__ void epilogue16(f32x4 (&acc)[NT], const float* __restrict__ bias,
                                           const float* __restrict__ ln_w, const float* __restrict__ ln_b, int use_ln,
                                           float eps, int g) {
  constexpr float inv_n = 1.0f / (16 * NT);
#pragma unroll
  for (int it = 0; it < NT; ++it) {
    const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias + it * 16 + 4 * g);
    acc[it] += b4;
  }
  if (use_ln) {
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < NT; ++it) s += (acc[it][0] + acc[it][1]) + (acc[it][2] + acc[it][3]);
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    const float mean = s * inv_n;
    float vs = 0.f;
#pragma unroll
    for (int it = 0; it < NT; ++it)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = acc[it][r] - mean;
        vs = fmaf(d, d, vs);
      }
    vs += __shfl_xor(vs, 16);
    vs += __shfl_xor(vs, 32);
    const float rstd = 1.0f / sqrtf(vs * inv_n + eps);
#pragma unroll
    for (int it = 0; it < NT; ++it) {
      const f32x4 g4 = *reinterpret_cast<const f32x4*>(ln_w + it * 16 + 4 * g);
      const f32x4 o4 = *reinterpret_cast<const f32x4*>(ln_b + it * 16 + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[it][r] = (acc[it][r] - mean) * rstd * g4[r] + o4[r];
    }
  }
#pragma unroll
  for (int it = 0; it < NT; ++it)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[it][r] = gelu_erf(acc[it][r]);
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

MlpPacked pack_mlp(int n_layers, int n_in, int n_hidden, int n_out, int use_ln, const float* const* w,
                   const float* const* b, const float* const* ln_w, const float* const* ln_b, const float* w_out,
                   const float* b_out, int tile) {
  if (!mlp_supported(n_layers, n_in, n_hidden, n_out))
    throw std::runtime_error("value net shape not supported by the MFMA forward (n_hidden in {64,128,256}, n_out <= 64)");
  if (tile == 16) return pack_mlp16(n_layers, n_in, n_hidden, n_out, use_ln, w, b, ln_w, ln_b, w_out, b_out);
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
