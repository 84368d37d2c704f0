// rebel_amd/csrc/net_kernels.hip -- fused value-net forward on CDNA4 matrix cores (gfx950).
//
// The net is the reference's Net2 (cfvpy/models.py:64-94, conf/c02_selfplay/liars_sp.yaml:28-33):
//     n_layers x [Linear -> LayerNorm -> GELU(erf)] -> Linear,   [rows, Q] f32 -> [rows, H] f32.
// The reference evaluates it once per CFR iteration per data-gen thread on ~66 rows (ModelLocker::forward,
// csrc/liars_dice/rela/model_locker.h:85-95); here every pseudo-leaf of every lane goes through ONE launch.
//
// Two kernels: the default persistent register-resident kernel lives in net_resident_kernel.hip (n_layers = 2,
// n_hidden = 256); this file holds the general n_hidden = 256 kernel ("feature split", any n_layers <= 7) that serves
// the shapes the resident kernel does not take, plus the host-side weight packing of both.  Both evaluate every
// product as an f16 x 2 split on v_mfma_f32_16x16x32_f16 (see below).  The earlier f32-MFMA and LDS-tape designs that
// led here are described in HISTORY.md section 3.2; their code was removed in round 2.
#include "net_kernels.h"

#include <cmath>
#include <stdexcept>
#include <type_traits>

namespace rbl {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// experiment: one-time start delay of ~half a tile period for half of the FIRST round of waves
__device__ __forceinline__ void stagger_sleep(bool first_round_odd, int n) {
  if (first_round_odd)
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(127);
}

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

// ================================================================================================ f16 x 2 split
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

// ================================================================================================ feature-split variant
// The round-1 tape kernels streamed the WEIGHTS through LDS and synchronised the workgroup once per 16 KiB chunk (every
// ~24 MFMAs per wave); PMC showed waves parked on those barriers / LDS waits 40 % of the time and every MFMA needing a
// fresh 1 KiB fragment from LDS.  This kernel turns the sharing around:
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
                                                                          int64_t rows, float* __restrict__ out,
                                                                          const long long* __restrict__ range) {
  if (range) {  // device-side row range: the launch is sized for the host's upper bound, surplus workgroups leave
    const long long r0 = range[0], r1 = range[1];
    queries += r0 * m.n_in;
    out += r0 * m.n_out;
    rows = r1 - r0;
    if ((int64_t)blockIdx.x * (WAVES * 8) >= rows) return;
  }
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
  return n_layers >= 1 && n_layers <= 7 && n_in >= 1 && n_in <= 128 && n_hidden == 256 && n_out >= 1 && n_out <= 64;
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
    throw std::runtime_error(
        "value net shape not supported by the MFMA forward (n_hidden = 256, n_layers <= 7, n_in <= 128, n_out <= 64)");
  if (tile != 3 && tile != 5)
    throw std::runtime_error("pack_mlp: kernel variant must be 5 (resident) or 3 (feature split)");
  if (tile == 5 && mlp_resident_supported(n_layers, n_in, n_hidden, n_out))
    return pack_mlp_fsplit(n_layers, n_in, n_hidden, n_out, use_ln, w, b, ln_w, ln_b, w_out, b_out, 8, true);
  return pack_mlp_fsplit(n_layers, n_in, n_hidden, n_out, use_ln, w, b, ln_w, ln_b, w_out, b_out, 8);
}

void launch_mlp_forward(const MlpDev& m, const float* queries, int64_t rows, float* out, hipStream_t stream,
                        const long long* range) {
  if (rows <= 0) return;
  if (m.tile == 5) return launch_mlp_resident(m, queries, rows, out, stream, range);
  if (m.tile != 3) throw std::runtime_error("launch_mlp_forward: unknown kernel variant");
#define RBL_FS(OT_) \
  hipLaunchKernelGGL((mlp_fsplit_forward_kernel<OT_, 8>), dim3((unsigned)((rows + 63) / 64)), dim3(512), 0, stream, m, \
                     queries, rows, out, range)
  switch (m.out_tiles) {
    case 1: RBL_FS(1); break;
    case 2: RBL_FS(2); break;
    case 3: RBL_FS(3); break;
    case 4: RBL_FS(4); break;
    default: throw std::runtime_error("launch_mlp_forward: unsupported n_out");
  }
#undef RBL_FS
}

}  // namespace rbl
