// rebel_amd/csrc/net_resident_kernel.hip -- value-net forward with REGISTER-RESIDENT weights (MlpDev::tile == 5).
//
// Why another variant.  The feature-split kernel (net_kernels.hip, tile 3) re-reads every weight fragment from L2 once
// per 64-row workgroup: 290 KB per 64 rows, 1.2 GB per 270k-row launch, ~65 % of what the L2 -> CU path can deliver while
// the GEMM phases run.  A build with the weight loads hoisted out of the k loop ran the hidden-layer GEMM phase in 4.7 K
// cycles instead of 8.5 K: the phase was bound by weight delivery, not by the matrix pipe.  Two more measurements
// (scripts/micro/) shaped this kernel:
//   * what the matrix pipe shares with the VALU (profiles/r03_micro_mfma32_fillers.txt; MI355X_MICROARCH.md "MFMA / VALU
//     co-issue"): SCALAR f32 VALU ops (v_fma_f32, v_mul, v_min, v_cvt_pkrtz, v_fma_mix) DO overlap f16 MFMAs -- about 5 per
//     32x32x16 MFMA ride for free (32.0 -> 33.5 cycles, part A), and a SIMD running MFMA waves beside VALU waves takes 520
//     cycles where the two alone take 544 + 408 (part C).  PACKED f32 ops do not: one v_pk_fma_f32 per MFMA already costs
//     32 -> 50 cycles, and the GELU below is 7 of them per element pair; v_exp_f32 hides up to 3 per MFMA.  So an
//     interleaved schedule can hide the split / LayerNorm part of the epilogue behind the GEMM but not the GELU polynomial
//     (as scalar FMAs it is twice the instructions and lost in tile 5's VALU-only phases: r03_net_pipe_kernel_measurements).
//     The binding constraint today is neither pipe but the socket: the kernel draws 1 340-1 395 W of the 1 400 W cap, and its
//     dynamic energy per launch / (cap - idle) is already the measured duration (profiles/r04_net_energy_attribution.txt) --
//     a denser interleaving is clocked down by the firmware instead of finishing sooner (profiles/r04_power_trace_*.txt).
//     Hence barrier-separated phases, each at its own pipe's rate, and a short VALU phase;
//   * v_mfma_f32_16x16x32_f16 honours f16 subnormals, so the low halves of the f16x2 split need no 2^11 pre-scale and
//     all three partial products can go into ONE f32 accumulator.
// Hence: a PERSISTENT workgroup per CU (8 waves, 2 per SIMD, 256-VGPR budget), looping over 64-row groups.  Wave w owns
// output features [32w, 32w+32) of both dense layers and keeps its slice of the 256x256 hidden layer (hi + lo halves,
// 128 VGPRs) in registers for the whole launch: weight traffic drops from once per group to once per workgroup (16x fewer
// bytes at 270k rows; SQ_INSTS_VMEM 3.67 M -> 0.48 M per launch).  Layer-0 weights (4 KB per wave) are re-read from L2
// once per group, requested as soon as registers are free; the next group's query rows are fetched in B-fragment order
// while the current group computes.
//
// Epilogue ON THE ACCUMULATORS.  The MFMA's D layout (lane (j, g): features 4g..4g+3 of each of the wave's two tiles,
// row j) is exactly one 16-byte B fragment of k-step `w` of the next layer, provided that layer's weights are packed
// with k running as (tile, g, r) inside a k-step (pack_mlp, tile 5).  So bias, LayerNorm and GELU are applied in
// registers and the result goes straight into the next layer's operand image: no f32 image, no transposition through LDS;
// per layer the waves exchange one float per row and wave (variance partials) and 2 LDS-only barriers.  The output
// layer's first tile is multiplied from registers too (each wave its own k slice; partials summed by four waves).
//
// VALU diet (the epilogue is about half of the time): per element pair, LayerNorm 3 (weights and biases are centred over
// the output features on the host, so only the variance is computed) + GELU 13 (gelu_z below; the 1/sqrt2 going in is
// folded into the LayerNorm scale and shift, the -sqrt2 coming out into the next layer's weights) + split 4 + bias 1 = 21
// instructions, against 37 in the tile-3 kernel.  Measured (MI355X, 270k rows): 18.8 k cycles per 64-row group and CU
// (tile 3: 24.7 k), ~145 us per launch alone; PMC: MFMA busy 33 % + VALU busy 45 % of the time -- by this kernel's barrier-separated phases, not by a hardware rule (above).
//
// Supported shapes: n_hidden == 256 and either n_layers == 2 (one hidden layer, the reference's configuration
// liars_sp.yaml:28-33; n_in <= 128, n_out <= 64) or n_layers == 3 (Net2's class default, cfvpy/models.py:73; n_in <= 64,
// n_out <= 16: every one-die game and 2 dice x 3 faces -- template parameter NH below).  Anything else stays on the tile-3
// kernel.
#include <stdexcept>
#include <type_traits>

#include "launch_timing.h"
#include "net_kernels.h"

namespace rbl {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));  // dword-aligned 16-byte access
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

union Frag {
  f32x4 v;
  f16x8 h;
};

constexpr int kRows = 64, kRT = 4, kKS = 8, kOTW = 2, kWaves = 8;
constexpr int kImageBytes = kKS * 2 * kRT * 64 * 16;  // activations of one group as f16x2 B fragments (64 KB)
constexpr int kStatBytes = kRows * kWaves * 4;        // per-row, per-wave partial sums of squares (LayerNorm)

__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 splat2(float v) { return f32x2{v, v}; }

// -GELU(sqrt2 z) / sqrt2 = t erfc(t) / 2 - max(z, 0), t = min(|z|, 4), for two elements.  t erfc(t) / 2 is evaluated as
// t 2^(t R(t) - 1) with R a degree-5 polynomial fitted (minimax on [0, 4], scripts/fit_gelu.py) to THAT product: max error
// 1.06e-7 in f32 arithmetic, i.e. 1.5e-7 on GELU itself; erfc(4) = 1.5e-8, so the clamp costs nothing.  13 VALU
// instructions per pair: 2 v_min (|z| clamp), 7 v_pk_fma, 2 v_exp, 2 v_min (-relu).
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

// two f32 -> (hi, lo) f16 pairs, lo = the exact remainder (subnormal f16 lo parts are fine: the MFMA honours them)
__device__ __forceinline__ void split2(float a, float b, f16x2* hi, f16x2* lo) {
  const f16x2 h = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(a, b));
  // remainder = a - (f32)hi, rounded to f16 and written into its half of the packed result by ONE mixed-precision fma per
  // element (v_fma_mixlo / mixhi_f16): three instructions per pair instead of four (round 3)
  unsigned l;
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l) : "v"(h), "v"(a));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(h), "v"(b));
  *hi = h;
  *lo = __builtin_bit_cast(f16x2, l);
}

// two f32 -> one f16 pair, round to nearest even (the half_inference modes keep no remainder, so the rounding is the error:
// cvt_pkrtz's truncation would bias it and double its bound)
__device__ __forceinline__ f16x2 pack_rne(float a, float b) { return __builtin_convertvector(f32x2{a, b}, f16x2); }

// sum over the 8 lanes {l ^ 1, l ^ 16, l ^ 32} that share an image row, on the VALU (DPP + the gfx950 row / half swaps)
// instead of three ds_bpermute round trips
__device__ __forceinline__ float row_sum8(float s) {
  s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0xB1, 0xF, 0xF, true));
  // v_permlane16_swap / v_permlane32_swap exchange the odd rows (upper half) of one register with the even rows (lower
  // half) of ANOTHER: copy, swap, add.  Written as asm: hipcc 7.2 mis-models the builtin's second result.
  float c;
  asm("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(s), "=&v"(c));
  s += c;
  asm("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(s), "=&v"(c));
  return s + c;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it would wait for the query
// prefetch of the next group and for the output stores of this one at every phase boundary.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Developer knock-outs for the ENERGY attribution of this kernel (scripts/power_attribution.sh; results are garbage, power and
// time only): RBL_KO_MFMA drops the matrix instructions but keeps their operands alive (weight registers, B fragments from
// LDS); RBL_KO_EPI drops LayerNorm scale / GELU / split (the activations are rounded to f16 and both halves carry the same).
#ifdef RBL_KO_MFMA
__device__ __forceinline__ f32x4 ko_mfma(f16x8 a, f16x8 b, f32x4 c) {
  asm volatile("" ::"v"(a), "v"(b));
  return c;
}
#define RBL_MFMA(A_, B_, C_) ko_mfma((A_), (B_), (C_))
#else
#define RBL_MFMA(A_, B_, C_) __builtin_amdgcn_mfma_f32_16x16x32_f16((A_), (B_), (C_), 0, 0, 0)
#endif

// One dense layer for this wave's 2 x 4 output tiles with the weights already in registers.  A "step" is one (k-step,
// row tile): two B fragments (hi, lo) from LDS feed 6 MFMAs (small products first, all into the same accumulator).
// hipcc on its own emits "4 ds_reads, wait, 12 MFMAs" with nothing in flight across the wait; the sched_barriers pin a
// 3-slot ring instead: the fragments of step s+2 are requested before the MFMAs of step s issue, so every LDS round trip
// has ~200 cycles of matrix work in front of it.
// NRES k-steps with weights in wh / wl, then NTAIL more from th / tl (fragments requested from L2 just before the call:
// they land while the resident k-steps are being multiplied); NT1 = max(NTAIL, 1) is only the array bound
// PROD: f16 products per multiply.  3 = x_h W_l + x_l W_h + x_h W_h (f32 parity, the default); 2 = x_h (W_l + W_h): the
// activations are rounded to f16, the weights keep 22 bits; 1 = x_h W_h: activations and weights rounded to f16, f32
// accumulation (the reference's half_inference, cfvpy/selfplay.py:42-43, with fewer roundings than a half torch module).
// With PROD < 3 the lo fragments of the X image are neither written nor read.
template <int NRES, int NTAIL, int NT1, int PF, int PROD>
__device__ __forceinline__ void gemm_resident(const Frag (&wh)[NRES][kOTW], const Frag (&wl)[NRES][kOTW],
                                              const Frag (&th)[NT1][kOTW], const Frag (&tl)[NT1][kOTW],
                                              const f32x4* __restrict__ X, int lane, f32x4 (&acc)[kOTW][kRT]) {
  constexpr int NKS = NRES + NTAIL;
  constexpr int NS = NKS * kRT;
  Frag xb[PF + 1][2];
#pragma unroll
  for (int s = 0; s < PF && s < NS; ++s) {
    xb[s][0].v = X[(((s / kRT) * 2 + 0) * kRT + (s % kRT)) * 64 + lane];
    if (PROD == 3) xb[s][1].v = X[(((s / kRT) * 2 + 1) * kRT + (s % kRT)) * 64 + lane];
  }
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int ks = s / kRT, rt = s % kRT, slot = s % (PF + 1);
    if (s + PF < NS) {
      const int n = s + PF, nslot = n % (PF + 1);
      xb[nslot][0].v = X[(((n / kRT) * 2 + 0) * kRT + (n % kRT)) * 64 + lane];
      if (PROD == 3) xb[nslot][1].v = X[(((n / kRT) * 2 + 1) * kRT + (n % kRT)) * 64 + lane];
    }
    __builtin_amdgcn_sched_barrier(0);
#ifdef RBL_SPLIT_KS
    if (rt == 0 && ks > 0 && ks % RBL_SPLIT_KS == 0) {
      int z = lane;
      asm volatile("" : "+v"(z));
      if (z == 0x7fffffff) asm volatile("s_nop 0");
    }
#endif
    const bool tail = ks >= NRES;
    const int kr = tail ? 0 : ks, kt = tail ? ks - NRES : 0;
    if (PROD >= 2) {
#pragma unroll
      for (int ot = 0; ot < kOTW; ++ot)
        acc[ot][rt] = RBL_MFMA(tail ? tl[kt][ot].h : wl[kr][ot].h, xb[slot][0].h, acc[ot][rt]);
    }
    if (PROD == 3) {
#pragma unroll
      for (int ot = 0; ot < kOTW; ++ot)
        acc[ot][rt] = RBL_MFMA(tail ? th[kt][ot].h : wh[kr][ot].h, xb[slot][1].h, acc[ot][rt]);
    }
#pragma unroll
    for (int ot = 0; ot < kOTW; ++ot)
      acc[ot][rt] = RBL_MFMA(tail ? th[kt][ot].h : wh[kr][ot].h, xb[slot][0].h, acc[ot][rt]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// The hidden layer when not all of its weights are resident (more than one input chunk).  k-steps are multiplied in the order
// [NEARLY streamed, NRES resident, NTAIL - NEARLY streamed]: the first streamed ones were requested before the layer-0 epilogue
// and are there; once they have been used, `fetch_late` requests the remaining ones INTO THE SAME REGISTERS, and the resident
// k-steps (>= 16 steps of matrix work per wave) cover that round trip.  Streamed weights so never hold more than NEARLY k-steps
// of registers (2 x 16 VGPRs instead of up to 64: what hipcc could not fit and spilled from the resident set).
template <int NRES, int NTAIL, int NEARLY, int NE1, int PF, int PROD, class FetchLate>
__device__ __forceinline__ void gemm_hidden(const Frag (&wh)[NRES][kOTW], const Frag (&wl)[NRES][kOTW], Frag (&th)[NE1][kOTW],
                                            Frag (&tl)[NE1][kOTW], const f32x4* __restrict__ X, int lane,
                                            f32x4 (&acc)[kOTW][kRT], FetchLate&& fetch_late) {
  static_assert(NTAIL - NEARLY <= NEARLY, "late k-steps reuse the early ones' registers");
  constexpr int NKS = NRES + NTAIL, NS = NKS * kRT;
  // position in the multiplication order -> k-step of the X image
  auto ks_of = [](int p) { return p < NEARLY ? NRES + p : (p < NEARLY + NRES ? p - NEARLY : p); };
  Frag xb[PF + 1][2];
#pragma unroll
  for (int s = 0; s < PF && s < NS; ++s) {
    const int k = ks_of(s / kRT);
    xb[s][0].v = X[((k * 2 + 0) * kRT + (s % kRT)) * 64 + lane];
    if (PROD == 3) xb[s][1].v = X[((k * 2 + 1) * kRT + (s % kRT)) * 64 + lane];
  }
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int p = s / kRT, rt = s % kRT, slot = s % (PF + 1);
    if (s + PF < NS) {
      const int n = s + PF, nslot = n % (PF + 1), k = ks_of(n / kRT);
      xb[nslot][0].v = X[((k * 2 + 0) * kRT + (n % kRT)) * 64 + lane];
      if (PROD == 3) xb[nslot][1].v = X[((k * 2 + 1) * kRT + (n % kRT)) * 64 + lane];
    }
    __builtin_amdgcn_sched_barrier(0);
    const bool res = p >= NEARLY && p < NEARLY + NRES;
    const int kr = res ? p - NEARLY : 0;
    const int kt = res ? 0 : (p < NEARLY ? p : p - NRES - NEARLY);  // register slot of a streamed k-step
    if (PROD >= 2) {
#pragma unroll
      for (int ot = 0; ot < kOTW; ++ot)
        acc[ot][rt] = RBL_MFMA(res ? wl[kr][ot].h : tl[kt][ot].h, xb[slot][0].h, acc[ot][rt]);
    }
    if (PROD == 3) {
#pragma unroll
      for (int ot = 0; ot < kOTW; ++ot)
        acc[ot][rt] = RBL_MFMA(res ? wh[kr][ot].h : th[kt][ot].h, xb[slot][1].h, acc[ot][rt]);
    }
#pragma unroll
    for (int ot = 0; ot < kOTW; ++ot)
      acc[ot][rt] = RBL_MFMA(res ? wh[kr][ot].h : th[kt][ot].h, xb[slot][0].h, acc[ot][rt]);
    __builtin_amdgcn_sched_barrier(0);
    if (NTAIL > NEARLY && s == NEARLY * kRT - 1) {
      fetch_late();
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// K0C: 32-wide k chunks of the input layer (1 for n_in <= 32: its weights stay resident); LN: LayerNorm on / off;
// NOTV: 1 = one output tile (16 outputs), multiplied straight from the last epilogue's registers; 3 = two or three tiles, all
// from registers (the third is skipped at run time when m.out_tiles == 2); 4 = four tiles: the first from registers, the rest
// through the X image.  (The X-image loop is compiled into variant 4 only: next to the prefetched layer-0 weights of the next
// group its fragments pushed the K0C = 4 kernel 40 VGPRs over budget, and hipcc spilled RESIDENT hidden weights for it --
// reloaded from scratch, with a full vmcnt wait, inside the hidden GEMM of every group.)
// NH: hidden 256 x 256 layers (n_layers - 1).  1 = the reference's configuration (liars_sp.yaml:28-33); 2 = the class default
// of Net2 (cfvpy/models.py:73, n_layers = 3).  Two hidden layers are 512 KB of hi + lo fragments -- the whole register file of
// a CU -- so each keeps HALF of its k-steps resident (2 x 64 VGPRs) and streams the other four from L2 per group through
// gemm_hidden: two requested before the preceding epilogue (its VALU work covers the round trip), two inside the GEMM behind
// the four resident k-steps.  (First built as "first layer resident, second through a ring of 2-3 k-step slots": 256 KB per CU
// had to arrive inside ONE 4 k-cycle GEMM phase -- exactly the 64 B/clk of the L2 -> CU path -- and that GEMM took 9 k cycles
// whatever the ring depth; split over both GEMMs and both epilogues the same bytes have 4x the time.)
template <int K0C, bool LN, int NOTV, int PROD, int NH>
__global__ void __launch_bounds__(kWaves * 64, 1) mlp_resident_kernel(const MlpDev m, const float* __restrict__ queries,
                                                                    int64_t rows, float* __restrict__ out, int n_groups,
                                                                    const long long* __restrict__ range) {
  constexpr int NOT = NOTV == 4 ? 1 : NOTV;  // tiles multiplied from registers
  const float* q_stat = m.q_stat;
  if (range) {  // device-side row range (resident self-play: the host never learns the row counts of an epoch)
    const long long r0 = range[0], r1 = range[1];
    queries += r0 * (q_stat ? m.q_dyn_stride : m.n_in);
    if (q_stat) q_stat += r0 * m.q_stat_stride;
    out += r0 * m.n_out;
    rows = r1 - r0;
    n_groups = (int)((rows + kRows - 1) / kRows);
  }
  constexpr int kParamFloats = (NH + 1) * 3 * 256 + 64;  // per layer: bias, gamma, beta / sqrt2; then the output bias
  constexpr int kOutBias = (NH + 1) * 3 * 256;
  constexpr int kWoF4 = kKS * 2 * 64;             // one output tile's weight fragments (16 KB)
  // kEarlyStage: the next group's queries are turned into B fragments in a region of their own (Xq) while this group's
  // hidden GEMM drains, so a group starts with its layer-0 GEMM instead of a staging phase and a barrier (round 4: ~830 of
  // 17.7 k cycles per group).  With three or four input chunks the region (24 / 32 KB) and the longer live ranges are not
  // worth it: those shapes stage at the head of the group as before, into the activation image itself.
  // (Tried for four chunks too, staging before the hidden epilogue or right after the layer-0 epilogue: the 16 more live floats
  // cost 20 bytes of scratch per lane in the 2 dice x 6 faces instantiation either way.
  // After the hidden epilogue instead (nothing in flight, accumulators dead) one resident weight fragment is spilled and the
  // stamped group is no shorter: at n_in = 99 the 3.9 k cycles of the head phase are the ISSUE of the next rows' 16 unaligned
  // loads per thread and the staging itself, not a wait -- they move, they do not shrink.)
  constexpr bool kEarlyStage = K0C <= 2;
  constexpr int kQImageBytes = kEarlyStage ? K0C * 2 * kRT * 64 * 16 : 0;
  // kW0Lds (round 5, one input chunk): the layer-0 weights of every wave (4 KB each) are copied to LDS once per launch and read
  // from there per group.  As global loads requested after the last epilogue they shared the vmcnt queue with the output
  // stores issued right behind them, whose number the compiler cannot know: the first MFMA of the next group sat behind
  // `s_waitcnt vmcnt(0)`, i.e. behind the store acknowledgements of the group that had just finished, in every group.
  constexpr bool kW0Lds = K0C == 1;
  constexpr int kW0LdsBytes = kW0Lds ? kWaves * K0C * kOTW * 2 * 64 * 16 : 0;
  // With one output tile its partials (32 KB) live in the activation image, like those of tiles 1 and 2 always did: nobody reads
  // X between the first barrier of the last epilogue and the next group's layer-0 epilogue.  The workgroup then needs 128 KB
  // instead of 148: room for this wave's output-layer fragments in LDS too (kWoLds: the last epilogue has no global load left
  // and never waits on the vmcnt queue, where the next-but-one group's rows are in flight), and 32 KB of a CU stay free for the
  // CFR lanes of the other lane part when two small parts interleave.
  constexpr bool kPinX = NOTV == 1;
  constexpr int kPBytes = kPinX ? 0 : kWaves * kRT * 64 * 16;
  constexpr bool kWoLds = K0C == 1 && NOTV == 1;
  constexpr int kWoLdsBytes = kWoLds ? kWaves * 2 * 64 * 16 : 0;
  __shared__ __align__(16) unsigned char smem[kImageBytes + kStatBytes + kPBytes + kParamFloats * 4 + kQImageBytes + kW0LdsBytes + kWoLdsBytes];
  f32x4* X = reinterpret_cast<f32x4*>(smem);                 // [ks][hi,lo][row tile][lane] B fragments
  unsigned long long* X8 = reinterpret_cast<unsigned long long*>(smem);
  float* S = reinterpret_cast<float*>(smem + kImageBytes);   // [row][wave] sums of squares
  // [wave][row tile][lane] partial outputs of tile 0 (k slice of a wave)
  f32x4* P = kPinX ? X : reinterpret_cast<f32x4*>(S + kRows * kWaves);
  float* prm = reinterpret_cast<float*>(smem + kImageBytes + kStatBytes + kPBytes);  // [layer][bias, gamma, beta'][256], output bias
  f32x4* Xq = kEarlyStage ? reinterpret_cast<f32x4*>(prm + kParamFloats) : X;  // B fragments of the query rows (layer 0's input)
  f32x4* W0s = reinterpret_cast<f32x4*>(smem + kImageBytes + kStatBytes + kPBytes + kParamFloats * 4 + kQImageBytes);
  f32x4* Wos = reinterpret_cast<f32x4*>(smem + kImageBytes + kStatBytes + kPBytes + kParamFloats * 4 + kQImageBytes + kW0LdsBytes);
  unsigned long long* Xq8 = reinterpret_cast<unsigned long long*>(Xq);
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave index in an SGPR
  const int lane0 = tid & 63;
  int lane = lane0, j = lane0 & 15, g = lane0 >> 4;
  const int n_in = m.n_in;
  long long* dbg = m.dbg && blockIdx.x < 1024 ? m.dbg + (size_t)blockIdx.x * 16 : nullptr;
  long long* const dbg_end = dbg;
  int dbg_k = 0;
#ifdef RBL_NO_STAMPS
#define RBL_NSTAMP() __builtin_amdgcn_sched_barrier(0)
#else
#define RBL_NSTAMP()                                              \
  do {                                                            \
    __builtin_amdgcn_sched_barrier(0);                            \
    if (dbg && tid == 0) dbg[dbg_k] = (long long)clock64();       \
    ++dbg_k;                                                      \
    __builtin_amdgcn_sched_barrier(0);                            \
  } while (0)
#endif

  // ---------------------------------------------------------------- this wave's weights, resident for the whole launch
  const f32x4* blob = reinterpret_cast<const f32x4*>(m.tape);
  // With more than one input k chunk (n_in > 32: 2dx3f, 2dx6f) the per-group state no longer fits next to all 8 hidden
  // k-steps (hipcc spilled 4-6 weight fragments and reloaded them from scratch every group); the last kTail k-steps are
  // then streamed from L2 per group instead, requested right before the hidden GEMM and used at its end.
#ifndef RBL_KTAIL4
#define RBL_KTAIL4 4
#endif
#ifndef RBL_KTAIL3
#define RBL_KTAIL3 3
#endif
#ifndef RBL_KEARLY
#define RBL_KEARLY 2
#endif
#ifndef RBL_NH2_TAIL
#define RBL_NH2_TAIL 5
#endif
#ifndef RBL_NH2_TAIL2
#define RBL_NH2_TAIL2 6
#endif
#ifndef RBL_NH2_EARLY
#define RBL_NH2_EARLY 3
#endif
#ifndef RBL_NH2_PF
#define RBL_NH2_PF 2
#endif
  // with two hidden layers each keeps kKS - kTail k-steps resident (see NH above)
  constexpr int kTail1 = K0C == 1 ? 0 : (K0C == 2 ? 1 : (K0C == 3 ? RBL_KTAIL3 : RBL_KTAIL4));
  constexpr int kTailNH2 = K0C == 1 ? RBL_NH2_TAIL : RBL_NH2_TAIL2;
  constexpr int kTail = NH == 2 && kTail1 < kTailNH2 ? kTailNH2 : kTail1, kRes = kKS - kTail,
                kEarlyMax = NH == 2 ? RBL_NH2_EARLY : RBL_KEARLY, kEarly = kTail < kEarlyMax ? kTail : kEarlyMax,
                kE1 = kEarly > 0 ? kEarly : 1;
  // B fragments requested this many steps ahead of their MFMAs; 1 with three or four input chunks (a third ring slot = 8
  // more VGPRs made hipcc spill 36-bytes' worth of resident weights at K0C = 3)
  constexpr int kPF = K0C >= 3 ? 1 : (NH == 2 ? RBL_NH2_PF : 2);
#ifndef RBL_W0_LATE_MIN
#define RBL_W0_LATE_MIN 2  // (round 5: 2 input chunks too -- 19.25 k -> 18.74 k cycles per group at 2 dice x 3 faces; one chunk reads an LDS copy)
#endif
  constexpr bool kW0Late = K0C >= RBL_W0_LATE_MIN;
  Frag w1h[kRes][kOTW], w1l[kRes][kOTW];
  const f32x4* w1 = reinterpret_cast<const f32x4*>(m.wh) + (size_t)wave * kKS * kOTW * 2 * 64;
  {
#pragma unroll
    for (int ks = 0; ks < kRes; ++ks)
#pragma unroll
      for (int ot = 0; ot < kOTW; ++ot) {
        w1h[ks][ot].v = w1[((ks * kOTW + ot) * 2 + 0) * 64 + lane];
        if (PROD >= 2) w1l[ks][ot].v = w1[((ks * kOTW + ot) * 2 + 1) * 64 + lane];
      }
  }
  constexpr int kRes2 = NH == 2 ? kRes : 1;
  Frag w2h[kRes2][kOTW], w2l[kRes2][kOTW];
  const f32x4* w2 = w1 + (size_t)kWaves * kKS * kOTW * 2 * 64;  // [layer][wave][k-step][tile][part]: one layer further
  if constexpr (NH == 2) {
#pragma unroll
    for (int ks = 0; ks < kRes; ++ks)
#pragma unroll
      for (int ot = 0; ot < kOTW; ++ot) {
        w2h[ks][ot].v = w2[((ks * kOTW + ot) * 2 + 0) * 64 + lane];
        if (PROD >= 2) w2l[ks][ot].v = w2[((ks * kOTW + ot) * 2 + 1) * 64 + lane];
      }
  }
  // per-feature parameters and the first output tile's weights: LDS copies (read every group, by every thread)
  for (int i = tid; i < 256; i += kWaves * 64) {
#pragma unroll
    for (int l = 0; l < NH + 1; ++l) {
      // the bias rides in the accumulators' initial value, in the GEMM's own scale (weights are packed times a power of two
      // S = 1 / inv_scale; the scaling back happens once per row, inside the LayerNorm factor)
      prm[(l * 3 + 0) * 256 + i] = K0C <= 2 ? m.bias[l * 256 + i] / m.inv_scale[l] : m.bias[l * 256 + i];
      prm[(l * 3 + 1) * 256 + i] = m.ln_w[l * 256 + i];
      prm[(l * 3 + 2) * 256 + i] = m.ln_b[l * 256 + i];
    }
  }
  if (tid < 64) prm[kOutBias + tid] = tid < m.n_out ? m.b_out[tid] : 0.f;

  // Query rows of a group, fetched straight in B-fragment order: thread (row tile = wave >> 1, half = wave & 1, j, g)
  // reads k = 32 ks + 8 g + 4 half + {0..3} of row 16 rt + j (a row is 4 n_in contiguous bytes; the 8 threads of a
  // row cover it in 16-byte pieces).  The loads for group n+1 are issued right after group n is staged and stay in
  // flight until the next staging step.
  float qn[K0C][4];
  auto fetch_queries = [&](int grp) {
    const int64_t row = (int64_t)grp * kRows + (wave >> 1) * 16 + j;
    const bool ok = grp < n_groups && row < rows;
    if (q_stat) {  // split layout: the virtual row is (dynamic row | static row), both strides multiples of 4 floats, so a
      // thread's four consecutive inputs are one aligned 16-byte load from one of the two
      const int DS = m.q_dyn_stride, SS = m.q_stat_stride;
      const int64_t r = ok ? row : 0;
#pragma unroll
      for (int ks = 0; ks < K0C; ++ks) {
        const int k0 = 32 * ks + 8 * g + 4 * (wave & 1);
        const float* src = k0 < DS ? queries + r * DS + k0 : q_stat + r * SS + (k0 - DS);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (ok && k0 < DS + SS) v = *reinterpret_cast<const f32x4*>(src);
#pragma unroll
        for (int e = 0; e < 4; ++e) qn[ks][e] = v[e];
      }
      return;
    }
    // canonical rows (n_in floats, 4-byte aligned): a thread's four consecutive inputs are ONE dword-aligned 16-byte load
    // when they all exist (gfx950 global loads need dword alignment only); the piece that straddles the end of the row is
    // read element by element.  (16 predicated dword loads per thread at n_in = 99 were ~2 k cycles of a 64-row group.)
    const float* q = queries + (ok ? row : 0) * n_in;
#pragma unroll
    for (int ks = 0; ks < K0C; ++ks) {
      const int k0 = 32 * ks + 8 * g + 4 * (wave & 1);
      if (32 * ks + 32 <= n_in || k0 + 4 <= n_in) {  // first clause: uniform, whole chunk inside the row
        f32x4u v = {0.f, 0.f, 0.f, 0.f};
        if (ok) v = *reinterpret_cast<const f32x4u*>(q + k0);
#pragma unroll
        for (int e = 0; e < 4; ++e) qn[ks][e] = v[e];
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) qn[ks][e] = (ok && k0 + e < n_in) ? q[k0 + e] : 0.f;
      }
    }
  };

  f32x4 acc[kOTW][kRT];
  // With one or two input chunks the accumulators start at the (scaled) bias of the wave's features; with three or four the
  // extra live address at the head of the GEMM pushed hipcc into scratch spills (20 bytes per lane), so there the bias is
  // added in the epilogue as before
  constexpr bool kBiasInAcc = K0C <= 2;
  auto init_acc = [&](const float* pl) {
#pragma unroll
    for (int ot = 0; ot < kOTW; ++ot) {
      f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
      if constexpr (kBiasInAcc) b4 = *reinterpret_cast<const f32x4*>(pl + 32 * wave + 16 * ot + 4 * g);
#pragma unroll
      for (int rt = 0; rt < kRT; ++rt) acc[ot][rt] = b4;
    }
  };

  // bias + LayerNorm + GELU on the accumulators, in place, and straight into the next layer's B fragments.
  // D layout of the MFMA: lane (j = lane & 15, g = lane >> 4) holds, for row tile rt, features 4g .. 4g+3 of each of this
  // wave's two 16-feature tiles: 8 values = exactly one 16-byte B fragment of k-step `wave` of the next layer, provided
  // that layer's weights are packed with k running as (tile, g, r) inside the k-step -- which is what pack_mlp does for
  // tile 5.  So there is no f32 image and no transposition: the only thing the waves exchange per layer is the row
  // variance (one float per row and wave).  The first barrier doubles as "everybody is done reading the old X".
  auto epilogue_regs = [&](auto last_tag, float inv_s, const float* pl) {
    constexpr bool kLast = decltype(last_tag)::value;  // feeds the output layer: tile 0 straight from registers
    // output layer, k-step `wave` of the first NOT tiles: 2 KB per tile from L2 per wave and group, requested here, used at
    // the end
    Frag woh[NOT], wol[NOT];
    if constexpr (kLast) {
#pragma unroll
      for (int t = 0; t < NOT; ++t) {
        const int tt = t < m.out_tiles ? t : 0;
        if constexpr (kWoLds) {
          woh[t].v = Wos[(wave * 2 + 0) * 64 + lane];
          if (PROD >= 2) wol[t].v = Wos[(wave * 2 + 1) * 64 + lane];
        } else {
          woh[t].v = reinterpret_cast<const f32x4*>(m.wo)[((size_t)tt * kKS * 2 + wave * 2 + 0) * 64 + lane];
          if (PROD >= 2) wol[t].v = reinterpret_cast<const f32x4*>(m.wo)[((size_t)tt * kKS * 2 + wave * 2 + 1) * 64 + lane];
        }
      }
    }
    // d = S x (pre-activation): the bias was the accumulators' initial value (init_acc), the 1 / S is part of rs[] below --
    // no per-element instruction is spent on either (round 4: -1 of 21 VALU instructions per element pair)
    f32x4 (&d)[kOTW][kRT] = acc;
    const float post = kBiasInAcc ? inv_s : 1.0f;  // what is left to scale back inside the LayerNorm factor
    if constexpr (!kBiasInAcc) {
#pragma unroll
      for (int ot = 0; ot < kOTW; ++ot) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(pl + 32 * wave + 16 * ot + 4 * g);
#pragma unroll
        for (int rt = 0; rt < kRT; ++rt) d[ot][rt] = acc[ot][rt] * inv_s + b4;
      }
    }
    constexpr float kC = 0.70710678118654752440f;
    float rs[kRT];
    // gamma and beta / sqrt2 of this lane's features: requested BEFORE the LayerNorm barrier (the barrier is a compiler fence
    // for LDS accesses), and d is multiplied by gamma while the wave would otherwise wait for the row statistics: the 16
    // multiplies per layer that used to form gamma x rs AFTER the barrier are off the critical path (same instruction count)
    // (With three or four input chunks the longer live ranges of gamma / beta cost scratch spills: those shapes fetch and
    // multiply after the barrier, as before.)
    constexpr bool kGammaEarly = K0C <= 2;
    f32x4 g4[kOTW], o4[kOTW];
    auto fetch_gamma_beta = [&]() {
#pragma unroll
      for (int ot = 0; ot < kOTW; ++ot) {
        g4[ot] = *reinterpret_cast<const f32x4*>(pl + 256 + 32 * wave + 16 * ot + 4 * g);
        o4[ot] = *reinterpret_cast<const f32x4*>(pl + 512 + 32 * wave + 16 * ot + 4 * g);  // beta / sqrt2
      }
    };
    auto times_gamma = [&]() {
#pragma unroll
      for (int ot = 0; ot < kOTW; ++ot)
#pragma unroll
        for (int rt = 0; rt < kRT; ++rt) d[ot][rt] = d[ot][rt] * g4[ot];
    };
    if constexpr (kGammaEarly) fetch_gamma_beta();
    if constexpr (LN) {
      // weights and biases are centred over the output features on the host: d has zero row mean, only the variance is left
      float q[kRT];
#pragma unroll
      for (int rt = 0; rt < kRT; ++rt) {
        f32x2 q2 = splat2(0.f);
#pragma unroll
        for (int ot = 0; ot < kOTW; ++ot) {
          const f32x2 lo = f32x2{d[ot][rt][0], d[ot][rt][1]}, hi = f32x2{d[ot][rt][2], d[ot][rt][3]};
          q2 = fma2(lo, lo, q2);
          q2 = fma2(hi, hi, q2);
        }
        q[rt] = q2[0] + q2[1];
      }
      // sum over the four lane groups g, all four row tiles at once: each swap + add halves two registers into one.
      // Afterwards lane group g holds the total of row tile {0, 2, 1, 3}[g].
      asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(q[0]), "+v"(q[1]));
      asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(q[2]), "+v"(q[3]));
      float a01 = q[0] + q[1], a23 = q[2] + q[3];
      asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a01), "+v"(a23));
      const int rtp = ((g & 1) << 1) | (g >> 1);
      S[(rtp * 16 + j) * kWaves + wave] = a01 + a23;
      if constexpr (kGammaEarly) times_gamma();
      lds_barrier();
      // A lane needs the factors of four rows (row j of every row tile).  Until round 4 every lane derived all four from the
      // partial sums (2 LDS reads + 11 VALU each, the same 64 results computed by all 64 lanes of all 8 waves); now lane
      // (j, g) derives the ONE of row tile g and the four lane groups exchange theirs with three lane swaps: swap16 of two
      // copies leaves [v0, v0, v2, v2] / [v1, v1, v3, v3] in the rows of the two registers, swap32 of each with a copy of
      // itself [v0 x4], [v2 x4] and [v1 x4], [v3 x4] (semantics: the variance reduction above).
      {
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(&S[(g * 16 + j) * kWaves]);
        const f32x4 s1 = *reinterpret_cast<const f32x4*>(&S[(g * 16 + j) * kWaves + 4]);
        const f32x4 t = s0 + s1;
        const float var = ((t[0] + t[1]) + (t[2] + t[3])) * (post * post * (1.0f / 256.0f)) + m.ln_eps;
        // v_rsq_f32 is accurate to 1 ulp: 6e-8 relative on the LayerNorm factor, far inside the f16x2 split's 4e-7 (the
        // Newton step of rounds 1-3 was 4 more instructions per row tile)
        const float mine = (kC * post) * __builtin_amdgcn_rsqf(var);
        float e0, e1, e2, e3;
        asm("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %4\n\ts_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1\n\t"
            "v_mov_b32 %2, %0\n\tv_mov_b32 %3, %1\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %2\n\t"
            "v_permlane32_swap_b32 %1, %3\n\ts_nop 1"
            : "=&v"(e0), "=&v"(e1), "=&v"(e2), "=&v"(e3)
            : "v"(mine));
        rs[0] = e0;
        rs[1] = e1;
        rs[2] = e2;
        rs[3] = e3;
      }
    } else {
      if constexpr (kGammaEarly) times_gamma();
      lds_barrier();
#pragma unroll
      for (int rt = 0; rt < kRT; ++rt) rs[rt] = kC * post;
    }
    if constexpr (!kGammaEarly) fetch_gamma_beta();
#pragma unroll
    for (int rt = 0; rt < kRT; ++rt) {
      f16x2 h[4], l[4];
#pragma unroll
      for (int ot = 0; ot < kOTW; ++ot)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
#ifdef RBL_KO_EPI
          h[2 * ot + h2] = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(d[ot][rt][2 * h2] * rs[rt], d[ot][rt][2 * h2 + 1]));
          l[2 * ot + h2] = h[2 * ot + h2];
#else
          const f32x2 a = kGammaEarly ? splat2(rs[rt]) : f32x2{g4[ot][2 * h2], g4[ot][2 * h2 + 1]} * splat2(rs[rt]);
          const f32x2 y = gelu_z(fma2(f32x2{d[ot][rt][2 * h2], d[ot][rt][2 * h2 + 1]}, a,
                                      f32x2{o4[ot][2 * h2], o4[ot][2 * h2 + 1]}));
          if (PROD == 3) {
            split2(y[0], y[1], &h[2 * ot + h2], &l[2 * ot + h2]);
          } else {
            h[2 * ot + h2] = pack_rne(y[0], y[1]);
            l[2 * ot + h2] = f16x2{(_Float16)0.f, (_Float16)0.f};
          }
#endif
        }
      Frag fh, fl;
      fh.h = f16x8{h[0][0], h[0][1], h[1][0], h[1][1], h[2][0], h[2][1], h[3][0], h[3][1]};
      fl.h = f16x8{l[0][0], l[0][1], l[1][0], l[1][1], l[2][0], l[2][1], l[3][0], l[3][1]};
      if (!kLast || NOTV == 4) {
        X[((wave * 2 + 0) * kRT + rt) * 64 + lane] = fh.v;
        if (PROD == 3) X[((wave * 2 + 1) * kRT + rt) * 64 + lane] = fl.v;
      }
      if constexpr (kLast) {  // this wave's 32-wide k slice of the output layer, summed over the waves below
#pragma unroll
        for (int t = 0; t < NOT; ++t) {
          if (t > 0 && t >= m.out_tiles) break;
          f32x4 o = {0.f, 0.f, 0.f, 0.f};
          if (PROD >= 2) o = RBL_MFMA(wol[t].h, fh.h, o);
          if (PROD == 3) o = RBL_MFMA(woh[t].h, fl.h, o);
          o = RBL_MFMA(woh[t].h, fh.h, o);
          // partials of tile 0 have their own buffer; those of tiles 1, 2 take over the X image, which nobody reads between
          // the first barrier of this epilogue (every wave is past the hidden GEMM) and the next group's staging
          f32x4* pt = t == 0 ? P : X + (size_t)(t - 1) * kWaves * kRT * 64;
          pt[(wave * kRT + rt) * 64 + lane] = o;
        }
      }
    }
    lds_barrier();
  };

  // Layer-0 weights of this wave (4 KB per 32-wide k chunk) are NOT kept across the epilogues: next to the hidden layer's
  // 128 VGPRs they pushed the kernel into scratch spills.  They are re-read from L2 once per group, requested as soon as the
  // registers are free again (after the last epilogue of the previous group), so the round trip is off the critical path.
  Frag th[K0C][kOTW], tl[K0C][kOTW];
  auto fetch_w0 = [&]() {
    // a fresh load every group (the values are loop invariant; hoisting = residency).  The opaque value is an OFFSET, not the
    // pointer: a pointer that went through an asm operand comes back in the generic address space, and hipcc then emits
    // FLAT loads -- which count on lgkmcnt as well as vmcnt, so every LDS wait and LDS barrier that followed (the output
    // reduction, the next GEMM's fragment ring) also waited for these L2 round trips (rounds 2-4 shipped that).
    int fresh = 0;
    asm volatile("" : "+s"(fresh));
    if constexpr (kW0Lds) {  // this wave's copy in LDS (written below, before the first call)
      const f32x4* w0 = W0s + wave * (K0C * kOTW * 2 * 64) + fresh;
#pragma unroll
      for (int ks = 0; ks < K0C; ++ks)
#pragma unroll
        for (int ot = 0; ot < kOTW; ++ot) {
          th[ks][ot].v = w0[((ks * kOTW + ot) * 2 + 0) * 64 + lane];
          if (PROD >= 2) tl[ks][ot].v = w0[((ks * kOTW + ot) * 2 + 1) * 64 + lane];
        }
    } else {
      const f32x4* w0 = blob + (size_t)wave * K0C * kOTW * 2 * 64 + fresh;
#pragma unroll
      for (int ks = 0; ks < K0C; ++ks)
#pragma unroll
        for (int ot = 0; ot < kOTW; ++ot) {
          th[ks][ot].v = w0[((ks * kOTW + ot) * 2 + 0) * 64 + lane];
          if (PROD >= 2) tl[ks][ot].v = w0[((ks * kOTW + ot) * 2 + 1) * 64 + lane];
        }
    }
  };
  if constexpr (kWoLds) {  // k-step `wave` of output tile 0, hi and lo
#pragma unroll
    for (int part = 0; part < 2; ++part)
      Wos[(wave * 2 + part) * 64 + lane] = reinterpret_cast<const f32x4*>(m.wo)[((size_t)wave * 2 + part) * 64 + lane];
  }
  if constexpr (kW0Lds) {
    const f32x4* w0g = blob + (size_t)wave * K0C * kOTW * 2 * 64;
    f32x4* w0l = W0s + wave * (K0C * kOTW * 2 * 64);
#pragma unroll
    for (int i = 0; i < K0C * kOTW * 2; ++i) w0l[i * 64 + lane] = w0g[i * 64 + lane];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }
  fetch_w0();
  fetch_queries(blockIdx.x);
  __syncthreads();  // parameter copies visible

  const int stamp_group = m.stagger;  // developer aid (RBL_MLP_STAGGER): which of the workgroup's groups the stamps describe
  int group_no = 0;
  for (int grp = blockIdx.x; grp < n_groups; grp += gridDim.x, ++group_no) {
    const int64_t row0 = (int64_t)grp * kRows;
    dbg = group_no == stamp_group ? dbg_end : nullptr;  // (0: the first group, which includes the cold start)
    dbg_k = 0;
    // lane-derived indices are re-derived every group: hoisted out of the loop, the dozens of LDS / global addresses built
    // from them stay live across all phases and push the 256-VGPR budget (128 of it weights) into scratch spills
    lane = lane0;
    asm volatile("" : "+v"(lane));
    j = lane & 15;
    g = lane >> 4;
    RBL_NSTAMP();  // 0
    // -------------------------------------------------------------- fetched queries -> f16x2 B fragments (8 bytes each)
    auto stage_queries = [&]() {
      const int rt = wave >> 1, half = wave & 1;
#pragma unroll
      for (int ks = 0; ks < K0C; ++ks) {
        f16x2 h0, l0, h1, l1;
        if (PROD == 3) {
          split2(qn[ks][0], qn[ks][1], &h0, &l0);
          split2(qn[ks][2], qn[ks][3], &h1, &l1);
        } else {  // the query, too, is rounded to f16 (the half module's first op: query.half())
          h0 = pack_rne(qn[ks][0], qn[ks][1]);
          h1 = pack_rne(qn[ks][2], qn[ks][3]);
          l0 = l1 = f16x2{(_Float16)0.f, (_Float16)0.f};
        }
        const f16x4 hh = f16x4{h0[0], h0[1], h1[0], h1[1]}, ll = f16x4{l0[0], l0[1], l1[0], l1[1]};
        Xq8[(((ks * 2 + 0) * kRT + rt) * 64 + lane) * 2 + half] = __builtin_bit_cast(unsigned long long, hh);
        if (PROD == 3) Xq8[(((ks * 2 + 1) * kRT + rt) * 64 + lane) * 2 + half] = __builtin_bit_cast(unsigned long long, ll);
      }
    };
    if (!kEarlyStage || group_no == 0) {
      stage_queries();
      lds_barrier();
      fetch_queries(grp + gridDim.x);  // in flight until the next group is staged
    }
    RBL_NSTAMP();  // 1: staged

    // -------------------------------------------------------------- layer 0
    init_acc(prm);
    gemm_resident<K0C, 0, K0C, kPF, PROD>(th, tl, th, tl, Xq, lane, acc);
    RBL_NSTAMP();  // 2: L0 gemm
    // The streamed k-steps of the hidden layer take over registers of the layer-0 weights (dead from here on): the first kEarly
    // of them are requested BEFORE the layer-0 epilogue, whose ~6 k cycles cover the L2 round trips; the rest is requested
    // inside the hidden GEMM into the same registers (gemm_hidden).  History: all requested after the epilogue, the hidden GEMM
    // waited for them (8.8 k cycles instead of 3.8 k at 2 dice x 6 faces).
    Frag t7h[kE1][kOTW], t7l[kE1][kOTW];
    int fresh_t = 0;
    asm volatile("" : "+s"(fresh_t));  // a fresh load every group (an opaque offset: see fetch_w0)
    const f32x4* wt = w1 + (size_t)kRes * kOTW * 2 * 64 + fresh_t;
    auto fetch_tail = [&](const f32x4* base, auto lo_tag, auto hi_tag, auto slot0_tag) {  // streamed k-steps [lo, hi) -> slots slot0 ..
#pragma unroll
      for (int ks = decltype(lo_tag)::value; ks < decltype(hi_tag)::value; ++ks)
#pragma unroll
        for (int ot = 0; ot < kOTW; ++ot) {
          const int sl = ks - decltype(lo_tag)::value + decltype(slot0_tag)::value;
          t7h[sl][ot].v = base[((ks * kOTW + ot) * 2 + 0) * 64 + lane];
          if (PROD >= 2) t7l[sl][ot].v = base[((ks * kOTW + ot) * 2 + 1) * 64 + lane];
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using IE = std::integral_constant<int, kEarly>;
    using IT = std::integral_constant<int, kTail>;
    fetch_tail(wt, I0{}, IE{}, I0{});
    RBL_NSTAMP();  // 3
    epilogue_regs(std::false_type{}, m.inv_scale[0], prm);
    RBL_NSTAMP();  // 4: L0 epilogue

    // -------------------------------------------------------------- hidden layer, weights from registers
    init_acc(prm + 768);
    if constexpr (kTail == 0)
      gemm_resident<kRes, 0, 1, kPF, PROD>(w1h, w1l, t7h, t7l, X, lane, acc);
    else
      gemm_hidden<kRes, kTail, kEarly, kE1, kPF, PROD>(w1h, w1l, t7h, t7l, X, lane, acc, [&]() { fetch_tail(wt, IE{}, IT{}, I0{}); });
    RBL_NSTAMP();  // 5: hidden gemm
    if constexpr (NH == 2) {
      // ------------------------------------------------------------ second hidden layer: the same half-resident scheme
      const f32x4* wt2 = w2 + (size_t)kRes * kOTW * 2 * 64 + fresh_t;
      fetch_tail(wt2, I0{}, IE{}, I0{});  // into the registers the first hidden layer's streamed k-steps just left
      epilogue_regs(std::false_type{}, m.inv_scale[1], prm + 768);
      init_acc(prm + 2 * 768);
      gemm_hidden<kRes, kTail, kEarly, kE1, kPF, PROD>(w2h, w2l, t7h, t7l, X, lane, acc, [&]() { fetch_tail(wt2, IE{}, IT{}, I0{}); });
    }
    if (kEarlyStage) {
      // the next group's queries (requested a whole group ago) become B fragments now: Xq was last read by this group's layer-0
      // GEMM, two barriers back, and the two barriers of the epilogue below order these stores before the next group's
      // layer-0 GEMM reads them.  The older wave of a SIMD gets here ~2.5 k cycles before its partner: this is idle time
      stage_queries();
      fetch_queries(grp + 2 * (int)gridDim.x);
    }
    RBL_NSTAMP();  // 6
    epilogue_regs(std::true_type{}, m.inv_scale[NH], prm + NH * 768);
    RBL_NSTAMP();  // 7: hidden epilogue (+ output tile 0 partials)
    // with three or four input chunks (48 / 64 KB per CU) the request goes out AFTER the output stores below: VMEM issues in
    // order, and behind 64 KB of weight loads the stores (and the waves issuing them) waited ~2 k cycles
    if constexpr (!kW0Late)
      if (grp + (int)gridDim.x < n_groups) fetch_w0();

    // -------------------------------------------------------------- register tiles: sum the 8 k slices of (tile, row tile)
    {
      const int n_reg = m.out_tiles < NOT ? m.out_tiles : NOT;
      for (int p = wave; p < n_reg * kRT; p += kWaves) {
        const int t = p >> 2, rt = p & 3;
        const f32x4* pt = t == 0 ? P : X + (size_t)(t - 1) * kWaves * kRT * 64;
        f32x4 o = pt[(0 * kRT + rt) * 64 + lane];
#pragma unroll
        for (int w = 1; w < kWaves; ++w) o += pt[(w * kRT + rt) * 64 + lane];
        // 32-bit lane offsets from a scalar group base (64-bit per-lane row indices were being spilled)
        int r_in = rt * 16 + j, col = t * 16 + 4 * g;
        const int n_out = m.n_out;
        asm volatile("" : "+v"(r_in), "+v"(col));  // computed here, every group: hoisted copies cost spills
        const int rows_here = (int)(rows - row0 < kRows ? rows - row0 : kRows);
        float* og = out + row0 * n_out;
        const f32x4 r4 = o * m.inv_scale[NH + 1] + *reinterpret_cast<const f32x4*>(prm + kOutBias + col);
        if (r_in < rows_here) {
          if ((n_out & 1) == 0) {  // even row length: a lane's four columns are two 8-byte aligned pairs (half the stores)
            f32x2* p2 = reinterpret_cast<f32x2*>(og + r_in * n_out + col);
            if (col + 1 < n_out) p2[0] = f32x2{r4[0], r4[1]};
            if (col + 3 < n_out) p2[1] = f32x2{r4[2], r4[3]};
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (col + r < n_out) og[r_in * n_out + col + r] = r4[r];
          }
        }
      }
    }
    // -------------------------------------------------------------- further output tiles (n_out > 16): from the X image
    if constexpr (NOTV == 4)
    for (int ot = NOT; ot < m.out_tiles; ++ot) {
      const int rt = wave & 3, kh = wave >> 2;
      const f32x4* wo = reinterpret_cast<const f32x4*>(m.wo) + (size_t)ot * kWoF4;
      f32x4 a1 = {0.f, 0.f, 0.f, 0.f}, a2 = a1, a3 = a1;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const int ks = kh * 4 + s4;
        Frag wh, wl, xh, xl;
        wh.v = wo[(ks * 2 + 0) * 64 + lane];
        xh.v = X[((ks * 2 + 0) * kRT + rt) * 64 + lane];
        if (PROD >= 2) {
          wl.v = wo[(ks * 2 + 1) * 64 + lane];
          a2 = RBL_MFMA(wl.h, xh.h, a2);
        }
        if (PROD == 3) {
          xl.v = X[((ks * 2 + 1) * kRT + rt) * 64 + lane];
          a3 = RBL_MFMA(wh.h, xl.h, a3);
        }
        a1 = RBL_MFMA(wh.h, xh.h, a1);
      }
      const f32x4 o = a1 + (a2 + a3);
      lds_barrier();  // the partials of the previous tile have been consumed
      if (kh == 1) P[rt * 64 + lane] = o;
      lds_barrier();
      if (kh == 0) {
        int r_in = rt * 16 + j, col = 4 * g;
        const int n_out = m.n_out;
        asm volatile("" : "+v"(r_in), "+v"(col));
        const int rows_here = (int)(rows - row0 < kRows ? rows - row0 : kRows);
        float* og = out + row0 * n_out;
        const f32x4 r4 = (o + P[rt * 64 + lane]) * m.inv_scale[NH + 1] + *reinterpret_cast<const f32x4*>(prm + kOutBias + ot * 16 + col);
        if (r_in < rows_here) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = ot * 16 + col + r;
            if (i < n_out) og[r_in * n_out + i] = r4[r];
          }
        }
      }
    }
    if (NOTV > 1 && m.out_tiles > 1) lds_barrier();  // X and P change hands
    RBL_NSTAMP();  // 8: output layer
    if constexpr (kW0Late)
      if (grp + (int)gridDim.x < n_groups) fetch_w0();
  }
  if (dbg_end && tid == 0) dbg_end[12] = (long long)clock64();  // whole workgroup: (this - stamp 0) / groups = steady state
#undef RBL_NSTAMP
}

}  // namespace

bool mlp_resident_supported(int n_layers, int n_in, int n_hidden, int n_out) {
  if (n_hidden != 256 || n_in < 1 || n_out < 1) return false;
  if (n_layers == 2) return n_in <= 128 && n_out <= 64;
  return n_layers == 3 && n_in <= 64 && n_out <= 16;  // two hidden layers: the second one streams (gemm_ring)
}

void launch_mlp_resident(const MlpDev& m, const float* queries, int64_t rows, float* out, hipStream_t stream,
                         const long long* range) {
  static int n_cu[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) throw std::runtime_error("launch_mlp_resident: no current device");
  if (dev < 64 && n_cu[dev] == 0) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    n_cu[dev] = v;
  }
  const int cus = dev < 64 ? n_cu[dev] : 256;
  const int n_groups = (int)((rows + kRows - 1) / kRows);
  const int cap = m.grid_cap > 0 && m.grid_cap < cus ? m.grid_cap : cus;
  const int grid = n_groups < cap ? n_groups : cap;
#define RBL_RES4(K0C_, LN_, NOT_, PROD_, NH_)                                                                                  \
  RBL_LAUNCH_TIMED((mlp_resident_kernel<K0C_, LN_, NOT_, PROD_, NH_>), dim3(grid), dim3(kWaves * 64), 0, stream, m, queries, \
                   rows, out, n_groups, range)
  // two hidden layers (n_layers = 3): built for one or two input chunks (every one-die game and 2 dice x 3 faces)
#define RBL_RES3(K0C_, LN_, NOT_, PROD_)                                                                      \
  do {                                                                                                        \
    if (m.n_layers == 2) RBL_RES4(K0C_, LN_, NOT_, PROD_, 1);                                                 \
    else if constexpr (K0C_ <= 2 && NOT_ == 1) RBL_RES4(K0C_, LN_, NOT_, PROD_, 2);                           \
    else throw std::runtime_error("launch_mlp_resident: n_layers = 3 needs n_in <= 64 and n_out <= 16");      \
  } while (0)
  // the half_inference modes (MlpDev::products 2 / 1) are built for LayerNorm nets only (the reference's configuration)
#define RBL_RES2(K0C_, LN_, NOT_)                                             \
  do {                                                                        \
    if (m.products == 3) RBL_RES3(K0C_, LN_, NOT_, 3);                        \
    else if (LN_ && m.products == 2) RBL_RES3(K0C_, true, NOT_, 2);           \
    else if (LN_ && m.products == 1) RBL_RES3(K0C_, true, NOT_, 1);           \
    else throw std::runtime_error("launch_mlp_resident: unsupported products / LayerNorm combination"); \
  } while (0)
#define RBL_RES(K0C_)                                    \
  do {                                                   \
    if (m.use_ln) {                                      \
      if (variant == 1) RBL_RES2(K0C_, true, 1);         \
      else if (variant == 3) RBL_RES2(K0C_, true, 3);    \
      else RBL_RES2(K0C_, true, 4);                      \
    } else {                                             \
      if (variant == 1) RBL_RES2(K0C_, false, 1);        \
      else if (variant == 3) RBL_RES2(K0C_, false, 3);   \
      else RBL_RES2(K0C_, false, 4);                     \
    }                                                    \
  } while (0)
  const int variant = m.out_tiles == 1 ? 1 : (m.out_tiles <= 3 ? 3 : 4);
  if (m.out_tiles < 1 || m.out_tiles > 4) throw std::runtime_error("launch_mlp_resident: unsupported n_out");
  if (m.n_layers != 2 && m.n_layers != 3) throw std::runtime_error("launch_mlp_resident: unsupported n_layers");
  switch (m.l0_chunks) {
    case 1: RBL_RES(1); break;
    case 2: RBL_RES(2); break;
    case 3: RBL_RES(3); break;
    default: RBL_RES(4); break;
  }
#undef RBL_RES
#undef RBL_RES2
#undef RBL_RES3
#undef RBL_RES4
}

}  // namespace rbl
