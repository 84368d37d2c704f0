// rebel_amd/csrc/net_kernels.h -- launch interface of the fused value-net forward (net_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

namespace rbl {

// Device-resident, MFMA-ready copy of a Net2 (cfvpy/models.py:64-94).  Built by pack_mlp() from torch-layout weights.
struct MlpDev {
  int n_layers = 0, n_in = 0, n_hidden = 0, n_out = 0, use_ln = 0;
  int tile = 0;       // kernel variant: 5 = register-resident, 3 = feature split
  const float* tape = nullptr;  // variant 0: the packed weights in consumption order, 32 KiB chunks
  int tape_chunks = 0, l0_chunks = 0;
  float inv_scale[8] = {1, 1, 1, 1, 1, 1, 1, 1};  // variant 2: 1 / (power-of-two weight scale) per layer (last = output)
  int stagger = 0;
  long long* dbg = nullptr;  // developer aid: [1024][16] phase stamps of the first 1024 workgroups (fsplit kernel)    // static wave priority by hardware slot parity (see stagger_priority)
  int k0_steps = 0;   // layer-0 k-pairs, padded to a multiple of 4
  int out_tiles = 0;  // ceil(n_out / 32)
  float ln_eps = 1e-5f;
  const float* w0 = nullptr;     // [NT][k0_steps/4][64][4]
  const float* wh = nullptr;     // [n_layers-1][NT][NT][4][64][4]
  const float* wo = nullptr;     // [out_tiles][NT][4][64][4]
  const float* bias = nullptr;   // [n_layers][n_hidden]
  const float* ln_w = nullptr;   // [n_layers][n_hidden]
  const float* ln_b = nullptr;   // [n_layers][n_hidden]
  const float* b_out = nullptr;  // [out_tiles*32]
  // split query layout (tile 5 only; engine.hip): `queries` of a launch are the dynamic rows [rows][q_dyn_stride], q_stat the
  // static rows [rows][q_stat_stride]; layer 0 is packed for the virtual input row (dyn row | stat row), n_in = the two strides
  const float* q_stat = nullptr;
  int q_dyn_stride = 0, q_stat_stride = 0;
  // f16 products per multiply (tile 5): 3 = f16x2 split on both operands (f32 parity), 2 = activations rounded to f16,
  // 1 = activations and weights rounded to f16 (half_inference); see net_resident_kernel.hip gemm_resident
  int products = 3;
  // tile 5: at most this many persistent workgroups (0 = one per CU).  The engine leaves a quarter of the CUs to the OTHER lane
  // part's CFR kernel when two small parts interleave on two streams (engine.hip: net_grid_cap)
  int grid_cap = 0;
};

// Host-side packing: returns one float blob plus the offsets of the members above (in floats).
struct MlpPacked {
  std::vector<float> blob;
  size_t off_w0, off_wh, off_wo, off_bias, off_lnw, off_lnb, off_bout;
  int k0_steps, out_tiles, tile;
  int tape_chunks = 0, l0_chunks = 0;
  std::vector<float> inv_scale;
};
MlpPacked pack_mlp(int n_layers, int n_in, int n_hidden, int n_out, int use_ln, const float* const* w,
                   const float* const* b, const float* const* ln_w, const float* const* ln_b, const float* w_out,
                   const float* b_out, int tile);

bool mlp_supported(int n_layers, int n_in, int n_hidden, int n_out);

// net_resident_kernel.hip (tile 5): persistent workgroups with register-resident weights; one hidden layer of 256 only
bool mlp_resident_supported(int n_layers, int n_in, int n_hidden, int n_out);
void launch_mlp_resident(const MlpDev& m, const float* queries, int64_t rows, float* out, hipStream_t stream,
                         const long long* range = nullptr);

// out[rows][n_out] = net(queries[rows][n_in]); f16x2-split MFMA (v_mfma_f32_16x16x32_f16), async on `stream`.
// `range` (optional, device memory): the rows to process are [range[0], range[1]) of queries / out, known only on the
// device; `rows` is then the host-side upper bound used to size the launch.
void launch_mlp_forward(const MlpDev& m, const float* queries, int64_t rows, float* out, hipStream_t stream,
                        const long long* range = nullptr);

}  // namespace rbl
