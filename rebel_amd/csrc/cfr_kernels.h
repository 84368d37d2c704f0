// rebel_amd/csrc/cfr_kernels.h -- launch interface of the batched CFR step kernel (cfr_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "tables.h"

namespace rbl {

enum CfrMode : int {
  kModeInit = 0,     // build_solver: uniform sigma, zero regrets, reach-weighted sum; then write queries
  kModeStep = 1,     // consume leaf values of the pending queries, CFR::step, then write the next queries
  kModeQueries = 2,  // only (re)write queries for `next_trav` from the current sigma
  kModeBestResponse = 3,  // BRSolver::compute_br (subgame_solving.cc:316-358) against sigma; root values -> br_out
  kModeFpStep = 4,        // FP::step (subgame_solving.cc:433-476): sigma holds the AVERAGE strategy, regrets hold `last`
  kModeEvaluate = 5,      // compute_ev (subgame_solving.cc:931-973): value of following sigma for the traverser -> br_out
};

// Everything the kernel needs; passed by value (fits the kernarg segment).
struct CfrArgs {
  // ---- static tables (per engine)
  const ShapeDev* shapes;
  const int* parent;
  const int* act;
  const int* cb;
  const int* ce;
  const int* depth;
  const int* leaves;
  const int* terms;
  const int* irank;     // node -> index of its reach row (root / nodes with children), -1 otherwise (cfr_rows_kernel)
  const int* leaf_row;  // node -> net row within the lane for pseudo-leaves, -1 otherwise (cfr_rows_kernel)
  const int* vrow;      // node -> rank among the nodes that are not pseudo-leaves, -1 for those (cfr_flat_kernel)
  const int* pack;      // node -> packed rows of its parent (tables.h: ShapeTables::pack; cfr_flat_kernel)
  const int8_t* matches;  // [faces][H]  Game::num_matches (liars_dice.h:83-91)
  const int8_t* wave_tabs;  // cfr_wave_kernel: per shape parent | act | cb | ce | depth | irank (N bytes each) | leaf nodes (L) |
  const int* wave_tab_off;  //                  terminal nodes (T) as one 4-byte aligned blob; byte offset of each shape's blob
  const unsigned short* wave_epv;  // cfr_wave_kernel: per shape, per edge element (c - 1) * H + h: parent(c) * H + h
  const int* wave_epv_off;         //                  element offset of each shape's table
  int H, A, Q, faces, dice;
  int Emax, Nmax;         // per-lane strides: Emax*H reals per strategy array
  // ---- per-lane descriptors
  const int* lane_shape;
  const int* lane_root_player;
  const int* lane_row_off;   // first net row of the lane
  const int* lane_act_iter;  // snapshot when steps_after == act_iter (may be null)
  const double* beliefs;     // [B][2][H]
  // ---- per-lane state, edge-indexed [B][Emax*H]
  double* sigma;     // last_strategies
  double* regrets;
  double* sums;      // sum_strategies
  double* snapshot;  // sigma at act_iteration
  double* root_mean; // [B][2][H] root_values_means
  // ---- net exchange
  float* queries;       // [rows][Q]
  const float* values;  // [rows][H]
  // split query layout (cfr_wave_kernel + the fused MLP forward): what changes between iterations -- traverser flag, the two
  // normalised reach vectors -- as contiguous rows [rows][q_dyn_stride] (stride a multiple of 4 floats, pads written as 0), so
  // that a step writes whole lines instead of 52-byte runs inside 108-byte rows; null = write `queries`
  float* q_dyn;
  int q_dyn_stride;
  // ---- global scratch for lanes too big for LDS: [B][work_reals]
  double* scratch;
  size_t work_stride;
  int use_lds;
  // ---- uniform step parameters (lanes are in lock-step)
  int lane0;  // first lane of this launch (half-batches run on separate streams)
  // cfr_rows_kernel<GS> only: workgroup b serves lane lane_order[lane0 + b] (lanes of a part sorted by tree size, so that a
  // launch can request the LDS of ITS largest tree and small trees share a CU); null = lane0 + b
  const int* lane_order;
  int mode, trav, next_trav, steps_after;
  double alpha;            // root-mean step size (subgame_solving.cc:580-590)
  double pos, neg, strat;  // discounts (:592-617); kModeFpStep: strat = linear factor (n+1)/(n+2) or 1
  int optimistic;          // FP only (util.h:50-60)
  double* br_out;          // [B][H] best-response root values (kModeBestResponse)
  long long* dbg;          // optional [B][16] phase timestamps (s_memtime) written by thread 0; null in production
};

// reals (8-byte units) of the per-lane working set: rho0, rho1, val [N][H], sigma and regrets [E][H], tmp, leaf values
// [L][H] f32, the lane's tree tables (5N + L + T ints) and the match table (faces x H bytes)
inline size_t cfr_work_reals(int N, int H, int L, int T, int dice, int faces) {
  const size_t tmp = (size_t)std::max(2 * L, L + T * (2 * dice + 2)) + 2;
  const size_t lv = ((size_t)L * H + 1) / 2 + 1;
  const size_t tabs = ((size_t)(5 * N + L + T) * 4 + (size_t)faces * H + 7) / 8 + 1;
  return (size_t)3 * N * H + (size_t)2 * (N - 1) * H + tmp + lv + tabs;
}

// test double of the value net (oracle/orc_api.h: orc_synthetic_net); lives in the -ffp-contract=off TU
void launch_synthetic_net(const float* queries, int64_t rows, int Q, float* out, int H, int A, hipStream_t stream,
                          const long long* range = nullptr);

void launch_cfr(const CfrArgs& a, int B, int block, size_t lds_bytes, hipStream_t stream);

// canonical query rows [rows][Q] = (player, traverser, one-hot last bid [A], reach0 [H], reach1 [H]) <-> split layout:
// dyn [rows][DS] = (traverser, reach0, reach1, 0...), stat [rows][SS] = (player, one-hot, 0...); `range` as in the net launch
void launch_split_queries(const float* canon, int A, int H, float* dyn, int DS, float* stat, int SS, int64_t rows,
                          hipStream_t stream, const long long* range = nullptr);
void launch_unsplit_queries(float* canon, int A, int H, const float* dyn, int DS, const float* stat, int SS, int64_t rows,
                            hipStream_t stream);

// cfr_rows_kernel.hip: kModeStep with one thread per tree row, for LDS-resident lanes of the common games.
// Returns false (nothing launched) when the game has no instantiation.
size_t cfr_rows_lds_bytes(int N, int NI, int H, int L, int faces);
bool cfr_rows_supported(int H, int A, int dice, int faces);
bool launch_cfr_rows(const CfrArgs& a, int B, int block, size_t lds_bytes, hipStream_t stream);
// cfr_wave_kernel.hip: kModeStep with ONE wavefront per lane, element-parallel (the default for the common games)
// its staging loads are unconditional: every global array it reads must be padded by this many ELEMENTS behind the last
// one a lane owns (sigma / values / the per-node and leaf tables / the match table)
constexpr size_t kWavePad = 2048;
// lo_d / lo_p: first node of the deepest level / of its parents' level (ShapeDev::lev_off[nlev - 1], [nlev - 2])
size_t cfr_wave_lds_bytes(int N, int NI, int H, int L, int T, int faces, int lo_d, int lo_p);
bool cfr_wave_supported(int H, int A, int dice, int faces, int max_EH, int max_LH, int max_N);
bool launch_cfr_wave(const CfrArgs& a, int B, size_t lds_bytes, hipStream_t stream);
// the same kernel for lanes whose state does not fit LDS (2 dice x 6 faces): node values and reach rows in LDS, sigma /
// regrets in place in global memory, 256 threads, one lane per CU
size_t cfr_rows_global_lds_bytes(int N, int NI, int H, int L, int faces);
bool cfr_rows_global_supported(int H, int A, int dice, int faces);
bool launch_cfr_rows_global(const CfrArgs& a, int B, size_t lds_bytes, hipStream_t stream);
// cfr_flat_kernel.hip: the same lanes with element-parallel passes and sigma resident in LDS (the default at 2 dice x 6 faces)
size_t cfr_flat_lds_bytes(int N, int NI, int H, int L, int T, int faces);
bool cfr_flat_supported(int H, int A, int dice, int faces);
bool launch_cfr_flat(const CfrArgs& a, int B, size_t lds_bytes, int threads, hipStream_t stream);

}  // namespace rbl
