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

// One 64-byte record per LAUNCH SLOT of the step kernels (cfr_wave_kernel: slot = lane; cfr_flat_kernel: slot = position in the
// size-sorted lane order of its part): everything a workgroup must know before it can issue its first vector load, fetched with ONE
// s_load_dwordx16.  It used to be lane_order -> lane -> lane_shape -> shapes[] (+ lane_root_player, lane_row_off, lane_act_iter,
// wave_tab_off[shape], wave_epv_off[shape]): up to four DEPENDENT scalar round trips in front of the staging loads.
// Written once per epoch by lane_rec_kernel (launch_lane_rec) from the lane descriptors and the per-shape templates.
// The level offsets of the depth <= 2 trees these kernels take are {0, 1, lo2, N} (lo2 == N when the tree has two levels).
struct alignas(64) LaneRec {
  int lane;         // the lane this slot serves
  int N, L, T, NI;  // nodes, pseudo-leaves (net rows), terminals, nodes with a reach row
  int nlev, lo2;    // BFS levels present; first node of level 2
  int root_player, row_off, act_iter;  // lane descriptors (act_iter < 0: no snapshot)
  int tab_off;      // cfr_wave_kernel: byte offset of the shape's table blob in wave_tabs; cfr_flat_kernel: int offset in flat_tabs
  int epv_off;      // cfr_wave_kernel: element offset of the shape's parent-offset table in wave_epv
  int node_off, term_off;  // ShapeDev's offsets into the per-node int tables (generic consumers)
  int flags;        // kRecSkip: no work for this slot (root de-duplication: a root lane served by the epoch's representative);
                    // kRecRep: the representative -- store sigma after EVERY iteration into CfrArgs::snap_all[steps_after]
  int shape;
};
constexpr int kRecSkip = 1, kRecRep = 2;
static_assert(sizeof(LaneRec) == 64, "one s_load_dwordx16");
// the record as the kernels load it: one 16-dword vector from the constant address space (a struct cannot be copied out of it)
typedef int LaneRecWords __attribute__((ext_vector_type(16)));
enum LaneRecWord : int {
  kRecLane = 0, kRecN, kRecL, kRecT, kRecNI, kRecNlev, kRecLo2, kRecRootPlayer, kRecRowOff, kRecActIter, kRecTabOff, kRecEpvOff,
  kRecNodeOff, kRecTermOff, kRecFlags, kRecShape
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
  // cfr_wave_kernel: ONE 4-byte aligned blob per shape in exactly the kernel's LDS layout: parent | act | cb | ce (N bytes each) |
  // leaf nodes (L) | terminal nodes (T) | match table (faces x H); LaneRec::tab_off = its byte offset
  const int8_t* wave_tabs;
  const unsigned short* wave_epv;  // cfr_wave_kernel: per shape, per edge element (c - 1) * H + h: parent(c) * H + h (LaneRec::epv_off)
  // cfr_flat_kernel: ONE 16-byte aligned int blob per shape in exactly the kernel's LDS layout: parent | act | cb | ce | depth | pack |
  // lrow (N ints each) | leaf nodes (L) | terminals (T) | pad to 8 bytes | match masks (faces x 2 x 8 bytes); LaneRec::tab_off
  const int* flat_tabs;
  const LaneRec* lane_rec;  // [max_lanes], by launch slot
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
  // root de-duplication (selfplay_kernels.h; null / unused when off): lane_skip[lane] 0 = normal, 1 = skipped, 2 = representative;
  // the representative also stores sigma after every iteration k into snap_all[k][Emax*H] (k = 0: the uniform start)
  const int* lane_skip;
  double* snap_all;
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

// out[slot] = shape_rec[lane_shape[lane]] with the lane's fields filled in, lane = lane_order ? lane_order[slot] : slot
// (lane_skip: root de-duplication flags per lane, or null -> LaneRec::flags)
void launch_lane_rec(const LaneRec* shape_rec, const int* lane_shape, const int* lane_player, const int* lane_row,
                     const int* lane_act, const int* lane_order, const int* lane_skip, int n, LaneRec* out, hipStream_t stream);

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
