// rebel_amd/csrc/selfplay_kernels.h -- launch interface of the device-side self-play walk (selfplay_kernels.hip).
//
// The recursive-solving outer loop of one reference data-gen thread (RlRunner::step, recursive_solving.cc:160-182;
// sample_state_to_leaf :192-246; sample_state_single :248-275) runs here for every lane on the GPU, between the CFR
// launches of two epochs: no per-epoch snapshot read-back, no host walk.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "tables.h"

namespace rbl {

constexpr int kSpMaxParts = 4;
constexpr int kSpSegs = 8;  // most launches per part of the size-sorted CFR kernel (cfr_rows_kernel<GS>, 2 dice x 6 faces)
// launches of a part with `lanes` lanes: each must still fill the GPU (256 CUs, one to four lanes per CU), or the small launches
// of a stream run one after the other on a half-empty chip
__host__ __device__ inline int sp_segments(int lanes) {
  const int s = lanes / 512;
  return s < 1 ? 1 : (s > kSpSegs ? kSpSegs : s);
}

// Written by sp_scan at the start of an epoch, read back (pinned, async) with the examples at its end.
struct SpEpochInfo {
  long long part_row[kSpMaxParts + 1];              // net-row boundaries of the lane parts (part p = rows [p, p+1))
  unsigned long long part_bytes[kSpMaxParts][2];    // algorithmic bytes of one CFR step per part and traverser
  unsigned long long games;                         // games finished so far (all epochs)
  long long rows;                                   // = part_row[n_parts]
  int seg_shape[kSpMaxParts][kSpSegs];              // shape of the largest tree of each launch segment (sp_order)
  int root_rep;                                     // root de-duplication: the lane that solves the root subgame this epoch (-1: none)
  int skipped;                                      // ... and how many root lanes read its result instead of solving their own
};

struct SpArgs {
  // ---- static tables of the engine (per shape-node, offset by ShapeDev::node_off)
  const ShapeDev* shapes;
  const int* act;
  const int* cb;
  const int* ce;
  const int* depth;
  const int* shape_epar;  // [n_shapes][2]: edges whose parent sits at even / odd depth (roofline accounting)
  int H, A, Q, liar, Emax, num_iters, n, sample_leaf;
  float rap;  // random_action_prob as the float the reference compares eps against (recursive_solving.cc:204)
  // ---- per-lane RNG: std::mt19937 state, [624][n] + position
  uint32_t* mt;
  int* mt_idx;
  // ---- per-lane game state (RlRunner::state_, beliefs_)
  int* bid;
  int* player;
  double* beliefs;  // [n][2][H]
  // ---- the engine's lane descriptors for the epoch (outputs of sp_begin / sp_scan)
  int* lane_shape;
  int* lane_player;
  int* lane_row;
  int* lane_act;
  double* eng_beliefs;  // [n][2][H] root beliefs of the epoch's subgames
  // ---- inputs of sp_end
  const double* snapshot;   // [n][Emax*H] sigma_last at act_iteration
  const double* root_mean;  // [n][2][H]
  // ---- outputs of sp_end: the epoch's training examples, lane-major, 2 per lane (update_value_network)
  float* ex_q;  // [2n][Q]
  float* ex_v;  // [2n][H]
  SpEpochInfo* info;
  // ---- root de-duplication (REBEL_AMD_ROOT_DEDUP=1, default off).  Every lane whose subgame starts at the
  // ROOT state computes the same thing: RlRunner::step resets to the root with uniform beliefs (recursive_solving.cc:160-163) and
  // CFR::step draws nothing, and the lanes are in lock-step under one net.  With dedup on, the lowest-indexed root lane of the
  // epoch (the representative, lane_skip == 2) solves it and keeps sigma after EVERY iteration (snap_all [num_iters + 1][Emax*H]);
  // the other root lanes (lane_skip == 1) get no net rows and no CFR launches, and sp_end reads the representative's sigma at
  // THEIR act_iteration and its root values: their trajectories and examples are bit-identical to the mode-off run.
  int dedup;
  int* lane_skip;          // [n] 0 = solves its own subgame, 1 = root lane served by the representative, 2 = the representative
  const double* snap_all;  // [num_iters + 1][Emax*H]
  int* lane_order;  // [n] lanes of each part sorted by tree size, largest first (null: not wanted)
  int n_parts;
  int part_lane[kSpMaxParts + 1];
};

// host: the state std::mt19937(seed) starts from (position 624: the first draw twists)
void mt19937_seed_state(uint32_t seed, uint32_t* state624);

void launch_sp_begin(const SpArgs& a, hipStream_t st);  // reset finished games, draw act_iteration, descriptors
void launch_sp_scan(const SpArgs& a, hipStream_t st);   // lane_row prefix sums, part boundaries, byte accounting
void launch_sp_order(const SpArgs& a, hipStream_t st);  // lanes of each part sorted by tree size; segment heads -> info
void launch_sp_end(const SpArgs& a, hipStream_t st);    // sampling walk, Bayes updates, examples

// test hook: n_draws of each kind from one lane's generator, in this order per round: uniform_int(0, hi), canonical
// float, discrete over `w` (nw weights)  -> out[3 * rounds] as doubles
void launch_sp_rng_probe(uint32_t* mt, int* mt_idx, int n, int lane, int rounds, int hi, const double* w, int nw,
                         double* out, hipStream_t st);

}  // namespace rbl
