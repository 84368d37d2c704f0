// rebel_amd/csrc/eval_stream.hip -- exploitability of the recursively solved full-tree strategy WITHOUT a dense tabulation.
//
// What it restates (BASELINE configs[4]: "2d x 6f ... + recursive_eval exploitability check"):
//   compute_strategy_recursive_to_leaf, use_sampling_strategy = false   /root/reference/csrc/liars_dice/recursive_solving.cc:76-134
//   compute_exploitability2 / BRSolver::compute_br                       /root/reference/csrc/liars_dice/subgame_solving.cc:802-816, 316-358
//   terminal values, reach sweep                                         subgame_solving.cc:80-98, 765-789, 54-78
// The reference (and rbl_strategy_recursive + rbl_exploitability2, engine.hip) hold the full-tree strategy as a dense
// [N][H][A] array: 241 GB for 2 dice x 6 faces (N = 33.5 M).  Here the strategy lives on the device, EDGE-indexed
// [N - 1][H] fp64 (edge = child node - 1: 9.7 GB), is written level by level -- every subgame of a recursion level is a lane
// of the engine; a scatter kernel copies the lanes' average strategies to their full-tree edges and builds the next level's
// frontier (node ids + beliefs) on the device -- and is consumed in place by two level-synchronous best-response sweeps
// (opponent reach top-down, values bottom-up).  The full tree itself is two flat tables: last bid (1 byte) and first child
// (4 bytes) per node in the reference's BFS order (tree.h:51-70); everything else follows from them.
//
// Arithmetic is the reference's, operation for operation (sequential sums in ascending child / hand order, the fp32
// truncation of the win probability, first-child-then-strictly-greater maximum): on games small enough for both, the
// result is bit-identical to the dense path and to the oracle (tests/test_eval_parity.py).  Compiled with -ffp-contract=off.
//
// Sharding (north_star: independent game batches per GPU).  Every shard solves the root subgame; its pseudo-leaves (the
// nodes at depth max_depth) are dealt to the shards largest subtree first, each to the least loaded shard (subtrees
// below one root ACTION cannot be balanced: the first bid's subtree is half of the game).  A shard then follows only its
// own frontier and returns the best-response values of every node of depth <= max_depth together with the owner of each
// depth-max_depth node; the caller redoes the sweep over those top levels on the host, taking each depth-max_depth
// value from its owner (rebel_amd/capi.py: combine_exploitability) -- no device collective.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <functional>
#include <limits>
#include <random>

#include "engine.h"

namespace rbl {

namespace {

constexpr double kEpsBelief = 1e-80;  // kReachSmoothingEps (subgame_solving.h)

struct FullTree {                   // BFS order of tree.h:51-70; mover of a node = depth & 1
  std::vector<int8_t> bid;          // last bid of the node (-1 at the root)
  std::vector<int32_t> cb;          // first child (children are contiguous: one per legal action, ascending)
  std::vector<int64_t> lev_off;     // node-id range of depth d is [lev_off[d], lev_off[d + 1])
  int64_t N = 0;
};

// max_levels: nodes deeper than that are not generated (the top of the tree: the shards' recombination sweep)
FullTree build_full_tree(const Rules& g, int max_levels = std::numeric_limits<int>::max()) {
  if (g.A > 120) throw std::runtime_error("exploitability_recursive: more than 120 actions");
  FullTree t;
  // N = 2^A nodes for A actions (every increasing bid sequence, optionally closed by liar): refuse what cannot be held
  if (g.A > 27 && max_levels > g.A)
    throw std::runtime_error("exploitability_recursive: full tree too large (2^" + std::to_string(g.A) + " nodes)");
  if (max_levels > g.A) {
    t.bid.reserve((size_t)1 << g.A);
    t.cb.reserve((size_t)1 << g.A);
  }
  t.bid.push_back(-1);
  t.cb.push_back(0);
  t.lev_off.push_back(0);
  int64_t level_end = 1;
  for (int64_t i = 0; i < (int64_t)t.bid.size(); ++i) {
    if (i == level_end) {
      t.lev_off.push_back(i);
      level_end = (int64_t)t.bid.size();
    }
    const int b = t.bid[i];
    if (b == g.liar) continue;
    if ((int)t.lev_off.size() - 1 >= max_levels) continue;  // node i sits at the last level that is kept
    int lo, hi;
    g.bid_range(b, &lo, &hi);
    if ((int64_t)t.bid.size() + (hi - lo) > (int64_t)std::numeric_limits<int32_t>::max())
      throw std::runtime_error("exploitability_recursive: node index overflow");
    t.cb[i] = (int32_t)t.bid.size();
    for (int a = lo; a < hi; ++a) {
      t.bid.push_back((int8_t)a);
      t.cb.push_back(0);
    }
  }
  t.lev_off.push_back((int64_t)t.bid.size());
  t.N = (int64_t)t.bid.size();
  return t;
}

struct ScatterArgs {
  // engine state
  const double* src;  // sum_strategies (CFR) or the average strategy itself (FP), [lane][Emax * H]
  int normalise;      // CFR: rows are sum_strategies, normalised on read (subgame_solving.cc:658-660)
  int steps0, steps1; // num_steps per player (0: the mover's rows keep the uniform initialisation, :518-519)
  int emax, H;
  const int* lane_shape;
  const int* lane_player;
  const double* beliefs;  // [lane][2][H]
  const ShapeDev* shapes;
  const int *parent, *cb, *ce, *depth, *leaves;
  // full tree
  const int32_t* f_cb;
  double* sigma_full;  // [N - 1][H]
  // per lane of this batch
  const int32_t* lane_node;  // full-tree node of the lane's root
  const int64_t* lane_out;   // first slot of the lane's pseudo-leaves in the next frontier
  const int32_t* lane_tag;   // index of the root subgame's pseudo-leaf the lane descends from (-1 at the root)
  // next frontier
  int32_t* nf_node;
  int32_t* nf_tag;
  double* nf_beliefs;  // [slot][2][H]
};

// One workgroup per lane: (1) full-tree ids of the subgame's nodes, (2) average strategy of every edge -> sigma_full and an
// LDS image, (3) per pseudo-leaf and player the reach under that strategy, normalised (normalize_beliefs_inplace,
// recursive_solving.cc:41-44) -> next frontier.
__global__ void __launch_bounds__(256) scatter_strategy_kernel(const ScatterArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = blockIdx.x, H = a.H;
  const ShapeDev& sh = a.shapes[a.lane_shape[lane]];
  const int N = sh.N, off = sh.node_off;
  const int root_player = a.lane_player[lane];
  double* avg = reinterpret_cast<double*>(smem);                      // [N - 1][H]
  int32_t* fid = reinterpret_cast<int32_t*>(avg + (size_t)a.emax * H);  // [N]
  const double* src = a.src + (size_t)lane * a.emax * H;

  if (threadIdx.x == 0) fid[0] = a.lane_node[lane];
  __syncthreads();
  for (int lev = 0; lev + 1 < sh.nlev; ++lev) {
    for (int n = sh.lev_off[lev] + threadIdx.x; n < sh.lev_off[lev + 1]; n += blockDim.x) {
      const int c0 = a.cb[off + n], c1 = a.ce[off + n];
      if (c0 == c1) continue;
      const int32_t base = a.f_cb[fid[n]];
      for (int k = 0; k < c1 - c0; ++k) fid[c0 + k] = base + k;
    }
    __syncthreads();
  }
  // (2) one thread per (node with children, hand)
  for (int i = threadIdx.x; i < N * H; i += blockDim.x) {
    const int n = i / H, h = i - n * H;
    const int c0 = a.cb[off + n], c1 = a.ce[off + n];
    if (c0 == c1) continue;
    const int mover = root_player ^ (a.depth[off + n] & 1);
    const bool untouched = a.normalise && (mover == 0 ? a.steps0 : a.steps1) == 0;
    double sum = 0;
    if (a.normalise && !untouched)
      for (int c = c0; c < c1; ++c) sum += src[(size_t)(c - 1) * H + h];
    for (int c = c0; c < c1; ++c) {
      double v = src[(size_t)(c - 1) * H + h];
      if (untouched) v = 1. / (c1 - c0);
      else if (a.normalise) v = v / sum;
      avg[(size_t)(c - 1) * H + h] = v;
      a.sigma_full[(size_t)(fid[c] - 1) * H + h] = v;
    }
  }
  __syncthreads();
  // (3) one thread per (pseudo-leaf, player)
  for (int i = threadIdx.x; i < sh.L * 2; i += blockDim.x) {
    const int k = i >> 1, pl = i & 1;
    const int leaf = a.leaves[sh.leaf_off + k];
    int path[kMaxLevels];  // nodes from the leaf up to (excluding) the root
    int d = 0;
    for (int n = leaf; n != 0; n = a.parent[off + n]) path[d++] = n;
    const int64_t slot = a.lane_out[lane] + k;
    double* out = a.nf_beliefs + ((size_t)slot * 2 + pl) * H;
    const double* b = a.beliefs + ((size_t)lane * 2 + pl) * H;
    double sum = 0;
    for (int h = 0; h < H; ++h) {
      double r = b[h];
      for (int j = d - 1; j >= 0; --j) {  // root first: child_reaches *= strategy[node][hand][action] on the mover's nodes
        const int n = path[j], p = a.parent[off + n];
        if ((root_player ^ (a.depth[off + p] & 1)) == pl) r *= avg[(size_t)(n - 1) * H + h];
      }
      out[h] = r;
      sum += r + kEpsBelief;
    }
    for (int h = 0; h < H; ++h) out[h] = (out[h] + kEpsBelief) / sum;
    if (pl == 0) {
      a.nf_node[slot] = fid[leaf];
      // which pseudo-leaf of the ROOT subgame a frontier item descends from: inherited, or (root subgame) its own slot
      a.nf_tag[slot] = a.lane_tag[lane] >= 0 ? a.lane_tag[lane] : (int32_t)slot;
    }
  }
}

struct SweepArgs {
  const int8_t* f_bid;
  const int32_t* f_cb;
  const double* sigma_full;  // [N - 1][H]
  double* reach;             // [N][H] opponent reach
  double* val;               // [N][H] traverser values
  const int8_t* matches;     // [faces][H]
  int64_t n0, n1;            // node range of the level being processed
  int H, A, faces, dice, liar;
  int trav, level;           // traverser; depth of the nodes n0..n1
  // forest solve (sampled_add with root_only): sub[n] = which dealt subtree node n belongs to, act_sub[s] = the iteration at
  // which subtree s stops; a node whose subtree has stopped is skipped by every pass (its sigma is that subgame's sampling
  // strategy from then on).  nullptr: no gating
  const int32_t* sub = nullptr;
  const int32_t* act_sub = nullptr;
  int iter = 0;
  int round32 = 0;  // regret reports: the strategy went through a float tensor (recursive_eval.cc:357-358): round on read
};

// opponent reach of the children of level `level` (precompute_reaches, subgame_solving.cc:54-78): thread per (node, hand)
__global__ void br_reach_kernel(const SweepArgs a) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int H = a.H;
  const int64_t n = a.n0 + i / H;
  if (n >= a.n1) return;
  const int h = (int)(i % H);
  const int b = a.f_bid[n];
  if (b == a.liar) return;
  if (a.sub && a.iter >= a.act_sub[a.sub[n]]) return;
  const int cnt = b < 0 ? a.A - 1 : a.A - 1 - b;  // bid_range: bids b + 1 .. A - 2, plus liar unless at the root
  const int64_t c0 = a.f_cb[n];
  const double up = a.reach[n * H + h];
  const bool opp_moves = (a.level & 1) != a.trav;
  for (int k = 0; k < cnt; ++k) {
    double sg = opp_moves ? a.sigma_full[(c0 + k - 1) * H + h] : 1.0;
    if (a.round32) sg = (double)(float)sg;
    a.reach[(c0 + k) * H + h] = opp_moves ? up * sg : up;
  }
}

// Terminal payoffs (compute_expected_terminal_values :80-98, compute_win_probability :765-789) of the liar children of the
// nodes of level `level`: ONE thread per parent (it knows the bid that was called) builds the match histogram of the
// opponent's reach at the terminal once and writes the terminal's H values (a thread per (node, hand) would rebuild the
// histogram H times: 18x the work on half of the tree's nodes).
__global__ void terminal_value_kernel(const SweepArgs a) {
  const int64_t n = a.n0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= a.n1) return;
  const int b = a.f_bid[n];
  if (b == a.liar || b < 0) return;  // terminals have no children; the root's children are all bids
  if (a.sub && a.iter >= a.act_sub[a.sub[n]]) return;
  const int H = a.H, cnt = a.A - 1 - b;
  const int64_t z = (int64_t)a.f_cb[n] + cnt - 1;  // the liar child is the last one
  const int qty = 1 + b / a.faces, face = b % a.faces;
  const int8_t* m = a.matches + face * H;
  const double* r = a.reach + z * H;
  double bins[2 * 8 + 2];
  const int nbins = 2 * a.dice + 1;
  for (int q = 0; q < nbins; ++q) bins[q] = 0.0;
  double s = 0;
  for (int g = 0; g < H; ++g) {
    bins[m[g]] += r[g];
    s += r[g];
  }
  for (int q = nbins - 2; q >= 0; --q) bins[q] += bins[q + 1];
  const bool inverse = ((a.level + 1) & 1) != a.trav;  // mover(z) = the player who did NOT call liar
  double* out = a.val + z * H;
  for (int h = 0; h < H; ++h) {
    const int left = max(0, qty - (int)m[h]);
    const float pwin = (float)bins[left];  // fp32 truncation (:785)
    double y = (double)pwin * 2 - s;
    if (inverse) y *= -1.0;
    out[h] = y;
  }
}

// values of level `level` from the values of level + 1 (compute_br, :326-355); terminal children were valued by
// terminal_value_kernel
__global__ void br_value_kernel(const SweepArgs a) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int H = a.H;
  const int64_t n = a.n0 + i / H;
  if (n >= a.n1) return;
  const int h = (int)(i % H);
  const int b = a.f_bid[n];
  if (b == a.liar) return;
  const int cnt = b < 0 ? a.A - 1 : a.A - 1 - b;
  const int64_t c0 = a.f_cb[n];
  const bool mine = (a.level & 1) == a.trav;
  double x = 0.0;
  for (int k = 0; k < cnt; ++k) {
    const double y = a.val[(c0 + k) * H + h];
    if (mine) {  // first child, then strictly greater (:336-344)
      if (k == 0 || y > x) x = y;
    } else {
      x += y;
    }
  }
  a.val[n * H + h] = x;
}

// beliefs of a batch's lanes gathered from the frontier's slots (lanes sorted by stop iteration sit in other slots than their
// frontier order): dst[i][2][H] = src[slot[i]][2][H]
__global__ void gather_beliefs_kernel(const double* __restrict__ src, const int64_t* __restrict__ slot, double* __restrict__ dst,
                                      int n, int row) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)n * row) return;
  dst[i] = src[slot[i / row] * row + i % row];
}

double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// The recursion of compute_strategy_with_solver_to_leaf (recursive_solving.cc:76-134), level by level: every subgame of a
// level is a lane of `e`; the scatter kernel writes its strategy to the full tree's edges (d_sigma, [N - 1][H]) and builds the
// next frontier.  act == nullptr: the average strategy after num_iters (compute_strategy_recursive_to_leaf, :289-299);
// otherwise (*act)[node] is the act_iteration of the subgame rooted at `node` and the strategy is the solver's
// get_sampling_strategy() at that iteration, which also carries the beliefs down (use_sampling_strategy = true, :117-124).
struct RecursionStats {
  int64_t n_subgames = 0, n_owned = 0;
  int levels = 0;
};

// deal_levels = K >= 1: the frontier PRODUCED by recursion level K - 1 (the non-terminal nodes at depth K * max_depth) is what
// the shards share out; the levels above it are solved by every shard (K = 2 at 2 dice x 6 faces: the root subgame and its 276
// depth-2 subgames, < 0.01 % of the 8.4 M subgames, redundantly -- and the largest dealt subtree is 1/16 of the game instead
// of 1/4, so eight shards balance).
RecursionStats recursive_fill(Engine& e, const FullTree& ft, const int32_t* d_cb, double* d_sigma, int shard, int n_shards,
                              const std::vector<int16_t>* act, int32_t* top_owner, int deal_levels = 1,
                              int max_levels = std::numeric_limits<int>::max(), std::vector<int32_t>* last_nodes = nullptr,
                              DevBuf<double>* last_beliefs = nullptr) {
  // max_levels: stop after that many recursion levels and hand the frontier they produced (node ids, and the beliefs
  // [count][2][H] on the device) to the caller instead of following it (root_only repeats: the forest solve takes over)
  const Rules& g = e.rules();
  const ShapeTables& tb = e.tables();
  const int H = g.H;
  if (n_shards < 1 || shard < 0 || shard >= n_shards) throw std::runtime_error("recursive solve on the device: bad shard");
  if (deal_levels < 1) throw std::runtime_error("recursive solve on the device: deal_levels must be >= 1");
  RBL_HIP_CHECK(hipSetDevice(e.device()));
  hipStream_t st = e.stream();
  const int D = e.params().max_depth;
  if (D < 1) throw std::runtime_error("recursive solve on the device: max_depth must be >= 1");
  if (top_owner) {  // (a recursion that ends above the dealt level leaves every node unowned: each shard then has everything)
    const int64_t top_lev = (int64_t)deal_levels * D;
    const int64_t M = (int64_t)ft.lev_off.size() - 1 > top_lev ? ft.lev_off[top_lev + 1] : ft.N;
    for (int64_t i = 0; i < M; ++i) top_owner[i] = -1;
  }
  const size_t lds = (size_t)e.emax() * H * sizeof(double) + (size_t)tb.max_N * sizeof(int32_t);
  if (lds > 160 * 1024)
    throw std::runtime_error("recursive solve on the device: subgames of depth " + std::to_string(D) +
                             " do not fit the scatter kernel's LDS image (" + std::to_string(lds) + " bytes)");
  RBL_HIP_CHECK(hipFuncSetAttribute((const void*)scatter_strategy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  DevBuf<int32_t> d_node[2], d_lane_node, d_tag[2], d_lane_tag;
  DevBuf<int64_t> d_lane_out, d_slot;
  DevBuf<double> d_bel[2], d_gather;

  // ------------------------------------------------------------------ recursion, level by level
  std::vector<int32_t> f_node{0};
  std::vector<int32_t> f_tag{-1};
  std::vector<int32_t> owner_of_tag;  // shard of each pseudo-leaf of the root subgame
  d_bel[0].alloc(2 * (size_t)H);
  {
    std::vector<double> b(2 * (size_t)H, 1.0 / H);  // get_initial_beliefs
    d_bel[0].upload(b, st);
  }
  int cur = 0, level = 0;
  int64_t n_subgames = 0, n_owned = 0;
  const int maxB = e.max_lanes();
  std::vector<int32_t> bids(maxB), players(maxB), lane_node(maxB), lane_act(maxB);
  std::vector<int32_t> lane_tag(maxB);
  std::vector<int64_t> lane_out(maxB);
  std::vector<double> bel_host((size_t)maxB * 2 * H);
  while (!f_node.empty()) {
    const size_t count = f_node.size();
    const int player = (int)(((int64_t)level * D) & 1);
    // Sampled strategies: a lane stops at its own act_iteration and a batch runs to its LARGEST one.  In frontier order every
    // batch of a wide level contains a lane near num_iters; sorted by stop iteration (descending, stable) the batches run
    // num_iters, ..., 0 steps: about half the work at the wide levels.  Subgames of a level are independent and every output is
    // keyed by node id, so the order of the lanes changes no result; only the beliefs must be fetched from the lanes' slots.
    std::vector<int64_t> slot_of;  // frontier slot (= position of the beliefs in d_bel[cur]) of the i-th lane in solve order
    if (act && count > (size_t)e.max_lanes()) {
      slot_of.resize(count);
      for (size_t i = 0; i < count; ++i) slot_of[i] = (int64_t)i;
      std::stable_sort(slot_of.begin(), slot_of.end(),
                       [&](int64_t x, int64_t y) { return (*act)[f_node[(size_t)x]] > (*act)[f_node[(size_t)y]]; });
      std::vector<int32_t> sn(count), stg(count);
      for (size_t i = 0; i < count; ++i) {
        sn[i] = f_node[(size_t)slot_of[i]];
        stg[i] = f_tag[(size_t)slot_of[i]];
      }
      f_node.swap(sn);
      f_tag.swap(stg);
    }
    // the frontier of the next level: every pseudo-leaf of every subgame of this level (a non-terminal node always has children)
    std::vector<int64_t> out_off(count + 1, 0);
    for (size_t i = 0; i < count; ++i) out_off[i + 1] = out_off[i] + tb.shapes[ft.bid[f_node[i]] + 1].L;
    const int64_t next_count = out_off[count];
    const int nxt = cur ^ 1;
    d_node[nxt].alloc((size_t)std::max<int64_t>(1, next_count));
    d_tag[nxt].alloc((size_t)std::max<int64_t>(1, next_count));
    d_bel[nxt].alloc((size_t)std::max<int64_t>(1, next_count) * 2 * H);
    for (size_t base = 0; base < count; base += maxB) {
      const int B = (int)std::min<size_t>(maxB, count - base);
      if (slot_of.empty()) {
        RBL_HIP_CHECK(hipMemcpyAsync(bel_host.data(), d_bel[cur].p + base * 2 * H, (size_t)B * 2 * H * sizeof(double),
                                     hipMemcpyDeviceToHost, st));
      } else {
        d_slot.upload(std::vector<int64_t>(slot_of.begin() + base, slot_of.begin() + base + B), st);
        if (d_gather.n < (size_t)B * 2 * H) d_gather.alloc((size_t)maxB * 2 * H);
        const int64_t work = (int64_t)B * 2 * H;
        hipLaunchKernelGGL(gather_beliefs_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, st, d_bel[cur].p, d_slot.p,
                           d_gather.p, B, 2 * H);
        RBL_HIP_CHECK(hipGetLastError());
        RBL_HIP_CHECK(hipMemcpyAsync(bel_host.data(), d_gather.p, (size_t)B * 2 * H * sizeof(double), hipMemcpyDeviceToHost, st));
      }
      RBL_HIP_CHECK(hipStreamSynchronize(st));
      int steps = act ? 0 : -1;
      for (int i = 0; i < B; ++i) {
        bids[i] = ft.bid[f_node[base + i]];
        players[i] = player;
        lane_node[i] = f_node[base + i];
        lane_tag[i] = level < deal_levels ? -1 : f_tag[base + i];  // the dealt level's output numbers its own slots
        lane_out[i] = out_off[base + i];
        if (act) {
          lane_act[i] = (*act)[f_node[base + i]];
          steps = std::max(steps, lane_act[i]);
        }
      }
      e.reset(B, bids.data(), players.data(), bel_host.data(), act ? lane_act.data() : nullptr);
      e.multistep(steps);
      e.sync();  // every lane part (stream) of the engine has finished before the scatter kernel reads the lanes
      const Engine::EvalView v = e.eval_view();
      d_lane_node.upload(std::vector<int32_t>(lane_node.begin(), lane_node.begin() + B), st);
      d_lane_tag.upload(std::vector<int32_t>(lane_tag.begin(), lane_tag.begin() + B), st);
      d_lane_out.upload(std::vector<int64_t>(lane_out.begin(), lane_out.begin() + B), st);
      ScatterArgs a{};
      a.src = act ? v.snapshot : (v.use_cfr ? v.sums : v.sigma);
      a.normalise = !act && v.use_cfr ? 1 : 0;
      a.steps0 = v.num_steps[0];
      a.steps1 = v.num_steps[1];
      a.emax = e.emax();
      a.H = H;
      a.lane_shape = v.lane_shape;
      a.lane_player = v.lane_player;
      a.beliefs = v.beliefs;
      a.shapes = v.shapes;
      a.parent = v.parent;
      a.cb = v.cb;
      a.ce = v.ce;
      a.depth = v.depth;
      a.leaves = v.leaves;
      a.f_cb = d_cb;
      a.sigma_full = d_sigma;
      a.lane_node = d_lane_node.p;
      a.lane_out = d_lane_out.p;
      a.lane_tag = d_lane_tag.p;
      a.nf_node = d_node[nxt].p;
      a.nf_tag = d_tag[nxt].p;
      a.nf_beliefs = d_bel[nxt].p;
      hipLaunchKernelGGL(scatter_strategy_kernel, dim3(B), dim3(256), lds, st, a);
      RBL_HIP_CHECK(hipGetLastError());
      RBL_HIP_CHECK(hipStreamSynchronize(st));  // the staging vectors above and the engine's lanes are reused next batch
      n_subgames += B;
    }
    n_owned += (int64_t)count;
    // next frontier: ids and tags back to the host (the beliefs stay on the device); a shard keeps its own root actions
    std::vector<int32_t> nn((size_t)next_count);
    std::vector<int32_t> nt((size_t)next_count);
    if (next_count) {
      RBL_HIP_CHECK(hipMemcpyAsync(nn.data(), d_node[nxt].p, (size_t)next_count * sizeof(int32_t), hipMemcpyDeviceToHost, st));
      RBL_HIP_CHECK(hipMemcpyAsync(nt.data(), d_tag[nxt].p, (size_t)next_count * sizeof(int32_t), hipMemcpyDeviceToHost, st));
      RBL_HIP_CHECK(hipStreamSynchronize(st));
    }
    if (level == deal_levels - 1) {
      // deal this level's pseudo-leaves to the shards: largest subtree first (a node with last bid b heads
      // 2^(liar - b) nodes), each to the least loaded shard; ties by index, so every shard computes the same map
      owner_of_tag.assign((size_t)next_count, 0);
      std::vector<int64_t> order((size_t)next_count);
      for (int64_t i = 0; i < next_count; ++i) order[i] = i;
      std::stable_sort(order.begin(), order.end(), [&](int64_t x, int64_t y) { return ft.bid[nn[x]] < ft.bid[nn[y]]; });
      std::vector<double> load(n_shards, 0.0);
      for (int64_t i : order) {
        int best = 0;
        for (int sdx = 1; sdx < n_shards; ++sdx)
          if (load[sdx] < load[best]) best = sdx;
        owner_of_tag[(size_t)nt[i]] = best;
        load[best] += std::ldexp(1.0, g.liar - ft.bid[nn[i]]);
      }
      if (top_owner)
        for (int64_t i = 0; i < next_count; ++i) top_owner[nn[i]] = owner_of_tag[(size_t)nt[i]];
    }
    if (n_shards > 1 && next_count && level >= deal_levels - 1) {  // compact the frontier (ids, tags and beliefs) to the items this shard owns
      std::vector<int64_t> keep;
      for (int64_t i = 0; i < next_count; ++i)
        if (owner_of_tag[(size_t)nt[i]] == shard) keep.push_back(i);
      DevBuf<double> compact;
      compact.alloc(std::max<size_t>(1, keep.size()) * 2 * H);
      // runs of consecutive kept slots are copied together
      size_t w = 0;
      for (size_t i = 0; i < keep.size();) {
        size_t j = i + 1;
        while (j < keep.size() && keep[j] == keep[j - 1] + 1) ++j;
        RBL_HIP_CHECK(hipMemcpyAsync(compact.p + w * 2 * H, d_bel[nxt].p + (size_t)keep[i] * 2 * H,
                                     (j - i) * 2 * H * sizeof(double), hipMemcpyDeviceToDevice, st));
        w += j - i;
        i = j;
      }
      RBL_HIP_CHECK(hipStreamSynchronize(st));
      std::swap(d_bel[nxt].p, compact.p);
      std::swap(d_bel[nxt].n, compact.n);
      std::vector<int32_t> kn(keep.size());
      std::vector<int32_t> kt(keep.size());
      for (size_t i = 0; i < keep.size(); ++i) {
        kn[i] = nn[keep[i]];
        kt[i] = nt[keep[i]];
      }
      nn.swap(kn);
      nt.swap(kt);
    }
    f_node.swap(nn);
    f_tag.swap(nt);
    cur = nxt;
    ++level;
    if (level >= max_levels) {
      if (last_nodes) *last_nodes = f_node;
      if (last_beliefs) {
        std::swap(last_beliefs->p, d_bel[cur].p);
        std::swap(last_beliefs->n, d_bel[cur].n);
      }
      break;
    }
  }
  RecursionStats rs;
  rs.n_subgames = n_subgames;
  rs.n_owned = n_owned;
  rs.levels = level;
  return rs;
}

}  // namespace

// nodes of depth <= deal_levels * max_depth of the full tree (the region the shards hand back: top_values / top_owner)
int64_t exploitability_top_nodes(const Rules& g, int max_depth, int deal_levels) {
  if (max_depth < 1 || deal_levels < 1) throw std::runtime_error("exploitability_top_nodes: max_depth and deal_levels must be >= 1");
  const int64_t lev = std::min<int64_t>((int64_t)max_depth * deal_levels, g.A + 1);
  return build_full_tree(g, (int)lev).N;
}

// Shards of exploitability_recursive -> the two exploitabilities (host only, no device): BRSolver::compute_br
// (subgame_solving.cc:326-355) over the nodes of depth <= deal_levels * max_depth.  A node whose subtree was dealt takes its
// value from its owner; every other childless node of the region (terminals; nodes of a recursion that ended early) from
// shard 0 -- those depend only on levels every shard solves.  Above them the traverser's nodes take the first-then-strictly-
// greater maximum over their children, the opponent's the sum in ascending order; then vector_sum / H (:813-814).
void exploitability_combine(const Rules& g, int max_depth, int deal_levels, int n_shards, const double* const* top_values,
                            const int32_t* top_owner, double* out2) {
  if (n_shards < 1 || !top_values || !top_owner) throw std::runtime_error("exploitability_combine: bad arguments");
  const int64_t lev = std::min<int64_t>((int64_t)max_depth * deal_levels, g.A + 1);
  const FullTree ft = build_full_tree(g, (int)lev);
  const int64_t M = ft.N;
  const int H = g.H;
  std::vector<int> depth((size_t)M, 0);
  for (size_t d = 0; d + 1 < ft.lev_off.size(); ++d)
    for (int64_t n = ft.lev_off[d]; n < ft.lev_off[d + 1]; ++n) depth[(size_t)n] = (int)d;
  std::vector<double> val((size_t)M * H);
  for (int t = 0; t < 2; ++t) {
    for (int64_t n = M - 1; n >= 0; --n) {
      const int b = ft.bid[(size_t)n];
      const bool childless = b == g.liar || depth[(size_t)n] >= lev;
      double* v = val.data() + (size_t)n * H;
      if (childless) {
        const int o = top_owner[n] >= 0 ? top_owner[n] : 0;
        if (o >= n_shards || !top_values[o]) throw std::runtime_error("exploitability_combine: owner shard missing");
        std::memcpy(v, top_values[o] + ((size_t)t * M + (size_t)n) * H, (size_t)H * sizeof(double));
        continue;
      }
      int lo, hi;
      g.bid_range(b, &lo, &hi);
      const int64_t c0 = ft.cb[(size_t)n];
      const bool mine = (depth[(size_t)n] & 1) == t;
      for (int h = 0; h < H; ++h) {
        double x = 0.0;
        for (int k = 0; k < hi - lo; ++k) {
          const double y = val[(size_t)(c0 + k) * H + h];
          if (mine) {
            if (k == 0 || y > x) x = y;
          } else {
            x += y;
          }
        }
        v[h] = x;
      }
    }
    double s = 0;
    for (int h = 0; h < H; ++h) s += val[h];
    out2[t] = s / H;
  }
}

// out2: exploitabilities of the two players (n_shards == 1, NaN otherwise).  top_values (optional): [2][M][H] best-response
// values per traverser of the M nodes of depth <= max_depth (M = size of unroll_tree(game, root, max_depth)); top_owner
// (optional): [M] shard that owns the subtree of each non-terminal depth-max_depth node, -1 for every other node.
// stats (optional): [8] = {nodes, subgames solved, levels, solve s, sweep s, bytes of the strategy, frontier items, M}.
void exploitability_recursive(Engine& e, int shard, int n_shards, double* out2, double* top_values, int32_t* top_owner,
                              double* stats, int deal_levels) {
  const Rules& g = e.rules();
  const int H = g.H, A = g.A;
  if (g.dice > 8) throw std::runtime_error("exploitability_recursive: more than 8 dice");
  RBL_HIP_CHECK(hipSetDevice(e.device()));
  hipStream_t st = e.stream();
  const int D = e.params().max_depth;
  const double t0 = now_s();
  const FullTree ft = build_full_tree(g);
  DevBuf<int8_t> d_bid, d_matches;
  DevBuf<int32_t> d_cb;
  DevBuf<double> d_sigma;
  d_bid.upload(ft.bid, st);
  d_cb.upload(ft.cb, st);
  d_sigma.alloc((size_t)std::max<int64_t>(1, ft.N - 1) * H);
  RBL_HIP_CHECK(hipMemsetAsync(d_sigma.p, 0, d_sigma.n * sizeof(double), st));
  {
    std::vector<int8_t> m((size_t)g.faces * H);
    for (int f = 0; f < g.faces; ++f)
      for (int h = 0; h < H; ++h) m[(size_t)f * H + h] = (int8_t)g.matches(h, f);
    d_matches.upload(m, st);
  }
  const RecursionStats rs = recursive_fill(e, ft, d_cb.p, d_sigma.p, shard, n_shards, nullptr, top_owner, deal_levels);
  const double t1 = now_s();

  // ------------------------------------------------------------------ best-response sweeps over the full tree
  DevBuf<double> d_reach, d_val;
  d_reach.alloc((size_t)ft.N * H);
  d_val.alloc((size_t)ft.N * H);
  const int nlev = (int)ft.lev_off.size() - 1;
  std::vector<double> root(H);
  const int64_t top_lev = (int64_t)deal_levels * D;  // nodes of depth <= deal_levels * max_depth
  const int64_t M = (int64_t)ft.lev_off.size() - 1 > top_lev ? ft.lev_off[top_lev + 1] : ft.N;
  for (int t = 0; t < 2; ++t) {
    std::vector<double> b(H, 1.0 / H);
    RBL_HIP_CHECK(hipMemcpyAsync(d_reach.p, b.data(), H * sizeof(double), hipMemcpyHostToDevice, st));
    RBL_HIP_CHECK(hipStreamSynchronize(st));
    SweepArgs a{};
    a.f_bid = d_bid.p;
    a.f_cb = d_cb.p;
    a.sigma_full = d_sigma.p;
    a.reach = d_reach.p;
    a.val = d_val.p;
    a.matches = d_matches.p;
    a.H = H;
    a.A = A;
    a.faces = g.faces;
    a.dice = g.dice;
    a.liar = g.liar;
    a.trav = t;
    for (int lev = 0; lev + 1 < nlev; ++lev) {
      a.n0 = ft.lev_off[lev];
      a.n1 = ft.lev_off[lev + 1];
      a.level = lev;
      const int64_t work = (a.n1 - a.n0) * H;
      hipLaunchKernelGGL(br_reach_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, st, a);
    }
    for (int lev = nlev - 2; lev >= 0; --lev) {
      a.n0 = ft.lev_off[lev];
      a.n1 = ft.lev_off[lev + 1];
      a.level = lev;
      const int64_t work = (a.n1 - a.n0) * H;
      hipLaunchKernelGGL(terminal_value_kernel, dim3((unsigned)((a.n1 - a.n0 + 255) / 256)), dim3(256), 0, st, a);
      hipLaunchKernelGGL(br_value_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, st, a);
    }
    RBL_HIP_CHECK(hipGetLastError());
    RBL_HIP_CHECK(hipMemcpyAsync(root.data(), d_val.p, H * sizeof(double), hipMemcpyDeviceToHost, st));
    if (top_values)
      RBL_HIP_CHECK(hipMemcpyAsync(top_values + (size_t)t * M * H, d_val.p, (size_t)M * H * sizeof(double),
                                   hipMemcpyDeviceToHost, st));
    RBL_HIP_CHECK(hipStreamSynchronize(st));
    double s = 0;
    for (int h = 0; h < H; ++h) s += root[h];  // vector_sum (util.h:87-90)
    out2[t] = n_shards == 1 ? s / H : std::numeric_limits<double>::quiet_NaN();
  }
  const double t2 = now_s();
  if (stats) {
    stats[0] = (double)ft.N;
    stats[1] = (double)rs.n_subgames;
    stats[2] = (double)rs.levels;
    stats[3] = t1 - t0;
    stats[4] = t2 - t1;
    stats[5] = (double)d_sigma.n * sizeof(double);
    stats[6] = (double)rs.n_owned;
    stats[7] = (double)M;
  }
}

// =================================================================================================== full-tree CFR, streamed
// The reference's evaluation tool first solves the WHOLE game with the same solver (recursive_eval.cc:269-296:
// build_solver with max_depth = 100000, `subgame_iters` steps, exploitability at iterations 2^k).  Its dense state does not
// fit at 2 dice x 6 faces either; this is the same CFR (subgame_solving.cc:509-670, no pseudo-leaves on a full tree) as
// level-synchronous sweeps over edge-indexed arrays in HBM -- sigma, regrets, sum_strategies [N-1][H] and two node arrays
// (reach, values) [N][H], 58 GB for 33.5 M nodes.  One step = opponent reach top-down, values + regret update + regret
// matching bottom-up (one thread per (node, hand), children walked in ascending order), then the traverser's reach under
// the new strategy + sum_strategies top-down.  Arithmetic as in cfr_kernels.hip (same operands, same order): on games that
// fit the engine it is bit-identical to a full-depth lane and to the oracle (tests/test_eval_parity.py).
namespace {

struct StepArgs {
  const int8_t* f_bid;
  const int32_t* f_cb;
  double *sigma, *regrets, *sums, *avg;  // [N - 1][H]
  double *reach, *val;                   // [N][H]
  const int8_t* matches;
  int64_t n0, n1;
  int H, A, faces, dice, liar, trav, level;
  double pos, neg, strat;
  int steps0, steps1;
  const int32_t* sub = nullptr;  // forest solve: see SweepArgs
  const int32_t* act_sub = nullptr;
  int iter = 0;
};

__device__ __forceinline__ int child_count(int b, int A) { return b < 0 ? A - 1 : A - 1 - b; }

// ctor (:509-534): uniform sigma, zero regrets, sum_strategies = uniform x the mover's reach under the uniform strategy
// (:125-149); `reach` holds both players' uniform reach interleaved as [N][2][H] here (val's storage is borrowed for it)
__global__ void st_init_kernel(const StepArgs a, double* reach2) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int H = a.H;
  const int64_t n = a.n0 + i / H;
  if (n >= a.n1) return;
  const int h = (int)(i % H);
  const int b = a.f_bid[n];
  if (b == a.liar) return;
  const int cnt = child_count(b, a.A), mover = a.level & 1;
  const int64_t c0 = a.f_cb[n];
  const double u = 1. / cnt;
  const double r0 = reach2[(n * 2 + 0) * H + h], r1 = reach2[(n * 2 + 1) * H + h];
  const double rm = mover == 0 ? r0 : r1;
  for (int k = 0; k < cnt; ++k) {
    const int64_t e = (c0 + k - 1) * H + h;
    a.sigma[e] = u;
    a.regrets[e] = 0.0;
    a.sums[e] = u * rm;
    reach2[((c0 + k) * 2 + 0) * H + h] = mover == 0 ? r0 * u : r0;
    reach2[((c0 + k) * 2 + 1) * H + h] = mover == 1 ? r1 * u : r1;
  }
}

// update_regrets (:542-574) fused with regret matching (:619-634) and the regret discount (:639-650) at the traverser's nodes
__global__ void st_value_kernel(const StepArgs a) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int H = a.H;
  const int64_t n = a.n0 + i / H;
  if (n >= a.n1) return;
  const int h = (int)(i % H);
  const int b = a.f_bid[n];
  if (b == a.liar) return;
  if (a.sub && a.iter >= a.act_sub[a.sub[n]]) return;
  const int cnt = child_count(b, a.A);
  const int64_t c0 = a.f_cb[n];
  const bool mine = (a.level & 1) == a.trav;
  auto child_val = [&](int k) { return a.val[(c0 + k) * H + h]; };  // terminals: terminal_value_kernel
  double x = 0.0;
  if (mine) {
    for (int k = 0; k < cnt; ++k) x += child_val(k) * a.sigma[(c0 + k - 1) * H + h];
    double s = 0.0;
    for (int k = 0; k < cnt; ++k) {
      const int64_t e = (c0 + k - 1) * H + h;
      double r = a.regrets[e];
      r += child_val(k);
      r -= x;
      const double mm = r > 1e-80 ? r : 1e-80;  // std::max(regret, kRegretSmoothingEps)
      s += mm;
      a.sigma[e] = mm;
      a.regrets[e] = r * (r > 0 ? a.pos : a.neg);
    }
    for (int k = 0; k < cnt; ++k) {
      const int64_t e = (c0 + k - 1) * H + h;
      a.sigma[e] = a.sigma[e] / s;
    }
  } else {
    for (int k = 0; k < cnt; ++k) x += child_val(k);
  }
  a.val[n * H + h] = x;
}

// traverser's reach under the NEW sigma (:636-638) + sum_strategies (:651-657), top-down
__global__ void st_sums_kernel(const StepArgs a) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int H = a.H;
  const int64_t n = a.n0 + i / H;
  if (n >= a.n1) return;
  const int h = (int)(i % H);
  const int b = a.f_bid[n];
  if (b == a.liar) return;
  const int cnt = child_count(b, a.A);
  const int64_t c0 = a.f_cb[n];
  const double r = a.reach[n * H + h];
  const bool mine = (a.level & 1) == a.trav;
  for (int k = 0; k < cnt; ++k) {
    const int64_t e = (c0 + k - 1) * H + h;
    if (mine) {
      const double sg = a.sigma[e];
      double sm = a.sums[e];
      sm *= a.strat;
      sm += r * sg;
      a.sums[e] = sm;
      a.reach[(c0 + k) * H + h] = r * sg;
    } else {
      a.reach[(c0 + k) * H + h] = r;
    }
  }
}

// get_strategy: sum_strategies normalised per (node, hand) on nodes whose mover has stepped (:658-660), else the uniform init
__global__ void st_average_kernel(const StepArgs a) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int H = a.H;
  const int64_t n = a.n0 + i / H;
  if (n >= a.n1) return;
  const int h = (int)(i % H);
  const int b = a.f_bid[n];
  if (b == a.liar) return;
  const int cnt = child_count(b, a.A);
  const int64_t c0 = a.f_cb[n];
  const bool untouched = ((a.level & 1) == 0 ? a.steps0 : a.steps1) == 0;
  double s = 0.0;
  if (!untouched)
    for (int k = 0; k < cnt; ++k) s += a.sums[(c0 + k - 1) * H + h];
  for (int k = 0; k < cnt; ++k) {
    const int64_t e = (c0 + k - 1) * H + h;
    a.avg[e] = untouched ? 1. / cnt : a.sums[e] / s;
  }
}

// ---- forest solve: every non-terminal node of one level is the root of an independent full-depth subgame (root_only repeats)
// CFR ctor on the nodes of a level (:509-534 without sum_strategies, which the sampling strategy does not read)
__global__ void forest_init_kernel(const StepArgs a) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int H = a.H;
  const int64_t n = a.n0 + i / H;
  if (n >= a.n1) return;
  const int h = (int)(i % H);
  const int b = a.f_bid[n];
  if (b == a.liar) return;
  const int cnt = child_count(b, a.A);
  const int64_t c0 = a.f_cb[n];
  const double u = 1. / cnt;
  for (int k = 0; k < cnt; ++k) {
    a.sigma[(c0 + k - 1) * H + h] = u;
    a.regrets[(c0 + k - 1) * H + h] = 0.0;
  }
}
// which subtree a node belongs to: roots number themselves, everybody below inherits (thread per parent)
__global__ void forest_sub_kernel(const int8_t* f_bid, const int32_t* f_cb, int32_t* sub, int64_t n0, int64_t n1, int A, int liar,
                                  int is_root_level) {
  const int64_t n = n0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= n1) return;
  if (is_root_level) sub[n] = (int32_t)(n - n0);
  const int b = f_bid[n];
  if (b == liar) return;
  const int me = is_root_level ? (int32_t)(n - n0) : sub[n];
  const int cnt = child_count(b, A);
  const int64_t c0 = f_cb[n];
  for (int k = 0; k < cnt; ++k) sub[c0 + k] = me;
}
// reach of player `pl` at the subgame roots = their beliefs: reach[node[i]][h] = bel[i][pl][h]
__global__ void forest_root_reach_kernel(const int32_t* node, const double* bel, double* reach, int64_t count, int H, int pl) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count * H) return;
  const int64_t k = i / H;
  const int h = (int)(i % H);
  reach[(int64_t)node[k] * H + h] = bel[(k * 2 + pl) * H + h];
}

// ---- sampled repeats (recursive_eval.cc:136-160, 336-363).  The reference keeps float32 tensors summed_strategy
// [N][H][A] and summed_reach [N][H][1]; the weight of (node, hand) in a repeat is the reach of the node's mover under that
// repeat's strategy from uniform beliefs (compute_stategy_stats: subgame_solving.cc:839-842), computed in fp64 from the fp64
// strategy and THEN rounded to float; the strategy is rounded to float on its own (tree_strategy_to_tensor :82-96); product
// and running sums are float operations (no contraction: -ffp-contract=off).
struct AccArgs {
  const int8_t* f_bid;
  const int32_t* f_cb;
  const double* samp;  // [N - 1][H] this repeat's strategy
  double *r0, *r1;     // [N][H] reach of player 0 / 1 under it
  float* sstrat;       // [N - 1][H]
  float* sreach;       // [N][H]
  double* fin;         // [N - 1][H]
  int64_t n0, n1;
  int H, A, liar, level;
};

__global__ void sm_accumulate_kernel(const AccArgs a) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int H = a.H;
  const int64_t n = a.n0 + i / H;
  if (n >= a.n1) return;
  const int h = (int)(i % H);
  const int b = a.f_bid[n];
  if (b == a.liar) return;  // a terminal's row of the strategy is zero: nothing to add
  const int cnt = child_count(b, a.A), mover = a.level & 1;
  const int64_t c0 = a.f_cb[n];
  const double r0 = a.r0[n * H + h], r1 = a.r1[n * H + h];
  const float w = (float)(mover == 0 ? r0 : r1);
  a.sreach[n * H + h] += w;
  for (int k = 0; k < cnt; ++k) {
    const int64_t e = (c0 + k - 1) * H + h;
    const double s = a.samp[e];
    const float prod = (float)s * w;
    a.sstrat[e] += prod;
    a.r0[(c0 + k) * H + h] = mover == 0 ? r0 * s : r0;
    a.r1[(c0 + k) * H + h] = mover == 1 ? r1 * s : r1;
  }
}

// final_strategy = summed_strategy / (summed_reach + 1e-6), float division, then widened (:352-353)
__global__ void sm_final_kernel(const AccArgs a) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int H = a.H;
  const int64_t n = a.n0 + i / H;
  if (n >= a.n1) return;
  const int h = (int)(i % H);
  const int b = a.f_bid[n];
  if (b == a.liar) return;
  const int cnt = child_count(b, a.A);
  const int64_t c0 = a.f_cb[n];
  const float den = a.sreach[n * H + h] + 1e-6f;
  for (int k = 0; k < cnt; ++k) {
    const int64_t e = (c0 + k - 1) * H + h;
    a.fin[e] = (double)(a.sstrat[e] / den);
  }
}

// compute_ev (:931-973): player 0 follows `sigma_own` at its nodes, values of the other player's nodes are sums
__global__ void ev_value_kernel(const SweepArgs a, const double* sigma_own) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int H = a.H;
  const int64_t n = a.n0 + i / H;
  if (n >= a.n1) return;
  const int h = (int)(i % H);
  const int b = a.f_bid[n];
  if (b == a.liar) return;
  const int cnt = child_count(b, a.A);
  const int64_t c0 = a.f_cb[n];
  const bool mine = (a.level & 1) == a.trav;
  double x = 0.0;
  for (int k = 0; k < cnt; ++k) {
    const double y = a.val[(c0 + k) * H + h];
    if (mine) x += sigma_own[(c0 + k - 1) * H + h] * y;
    else x += y;
  }
  a.val[n * H + h] = x;
}

// compute_immediate_regrets (subgame_solving.cc:984-1050), one (strategy, traverser) sweep: values bottom-up under the strategy,
// regrets[node][hand][action] += value(child), then -= value(node) at the traverser's nodes -- accumulated over strategies
__global__ void ir_value_kernel(const SweepArgs a, double* regrets) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int H = a.H;
  const int64_t n = a.n0 + i / H;
  if (n >= a.n1) return;
  const int h = (int)(i % H);
  const int b = a.f_bid[n];
  if (b == a.liar) return;
  const int cnt = child_count(b, a.A);
  const int64_t c0 = a.f_cb[n];
  const bool mine = (a.level & 1) == a.trav;
  double x = 0.0;
  if (mine) {
    for (int k = 0; k < cnt; ++k) {
      double sg = a.sigma_full[(c0 + k - 1) * H + h];
      if (a.round32) sg = (double)(float)sg;
      x += a.val[(c0 + k) * H + h] * sg;
    }
    for (int k = 0; k < cnt; ++k) {
      const int64_t e = (c0 + k - 1) * H + h;
      double r = regrets[e];
      r += a.val[(c0 + k) * H + h];
      r -= x;
      regrets[e] = r;
    }
  } else {
    for (int k = 0; k < cnt; ++k) x += a.val[(c0 + k) * H + h];
  }
  a.val[n * H + h] = x;
}
// immediate_regrets[node][hand] = max over the DENSE action row / n_strategies (:1041-1046): every node has at least one
// illegal action, whose regret stays 0, so the maximum is never negative; 0 on nodes without children
__global__ void ir_max_kernel(const int8_t* f_bid, const int32_t* f_cb, const double* regrets, double* out, int64_t N, int H, int A,
                              int liar, double n_strategies) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * H) return;
  const int64_t n = i / H;
  const int h = (int)(i % H);
  const int b = f_bid[n];
  double best = 0.0;
  if (b != liar) {
    const int cnt = child_count(b, A);
    const int64_t c0 = f_cb[n];
    for (int k = 0; k < cnt; ++k) {
      const double r = regrets[(c0 + k - 1) * H + h];
      best = r > best ? r : best;
    }
    best = best / n_strategies;
  }
  out[i] = best;
}
__global__ void ir_node_sum_kernel(const double* v, double* out, int64_t N, int H) {  // vector_sum per node (util.h:87-90)
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  double sum = 0.0;
  for (int h = 0; h < H; ++h) sum += v[n * H + h];
  out[n] = sum;
}

// act_iteration of every subgame of the recursion, drawn in the order the reference constructs its solvers
// (recursive_solving.cc:301-327 + :96-131: a subgame, then depth-first each of its pseudo-leaves in the partial tree's
// BFS order = ascending full-tree id).  The weights emulate linear averaging: even iterations only, weight i / 2 + 1.
// std::discrete_distribution keeps no state between draws, so one instance stands for the reference's per-solver ones.
std::vector<int16_t> draw_act_iterations(const FullTree& ft, const Rules& g, int D, int num_iters, int seed) {
  if (num_iters > 32767) throw std::runtime_error("sampled recursion: num_iters above 32767");
  std::vector<int16_t> act((size_t)ft.N, (int16_t)-1);
  std::mt19937 gen(seed);
  std::vector<double> w;
  for (int i = 0; i < num_iters; ++i) w.push_back(i % 2 ? 0.0 : (i / 2. + 1));
  std::discrete_distribution<int> dist(w.begin(), w.end());
  std::vector<std::vector<int32_t>> scratch;  // two node lists per recursion depth
  std::function<void(int32_t, int)> visit = [&](int32_t root, int rd) {
    act[root] = (int16_t)dist(gen);
    if (scratch.size() < 2 * (size_t)rd + 2) scratch.resize(2 * (size_t)rd + 2);
    int cur = 2 * rd, nxt = 2 * rd + 1;
    scratch[cur].assign(1, root);
    for (int d = 0; d < D && !scratch[cur].empty(); ++d) {
      scratch[nxt].clear();
      for (size_t j = 0; j < scratch[cur].size(); ++j) {
        const int32_t n = scratch[cur][j];
        const int b = ft.bid[n];
        if (b == g.liar) continue;
        const int cnt = b < 0 ? g.A - 1 : g.A - 1 - b;
        for (int k = 0; k < cnt; ++k) scratch[nxt].push_back(ft.cb[n] + k);
      }
      std::swap(cur, nxt);
    }
    for (size_t j = 0; j < scratch[cur].size(); ++j) {  // visit() below may grow `scratch`: index, do not hold references
      const int32_t n = scratch[cur][j];
      if (ft.bid[n] != g.liar) visit(n, rd + 1);
    }
  };
  visit(0, 0);
  return act;
}

// The same with root_only (:318-320): the subgames below the root subgame run to the end of the game, so the recursion is two
// levels deep -- the root's draw, then one draw per pseudo-leaf of the root subgame in ascending full-tree id.  Returns the root's
// act_iteration; `leaf_act[i]` belongs to the i-th non-terminal node of depth D.
int draw_act_iterations_root_only(const FullTree& ft, const Rules& g, int D, int num_iters, int seed, std::vector<int32_t>* leaf_act) {
  std::mt19937 gen(seed);
  std::vector<double> w;
  for (int i = 0; i < num_iters; ++i) w.push_back(i % 2 ? 0.0 : (i / 2. + 1));
  std::discrete_distribution<int> dist(w.begin(), w.end());
  const int root = dist(gen);
  leaf_act->clear();
  if ((int)ft.lev_off.size() - 1 > D)
    for (int64_t n = ft.lev_off[D]; n < ft.lev_off[D + 1]; ++n)
      if (ft.bid[n] != g.liar) leaf_act->push_back(dist(gen));
  return root;
}

}  // namespace

struct StreamSolver {
  int device;
  Rules g;
  rbl_params p;
  FullTree ft;
  hipStream_t st = nullptr;
  DevBuf<int8_t> d_bid, d_matches;
  DevBuf<int32_t> d_cb;
  DevBuf<double> d_sigma, d_regrets, d_sums, d_avg, d_reach, d_val;
  DevBuf<double> d_samp, d_final;  // sampled repeats: the current repeat's strategy, the reach-weighted mean of all
  DevBuf<float> d_sstrat, d_sreach;
  DevBuf<double> d_ir;  // regret reports: regrets accumulated over a list of strategies (compute_immediate_regrets)
  int n_ir = 0;
  DevBuf<double> f_sigma, f_regrets;  // forest solve (root_only repeats): sigma and regrets of the subgames below the root's
  DevBuf<int32_t> f_sub, f_act, f_nodes;
  int f_sub_depth = -1;  // max_depth the subtree map f_sub was built for
  int n_samples = 0;
  double sample_seconds = 0;
  int iter = 0, num_steps[2] = {0, 0};
  double step_seconds = 0;

  StreamSolver(int dev, int dice, int faces, const rbl_params& params) : device(dev), g(dice, faces), p(params) {
    if (!p.use_cfr) throw std::runtime_error("stream solver: fictitious play is not available at this scale (use_cfr = 1)");
    if (g.dice > 8) throw std::runtime_error("stream solver: more than 8 dice");
    RBL_HIP_CHECK(hipSetDevice(device));
    RBL_HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    ft = build_full_tree(g);
    const int H = g.H;
    d_bid.upload(ft.bid, st);
    d_cb.upload(ft.cb, st);
    std::vector<int8_t> m((size_t)g.faces * H);
    for (int f = 0; f < g.faces; ++f)
      for (int h = 0; h < H; ++h) m[(size_t)f * H + h] = (int8_t)g.matches(h, f);
    d_matches.upload(m, st);
    const size_t eh = (size_t)std::max<int64_t>(1, ft.N - 1) * H, nh = (size_t)ft.N * H;
    d_sigma.alloc(eh);
    d_regrets.alloc(eh);
    d_sums.alloc(eh);
    d_avg.alloc(eh);
    d_reach.alloc(nh);
    d_val.alloc(std::max(nh, (size_t)0));
    // ctor: both players' uniform reach top-down through a [N][2][H] scratch (reach + val are exactly that large together)
    DevBuf<double> reach2;
    reach2.alloc(2 * nh);
    std::vector<double> b(2 * (size_t)H, 1.0 / H);
    RBL_HIP_CHECK(hipMemcpyAsync(reach2.p, b.data(), b.size() * sizeof(double), hipMemcpyHostToDevice, st));
    StepArgs a = args(0);
    for (int lev = 0; lev + 1 < nlev(); ++lev) {
      level(a, lev);
      hipLaunchKernelGGL(st_init_kernel, grid(a), dim3(256), 0, st, a, reach2.p);
    }
    RBL_HIP_CHECK(hipGetLastError());
    RBL_HIP_CHECK(hipStreamSynchronize(st));
  }
  ~StreamSolver() {
    if (st) (void)hipStreamDestroy(st);
  }
  int nlev() const { return (int)ft.lev_off.size() - 1; }
  StepArgs args(int trav) const {
    StepArgs a{};
    a.f_bid = d_bid.p;
    a.f_cb = d_cb.p;
    a.sigma = d_sigma.p;
    a.regrets = d_regrets.p;
    a.sums = d_sums.p;
    a.avg = d_avg.p;
    a.reach = d_reach.p;
    a.val = d_val.p;
    a.matches = d_matches.p;
    a.H = g.H;
    a.A = g.A;
    a.faces = g.faces;
    a.dice = g.dice;
    a.liar = g.liar;
    a.trav = trav;
    a.steps0 = num_steps[0];
    a.steps1 = num_steps[1];
    return a;
  }
  void level(StepArgs& a, int lev) const {
    a.n0 = ft.lev_off[lev];
    a.n1 = ft.lev_off[lev + 1];
    a.level = lev;
  }
  static dim3 grid(const StepArgs& a) { return dim3((unsigned)(((a.n1 - a.n0) * a.H + 255) / 256)); }
  SweepArgs sweep_args(const double* sigma, int trav) const {
    SweepArgs w{};
    w.f_bid = d_bid.p;
    w.f_cb = d_cb.p;
    w.sigma_full = sigma;
    w.reach = d_reach.p;
    w.val = d_val.p;
    w.matches = d_matches.p;
    w.H = g.H;
    w.A = g.A;
    w.faces = g.faces;
    w.dice = g.dice;
    w.liar = g.liar;
    w.trav = trav;
    return w;
  }
  void root_beliefs() {
    std::vector<double> b(g.H, 1.0 / g.H);
    RBL_HIP_CHECK(hipMemcpyAsync(d_reach.p, b.data(), g.H * sizeof(double), hipMemcpyHostToDevice, st));
    RBL_HIP_CHECK(hipStreamSynchronize(st));
  }

  void step(int t) {  // CFR::step (:577-664) on the full tree
    RBL_HIP_CHECK(hipSetDevice(device));
    const int k = num_steps[t];
    double pos = 1, neg = 1, strat = 1;
    const double s = k + 1;
    if (p.linear_update) {
      pos = neg = strat = s / (s + 1);
    } else if (p.dcfr) {
      pos = p.dcfr_alpha >= 5 ? 1 : std::pow(s, p.dcfr_alpha) / (std::pow(s, p.dcfr_alpha) + 1.);
      neg = p.dcfr_beta <= -5 ? 0 : std::pow(s, p.dcfr_beta) / (std::pow(s, p.dcfr_beta) + 1.);
      strat = std::pow(s / (s + 1), p.dcfr_gamma);
    }
    // opponent reach under sigma, top-down
    root_beliefs();
    SweepArgs w = sweep_args(d_sigma.p, t);
    for (int lev = 0; lev + 1 < nlev(); ++lev) {
      w.n0 = ft.lev_off[lev];
      w.n1 = ft.lev_off[lev + 1];
      w.level = lev;
      hipLaunchKernelGGL(br_reach_kernel, dim3((unsigned)(((w.n1 - w.n0) * g.H + 255) / 256)), dim3(256), 0, st, w);
    }
    StepArgs a = args(t);
    a.pos = pos;
    a.neg = neg;
    a.strat = strat;
    for (int lev = nlev() - 2; lev >= 0; --lev) {
      level(a, lev);
      w.n0 = a.n0;
      w.n1 = a.n1;
      w.level = lev;
      hipLaunchKernelGGL(terminal_value_kernel, dim3((unsigned)((w.n1 - w.n0 + 255) / 256)), dim3(256), 0, st, w);
      hipLaunchKernelGGL(st_value_kernel, grid(a), dim3(256), 0, st, a);
    }
    root_beliefs();  // the traverser's beliefs at the root; its reach under the new sigma follows top-down
    for (int lev = 0; lev + 1 < nlev(); ++lev) {
      level(a, lev);
      hipLaunchKernelGGL(st_sums_kernel, grid(a), dim3(256), 0, st, a);
    }
    RBL_HIP_CHECK(hipGetLastError());
    ++num_steps[t];
    ++iter;
  }

  void average() {
    StepArgs a = args(0);
    for (int lev = 0; lev + 1 < nlev(); ++lev) {
      level(a, lev);
      hipLaunchKernelGGL(st_average_kernel, grid(a), dim3(256), 0, st, a);
    }
    RBL_HIP_CHECK(hipGetLastError());
  }

  // compute_exploitability2 (:802-816) of `sigma` (an edge-indexed strategy on this device)
  void exploitability(const double* sigma, double out2[2]) {
    std::vector<double> root(g.H);
    for (int t = 0; t < 2; ++t) {
      root_beliefs();
      SweepArgs w = sweep_args(sigma, t);
      for (int lev = 0; lev + 1 < nlev(); ++lev) {
        w.n0 = ft.lev_off[lev];
        w.n1 = ft.lev_off[lev + 1];
        w.level = lev;
        hipLaunchKernelGGL(br_reach_kernel, dim3((unsigned)(((w.n1 - w.n0) * g.H + 255) / 256)), dim3(256), 0, st, w);
      }
      for (int lev = nlev() - 2; lev >= 0; --lev) {
        w.n0 = ft.lev_off[lev];
        w.n1 = ft.lev_off[lev + 1];
        w.level = lev;
        hipLaunchKernelGGL(terminal_value_kernel, dim3((unsigned)((w.n1 - w.n0 + 255) / 256)), dim3(256), 0, st, w);
        hipLaunchKernelGGL(br_value_kernel, dim3((unsigned)(((w.n1 - w.n0) * g.H + 255) / 256)), dim3(256), 0, st, w);
      }
      RBL_HIP_CHECK(hipGetLastError());
      RBL_HIP_CHECK(hipMemcpyAsync(root.data(), d_val.p, g.H * sizeof(double), hipMemcpyDeviceToHost, st));
      RBL_HIP_CHECK(hipStreamSynchronize(st));
      double sum = 0;
      for (int h = 0; h < g.H; ++h) sum += root[h];
      out2[t] = sum / g.H;
    }
  }


  // ---- "Recursive solving" of the reference tool (recursive_eval.cc:320-388) without dense tensors
  void sampled_reset() {
    RBL_HIP_CHECK(hipSetDevice(device));
    const size_t eh = (size_t)std::max<int64_t>(1, ft.N - 1) * g.H, nh = (size_t)ft.N * g.H;
    if (!d_samp.p) {
      d_samp.alloc(eh);
      d_final.alloc(eh);
      d_sstrat.alloc(eh);
      d_sreach.alloc(nh);
    }
    RBL_HIP_CHECK(hipMemsetAsync(d_sstrat.p, 0, eh * sizeof(float), st));
    RBL_HIP_CHECK(hipMemsetAsync(d_sreach.p, 0, nh * sizeof(float), st));
    RBL_HIP_CHECK(hipStreamSynchronize(st));
    n_samples = 0;
  }
  AccArgs acc_args() const {
    AccArgs a{};
    a.f_bid = d_bid.p;
    a.f_cb = d_cb.p;
    a.samp = d_samp.p;
    a.r0 = d_reach.p;
    a.r1 = d_val.p;
    a.sstrat = d_sstrat.p;
    a.sreach = d_sreach.p;
    a.fin = d_final.p;
    a.H = g.H;
    a.A = g.A;
    a.liar = g.liar;
    return a;
  }
  // one repeat: compute_sampled_strategy_recursive_to_leaf(game, params, net, seed, root_only = false) on the lanes of `e`
  // (its max_depth is the tool's mdp_depth, its net the value net), then summed_strategy / summed_reach
  void sampled_add(Engine& e, int seed) {
    if (e.device() != device || e.rules().dice != g.dice || e.rules().faces != g.faces)
      throw std::runtime_error("sampled_add: the engine is for another device or game");
    if (!d_samp.p) sampled_reset();
    const double t0 = now_s();
    const std::vector<int16_t> act = draw_act_iterations(ft, g, e.params().max_depth, e.params().num_iters, seed);
    recursive_fill(e, ft, d_cb.p, d_samp.p, 0, 1, &act, nullptr);
    RBL_HIP_CHECK(hipSetDevice(device));
    accumulate_sample();
    ++n_samples;
    sample_seconds += now_s() - t0;
  }
  // one repeat with root_only (recursive_solving.cc:318-320; recursive_eval --root_only): the root subgame on a lane of `e`
  // (depth max_depth, value net, stopped at its act_iteration), then EVERY subgame below it solved to the end of the game --
  // at 2 dice x 6 faces those are 276 trees of up to 4 M nodes, far beyond a lane: they are solved together as a FOREST on the
  // full tree's edge-indexed arrays, level-synchronously like the full-tree solver above, every node gated by the stop
  // iteration of the subtree it belongs to.  The subgames are independent, start together (so the traverser of iteration i
  // is i mod 2 for all, :666-670) and a stopped subtree's sigma IS its sampling strategy (get_sampling_strategy, :682-688).
  void sampled_add_root_only(Engine& e, int seed) {
    if (e.device() != device || e.rules().dice != g.dice || e.rules().faces != g.faces)
      throw std::runtime_error("sampled_add: the engine is for another device or game");
    if (!d_samp.p) sampled_reset();
    const double t0 = now_s();
    const int H = g.H, D = e.params().max_depth;
    // ONE subgame_params for every subgame of the repeat, as in the reference (recursive_solving.cc:301-327): the forest below
    // the root discounts with the ENGINE's parameters, never with this stream solver's own `p` (they may have been created
    // with different ones); what the forest's kernels do not implement is refused instead of silently diverging
    const rbl_params& ep = e.params();
    if (!ep.use_cfr) throw std::runtime_error("sampled_add(root_only): the forest solve is CFR only (engine params use_cfr = 0)");
    // (`optimistic` is read by fictitious play only, subgame_solving.cc:452: CFR ignores it, here as there; dcfr_gamma discounts
    //  sum_strategies, which a sampling strategy never reads)
    std::vector<int32_t> leaf_act;
    const int root_act = draw_act_iterations_root_only(ft, g, D, e.params().num_iters, seed, &leaf_act);
    // the root subgame: recursive_fill's first level with the root's own stop iteration; it scatters the sampling strategy
    // to the edges above depth D and leaves the frontier (nodes in ascending id, normalised beliefs) on the device
    std::vector<int16_t> act((size_t)ft.N, (int16_t)-1);
    act[0] = (int16_t)root_act;
    std::vector<int32_t> nodes;
    DevBuf<double> bel;
    recursive_fill(e, ft, d_cb.p, d_samp.p, 0, 1, &act, nullptr, 1, 1, &nodes, &bel);
    RBL_HIP_CHECK(hipSetDevice(device));
    const int64_t count = (int64_t)nodes.size();
    if (count != (int64_t)leaf_act.size()) throw std::runtime_error("sampled_add(root_only): frontier / draw count mismatch");
    if (count > 0) {
      const int nl = nlev();
      const size_t eh = (size_t)std::max<int64_t>(1, ft.N - 1) * H;
      if (!f_sigma.p) {
        f_sigma.alloc(eh);
        f_regrets.alloc(eh);
        f_sub.alloc((size_t)ft.N);
      }
      if (f_sub_depth != D) {  // the subtree map belongs to one root-subgame depth (an engine with another max_depth: rebuild)
        f_sub_depth = D;
        // subtree ids: the nodes of depth D number themselves (terminals included: they have no descendants), top-down
        for (int lev = D; lev + 1 <= nl; ++lev) {
          const int64_t n0 = ft.lev_off[lev], n1 = ft.lev_off[lev + 1];
          hipLaunchKernelGGL(forest_sub_kernel, dim3((unsigned)((n1 - n0 + 255) / 256)), dim3(256), 0, st, d_bid.p, d_cb.p, f_sub.p,
                             n0, n1, g.A, g.liar, lev == D ? 1 : 0);
        }
        RBL_HIP_CHECK(hipGetLastError());
      }
      // stop iteration per subtree id (= position in level D); terminals of that level never take part
      std::vector<int32_t> act_sub((size_t)(ft.lev_off[D + 1] - ft.lev_off[D]), 0);
      int max_act = 0;
      for (int64_t i = 0; i < count; ++i) {
        act_sub[(size_t)(nodes[(size_t)i] - ft.lev_off[D])] = leaf_act[(size_t)i];
        max_act = std::max(max_act, leaf_act[(size_t)i]);
      }
      f_act.upload(act_sub, st);
      f_nodes.upload(nodes, st);
      StepArgs a = args(0);
      a.sigma = f_sigma.p;
      a.regrets = f_regrets.p;
      for (int lev = D; lev + 1 < nl; ++lev) {
        level(a, lev);
        hipLaunchKernelGGL(forest_init_kernel, grid(a), dim3(256), 0, st, a);
      }
      RBL_HIP_CHECK(hipGetLastError());
      int steps[2] = {0, 0};
      for (int it = 0; it < max_act; ++it) {
        const int t = it % 2;
        const double sdisc = steps[t] + 1;
        double pos = 1, neg = 1;
        if (ep.linear_update) {
          pos = neg = sdisc / (sdisc + 1);
        } else if (ep.dcfr) {
          pos = ep.dcfr_alpha >= 5 ? 1 : std::pow(sdisc, ep.dcfr_alpha) / (std::pow(sdisc, ep.dcfr_alpha) + 1.);
          neg = ep.dcfr_beta <= -5 ? 0 : std::pow(sdisc, ep.dcfr_beta) / (std::pow(sdisc, ep.dcfr_beta) + 1.);
        }
        // opponent reach from the roots' beliefs, top-down under sigma
        hipLaunchKernelGGL(forest_root_reach_kernel, dim3((unsigned)((count * H + 255) / 256)), dim3(256), 0, st, f_nodes.p, bel.p,
                           d_reach.p, count, H, 1 - t);
        SweepArgs w = sweep_args(f_sigma.p, t);
        w.sub = f_sub.p;
        w.act_sub = f_act.p;
        w.iter = it;
        for (int lev = D; lev + 1 < nl; ++lev) {
          w.n0 = ft.lev_off[lev];
          w.n1 = ft.lev_off[lev + 1];
          w.level = lev;
          hipLaunchKernelGGL(br_reach_kernel, dim3((unsigned)(((w.n1 - w.n0) * H + 255) / 256)), dim3(256), 0, st, w);
        }
        a = args(t);
        a.sigma = f_sigma.p;
        a.regrets = f_regrets.p;
        a.pos = pos;
        a.neg = neg;
        a.strat = 1;
        a.sub = f_sub.p;
        a.act_sub = f_act.p;
        a.iter = it;
        for (int lev = nl - 2; lev >= D; --lev) {
          level(a, lev);
          w.n0 = a.n0;
          w.n1 = a.n1;
          w.level = lev;
          hipLaunchKernelGGL(terminal_value_kernel, dim3((unsigned)((w.n1 - w.n0 + 255) / 256)), dim3(256), 0, st, w);
          hipLaunchKernelGGL(st_value_kernel, grid(a), dim3(256), 0, st, a);
        }
        RBL_HIP_CHECK(hipGetLastError());
        ++steps[t];
      }
      // the subgames' sampling strategies: sigma of every edge below depth D (edge = child - 1, children of depth > D are the
      // tail of the BFS order)
      const int64_t e0 = ft.lev_off[D + 1] - 1;
      RBL_HIP_CHECK(hipMemcpyAsync(d_samp.p + e0 * H, f_sigma.p + e0 * H, (size_t)(ft.N - 1 - e0) * H * sizeof(double),
                                   hipMemcpyDeviceToDevice, st));
    }
    accumulate_sample();
    ++n_samples;
    sample_seconds += now_s() - t0;
  }
  void accumulate_sample() {  // summed_strategy / summed_reach of the repeat in d_samp (recursive_eval.cc:136-160)
    std::vector<double> b(g.H, 1.0 / g.H);
    RBL_HIP_CHECK(hipMemcpyAsync(d_reach.p, b.data(), g.H * sizeof(double), hipMemcpyHostToDevice, st));
    RBL_HIP_CHECK(hipMemcpyAsync(d_val.p, b.data(), g.H * sizeof(double), hipMemcpyHostToDevice, st));
    RBL_HIP_CHECK(hipStreamSynchronize(st));
    AccArgs a = acc_args();
    for (int lev = 0; lev + 1 < nlev(); ++lev) {
      a.n0 = ft.lev_off[lev];
      a.n1 = ft.lev_off[lev + 1];
      a.level = lev;
      hipLaunchKernelGGL(sm_accumulate_kernel, dim3((unsigned)(((a.n1 - a.n0) * g.H + 255) / 256)), dim3(256), 0, st, a);
    }
    RBL_HIP_CHECK(hipGetLastError());
    RBL_HIP_CHECK(hipStreamSynchronize(st));
  }
  // ---- report_regrets (recursive_eval.cc:28-53) without the list of dense strategies: the regrets of
  // compute_immediate_regrets accumulate on the device as the strategies come by
  void regrets_reset() {
    RBL_HIP_CHECK(hipSetDevice(device));
    const size_t eh = (size_t)std::max<int64_t>(1, ft.N - 1) * g.H;
    if (!d_ir.p) d_ir.alloc(eh);
    RBL_HIP_CHECK(hipMemsetAsync(d_ir.p, 0, eh * sizeof(double), st));
    RBL_HIP_CHECK(hipStreamSynchronize(st));
    n_ir = 0;
  }
  void regrets_add(const double* sigma, bool round32) {  // one more strategy of the list: both traversers' sweeps
    if (!d_ir.p) regrets_reset();
    for (int t = 0; t < 2; ++t) {
      root_beliefs();
      SweepArgs w = sweep_args(sigma, t);
      w.round32 = round32 ? 1 : 0;
      for (int lev = 0; lev + 1 < nlev(); ++lev) {
        w.n0 = ft.lev_off[lev];
        w.n1 = ft.lev_off[lev + 1];
        w.level = lev;
        hipLaunchKernelGGL(br_reach_kernel, dim3((unsigned)(((w.n1 - w.n0) * g.H + 255) / 256)), dim3(256), 0, st, w);
      }
      for (int lev = nlev() - 2; lev >= 0; --lev) {
        w.n0 = ft.lev_off[lev];
        w.n1 = ft.lev_off[lev + 1];
        w.level = lev;
        hipLaunchKernelGGL(terminal_value_kernel, dim3((unsigned)((w.n1 - w.n0 + 255) / 256)), dim3(256), 0, st, w);
        hipLaunchKernelGGL(ir_value_kernel, dim3((unsigned)(((w.n1 - w.n0) * g.H + 255) / 256)), dim3(256), 0, st, w, d_ir.p);
      }
      RBL_HIP_CHECK(hipGetLastError());
    }
    RBL_HIP_CHECK(hipStreamSynchronize(st));
    ++n_ir;
  }
  // first: [n_first][H] immediate regrets of the first nodes (the tool prints 20); sums: {sum over the nodes of depth < depth,
  // sum over the rest}, each node's vector_sum added in node order as the reference does
  void regrets_report(int depth, int n_first, double* first, double sums[2]) {
    if (!n_ir) throw std::runtime_error("regrets_report: no strategy was added");
    const int H = g.H;
    hipLaunchKernelGGL(ir_max_kernel, dim3((unsigned)((ft.N * H + 255) / 256)), dim3(256), 0, st, d_bid.p, d_cb.p, d_ir.p, d_val.p,
                       ft.N, H, g.A, g.liar, (double)n_ir);
    hipLaunchKernelGGL(ir_node_sum_kernel, dim3((unsigned)((ft.N + 255) / 256)), dim3(256), 0, st, d_val.p, d_reach.p, ft.N, H);
    RBL_HIP_CHECK(hipGetLastError());
    std::vector<double> node_sum((size_t)ft.N);
    RBL_HIP_CHECK(hipMemcpyAsync(node_sum.data(), d_reach.p, node_sum.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    const int64_t nf = std::min<int64_t>(n_first, ft.N);
    if (first && nf > 0)
      RBL_HIP_CHECK(hipMemcpyAsync(first, d_val.p, (size_t)nf * H * sizeof(double), hipMemcpyDeviceToHost, st));
    RBL_HIP_CHECK(hipStreamSynchronize(st));
    const int64_t split = depth < nlev() ? ft.lev_off[std::max(depth, 0)] : ft.N;  // nodes of depth < `depth` come first (BFS)
    double top = 0, bottom = 0;
    for (int64_t n = 0; n < ft.N; ++n) (n < split ? top : bottom) += node_sum[(size_t)n];
    sums[0] = top;
    sums[1] = bottom;
  }

  void sampled_final() {
    if (!n_samples) throw std::runtime_error("sampled_final: no repeats were added");
    AccArgs a = acc_args();
    for (int lev = 0; lev + 1 < nlev(); ++lev) {
      a.n0 = ft.lev_off[lev];
      a.n1 = ft.lev_off[lev + 1];
      a.level = lev;
      hipLaunchKernelGGL(sm_final_kernel, dim3((unsigned)(((a.n1 - a.n0) * g.H + 255) / 256)), dim3(256), 0, st, a);
    }
    RBL_HIP_CHECK(hipGetLastError());
  }

  // compute_ev2 (:975-982): {EV of s1 as player 0 against s2, -(EV of s2 as player 0 against s1)}
  void ev2(const double* s1, const double* s2, double out2[2]) {
    std::vector<double> root(g.H);
    for (int t = 0; t < 2; ++t) {
      const double* own = t == 0 ? s1 : s2;
      const double* opp = t == 0 ? s2 : s1;
      root_beliefs();
      SweepArgs w = sweep_args(opp, 0);
      for (int lev = 0; lev + 1 < nlev(); ++lev) {
        w.n0 = ft.lev_off[lev];
        w.n1 = ft.lev_off[lev + 1];
        w.level = lev;
        hipLaunchKernelGGL(br_reach_kernel, dim3((unsigned)(((w.n1 - w.n0) * g.H + 255) / 256)), dim3(256), 0, st, w);
      }
      for (int lev = nlev() - 2; lev >= 0; --lev) {
        w.n0 = ft.lev_off[lev];
        w.n1 = ft.lev_off[lev + 1];
        w.level = lev;
        hipLaunchKernelGGL(terminal_value_kernel, dim3((unsigned)((w.n1 - w.n0 + 255) / 256)), dim3(256), 0, st, w);
        hipLaunchKernelGGL(ev_value_kernel, dim3((unsigned)(((w.n1 - w.n0) * g.H + 255) / 256)), dim3(256), 0, st, w, own);
      }
      RBL_HIP_CHECK(hipGetLastError());
      RBL_HIP_CHECK(hipMemcpyAsync(root.data(), d_val.p, g.H * sizeof(double), hipMemcpyDeviceToHost, st));
      RBL_HIP_CHECK(hipStreamSynchronize(st));
      double sum = 0;
      for (int h = 0; h < g.H; ++h) sum += root[h];
      out2[t] = (t == 0 ? sum : -sum) / g.H;
    }
  }

  // edge-indexed [N-1][H] device array -> the reference's dense TreeStrategy [N][H][A] on the host (small games: tests)
  void dense(const double* edge_dev, double* out) {
    const int H = g.H, A = g.A;
    std::vector<double> e((size_t)(ft.N - 1) * H);
    RBL_HIP_CHECK(hipMemcpyAsync(e.data(), edge_dev, e.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    RBL_HIP_CHECK(hipStreamSynchronize(st));
    std::fill(out, out + (size_t)ft.N * H * A, 0.0);
    for (int64_t n = 0; n < ft.N; ++n) {
      const int b = ft.bid[n];
      if (b == g.liar) continue;
      int lo, hi;
      g.bid_range(b, &lo, &hi);
      for (int k = 0; k < hi - lo; ++k)
        for (int h = 0; h < H; ++h) out[((size_t)n * H + h) * A + lo + k] = e[(size_t)(ft.cb[n] + k - 1) * H + h];
    }
  }
};

}  // namespace rbl

// ---------------------------------------------------------------------------------------------- C ABI of the stream solver
struct rbl_stream {
  rbl::StreamSolver impl;
  rbl_stream(int dev, int d, int f, const rbl_params& p) : impl(dev, d, f, p) {}
};

namespace {
thread_local std::string g_stream_err;
template <class F>
int stream_guard(F&& f) {
  try {
    f();
    return 0;
  } catch (const std::exception& ex) {
    g_stream_err = ex.what();
    return 1;
  }
}
}  // namespace

extern "C" {
const char* rbl_stream_last_error(void) { return g_stream_err.c_str(); }
rbl_stream* rbl_stream_create(int device, int dice, int faces, const rbl_params* params) {
  rbl_stream* s = nullptr;
  if (!params || stream_guard([&] { s = new rbl_stream(device, dice, faces, *params); })) return nullptr;
  return s;
}
void rbl_stream_destroy(rbl_stream* s) { delete s; }
int64_t rbl_stream_num_nodes(rbl_stream* s) { return s ? s->impl.ft.N : -1; }
int rbl_stream_step(rbl_stream* s, int n_steps) {
  return stream_guard([&] {
    if (!s) throw std::runtime_error("null stream solver");
    for (int i = 0; i < n_steps; ++i) s->impl.step(s->impl.iter % 2);
    RBL_HIP_CHECK(hipStreamSynchronize(s->impl.st));
  });
}
int rbl_stream_exploitability(rbl_stream* s, double out[2]) {
  return stream_guard([&] {
    if (!s) throw std::runtime_error("null stream solver");
    s->impl.average();
    s->impl.exploitability(s->impl.d_avg.p, out);
  });
}
int rbl_stream_get(rbl_stream* s, int which, double* out) {
  return stream_guard([&] {
    if (!s) throw std::runtime_error("null stream solver");
    if (s->impl.ft.N > (1 << 22)) throw std::runtime_error("rbl_stream_get: the dense strategy of this game does not fit (that is the point)");
    switch (which) {
      case RBL_GET_AVERAGE: s->impl.average(); s->impl.dense(s->impl.d_avg.p, out); break;
      case RBL_GET_LAST: s->impl.dense(s->impl.d_sigma.p, out); break;
      case RBL_GET_REGRETS: s->impl.dense(s->impl.d_regrets.p, out); break;
      case RBL_GET_SUM: s->impl.dense(s->impl.d_sums.p, out); break;
      case RBL_GET_SAMPLED:
        if (!s->impl.n_samples) throw std::runtime_error("rbl_stream_get: no sampled repeat yet");
        s->impl.dense(s->impl.d_samp.p, out);
        break;
      case RBL_GET_FINAL: s->impl.sampled_final(); s->impl.dense(s->impl.d_final.p, out); break;
      default: throw std::runtime_error("rbl_stream_get: bad selector");
    }
  });
}
int rbl_stream_sampled_reset(rbl_stream* s) {
  return stream_guard([&] {
    if (!s) throw std::runtime_error("null stream solver");
    s->impl.sampled_reset();
  });
}
int rbl_stream_sampled_add(rbl_stream* s, rbl_engine* e, int seed) {
  return stream_guard([&] {
    if (!s || !e) throw std::runtime_error("null stream solver or engine");
    s->impl.sampled_add(rbl::engine_impl(e), seed);
  });
}
int rbl_stream_sampled_add_root_only(rbl_stream* s, rbl_engine* e, int seed) {
  return stream_guard([&] {
    if (!s || !e) throw std::runtime_error("null stream solver or engine");
    s->impl.sampled_add_root_only(rbl::engine_impl(e), seed);
  });
}
int rbl_stream_regrets_reset(rbl_stream* s) {
  return stream_guard([&] {
    if (!s) throw std::runtime_error("null stream solver");
    s->impl.regrets_reset();
  });
}
int rbl_stream_regrets_add(rbl_stream* s, int which) {
  return stream_guard([&] {
    if (!s) throw std::runtime_error("null stream solver");
    if (which == RBL_GET_LAST) {
      s->impl.regrets_add(s->impl.d_sigma.p, false);
    } else if (which == RBL_GET_SAMPLED) {
      if (!s->impl.n_samples) throw std::runtime_error("rbl_stream_regrets_add: no sampled repeat yet");
      s->impl.regrets_add(s->impl.d_samp.p, true);
    } else {
      throw std::runtime_error("rbl_stream_regrets_add: which must be RBL_GET_LAST or RBL_GET_SAMPLED");
    }
  });
}
int rbl_stream_regrets_report(rbl_stream* s, int depth, int n_first, double* first, double sums[2]) {
  return stream_guard([&] {
    if (!s || !sums) throw std::runtime_error("null stream solver or output");
    s->impl.regrets_report(depth, n_first, first, sums);
  });
}
int rbl_stream_sampled_eval(rbl_stream* s, double exploitability[2], double ev_of_full[2]) {
  return stream_guard([&] {
    if (!s) throw std::runtime_error("null stream solver");
    s->impl.sampled_final();
    s->impl.exploitability(s->impl.d_final.p, exploitability);
    s->impl.average();
    s->impl.ev2(s->impl.d_avg.p, s->impl.d_final.p, ev_of_full);
  });
}
}
