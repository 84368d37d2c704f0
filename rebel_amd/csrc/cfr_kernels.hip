// rebel_amd/csrc/cfr_kernels.hip -- the batched CFR subgame-solver step for gfx950 (MI355X).
//
// One workgroup = one lane (one subgame); thousands of lanes per launch.  A launch performs, for every lane,
//   [consume the leaf values the value net produced for the pending queries]  -> CFR::step(trav)
//   -> [write the queries of the NEXT step], so that one CFR iteration = one launch of this kernel + one net forward.
//
// What it computes is CFR::step of the reference (/root/reference/csrc/liars_dice/subgame_solving.cc:577-664) with
// PartialTreeTraverser's reach sweep (:54-78), query encoding (:104-123, :253-269), terminal payoffs (:80-98,
// :285-293, :765-789) and regret update (:538-575).  How it computes it is ours: edge-indexed [edge][hand] arrays,
// the lane's working set (reach of both players, node values, sigma) staged in LDS, level-synchronous sweeps with one
// thread per (node, hand) that walks its actions sequentially.
//
// Arithmetic contract (tests/test_cfr_parity.py): fp64 state, every reduction sequential in ascending index order,
// no FMA contraction (this file is compiled with -ffp-contract=off), float truncations where the reference has them
// (:785 win probability, :268 leaf values, :109-120 queries).  Under that contract the results are bit-identical to the
// reference's for the same leaf values.
#include "cfr_kernels.h"

namespace rbl {

namespace {

constexpr double kEps = 1e-80;  // kReachSmoothingEps == kRegretSmoothingEps (subgame_solving.h:34-36)

struct LaneView {
  int N, L, T, nlev;
  const int* lev_off;
  const int* parent;
  const int* act;
  const int* cb;
  const int* ce;
  const int* depth;
  const int* leaves;
  const int* terms;
};

// Reach of `player` under sigma, top-down by BFS level (compute_reach_probabilities, subgame_solving.cc:54-78).
// rho[0] must hold the root beliefs.  Level d reads level d-1 only, so one barrier per level.
__device__ __forceinline__ void sweep_reach(const LaneView& v, const double* sig, double* rho, int player,
                                            int root_player, int H) {
  for (int lev = 1; lev < v.nlev; ++lev) {
    const int n0 = v.lev_off[lev], cnt = (v.lev_off[lev + 1] - n0) * H;
    const bool own = ((root_player ^ ((lev - 1) & 1)) == player);  // mover of the parents of this level
    for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
      const int n = n0 + i / H, h = i % H;
      const double up = rho[v.parent[n] * H + h];
      rho[n * H + h] = own ? up * sig[(n - 1) * H + h] : up;
    }
    __syncthreads();
  }
}

__global__ void cfr_step_kernel(const CfrArgs a) {
  extern __shared__ __align__(16) double lds[];
  const int lane = blockIdx.x;
  const int H = a.H, A = a.A, Q = a.Q;
  const ShapeDev& sh = a.shapes[a.lane_shape[lane]];
  LaneView v;
  v.N = sh.N;
  v.L = sh.L;
  v.T = sh.T;
  v.nlev = sh.nlev;
  v.lev_off = sh.lev_off;
  v.parent = a.parent + sh.node_off;
  v.act = a.act + sh.node_off;
  v.cb = a.cb + sh.node_off;
  v.ce = a.ce + sh.node_off;
  v.depth = a.depth + sh.node_off;
  v.leaves = a.leaves + sh.leaf_off;
  v.terms = a.terms + sh.term_off;
  const int N = v.N, E = N - 1, EH = E * H, NH = N * H;
  const int root_player = a.lane_root_player[lane];
  const int row_off = a.lane_row_off[lane];
  const int nbins = 2 * a.dice + 1;

  // working set: LDS when it fits, else a per-lane slab of global scratch (big trees: 2 dice x 6 faces, full trees)
  double* W = a.use_lds ? lds : a.scratch + (size_t)lane * a.work_stride;
  double* rho0 = W;
  double* rho1 = rho0 + NH;
  double* val = rho1 + NH;
  double* sig = val + NH;
  double* tmp = sig + EH;

  const size_t lane_e = (size_t)lane * a.Emax * H;
  double* g_sig = a.sigma + lane_e;
  double* g_reg = a.regrets + lane_e;
  double* g_sum = a.sums + lane_e;
  const double* bel = a.beliefs + (size_t)lane * 2 * H;
  double* rmean = a.root_mean + (size_t)lane * 2 * H;

  // ---------------------------------------------------------------- stage sigma (or build the uniform one)
  if (a.mode == kModeInit) {
    // get_uniform_strategy (subgame_solving.cc:718-730): 1/#children on the edges out of every internal node
    for (int i = threadIdx.x; i < EH; i += blockDim.x) {
      const int n = 1 + i / H;
      const int p = v.parent[n];
      const double u = 1. / (v.ce[p] - v.cb[p]);
      sig[i] = u;
      g_sig[i] = u;
      g_reg[i] = 0.0;
    }
    for (int i = threadIdx.x; i < 2 * H; i += blockDim.x) rmean[i] = 0.0;
  } else {
    for (int i = threadIdx.x; i < EH; i += blockDim.x) sig[i] = g_sig[i];
  }
  for (int i = threadIdx.x; i < H; i += blockDim.x) {
    rho0[i] = bel[i];
    rho1[i] = bel[H + i];
  }
  __syncthreads();

  // ---------------------------------------------------------------- reach of both players under sigma (:539)
  sweep_reach(v, sig, rho0, 0, root_player, H);
  sweep_reach(v, sig, rho1, 1, root_player, H);

  if (a.mode == kModeInit) {
    // sum_strategies = uniform * own reach on the mover's nodes (get_uniform_reach_weigted_strategy, :125-149)
    for (int i = threadIdx.x; i < EH; i += blockDim.x) {
      const int n = 1 + i / H, h = i % H;
      const int p = v.parent[n];
      const int mover = root_player ^ (v.depth[p] & 1);
      const double r = (mover == 0 ? rho0 : rho1)[p * H + h];
      g_sum[i] = sig[i] * r;
    }
    if (a.lane_act_iter && a.lane_act_iter[lane] == 0) {
      double* snap = a.snapshot + lane_e;
      for (int i = threadIdx.x; i < EH; i += blockDim.x) snap[i] = sig[i];
    }
  }

  if (a.mode == kModeStep) {
    const int t = a.trav;
    double* rho_t = t == 0 ? rho0 : rho1;
    const double* rho_o = t == 0 ? rho1 : rho0;
    double* lscale = tmp;           // [L]
    double* tbins = tmp + v.L;      // [T][nbins + 1]  (suffix-summed match counts, then the plain sum)

    // -------------------------------------------------------------- leaf scalers and terminal match histograms
    for (int i = threadIdx.x; i < v.L + v.T; i += blockDim.x) {
      if (i < v.L) {  // query_value_net (:257-268): sum of opponent reach at the pseudo-leaf, sequential
        const double* r = rho_o + v.leaves[i] * H;
        double s = 0;
        for (int h = 0; h < H; ++h) s += r[h];
        lscale[i] = s;
      } else {  // compute_win_probability (:765-789), the belief-histogram part
        const int zi = i - v.L, z = v.terms[zi];
        const int bid = v.act[v.parent[z]];
        const int face = bid % a.faces;
        const int8_t* m = a.matches + face * H;
        const double* r = rho_o + z * H;
        double* b = tbins + zi * (nbins + 1);
        for (int k = 0; k < nbins; ++k) b[k] = 0.0;
        double s = 0;
        for (int h = 0; h < H; ++h) {
          b[m[h]] += r[h];
          s += r[h];
        }
        for (int k = nbins - 2; k >= 0; --k) b[k] += b[k + 1];
        b[nbins] = s;
      }
    }
    __syncthreads();

    // -------------------------------------------------------------- leaf and terminal values for the traverser
    for (int i = threadIdx.x; i < (v.L + v.T) * H; i += blockDim.x) {
      const int k = i / H, h = i % H;
      if (k < v.L) {  // leaf_values(float) *= scalers(double), stored back as float (:268), read as double (:275)
        const float x = a.values[(size_t)(row_off + k) * H + h];
        val[v.leaves[k] * H + h] = (double)(float)((double)x * lscale[k]);
      } else {  // compute_expected_terminal_values (:80-98)
        const int zi = k - v.L, z = v.terms[zi];
        const int bid = v.act[v.parent[z]];
        const int qty = 1 + bid / a.faces, face = bid % a.faces;
        const double* b = tbins + zi * (nbins + 1);
        const int left = max(0, qty - (int)a.matches[face * H + h]);
        const float pwin = (float)b[left];  // fp32 truncation (:785)
        double x = (double)pwin * 2 - b[nbins];
        // mover(z) is the player who did NOT call liar; inverse when that is not the traverser (:88-95)
        if ((root_player ^ (v.depth[z] & 1)) != t) x *= -1.0;
        val[z * H + h] = x;
      }
    }
    __syncthreads();

    // -------------------------------------------------------------- bottom-up sweep (update_regrets, :542-574)
    // fused with regret matching (:619-634) and the regret discount (:639-650) of the same (node, hand) row.
    for (int lev = v.nlev - 2; lev >= 0; --lev) {
      const int n0 = v.lev_off[lev], cnt = (v.lev_off[lev + 1] - n0) * H;
      const bool mine = ((root_player ^ (lev & 1)) == t);
      for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
        const int n = n0 + i / H, h = i % H;
        const int c0 = v.cb[n], c1 = v.ce[n];
        if (c0 == c1) continue;
        double x = 0.0;
        if (mine) {
          for (int c = c0; c < c1; ++c) x += val[c * H + h] * sig[(c - 1) * H + h];
          double s = 0.0;
          for (int c = c0; c < c1; ++c) {
            const int e = (c - 1) * H + h;
            double r = g_reg[e];
            r += val[c * H + h];
            r -= x;
            const double m = r > kEps ? r : kEps;  // std::max(regret, eps)
            s += m;
            sig[e] = m;
            g_reg[e] = r * (r > 0 ? a.pos : a.neg);
          }
          for (int c = c0; c < c1; ++c) {
            const int e = (c - 1) * H + h;
            sig[e] = sig[e] / s;
          }
        } else {
          for (int c = c0; c < c1; ++c) x += val[c * H + h];
        }
        val[n * H + h] = x;
      }
      __syncthreads();
    }

    // -------------------------------------------------------------- running mean of the root values (:579-590)
    for (int h = threadIdx.x; h < H; h += blockDim.x) {
      double m = rmean[t * H + h];
      m += (val[h] - m) * a.alpha;
      rmean[t * H + h] = m;
    }

    // -------------------------------------------------------------- traverser's reach under the NEW sigma (:636-638)
    sweep_reach(v, sig, rho_t, t, root_player, H);

    // -------------------------------------------------------------- sum_strategies (:651-657) + write sigma back
    for (int i = threadIdx.x; i < EH; i += blockDim.x) {
      const int n = 1 + i / H, h = i % H;
      const int p = v.parent[n];
      if ((root_player ^ (v.depth[p] & 1)) == t) {
        double s = g_sum[i];
        s *= a.strat;
        s += rho_t[p * H + h] * sig[i];
        g_sum[i] = s;
        g_sig[i] = sig[i];
      }
    }
    if (a.lane_act_iter && a.lane_act_iter[lane] == a.steps_after) {
      double* snap = a.snapshot + lane_e;
      for (int i = threadIdx.x; i < EH; i += blockDim.x) snap[i] = sig[i];
    }
  }

  // ---------------------------------------------------------------- queries for the next step (:253-269, :104-123)
  if (a.next_trav >= 0 && v.L > 0) {
    double* qsum = tmp;  // [L][2]
    __syncthreads();     // tmp (lscale/tbins) is dead; rho_t final
    for (int i = threadIdx.x; i < 2 * v.L; i += blockDim.x) {
      const double* r = (i & 1 ? rho1 : rho0) + v.leaves[i >> 1] * H;
      double s = 0;
      for (int h = 0; h < H; ++h) s += r[h] + kEps;  // normalize_probabilities_safe (util.h:68-78)
      qsum[i] = s;
    }
    __syncthreads();
    float* q = a.queries + (size_t)row_off * Q;
    for (int i = threadIdx.x; i < v.L * Q; i += blockDim.x) {
      const int k = i / Q, j = i % Q;
      const int n = v.leaves[k];
      float x;
      if (j == 0) {
        x = (float)(root_player ^ (v.depth[n] & 1));  // state.player_id
      } else if (j == 1) {
        x = (float)a.next_trav;
      } else if (j < 2 + A) {
        x = (j - 2 == v.act[n]) ? 1.0f : 0.0f;
      } else {
        const int pj = j - 2 - A;
        const int p = pj >= H, h = pj - p * H;
        x = (float)(((p ? rho1 : rho0)[n * H + h] + kEps) / qsum[2 * k + p]);
      }
      q[i] = x;
    }
  }
}

// Test double of the value net (oracle/orc_api.h: orc_synthetic_net), elementwise, exact in IEEE float:
//   v[h] = ((0.5f*q[2+A+h] - 0.25f*q[2+A+H+h]) + 0.125f*(q[1]-q[0])) + 0.0625f*q[2 + h % A]
__global__ void synthetic_net_kernel(const float* __restrict__ queries, int64_t rows, int Q, float* __restrict__ out,
                                     int H, int A) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * H) return;
  const int64_t r = i / H;
  const int h = (int)(i % H);
  const float* q = queries + r * Q;
  const float a = 0.5f * q[2 + A + h];
  const float b = 0.25f * q[2 + A + H + h];
  const float c = 0.125f * (q[1] - q[0]);
  const float d = 0.0625f * q[2 + h % A];
  out[i] = ((a - b) + c) + d;
}

}  // namespace

void launch_synthetic_net(const float* queries, int64_t rows, int Q, float* out, int H, int A, hipStream_t stream) {
  if (rows <= 0) return;
  const int64_t n = rows * H;
  hipLaunchKernelGGL(synthetic_net_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, queries, rows, Q,
                     out, H, A);
}

void launch_cfr(const CfrArgs& a, int B, int block, size_t lds_bytes, hipStream_t stream) {
  hipLaunchKernelGGL(cfr_step_kernel, dim3(B), dim3(block), a.use_lds ? lds_bytes : 0, stream, a);
}

}  // namespace rbl
