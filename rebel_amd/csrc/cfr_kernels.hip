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

// Loops over (node, hand) / (row, column) pairs are division-free: a thread fixes its column (hand) once and strides
// over rows; threads beyond rows_per_pass * width idle (2 of 128 for H = 6).
struct Grid2 {
  int col, row0, rpp;  // this thread's column, first row, rows per pass
  bool active;
  __device__ Grid2(int width) {
    rpp = blockDim.x / width;
    if (rpp < 1) rpp = 1;
    col = threadIdx.x % width;
    row0 = threadIdx.x / width;
    active = (int)threadIdx.x < rpp * width;
  }
};

// Reach of BOTH players under sigma, top-down by BFS level (compute_reach_probabilities, subgame_solving.cc:54-78).
// rho*[0] must hold the root beliefs.  Level d reads level d-1 only, so one barrier per level.  `only` >= 0 restricts
// the sweep to that player (the traverser's reach under the new sigma, :636-638).
__device__ __forceinline__ void sweep_reach(const LaneView& v, const double* sig, double* rho0, double* rho1, int only,
                                            int root_player, int H, const Grid2& gh) {
  for (int lev = 1; lev < v.nlev; ++lev) {
    const int n0 = v.lev_off[lev], n1 = v.lev_off[lev + 1];
    const int mover = root_player ^ ((lev - 1) & 1);  // mover of the parents of this level
    if (gh.active && gh.row0 < n1 - n0) {
      const int h = gh.col;
      for (int n = n0 + gh.row0; n < n1; n += gh.rpp) {
        const int p = v.parent[n] * H + h;
        const double s = sig[(n - 1) * H + h];
        if (only != 1) {
          const double up = rho0[p];
          rho0[n * H + h] = mover == 0 ? up * s : up;
        }
        if (only != 0) {
          const double up = rho1[p];
          rho1[n * H + h] = mover == 1 ? up * s : up;
        }
      }
    }
    __syncthreads();
  }
}

// HT / AT: compile-time number of hands / actions for the common games (0 = read them from the arguments); with them
// known every `* H`, `% H`, `/ Q` below is strength-reduced and the per-hand loops unroll.
// LDS: compile-time choice of where the lane's working set lives.  It has to be a template parameter: with a run-time
// `use_lds ? lds : scratch` every working-set pointer is generic and hipcc emits FLAT loads/stores with 64-bit address
// arithmetic for what should be ds_read/ds_write (and spills SGPRs holding the pointers).
template <int HT, int AT, bool LDS>
__global__ void cfr_step_kernel(const CfrArgs a) {
  extern __shared__ __align__(16) double lds[];
  const int lane = a.lane0 + blockIdx.x;
  // root de-duplication (selfplay_kernels.h): a served root lane has no rows and takes no part in any launch
  const int dedup_role = a.lane_skip ? a.lane_skip[lane] : 0;
  if (dedup_role == 1) return;
  const int H = HT > 0 ? HT : a.H, A = AT > 0 ? AT : a.A, Q = HT > 0 ? 2 + AT + 2 * HT : a.Q;
  const ShapeDev& sh = a.shapes[a.lane_shape[lane]];
  LaneView v;  // the tables the phases read: LDS copies when LDS, the global tables otherwise
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
  const Grid2 gh(H);
  long long* dbg = a.dbg ? a.dbg + (size_t)lane * 16 : nullptr;
  int dbg_k = 0;
#define RBL_STAMP()                                                    \
  do {                                                                 \
    if (dbg && threadIdx.x == 0) dbg[dbg_k] = (long long)clock64();    \
    ++dbg_k;                                                           \
  } while (0)
  RBL_STAMP();  // 0: start

  // working set: LDS when it fits, else a per-lane slab of global scratch (big trees: 2 dice x 6 faces, full trees)
  double* W;
  if constexpr (LDS)
    W = lds;
  else
    W = a.scratch + (size_t)lane * a.work_stride;
  double* rho0 = W;
  double* rho1 = rho0 + NH;
  double* val = rho1 + NH;
  double* sig = val + NH;
  double* reg = sig + EH;  // regrets staged like sigma: all irregular access happens here, global traffic is flat
  double* tmp = reg + EH;
  const int tmp_n = max(2 * v.L, v.L + v.T * (nbins + 1)) + 2;
  float* lvals = reinterpret_cast<float*>(tmp + tmp_n);  // [L][H] leaf values of the pending queries
  const int8_t* mtab = a.matches;
  const size_t lane_e = (size_t)lane * a.Emax * H;
  double* g_sig = a.sigma + lane_e;
  double* g_reg = a.regrets + lane_e;
  double* g_sum = a.sums + lane_e;
  const double* bel = a.beliefs + (size_t)lane * 2 * H;
  double* rmean = a.root_mean + (size_t)lane * 2 * H;

  // ---------------------------------------------------------------- stage everything the lane touches, in ONE sweep
  // (all loads of an iteration are in flight together): sigma, regrets, the leaf values of the pending queries, and --
  // LDS mode only -- the lane's tree tables and the match table.  Every phase below is a chain of dependent look-ups;
  // an L2 round trip per look-up, times ~10 phases, is what a lane's latency used to be made of.
  {
    int* it = reinterpret_cast<int*>(lvals + ((v.L * H + 1) & ~1));
    int* t_parent = it, *t_act = it + N, *t_cb = it + 2 * N, *t_ce = it + 3 * N, *t_depth = it + 4 * N;
    int* t_leaves = it + 5 * N, *t_terms = t_leaves + v.L;
    int8_t* t_match = reinterpret_cast<int8_t*>(t_terms + v.T);
    const bool br = a.mode == kModeBestResponse || a.mode == kModeFpStep || a.mode == kModeEvaluate;
    const bool step = a.mode == kModeStep, load_sig = a.mode != kModeInit;
    const float* gv = a.values + (size_t)row_off * H;
    const int n_lv = ((step || br) && LDS) ? v.L * H : 0, n_mt = LDS ? a.faces * H : 0, n_tab = LDS ? N : 0;
    const LaneView gt = v;  // global tables (source of the staging copy)
    const int n_all = max(max(load_sig ? EH : 0, n_tab), max(n_lv, n_mt));
    for (int i = threadIdx.x; i < n_all; i += blockDim.x) {
      if (load_sig && i < EH) {
        sig[i] = g_sig[i];
        if (step) reg[i] = g_reg[i];
      }
      if (i < n_lv) lvals[i] = gv[i];
      if (i < n_tab) {
        t_parent[i] = gt.parent[i];
        t_act[i] = gt.act[i];
        t_cb[i] = gt.cb[i];
        t_ce[i] = gt.ce[i];
        t_depth[i] = gt.depth[i];
        if (i < v.L) t_leaves[i] = gt.leaves[i];
        if (i < v.T) t_terms[i] = gt.terms[i];
      }
      if (i < n_mt) t_match[i] = a.matches[i];
    }
    for (int i = threadIdx.x; i < H; i += blockDim.x) {
      rho0[i] = bel[i];
      rho1[i] = bel[H + i];
    }
    __syncthreads();
    if constexpr (LDS) {
      v.parent = t_parent;
      v.act = t_act;
      v.cb = t_cb;
      v.ce = t_ce;
      v.depth = t_depth;
      v.leaves = t_leaves;
      v.terms = t_terms;
      mtab = t_match;
    }
  }

  if (a.mode == kModeInit) {
    // get_uniform_strategy (subgame_solving.cc:718-730): 1/#children on the edges out of every internal node
    if (gh.active)
      for (int n = 1 + gh.row0; n < N; n += gh.rpp) {
        const int p = v.parent[n], i = (n - 1) * H + gh.col;
        const double u = 1. / (v.ce[p] - v.cb[p]);
        sig[i] = u;
        g_sig[i] = u;
        g_reg[i] = 0.0;
      }
    for (int i = threadIdx.x; i < 2 * H; i += blockDim.x) rmean[i] = 0.0;
    __syncthreads();
  }
  RBL_STAMP();  // 1: staged

  // ---------------------------------------------------------------- reach of both players under sigma (:539)
  sweep_reach(v, sig, rho0, rho1, -1, root_player, H, gh);
  RBL_STAMP();  // 2: reach

  if (a.mode == kModeInit) {
    // sum_strategies = uniform * own reach on the mover's nodes (get_uniform_reach_weigted_strategy, :125-149)
    const bool snap_now = a.lane_act_iter && a.lane_act_iter[lane] == 0;
    double* snap = a.snapshot + lane_e;
    if (gh.active)
      for (int n = 1 + gh.row0; n < N; n += gh.rpp) {
        const int p = v.parent[n], h = gh.col, i = (n - 1) * H + h;
        const int mover = root_player ^ (v.depth[p] & 1);
        const double r = (mover == 0 ? rho0 : rho1)[p * H + h];
        g_sum[i] = sig[i] * r;
        if (snap_now) snap[i] = sig[i];
        if (dedup_role == 2) a.snap_all[i] = sig[i];  // sigma "after iteration 0": the uniform start
      }
  }

  if (a.mode == kModeStep || a.mode == kModeBestResponse || a.mode == kModeFpStep || a.mode == kModeEvaluate) {
    const bool fp = a.mode == kModeFpStep;
    const bool br = a.mode == kModeBestResponse || fp;
    const bool ev = a.mode == kModeEvaluate;  // policy evaluation: sum of sigma * value at the traverser's nodes, nothing updated
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
        const int8_t* m = mtab + face * H;
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
    RBL_STAMP();  // 3: leaf scalers

    // -------------------------------------------------------------- leaf and terminal values for the traverser
    if (gh.active) {
      const int h = gh.col;
      for (int k = gh.row0; k < v.L + v.T; k += gh.rpp) {
        if (k < v.L) {  // leaf_values(float) *= scalers(double), stored back as float (:268), read as double (:275)
          float x;
          if constexpr (LDS)
            x = lvals[k * H + h];
          else
            x = a.values[(size_t)(row_off + k) * H + h];
          val[v.leaves[k] * H + h] = (double)(float)((double)x * lscale[k]);
        } else {  // compute_expected_terminal_values (:80-98)
          const int zi = k - v.L, z = v.terms[zi];
          const int bid = v.act[v.parent[z]];
          const int qty = 1 + bid / a.faces, face = bid % a.faces;
          const double* b = tbins + zi * (nbins + 1);
          const int left = max(0, qty - (int)mtab[face * H + h]);
          const float pwin = (float)b[left];  // fp32 truncation (:785)
          double x = (double)pwin * 2 - b[nbins];
          // mover(z) is the player who did NOT call liar; inverse when that is not the traverser (:88-95)
          if ((root_player ^ (v.depth[z] & 1)) != t) x *= -1.0;
          val[z * H + h] = x;
        }
      }
    }
    __syncthreads();
    RBL_STAMP();  // 4: leaf values

    // -------------------------------------------------------------- bottom-up sweep (update_regrets, :542-574)
    // fused with regret matching (:619-634) and the regret discount (:639-650) of the same (node, hand) row.
    for (int lev = v.nlev - 2; lev >= 0; --lev) {
      const int n0 = v.lev_off[lev], n1 = v.lev_off[lev + 1];
      const bool mine = ((root_player ^ (lev & 1)) == t);
      if (gh.active) {
        const int h = gh.col;
        for (int n = n0 + gh.row0; n < n1; n += gh.rpp) {
          const int c0 = v.cb[n], c1 = v.ce[n];
          if (c0 == c1) continue;
          const double* vc = val + c0 * H + h;
          double x = 0.0;
          if (mine && br) {  // best response: first child, then strictly-greater children (subgame_solving.cc:336-344)
            const int cnt = c1 - c0;
            int best = 0;
            x = vc[0];
            for (int k = 1; k < cnt; ++k) {
              const double y = vc[k * H];
              if (y > x) {
                x = y;
                best = k;
              }
            }
            if (fp)  // br_strategies as a one-hot over the node's edges (:345-348), kept in the staged `reg` rows
              for (int k = 0; k < cnt; ++k) reg[(c0 - 1 + k) * H + h] = k == best ? 1.0 : 0.0;
          } else if (mine && ev) {  // compute_ev :955-962 (strategy * value, ascending actions)
            const double* sc = sig + (c0 - 1) * H + h;
            const int cnt = c1 - c0;
            for (int k = 0; k < cnt; ++k) x += sc[k * H] * vc[k * H];
          } else if (mine) {
            const double* sc = sig + (c0 - 1) * H + h;
            double* rc = reg + (c0 - 1) * H + h;
            const int cnt = c1 - c0;
            for (int k = 0; k < cnt; ++k) x += vc[k * H] * sc[k * H];
            double s = 0.0;
            for (int k = 0; k < cnt; ++k) {
              double r = rc[k * H];
              r += vc[k * H];
              r -= x;
              const double m = r > kEps ? r : kEps;  // std::max(regret, eps)
              s += m;
              sig[(c0 - 1 + k) * H + h] = m;
              rc[k * H] = r * (r > 0 ? a.pos : a.neg);
            }
            for (int k = 0; k < cnt; ++k) sig[(c0 - 1 + k) * H + h] = sc[k * H] / s;
          } else {
            const int cnt = c1 - c0;
            for (int k = 0; k < cnt; ++k) x += vc[k * H];
          }
          val[n * H + h] = x;
        }
      }
      __syncthreads();
    }

    RBL_STAMP();  // 5: bottom-up
    if ((br && !fp) || ev) {
      for (int h = threadIdx.x; h < H; h += blockDim.x) a.br_out[(size_t)lane * H + h] = val[h];
      return;
    }
    if (fp) {
      // ------------------------------------------------------------ fictitious play (FP::step, :433-476)
      for (int h = threadIdx.x; h < H; h += blockDim.x) {  // root_values_means (:441-452)
        double m = rmean[t * H + h];
        m += (val[h] - m) * a.alpha;
        rmean[t * H + h] = m;
      }
      __syncthreads();  // val[root] consumed; val is reused below for the updated sum rows
      // update_sum_strat (:401-431): traverser's reach under its best response, top-down from the lane's beliefs
      sweep_reach(v, reg, rho0, rho1, t, root_player, H, gh);
      double* nsum = val;  // edge-indexed scratch: E*H <= N*H
      if (gh.active)
        for (int n = 1 + gh.row0; n < N; n += gh.rpp) {
          const int p = v.parent[n], h = gh.col, i = (n - 1) * H + h;
          if ((root_player ^ (v.depth[p] & 1)) == t) {
            const double c = rho_t[p * H + h] * reg[i];  // traverser_beliefs * br
            double s = g_sum[i] + c;
            if (a.strat != 1.0) s *= a.strat;  // linear_update: sum *= (n+1)/(n+2) (:459-461)
            nsum[i] = s;
            reg[i] = c;  // last_strategies
          }
        }
      __syncthreads();
      {  // normalise rows of the traverser's nodes (:456-472); row sums sequential over the actions
        const bool snap_now = a.lane_act_iter && a.lane_act_iter[lane] == a.steps_after;
        double* snap = a.snapshot + lane_e;
        if (gh.active) {
          const int h = gh.col;
          for (int n = gh.row0; n < N; n += gh.rpp) {
            const int c0 = v.cb[n], c1 = v.ce[n];
            if (c0 == c1 || (root_player ^ (v.depth[n] & 1)) != t) continue;
            double s = 0.0;
            for (int c = c0; c < c1; ++c) s += nsum[(c - 1) * H + h];
            if (a.optimistic) {
              double sl = 0.0;
              for (int c = c0; c < c1; ++c) sl += reg[(c - 1) * H + h];
              s = s + sl;
            }
            for (int c = c0; c < c1; ++c) {
              const int i = (c - 1) * H + h;
              const double num = a.optimistic ? nsum[i] + reg[i] : nsum[i];
              const double av = num / s;
              sig[i] = av;
              g_sig[i] = av;
              g_sum[i] = nsum[i];
              g_reg[i] = reg[i];
            }
          }
        }
        __syncthreads();
        if (snap_now)
          for (int i = threadIdx.x; i < EH; i += blockDim.x) snap[i] = sig[i];
      }
      // reach of the traverser under the NEW average strategy, for the next queries
      for (int i = threadIdx.x; i < H; i += blockDim.x) (t == 0 ? rho0 : rho1)[i] = bel[t * H + i];
      __syncthreads();
      sweep_reach(v, sig, rho0, rho1, t, root_player, H, gh);
    } else {
    // -------------------------------------------------------------- running mean of the root values (:579-590)
    for (int h = threadIdx.x; h < H; h += blockDim.x) {
      double m = rmean[t * H + h];
      m += (val[h] - m) * a.alpha;
      rmean[t * H + h] = m;
    }

    // -------------------------------------------------------------- traverser's reach under the NEW sigma (:636-638)
    sweep_reach(v, sig, rho0, rho1, t, root_player, H, gh);
    RBL_STAMP();  // 6: new reach

    // -------------------------------------------------------------- sum_strategies (:651-657) + write back what changed
    {
      const bool snap_now = a.lane_act_iter && a.lane_act_iter[lane] == a.steps_after;
      double* snap = a.snapshot + lane_e;
      if (gh.active)
        for (int n = 1 + gh.row0; n < N; n += gh.rpp) {
          const int p = v.parent[n], h = gh.col, i = (n - 1) * H + h;
          if ((root_player ^ (v.depth[p] & 1)) == t) {
            double s = g_sum[i];
            s *= a.strat;
            s += rho_t[p * H + h] * sig[i];
            g_sum[i] = s;
            g_sig[i] = sig[i];
            g_reg[i] = reg[i];
          }
          if (snap_now) snap[i] = sig[i];
          if (dedup_role == 2) a.snap_all[(size_t)a.steps_after * a.Emax * H + i] = sig[i];
        }
    }
    }  // CFR (not FP)
  }

  RBL_STAMP();  // 7 (3 in init/query modes): write-back
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
    const Grid2 gq(Q);
    if (gq.active) {
      const int jq = gq.col;
      // column kind is fixed per thread: 0 = mover flag, 1 = traverser flag, 2 = one-hot of the last bid, 3/4 = beliefs
      const int pj = jq - 2 - A;
      const int kind = jq == 0 ? 0 : jq == 1 ? 1 : jq < 2 + A ? 2 : (pj >= H ? 4 : 3);
      const int hq = kind == 4 ? pj - H : pj;
      for (int k = gq.row0; k < v.L; k += gq.rpp) {
        const int n = v.leaves[k];
        float x;
        if (kind == 0)
          x = (float)(root_player ^ (v.depth[n] & 1));  // state.player_id
        else if (kind == 1)
          x = (float)a.next_trav;
        else if (kind == 2)
          x = (jq - 2 == v.act[n]) ? 1.0f : 0.0f;
        else
          x = (float)(((kind == 4 ? rho1 : rho0)[n * H + hq] + kEps) / qsum[2 * k + (kind == 4)]);
        q[k * Q + jq] = x;
      }
    }
  }
  __syncthreads();
  RBL_STAMP();  // 8: queries
#undef RBL_STAMP
}

// Test double of the value net (oracle/orc_api.h: orc_synthetic_net), elementwise, exact in IEEE float:
//   v[h] = ((0.5f*q[2+A+h] - 0.25f*q[2+A+H+h]) + 0.125f*(q[1]-q[0])) + 0.0625f*q[2 + h % A]
__global__ void synthetic_net_kernel(const float* __restrict__ queries, int64_t rows, int Q, float* __restrict__ out,
                                     int H, int A, const long long* __restrict__ range) {
  if (range) {  // device-side row range (see launch_mlp_forward)
    queries += range[0] * Q;
    out += range[0] * H;
    rows = range[1] - range[0];
  }
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

__global__ void split_queries_kernel(const float* __restrict__ canon, int A, int H, float* __restrict__ dyn, int DS,
                                     float* __restrict__ stat, int SS, int64_t rows, const long long* __restrict__ range) {
  const int Q = 2 + A + 2 * H, W = DS + SS;
  if (range) {
    canon += range[0] * Q;
    dyn += range[0] * DS;
    stat += range[0] * SS;
    rows = range[1] - range[0];
  }
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t r = i / W;
  if (r >= rows) return;
  const int j = (int)(i % W);
  const float* c = canon + r * Q;
  if (j < DS)
    dyn[r * DS + j] = j == 0 ? c[1] : (j <= 2 * H ? c[2 + A + j - 1] : 0.f);
  else {
    const int s = j - DS;
    stat[r * SS + s] = s == 0 ? c[0] : (s <= A ? c[2 + s - 1] : 0.f);
  }
}

__global__ void unsplit_queries_kernel(float* __restrict__ canon, int A, int H, const float* __restrict__ dyn, int DS,
                                       const float* __restrict__ stat, int SS, int64_t rows) {
  const int Q = 2 + A + 2 * H;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t r = i / Q;
  if (r >= rows) return;
  const int k = (int)(i % Q);
  canon[i] = k == 0 ? stat[r * SS] : (k == 1 ? dyn[r * DS] : (k < 2 + A ? stat[r * SS + 1 + k - 2] : dyn[r * DS + 1 + k - 2 - A]));
}

// LaneRec of every launch slot (cfr_kernels.h): the shape's template with the lane's descriptors filled in; one thread per slot,
// the record written as four 16-byte pieces
__global__ void __launch_bounds__(256) lane_rec_kernel(const LaneRec* __restrict__ shape_rec, const int* __restrict__ lane_shape,
                                                       const int* __restrict__ lane_player, const int* __restrict__ lane_row,
                                                       const int* __restrict__ lane_act, const int* __restrict__ lane_order,
                                                       const int* __restrict__ lane_skip, int n, LaneRec* __restrict__ out) {
  const int slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= n) return;
  const int lane = lane_order ? lane_order[slot] : slot;
  LaneRec r = shape_rec[lane_shape[lane]];
  r.lane = lane;
  r.root_player = lane_player[lane];
  r.row_off = lane_row[lane];
  r.act_iter = lane_act ? lane_act[lane] : -1;
  const int sk = lane_skip ? lane_skip[lane] : 0;
  r.flags = sk == 1 ? kRecSkip : (sk == 2 ? kRecRep : 0);
  out[slot] = r;
}

void launch_lane_rec(const LaneRec* shape_rec, const int* lane_shape, const int* lane_player, const int* lane_row,
                     const int* lane_act, const int* lane_order, const int* lane_skip, int n, LaneRec* out, hipStream_t stream) {
  if (n <= 0) return;
  hipLaunchKernelGGL(lane_rec_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, shape_rec, lane_shape, lane_player, lane_row,
                     lane_act, lane_order, lane_skip, n, out);
}

void launch_split_queries(const float* canon, int A, int H, float* dyn, int DS, float* stat, int SS, int64_t rows,
                          hipStream_t stream, const long long* range) {
  if (rows <= 0) return;
  const int64_t n = rows * (DS + SS);
  hipLaunchKernelGGL(split_queries_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, canon, A, H, dyn, DS,
                     stat, SS, rows, range);
}

void launch_unsplit_queries(float* canon, int A, int H, const float* dyn, int DS, const float* stat, int SS, int64_t rows,
                            hipStream_t stream) {
  if (rows <= 0) return;
  const int64_t n = rows * (2 + A + 2 * H);
  hipLaunchKernelGGL(unsplit_queries_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, canon, A, H, dyn, DS,
                     stat, SS, rows);
}

void launch_synthetic_net(const float* queries, int64_t rows, int Q, float* out, int H, int A, hipStream_t stream,
                          const long long* range) {
  if (rows <= 0) return;
  const int64_t n = rows * H;
  hipLaunchKernelGGL(synthetic_net_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, queries, rows, Q,
                     out, H, A, range);
}

void launch_cfr(const CfrArgs& a, int B, int block, size_t lds_bytes, hipStream_t stream) {
#define RBL_CFR(HT_, AT_)                                                                                         \
  do {                                                                                                            \
    if (a.use_lds)                                                                                                \
      hipLaunchKernelGGL((cfr_step_kernel<HT_, AT_, true>), dim3(B), dim3(block), lds_bytes, stream, a);          \
    else                                                                                                          \
      hipLaunchKernelGGL((cfr_step_kernel<HT_, AT_, false>), dim3(B), dim3(block), 0, stream, a);                 \
  } while (0)
  if (a.H == 6 && a.A == 13) RBL_CFR(6, 13);        // 1 die x 6 faces
  else if (a.H == 4 && a.A == 9) RBL_CFR(4, 9);     // 1 die x 4 faces
  else if (a.H == 9 && a.A == 13) RBL_CFR(9, 13);   // 2 dice x 3 faces
  else if (a.H == 5 && a.A == 11) RBL_CFR(5, 11);   // 1 die x 5 faces
  else RBL_CFR(0, 0);
#undef RBL_CFR
}

}  // namespace rbl
