// rebel_amd/csrc/cfr_rows_kernel.hip -- CFR::step for the common games, one thread per tree ROW.
//
// cfr_kernels.hip maps one thread to a (node, hand) pair; rocprofv3 PMC showed that kernel issue-bound (SIMDs issuing
// 84 % of the time, ~2 100 instructions per wave, a third of them scalar control flow) for ~5 kFLOP of real work per
// lane: almost all of it is index arithmetic and loop control around single fp64 operations.  Here a thread owns a
// whole row of H hands of one node / edge (H, A, dice, faces are template parameters, so the per-hand loops unroll
// into straight-line fp64 code on registers, rows move as 16-byte LDS transfers), which amortises every table
// look-up and address computation over H elements, and folds whole phases together:
//   * a leaf's reach is never stored: the thread that computes it uses it on the spot for the leaf value (scale sum /
//     terminal match histogram, both in-thread), and recomputes it for the query row at the end;
//   * reach rows are kept for nodes with children only (irank table), which shrinks the lane's LDS footprint;
//   * regret update, regret matching and normalisation of a level run as row-parallel passes over the level's edges
//     with the sequential-in-action reductions (node value x, row sum s) as row passes over the level's nodes.
// Arithmetic is operation-for-operation what cfr_kernels.hip does (same operands, same order, -ffp-contract=off), so
// the bit-exactness contract with the reference (subgame_solving.cc:538-664) is unchanged; tests/test_cfr_parity.py
// runs against this kernel.  Only kModeStep of LDS-resident lanes runs here; init / query-only / best-response / FP
// modes and big trees stay on the generic kernel (they share the global state layout).
#include <type_traits>

#include <mutex>
#include <stdexcept>
#include <string>

#include "cfr_kernels.h"

namespace rbl {

namespace {

constexpr double kEps = 1e-80;

template <int H>
struct Row {
  double v[H];
};

template <int H>
__device__ __forceinline__ Row<H> load_row(const double* p) {
  Row<H> r;
#pragma unroll
  for (int h = 0; h < H; ++h) r.v[h] = p[h];
  return r;
}
template <int H>
__device__ __forceinline__ void store_row(double* p, const Row<H>& r) {
#pragma unroll
  for (int h = 0; h < H; ++h) p[h] = r.v[h];
}

// GS ("global state"): for games whose per-lane state does not fit LDS (2 dice x 6 faces: H = 36, root tree N = 325, 93 KB
// per strategy array) the kernel keeps only what is accessed irregularly -- node values [N][H], the reach rows of nodes
// with children, the tree tables -- in LDS (124 KB at the root: one lane per CU) and works on sigma / regrets IN PLACE in
// the lane's global slab: every element of them is touched a bounded number of times per step by the thread that owns
// its row, in 16-byte pieces, and the rows of a lane sit in one contiguous stretch that stays in L2 for the whole step.
// Leaf values are read from the net's output rows directly and query rows are written straight to the exchange buffer
// by the pseudo-leaf's thread.  Same operations, same order: bit-exactness is unchanged.
template <int H, int A, int DICE, int FACES, bool GS>
__global__ void __launch_bounds__(GS ? 256 : 128) cfr_rows_kernel(const CfrArgs a) {
  extern __shared__ __align__(16) double lds[];
  constexpr int Q = 2 + A + 2 * H, NB = 2 * DICE + 1;
  const int lane = (GS && a.lane_order)
                       ? ((const int __attribute__((address_space(4)))*)a.lane_order)[a.lane0 + blockIdx.x]
                       : a.lane0 + (int)blockIdx.x;
  const int tid = threadIdx.x, nthr = blockDim.x;
  // The lane's strategy and regrets are requested BEFORE its shape is known (their addresses only depend on the lane; the
  // bound is the slab size, the real extent is applied at the LDS store): the shape look-up (lane -> shape id -> shape
  // record -> tables) is a chain of dependent loads, and the state comes from Infinity Cache / HBM, not from L2.
  // (even H: rows are 16-byte aligned, the state moves as double2)
  constexpr int kW = H % 2 == 0 ? 2 : 1;                        // doubles per transfer
  constexpr int kB = H % 2 == 0 ? 3 : (H <= 6 ? 5 : 7);         // strides of blockDim = 128 that cover the depth-2 root tree
  typedef double d2_t __attribute__((ext_vector_type(2)));
  typedef typename std::conditional<kW == 2, d2_t, double>::type dw;
  dw s_[GS ? 1 : kB], r_[GS ? 1 : kB];
  if constexpr (!GS) {
    const size_t le = (size_t)lane * a.Emax * H;
    const int cap = a.Emax * H / kW;
    const dw* gs = reinterpret_cast<const dw*>(a.sigma + le);
    const dw* gr = reinterpret_cast<const dw*>(a.regrets + le);
#pragma unroll
    for (int u = 0; u < kB; ++u) {  // index clamped instead of predicated: no exec juggling, no phi copies
      const int i = min(tid + u * nthr, cap - 1);
      s_[u] = gs[i];
      r_[u] = gr[i];
    }
  }
  // Lane descriptors and the shape record are read-only for the whole launch: read them through the constant address
  // space so that they are SCALAR loads.  As plain global pointers hipcc fetched sh.node_off with a vector load +
  // readfirstlane, and the vmcnt(0) in front of the readfirstlane also waited for the speculative state loads above
  // (vmcnt retires in order): the table addresses could not even be formed before the state had arrived.
  typedef const int __attribute__((address_space(4)))* cint_p;
  typedef const ShapeDev __attribute__((address_space(4)))* cshape_p;
  const cshape_p shc = (cshape_p)a.shapes + ((cint_p)a.lane_shape)[lane];
  const int N = shc->N, E = N - 1, L = shc->L, NI = shc->NI, nlev = shc->nlev, node_off = shc->node_off;
  const int root_player = ((cint_p)a.lane_root_player)[lane], row_off = ((cint_p)a.lane_row_off)[lane];
  const int t = a.trav, opp = 1 - t;

  // ---- LDS layout (doubles): rho0, rho1, yrow [NI][H] | sig [E][H] | val [N][H] | reg [E][H] | leaf values | tables
  // GS layout: rho0, rho1, yrow [NI][H] | val [N][H] | tables  (sigma, regrets, leaf values, query rows stay in global memory)
  const size_t lane_e = (size_t)lane * a.Emax * H;
  double* g_sig = a.sigma + lane_e;
  double* g_reg = a.regrets + lane_e;
  double* rho0 = lds;
  double* rho1 = rho0 + NI * H;
  double* yrow = rho1 + NI * H;  // refined reciprocals of the regret-matching row sums (see the normalisation pass)
  double* sig;
  double* val;
  double* reg;
  const float* lvals;
  int* tb;
  if constexpr (GS) {
    sig = g_sig;
    reg = g_reg;
    val = yrow + NI * H;
    lvals = a.values + (size_t)row_off * H;
    tb = reinterpret_cast<int*>(val + N * H);
  } else {
    sig = yrow + NI * H;
    val = sig + E * H;
    reg = val + N * H;
    float* lv = reinterpret_cast<float*>(reg + E * H);
    lvals = lv;
    tb = reinterpret_cast<int*>(lv + ((L * H + 3) & ~3));
  }
  int* t_parent = tb, *t_act = tb + N, *t_cb = tb + 2 * N, *t_ce = tb + 3 * N, *t_depth = tb + 4 * N;
  int* t_irank = tb + 5 * N, *t_lrow = tb + 6 * N;
  int8_t* t_match = reinterpret_cast<int8_t*>(tb + 7 * N);
  float* qstage = reinterpret_cast<float*>(val);  // [L][Q], aliases val + reg once both are dead (host checks the size)

  double* g_sum = a.sums + lane_e;
  const double* bel = a.beliefs + (size_t)lane * 2 * H;
  double* rmean = a.root_mean + (size_t)lane * 2 * H;

  long long* dbg = a.dbg ? a.dbg + (size_t)lane * 16 : nullptr;
  int dbg_k = 0;
#define RBL_STAMP()                                                    \
  do {                                                                 \
    if (dbg && threadIdx.x == 0) dbg[dbg_k] = (long long)clock64();    \
    ++dbg_k;                                                           \
  } while (0)
  RBL_STAMP();  // 0: start
  double bel_t = 0.0, rmean_t = 0.0;  // threads < H: the traverser's root belief and running root value mean of hand tid
  const bool snap_now = a.lane_act_iter && ((cint_p)a.lane_act_iter)[lane] == a.steps_after;
  // ---------------------------------------------------------------- stage (flat, coalesced)
  {
    const int* gp = a.parent + node_off;
    const int* ga = a.act + node_off;
    const int* gb = a.cb + node_off;
    const int* ge = a.ce + node_off;
    const int* gd = a.depth + node_off;
    const int* gi = a.irank + node_off;
    const int* gl = a.leaf_row + node_off;
    const float* gv = a.values + (size_t)row_off * H;
    const int EH = E * H, LH = L * H;
    // every global load is issued before the first LDS store (a plain "load, store, next i" loop paid one memory round
    // trip per stride: 5 for the root tree at 128 threads, a third of the whole step)
    const int nn = min(tid, N - 1);  // clamped, unpredicated loads; the stores below apply the real bounds
    int tp = gp[nn], ta = ga[nn], tcb = gb[nn], te = ge[nn], td = gd[nn], ti = gi[nn], tl = gl[nn];
    const int8_t tm = a.matches[min(tid, FACES * H - 1)];
    constexpr int kV = H <= 6 ? 4 : 5;  // strides that cover the root's L * H leaf values (396 / 594)
    float v_[kV];
    if (!GS && LH > 0) {  // uniform
#pragma unroll
      for (int u = 0; u < kV; ++u) v_[u] = gv[min(tid + u * nthr, LH - 1)];
    }
    if constexpr (!GS) {
      dw* lsig = reinterpret_cast<dw*>(sig);
      dw* lreg = reinterpret_cast<dw*>(reg);
      const dw* gs = reinterpret_cast<const dw*>(g_sig);
      const dw* gr = reinterpret_cast<const dw*>(g_reg);
      const int EW = EH / kW;  // E * H is even whenever kW == 2
#pragma unroll
      for (int u = 0; u < kB; ++u) {
        const int i = tid + u * nthr;
        if (i < EW) {
          lsig[i] = s_[u];
          lreg[i] = r_[u];
        }
      }
      for (int i = tid + kB * nthr; i < EW; i += nthr) {  // what the first kB strides did not cover
        lsig[i] = gs[i];
        lreg[i] = gr[i];
      }
    }
    if constexpr (!GS) {
      float* lv = const_cast<float*>(lvals);
#pragma unroll
      for (int u = 0; u < kV; ++u) {
        const int i = tid + u * nthr;
        if (i < LH) lv[i] = v_[u];
      }
      for (int i = tid + kV * nthr; i < LH; i += nthr) lv[i] = gv[i];
    }
    if (tid < N) {
      t_parent[tid] = tp;
      t_act[tid] = ta;
      t_cb[tid] = tcb;
      t_ce[tid] = te;
      t_depth[tid] = td;
      t_irank[tid] = ti;
      t_lrow[tid] = tl;
    }
    for (int i = tid + nthr; i < N; i += nthr) {  // trees wider than the block (64-thread launches, deeper subgames)
      t_parent[i] = gp[i];
      t_act[i] = ga[i];
      t_cb[i] = gb[i];
      t_ce[i] = ge[i];
      t_depth[i] = gd[i];
      t_irank[i] = gi[i];
      t_lrow[i] = gl[i];
    }
    if (tid < FACES * H) t_match[tid] = tm;
    for (int i = tid + nthr; i < FACES * H; i += nthr) t_match[i] = a.matches[i];
    if (tid < H) {
      bel_t = bel[t * H + tid];
      rho0[tid] = t == 0 ? bel_t : bel[tid];
      rho1[tid] = t == 1 ? bel_t : bel[H + tid];
      rmean_t = rmean[t * H + tid];  // used after the bottom-up sweep: requested now, not in front of a barrier there
    }
    __syncthreads();
  }

  RBL_STAMP();  // 1: staged
  // value of a node without children, from its opponent-reach row (query_value_net :257-268 / terminal payoffs :80-98)
  auto leaf_value = [&](int n, const Row<H>& ro) {
    Row<H> out;
    if (t_act[n] == A - 1) {  // terminal: the bid that was called is the parent's last bid
      const int bid = t_act[t_parent[n]];
      const int qty = 1 + bid / FACES, face = bid % FACES;
      const int8_t* m = t_match + face * H;
      double b[NB];
#pragma unroll
      for (int k = 0; k < NB; ++k) b[k] = 0.0;
      double s = 0.0;
#pragma unroll
      for (int h = 0; h < H; ++h) {  // b[m[h]] += r, without dynamic register indexing (x + 0.0 == x for x >= +0)
        const int mh = m[h];
#pragma unroll
        for (int k = 0; k <= DICE; ++k) b[k] += (mh == k) ? ro.v[h] : 0.0;  // one hand shows at most DICE matches
        s += ro.v[h];
      }
#pragma unroll
      for (int k = DICE - 1; k >= 0; --k) b[k] += b[k + 1];  // bins above DICE are +0.0: adding them changes no bit
      const bool inverse = (root_player ^ (t_depth[n] & 1)) != t;
      // hands with the same number of matches need the same entry b[max(0, qty - matches)]: DICE + 1 candidates are picked
      // once per node, then each hand selects among those (instead of a 2 * DICE deep select chain per hand)
      double cand[DICE + 1];
#pragma unroll
      for (int mm = 0; mm <= DICE; ++mm) {
        const int left = max(0, qty - mm);
        double bl = b[0];
#pragma unroll
        for (int k = 1; k < NB; ++k) bl = (left == k) ? b[k] : bl;
        cand[mm] = (double)(float)bl * 2 - s;  // fp32 truncation (:785)
        if (inverse) cand[mm] *= -1.0;
      }
#pragma unroll
      for (int h = 0; h < H; ++h) {
        const int mh = m[h];
        double x = cand[0];
#pragma unroll
        for (int mm = 1; mm <= DICE; ++mm) x = (mh == mm) ? cand[mm] : x;
        out.v[h] = x;
      }
    } else {
      double s = 0.0;
#pragma unroll
      for (int h = 0; h < H; ++h) s += ro.v[h];
      const float* lv = lvals + t_lrow[n] * H;
#pragma unroll
      for (int h = 0; h < H; ++h) out.v[h] = (double)(float)((double)lv[h] * s);
    }
    store_row<H>(val + n * H, out);
  };

  // ---------------------------------------------------------------- reach of both players under sigma, level by level;
  // rows are stored for nodes with children, leaves turn theirs into a value right away
  if (tid == 0 && t_cb[0] == t_ce[0]) leaf_value(0, load_row<H>(opp == 0 ? rho0 : rho1));
  for (int lev = 1; lev < nlev; ++lev) {
    const int n0 = shc->lev_off[lev], n1 = shc->lev_off[lev + 1];
    const int mover = root_player ^ ((lev - 1) & 1);
    // the mover of the parents is uniform per level: it picks POINTERS (whose row is multiplied by sigma, whose row a leaf
    // needs), not registers -- with `if (mover == 0) r0 *= s else r1 *= s` and `opp == 0 ? r0 : r1` on register rows hipcc
    // kept both variants alive and selected element by element (12 v_cndmask + 6 copies per node)
    double* rho_m = mover == 0 ? rho0 : rho1;  // reach of the player who acted
    double* rho_n = mover == 0 ? rho1 : rho0;  // reach of the other one: copied
    for (int n = n0 + tid; n < n1; n += nthr) {
      const int pr = t_irank[t_parent[n]];
      const int ir = t_irank[n];
      if (ir >= 0) {
        Row<H> rm = load_row<H>(rho_m + pr * H);
        const Row<H> rn = load_row<H>(rho_n + pr * H), s = load_row<H>(sig + (n - 1) * H);
#pragma unroll
        for (int h = 0; h < H; ++h) rm.v[h] = rm.v[h] * s.v[h];
        store_row<H>(rho_m + ir * H, rm);
        store_row<H>(rho_n + ir * H, rn);
      } else if (mover == opp) {  // a leaf only needs the opponent's reach
        Row<H> ro = load_row<H>(rho_m + pr * H);
        const Row<H> s = load_row<H>(sig + (n - 1) * H);
#pragma unroll
        for (int h = 0; h < H; ++h) ro.v[h] = ro.v[h] * s.v[h];
        leaf_value(n, ro);
      } else {
        leaf_value(n, load_row<H>(rho_n + pr * H));
      }
    }
    __syncthreads();
  }

  RBL_STAMP();  // 2: reach + leaf values
  RBL_STAMP();  // 3
  RBL_STAMP();  // 4
  // ---------------------------------------------------------------- bottom-up (update_regrets :542-574) fused with regret
  // matching (:619-634) and the regret discount (:639-650)
  double* rho_t = t == 0 ? rho0 : rho1;
  for (int lev = nlev - 2; lev >= 0; --lev) {
    const int n0 = shc->lev_off[lev], n1 = shc->lev_off[lev + 1];
    const int c_lo = shc->lev_off[lev + 1], c_hi = shc->lev_off[lev + 2];
    const bool mine = (root_player ^ (lev & 1)) == t;
    for (int n = n0 + tid; n < n1; n += nthr) {  // node value: sequential over the actions
      const int c0 = t_cb[n], c1 = t_ce[n];
      if (c0 == c1) continue;
      Row<H> x;
#pragma unroll
      for (int h = 0; h < H; ++h) x.v[h] = 0.0;
      // ascending order over the actions; `mine` is uniform, so it selects the loop, not a branch inside it (with the test
      // in the body hipcc kept two copies of the accumulator row and moved one into the other every iteration)
      if (mine) {
        for (int c = c0; c < c1; ++c) {
          const Row<H> vc = load_row<H>(val + c * H), sc = load_row<H>(sig + (c - 1) * H);
#pragma unroll
          for (int h = 0; h < H; ++h) x.v[h] += vc.v[h] * sc.v[h];
        }
      } else {
        for (int c = c0; c < c1; ++c) {
          const Row<H> vc = load_row<H>(val + c * H);
#pragma unroll
          for (int h = 0; h < H; ++h) x.v[h] += vc.v[h];
        }
      }
      store_row<H>(val + n * H, x);
    }
    __syncthreads();
    if (!mine) continue;
    for (int c = c_lo + tid; c < c_hi; c += nthr) {  // one thread per edge into the level below
      const int p = t_parent[c];
      const Row<H> vc = load_row<H>(val + c * H), xp = load_row<H>(val + p * H);
      Row<H> r = load_row<H>(reg + (c - 1) * H), m;
#pragma unroll
      for (int h = 0; h < H; ++h) {
        double q = r.v[h];
        q += vc.v[h];
        q -= xp.v[h];
        m.v[h] = q > kEps ? q : kEps;
        r.v[h] = q * (q > 0 ? a.pos : a.neg);
      }
      store_row<H>(sig + (c - 1) * H, m);
      store_row<H>(reg + (c - 1) * H, r);
    }
    __syncthreads();
    for (int n = n0 + tid; n < n1; n += nthr) {  // row sums, sequential over the actions; parked in the (dead) rho_t row
      const int c0 = t_cb[n], c1 = t_ce[n];
      if (c0 == c1) continue;
      Row<H> s;
#pragma unroll
      for (int h = 0; h < H; ++h) s.v[h] = 0.0;
      for (int c = c0; c < c1; ++c) {
        const Row<H> mc = load_row<H>(sig + (c - 1) * H);
#pragma unroll
        for (int h = 0; h < H; ++h) s.v[h] += mc.v[h];
      }
      // m / s below is hipcc's f64 division sequence with its denominator-only part (v_rcp_f64 + two Newton steps) done here,
      // once per node instead of once per edge, and without v_div_scale / v_div_fixup, which are the identity for these
      // operands (1e-80 <= m <= s, exponents < 768 apart; scripts/micro/div_shared_rcp.hip checks 3e9 cases against `/`)
      Row<H> y;
#pragma unroll
      for (int h = 0; h < H; ++h) {
        double yy = __builtin_amdgcn_rcp(s.v[h]);
        double e = __builtin_fma(-s.v[h], yy, 1.0);
        yy = __builtin_fma(yy, e, yy);
        e = __builtin_fma(-s.v[h], yy, 1.0);
        y.v[h] = __builtin_fma(yy, e, yy);
      }
      store_row<H>(rho_t + t_irank[n] * H, s);
      store_row<H>(yrow + t_irank[n] * H, y);
    }
    __syncthreads();
    for (int c = c_lo + tid; c < c_hi; c += nthr) {
      const int pr = t_irank[t_parent[c]];
      const Row<H> s = load_row<H>(rho_t + pr * H), y = load_row<H>(yrow + pr * H);
      Row<H> m = load_row<H>(sig + (c - 1) * H);
#pragma unroll
      for (int h = 0; h < H; ++h) {
        const double q0 = m.v[h] * y.v[h];
        const double rem = __builtin_fma(-s.v[h], q0, m.v[h]);
        m.v[h] = __builtin_fma(rem, y.v[h], q0);
      }
      store_row<H>(sig + (c - 1) * H, m);
    }
    __syncthreads();
  }

  RBL_STAMP();  // 5: bottom-up
  // ---------------------------------------------------------------- running mean of the root values (:579-590)
  if (tid < H) {
    double m = rmean_t;
    m += (val[tid] - m) * a.alpha;
    rmean[t * H + tid] = m;
    rho_t[tid] = bel_t;  // root row of the traverser (it served as scratch above)
  }
  __syncthreads();

  // ---------------------------------------------------------------- traverser's reach under the NEW sigma (:636-638), rows
  // of nodes with children only
  for (int lev = 1; lev < nlev - 1; ++lev) {
    const int n0 = shc->lev_off[lev], n1 = shc->lev_off[lev + 1];
    const bool own = (root_player ^ ((lev - 1) & 1)) == t;
    for (int n = n0 + tid; n < n1; n += nthr) {
      const int ir = t_irank[n];
      if (ir < 0) continue;
      Row<H> r = load_row<H>(rho_t + t_irank[t_parent[n]] * H);
      if (own) {
        const Row<H> s = load_row<H>(sig + (n - 1) * H);
#pragma unroll
        for (int h = 0; h < H; ++h) r.v[h] = r.v[h] * s.v[h];
      }
      store_row<H>(rho_t + ir * H, r);
    }
    __syncthreads();
  }

  RBL_STAMP();  // 6: new reach
  // ---------------------------------------------------------------- sum_strategies (:651-657) + write back what changed
  {
    double* snap = a.snapshot + lane_e;
    for (int c = 1 + tid; c < N; c += nthr) {
      const int p = t_parent[c], e = (c - 1) * H;
      const Row<H> s = load_row<H>(sig + e);
      if ((root_player ^ (t_depth[p] & 1)) == t) {
        const Row<H> rp = load_row<H>(rho_t + t_irank[p] * H);
        const Row<H> rg = load_row<H>(reg + e);
        if constexpr (H % 2 == 0) {  // rows are 16-byte aligned: half as many global instructions
          typedef double d2 __attribute__((ext_vector_type(2)));
          d2* gs = reinterpret_cast<d2*>(g_sum + e);
          d2* gg = reinterpret_cast<d2*>(g_sig + e);
          d2* gr = reinterpret_cast<d2*>(g_reg + e);
          d2 acc2[H / 2];
#pragma unroll
          for (int h = 0; h < H / 2; ++h) acc2[h] = gs[h];
#pragma unroll
          for (int h = 0; h < H / 2; ++h) {
            double x0 = acc2[h][0], x1 = acc2[h][1];
            x0 *= a.strat;
            x1 *= a.strat;
            x0 += rp.v[2 * h] * s.v[2 * h];
            x1 += rp.v[2 * h + 1] * s.v[2 * h + 1];
            gs[h] = d2{x0, x1};
            if constexpr (!GS) {  // GS: sigma and regrets were updated in place
              gg[h] = d2{s.v[2 * h], s.v[2 * h + 1]};
              gr[h] = d2{rg.v[2 * h], rg.v[2 * h + 1]};
            }
          }
        } else {
#pragma unroll
          for (int h = 0; h < H; ++h) {
            double x = g_sum[e + h];
            x *= a.strat;
            x += rp.v[h] * s.v[h];
            g_sum[e + h] = x;
            if constexpr (!GS) {
              g_sig[e + h] = s.v[h];
              g_reg[e + h] = rg.v[h];
            }
          }
        }
      }
      if (snap_now) {
        if constexpr (H % 2 == 0) {
          typedef double d2 __attribute__((ext_vector_type(2)));
          d2* sn = reinterpret_cast<d2*>(snap + e);
#pragma unroll
          for (int h = 0; h < H / 2; ++h) sn[h] = d2{s.v[2 * h], s.v[2 * h + 1]};
        } else {
#pragma unroll
          for (int h = 0; h < H; ++h) snap[e + h] = s.v[h];
        }
      }
    }
  }

  RBL_STAMP();  // 7: write-back
  // (x + eps) / s for the H elements of a row, bit-identical to H IEEE divisions: this IS hipcc's f64 division sequence
  // (v_rcp_f64, two Newton steps on the reciprocal, q0 = a y, r = a - s q0, q = q0 + r y) with the part that only depends
  // on the denominator done once per row.  What is left out -- v_div_scale and v_div_fixup -- is the identity here: both
  // only act when the operands' exponents are >= 768 apart, denormal, zero, infinite or NaN, and 1e-80 <= a <= s <= H + 1.
  auto norm_row = [&](const Row<H>& r, double ssum, float* dst) {
    double y = __builtin_amdgcn_rcp(ssum);
    double e = __builtin_fma(-ssum, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-ssum, y, 1.0);
    y = __builtin_fma(y, e, y);
#pragma unroll
    for (int h = 0; h < H; ++h) {
      const double num = r.v[h] + kEps;
      const double q0 = num * y;
      const double rem = __builtin_fma(-ssum, q0, num);
      dst[h] = (float)__builtin_fma(rem, y, q0);
    }
  };
  // ---------------------------------------------------------------- queries for the next step (:253-269, :104-123)
  if (a.next_trav >= 0 && L > 0) {
    __syncthreads();  // val / reg are dead from here on: their bytes stage the query rows
    float* gq = a.queries + (size_t)row_off * Q;
    for (int n = tid; n < N; n += nthr) {  // a pseudo-leaf's thread writes its row (no look-up of the global leaf list)
      const int k = t_lrow[n];
      if (k < 0) continue;
      float* q = (GS ? gq : qstage) + k * Q;
      // rm: reach of the player who acted at the parent (times sigma), rn: the other player's (copied); which of the two is
      // player 0 only decides WHERE in the row they are written (a per-thread offset, not a per-element register select)
      Row<H> rm, rn;
      int pm = 0;  // player whose reach is rm
      const int ir = t_irank[n];
      if (ir >= 0) {  // only the root can be a pseudo-leaf with a stored row (max_depth = 0)
        rm = load_row<H>(rho0 + ir * H);
        rn = load_row<H>(rho1 + ir * H);
      } else {
        const int p = t_parent[n], pr = t_irank[p];
        pm = root_player ^ (t_depth[p] & 1);
        rm = load_row<H>((pm == 0 ? rho0 : rho1) + pr * H);
        rn = load_row<H>((pm == 0 ? rho1 : rho0) + pr * H);
        const Row<H> s = load_row<H>(sig + (n - 1) * H);
#pragma unroll
        for (int h = 0; h < H; ++h) rm.v[h] = rm.v[h] * s.v[h];
      }
      double sm = 0, sn = 0;
#pragma unroll
      for (int h = 0; h < H; ++h) sm += rm.v[h] + kEps;  // normalize_probabilities_safe (util.h:68-78)
#pragma unroll
      for (int h = 0; h < H; ++h) sn += rn.v[h] + kEps;
      q[0] = (float)(root_player ^ (t_depth[n] & 1));
      q[1] = (float)a.next_trav;
      const int lb = t_act[n];
#pragma unroll
      for (int j = 0; j < A; ++j) q[2 + j] = (j == lb) ? 1.0f : 0.0f;
      float* qm = q + 2 + A + pm * H;
      float* qn = q + 2 + A + (1 - pm) * H;
      norm_row(rm, sm, qm);
      norm_row(rn, sn, qn);
    }
    if constexpr (!GS) {
      __syncthreads();
      for (int i = tid; i < L * Q; i += nthr) gq[i] = qstage[i];
    }
  }
  RBL_STAMP();  // 8: queries
#undef RBL_STAMP
}

}  // namespace

size_t cfr_rows_lds_bytes(int N, int NI, int H, int L, int faces) {
  size_t d = (size_t)3 * NI * H + (size_t)(N - 1) * H + (size_t)N * H + (size_t)(N - 1) * H;  // doubles
  size_t b = d * 8 + (size_t)((L * H + 3) & ~3) * 4 + (size_t)7 * N * 4 + (size_t)faces * H;
  return (b + 15) & ~(size_t)15;
}

// (L is part of the signature the engine sizes both 2 dice x 6 faces kernels with; this one keeps a value row per node)
size_t cfr_rows_global_lds_bytes(int N, int NI, int H, int /*L*/, int faces) {
  const size_t b = ((size_t)3 * NI * H + (size_t)N * H) * 8 + (size_t)7 * N * 4 + (size_t)faces * H;
  return (b + 15) & ~(size_t)15;
}

bool cfr_rows_global_supported(int H, int A, int dice, int faces) { return H == 36 && A == 25 && dice == 2 && faces == 6; }

bool launch_cfr_rows_global(const CfrArgs& a, int B, size_t lds_bytes, hipStream_t stream) {
  if (!(a.H == 36 && a.A == 25 && a.dice == 2)) return false;
  auto kern = cfr_rows_kernel<36, 25, 2, 6, true>;
  // more than 64 KB of dynamic LDS has to be requested explicitly -- per DEVICE (one engine and driver thread per GPU in
  // the trainer's in-process topology), once, and a refusal is an error, not something to swallow
  static std::once_flag attr_once[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) throw std::runtime_error("cfr_rows_global: no current device");
  hipError_t attr_err = hipSuccess;
  std::call_once(attr_once[dev], [&] {
    attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  if (attr_err != hipSuccess)
    throw std::runtime_error(std::string("cfr_rows_global: cannot request 160 KB of LDS: ") + hipGetErrorString(attr_err));
  hipLaunchKernelGGL(kern, dim3(B), dim3(256), lds_bytes, stream, a);
  return true;
}

bool cfr_rows_supported(int H, int A, int dice, int faces) {
  return (H == 6 && A == 13 && dice == 1 && faces == 6) || (H == 4 && A == 9 && dice == 1 && faces == 4) ||
         (H == 5 && A == 11 && dice == 1 && faces == 5) || (H == 9 && A == 13 && dice == 2 && faces == 3);
}

bool launch_cfr_rows(const CfrArgs& a, int B, int block, size_t lds_bytes, hipStream_t stream) {
#define RBL_ROWS(H_, A_, D_, F_)                                                                              \
  do {                                                                                                        \
    hipLaunchKernelGGL((cfr_rows_kernel<H_, A_, D_, F_, false>), dim3(B), dim3(block), lds_bytes, stream, a); \
    return true;                                                                                              \
  } while (0)
  if (a.H == 6 && a.A == 13 && a.dice == 1) RBL_ROWS(6, 13, 1, 6);
  if (a.H == 4 && a.A == 9 && a.dice == 1) RBL_ROWS(4, 9, 1, 4);
  if (a.H == 5 && a.A == 11 && a.dice == 1) RBL_ROWS(5, 11, 1, 5);
  if (a.H == 9 && a.A == 13 && a.dice == 2) RBL_ROWS(9, 13, 2, 3);
#undef RBL_ROWS
  return false;
}

}  // namespace rbl
