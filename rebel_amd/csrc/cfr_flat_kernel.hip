// rebel_amd/csrc/cfr_flat_kernel.hip -- CFR::step for 2 dice x 6 faces: element-parallel passes, sigma resident in LDS.
//
// Why another kernel.  cfr_rows_kernel<GS> gives every tree ROW (36 hands, 288 bytes) to one thread and keeps sigma and the
// regrets in the lane's global slab.  Measured on MI355X (scripts/probe_cfr_phases_2d6f.py): a root-sized lane (325 nodes)
// takes 115 k cycles per step, and it is neither HBM nor arithmetic that it waits for:
//   * a thread reading ITS row in 16-byte pieces makes every wave-level load touch 64 different cache lines (the rows are 288
//     bytes apart): the texture addresser spends ~64 cycles per instruction instead of 16, and sigma is read six times a step;
//   * at most 325 rows = 5 waves per lane, one lane per CU (a value row per node: 124 KB of LDS): one wave per SIMD, every
//     LDS / L2 / HBM round trip fully exposed; the node-value pass has 24 busy threads walking 24 children each.
// (Halving the LDS footprint to get two lanes per CU made it slower: 153 k cycles -- the extra lanes only queue at the
// addresser.)  Here instead:
//   * every pass that is elementwise in the hand index -- reach products, node values (sequential over the children),
//     regret update, regret matching, normalisation, sum_strategies, the query rows -- runs over (row, hand) ITEMS spread
//     over 512 threads: consecutive threads touch consecutive addresses, global traffic is coalesced;
//   * only what is sequential over the HANDS stays one thread per row, on LDS rows: a leaf's reach sum, a terminal's match
//     histogram, the two normalisation sums of a query row;
//   * sigma lives in LDS for the whole step (93 KB at the root): read from global once, the traverser's rows written back
//     once; regrets and sum_strategies are touched once (read + write) by the pass that needs them;
//   * a pseudo-leaf keeps ONE double (the reach sum that scales the net's row); its value row float(net row x sum) is
//     recomputed where it is read -- the same expression, bit for bit.  Value rows exist for nodes with children and
//     terminals only (49 of 325 at the root).
// LDS at the root: 143 KB (one lane per CU, 8 waves); a 160-node tree needs 70 KB (two lanes).  Arithmetic is
// operation-for-operation what cfr_rows_kernel / cfr_kernels.hip do (same operands, same order, -ffp-contract=off):
// tests/test_cfr_parity.py and tests/test_selfplay_parity.py run the 2 dice x 6 faces cases against the oracle through it.
//
// Reference: CFR::step and its helpers, /root/reference/csrc/liars_dice/subgame_solving.cc:538-664; leaf queries :253-269,
// terminal payoffs :80-98, :765-789.
#include <mutex>
#include <stdexcept>
#include <string>

#include "cfr_kernels.h"

namespace rbl {

namespace {

constexpr double kEps = 1e-80;
constexpr int kFlatMaxThreads = 1024;
constexpr int kG = 12;  // items per thread whose GLOBAL operands are requested together (12 x 1024 threads cover a root tree)
constexpr int kU = 4;  // items whose loads are issued together: a pass is a chain of dependent LDS / memory round trips per
                       // item (table word -> operands -> store), and with 16 waves per CU only independent work inside a
                       // thread hides them (one item at a time: ~700 cycles per item, 137 k cycles per root lane)

// per-node word built at staging: what a pass needs to know about a node's PARENT without a second table round trip
__device__ __forceinline__ int pk_pr(int w) { return w & 255; }          // reach row of the parent
__device__ __forceinline__ int pk_pv(int w) { return (w >> 8) & 255; }   // value row of the parent
__device__ __forceinline__ int pk_pdp(int w) { return (w >> 16) & 1; }   // depth parity of the parent
__device__ __forceinline__ int pk_ir(int w) { return ((w >> 17) & 63) - 1; }  // reach row of the node itself, -1: none

template <int H, int A, int DICE, int FACES>
__global__ void __launch_bounds__(kFlatMaxThreads) cfr_flat_kernel(const CfrArgs a) {
  extern __shared__ __align__(16) double lds[];
  constexpr int Q = 2 + A + 2 * H, NB = 2 * DICE + 1;
  typedef const int __attribute__((address_space(4)))* cint_p;
  typedef const ShapeDev __attribute__((address_space(4)))* cshape_p;
  const int lane = a.lane_order ? ((cint_p)a.lane_order)[a.lane0 + blockIdx.x] : a.lane0 + (int)blockIdx.x;
  const int tid = threadIdx.x, NT = blockDim.x;
  const cshape_p shc = (cshape_p)a.shapes + ((cint_p)a.lane_shape)[lane];
  const int N = shc->N, E = N - 1, L = shc->L, NI = shc->NI, nlev = shc->nlev, node_off = shc->node_off;
  const int NV = N - L, T = shc->T;
  const int root_player = ((cint_p)a.lane_root_player)[lane], row_off = ((cint_p)a.lane_row_off)[lane];
  const int t = a.trav, opp = 1 - t;

  // ---- LDS layout (doubles): rho0, rho1, yrow [NI][H] | val [NV][H] | lsum [L] | qs [4][L] | sig [E][H] | tables
  double* rho0 = lds;
  double* rho1 = rho0 + NI * H;
  double* yrow = rho1 + NI * H;  // refined reciprocals of the regret-matching row sums
  double* val = yrow + NI * H;   // values of the nodes that are not pseudo-leaves (row = -1 - t_lrow[n])
  double* lsum = val + NV * H;   // pseudo-leaf: sum of the opponent's reach
  double* qs = lsum + L;         // query rows: [sum, reciprocal] of the acting player's reach, then of the other's
  double* sig = qs + 4 * L;
  int* tb = reinterpret_cast<int*>(sig + E * H);
  int* t_parent = tb, *t_act = tb + N, *t_cb = tb + 2 * N, *t_ce = tb + 3 * N, *t_depth = tb + 4 * N;
  int* t_pack = tb + 5 * N, *t_lrow = tb + 6 * N, *t_leaf = tb + 7 * N;  // t_leaf [L]: node of net row k
  int* t_term = t_leaf + L;                                              // t_term [T]: the terminals, ascending
  int* t_qrec = t_term + T;  // [L]: node | parent's reach row << 9 | who acted there << 17 | player to move << 18 | last bid << 19
  // t_mask [FACES][2]: bit h set when hand h shows exactly 1 / exactly 2 of the face (8-byte aligned slot after the ints)
  unsigned long long* t_mask = reinterpret_cast<unsigned long long*>(tb + ((7 * N + 2 * L + T + 1) & ~1));

  const size_t lane_e = (size_t)lane * a.Emax * H;
  double* g_sig = a.sigma + lane_e;
  double* g_reg = a.regrets + lane_e;
  double* g_sum = a.sums + lane_e;
  const float* lvals = a.values + (size_t)row_off * H;
  const double* bel = a.beliefs + (size_t)lane * 2 * H;
  double* rmean = a.root_mean + (size_t)lane * 2 * H;
  const bool snap_now = a.lane_act_iter && ((cint_p)a.lane_act_iter)[lane] == a.steps_after;

  long long* dbg = a.dbg ? a.dbg + (size_t)lane * 16 : nullptr;
  int dbg_k = 0;
#define RBL_STAMP()                                                    \
  do {                                                                 \
    if (dbg && threadIdx.x == 0) dbg[dbg_k] = (long long)clock64();    \
    ++dbg_k;                                                           \
  } while (0)
  RBL_STAMP();  // 0: start
#if defined(RBL_FINE) && RBL_FINE == 1
#define RBL_F1() RBL_STAMP()
#else
#define RBL_F1() do {} while (0)
#endif
#if defined(RBL_FINE) && RBL_FINE == 2
#define RBL_F2() RBL_STAMP()
#else
#define RBL_F2() do {} while (0)
#endif

  // ---------------------------------------------------------------- stage: sigma (coalesced 16-byte pieces), tables
  double bel_t = 0.0, rmean_t = 0.0;
  {
    typedef double d2 __attribute__((ext_vector_type(2)));
    const d2* gs = reinterpret_cast<const d2*>(g_sig);
    d2* ls = reinterpret_cast<d2*>(sig);
    const int EW = E * H / 2;  // H is even
    // batches of independent loads before their stores (one load -> store per iteration paid a memory round trip each)
    for (int i0 = tid; i0 < EW; i0 += 12 * NT) {
      d2 v[12];
#pragma unroll
      for (int u = 0; u < 12; ++u) v[u] = gs[min(i0 + u * NT, EW - 1)];
#pragma unroll
      for (int u = 0; u < 12; ++u)
        if (i0 + u * NT < EW) ls[i0 + u * NT] = v[u];
    }
    const int* gp = a.parent + node_off;
    const int* ga = a.act + node_off;
    const int* gb = a.cb + node_off;
    const int* ge = a.ce + node_off;
    const int* gd = a.depth + node_off;
    const int* gi = a.irank + node_off;
    const int* gl = a.leaf_row + node_off;
    const int* gv = a.vrow + node_off;
    for (int i = tid; i < N; i += NT) {
      const int p = gp[i], lr = gl[i], vr = gv[i], ir = gi[i];
      const int pp = max(p, 0);
      // second round trip: the parent's rows (once per step; the passes below then need one table word per item)
      const int pir = gi[pp], pvr = gv[pp], pd = gd[pp];
      t_parent[i] = p;
      t_act[i] = ga[i];
      t_cb[i] = gb[i];
      t_ce[i] = ge[i];
      t_depth[i] = gd[i];
      t_pack[i] = (pir & 255) | ((pvr & 255) << 8) | ((pd & 1) << 16) | ((ir + 1) << 17);
      t_lrow[i] = lr >= 0 ? lr : -1 - vr;
      if (lr >= 0) t_leaf[lr] = i;
    }
    if (tid < 2 * FACES) {
      const int8_t* mrow = a.matches + (tid >> 1) * H;
      unsigned long long bits = 0;
      for (int h = 0; h < H; ++h) bits |= (unsigned long long)(mrow[h] == (tid & 1) + 1) << h;
      t_mask[tid] = bits;
    }
    for (int i = tid; i < T; i += NT) t_term[i] = (a.terms + shc->term_off)[i];
    if (tid < H) {
      bel_t = bel[t * H + tid];
      rho0[tid] = t == 0 ? bel_t : bel[tid];
      rho1[tid] = t == 1 ? bel_t : bel[H + tid];
      rmean_t = rmean[t * H + tid];
    }
    __syncthreads();
  }
  RBL_STAMP();  // 1: staged

  // value of node c for hand h, as its parent reads it (lr = t_lrow[c])
  auto child_val = [&](int lr, int h) {
    if (lr >= 0) return (double)(float)((double)lvals[lr * H + h] * lsum[lr]);  // query_value_net (:257-268)
    return val[(-1 - lr) * H + h];
  };

  // ---------------------------------------------------------------- reach of both players under sigma, level by level
  for (int lev = 1; lev < nlev; ++lev) {
    const int n0 = shc->lev_off[lev], n1 = shc->lev_off[lev + 1];
    const int mover = root_player ^ ((lev - 1) & 1);
    double* rho_m = mover == 0 ? rho0 : rho1;  // reach of the player who acted: times sigma
    double* rho_n = mover == 0 ? rho1 : rho0;  // the other one's: copied
    if (lev < nlev - 1) {  // rows are kept for nodes with children (the last level has none)
      const int cnt = (n1 - n0) * H;
      for (int i0 = tid; i0 < cnt; i0 += kU * NT) {
        double rm[kU], rn[kU], sg[kU];
        int dst[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
          const int i = min(i0 + u * NT, cnt - 1);
          const int n = n0 + i / H, h = i % H;
          const int w = t_pack[n];
          dst[u] = (i0 + u * NT < cnt && pk_ir(w) >= 0) ? pk_ir(w) * H + h : -1;
          rm[u] = rho_m[pk_pr(w) * H + h];
          rn[u] = rho_n[pk_pr(w) * H + h];
          sg[u] = sig[(n - 1) * H + h];
        }
#pragma unroll
        for (int u = 0; u < kU; ++u)
          if (dst[u] >= 0) {
            rho_m[dst[u]] = rm[u] * sg[u];
            rho_n[dst[u]] = rn[u];
          }
      }
    }
    RBL_F1();  // rows
    // leaves: one thread per node, sequential over the hands (the sums below are order-sensitive).  Pseudo-leaves are taken
    // from the front of the block, terminals from its back: different waves, so the two paths run side by side instead of
    // one after the other in every wave that holds both kinds
    const bool times_sigma = mover == opp;  // the opponent acted: its reach at the leaf is the parent's times sigma
    for (int n = n0 + tid; n < n1; n += NT) {
      const int lr = t_lrow[n];
      if (lr < 0) continue;
      const int pr = pk_pr(t_pack[n]);
      const double* ro = (times_sigma ? rho_m : rho_n) + pr * H;
      const double* sg = sig + (n - 1) * H;
      double s = 0.0;
      if (times_sigma) {
#pragma unroll 6
        for (int h = 0; h < H; ++h) s += ro[h] * sg[h];
      } else {
#pragma unroll 6
        for (int h = 0; h < H; ++h) s += ro[h];
      }
      lsum[lr] = s;
    }
    RBL_F1();  // pseudo-leaves
    // terminals (:80-98, :765-789): FOUR threads per node -- one per match bin (a hand shows 0..DICE matches) and one for the
    // total, each a sequential sum over the hands (a single thread doing all four: ~600 dependent fp64 instructions, 8 k
    // cycles); then each writes a quarter of the value row.  x + 0.0 == x for x >= +0, so a bin's masked sum is the sum of
    // its members in ascending hand order.
    {
      static_assert(DICE == 2 && H % 4 == 0, "three match bins + the total = four threads per terminal");
      const int rt = NT - 1 - tid, g = rt >> 2, role = rt & 3;
      const int ln = tid & 63;
      for (int g0 = g; g0 < T; g0 += NT / 4) {
        const int n = t_term[g0];
        if (n < n0 || n >= n1) continue;  // uniform over the four threads of a group
#ifdef RBL_SKIP_TERM
        if (a.dbg) continue;
#endif
        const int pr = pk_pr(t_pack[n]);
        const double* ro = (times_sigma ? rho_m : rho_n) + pr * H;
        const double* sg = sig + (n - 1) * H;
        const int bid = t_act[t_parent[n]];  // the bid that was called is the parent's last bid
        const int qty = 1 + bid / FACES, face = bid % FACES;
        // which hands show exactly 1 / exactly 2 matches of `face`: two 36-bit masks in registers (reading the byte table per
        // hand between the stores below made every iteration an LDS round trip: hipcc keeps LDS loads behind earlier stores)
        const unsigned long long m1 = t_mask[2 * face], m2 = t_mask[2 * face + 1];
        const unsigned long long mine_mask = role == 3 ? ~0ull : (role == 0 ? ~(m1 | m2) : (role == 1 ? m1 : m2));
        double acc = 0.0;
#pragma unroll 6
        for (int h = 0; h < H; ++h) {
          double r = ro[h];
          if (times_sigma) r = r * sg[h];
          acc += ((mine_mask >> h) & 1) ? r : 0.0;
        }
        // the lane that holds role j of this group: reversed thread order, so role j sits at (ln | 3) - j
        double b[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) b[k] = 0.0;
#pragma unroll
        for (int k = 0; k <= DICE; ++k) b[k] = __shfl(acc, (ln | 3) - k);
        const double s = __shfl(acc, (ln | 3) - 3);
#pragma unroll
        for (int k = DICE - 1; k >= 0; --k) b[k] += b[k + 1];  // bins above DICE are +0.0
        const bool inverse = (root_player ^ (t_depth[n] & 1)) != t;
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
        double* out = val + (-1 - t_lrow[n]) * H + role * (H / 4);
        double o[H / 4];
#pragma unroll
        for (int hh = 0; hh < H / 4; ++hh) {
          const int h = role * (H / 4) + hh;
          o[hh] = ((m2 >> h) & 1) ? cand[2] : (((m1 >> h) & 1) ? cand[1] : cand[0]);
        }
#pragma unroll
        for (int hh = 0; hh < H / 4; ++hh) out[hh] = o[hh];
      }
    }
    RBL_F1();  // terminals
    __syncthreads();
    RBL_F1();  // barrier
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
    // node values: one item per (node, hand), sequential over the actions in ascending order; the children's operands are
    // requested eight at a time (a pseudo-leaf's value comes from the net's rows in global memory)
    for (int i = tid; i < (n1 - n0) * H; i += NT) {
      const int n = n0 + i / H, h = i % H;
      const int c0 = t_cb[n], c1 = t_ce[n];
      if (c0 == c1) continue;
      // a pseudo-leaf's value comes from the net's rows in global memory: twelve children's are requested before the first is
      // used (all 24 at once spilled registers to scratch: 24 k cycles for this pass at the root)
      constexpr int kC = 12;
      double x = 0.0;
      for (int cb = c0; cb < c1; cb += kC) {
        int lr[kC];
        float lvf[kC];
#pragma unroll
        for (int u = 0; u < kC; ++u) lr[u] = t_lrow[min(cb + u, c1 - 1)];
        // unconditional loads from a row that exists (row 0 when the child is not a pseudo-leaf): a load under a condition is a
        // branch, and loads in different basic blocks are not requested together
#pragma unroll
        for (int u = 0; u < kC; ++u) lvf[u] = lvals[max(lr[u], 0) * H + h];
#pragma unroll
        for (int u = 0; u < kC; ++u)
          if (cb + u < c1) {
            const double v = lr[u] >= 0 ? (double)(float)((double)lvf[u] * lsum[lr[u]]) : val[(-1 - lr[u]) * H + h];
            if (mine) x += v * sig[(cb + u - 1) * H + h];
            else x += v;
          }
      }
      val[(-1 - t_lrow[n]) * H + h] = x;
    }
    RBL_F2();  // node values
    __syncthreads();
    if (!mine) continue;
    // regrets of the edges into the level below; sigma receives the clamped regrets
    {
      const int cnt = (c_hi - c_lo) * H;
      for (int i0 = tid; i0 < cnt; i0 += kG * NT) {  // every global operand of the thread's items in one round trip
        double q[kG];
        float lvf[kG];
        int lr[kG];
#pragma unroll
        for (int u = 0; u < kG; ++u) {  // clamped, unconditional: straight-line code
          const int i = min(i0 + u * NT, cnt - 1);
          const int c = c_lo + i / H, h = i % H;
          q[u] = g_reg[(c - 1) * H + h];
          lr[u] = t_lrow[c];
          lvf[u] = lvals[max(lr[u], 0) * H + h];
        }
#pragma unroll
        for (int u = 0; u < kG; ++u)
          if (i0 + u * NT < cnt) {
            const int i = i0 + u * NT;
            const int c = c_lo + i / H, h = i % H;
            const int e = (c - 1) * H + h;
            const double cv = lr[u] >= 0 ? (double)(float)((double)lvf[u] * lsum[lr[u]]) : val[(-1 - lr[u]) * H + h];
            double qq = q[u];
            qq += cv;
            qq -= val[pk_pv(t_pack[c]) * H + h];
            sig[e] = qq > kEps ? qq : kEps;
            g_reg[e] = qq * (qq > 0 ? a.pos : a.neg);
          }
      }
    }
    RBL_F2();  // regrets
    __syncthreads();
    // row sums, sequential over the actions; parked in the (dead) rho_t row.  m / s below is hipcc's f64 division sequence
    // with its denominator-only part (v_rcp_f64 + two Newton steps) done here once per (node, hand); v_div_scale / v_div_fixup
    // are the identity for these operands (1e-80 <= m <= s; scripts/micro/div_shared_rcp.hip checks 3e9 cases against `/`)
    for (int i = tid; i < (n1 - n0) * H; i += NT) {
      const int n = n0 + i / H, h = i % H;
      const int c0 = t_cb[n], c1 = t_ce[n];
      if (c0 == c1) continue;
      double s = 0.0;
      for (int cb = c0; cb < c1; cb += 8) {
        double m8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) m8[u] = sig[(min(cb + u, c1 - 1) - 1) * H + h];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (cb + u < c1) s += m8[u];
      }
      double yy = __builtin_amdgcn_rcp(s);
      double er = __builtin_fma(-s, yy, 1.0);
      yy = __builtin_fma(yy, er, yy);
      er = __builtin_fma(-s, yy, 1.0);
      const int ir = pk_ir(t_pack[n]);
      rho_t[ir * H + h] = s;
      yrow[ir * H + h] = __builtin_fma(yy, er, yy);
    }
    RBL_F2();  // row sums
    __syncthreads();
    {
      const int cnt = (c_hi - c_lo) * H;
      for (int i0 = tid; i0 < cnt; i0 += kU * NT) {
        double s[kU], y[kU], m[kU];
        int e[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
          const int i = min(i0 + u * NT, cnt - 1);
          const int c = c_lo + i / H, h = i % H;
          const int pr = pk_pr(t_pack[c]);
          e[u] = i0 + u * NT < cnt ? (c - 1) * H + h : -1;
          s[u] = rho_t[pr * H + h];
          y[u] = yrow[pr * H + h];
          m[u] = sig[(c - 1) * H + h];
        }
#pragma unroll
        for (int u = 0; u < kU; ++u)
          if (e[u] >= 0) {
            const double q0 = m[u] * y[u];
            const double rem = __builtin_fma(-s[u], q0, m[u]);
            sig[e[u]] = __builtin_fma(rem, y[u], q0);
          }
      }
    }
    __syncthreads();
  }
  RBL_STAMP();  // 5: bottom-up

  // ---------------------------------------------------------------- running mean of the root values (:579-590)
  if (tid < H) {
    double m = rmean_t;
    m += (val[tid] - m) * a.alpha;  // the root is row 0 of val (it has children)
    rmean[t * H + tid] = m;
    rho_t[tid] = bel_t;  // root row of the traverser (it served as scratch above)
  }
  __syncthreads();
  // ---------------------------------------------------------------- traverser's reach under the NEW sigma (:636-638)
  for (int lev = 1; lev < nlev - 1; ++lev) {
    const int n0 = shc->lev_off[lev], n1 = shc->lev_off[lev + 1];
    const bool own = (root_player ^ ((lev - 1) & 1)) == t;
    const int cnt = (n1 - n0) * H;
    for (int i0 = tid; i0 < cnt; i0 += kU * NT) {
      double r[kU], sg[kU];
      int dst[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int i = min(i0 + u * NT, cnt - 1);
        const int n = n0 + i / H, h = i % H;
        const int w = t_pack[n];
        dst[u] = (i0 + u * NT < cnt && pk_ir(w) >= 0) ? pk_ir(w) * H + h : -1;
        r[u] = rho_t[pk_pr(w) * H + h];
        sg[u] = sig[(n - 1) * H + h];
      }
#pragma unroll
      for (int u = 0; u < kU; ++u)
        if (dst[u] >= 0) rho_t[dst[u]] = own ? r[u] * sg[u] : r[u];
    }
    __syncthreads();
  }
  RBL_STAMP();  // 6: new reach

  // ---------------------------------------------------------------- sum_strategies (:651-657) + write back what changed
  {
    double* snap = a.snapshot + lane_e;
    const int cnt = E * H;
    for (int i0 = tid; i0 < cnt; i0 += kG * NT) {
      double x[kG];
      bool own[kG];
#pragma unroll
      for (int u = 0; u < kG; ++u) {  // clamped, unconditional: straight-line code
        const int i = min(i0 + u * NT, cnt - 1);
        own[u] = (root_player ^ pk_pdp(t_pack[1 + i / H])) == t;
        x[u] = g_sum[i];
      }
#pragma unroll
      for (int u = 0; u < kG; ++u)
        if (i0 + u * NT < cnt) {
          const int i = i0 + u * NT;
          const int h = i % H;
          const double s = sig[i];
          if (own[u]) {
            double xx = x[u];
            xx *= a.strat;
            xx += rho_t[pk_pr(t_pack[1 + i / H]) * H + h] * s;
            g_sum[i] = xx;
            g_sig[i] = s;
          }
          if (snap_now) snap[i] = s;
        }
    }
  }
  RBL_STAMP();  // 7: write-back

  // ---------------------------------------------------------------- queries for the next step (:253-269, :104-123)
  if (a.next_trav >= 0 && L > 0) {
    // sums of (reach + eps) over the hands (normalize_probabilities_safe, util.h:68-78) and the denominator-only part of the
    // division, one thread per pseudo-leaf; rm: reach of the player who acted at the parent (times sigma), rn: the other's
    for (int k = tid; k < L; k += NT) {
      const int n = t_leaf[k];
      const int w = t_pack[n];
      const int pr = pk_pr(w), pm = root_player ^ pk_pdp(w);
      const double* rm = (pm == 0 ? rho0 : rho1) + pr * H;
      const double* rn = (pm == 0 ? rho1 : rho0) + pr * H;
      const double* sg = sig + (n - 1) * H;
      double sm = 0, sn = 0;
#pragma unroll 6
      for (int h = 0; h < H; ++h) sm += rm[h] * sg[h] + kEps;
#pragma unroll 6
      for (int h = 0; h < H; ++h) sn += rn[h] + kEps;
      double y = __builtin_amdgcn_rcp(sm);
      double er = __builtin_fma(-sm, y, 1.0);
      y = __builtin_fma(y, er, y);
      er = __builtin_fma(-sm, y, 1.0);
      qs[4 * k + 0] = sm;
      qs[4 * k + 1] = __builtin_fma(y, er, y);
      y = __builtin_amdgcn_rcp(sn);
      er = __builtin_fma(-sn, y, 1.0);
      y = __builtin_fma(y, er, y);
      er = __builtin_fma(-sn, y, 1.0);
      qs[4 * k + 2] = sn;
      qs[4 * k + 3] = __builtin_fma(y, er, y);
      t_qrec[k] = n | (pr << 9) | (pm << 17) | ((root_player ^ (t_depth[n] & 1)) << 18) | (t_act[n] << 19);
    }
    RBL_F2();  // query sums
    __syncthreads();
    // the rows themselves: one item per element, consecutive threads = consecutive floats of the exchange buffer.
    // (x + eps) / s is hipcc's f64 division sequence minus v_div_scale / v_div_fixup, the identity here
    // (1e-80 <= x + eps <= s <= H + 1)
    float* gq = a.queries + (size_t)row_off * Q;
    const int cnt = L * Q;
    for (int i0 = tid; i0 < cnt; i0 += kU * NT) {
      double num[kU], s[kU], y[kU];
      float flat[kU];
      int kind[kU];  // -1: nothing to store, 0: `flat`, 1: the quotient
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int i = min(i0 + u * NT, cnt - 1);
        const int k = i / Q, j = i - k * Q;
        const int rec = t_qrec[k];
        const int n = rec & 511, pr = (rec >> 9) & 255, pm = (rec >> 17) & 1;
        const int jj = max(j - 2 - A, 0);
        const int wh = jj / H, h = jj % H;  // player whose reach this is
        const bool acted = wh == pm;
        const double r = (wh == 0 ? rho0 : rho1)[pr * H + h];
        const double sg = acted ? sig[(n - 1) * H + h] : 1.0;
        num[u] = (acted ? r * sg : r) + kEps;
        s[u] = qs[4 * k + (acted ? 0 : 2)];
        y[u] = qs[4 * k + (acted ? 1 : 3)];
        flat[u] = j == 0 ? (float)((rec >> 18) & 1) : (j == 1 ? (float)a.next_trav : (j - 2 == (rec >> 19) ? 1.0f : 0.0f));
        kind[u] = i0 + u * NT < cnt ? (j < 2 + A ? 0 : 1) : -1;
      }
#pragma unroll
      for (int u = 0; u < kU; ++u)
        if (kind[u] >= 0) {
          const double q0 = num[u] * y[u];
          const double rem = __builtin_fma(-s[u], q0, num[u]);
          gq[i0 + u * NT] = kind[u] == 0 ? flat[u] : (float)__builtin_fma(rem, y[u], q0);
        }
    }
  }
  RBL_STAMP();  // 8: queries
#undef RBL_STAMP
}

}  // namespace

size_t cfr_flat_lds_bytes(int N, int NI, int H, int L, int T, int faces) {
  const size_t d = (size_t)3 * NI * H + (size_t)(N - L) * H + (size_t)5 * L + (size_t)(N - 1) * H;  // doubles
  const size_t b = d * 8 + (((size_t)7 * N + 2 * L + T + 1) & ~(size_t)1) * 4 + (size_t)faces * 16;
  return (b + 15) & ~(size_t)15;
}

bool cfr_flat_supported(int H, int A, int dice, int faces) { return H == 36 && A == 25 && dice == 2 && faces == 6; }

bool launch_cfr_flat(const CfrArgs& a, int B, size_t lds_bytes, int threads, hipStream_t stream) {
  if (threads < 64 || threads > kFlatMaxThreads || threads % 64) throw std::runtime_error("cfr_flat: bad block size");
  if (!(a.H == 36 && a.A == 25 && a.dice == 2)) return false;
  auto kern = cfr_flat_kernel<36, 25, 2, 6>;
  static std::once_flag attr_once[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) throw std::runtime_error("cfr_flat: no current device");
  hipError_t attr_err = hipSuccess;
  std::call_once(attr_once[dev], [&] {
    attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  if (attr_err != hipSuccess)
    throw std::runtime_error(std::string("cfr_flat: cannot request 160 KB of LDS: ") + hipGetErrorString(attr_err));
  hipLaunchKernelGGL(kern, dim3(B), dim3(threads), lds_bytes, stream, a);
  return true;
}

}  // namespace rbl
