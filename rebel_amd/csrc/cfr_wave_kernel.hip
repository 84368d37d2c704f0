// rebel_amd/csrc/cfr_wave_kernel.hip -- CFR::step for the common games, ONE WAVEFRONT per lane, element-parallel.
//
// cfr_rows_kernel.hip gives every tree ROW (H hands of one node / edge) to a thread of a 128-thread workgroup.  PMC
// (profiles/r01_pmc_summary.txt) showed it issue-bound at ~3 260 wave-instructions per lane and step: the sequential-
// in-action sums of a level run on as many threads as the level has parents (12, or 1 at the root) while each of them
// issues H fp64 operations per child, the second wave of a lane mostly executes loop control, and a step has ~15
// workgroup barriers.  Here a lane is one wave and the unit of work is an ELEMENT (row, hand):
//   * sums over the actions of a node run on one thread per (node, hand): the same 12-long sequential chains, but 6x
//     more of them per instruction (root tree: 78 threads instead of 13);
//   * regret update, regret matching, normalisation, new reach, strategy sums and the query rows are flat maps over
//     the edge elements of the traverser's level; each thread's view of its elements (regret, strategy sum, the LDS offset of
//     the parent's rows from a per-shape table) is built once and only for the strides that level has (round 5);
//   * only the per-leaf work that needs a whole reach row at once (scale sum, terminal match histogram) stays
//     row-per-thread;
//   * no workgroup barriers: the phases of a lane are ordered by the wave's own in-order LDS pipeline
//     (wave-scope fences only keep the compiler from reordering);
//   * regrets never enter LDS (they are read and written once, by the owning thread, straight from / to the lane's
//     slab), and of a query row only what changes between iterations is rewritten (the traverser flag and the two
//     normalised reach vectors; player id and last-bid one-hot were written when the solver was built).
// Arithmetic is operation for operation that of cfr_rows_kernel.hip / cfr_kernels.hip (same operands, same order,
// -ffp-contract=off, explicit fmas only where those kernels have them), so the bit-exactness contract with the reference
// (subgame_solving.cc:538-664) is unchanged; tests/test_cfr_parity.py and tests/test_selfplay_parity.py run against it.
// The LDS image of a lane (8.7 KB at the 1 die x 6 faces root: 18 lanes per CU) keeps pseudo-leaf values as floats -- they are
// floats by construction -- and the deepest level's terminals as one fp64 row per parent: see "LDS layout" below.
// Only kModeStep of LDS-resident lanes runs here (RBL_CFR_WAVE=0 switches back to the row kernel).
#include <algorithm>
#include <type_traits>

#include "cfr_kernels.h"
#include "launch_timing.h"

namespace rbl {

namespace {

constexpr double kEps = 1e-80;
constexpr size_t kWaveLdsSlack = 128;  // bytes behind the lane's LDS image that the unconditional staging stores may touch
                                       // (cfr_wave_lds_bytes additionally covers the KT*64-dword store of the byte tables)

template <int H>
struct Row {
  double v[H];
};
template <int H>
__device__ __forceinline__ Row<H> load_row(const double* p) {
  Row<H> r;
  if constexpr (H % 2 == 0) {
    // every row of the LDS image starts at a multiple of 8 H bytes from a 16-byte aligned base: with an even H a thread reads its
    // row in 16-byte pieces (round 6: a wave's ds_read_b64 of rows 8 H bytes apart is a 2-way bank conflict at H = 6, its
    // ds_read_b128 none -- the lesson of the 2 dice x 6 faces kernel, profiles/r06_cfr_flat_node_values.txt)
    typedef double d2 __attribute__((ext_vector_type(2)));
    const d2* p2 = reinterpret_cast<const d2*>(p);
#pragma unroll
    for (int h = 0; h < H / 2; ++h) {
      const d2 v = p2[h];
      r.v[2 * h] = v[0];
      r.v[2 * h + 1] = v[1];
    }
  } else {
#pragma unroll
    for (int h = 0; h < H; ++h) r.v[h] = p[h];
  }
  return r;
}
template <int H>
__device__ __forceinline__ void store_row(double* p, const Row<H>& r) {
  if constexpr (H % 2 == 0) {  // 16-byte pieces, as load_row
    typedef double d2 __attribute__((ext_vector_type(2)));
    d2* p2 = reinterpret_cast<d2*>(p);
#pragma unroll
    for (int h = 0; h < H / 2; ++h) p2[h] = d2{r.v[2 * h], r.v[2 * h + 1]};
  } else {
#pragma unroll
    for (int h = 0; h < H; ++h) p[h] = r.v[h];
  }
}

// orders this wave's LDS / global accesses across a phase boundary: the LDS pipeline serves a wave's requests in
// order, so no hardware barrier is needed -- only the compiler must not move accesses across
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ double refine_rcp(double s) {  // v_rcp_f64 + two Newton steps: the denominator part of `/`
  double y = __builtin_amdgcn_rcp(s);
  double e = __builtin_fma(-s, y, 1.0);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(-s, y, 1.0);
  return __builtin_fma(y, e, y);
}
__device__ __forceinline__ double div_by(double num, double s, double y) {  // == num / s for 1e-80 <= num <= s (see rows kernel)
  const double q0 = num * y;
  const double rem = __builtin_fma(-s, q0, num);
  return __builtin_fma(rem, y, q0);
}

// EHM / LHM / NM: upper bounds of E*H, L*H, N over the engine's shapes (fix the number of staging strides)
template <int H, int A, int DICE, int FACES, int EHM, int LHM, int NM>
__global__ void __launch_bounds__(64, H == 9 ? 4 : 5) cfr_wave_kernel(const CfrArgs a) {  // (>= 5 waves per SIMD: the LDS image allows 18 lanes per CU)
  extern __shared__ __align__(16) double lds[];
  constexpr int Q = 2 + A + 2 * H, NB = 2 * DICE + 1, W = 64;
  constexpr int KS = (EHM + W - 1) / W;
  constexpr int KT = ((5 * NM + FACES * H + 3) / 4 + W - 1) / W;  // dword strides of the table blob (4 N + L + T <= 5 N bytes, + the match table)
  const int lane = a.lane0 + blockIdx.x, tid = threadIdx.x;
  // everything this lane must know before its first vector load, in ONE scalar load (cfr_kernels.h: LaneRec; it was lane_shape ->
  // shapes[] -> wave_tab_off / wave_epv_off plus three per-lane arrays: three dependent scalar round trips in front of the staging)
  typedef const LaneRecWords __attribute__((address_space(4)))* crec_p;
  const LaneRecWords rec = ((crec_p)a.lane_rec)[lane];
  if (rec[kRecFlags] & kRecSkip) return;  // root de-duplication: this root lane is served by the epoch's representative
  const int N = rec[kRecN], E = N - 1, L = rec[kRecL], T = rec[kRecT], NI = rec[kRecNI], nlev = rec[kRecNlev];
  const int root_player = rec[kRecRootPlayer], row_off = rec[kRecRowOff];
  // level offsets of a tree of depth <= 2: {0, 1, lo2, N} (lo2 == N when it has two levels)
  auto lev_off = [&](int d) { return d <= 0 ? 0 : (d == 1 ? 1 : (d == 2 ? rec[kRecLo2] : N)); };
  const int t = a.trav, opp = 1 - t;
  const int EH = E * H, LH = L * H;

  const size_t lane_e = (size_t)lane * a.Emax * H;
  double* g_sig = a.sigma + lane_e;
  double* g_reg = a.regrets + lane_e;
  double* g_sum = a.sums + lane_e;
  const double* bel = a.beliefs + (size_t)lane * 2 * H;
  double* rmean = a.root_mean + (size_t)lane * 2 * H;
  const bool snap_now = rec[kRecActIter] == a.steps_after;  // (act_iter < 0: no snapshot; steps_after >= 1)

  // ---- LDS layout
  // Node values.  A pseudo-leaf's value is a float by construction ((double)(float)(net row x reach sum), :257-268), and
  // pseudo-leaves only exist on the DEEPEST level (a node above max_depth that is not terminal has children): that level's rows
  // are stored as floats, indexed by node (the slots of its terminals stay unused).  A terminal of the deepest level is the
  // LAST child of its parent (the liar call is the highest action; the host checks both facts per shape): its fp64 row is
  // kept per PARENT.  Everything above the deepest level keeps an fp64 row per node.  8.7 KB per root lane instead of 10.0 KB at
  // 1 die x 6 faces = 18 lanes per CU instead of 16 (12.7 instead of 14.7 KB at 2 dice x 3 faces: 12 instead of 10).
  const int lo1 = 1, lo2 = rec[kRecLo2];
  const int lo_d = nlev == 3 ? lo2 : lo1, lo_p = nlev == 3 ? lo1 : 0;  // first node of the deepest level / of its parents' level
  const int NP = lo_d - lo_p, ND = N - lo_d;
  double* sig = lds;                  // [E][H]
  double* vald = sig + EH;            // [lo_d][H]   values of the nodes above the deepest level
  double* vterm = vald + lo_d * H;    // [NP][H]     value of the terminal child of deepest-level parent p at row p - lo_p
  // Reach rows are kept for the nodes that have children.  In a tree of depth <= 2 (all this kernel takes) only the ROOT's
  // mover has a reach that differs between those rows; the other player's reach is the same at the root and at every depth-1
  // node (it moves at depth 1, i.e. into nodes that keep no row): one row instead of NI.
  double* rho_rp = vterm + NP * H;    // [NI][H] reach of the root's mover
  double* rho_op = rho_rp + NI * H;   // [1][H]  reach of the other player
  float* valf = reinterpret_cast<float*>(rho_op + H);  // [ND][H] floats: pseudo-leaf values of the deepest level, row c - lo_d
  auto rrow = [&](int pl, int ir) -> double* { return pl == root_player ? rho_rp + ir * H : rho_op; };
  // nodes with children are a prefix of the BFS order here (the host checks it), and a node's depth follows from the level
  // offsets: no irank / depth tables
  auto irank = [&](int n) { return n < NI ? n : -1; };
  auto dpar = [&](int n) { return ((n >= lo1) + (n >= lo2)) & 1; };
  // (the regret-matching row sums live in value rows that are dead by then -- the terminal rows of the level they normalise, or
  // the depth-1 rows when the root is normalised; their refined reciprocals are recomputed where they are used)
  // tree tables as bytes: every entry is a node id, an action, a row index or -1, all < 128 for these games (NM <= 127)
  int8_t* tb = reinterpret_cast<int8_t*>(valf + ((ND * H + 1) & ~1));
  int8_t *t_parent = tb, *t_act = tb + N, *t_cb = tb + 2 * N, *t_ce = tb + 3 * N;
  int8_t *t_leaf = tb + 4 * N, *t_term = t_leaf + L;  // t_leaf[k]: node of net row k; t_term[j]: j-th terminal
  int8_t* t_match = t_term + T;

  long long* dbg = a.dbg ? a.dbg + (size_t)lane * 16 : nullptr;
  int dbg_k = 0;
#define RBL_STAMP()                                                 \
  do {                                                              \
    if (dbg && tid == 0) dbg[dbg_k] = (long long)clock64();         \
    ++dbg_k;                                                        \
  } while (0)
  RBL_STAMP();  // 0

  // ---------------------------------------------------------------- stage: every global load in flight before the first store.
  // Loads and LDS stores are UNCONDITIONAL, at compile-time offsets from one per-thread base (no exec-mask branches -- they
  // were a third of this kernel's scalar instructions -- and clamps only on the strides that can reach past the lane's own
  // data: round 3, PMC reads 140.7 -> 109.9 MB per 16 384-lane launch for +0.3 us): the engine pads every global array these loads
  // can overrun (kWavePad), and the LDS stores run front to back through the lane's image -- sigma, root reach rows,
  // leaf values, the byte tables in layout order -- so whatever a too-long store spills into the following arrays is
  // overwritten by the stores that own them (the LDS pipeline keeps a wave's stores in order); the image is allocated
  // with kWaveLdsSlack bytes behind the last table for the final overrun.
  double bel_t = 0.0, rmean_t = 0.0;
  // leaf values of net rows k = tid + 64 u, read by the thread that turns them into node values (unconditional loads:
  // the values buffer is padded)
  constexpr int KL = (LHM / H + W - 1) / W;
  float lv_[KL][H];
  {
    const float* gv = a.values + (size_t)row_off * H;
    // one wave-uniform choice of how many 64-row strides the lane's L rows need (a test per stride would serialise the
    // loads; always loading the maximum made small trees read their neighbours' rows: 1.9x fabric traffic).  Within a stride
    // the row index is CLAMPED to the lane's last row: threads past it re-read that row (a cache hit) instead of the next
    // lanes' rows, which live in other XCDs' L2 slices and came over the fabric a second time.
    auto load_lv = [&](auto kc) {
      constexpr int K = decltype(kc)::value;
#pragma unroll
      for (int u = 0; u < K; ++u) {  // only the last stride of the choice can reach past the lane's rows
        const float* gr = gv + (size_t)(u + 1 < K ? tid + u * W : min(tid + u * W, max(L - 1, 0))) * H;  // (L == 0 never gets here; the max keeps the index non-negative regardless)
#pragma unroll
        for (int h = 0; h < H; ++h) lv_[u][h] = gr[h];
      }
    };
    if (KL >= 2 && L > W)
      load_lv(std::integral_constant<int, KL>{});
    else if (L > 0)
      load_lv(std::integral_constant<int, 1>{});
  }
  {
    double s_[KS];
    const double* gs0 = g_sig + tid;
    constexpr int KS3 = (KS + 2) / 3, KS23 = (2 * KS + 2) / 3;  // thirds of the stride count, same wave-uniform choice
    auto load_sig = [&](auto kc) {
      constexpr int K = decltype(kc)::value;
      // clamped to the lane's last element: the tail of the last stride re-reads it instead of the next lane's slab
      // (only the last third of the chosen stride count can reach past it: the choice is the smallest third that covers EH)
#pragma unroll
      for (int u = 0; u < K; ++u) s_[u] = u < K - KS3 ? gs0[u * W] : g_sig[min(tid + u * W, max(EH - 1, 0))];
    };
    const int sig_strides = EH <= KS3 * W ? KS3 : (EH <= KS23 * W ? KS23 : KS);
    if (sig_strides == KS3)
      load_sig(std::integral_constant<int, KS3>{});
    else if (sig_strides == KS23)
      load_sig(std::integral_constant<int, KS23>{});
    else
      load_sig(std::integral_constant<int, KS>{});
    // (the net's output rows go straight into the registers of the thread that will consume them: lv_ below)
    // the lane's tree tables: one byte blob per shape in exactly the LDS layout (parent, act, cb, ce: N bytes each; leaf
    // nodes: L; terminal nodes: T; the game's match table: FACES x H), copied dword-wise (was: eight int tables, 16 loads per
    // thread, 4 bytes per entry, and a separate load + store for the match table)
    const int* gt = reinterpret_cast<const int*>(a.wave_tabs + rec[kRecTabOff]) + tid;
    int tw[KT];
#pragma unroll
    for (int u = 0; u < KT; ++u) tw[u] = gt[u * W];
    if (tid < H) {
      bel_t = bel[t * H + tid];
      rmean_t = rmean[t * H + tid];
    }
    const double b0 = tid < H ? (t == 0 ? bel_t : bel[tid]) : 0.0, b1 = tid < H ? (t == 1 ? bel_t : bel[H + tid]) : 0.0;
    auto store_sig = [&](auto kc) {
      constexpr int K = decltype(kc)::value;
#pragma unroll
      for (int u = 0; u < K; ++u) sig[tid + u * W] = s_[u];
    };
    if (sig_strides == KS3)
      store_sig(std::integral_constant<int, KS3>{});
    else if (sig_strides == KS23)
      store_sig(std::integral_constant<int, KS23>{});
    else
      store_sig(std::integral_constant<int, KS>{});
    if (tid < H) {
      rrow(0, 0)[tid] = b0;
      rrow(1, 0)[tid] = b1;
    }
#pragma unroll
    for (int u = 0; u < KT; ++u) reinterpret_cast<int*>(tb)[tid + u * W] = tw[u];
    static_assert(NM <= 127, "byte tables");
  }
  wave_sync();
  RBL_STAMP();  // 1: staged

  // ---------------------------------------------------------------- reach of both players under sigma (:54-78) and the values
  // of the nodes without children (query_value_net :257-268, terminal payoffs :80-98).  Three uniform passes instead of
  // one per level with a three-way branch per node (pseudo-leaves and liar-call terminals alternate in node order, so a
  // wave ran every branch for every node): reach rows of the nodes WITH children, level by level; then every
  // pseudo-leaf (its net row order is the leaf list); then every terminal.  A childless node's value needs its whole
  // opponent-reach row at once (scale sum / match histogram): those two passes are row-per-thread.
  auto opp_reach_row = [&](int n) {  // opponent's reach at childless node n: the parent's row, times sigma if the opponent acted
    const int p = t_parent[n];
    Row<H> ro = load_row<H>(rrow(opp, irank(p)));
    if ((root_player ^ dpar(p)) == opp) {
      const Row<H> sg = load_row<H>(sig + (n - 1) * H);
#pragma unroll
      for (int h = 0; h < H; ++h) ro.v[h] = ro.v[h] * sg.v[h];
    }
    return ro;
  };
  if (nlev > 2) {  // depth-1 nodes with children: the root's mover acted on the way in (the other player's row is the root's)
    for (int n = lo1 + tid; n < NI; n += W) {
      Row<H> rm = load_row<H>(rho_rp);
      const Row<H> sg = load_row<H>(sig + (n - 1) * H);
#pragma unroll
      for (int h = 0; h < H; ++h) rm.v[h] = rm.v[h] * sg.v[h];
      store_row<H>(rho_rp + n * H, rm);
    }
    wave_sync();
  }
#pragma unroll
  for (int u = 0; u < KL; ++u) {  // pseudo-leaves: net output row k, scaled by the opponent's total reach (:257-268)
    const int k = tid + u * W;
    if (k < L) {
      const int n = t_leaf[k];
      const Row<H> ro = opp_reach_row(n);  // (n >= 1: the host only sends trees whose root has children)
      double ssum = 0.0;
#pragma unroll
      for (int h = 0; h < H; ++h) ssum += ro.v[h];
      float* out = valf + (n - lo_d) * H;
#pragma unroll
      for (int h = 0; h < H; ++h) out[h] = (float)((double)lv_[u][h] * ssum);
    }
  }
  for (int j = tid; j < T; j += W) {  // terminals: the bid that was called is the parent's last bid (:80-98, :765-789)
    const int n = t_term[j];
    const Row<H> ro = opp_reach_row(n);
    const int par = t_parent[n];
    const int bid = t_act[par];
    const int qty = 1 + bid / FACES, face = bid % FACES;
    const int8_t* m = t_match + face * H;
    double b[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) b[k] = 0.0;
    double ssum = 0.0;
#pragma unroll
    for (int h = 0; h < H; ++h) {  // b[m[h]] += r, without dynamic register indexing (x + 0.0 == x for x >= +0)
      const int mh = m[h];
#pragma unroll
      for (int k = 0; k <= DICE; ++k) b[k] += (mh == k) ? ro.v[h] : 0.0;  // one hand shows at most DICE matches
      ssum += ro.v[h];
    }
#pragma unroll
    for (int k = DICE - 1; k >= 0; --k) b[k] += b[k + 1];  // bins above DICE are +0.0: adding them changes no bit
    const bool inverse = (root_player ^ dpar(n)) != t;
    double cand[DICE + 1];
#pragma unroll
    for (int mm = 0; mm <= DICE; ++mm) {
      const int left = max(0, qty - mm);
      double bl = b[0];
#pragma unroll
      for (int k = 1; k < NB; ++k) bl = (left == k) ? b[k] : bl;
      cand[mm] = (double)(float)bl * 2 - ssum;  // fp32 truncation (:785)
      if (inverse) cand[mm] *= -1.0;
    }
    Row<H> out;
#pragma unroll
    for (int h = 0; h < H; ++h) {
      const int mh = m[h];
      double x = cand[0];
#pragma unroll
      for (int mm = 1; mm <= DICE; ++mm) x = (mh == mm) ? cand[mm] : x;
      out.v[h] = x;
    }
    store_row<H>(n >= lo_d ? vterm + (par - lo_p) * H : vald + n * H, out);
  }
  wave_sync();
  RBL_STAMP();  // 2: reach + leaf values
  RBL_STAMP();  // 3
  RBL_STAMP();  // 4

  // ---------------------------------------------------------------- bottom-up (update_regrets :542-574) fused with regret
  // matching (:619-634) and the regret discount (:639-650); element-parallel
  // the traverser's reach rows: the full set if it moves at the root, else the single row (index multiplier 0)
  double* rho_t = t == root_player ? rho_rp : rho_op;
  const int tmul = t == root_player ? 1 : 0;
  // Per-thread view of a traverser level's edge elements i = tid + 64 u (element = (child c, hand h) of the level below):
  // its regret and strategy-sum values (requested from global memory up front) and the LDS offset pv = parent * H + h of its
  // parent's value row -- which is also the offset of the parent's reach row and row-sum row (a parent has children, so its
  // reach-row rank is its node id).  pv comes from a per-shape table (one 2-byte load next to the regret load; it used to be a
  // division by H, a modulo, a byte-table look-up and a multiply-add per element).  The regret update, the normalisation and
  // the strategy-sum update all walk this same element set.  Depth-2 subgames have exactly one traverser level; for deeper
  // trees the arrays of the LAST processed (shallowest) one stay alive for the write-back and the others are redone there.
  // K: how many 64-element strides the traverser's widest level needs -- ONE wave-uniform choice among thirds of the maximum
  // (the same idea as the sigma staging): when the traverser owns the 12 root edges (72 elements at 1 die x 6 faces) the other
  // 6-7 strides used to issue their clamped loads and exec-mask branches for nothing in every pass.  Only the strides of the
  // last third can reach past the level (K is the smallest third that covers it).
  constexpr int KS3 = (KS + 2) / 3, KS23 = (2 * KS + 2) / 3;
  int ch_max = 0;
  for (int lev = nlev - 2; lev >= 0; --lev)
    if ((root_player ^ (lev & 1)) == t) ch_max = max(ch_max, (lev_off(lev + 2) - lev_off(lev + 1)) * H);
  const unsigned short* epv = a.wave_epv + rec[kRecEpvOff];
  auto tail = [&](auto kc) {
  constexpr int K = decltype(kc)::value;
  constexpr int KFULL = K > KS3 ? K - KS3 : 0;  // strides that lie wholly inside the level
  double rq[K], gs_[K];
  int pv[K];
  int lev_kept = -1;
  for (int lev = nlev - 2; lev >= 0; --lev) {
    const int n0 = lev_off(lev), n1 = lev_off(lev + 1);
    const int c_lo = n1, c_hi = lev_off(lev + 2);
    const bool mine = (root_player ^ (lev & 1)) == t;
    const int nh = (n1 - n0) * H, ch = (c_hi - c_lo) * H;
    if (mine) {
      lev_kept = lev;
#pragma unroll
      for (int u = 0; u < K; ++u) {
        const int i = u < KFULL && ch == ch_max ? tid + u * W : max(0, min(tid + u * W, ch - 1));
        rq[u] = g_reg[(c_lo - 1) * H + i];
        gs_[u] = g_sum[(c_lo - 1) * H + i];
        pv[u] = epv[(c_lo - 1) * H + i];
      }
    }
    const bool deep = lev + 2 == nlev;  // the children are the deepest level: float rows + one terminal row per parent
    for (int i = tid; i < nh; i += W) {  // node value of (n, h): sequential over the actions, ascending
      const int n = n0 + i / H, h = i % H;
      const int c0 = t_cb[n], c1 = t_ce[n];
      if (c0 == c1) continue;
      // the additions stay sequential in ascending action order; the LDS reads of four actions are issued together
      // (reads past the node's last child land in other rows of the lane's LDS image and are never added: a SELECT replaces
      // them by +0.0 -- adding +0.0 changes no bit, the accumulator starts at +0.0 and a round-to-nearest sum is never -0.0
      // unless both operands are -- where round 4 had an exec-mask branch per child)
      double x = 0.0;
      const double* ps_ = sig + (c0 - 1) * H + h;
      if (deep) {
        const bool has_term = t_act[c1 - 1] == A - 1;  // the liar call, always the last child
        const int cl = c1 - (has_term ? 1 : 0);        // the children before it are pseudo-leaves
        const float* pf_ = valf + (c0 - lo_d) * H + h;
        if (mine) {
          for (int c = c0; c < cl; c += 4, pf_ += 4 * H, ps_ += 4 * H) {
            const float v0 = pf_[0], v1 = pf_[H], v2 = pf_[2 * H], v3 = pf_[3 * H];
            const double s0 = ps_[0], s1 = ps_[H], s2 = ps_[2 * H], s3 = ps_[3 * H];
            x += (double)v0 * s0;
            x += c + 1 < cl ? (double)v1 * s1 : 0.0;
            x += c + 2 < cl ? (double)v2 * s2 : 0.0;
            x += c + 3 < cl ? (double)v3 * s3 : 0.0;
          }
          if (has_term) x += vterm[(n - lo_p) * H + h] * sig[(c1 - 2) * H + h];
        } else {
          for (int c = c0; c < cl; c += 4, pf_ += 4 * H) {
            const float v0 = pf_[0], v1 = pf_[H], v2 = pf_[2 * H], v3 = pf_[3 * H];
            x += (double)v0;
            x += c + 1 < cl ? (double)v1 : 0.0;
            x += c + 2 < cl ? (double)v2 : 0.0;
            x += c + 3 < cl ? (double)v3 : 0.0;
          }
          if (has_term) x += vterm[(n - lo_p) * H + h];
        }
      } else {
        const double* pv_ = vald + c0 * H + h;
        if (mine) {
          for (int c = c0; c < c1; c += 4, pv_ += 4 * H, ps_ += 4 * H) {
            const double v0 = pv_[0], v1 = pv_[H], v2 = pv_[2 * H], v3 = pv_[3 * H];
            const double s0 = ps_[0], s1 = ps_[H], s2 = ps_[2 * H], s3 = ps_[3 * H];
            x += v0 * s0;
            x += c + 1 < c1 ? v1 * s1 : 0.0;
            x += c + 2 < c1 ? v2 * s2 : 0.0;
            x += c + 3 < c1 ? v3 * s3 : 0.0;
          }
        } else {
          for (int c = c0; c < c1; c += 4, pv_ += 4 * H) {
            const double v0 = pv_[0], v1 = pv_[H], v2 = pv_[2 * H], v3 = pv_[3 * H];
            x += v0;
            x += c + 1 < c1 ? v1 : 0.0;
            x += c + 2 < c1 ? v2 : 0.0;
            x += c + 3 < c1 ? v3 : 0.0;
          }
        }
      }
      vald[n * H + h] = x;
    }
    wave_sync();
    if (!mine) continue;
    double* lsig = sig + (c_lo - 1) * H;
    double* greg = g_reg + (c_lo - 1) * H;
    // the child's value: a float row or (flag of the element's table entry) its parent's terminal row on the deepest level,
    // an fp64 row per node above it
    const double* lval = vald + c_lo * H;
    const double* ltrm = vterm - lo_p * H;
#pragma unroll
    for (int u = 0; u < K; ++u) {  // regret update + regret matching numerators, one thread per edge element
      const int i = tid + u * W;
      if ((u < KFULL && ch == ch_max) || i < ch) {
        double q = rq[u];
        const int po = pv[u] & 0x7fff;
        double cv;
        if (deep) {
          const double ct = ltrm[po];
          const double cf = (double)valf[i];
          cv = (pv[u] & 0x8000) ? ct : cf;
        } else {
          cv = lval[i];
        }
        q += cv;
        q -= vald[po];
        lsig[i] = q > kEps ? q : kEps;
        greg[i] = q * (q > 0 ? a.pos : a.neg);
      }
    }
    wave_sync();
    // the children's values have been consumed (regret update above): their rows now hold the row sums, one row per node
    // of this level that has children (those nodes' reach-row ranks are consecutive; every one of them has >= 1 child)
    double* ysum = deep ? vterm : vald + c_lo * H;  // [n - n0][H]
    for (int i = tid; i < nh; i += W) {  // row sums of (n, h), sequential over the actions
      const int n = n0 + i / H, h = i % H;
      const int c0 = t_cb[n], c1 = t_ce[n];
      if (c0 == c1) continue;
      double s = 0.0;
      const double* ps_ = sig + (c0 - 1) * H + h;
      for (int c = c0; c < c1; c += 4, ps_ += 4 * H) {
        const double s0 = ps_[0], s1 = ps_[H], s2 = ps_[2 * H], s3 = ps_[3 * H];
        s += s0;
        s += c + 1 < c1 ? s1 : 0.0;
        s += c + 2 < c1 ? s2 : 0.0;
        s += c + 3 < c1 ? s3 : 0.0;
      }
      ysum[(n - n0) * H + h] = s;  // (a node with children: its reach-row rank is its node id)
    }
    wave_sync();
#pragma unroll
    for (int u = 0; u < K; ++u) {
      const int i = tid + u * W;
      if ((u < KFULL && ch == ch_max) || i < ch) {
        const double s = ysum[(pv[u] & 0x7fff) - n0 * H];
        lsig[i] = div_by(lsig[i], s, refine_rcp(s));
      }
    }
    wave_sync();
  }
  RBL_STAMP();  // 5: bottom-up

  // ---------------------------------------------------------------- running mean of the root values (:579-590)
  if (tid < H) {
    double m = rmean_t;
    m += (vald[tid] - m) * a.alpha;
    rmean[t * H + tid] = m;
  }
  // ---------------------------------------------------------------- traverser's reach under the NEW sigma (:636-638), rows of
  // nodes with children only (the root row still holds the traverser's beliefs)
  if (nlev > 2 && tmul) {  // depth-1 rows of the root's mover, if it is the traverser (the other player's single row stands)
    const int nh = (NI - lo1) * H;
    for (int i = tid; i < nh; i += W) {
      const int n = lo1 + i / H, h = i % H;
      rho_rp[n * H + h] = rho_rp[h] * sig[(n - 1) * H + h];
    }
    wave_sync();
  }
  RBL_STAMP();  // 6: new reach

  // ---------------------------------------------------------------- sum_strategies (:651-657) + write back what changed:
  // the traverser's levels only (contiguous edge ranges), from the per-thread view built in the bottom-up sweep
  for (int lev = 0; lev < nlev - 1; ++lev) {
    if ((root_player ^ (lev & 1)) != t) continue;
    const int c_lo = lev_off(lev + 1), c_hi = lev_off(lev + 2);
    const int ch = (c_hi - c_lo) * H, e0 = (c_lo - 1) * H;
    if (lev == lev_kept) {
#pragma unroll
      for (int u = 0; u < K; ++u) {
        const int i = tid + u * W;
        if ((u < KFULL && ch == ch_max) || i < ch) {
          const double sg = sig[e0 + i];
          double x = gs_[u];
          x *= a.strat;
          x += rho_t[tmul ? (pv[u] & 0x7fff) : (tid + u * W) % H] * sg;
          g_sum[e0 + i] = x;
          g_sig[e0 + i] = sg;
        }
      }
    } else {  // (only trees deeper than this kernel takes have a second traverser level)
      for (int i = tid; i < ch; i += W) {
        const int ir = irank(t_parent[c_lo + i / H]), h = i % H;
        const double sg = sig[e0 + i];
        double x = g_sum[e0 + i];
        x *= a.strat;
        x += rho_t[ir * tmul * H + h] * sg;
        g_sum[e0 + i] = x;
        g_sig[e0 + i] = sg;
      }
    }
  }
  };
  if (ch_max <= KS3 * W)
    tail(std::integral_constant<int, KS3>{});
  else if (ch_max <= KS23 * W)
    tail(std::integral_constant<int, KS23>{});
  else
    tail(std::integral_constant<int, KS>{});
  if (snap_now) {
    double* snap = a.snapshot + lane_e;
    for (int i = tid; i < EH; i += W) snap[i] = sig[i];
  }
  if (rec[kRecFlags] & kRecRep) {  // the representative keeps sigma after EVERY iteration for the root lanes it serves
    double* snap = a.snap_all + (size_t)a.steps_after * a.Emax * H;
    for (int i = tid; i < EH; i += W) snap[i] = sig[i];
  }
  RBL_STAMP();  // 7: write-back

  // ---------------------------------------------------------------- queries for the next step (:253-269, :104-123): the
  // traverser flag and the two normalised reach vectors of every pseudo-leaf row
  if (a.next_trav >= 0 && L > 0) {
    wave_sync();  // the traverser's reach rows were rewritten above
    // One thread per (pseudo-leaf, player): that player's reach row at the leaf = the parent's row (times the new sigma
    // where the player acted at the parent), normalised (normalize_probabilities_safe, util.h:68-78) and written as H
    // floats into the leaf's query row.  The opponent's rows still hold the reach under the sigma of this step's start
    // (it did not change), the traverser's were just recomputed: together the current strategy profile's reach.
    float* gq = a.queries + (size_t)row_off * Q;
    const float trav_flag = (float)a.next_trav;
    for (int j = tid; j < 2 * L; j += W) {
      const int k = j >> 1, pl = j & 1, n = t_leaf[k];
      Row<H> r;
      const int ir = irank(n);
      if (ir >= 0) {  // only the root can be a pseudo-leaf with a stored row (max_depth = 0)
        r = load_row<H>(rrow(pl, ir));
      } else {
        const int p = t_parent[n];
        r = load_row<H>(rrow(pl, irank(p)));
        if ((root_player ^ dpar(p)) == pl) {
          const Row<H> sg = load_row<H>(sig + (n - 1) * H);
#pragma unroll
          for (int h = 0; h < H; ++h) r.v[h] = r.v[h] * sg.v[h];
        }
      }
      double ssum = 0.0;
#pragma unroll
      for (int h = 0; h < H; ++h) ssum += r.v[h] + kEps;
      const double y = refine_rcp(ssum);
      if (a.q_dyn) {  // split layout: (traverser, reach0, reach1, pads) as one contiguous row per pseudo-leaf
        float* qd = a.q_dyn + ((size_t)row_off + k) * a.q_dyn_stride;
#pragma unroll
        for (int h = 0; h < H; ++h) qd[1 + pl * H + h] = (float)div_by(r.v[h] + kEps, ssum, y);
        if (pl == 0) qd[0] = trav_flag;
        else
          for (int z = 1 + 2 * H; z < a.q_dyn_stride; ++z) qd[z] = 0.f;
        continue;
      }
      float* q = gq + k * Q + 2 + A + pl * H;
#pragma unroll
      for (int h = 0; h < H; ++h) q[h] = (float)div_by(r.v[h] + kEps, ssum, y);
      if (pl == 0) gq[k * Q + 1] = trav_flag;
    }
  }
  RBL_STAMP();  // 8: queries
#undef RBL_STAMP
}

}  // namespace

size_t cfr_wave_lds_bytes(int N, int NI, int H, int L, int T, int faces, int lo_d, int lo_p) {
  // doubles: sigma, values above the deepest level, one terminal row per deepest-level parent, reach rows; floats: the
  // deepest level's rows (kernel: "LDS layout")
  const size_t d = (size_t)(N - 1) * H + (size_t)lo_d * H + (size_t)(lo_d - lo_p) * H + (size_t)(NI + 1) * H;
  const size_t f = ((size_t)(N - lo_d) * H + 1) & ~(size_t)1;
  const size_t v = d * 8 + f * 4;
  size_t b = v + (size_t)(4 * N + L + T) + (size_t)faces * H;
  // the byte tables are staged as KT strides of 64 dwords (KT from the instantiation's NM, see launch_cfr_wave) and the
  // match table as one 64-byte store behind them: the image must hold whichever reaches further (1 die x 5 faces: the
  // 512-byte table store ends 40 bytes behind the slack of the layout itself)
  const int NM = H == 4 ? 45 : (H == 5 ? 66 : 91);
  const size_t KT = (size_t)(((5 * NM + faces * H + 3) / 4 + 63) / 64);
  b = std::max(b, v + KT * 256);
  b = std::max(b, v + (size_t)(4 * N + L + T) + 64);
  // the sigma staging stores run up to a third of the stride count past the lane's own elements (kernel: "stage"): they must
  // stay inside the image whatever follows sigma in it
  const int EHM = H == 4 ? 176 : (H == 5 ? 325 : (H == 6 ? 540 : 810));
  const size_t KS = (size_t)((EHM + 63) / 64);
  b = std::max(b, ((size_t)(N - 1) * H + ((KS + 2) / 3) * 64) * 8);
  return ((b + 15) & ~(size_t)15) + kWaveLdsSlack;
}

// the instantiations cover depth-2 subgames of the game (max E*H, L*H, N over the shapes)
bool cfr_wave_supported(int H, int A, int dice, int faces, int max_EH, int max_LH, int max_N) {
  if (H == 6 && A == 13 && dice == 1 && faces == 6) return max_EH <= 540 && max_LH <= 396 && max_N <= 91;
  if (H == 4 && A == 9 && dice == 1 && faces == 4) return max_EH <= 176 && max_LH <= 112 && max_N <= 45;
  if (H == 5 && A == 11 && dice == 1 && faces == 5) return max_EH <= 325 && max_LH <= 225 && max_N <= 66;
  if (H == 9 && A == 13 && dice == 2 && faces == 3) return max_EH <= 810 && max_LH <= 594 && max_N <= 91;
  return false;
}

bool launch_cfr_wave(const CfrArgs& a, int B, size_t lds_bytes, hipStream_t stream) {
#define RBL_WAVE(H_, A_, D_, F_, EH_, LH_, N_)                                                                       \
  do {                                                                                                               \
    RBL_LAUNCH_TIMED((cfr_wave_kernel<H_, A_, D_, F_, EH_, LH_, N_>), dim3(B), dim3(64), lds_bytes, stream, a);      \
    return true;                                                                                                     \
  } while (0)
  if (a.H == 6 && a.A == 13 && a.dice == 1) RBL_WAVE(6, 13, 1, 6, 540, 396, 91);
  if (a.H == 4 && a.A == 9 && a.dice == 1) RBL_WAVE(4, 9, 1, 4, 176, 112, 45);
  if (a.H == 5 && a.A == 11 && a.dice == 1) RBL_WAVE(5, 11, 1, 5, 325, 225, 66);
  if (a.H == 9 && a.A == 13 && a.dice == 2) RBL_WAVE(9, 13, 2, 3, 810, 594, 91);
#undef RBL_WAVE
  return false;
}

}  // namespace rbl
