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
//     regret update, regret matching, normalisation, sum_strategies, the query rows -- runs over (row, hand PAIR) items: a
//     thread owns 16 bytes of every row and walks the rows with a constant stride, so consecutive threads touch consecutive
//     addresses (coalesced global traffic, conflict-free LDS) and no item needs a division;
//   * only what is sequential over the HANDS stays one thread per row, on LDS rows: a leaf's reach sum, the two normalisation
//     sums of a query row, and a terminal's match histogram, which four threads share (one per match bin + the total);
//   * sigma lives in LDS for the whole step (93 KB at the root, staged with LDS-direct loads): read from global once, the
//     traverser's rows written back once; regrets and sum_strategies are touched once (read + write) by the pass that needs them;
//   * a pseudo-leaf keeps ONE double (the reach sum that scales the net's row); its value row float(net row x sum) is
//     recomputed where it is read -- the same expression, bit for bit.  Value rows exist for nodes with children and
//     terminals only (49 of 325 at the root);
//   * every pass is "gather, compute, scatter": all operands of a batch of rows are requested before the first is used,
//     unconditionally and clamped.  hipcc keeps LDS loads behind earlier LDS stores (the arrays of a dynamic LDS block may
//     alias) and does not batch loads across basic blocks, so a pass written "load, compute, store" per item is one LDS or
//     memory round trip per item -- with 16 waves per CU nothing hides that (first version of this kernel: 137 k cycles).
// The workgroup is sized to the largest tree of its launch segment (1024 / 512 / 256 / 128 threads: engine.h), LDS at the
// root is 147 KB (one lane per CU, 16 waves); 105 VGPRs, no scratch.  Measured: root lane 115 k -> 69 k cycles per step,
// 2 dice x 6 faces self-play 3.5 -> 4.8 M it/s (profiles/r03_bench_2d6f.txt).  Arithmetic is operation-for-operation what
// cfr_rows_kernel / cfr_kernels.hip do (same operands, same order, -ffp-contract=off): tests/test_cfr_parity.py and
// tests/test_selfplay_parity.py run the 2 dice x 6 faces cases against the oracle through it.
//
// Reference: CFR::step and its helpers, /root/reference/csrc/liars_dice/subgame_solving.cc:538-664; leaf queries :253-269,
// terminal payoffs :80-98, :765-789.
#include <mutex>
#include <stdexcept>
#include <string>

#include "cfr_kernels.h"
#include "launch_timing.h"

namespace rbl {

namespace {

constexpr double kEps = 1e-80;
constexpr int kFlatMaxThreads = 1024;
constexpr int kG = 6;   // rows per thread whose GLOBAL operands are requested together (6 x 56 rows cover a root tree)
constexpr int kU = 4;  // items whose loads are issued together: a pass is a chain of dependent LDS / memory round trips per
                       // item (table word -> operands -> store), and with 16 waves per CU only independent work inside a
                       // thread hides them (one item at a time: ~700 cycles per item, 137 k cycles per root lane)

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt: behind the regret pass and the write-back
// it made every wave wait for the acknowledgement of global STORES nobody in this launch reads back (regrets, strategy sums and
// sigma are written once, by the thread that owns the element).  What the passes exchange lives in LDS; the one barrier that
// does need the global loads to have landed -- the staging barrier, behind the LDS-direct sigma loads -- keeps its vmcnt(0).
// (Measured, round 5: no difference -- 63.5 k vs 63.1 k ticks per root lane-step; the acknowledgements were never the wait.)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// per-node word built at staging: what a pass needs to know about a node's PARENT without a second table round trip
__device__ __forceinline__ int pk_pr(int w) { return w & 255; }          // reach row of the parent
__device__ __forceinline__ int pk_pv(int w) { return (w >> 8) & 255; }   // value row of the parent
__device__ __forceinline__ int pk_pdp(int w) { return (w >> 16) & 1; }   // depth parity of the parent
__device__ __forceinline__ int pk_ir(int w) { return ((w >> 17) & 63) - 1; }  // reach row of the node itself, -1: none

template <int H, int A, int DICE, int FACES>
__global__ void __launch_bounds__(kFlatMaxThreads) cfr_flat_kernel(const CfrArgs a) {
  extern __shared__ __align__(16) double lds[];
  constexpr int Q = 2 + A + 2 * H, NB = 2 * DICE + 1;
  // everything the workgroup must know before its first vector load, in ONE scalar load (cfr_kernels.h: LaneRec, indexed by launch
  // slot; it was lane_order -> lane -> lane_shape -> shapes[] -> node_off: four dependent round trips -- 20 % of a root lane-step)
  typedef const LaneRecWords __attribute__((address_space(4)))* crec_p;
  const LaneRecWords rec = ((crec_p)a.lane_rec)[a.lane0 + blockIdx.x];
  if (rec[kRecFlags] & kRecSkip) return;  // root de-duplication: this root lane is served by the epoch's representative
  const int lane = rec[kRecLane];
  const int tid = threadIdx.x, NT = blockDim.x;
  const int N = rec[kRecN], E = N - 1, L = rec[kRecL], NI = rec[kRecNI], nlev = rec[kRecNlev];
  const int NV = N - L, T = rec[kRecT];
  const int root_player = rec[kRecRootPlayer], row_off = rec[kRecRowOff];
  // level offsets of a tree of depth <= 2: {0, 1, lo2, N} (lo2 == N when it has two levels)
  auto lev_off = [&](int d) { return d <= 0 ? 0 : (d == 1 ? 1 : (d == 2 ? rec[kRecLo2] : N)); };
  const int t = a.trav, opp = 1 - t;

  // ---- LDS layout (doubles): rho0, rho1, yrow [NI][H] | val [NV][H] | lsum [L] | qs [4][L] | sig [E][H] | tables
  double* rho0 = lds;
  double* rho1 = rho0 + NI * H;
  double* yrow = rho1 + NI * H;  // refined reciprocals of the regret-matching row sums
  double* val = yrow + NI * H;   // values of the nodes that are not pseudo-leaves (row = -1 - t_lrow[n])
  double* lsum = val + NV * H;   // pseudo-leaf: sum of the opponent's reach
  double* qs = lsum + ((L + 1) & ~1);  // (16-byte aligned) query rows: [sum, reciprocal] of the acting player's reach, then of the other's
  double* sig = qs + 4 * L;
  int* tb = reinterpret_cast<int*>(sig + E * H);
  int* t_parent = tb, *t_act = tb + N, *t_cb = tb + 2 * N, *t_ce = tb + 3 * N, *t_depth = tb + 4 * N;
  int* t_pack = tb + 5 * N, *t_lrow = tb + 6 * N, *t_leaf = tb + 7 * N;  // t_leaf [L]: node of net row k
  int* t_term = t_leaf + L;                                              // t_term [T]: the terminals, ascending
  // t_mask [FACES][2]: bit h set when hand h shows exactly 1 / exactly 2 of the face (8-byte aligned slot after the ints)
  const int TI = ((7 * N + L + T + 1) & ~1) + 4 * FACES;  // ints of the shape's table blob: everything up to here, in this layout
  unsigned long long* t_mask = reinterpret_cast<unsigned long long*>(tb + TI - 4 * FACES);
  int* t_qrec = tb + TI;  // [L]: node | parent's reach row << 9 | who acted there << 17 | player to move << 18 | last bid << 19

  const size_t lane_e = (size_t)lane * a.Emax * H;
  double* g_sig = a.sigma + lane_e;
  double* g_reg = a.regrets + lane_e;
  double* g_sum = a.sums + lane_e;
  const float* lvals = a.values + (size_t)row_off * H;
  const double* bel = a.beliefs + (size_t)lane * 2 * H;
  double* rmean = a.root_mean + (size_t)lane * 2 * H;
  const bool snap_now = rec[kRecActIter] == a.steps_after;  // (act_iter < 0: no snapshot; steps_after >= 1)

  long long* dbg = a.dbg ? a.dbg + (size_t)lane * 16 : nullptr;
  int dbg_k = 0;
#define RBL_STAMP()                                                    \
  do {                                                                 \
    if (dbg && threadIdx.x == 0) dbg[dbg_k] = (long long)clock64();    \
    ++dbg_k;                                                           \
  } while (0)
  RBL_STAMP();  // 0: start
  if (dbg && threadIdx.x == 0)  // where this workgroup runs (HW_ID | XCC_ID << 32): lets a probe line the workgroups of a CU up in time
    dbg[15] = (long long)(unsigned)__builtin_amdgcn_s_getreg(63492) | ((long long)(unsigned)__builtin_amdgcn_s_getreg(63508) << 32);
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
    // sigma AND the shape's table blob (tables, pseudo-leaf / value-row map, match masks: flat_tabs, already in this LDS layout):
    // global -> LDS without a stop in registers (global_load_lds_dwordx4: lane l of a wave lands at base + 16 l,
    // scripts/micro/global_load_lds.hip): a wave requests its 1 KB chunks back to back and the whole working set is in flight at
    // once, one round trip behind the record.  (Round 5: eight int tables through registers behind the shape record -- 8 loads per
    // thread, wait, 8 stores; sigma through registers, 12 pieces per thread at a time, was four memory round trips: 8 k cycles.)
    {
      typedef int i4 __attribute__((ext_vector_type(4)));
      const i4* gt = reinterpret_cast<const i4*>(a.flat_tabs + rec[kRecTabOff]);
      i4* lt = reinterpret_cast<i4*>(tb);
      const int TW = (TI + 3) / 4;  // 16-byte pieces (the blob is padded; the LDS image has 16 bytes of slack behind t_qrec)
      const int wave = tid >> 6, ln = tid & 63, nw = NT >> 6;
      const int full = EW / 64, tfull = TW / 64;  // 1 KB chunks
      for (int c = wave; c < full + tfull; c += nw) {
        if (c < full)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gs + c * 64 + ln),
                                           (__attribute__((address_space(3))) void*)(ls + c * 64), 16, 0, 0);
        else
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gt + (c - full) * 64 + ln),
                                           (__attribute__((address_space(3))) void*)(lt + (c - full) * 64), 16, 0, 0);
      }
      if (tid < EW - full * 64) ls[full * 64 + tid] = gs[full * 64 + tid];
      const int tt = NT - 1 - tid;  // (the tables' tail on the block's last threads: other waves than sigma's tail where there are several)
      if (tt < TW - tfull * 64) lt[tfull * 64 + tt] = gt[tfull * 64 + tt];
    }
    if (tid < H) {
      bel_t = bel[t * H + tid];
      rho0[tid] = t == 0 ? bel_t : bel[tid];
      rho1[tid] = t == 1 ? bel_t : bel[H + tid];
      rmean_t = rmean[t * H + tid];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the LDS-direct loads are not tied to a register the compiler could wait on
    __syncthreads();
  }
  RBL_STAMP();  // 1: staged

  // Geometry of the elementwise passes: a thread owns one PAIR of hands (16 bytes of every row) and walks the rows with a
  // stride of R = threads / 18; consecutive threads cover a row, then the next one: rows are contiguous, so global traffic is
  // coalesced 16-byte accesses and no item needs a division.  (One (row, hand) item per thread and iteration was ~45 VALU
  // instructions per element, most of them index arithmetic: the passes were instruction-issue-bound.)
  typedef double d2 __attribute__((ext_vector_type(2)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  constexpr int HP = H / 2;
  const int R = NT / HP, my_r = tid / HP, h2 = tid - my_r * HP;
  const bool in_grid = my_r < R;  // the last threads of the block (NT is not a multiple of 18) sit the pair passes out
  d2* rho0_2 = reinterpret_cast<d2*>(rho0);
  d2* rho1_2 = reinterpret_cast<d2*>(rho1);
  d2* yrow_2 = reinterpret_cast<d2*>(yrow);
  d2* val_2 = reinterpret_cast<d2*>(val);
  d2* sig_2 = reinterpret_cast<d2*>(sig);
  const f2* lvals_2 = reinterpret_cast<const f2*>(lvals);
  // value pair of node c as its parent reads it (lr = t_lrow[c], lv = the net's pair when c is a pseudo-leaf)
  auto child_val2 = [&](int lr, f2 lv) {
    if (lr >= 0) {  // query_value_net (:257-268)
      const double ls = lsum[lr];
      return d2{(double)(float)((double)lv[0] * ls), (double)(float)((double)lv[1] * ls)};
    }
    return val_2[(-1 - lr) * HP + h2];
  };

  // ---------------------------------------------------------------- reach of both players under sigma + leaf values, ONE pass.
  // In a tree of depth <= 2 (all this kernel takes: the host checks nlev <= 3) the opponent acts on at most one edge of the
  // path root -> leaf, so a leaf's opponent reach is the ROOT row times sigma of that edge (or the root row itself): the same
  // two operands and the same single rounding as "level-1 row = root row x sigma, leaf = copy" or "level-1 row = copy, leaf =
  // row x sigma".  Leaves therefore do not wait for the level-1 rows: those (needed by the query rows later), the
  // pseudo-leaves' sums and the terminals' values are all computed side by side, one barrier for the whole phase.
  {
    const bool opp_at_root = root_player == opp;  // the opponent acts at the root (and the traverser at depth 1), or vice versa
    const double* ro = opp == 0 ? rho0 : rho1;    // the opponent's root row
    // which edge's sigma multiplies the root row on the way to leaf n (-1: none -- a depth-1 leaf below a traverser's root)
    auto opp_edge = [&](int n) {
      if (t_depth[n] == 1) return opp_at_root ? n - 1 : -1;
      return opp_at_root ? t_parent[n] - 1 : n - 1;
    };
    if (nlev == 3 && in_grid) {  // rows of the depth-1 nodes with children
      const int n0 = 1, n1 = rec[kRecLo2];
      d2* rho_m2 = root_player == 0 ? rho0_2 : rho1_2;  // the root's mover: times sigma
      d2* rho_n2 = root_player == 0 ? rho1_2 : rho0_2;  // the other player: copied
      for (int nb = n0 + my_r; nb < n1; nb += kU * R) {
        d2 sg[kU];
        int dst[kU];
        const d2 rm = rho_m2[h2], rn = rho_n2[h2];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
          const int n = min(nb + u * R, n1 - 1);
          const int w = t_pack[n];
          dst[u] = (nb + u * R < n1 && pk_ir(w) >= 0) ? pk_ir(w) * HP + h2 : -1;
          sg[u] = sig_2[(n - 1) * HP + h2];
        }
#pragma unroll
        for (int u = 0; u < kU; ++u)
          if (dst[u] >= 0) {
            rho_m2[dst[u]] = rm * sg[u];
            rho_n2[dst[u]] = rn;
          }
      }
    }
    RBL_F1();  // rows
    // leaves: one thread per node, sequential over the hands (the sums below are order-sensitive).  Pseudo-leaves are taken
    // from the front of the block, terminals from its back: different waves, so the two paths run side by side instead of
    // one after the other in every wave that holds both kinds
    for (int k = tid; k < L; k += NT) {
      const int n = t_leaf[k];
      const int e = opp_edge(n);
      // (16-byte reads: a thread walking ITS 288-byte row collides with the threads whose rows are a multiple of 8 away -- 2 304 bytes
      // = 9 x 64 banks -- and the LDS pipe, not the 36 dependent additions, is what these row-per-thread loops wait for; a b128
      // read moves twice the bytes per conflicted pass.  Same products, same additions, ascending hands.)
      const d2* sg2 = reinterpret_cast<const d2*>(sig + max(e, 0) * H);
      const d2* ro2 = reinterpret_cast<const d2*>(ro);
      double s = 0.0;
      if (e >= 0) {
#pragma unroll 6
        for (int p = 0; p < HP; ++p) {
          const d2 r = ro2[p], g = sg2[p];
          s += r[0] * g[0];
          s += r[1] * g[1];
        }
      } else {
#pragma unroll 6
        for (int p = 0; p < HP; ++p) {
          const d2 r = ro2[p];
          s += r[0];
          s += r[1];
        }
      }
      lsum[k] = s;
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
        const int e = opp_edge(n);
        const double* sg = sig + max(e, 0) * H;
        const int bid = t_act[t_parent[n]];  // the bid that was called is the parent's last bid
        const int qty = 1 + bid / FACES, face = bid % FACES;
        // which hands show exactly 1 / exactly 2 matches of `face`: two 36-bit masks in registers (reading the byte table per
        // hand between the stores below made every iteration an LDS round trip: hipcc keeps LDS loads behind earlier stores)
        const unsigned long long m1 = t_mask[2 * face], m2 = t_mask[2 * face + 1];
        const unsigned long long mine_mask = role == 3 ? ~0ull : (role == 0 ? ~(m1 | m2) : (role == 1 ? m1 : m2));
        double acc = 0.0;
        const d2* sg2 = reinterpret_cast<const d2*>(sg);
        const d2* ro2 = reinterpret_cast<const d2*>(ro);
#pragma unroll 6
        for (int p = 0; p < HP; ++p) {  // (16-byte reads, see the pseudo-leaf loop)
          d2 r = ro2[p];
          if (e >= 0) r = r * sg2[p];
          acc += ((mine_mask >> (2 * p)) & 1) ? r[0] : 0.0;
          acc += ((mine_mask >> (2 * p + 1)) & 1) ? r[1] : 0.0;
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
    lds_barrier();
    RBL_F1();  // barrier
  }
  RBL_STAMP();  // 2: reach + leaf values
  RBL_STAMP();  // 3
  RBL_STAMP();  // 4

  // ---------------------------------------------------------------- bottom-up (update_regrets :542-574) fused with regret
  // matching (:619-634) and the regret discount (:639-650)
  double* rho_t = t == 0 ? rho0 : rho1;
  for (int lev = nlev - 2; lev >= 0; --lev) {
    const int n0 = lev_off(lev), n1 = lev_off(lev + 1);
    const int c_lo = n1, c_hi = lev_off(lev + 2);
    const bool mine = (root_player ^ (lev & 1)) == t;
    const bool deepest = lev == nlev - 2;  // the children are the last level
    d2* rho_t2 = reinterpret_cast<d2*>(rho_t);
    // node values: one item per (node, hand pair), sequential over the actions in ascending order.  A pseudo-leaf's value comes
    // from the net's rows in global memory: twelve children's are requested before the first is used (all 24 at once spilled
    // registers), unconditionally and from a row that exists (a load under a condition is a branch, and loads in different
    // basic blocks are not requested together)
    // Round 6 (fine stamps of root lanes, profiles/r06_cfr_flat_node_values.txt): this pass was 10-12 k cycles on the deepest level
    // and 7-10 k at the ROOT level of a 62-77 k cycle lane-step, both as chains of LDS round trips:
    //  * the root level is ONE node with up to 24 children: 18 threads walked them one dependent LDS read (value, and sigma when the
    //    traverser moves there) at a time while the other 1 006 waited at the barrier.  Now every (child, hand pair) product is formed
    //    by its own thread into the query-sum scratch (dead until the query phase), and the 18 threads only run the ordered sum over
    //    operands requested eight at a time: the same products, the same additions in the same order;
    //  * on the other levels the pseudo-leaves' reach sums (lsum) are requested with the table words, unconditionally, instead of
    //    inside the per-child branch.
    const int rc0 = t_cb[n0], rc1 = t_ce[n0];
    const bool root_fast = n1 - n0 == 1 && !deepest && rc1 > rc0 && (rc1 - rc0) * H <= 4 * L;
    if (root_fast) {
      d2* scr_2 = reinterpret_cast<d2*>(qs);
      if (in_grid)
        for (int c = rc0 + my_r; c < rc1; c += R) {
          const d2 v = val_2[(-1 - t_lrow[c]) * HP + h2];  // above the deepest level every child has a value row
          scr_2[(c - rc0) * HP + h2] = mine ? v * sig_2[(c - 1) * HP + h2] : v;
        }
      lds_barrier();
      if (tid < HP) {
        const int nc = rc1 - rc0;
        d2 x = {0.0, 0.0};
        for (int k0 = 0; k0 < nc; k0 += 8) {
          d2 pr[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) pr[u] = scr_2[min(k0 + u, nc - 1) * HP + tid];
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (k0 + u < nc) x += pr[u];
        }
        val_2[(-1 - t_lrow[n0]) * HP + tid] = x;
      }
    } else if (deepest && 2 * (n1 - n0) <= R) {
      //  * on the deepest level a node has up to 24 children whose values come from the net's rows in global memory: two batches of
      //    twelve were two memory round trips, one behind the other, on 432 of the 1 024 threads.  Now TWO thread rows serve a
      //    node: row A requests children 0..11 and row B children 12..23 at the same time (one round trip); A folds its twelve
      //    into the node's value row, and behind a barrier B continues the SAME ordered sum from A's partial.  The pseudo-leaves'
      //    reach sums are requested with the value pairs, sigma (when the traverser moves here) six at a time.
      constexpr int kC = 12;
      static_assert(A - 1 <= 2 * kC, "two thread rows of twelve children cover a node");
      const int nn = n1 - n0;
      const int half = my_r >= nn ? 1 : 0;
      const bool act = in_grid && my_r < 2 * nn;
      const int n = n0 + (act ? my_r - half * nn : 0);
      const int c0 = t_cb[n], c1 = t_ce[n], cb = c0 + half * kC;
      const bool work = act && cb < c1;
      int lr[kC];
      f2 lvf[kC];
      double ls[kC];
#pragma unroll
      for (int u = 0; u < kC; ++u) lr[u] = t_lrow[max(min(cb + u, c1 - 1), 0)];
#pragma unroll
      for (int u = 0; u < kC; ++u) lvf[u] = lvals_2[max(lr[u], 0) * HP + h2];
#pragma unroll
      for (int u = 0; u < kC; ++u) ls[u] = lsum[max(lr[u], 0)];
      // the pseudo-leaf values as the floats they are ((double)(float)(net row x reach sum), :257-268): 2 registers per child
      // instead of the 4 + 2 of (value pair, reach sum) while sigma is gathered
      f2 vf[kC];
#pragma unroll
      for (int u = 0; u < kC; ++u) vf[u] = f2{(float)((double)lvf[u][0] * ls[u]), (float)((double)lvf[u][1] * ls[u])};
      const int vrow = (-1 - t_lrow[n]) * HP + h2;
      auto fold = [&](d2 x) {
#pragma unroll
        for (int ub = 0; ub < kC; ub += 6) {
          d2 sg[6];
          if (mine) {
#pragma unroll
            for (int u = 0; u < 6; ++u) sg[u] = sig_2[max(min(cb + ub + u, c1 - 1) - 1, 0) * HP + h2];
          }
#pragma unroll
          for (int u = 0; u < 6; ++u)
            if (cb + ub + u < c1) {  // query_value_net (:257-268) for a pseudo-leaf (as child_val2), else the child's value row
              const int k = ub + u;
              const d2 v = lr[k] >= 0 ? d2{(double)vf[k][0], (double)vf[k][1]} : val_2[(-1 - lr[k]) * HP + h2];
              if (mine) x += v * sg[u];
              else x += v;
            }
        }
        return x;
      };
      if (work && half == 0) val_2[vrow] = fold(d2{0.0, 0.0});
      lds_barrier();
      if (work && half == 1) val_2[vrow] = fold(val_2[vrow]);
    } else if (in_grid) {  // the general form: any level, any workgroup size (twelve children at a time)
      for (int n = n0 + my_r; n < n1; n += R) {
        const int c0 = t_cb[n], c1 = t_ce[n];
        if (c0 == c1) continue;
        constexpr int kC = 12;
        d2 x = {0.0, 0.0};
        for (int cb = c0; cb < c1; cb += kC) {
          int lr[kC];
          f2 vf[kC];
#pragma unroll
          for (int u = 0; u < kC; ++u) lr[u] = t_lrow[min(cb + u, c1 - 1)];
          if (deepest) {  // pseudo-leaves only exist on the last level: above it no child needs the net's rows
            f2 lvf[kC];
            double ls[kC];
#pragma unroll
            for (int u = 0; u < kC; ++u) lvf[u] = lvals_2[max(lr[u], 0) * HP + h2];
#pragma unroll
            for (int u = 0; u < kC; ++u) ls[u] = lsum[max(lr[u], 0)];
#pragma unroll
            for (int u = 0; u < kC; ++u) vf[u] = f2{(float)((double)lvf[u][0] * ls[u]), (float)((double)lvf[u][1] * ls[u])};
          } else {
#pragma unroll
            for (int u = 0; u < kC; ++u) vf[u] = f2{0.f, 0.f};
          }
#pragma unroll
          for (int u = 0; u < kC; ++u)
            if (cb + u < c1) {
              const d2 v = lr[u] >= 0 ? d2{(double)vf[u][0], (double)vf[u][1]} : val_2[(-1 - lr[u]) * HP + h2];
              if (mine) x += v * sig_2[(cb + u - 1) * HP + h2];
              else x += v;
            }
        }
        val_2[(-1 - t_lrow[n]) * HP + h2] = x;
      }
    }
    RBL_F2();  // node values
    lds_barrier();
    if (!mine) continue;
    // regrets of the edges into the level below; sigma receives the clamped regrets.  Every global operand of the thread's rows
    // in one round trip (clamped, unconditional: straight-line code)
    if (in_grid) {
      d2* greg_2 = reinterpret_cast<d2*>(g_reg);
      // (round 6: the pseudo-leaf's reach sum, the table word and the parent's value row are gathered with the other operands -- they
      // were read per item behind the previous item's LDS store, three dependent round trips each)
      for (int cb = c_lo + my_r; cb < c_hi; cb += kG * R) {
        d2 q[kG], cv[kG], pv[kG];
        f2 lvf[kG];
        int lr[kG], pw[kG];
#pragma unroll
        for (int u = 0; u < kG; ++u) {
          const int c = min(cb + u * R, c_hi - 1);
          q[u] = greg_2[(c - 1) * HP + h2];
          lr[u] = t_lrow[c];
          lvf[u] = deepest ? lvals_2[max(lr[u], 0) * HP + h2] : f2{0.f, 0.f};
          pw[u] = t_pack[c];
        }
#pragma unroll
        for (int u = 0; u < kG; ++u) {
          const double ls = lsum[max(lr[u], 0)];
          const d2 row = val_2[max(-1 - lr[u], 0) * HP + h2];
          // child_val2: query_value_net (:257-268) for a pseudo-leaf, else the child's value row
          cv[u] = lr[u] >= 0 ? d2{(double)(float)((double)lvf[u][0] * ls), (double)(float)((double)lvf[u][1] * ls)} : row;
          pv[u] = val_2[pk_pv(pw[u]) * HP + h2];
        }
#pragma unroll
        for (int u = 0; u < kG; ++u)
          if (cb + u * R < c_hi) {
            const int e = (cb + u * R - 1) * HP + h2;
            d2 qq = q[u];
            qq += cv[u];
            qq -= pv[u];
            sig_2[e] = d2{qq[0] > kEps ? qq[0] : kEps, qq[1] > kEps ? qq[1] : kEps};
            greg_2[e] = d2{qq[0] * (qq[0] > 0 ? a.pos : a.neg), qq[1] * (qq[1] > 0 ? a.pos : a.neg)};
          }
      }
    }
    RBL_F2();  // regrets
    lds_barrier();
    // row sums, sequential over the actions; parked in the (dead) rho_t row.  m / s below is hipcc's f64 division sequence
    // with its denominator-only part (v_rcp_f64 + two Newton steps) done here once per (node, hand); v_div_scale / v_div_fixup
    // are the identity for these operands (1e-80 <= m <= s; scripts/micro/div_shared_rcp.hip checks 3e9 cases against `/`)
    if (in_grid) {
      for (int n = n0 + my_r; n < n1; n += R) {
        const int c0 = t_cb[n], c1 = t_ce[n];
        if (c0 == c1) continue;
        d2 s = {0.0, 0.0};
        for (int cb = c0; cb < c1; cb += 8) {
          d2 m8[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) m8[u] = sig_2[(min(cb + u, c1 - 1) - 1) * HP + h2];
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (cb + u < c1) s += m8[u];
        }
        d2 y;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          double yy = __builtin_amdgcn_rcp(s[k]);
          double er = __builtin_fma(-s[k], yy, 1.0);
          yy = __builtin_fma(yy, er, yy);
          er = __builtin_fma(-s[k], yy, 1.0);
          y[k] = __builtin_fma(yy, er, yy);
        }
        const int ir = pk_ir(t_pack[n]);
        rho_t2[ir * HP + h2] = s;
        yrow_2[ir * HP + h2] = y;
      }
    }
    RBL_F2();  // row sums
    lds_barrier();
    if (in_grid) {
      for (int cb = c_lo + my_r; cb < c_hi; cb += kU * R) {
        d2 s[kU], y[kU], m[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
          const int c = min(cb + u * R, c_hi - 1);
          const int pr = pk_pr(t_pack[c]);
          s[u] = rho_t2[pr * HP + h2];
          y[u] = yrow_2[pr * HP + h2];
          m[u] = sig_2[(c - 1) * HP + h2];
        }
#pragma unroll
        for (int u = 0; u < kU; ++u)
          if (cb + u * R < c_hi) {
            d2 o;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
              const double q0 = m[u][k] * y[u][k];
              const double rem = __builtin_fma(-s[u][k], q0, m[u][k]);
              o[k] = __builtin_fma(rem, y[u][k], q0);
            }
            sig_2[(cb + u * R - 1) * HP + h2] = o;
          }
      }
    }
    lds_barrier();
  }
  RBL_STAMP();  // 5: bottom-up

  // the write-back's strategy sums are requested HERE, two phases early (round 6): their memory round trip runs under the root
  // mean, the barrier and the new-reach pass instead of at the head of the write-back (where it was 3-4 k exposed cycles of a
  // root lane-step; a lane of up to 6 x R + 1 nodes needs this one batch only)
  d2 wb_x[kG];
  {
    const d2* gsum_2 = reinterpret_cast<const d2*>(g_sum);
#pragma unroll
    for (int u = 0; u < kG; ++u) wb_x[u] = gsum_2[(min(1 + my_r + u * R, N - 1) - 1) * HP + h2];  // clamped, unconditional
  }
  // ---------------------------------------------------------------- running mean of the root values (:579-590)
  if (tid < H) {
    double m = rmean_t;
    m += (val[tid] - m) * a.alpha;  // the root is row 0 of val (it has children)
    rmean[t * H + tid] = m;
    rho_t[tid] = bel_t;  // root row of the traverser (it served as scratch above)
  }
  lds_barrier();
  // ---------------------------------------------------------------- traverser's reach under the NEW sigma (:636-638)
  for (int lev = 1; lev < nlev - 1; ++lev) {
    const int n0 = lev_off(lev), n1 = lev_off(lev + 1);
    const bool own = (root_player ^ ((lev - 1) & 1)) == t;
    d2* rho_t2 = reinterpret_cast<d2*>(rho_t);
    if (in_grid) {
      for (int nb = n0 + my_r; nb < n1; nb += kU * R) {
        d2 r[kU], sg[kU];
        int dst[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
          const int n = min(nb + u * R, n1 - 1);
          const int w = t_pack[n];
          dst[u] = (nb + u * R < n1 && pk_ir(w) >= 0) ? pk_ir(w) * HP + h2 : -1;
          r[u] = rho_t2[pk_pr(w) * HP + h2];
          sg[u] = sig_2[(n - 1) * HP + h2];
        }
#pragma unroll
        for (int u = 0; u < kU; ++u)
          if (dst[u] >= 0) rho_t2[dst[u]] = own ? r[u] * sg[u] : r[u];
      }
    }
    lds_barrier();
  }
  RBL_STAMP();  // 6: new reach

  // ---------------------------------------------------------------- sum_strategies (:651-657) + write back what changed
  {
    d2* snap_2 = reinterpret_cast<d2*>(a.snapshot + lane_e);
    // (root de-duplication: the representative also keeps sigma after EVERY iteration for the root lanes it serves)
    const bool rep = (rec[kRecFlags] & kRecRep) != 0;
    d2* snap_all_2 = reinterpret_cast<d2*>(a.snap_all + (size_t)(rep ? a.steps_after : 0) * a.Emax * H);
    d2* gsum_2 = reinterpret_cast<d2*>(g_sum);
    d2* gsig_2 = reinterpret_cast<d2*>(g_sig);
    const d2* rho_t2 = reinterpret_cast<const d2*>(rho_t);
    if (in_grid) {
      // gather, compute, scatter (round 6): the table word, sigma and the traverser's reach of every item of the batch are requested
      // before the first is used -- they sat inside the per-item conditions, one dependent LDS round trip after the other (12 k
      // cycles of a root lane-step when the traverser owns the 300 depth-1 edges)
      for (int cb = 1 + my_r; cb < N; cb += kG * R) {
        d2 x[kG], sg[kG], rt[kG];
        int w[kG];
        const bool first = cb == 1 + my_r;  // the first batch was requested before the new-reach pass (wb_x)
#pragma unroll
        for (int u = 0; u < kG; ++u) {
          const int c = min(cb + u * R, N - 1);
          if (first) x[u] = wb_x[u];
          else x[u] = gsum_2[(c - 1) * HP + h2];  // clamped, unconditional
          w[u] = t_pack[c];
          sg[u] = sig_2[(c - 1) * HP + h2];
        }
#pragma unroll
        for (int u = 0; u < kG; ++u) rt[u] = rho_t2[pk_pr(w[u]) * HP + h2];
#pragma unroll
        for (int u = 0; u < kG; ++u)
          if (cb + u * R < N) {
            const int e = (cb + u * R - 1) * HP + h2;
            if ((root_player ^ pk_pdp(w[u])) == t) {
              d2 xx = x[u];
              xx *= a.strat;
              xx += rt[u] * sg[u];
              gsum_2[e] = xx;
              gsig_2[e] = sg[u];
            }
            if (snap_now) snap_2[e] = sg[u];
            if (rep) snap_all_2[e] = sg[u];
          }
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
      const d2 *rm2 = reinterpret_cast<const d2*>(rm), *rn2 = reinterpret_cast<const d2*>(rn), *sg2 = reinterpret_cast<const d2*>(sg);
#pragma unroll 6
      for (int p = 0; p < HP; ++p) {  // (16-byte reads, see the leaf sums of the reach phase)
        const d2 r = rm2[p], g = sg2[p];
        sm += r[0] * g[0] + kEps;
        sm += r[1] * g[1] + kEps;
      }
#pragma unroll 6
      for (int p = 0; p < HP; ++p) {
        const d2 r = rn2[p];
        sn += r[0] + kEps;
        sn += r[1] + kEps;
      }
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
    lds_barrier();
    // the rows themselves: one item per element, consecutive threads = consecutive floats of the exchange buffer.
    // (x + eps) / s is hipcc's f64 division sequence minus v_div_scale / v_div_fixup, the identity here
    // (1e-80 <= x + eps <= s <= H + 1)
    // Split layout (a.q_dyn; engine.hip "Split query layout", round 4 for this kernel): what changes between iterations -- the
    // traverser flag and the two reach vectors -- goes to contiguous 16-byte aligned rows [rows][q_dyn_stride] that the value
    // net reads with aligned 16-byte loads (at n_in = 99 the canonical rows cost the net 16 unaligned loads per thread and
    // group); player and last-bid one-hot were split off once per epoch behind the solver-init launch and are not touched here
    float* gq = a.q_dyn ? a.q_dyn + (size_t)row_off * a.q_dyn_stride : a.queries + (size_t)row_off * Q;
    const int qstride = a.q_dyn ? a.q_dyn_stride : Q, qreach = a.q_dyn ? 1 : 2 + A;
    if (a.q_dyn) {
      for (int k = tid; k < L; k += NT) gq[(size_t)k * qstride] = (float)a.next_trav;
    } else {
      // the head of a row: player to move, traverser, one-hot last bid
      for (int i = tid; i < L * (2 + A); i += NT) {
        const int k = i / (2 + A), j = i - k * (2 + A);
        const int rec = t_qrec[k];
        gq[(size_t)k * Q + j] = j == 0 ? (float)((rec >> 18) & 1) : (j == 1 ? (float)a.next_trav : (j - 2 == (rec >> 19) ? 1.0f : 0.0f));
      }
    }
    // the two reach vectors: a pair of hands of BOTH players per thread and row
    if (in_grid) {
      for (int kb = my_r; kb < L; kb += kU * R) {
        d2 r0[kU], r1[kU], sg[kU], q4[kU][2];
        int pm[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
          const int k = min(kb + u * R, L - 1);
          const int rec = t_qrec[k];
          const int n = rec & 511, pr = (rec >> 9) & 255;
          pm[u] = (rec >> 17) & 1;
          r0[u] = rho0_2[pr * HP + h2];
          r1[u] = rho1_2[pr * HP + h2];
          sg[u] = sig_2[(n - 1) * HP + h2];
          q4[u][0] = reinterpret_cast<const d2*>(qs)[2 * k];      // sum, reciprocal of the player who acted
          q4[u][1] = reinterpret_cast<const d2*>(qs)[2 * k + 1];  // ... of the other one
        }
#pragma unroll
        for (int u = 0; u < kU; ++u)
          if (kb + u * R < L) {
            float* row = gq + (size_t)(kb + u * R) * qstride + qreach + 2 * h2;
#pragma unroll
            for (int wh = 0; wh < 2; ++wh) {
              const bool acted = wh == pm[u];
              const d2 r = wh == 0 ? r0[u] : r1[u];
              const d2 sy = acted ? q4[u][0] : q4[u][1];
#pragma unroll
              for (int k = 0; k < 2; ++k) {
                const double num = (acted ? r[k] * sg[u][k] : r[k]) + kEps;
                const double q0 = num * sy[1];
                const double rem = __builtin_fma(-sy[0], q0, num);
                row[wh * H + k] = (float)__builtin_fma(rem, sy[1], q0);
              }
            }
          }
      }
    }
  }
  RBL_STAMP();  // 8: queries
#undef RBL_STAMP
}

}  // namespace

size_t cfr_flat_lds_bytes(int N, int NI, int H, int L, int T, int faces) {
  const size_t d = (size_t)3 * NI * H + (size_t)(N - L) * H + (size_t)((L + 1) & ~1) + (size_t)4 * L + (size_t)(N - 1) * H;  // doubles
  // ints: the table blob (7 N + L + T, padded to 8 bytes, + the match masks) staged in 16-byte pieces, then t_qrec [L]; 16 bytes of
  // slack for the last piece
  const size_t b = d * 8 + ((((size_t)7 * N + L + T + 1) & ~(size_t)1) + (size_t)4 * faces + (size_t)L) * 4 + 16;
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
  RBL_LAUNCH_TIMED(kern, dim3(B), dim3(threads), lds_bytes, stream, a);
  return true;
}

}  // namespace rbl
