// rebel_amd/csrc/selfplay_kernels.hip -- the recursive-solving tree walk on the device (gfx950).
//
// What runs here, per lane, is RlRunner::step of the reference (/root/reference/csrc/liars_dice/recursive_solving.cc:
// 160-182) cut at subgame boundaries:
//   sp_begin : [game over -> root state, uniform beliefs (:161-163)]; act_iteration ~ U{0..num_iters} (:168-169);
//              the engine's lane descriptors for the subgame (shape = last_bid + 1, root player, root beliefs)
//   sp_scan  : first net row of every lane (prefix sum of the shapes' pseudo-leaf counts), part boundaries
//   ... num_iters x (value-net forward, CFR step) on the engine's streams; sigma_last snapshotted at act_iteration ...
//   sp_end   : sample_state_to_leaf (:192-246) or sample_state_single (:248-275) from the snapshot, Bayes update +
//              normalize_beliefs_inplace (:41-44), and the subgame's two training examples (update_value_network,
//              subgame_solving.cc:672-676: root query per traverser, root value means as float)
//
// Trajectory parity ("node indices bit-exact on identical seeds") needs the reference's random draws, which come from
// libstdc++: std::mt19937 and its uniform_int / uniform_real<float> / discrete distributions.  Those are deterministic
// published algorithms (GCC 11 <bits/random.tcc>, <bits/uniform_int_dist.h>); they are restated below and checked
// draw for draw against the host library on this toolchain (tests/test_selfplay_parity.py::test_device_rng_*):
//   mt19937            : MT19937, lazy in-place twist every 624 draws
//   uniform_int(a, b)  : Lemire's nearly-divisionless method on 32-bit draws (_S_nd<uint64_t>)
//   canonical<float>   : one draw; float(u) / 2^32, clamped below 1 with nextafter
//   canonical<double>  : two draws; (double(u1) + double(u2) * 2^32) / 2^64
//   discrete(w)        : p = w / sum(w) (sequential sum), running partial sums, last forced to 1, first cp >= u;
//                        fewer than two weights -> 0 without a draw
// One thread per lane: the walk is a few dozen dependent scalar steps per epoch (against 1024 CFR iterations), it only
// has to stay off the host.  Compiled with -ffp-contract=off like everything on the fp64 parity path.
#include "selfplay_kernels.h"

namespace rbl {

namespace {

constexpr double kEps = 1e-80;  // kReachSmoothingEps (subgame_solving.h:34-36)
constexpr int kMaxHands = 64;   // local belief copy; the engine refuses larger games for the device walk
constexpr int kMaxPath = 64;

struct Rng {
  uint32_t* mt;  // [624][n], this lane's column
  int n, p;
  __device__ uint32_t next() {
    if (p >= 624) {
      twist();
      p = 0;
    }
    uint32_t y = mt[(size_t)p * n];
    ++p;
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
  }
  __device__ void twist() {
    const size_t s = (size_t)n;
    uint32_t first = mt[0], cur = first;
    for (int k = 0; k < 624; ++k) {
      const uint32_t nxt = k + 1 < 624 ? mt[(size_t)(k + 1) * s] : first;  // mt[0] was already replaced: use the new one
      const uint32_t y = (cur & 0x80000000u) | (nxt & 0x7fffffffu);
      const int m = k + 397 < 624 ? k + 397 : k + 397 - 624;
      const uint32_t v = mt[(size_t)m * s] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
      mt[(size_t)k * s] = v;
      if (k == 0) first = v;
      cur = nxt;
    }
  }
  // std::uniform_int_distribution<int>(a, b) on a 32-bit engine (GCC 11: _S_nd<uint64_t>)
  __device__ int uniform_int(int a, int b) {
    const uint32_t range = (uint32_t)b - (uint32_t)a + 1u;  // never 0 here: b - a < 2^32 - 1
    unsigned long long product = (unsigned long long)next() * range;
    uint32_t low = (uint32_t)product;
    if (low < range) {
      const uint32_t threshold = (0u - range) % range;
      while (low < threshold) {
        product = (unsigned long long)next() * range;
        low = (uint32_t)product;
      }
    }
    return a + (int)(product >> 32);
  }
  // std::generate_canonical<float, 24>: one draw
  __device__ float canonical_float() {
    const float r = (float)next() / 4294967296.0f;
    return r >= 1.0f ? 0.99999994f : r;  // nextafter(1.f, 0.f)
  }
  // std::generate_canonical<double, 53>: two draws
  __device__ double canonical_double() {
    double sum = (double)next();
    sum += (double)next() * 4294967296.0;
    const double r = sum / 18446744073709551616.0;
    return r >= 1.0 ? 0.99999999999999989 : r;
  }
  // std::discrete_distribution<int>(w, w + n)(gen)
  template <class W>
  __device__ int discrete(W w, int cnt) {
    if (cnt < 2) return 0;
    double sum = 0.0;
    for (int i = 0; i < cnt; ++i) sum += w(i);
    const double u = canonical_double();
    double acc = 0.0;
    for (int i = 0; i < cnt; ++i) {
      const double p = w(i) / sum;
      acc = i == 0 ? p : acc + p;
      const double cp = i == cnt - 1 ? 1.0 : acc;
      if (!(cp < u)) return i;
    }
    return cnt;
  }
};

__device__ void normalize_safe_inplace(double* x, int n) {  // util.h:68-78 with eps = kReachSmoothingEps
  double sum = 0;
  for (int i = 0; i < n; ++i) sum += x[i] + kEps;
  for (int i = 0; i < n; ++i) x[i] = (x[i] + kEps) / sum;
}

// beliefs[h] *= sigma[child edge][h]; normalize_beliefs_inplace (recursive_solving.cc:41-44)
__device__ void bayes(double* b, const double* sigma, int child, int H) {
  for (int h = 0; h < H; ++h) b[h] *= sigma[(size_t)(child - 1) * H + h];
  normalize_safe_inplace(b, H);
}

__global__ void __launch_bounds__(256) sp_begin_kernel(const SpArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  const int H = a.H;
  int bid = a.bid[i], pl = a.player[i];
  double* b = a.beliefs + (size_t)i * 2 * H;
  if (bid == a.liar) {  // previous game over: state_ = root, beliefs_ uniform (recursive_solving.cc:161-163)
    bid = -1;
    pl = 0;
    const double u = 1.0 / H;
    for (int k = 0; k < 2 * H; ++k) b[k] = u;
    a.bid[i] = bid;
    a.player[i] = pl;
  }
  Rng r{a.mt + i, a.n, a.mt_idx[i]};
  a.lane_act[i] = r.uniform_int(0, a.num_iters);  // inclusive (:168-169)
  a.mt_idx[i] = r.p;
  a.lane_shape[i] = bid + 1;
  a.lane_player[i] = pl;
  double* eb = a.eng_beliefs + (size_t)i * 2 * H;
  for (int k = 0; k < 2 * H; ++k) eb[k] = b[k];
}

// One workgroup: exclusive prefix sum of L(shape(lane)) over the lanes in order -> lane_row; part boundaries; byte sums.
__global__ void __launch_bounds__(1024) sp_scan_kernel(const SpArgs a) {
  __shared__ long long part[1024];
  __shared__ int rep_s, skipped_s;
  const int t = threadIdx.x, per = (a.n + 1023) / 1024;
  const int l0 = t * per, l1 = l0 + per < a.n ? l0 + per : a.n;
  // root de-duplication (selfplay_kernels.h): the lowest-indexed lane in the root state is the epoch's representative; the other
  // root lanes take no net rows (shape 0 <=> last bid -1 <=> the root state, whose beliefs are uniform by construction)
  if (t == 0) {
    rep_s = 0x7fffffff;
    skipped_s = 0;
  }
  __syncthreads();
  if (a.dedup) {
    for (int i = l0; i < l1; ++i)
      if (a.lane_shape[i] == 0) {
        atomicMin(&rep_s, i);
        break;
      }
  }
  __syncthreads();
  const int rep = a.dedup && rep_s < a.n ? rep_s : -1;
  auto skip_of = [&](int i) { return rep < 0 || a.lane_shape[i] != 0 ? 0 : (i == rep ? 2 : 1); };
  long long mine = 0;
  for (int i = l0; i < l1; ++i) mine += skip_of(i) == 1 ? 0 : a.shapes[a.lane_shape[i]].L;
  part[t] = mine;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {  // inclusive Hillis-Steele scan
    const long long v = t >= off ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  long long row = part[t] - mine;
  int my_skipped = 0;
  for (int i = l0; i < l1; ++i) {
    const int sk = skip_of(i);
    a.lane_row[i] = (int)row;
    if (a.lane_skip) a.lane_skip[i] = sk;
    my_skipped += sk == 1;
    for (int p = 0; p < a.n_parts; ++p)
      if (i == a.part_lane[p]) a.info->part_row[p] = row;
    row += sk == 1 ? 0 : a.shapes[a.lane_shape[i]].L;
  }
  if (my_skipped) atomicAdd(&skipped_s, my_skipped);
  if (t == 1023) {
    a.info->part_row[a.n_parts] = part[1023];
    a.info->rows = part[1023];
    a.info->root_rep = rep;
  }
  // algorithmic bytes of one CFR step per part and traverser: read sigma over E, RMW regrets and sums +
  // write sigma over E_t, write L queries, read L value rows
  __shared__ unsigned long long bytes[kSpMaxParts][2];
  if (t < kSpMaxParts * 2) bytes[t >> 1][t & 1] = 0;
  __syncthreads();
  for (int i = l0; i < l1; ++i) {
    if (skip_of(i) == 1) continue;  // no launch touches this lane
    const int s = a.lane_shape[i];
    const ShapeDev& sh = a.shapes[s];
    int p = 0;
    while (p + 1 < a.n_parts && i >= a.part_lane[p + 1]) ++p;
    for (int tr = 0; tr < 2; ++tr) {
      const int et = a.shape_epar[2 * s + (a.lane_player[i] == tr ? 0 : 1)];
      const unsigned long long v =
          8ull * a.H * ((unsigned long long)(sh.N - 1) + 5ull * et) + 4ull * sh.L * (unsigned long long)(a.Q + a.H);
      atomicAdd(&bytes[p][tr], v);
    }
  }
  __syncthreads();
  if (t < kSpMaxParts * 2) a.info->part_bytes[t >> 1][t & 1] = bytes[t >> 1][t & 1];
  if (t == 0) a.info->skipped = skipped_s;
}

// Lanes of each part in order of tree size, largest first, ties by lane index: a stable counting sort by shape id (a
// subgame's tree shrinks as its root's last bid grows, so the shape id IS the size rank).  One thread per shape; the lane
// count of a 2 dice x 6 faces engine is a few thousand, and this runs once per epoch.  Then the shape at the head of each
// of the kSpSegs equal launch segments of a part: the host sizes each launch's LDS request by it.
__global__ void __launch_bounds__(128) sp_order_kernel(const SpArgs a) {
  __shared__ int start[kSpMaxParts][128];
  // shapes 0 .. A-1 (root_last_bid + 1); one more key, A, for the root lanes that root de-duplication serves from the
  // representative: their workgroups exit at once, so they sort behind every real tree (launch_sp_order: A + 1 <= 128)
  const int s = threadIdx.x, n_shapes = a.A + (a.dedup ? 1 : 0);
  // the epoch's representative, recomputed here from the lane shapes (the lowest-indexed root lane, as in sp_scan)
  __shared__ int rep_s;
  if (s == 0) rep_s = 0x7fffffff;
  __syncthreads();
  if (a.dedup)
    for (int i = s; i < a.n; i += blockDim.x)
      if (a.lane_shape[i] == 0) {
        atomicMin(&rep_s, i);
        break;
      }
  __syncthreads();
  const int rep = rep_s;
  // (Written with the shape in a register and the served key as arithmetic on a register: the first version, `flag ? a.A :
  // a.lane_shape[i]`, was MISCOMPILED by hipcc 7.2 -- it turned the select into ONE load through a selected pointer (&kernarg.A or
  // &lane_shape[i]) and lost the condition, so every lane sorted under key A and the launch segments got the LDS request of
  // whatever lane came first: scripts/micro/scalar_load_after_store.hip rules the memory system out, the ISA shows the select.)
  const int served_key = n_shapes - 1, dedup = a.dedup;
  auto key = [&](int i) {
    const int sh = a.lane_shape[i];
    const bool served = dedup != 0 && sh == 0 && i != rep;
    return served ? served_key : sh;
  };
  for (int p = 0; p < a.n_parts; ++p) {
    const int l0 = a.part_lane[p], l1 = a.part_lane[p + 1];
    int cnt = 0;
    if (s < n_shapes)
      for (int i = l0; i < l1; ++i) cnt += key(i) == s;
    start[p][s] = cnt;
    __syncthreads();
    if (s == 0) {
      int run = l0;
      for (int k = 0; k < n_shapes; ++k) {
        const int c = start[p][k];
        start[p][k] = run;
        run += c;
      }
    }
    __syncthreads();
    if (s < n_shapes) {
      int w = start[p][s];
      for (int i = l0; i < l1; ++i)
        if (key(i) == s) a.lane_order[w++] = i;
    }
    __syncthreads();
    if (s < kSpSegs) {
      const int cnt_p = l1 - l0, ns = sp_segments(cnt_p), first = l0 + (int)((long long)cnt_p * (s < ns ? s : 0) / ns);
      a.info->seg_shape[p][s] = cnt_p > 0 ? a.lane_shape[a.lane_order[first < l1 ? first : l1 - 1]] : 0;
    }
    __syncthreads();
  }
}

// write_query_to (subgame_solving.cc:104-123) for the subgame root
__device__ void write_root_query(const SpArgs& a, int traverser, int last_bid, int player, const double* b0,
                                 const double* b1, float* q) {
  int w = 0;
  q[w++] = (float)player;
  q[w++] = (float)traverser;
  for (int act = 0; act < a.A; ++act) q[w++] = act == last_bid ? 1.0f : 0.0f;
  for (int pl = 0; pl < 2; ++pl) {
    const double* b = pl == 0 ? b0 : b1;
    double sum = 0;
    for (int h = 0; h < a.H; ++h) sum += b[h] + kEps;
    for (int h = 0; h < a.H; ++h) q[w++] = (float)((b[h] + kEps) / sum);
  }
}

__global__ void __launch_bounds__(128) sp_end_kernel(const SpArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  const int H = a.H, A = a.A, Q = a.Q;
  const int root_bid = a.lane_shape[i] - 1, root_player = a.lane_player[i];
  const ShapeDev& s = a.shapes[root_bid + 1];
  const int* t_act = a.act + s.node_off;
  const int* t_cb = a.cb + s.node_off;
  const int* t_ce = a.ce + s.node_off;
  const int* t_depth = a.depth + s.node_off;
  // root de-duplication: a served root lane reads the representative's sigma after ITS act_iteration and the representative's
  // root values -- bit for bit what its own solve would have left in its own slabs
  const bool served = a.dedup && a.lane_skip[i] == 1;
  const int src = served ? a.info->root_rep : i;
  const double* sigma = served ? a.snap_all + (size_t)a.lane_act[i] * a.Emax * H : a.snapshot + (size_t)i * a.Emax * H;
  double* bel = a.beliefs + (size_t)i * 2 * H;
  // ---- the subgame's training examples (root state of the epoch = the engine's descriptors)
  {
    const double* rb = a.eng_beliefs + (size_t)i * 2 * H;
    for (int t = 0; t < 2; ++t) {
      const size_t k = (size_t)2 * i + t;
      write_root_query(a, t, root_bid, root_player, rb, rb + H, a.ex_q + k * Q);
      for (int h = 0; h < H; ++h) a.ex_v[k * H + h] = (float)a.root_mean[((size_t)src * 2 + t) * H + h];  // :224
    }
  }
  Rng r{a.mt + i, a.n, a.mt_idx[i]};
  int bid = root_bid, pl = root_player;
  if (a.sample_leaf) {  // sample_state_to_leaf (recursive_solving.cc:192-246)
    int path[kMaxPath], np = 0;
    double sb[2 * kMaxHands];
    for (int k = 0; k < 2 * H; ++k) sb[k] = bel[k];
    int n = 0;
    const int br_sampler = r.uniform_int(0, 1);
    while (t_cb[n] != t_ce[n]) {
      const float eps = r.canonical_float();
      const int mover = root_player ^ (t_depth[n] & 1);
      const int c0 = t_cb[n], c1 = t_ce[n];
      const int lo = t_act[c0];
      int action;
      if (mover == br_sampler && eps < a.rap) {
        action = r.uniform_int(lo, lo + (c1 - c0) - 1);
      } else {
        const double* hb = sb + mover * H;
        const int hand = r.discrete([&](int h) { return hb[h]; }, H);
        // the reference samples over a dense [A] row that is zero outside the legal range [lo, lo + c1 - c0)
        action = r.discrete(
            [&](int act) {
              const int c = act - lo;
              return (c >= 0 && c < c1 - c0) ? sigma[(size_t)(c0 + c - 1) * H + hand] : 0.0;
            },
            A);
      }
      const int child = c0 + action - lo;
      bayes(sb + mover * H, sigma, child, H);
      if (np < kMaxPath) path[np++] = child;
      n = child;
    }
    for (int k = 0; k < np; ++k) {  // second pass on the lane's real beliefs (:235-245)
      bayes(bel + pl * H, sigma, path[k], H);
      bid = t_act[path[k]];
      pl = 1 - pl;
    }
  } else {  // sample_state_single (:248-275)
    const int br_sampler = r.uniform_int(0, 1);
    const float eps = r.canonical_float();
    const int c0 = t_cb[0], c1 = t_ce[0];
    const int lo = root_bid < 0 ? 0 : root_bid + 1, hi = root_bid < 0 ? A - 1 : A;  // liars_dice.h:110-115
    int action;
    if (pl == br_sampler && eps < a.rap) {
      action = r.uniform_int(lo, hi - 1);
    } else {
      const double* hb = bel + pl * H;
      const int hand = r.discrete([&](int h) { return hb[h]; }, H);
      action = r.discrete(
          [&](int act) {
            const int c = act - lo;
            return (c >= 0 && c < c1 - c0) ? sigma[(size_t)(c0 + c - 1) * H + hand] : 0.0;
          },
          A);
    }
    bayes(bel + pl * H, sigma, c0 + action - lo, H);
    bid = action;  // Game::act (liars_dice.h:121-129)
    pl = 1 - pl;
  }
  a.mt_idx[i] = r.p;
  a.bid[i] = bid;
  a.player[i] = pl;
  if (bid == a.liar) atomicAdd(&a.info->games, 1ull);
}

__global__ void sp_rng_probe_kernel(uint32_t* mt, int* mt_idx, int n, int lane, int rounds, int hi, const double* w,
                                    int nw, double* out) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  Rng r{mt + lane, n, mt_idx[lane]};
  for (int k = 0; k < rounds; ++k) {
    out[3 * k + 0] = (double)r.uniform_int(0, hi);
    out[3 * k + 1] = (double)r.canonical_float();
    out[3 * k + 2] = (double)r.discrete([&](int i) { return w[i]; }, nw);
  }
  mt_idx[lane] = r.p;
}

}  // namespace

void mt19937_seed_state(uint32_t seed, uint32_t* x) {
  x[0] = seed;
  for (uint32_t i = 1; i < 624; ++i) x[i] = 1812433253u * (x[i - 1] ^ (x[i - 1] >> 30)) + i;
}

void launch_sp_begin(const SpArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(sp_begin_kernel, dim3((a.n + 255) / 256), dim3(256), 0, st, a);
}
void launch_sp_order(const SpArgs& a, hipStream_t st) {
  if (a.A + 1 > 128) return;
  hipLaunchKernelGGL(sp_order_kernel, dim3(1), dim3(128), 0, st, a);
}

void launch_sp_scan(const SpArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(sp_scan_kernel, dim3(1), dim3(1024), 0, st, a);
}
void launch_sp_end(const SpArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(sp_end_kernel, dim3((a.n + 127) / 128), dim3(128), 0, st, a);
}
void launch_sp_rng_probe(uint32_t* mt, int* mt_idx, int n, int lane, int rounds, int hi, const double* w, int nw,
                         double* out, hipStream_t st) {
  hipLaunchKernelGGL(sp_rng_probe_kernel, dim3(1), dim3(64), 0, st, mt, mt_idx, n, lane, rounds, hi, w, nw, out);
}

}  // namespace rbl
