// oracle/cfr_oracle.cc -- TEST INFRASTRUCTURE ONLY.  "port" oracle: a CPU restatement of the reference's hot path.
//
// A from-scratch restatement (flat arrays, one struct per subgame) of the algorithm in
// /root/reference/csrc/liars_dice/{liars_dice.h,tree.h,util.h,subgame_solving.cc,recursive_solving.cc}; every
// function cites the reference lines it follows.  It exists so the HIP path can be checked where the reference
// itself is not available, and it is itself PINNED: tests/test_oracle_pin.py compares it bit-for-bit against the
// compiled reference (oracle/_ref/libref_driver.so) and tests/test_golden.py against the committed golden vectors
// generated from the reference (tests/golden/make_golden.py).
//
// Arithmetic contract: fp64 state; every reduction sequential in ascending index order;
// compile with -ffp-contract=off; float truncations where the reference has them.  The RNG draws go through the
// very same libstdc++ <random> objects the reference uses, in the same order (Appendix A.4).
//
// Never linked into the product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may load it.

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <random>
#include <stdexcept>
#include <vector>

#include "orc_api.h"

namespace {

constexpr double kEps = 1e-80;  // kReachSmoothingEps == kRegretSmoothingEps (subgame_solving.h:34-36)

// ------------------------------------------------------------------------------------------------ game rules
struct Rules {  // liars_dice.h:46-155
  int dice, faces, A, H, liar, wild;
  Rules(int d, int f) : dice(d), faces(f) {
    A = 1 + 2 * d * f;  // liars_dice.h:55
    H = 1;
    for (int i = 0; i < d; ++i) H *= f;  // liars_dice.h:56
    liar = A - 1;                        // :57
    wild = f - 1;                        // :58
  }
  int matches(int hand, int face) const {  // liars_dice.h:83-91
    int m = 0;
    for (int i = 0; i < dice; ++i) {
      const int d = hand % faces;
      m += (d == face || d == wild) ? 1 : 0;
      hand /= faces;
    }
    return m;
  }
  void bid_range(int last_bid, int* lo, int* hi) const {  // liars_dice.h:110-115
    if (last_bid < 0) {
      *lo = 0;
      *hi = A - 1;
    } else {
      *lo = last_bid + 1;
      *hi = A;
    }
  }
};

// ------------------------------------------------------------------------------------------------ public tree
struct Node {  // tree.h:31-47
  int last_bid, player, cb, ce, parent, depth;
};

std::vector<Node> unroll(const Rules& g, int root_bid, int root_player, int max_depth) {  // tree.h:51-70
  std::vector<Node> t;
  t.push_back(Node{root_bid, root_player, 0, 0, -1, 0});
  for (size_t i = 0; i < t.size() && t[i].depth < max_depth; ++i) {
    int lo, hi;
    g.bid_range(t[i].last_bid, &lo, &hi);
    t[i].cb = (int)t.size();
    t[i].ce = (int)t.size() + (hi - lo);
    for (int a = lo; a < hi; ++a) t.push_back(Node{a, 1 - t[i].player, 0, 0, (int)i, t[i].depth + 1});
  }
  return t;
}

// ------------------------------------------------------------------------------------------------ prob utils
template <class T>
void normalize_safe(const double* x, int n, double eps, T* out) {  // util.h:68-78
  double sum = 0;
  for (int i = 0; i < n; ++i) sum += x[i] + eps;
  for (int i = 0; i < n; ++i) out[i] = (T)((x[i] + eps) / sum);
}

double seq_sum(const double* x, int n) {  // util.h:87-90 (std::accumulate from 0.0)
  double s = 0;
  for (int i = 0; i < n; ++i) s += x[i];
  return s;
}

void win_probability(const Rules& g, int bet, const double* beliefs, double* out) {  // subgame_solving.cc:765-789
  const int qty = 1 + bet / g.faces, face = bet % g.faces;  // liars_dice.h:74-80
  std::vector<double> cnt(2 * g.dice + 1, 0.0);
  for (int h = 0; h < g.H; ++h) cnt[g.matches(h, face)] += beliefs[h];
  for (size_t i = cnt.size() - 1; i-- > 0;) cnt[i] += cnt[i + 1];
  for (int h = 0; h < g.H; ++h) {
    const int left = std::max(0, qty - g.matches(h, face));
    const float p = (float)cnt[left];  // fp32 truncation, subgame_solving.cc:785
    out[h] = p;
  }
}

void write_query(const Rules& g, int traverser, int last_bid, int player, const double* r0, const double* r1,
                 float* q) {  // subgame_solving.cc:104-123
  int w = 0;
  q[w++] = (float)player;
  q[w++] = (float)traverser;
  for (int a = 0; a < g.A; ++a) q[w++] = (a == last_bid) ? 1.0f : 0.0f;
  normalize_safe(r0, g.H, kEps, q + w);
  w += g.H;
  normalize_safe(r1, g.H, kEps, q + w);
}

void synthetic_net(const float* queries, int64_t rows, int64_t qsize, float* out, int64_t H, int A) {
  for (int64_t r = 0; r < rows; ++r) {
    const float* q = queries + r * qsize;
    for (int64_t h = 0; h < H; ++h) {
      const float a = 0.5f * q[2 + A + h];
      const float b = 0.25f * q[2 + A + H + h];
      const float c = 0.125f * (q[1] - q[0]);
      const float d = 0.0625f * q[2 + h % A];
      out[r * H + h] = ((a - b) + c) + d;
    }
  }
}

// ------------------------------------------------------------------------------------------------ net double
struct Net {
  int mode = ORC_NET_ZERO;
  orc_net_fn fn = nullptr;
  void* user = nullptr;
  orc_example_fn ex_fn = nullptr;
  void* ex_user = nullptr;
  void forward(const Rules& g, const float* q, int64_t rows, int64_t qs, float* out) const {
    if (mode == ORC_NET_CALLBACK) {
      std::memset(out, 0, sizeof(float) * rows * g.H);
      fn(user, q, rows, qs, out, g.H);
    } else if (mode == ORC_NET_SYNTHETIC) {
      synthetic_net(q, rows, qs, out, g.H, g.A);
    } else if (mode == ORC_NET_ZERO) {
      std::memset(out, 0, sizeof(float) * rows * g.H);
    } else {
      throw std::runtime_error("port oracle: net mode not supported (TorchScript lives in the reference build)");
    }
  }
};

// ------------------------------------------------------------------------------------------------ tree traverser
// Shared by CFR / BR / FP: reach sweep, leaf queries, terminal payoffs (PartialTreeTraverser, subgame_solving.cc:152-303).
struct Traverser {
  Rules g;
  std::vector<Node> tree;
  int N, H, A, Q;
  Net net;
  bool has_net;
  std::vector<int> leaves, terminals;  // pseudo-leaves (:189-195) and terminals (:198-202), ascending node order
  std::vector<double> reach[2];        // [N][H]
  std::vector<double> value;           // traverser_values [N][H]
  std::vector<float> qbuf, leafv;

  Traverser(const Rules& rules, std::vector<Node> t, const Net& n, bool has)
      : g(rules), tree(std::move(t)), net(n), has_net(has) {
    N = (int)tree.size();
    H = g.H;
    A = g.A;
    Q = 2 + A + 2 * H;  // subgame_solving.cc:100-102
    for (int i = 0; i < N; ++i) {
      const bool term = tree[i].last_bid == g.liar;
      if (tree[i].cb == tree[i].ce && !term) {
        if (!has_net)  // subgame_solving.cc:177-186
          throw std::runtime_error("non-final leaf without a value net: provide a net or increase max_depth");
        leaves.push_back(i);
      }
      if (term) terminals.push_back(i);
    }
    reach[0].assign((size_t)N * H, 0.0);
    reach[1].assign((size_t)N * H, 0.0);
    value.assign((size_t)N * H, 0.0);
    qbuf.assign(leaves.size() * (size_t)Q, 0.f);
    leafv.assign(leaves.size() * (size_t)H, 0.f);
  }

  // reach[node][h] = reach[parent][h] * sigma[parent][h][a] when the parent's mover is `player` (subgame_solving.cc:54-78)
  void sweep_reach(const std::vector<double>& sigma, const double* beliefs, int player, std::vector<double>& out) const {
    for (int n = 0; n < N; ++n) {
      double* dst = &out[(size_t)n * H];
      if (n == 0) {
        for (int h = 0; h < H; ++h) dst[h] = beliefs[h];
        continue;
      }
      const int par = tree[n].parent, a = tree[n].last_bid;
      const double* src = &out[(size_t)par * H];
      if (tree[par].player == player) {
        for (int h = 0; h < H; ++h) dst[h] = src[h] * sigma[((size_t)par * H + h) * A + a];
      } else {
        for (int h = 0; h < H; ++h) dst[h] = src[h];
      }
    }
  }

  void query_for(int node, int traverser, float* q) const {  // write_query, subgame_solving.cc:211-218
    write_query(g, traverser, tree[node].last_bid, tree[node].player, &reach[0][(size_t)node * H],
                &reach[1][(size_t)node * H], q);
  }

  // precompute_all_leaf_values (subgame_solving.cc:238-242) = query net + scatter + terminals
  void leaf_values(int traverser) {
    const int opp = 1 - traverser;
    if (!leaves.empty()) {  // query_value_net, :253-269 (net not called when there are no pseudo-leaves, :254)
      const int64_t L = (int64_t)leaves.size();
      std::vector<double> scale(L);
      for (int64_t r = 0; r < L; ++r) {
        query_for(leaves[r], traverser, &qbuf[r * Q]);
        scale[r] = seq_sum(&reach[opp][(size_t)leaves[r] * H], H);
      }
      net.forward(g, qbuf.data(), L, Q, leafv.data());
      // leaf_values(float) *= scalers(double): computed in double, stored back as float (:268)
      for (int64_t r = 0; r < L; ++r)
        for (int h = 0; h < H; ++h) leafv[r * H + h] = (float)((double)leafv[r * H + h] * scale[r]);
      for (int64_t r = 0; r < L; ++r)  // populate_leaf_values, :273-282
        for (int h = 0; h < H; ++h) value[(size_t)leaves[r] * H + h] = leafv[r * H + h];
    }
    std::vector<double> w(H);
    for (int z : terminals) {  // precompute_terminal_leaves_values, :285-293 + :80-98
      const int bid = tree[tree[z].parent].last_bid;
      const double* ro = &reach[opp][(size_t)z * H];
      win_probability(g, bid, ro, w.data());
      const double bsum = seq_sum(ro, H);
      const bool inverse = tree[z].player != traverser;
      for (int h = 0; h < H; ++h) {
        double v = w[h] * 2 - bsum;
        if (inverse) v *= -1.0;
        value[(size_t)z * H + h] = v;
      }
    }
  }

  void emit_example(int traverser, const double* vals) const {  // add_training_example, :220-226
    if (!net.ex_fn) return;
    std::vector<float> q(Q), v(H);
    query_for(0, traverser, q.data());
    for (int h = 0; h < H; ++h) v[h] = (float)vals[h];
    net.ex_fn(net.ex_user, q.data(), Q, v.data(), H);
  }
};

std::vector<double> uniform_strategy(const Traverser& t) {  // get_uniform_strategy, subgame_solving.cc:718-730
  std::vector<double> s((size_t)t.N * t.H * t.A, 0.0);
  for (int n = 0; n < t.N; ++n) {
    int lo, hi;
    t.g.bid_range(t.tree[n].last_bid, &lo, &hi);
    const int cnt = t.tree[n].ce - t.tree[n].cb;
    for (int h = 0; h < t.H; ++h)
      for (int a = lo; a < lo + cnt; ++a) s[((size_t)n * t.H + h) * t.A + a] = 1. / cnt;
  }
  return s;
}

// get_uniform_reach_weigted_strategy, subgame_solving.cc:125-149
std::vector<double> uniform_reach_weighted(const Traverser& t, const std::vector<double> beliefs[2]) {
  std::vector<double> s = uniform_strategy(t);
  std::vector<double> rbuf((size_t)t.N * t.H, 0.0);
  for (int p = 0; p < 2; ++p) {
    t.sweep_reach(s, beliefs[p].data(), p, rbuf);
    for (int n = 0; n < t.N; ++n) {
      if (t.tree[n].cb == t.tree[n].ce || t.tree[n].player != p) continue;
      int lo, hi;
      t.g.bid_range(t.tree[n].last_bid, &lo, &hi);
      for (int h = 0; h < t.H; ++h)
        for (int a = lo; a < hi; ++a) s[((size_t)n * t.H + h) * t.A + a] *= rbuf[(size_t)n * t.H + h];
    }
  }
  return s;
}

// ------------------------------------------------------------------------------------------------ solvers
struct Solver {
  virtual ~Solver() {}
  virtual void step(int traverser) = 0;
  virtual const std::vector<double>& average() const = 0;
  virtual const std::vector<double>& sampling() const = 0;  // == belief-propagation strategy for both solvers
  virtual const std::vector<double>& hand_values(int p) const = 0;
  virtual void update_value_network() = 0;
  virtual Traverser& trav() = 0;
  orc_params params;
  void multistep() {
    for (int i = 0; i < params.num_iters; ++i) step(i % 2);
  }
};

struct CfrSolver : Solver {  // CFR, subgame_solving.cc:508-715
  Traverser t;
  std::vector<double> beliefs[2];
  std::vector<double> avg, sum, last, regrets, rbuf;
  std::vector<double> root_mean[2];
  int num_steps[2] = {0, 0};

  CfrSolver(const Rules& g, std::vector<Node> tree, const Net& net, bool has_net, const double* b0, const double* b1,
            const orc_params& p)
      : t(g, std::move(tree), net, has_net) {
    params = p;
    beliefs[0].assign(b0, b0 + g.H);
    beliefs[1].assign(b1, b1 + g.H);
    avg = uniform_strategy(t);  // :518-523
    last = avg;
    sum = uniform_reach_weighted(t, beliefs);
    regrets.assign(avg.size(), 0.0);
    rbuf.assign((size_t)t.N * t.H, 0.0);
  }
  size_t at(int n, int h, int a) const { return ((size_t)n * t.H + h) * t.A + a; }

  void update_regrets(int tr) {  // :538-575
    t.sweep_reach(last, beliefs[0].data(), 0, t.reach[0]);
    t.sweep_reach(last, beliefs[1].data(), 1, t.reach[1]);
    t.leaf_values(tr);
    const int H = t.H;
    for (int n = t.N; n-- > 0;) {
      const Node& nd = t.tree[n];
      if (nd.cb == nd.ce) continue;
      double* v = &t.value[(size_t)n * H];
      for (int h = 0; h < H; ++h) v[h] = 0.0;
      int lo, hi;
      t.g.bid_range(nd.last_bid, &lo, &hi);
      if (nd.player == tr) {
        for (int c = nd.cb, a = lo; c < nd.ce; ++c, ++a) {
          const double* cv = &t.value[(size_t)c * H];
          for (int h = 0; h < H; ++h) {
            regrets[at(n, h, a)] += cv[h];
            v[h] += cv[h] * last[at(n, h, a)];
          }
        }
        for (int h = 0; h < H; ++h)
          for (int a = lo; a < lo + (nd.ce - nd.cb); ++a) regrets[at(n, h, a)] -= v[h];
      } else {
        for (int c = nd.cb; c < nd.ce; ++c) {
          const double* cv = &t.value[(size_t)c * H];
          for (int h = 0; h < H; ++h) v[h] += cv[h];
        }
      }
    }
  }

  void step(int tr) override {  // :577-664
    update_regrets(tr);
    const int H = t.H, A = t.A;
    {
      const double alpha = params.linear_update ? 2. / (num_steps[tr] + 2) : 1. / (num_steps[tr] + 1);
      root_mean[tr].resize(H);
      for (int h = 0; h < H; ++h) root_mean[tr][h] += (t.value[h] - root_mean[tr][h]) * alpha;
    }
    double pos = 1, neg = 1, strat = 1;
    {
      const double k = num_steps[tr] + 1;  // "+1": the uniform strategy counts (:596)
      if (params.linear_update) {
        pos = neg = strat = k / (k + 1);
      } else if (params.dcfr) {
        pos = params.dcfr_alpha >= 5 ? 1 : std::pow(k, params.dcfr_alpha) / (std::pow(k, params.dcfr_alpha) + 1.);
        neg = params.dcfr_beta <= -5 ? 0 : std::pow(k, params.dcfr_beta) / (std::pow(k, params.dcfr_beta) + 1.);
        strat = std::pow(k / (k + 1), params.dcfr_gamma);
      }
    }
    for (int n = 0; n < t.N; ++n) {  // regret matching, :619-634
      const Node& nd = t.tree[n];
      if (nd.cb == nd.ce || nd.player != tr) continue;
      int lo, hi;
      t.g.bid_range(nd.last_bid, &lo, &hi);
      for (int h = 0; h < H; ++h) {
        double* row = &last[at(n, h, 0)];
        for (int a = lo; a < hi; ++a) row[a] = std::max(regrets[at(n, h, a)], kEps);
        const double s = seq_sum(row, A);  // normalize_probabilities sums the whole row (util.h:24-34)
        for (int a = 0; a < A; ++a) row[a] = row[a] / s;
      }
    }
    t.sweep_reach(last, beliefs[tr].data(), tr, rbuf);  // :636-638
    for (int n = 0; n < t.N; ++n) {                     // :639-661
      const Node& nd = t.tree[n];
      if (nd.cb == nd.ce || nd.player != tr) continue;
      int lo, hi;
      t.g.bid_range(nd.last_bid, &lo, &hi);
      for (int h = 0; h < H; ++h) {
        for (int a = lo; a < hi; ++a) {
          double& r = regrets[at(n, h, a)];
          r *= r > 0 ? pos : neg;
        }
        for (int a = lo; a < hi; ++a) sum[at(n, h, a)] *= strat;
        for (int a = lo; a < hi; ++a) sum[at(n, h, a)] += rbuf[(size_t)n * H + h] * last[at(n, h, a)];
        const double s = seq_sum(&sum[at(n, h, 0)], A);
        for (int a = 0; a < A; ++a) avg[at(n, h, a)] = sum[at(n, h, a)] / s;
      }
    }
    ++num_steps[tr];
  }

  const std::vector<double>& average() const override { return avg; }
  const std::vector<double>& sampling() const override { return last; }  // :682-688
  const std::vector<double>& hand_values(int p) const override { return root_mean[p]; }
  void update_value_network() override {  // :672-676
    t.emit_example(0, root_mean[0].data());
    t.emit_example(1, root_mean[1].data());
  }
  Traverser& trav() override { return t; }
};

// Best response sweep (BRSolver::compute_br, subgame_solving.cc:316-358); returns root values, fills br (dense).
void best_response(Traverser& t, int tr, const std::vector<double>& strategy, const std::vector<double> beliefs[2],
                   std::vector<double>* br, std::vector<double>* root_values) {
  t.sweep_reach(strategy, beliefs[0].data(), 0, t.reach[0]);
  t.sweep_reach(strategy, beliefs[1].data(), 1, t.reach[1]);
  t.leaf_values(tr);
  const int H = t.H, A = t.A;
  std::vector<int> best(H);
  for (int n = t.N; n-- > 0;) {
    const Node& nd = t.tree[n];
    if (nd.cb == nd.ce) continue;
    double* v = &t.value[(size_t)n * H];
    for (int h = 0; h < H; ++h) v[h] = 0.0;
    int lo, hi;
    t.g.bid_range(nd.last_bid, &lo, &hi);
    if (nd.player == tr) {
      for (int c = nd.cb, a = lo; c < nd.ce; ++c, ++a) {
        const double* cv = &t.value[(size_t)c * H];
        for (int h = 0; h < H; ++h)
          if (c == nd.cb || cv[h] > v[h]) {
            v[h] = cv[h];
            best[h] = a;
          }
      }
      if (br)
        for (int h = 0; h < H; ++h) {
          double* row = &(*br)[((size_t)n * H + h) * A];
          for (int a = 0; a < A; ++a) row[a] = 0.;
          row[best[h]] = 1.0;
        }
    } else {
      for (int c = nd.cb; c < nd.ce; ++c) {
        const double* cv = &t.value[(size_t)c * H];
        for (int h = 0; h < H; ++h) v[h] += cv[h];
      }
    }
  }
  root_values->assign(t.value.begin(), t.value.begin() + H);
}

struct FpSolver : Solver {  // FP, subgame_solving.cc:364-506
  Traverser t;
  std::vector<double> beliefs[2];
  std::vector<double> avg, sum, last, br;
  std::vector<double> root_values[2], root_mean[2];
  int num_strategies = 0;

  FpSolver(const Rules& g, std::vector<Node> tree, const Net& net, bool has_net, const double* b0, const double* b1,
           const orc_params& p)
      : t(g, std::move(tree), net, has_net) {
    params = p;
    beliefs[0].assign(b0, b0 + g.H);
    beliefs[1].assign(b1, b1 + g.H);
    avg = uniform_strategy(t);
    last = avg;
    sum = uniform_reach_weighted(t, beliefs);
    br.assign(avg.size(), 0.0);
  }
  size_t at(int n, int h, int a) const { return ((size_t)n * t.H + h) * t.A + a; }

  void update_sum(int n, int tr, const std::vector<double>& tb) {  // update_sum_strat, :401-431
    const Node& nd = t.tree[n];
    if (nd.cb == nd.ce) return;
    const int H = t.H;
    int lo, hi;
    t.g.bid_range(nd.last_bid, &lo, &hi);
    if (nd.player == tr) {
      std::vector<double> nb(H);
      for (int c = nd.cb, a = lo; c < nd.ce; ++c, ++a) {
        for (int h = 0; h < H; ++h) {
          sum[at(n, h, a)] += tb[h] * br[at(n, h, a)];
          last[at(n, h, a)] = tb[h] * br[at(n, h, a)];
        }
        for (int h = 0; h < H; ++h) nb[h] = tb[h] * br[at(n, h, a)];
        update_sum(c, tr, nb);
      }
    } else {
      for (int c = nd.cb; c < nd.ce; ++c) update_sum(c, tr, tb);
    }
  }

  void step(int tr) override {  // :433-476
    best_response(t, tr, avg, beliefs, &br, &root_values[tr]);
    const int H = t.H, A = t.A;
    const int num_update = num_strategies / 2 + 1;
    {
      const double alpha = params.linear_update ? 2. / (num_update + 1) : 1. / (num_update);
      root_mean[tr].resize(H);
      for (int h = 0; h < H; ++h) root_mean[tr][h] += (root_values[tr][h] - root_mean[tr][h]) * alpha;
    }
    update_sum(0, tr, beliefs[tr]);
    for (int n = 0; n < t.N; ++n) {
      const Node& nd = t.tree[n];
      if (nd.cb == nd.ce || nd.player != tr) continue;
      for (int h = 0; h < H; ++h) {
        double* srow = &sum[at(n, h, 0)];
        if (params.linear_update)
          for (int a = 0; a < A; ++a) srow[a] *= static_cast<double>(num_update + 1) / (num_update + 2);
        if (params.optimistic) {  // util.h:50-60
          const double* lrow = &last[at(n, h, 0)];
          const double s = seq_sum(srow, A) + seq_sum(lrow, A);
          for (int a = 0; a < A; ++a) avg[at(n, h, a)] = (srow[a] + lrow[a]) / s;
        } else {
          const double s = seq_sum(srow, A);
          for (int a = 0; a < A; ++a) avg[at(n, h, a)] = srow[a] / s;
        }
      }
    }
    ++num_strategies;
  }

  const std::vector<double>& average() const override { return avg; }
  const std::vector<double>& sampling() const override { return avg; }  // subgame_solving.h:75-82 defaults
  const std::vector<double>& hand_values(int p) const override { return root_mean[p]; }
  void update_value_network() override {  // :484-487
    t.emit_example(0, root_mean[0].data());
    t.emit_example(1, root_mean[1].data());
  }
  Traverser& trav() override { return t; }
};

Solver* build(const Rules& g, int root_bid, int root_player, const double* b0, const double* b1, const orc_params& p,
              const Net& net, bool has_net) {  // build_solver, subgame_solving.cc:791-800
  auto tree = unroll(g, root_bid, root_player, p.max_depth);
  if (p.use_cfr) return new CfrSolver(g, std::move(tree), net, has_net, b0, b1, p);
  return new FpSolver(g, std::move(tree), net, has_net, b0, b1, p);
}

// ------------------------------------------------------------------------------------------------ self-play walk
struct Runner {  // RlRunner, recursive_solving.h:40-86
  Rules g;
  orc_params sp;
  float random_action_prob;
  bool sample_leaf;
  Net net;
  int state_bid = -1, state_player = 0;
  std::vector<double> beliefs[2];
  std::mt19937 gen;

  Runner(const Rules& rules, const orc_params& p, double rap, bool leaf, const Net& n, int seed)
      : g(rules), sp(p), random_action_prob((float)rap), sample_leaf(leaf), net(n), gen(seed) {}

  void play_one_game() {  // RlRunner::step, recursive_solving.cc:160-182
    state_bid = -1;
    state_player = 0;
    beliefs[0].assign(g.H, 1.0 / g.H);
    beliefs[1].assign(g.H, 1.0 / g.H);
    while (state_bid != g.liar) {
      Solver* s = build(g, state_bid, state_player, beliefs[0].data(), beliefs[1].data(), sp, net, true);
      const int act_iteration = std::uniform_int_distribution<>(0, sp.num_iters)(gen);  // inclusive, :168-169
      for (int it = 0; it < act_iteration; ++it) s->step(it % 2);
      if (sample_leaf)
        sample_to_leaf(s);
      else
        sample_single(s);
      for (int it = act_iteration; it < sp.num_iters; ++it) s->step(it % 2);
      s->update_value_network();
      delete s;
    }
  }

  void bayes(std::vector<double>& b, const std::vector<double>& sigma, int n, int action) const {
    for (int h = 0; h < g.H; ++h) b[h] *= sigma[((size_t)n * g.H + h) * g.A + action];
    normalize_safe(b.data(), g.H, kEps, b.data());  // normalize_beliefs_inplace, recursive_solving.cc:41-44
  }

  void sample_to_leaf(Solver* s) {  // recursive_solving.cc:192-246
    const Traverser& t = s->trav();
    const std::vector<double>& sigma = s->sampling();
    std::vector<std::pair<int, int>> path;
    {
      int n = 0;
      const int br_sampler = std::uniform_int_distribution<>(0, 1)(gen);
      std::vector<double> sb[2] = {beliefs[0], beliefs[1]};
      while (t.tree[n].cb != t.tree[n].ce) {
        const float eps = std::uniform_real_distribution<float>(0, 1)(gen);
        const int mover = t.tree[n].player;
        int lo, hi, action;
        g.bid_range(t.tree[n].last_bid, &lo, &hi);
        if (mover == br_sampler && eps < random_action_prob) {
          std::uniform_int_distribution<> dis(lo, hi - 1);
          action = dis(gen);
        } else {
          std::discrete_distribution<> hd(sb[mover].begin(), sb[mover].end());
          const int hand = hd(gen);
          const double* row = &sigma[((size_t)n * g.H + hand) * g.A];
          std::discrete_distribution<> ad(row, row + g.A);
          action = ad(gen);
        }
        bayes(sb[mover], sigma, n, action);
        path.emplace_back(n, action);
        n = t.tree[n].cb + action - lo;
      }
    }
    for (auto [n, action] : path) {  // second pass on the real beliefs, :235-245
      int lo, hi;
      g.bid_range(state_bid, &lo, &hi);
      bayes(beliefs[state_player], sigma, n, action);
      const int child = t.tree[n].cb + action - lo;
      state_bid = t.tree[child].last_bid;
      state_player = t.tree[child].player;
    }
  }

  void sample_single(Solver* s) {  // recursive_solving.cc:248-275
    const std::vector<double>& sigma = s->sampling();
    int action;
    const int br_sampler = std::uniform_int_distribution<>(0, 1)(gen);
    const float eps = std::uniform_real_distribution<float>(0, 1)(gen);
    int lo, hi;
    g.bid_range(state_bid, &lo, &hi);
    if (state_player == br_sampler && eps < random_action_prob) {
      std::uniform_int_distribution<> dis(lo, hi - 1);
      action = dis(gen);
    } else {
      auto& b = beliefs[state_player];
      std::discrete_distribution<> hd(b.begin(), b.end());
      const int hand = hd(gen);
      const double* row = &sigma[((size_t)0 * g.H + hand) * g.A];
      std::discrete_distribution<> ad(row, row + g.A);
      action = ad(gen);
    }
    bayes(beliefs[state_player], sigma, 0, action);
    state_bid = action;  // Game::act, liars_dice.h:121-129
    state_player = 1 - state_player;
  }
};

Net make_net(int mode, orc_net_fn fn, void* user, orc_example_fn ex_fn, void* ex_user) {
  Net n;
  n.mode = mode;
  n.fn = fn;
  n.user = user;
  n.ex_fn = ex_fn;
  n.ex_user = ex_user;
  return n;
}

}  // namespace

extern "C" {

const char* orc_impl_name(void) { return "port"; }

int orc_num_actions(int dice, int faces) { return Rules(dice, faces).A; }
int orc_num_hands(int dice, int faces) { return Rules(dice, faces).H; }
int orc_num_matches(int dice, int faces, int hand, int face) { return Rules(dice, faces).matches(hand, face); }
void orc_unpack_action(int dice, int faces, int action, int* quantity, int* face) {
  (void)dice;
  *quantity = 1 + action / faces;  // liars_dice.h:74-80
  *face = action % faces;
}
void orc_bid_range(int dice, int faces, int last_bid, int* lo, int* hi) { Rules(dice, faces).bid_range(last_bid, lo, hi); }

int orc_unroll_tree(int dice, int faces, int root_last_bid, int root_player, int max_depth, int32_t* out,
                    int cap_nodes) {
  Rules g(dice, faces);
  auto t = unroll(g, root_last_bid, root_player, max_depth);
  const int n = (int)t.size();
  for (int i = 0; i < n && i < cap_nodes; ++i) {
    out[i * 6 + 0] = t[i].last_bid;
    out[i * 6 + 1] = t[i].player;
    out[i * 6 + 2] = t[i].cb;
    out[i * 6 + 3] = t[i].ce;
    out[i * 6 + 4] = t[i].parent;
    out[i * 6 + 5] = t[i].depth;
  }
  return n;
}

void orc_compute_win_probability(int dice, int faces, int bet, const double* beliefs, double* out) {
  win_probability(Rules(dice, faces), bet, beliefs, out);
}

void orc_get_query(int dice, int faces, int traverser, int last_bid, int player_id, const double* reach0,
                   const double* reach1, float* out) {
  write_query(Rules(dice, faces), traverser, last_bid, player_id, reach0, reach1, out);
}

void orc_normalize_probabilities_safe(const double* in, int n, double eps, double* out_d, float* out_f) {
  if (out_d) normalize_safe(in, n, eps, out_d);
  if (out_f) normalize_safe(in, n, eps, out_f);
}

void* orc_solver_create(int dice, int faces, int root_last_bid, int root_player, const double* beliefs0,
                        const double* beliefs1, const orc_params* params, int net_mode, orc_net_fn net_fn,
                        void* net_user, const char* torchscript_path, orc_example_fn ex_fn, void* ex_user) {
  (void)torchscript_path;
  Rules g(dice, faces);
  Net net = make_net(net_mode, net_fn, net_user, ex_fn, ex_user);
  return build(g, root_last_bid, root_player, beliefs0, beliefs1, *params, net, net_mode != ORC_NET_NONE);
}

void orc_solver_destroy(void* s) { delete static_cast<Solver*>(s); }
int orc_solver_tree_size(void* s) { return static_cast<Solver*>(s)->trav().N; }
void orc_solver_step(void* s, int traverser) { static_cast<Solver*>(s)->step(traverser); }
void orc_solver_multistep(void* s) { static_cast<Solver*>(s)->multistep(); }

void orc_solver_get(void* s, int which, double* out) {
  Solver* sv = static_cast<Solver*>(s);
  const std::vector<double>* src = nullptr;
  if (which == ORC_GET_AVERAGE) src = &sv->average();
  if (which == ORC_GET_LAST) src = &sv->sampling();
  if (which == ORC_GET_REGRETS) {
    auto* c = dynamic_cast<CfrSolver*>(sv);
    if (!c) throw std::runtime_error("regrets: CFR only");
    src = &c->regrets;
  }
  if (which == ORC_GET_SUM) {
    if (auto* c = dynamic_cast<CfrSolver*>(sv))
      src = &c->sum;
    else
      src = &dynamic_cast<FpSolver*>(sv)->sum;
  }
  if (!src) throw std::runtime_error("orc_solver_get: bad selector");
  std::copy(src->begin(), src->end(), out);
}

void orc_solver_hand_values(void* s, int player, double* out) {
  const auto& v = static_cast<Solver*>(s)->hand_values(player);
  std::copy(v.begin(), v.end(), out);
}

void orc_solver_update_value_network(void* s) { static_cast<Solver*>(s)->update_value_network(); }

void orc_rl_run(int dice, int faces, double random_action_prob, int sample_leaf, const orc_params* params, int seed,
                int num_games, int net_mode, orc_net_fn net_fn, void* net_user, const char* torchscript_path,
                orc_example_fn ex_fn, void* ex_user) {
  (void)torchscript_path;
  Rules g(dice, faces);
  Runner r(g, *params, random_action_prob, sample_leaf != 0, make_net(net_mode, net_fn, net_user, ex_fn, ex_user), seed);
  for (int i = 0; i < num_games; ++i) r.play_one_game();
}

// compute_strategy_recursive / _to_leaf (recursive_solving.cc:47-134): depth-first like the reference (the order does not
// change the result: every subgame depends only on the beliefs handed down to it).
namespace {
struct Recursive {
  Rules g;
  orc_params sp;
  Net net;
  std::vector<Node> full;
  double* out;
  int H, A;
  Recursive(const Rules& r, const orc_params& p, const Net& n, double* o)
      : g(r), sp(p), net(n), full(unroll(r, -1, 0, 1000000)), out(o), H(r.H), A(r.A) {}
  void copy_row(int full_node, const std::vector<double>& dense, int partial_node) {
    std::copy(dense.begin() + (size_t)partial_node * H * A, dense.begin() + (size_t)(partial_node + 1) * H * A,
              out + (size_t)full_node * H * A);
  }
  void per_node(int node, const std::vector<double> b[2]) {  // :47-74
    const Node& nd = full[node];
    if (nd.last_bid == g.liar) return;
    Solver* s = build(g, nd.last_bid, nd.player, b[0].data(), b[1].data(), sp, net, true);
    s->multistep();
    copy_row(node, s->average(), 0);
    delete s;
    int lo, hi;
    g.bid_range(nd.last_bid, &lo, &hi);
    for (int c = nd.cb; c < nd.ce; ++c) {
      std::vector<double> nb[2] = {b[0], b[1]};
      const int action = c - nd.cb + lo;
      for (int h = 0; h < H; ++h) nb[nd.player][h] *= out[((size_t)node * H + h) * A + action];
      normalize_safe(nb[nd.player].data(), H, kEps, nb[nd.player].data());
      per_node(c, nb);
    }
  }
  // sampled variant (:301-327): one draw per subgame, in the order the depth-first recursion builds the solvers
  bool sampled = false, root_only = false;
  std::mt19937 gen;
  std::vector<double> iteration_weights;
  void to_leaf(int node, const std::vector<double> b[2]) {  // :76-134
    const Node& nd = full[node];
    if (nd.last_bid == g.liar) return;
    orc_params p = sp;
    if (sampled) {
      std::discrete_distribution<int> dist(iteration_weights.begin(), iteration_weights.end());
      p.num_iters = dist(gen);
      if (root_only && node != 0) p.max_depth = 100000;
    }
    Solver* s = build(g, nd.last_bid, nd.player, b[0].data(), b[1].data(), p, net, true);
    s->multistep();
    const std::vector<double> strat = sampled ? s->sampling() : s->average();
    const std::vector<Node> part = s->trav().tree;
    delete s;
    struct Item {
      int f, p;
      std::vector<double> r[2];
    };
    std::vector<Item> queue;
    queue.push_back(Item{node, 0, {b[0], b[1]}});
    for (size_t qi = 0; qi < queue.size(); ++qi) {
      Item it = queue[qi];
      copy_row(it.f, strat, it.p);
      const Node& fn = full[it.f];
      const Node& pn = part[it.p];
      int lo, hi;
      g.bid_range(fn.last_bid, &lo, &hi);
      for (int i = 0; i < pn.ce - pn.cb; ++i) {
        Item ch{fn.cb + i, pn.cb + i, {it.r[0], it.r[1]}};
        const int action = lo + i;
        for (int h = 0; h < H; ++h) ch.r[fn.player][h] *= strat[((size_t)it.p * H + h) * A + action];
        queue.push_back(ch);
      }
      if (pn.ce == pn.cb && fn.ce != fn.cb) {
        normalize_safe(it.r[0].data(), H, kEps, it.r[0].data());
        normalize_safe(it.r[1].data(), H, kEps, it.r[1].data());
        to_leaf(it.f, it.r);
      }
    }
  }
};
}  // namespace

void orc_strategy_recursive(int dice, int faces, const orc_params* params, int to_leaf, int net_mode, orc_net_fn net_fn,
                            void* net_user, const char* torchscript_path, double* out) {
  (void)torchscript_path;
  Rules g(dice, faces);
  Recursive r(g, *params, make_net(net_mode, net_fn, net_user, nullptr, nullptr), out);
  std::fill(out, out + r.full.size() * (size_t)g.H * g.A, 0.0);
  std::vector<double> b[2];
  b[0].assign(g.H, 1. / g.H);
  b[1].assign(g.H, 1. / g.H);
  if (to_leaf)
    r.to_leaf(0, b);
  else
    r.per_node(0, b);
}

void orc_strategy_recursive_sampled(int dice, int faces, const orc_params* params, int seed, int root_only, int net_mode,
                                    orc_net_fn net_fn, void* net_user, const char* torchscript_path, double* out) {
  (void)torchscript_path;
  Rules g(dice, faces);
  Recursive r(g, *params, make_net(net_mode, net_fn, net_user, nullptr, nullptr), out);
  std::fill(out, out + r.full.size() * (size_t)g.H * g.A, 0.0);
  r.sampled = true;
  r.root_only = root_only != 0;
  r.gen.seed(seed);
  for (int i = 0; i < params->num_iters; ++i) r.iteration_weights.push_back(i % 2 ? 0.0 : (i / 2. + 1));  // :306-310
  std::vector<double> b[2];
  b[0].assign(g.H, 1. / g.H);
  b[1].assign(g.H, 1. / g.H);
  r.to_leaf(0, b);
}

void orc_compute_exploitability2(int dice, int faces, const double* strategy, double out[2]) {
  // compute_exploitability2, subgame_solving.cc:802-816: two full-tree best-response sweeps against uniform beliefs.
  Rules g(dice, faces);
  Net none;
  Traverser t(g, unroll(g, -1, 0, 1000000), none, false);
  std::vector<double> s(strategy, strategy + (size_t)t.N * g.H * g.A);
  std::vector<double> beliefs[2];
  beliefs[0].assign(g.H, 1. / g.H);
  beliefs[1].assign(g.H, 1. / g.H);
  for (int p = 0; p < 2; ++p) {
    std::vector<double> rv;
    best_response(t, p, s, beliefs, nullptr, &rv);
    out[p] = seq_sum(rv.data(), g.H) / rv.size();
  }
}

void orc_compute_ev2(int dice, int faces, const double* strategy1, const double* strategy2, double out[2]) {
  // compute_ev / compute_ev2, subgame_solving.cc:931-982: player 0 follows the first strategy, the opponent's reach comes
  // from the second; terminal values as in the solvers; then the roles are swapped and the sign flipped.
  Rules g(dice, faces);
  Net none;
  Traverser t(g, unroll(g, -1, 0, 1000000), none, false);
  const int H = g.H, A = g.A;
  const double* s[2] = {strategy1, strategy2};
  for (int k = 0; k < 2; ++k) {
    std::vector<double> mine(s[k], s[k] + (size_t)t.N * H * A), other(s[1 - k], s[1 - k] + (size_t)t.N * H * A);
    std::vector<double> uniform(H, 1. / H);
    t.sweep_reach(other, uniform.data(), 1, t.reach[1]);  // :943-944, opponent = player 1
    t.leaf_values(0);                                      // terminal payoffs for player 0 (:949-954)
    for (int n = t.N; n-- > 0;) {
      const Node& nd = t.tree[n];
      if (nd.cb == nd.ce) continue;
      double* v = &t.value[(size_t)n * H];
      for (int h = 0; h < H; ++h) v[h] = 0.0;
      int lo, hi;
      g.bid_range(nd.last_bid, &lo, &hi);
      for (int c = nd.cb, a = lo; c < nd.ce; ++c, ++a) {
        const double* cv = &t.value[(size_t)c * H];
        if (nd.player == 0)
          for (int h = 0; h < H; ++h) v[h] += mine[((size_t)n * H + h) * A + a] * cv[h];
        else
          for (int h = 0; h < H; ++h) v[h] += cv[h];
      }
    }
    const double ev = seq_sum(&t.value[0], H) / H;
    out[k] = k == 0 ? ev : -ev;
  }
}

void orc_synthetic_net(const float* queries, int64_t rows, int64_t qsize, float* out, int64_t osize,
                       int num_actions) {
  synthetic_net(queries, rows, qsize, out, osize, num_actions);
}
// compute_immediate_regrets (subgame_solving.cc:984-1050) restated: strategies [K][N][H][A] back to back -> out [N][H]
void orc_immediate_regrets(int dice, int faces, const double* strategies, int n_strategies, double* out) {
  Rules g(dice, faces);
  Net none;
  Traverser t(g, unroll(g, -1, 0, 1000000), none, false);
  const int H = g.H, A = g.A, N = t.N;
  const size_t stride = (size_t)N * H * A;
  std::vector<double> regrets(stride, 0.0), uniform(H, 1. / H);
  for (int k = 0; k < n_strategies; ++k) {
    const std::vector<double> sg(strategies + (size_t)k * stride, strategies + (size_t)(k + 1) * stride);
    t.sweep_reach(sg, uniform.data(), 0, t.reach[0]);  // precompute_reaches(last_strategies, initial_beliefs, 0 / 1)
    t.sweep_reach(sg, uniform.data(), 1, t.reach[1]);
    for (int trav = 0; trav < 2; ++trav) {
      t.leaf_values(trav);
      for (int n = N; n-- > 0;) {
        const Node& nd = t.tree[n];
        if (nd.cb == nd.ce) continue;
        double* v = &t.value[(size_t)n * H];
        for (int h = 0; h < H; ++h) v[h] = 0.0;
        int lo, hi;
        g.bid_range(nd.last_bid, &lo, &hi);
        if (nd.player == trav) {
          for (int c = nd.cb, a = lo; c < nd.ce; ++c, ++a) {
            const double* cv = &t.value[(size_t)c * H];
            for (int h = 0; h < H; ++h) {
              regrets[((size_t)n * H + h) * A + a] += cv[h];
              v[h] += cv[h] * sg[((size_t)n * H + h) * A + a];
            }
          }
          for (int h = 0; h < H; ++h)
            for (int c = nd.cb, a = lo; c < nd.ce; ++c, ++a) regrets[((size_t)n * H + h) * A + a] -= v[h];
        } else {
          for (int c = nd.cb; c < nd.ce; ++c) {
            const double* cv = &t.value[(size_t)c * H];
            for (int h = 0; h < H; ++h) v[h] += cv[h];
          }
        }
      }
    }
  }
  for (int n = 0; n < N; ++n)
    for (int h = 0; h < H; ++h) {
      double best = 0.0;
      if (t.tree[n].cb != t.tree[n].ce) {
        const double* r = &regrets[((size_t)n * H + h) * A];
        best = *std::max_element(r, r + A) / n_strategies;
      }
      out[(size_t)n * H + h] = best;
    }
}

// The reference's random draws come from libstdc++ <random> driven by std::mt19937 (recursive_solving.cc:168-169, 198-215):
// per round uniform_int_distribution<int>(0, hi), uniform_real_distribution<float>(0, 1), discrete_distribution<int>(w).
// The device restatement (rebel_amd/csrc/selfplay_kernels.hip) is checked against this draw for draw.
void orc_rng_probe(int seed, int rounds, int hi, const double* w, int nw, double* out) {
  std::mt19937 gen(seed);
  for (int k = 0; k < rounds; ++k) {
    out[3 * k + 0] = (double)std::uniform_int_distribution<>(0, hi)(gen);
    out[3 * k + 1] = (double)std::uniform_real_distribution<float>(0, 1)(gen);
    std::discrete_distribution<int> d(w, w + nw);
    out[3 * k + 2] = (double)d(gen);
  }
}

}  // extern "C"
