// rebel_amd/csrc/tables.h -- host side: Liar's Dice rules, the BFS public tree and its flattened device tables.
//
// Reference semantics (cited per item; nothing here is derived from the reference's code layout):
//   rules      /root/reference/csrc/liars_dice/liars_dice.h:46-155
//   BFS tree   /root/reference/csrc/liars_dice/tree.h:31-70   (node index = BFS order, children contiguous)
//
// Layout decision (DESIGN.md "HBM layout"): a subgame's shape depends only on (root_last_bid, max_depth); the mover
// of a node is root_player ^ (depth & 1).  An engine therefore holds at most A shapes, uploaded once; lanes refer to
// them by id.  Strategy-like arrays are edge-indexed: edge e = child_node - 1, value [e][hand].
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace rbl {

constexpr int kMaxLevels = 64;

struct Rules {  // liars_dice.h:51-58
  int dice, faces, A, H, liar, wild;
  Rules(int d, int f) : dice(d), faces(f) {
    if (d < 1 || f < 1) throw std::runtime_error("rules: dice and faces must be >= 1");
    A = 1 + 2 * d * f;
    H = 1;
    for (int i = 0; i < d; ++i) H *= f;
    liar = A - 1;
    wild = f - 1;
  }
  int query_size() const { return 2 + A + 2 * H; }  // subgame_solving.cc:100-102
  int matches(int hand, int face) const {           // liars_dice.h:83-91: dice showing `face` or the wild face
    int m = 0;
    for (int i = 0; i < dice; ++i) {
      const int v = hand % faces;
      m += (v == face || v == wild);
      hand /= faces;
    }
    return m;
  }
  void bid_range(int last_bid, int* lo, int* hi) const {  // liars_dice.h:110-115
    if (last_bid < 0) {
      *lo = 0;
      *hi = A - 1;  // no liar call before the first bid
    } else {
      *lo = last_bid + 1;
      *hi = A;
    }
  }
};

struct Node {  // tree.h:31-47
  int last_bid, player, cb, ce, parent, depth;
};

inline std::vector<Node> unroll_tree(const Rules& g, int root_bid, int root_player, int max_depth) {  // tree.h:51-70
  std::vector<Node> t;
  t.push_back(Node{root_bid, root_player, 0, 0, -1, 0});
  for (size_t i = 0; i < t.size(); ++i) {
    if (t[i].depth >= max_depth) continue;
    int lo, hi;
    g.bid_range(t[i].last_bid, &lo, &hi);
    t[i].cb = (int)t.size();
    t[i].ce = (int)t.size() + (hi - lo);
    for (int a = lo; a < hi; ++a) t.push_back(Node{a, 1 - t[i].player, 0, 0, (int)i, t[i].depth + 1});
  }
  return t;
}

// Per-shape header as the kernels see it.  Offsets are into the concatenated int tables below.
struct ShapeDev {
  int node_off;  // first node of this shape in parent/act/cb/ce/leaf_row
  int N, L, T;   // nodes, pseudo-leaves (net rows), terminals
  int NI;        // nodes that keep a reach row: the root and every node with children (irank >= 0)
  int leaf_off;  // into `leaves` (node ids, ascending = net row order, subgame_solving.cc:189-195)
  int term_off;  // into `terms`  (node ids, ascending, :198-202)
  int nlev;      // number of BFS levels present
  int lev_off[kMaxLevels + 1];  // node-id range of level d is [lev_off[d], lev_off[d+1])
};

struct ShapeTables {
  std::vector<ShapeDev> shapes;  // index = root_last_bid + 1
  std::vector<int> parent, act, cb, ce, depth, leaf_row, irank, leaves, terms;
  std::vector<int> vrow;  // node -> its rank among the nodes that are NOT pseudo-leaves (-1 for pseudo-leaves): cfr_flat_kernel
  // node -> reach row of its parent | value row of its parent << 8 | depth parity of its parent << 16 | (own reach row + 1)
  // << 17 (cfr_flat_kernel: one table word per item instead of a chain of look-ups); 0 in the low 17 bits for the root
  std::vector<int> pack;
  int max_N = 0, max_L = 0, max_T = 0;

  // has_net=false reproduces the reference's refusal to build a truncated tree without a value net
  // (subgame_solving.cc:177-186) at reset time, not here.
  static ShapeTables build(const Rules& g, int max_depth) {
    ShapeTables t;
    for (int rb = -1; rb < g.A - 1; ++rb) {
      auto tree = unroll_tree(g, rb, 0, max_depth);
      ShapeDev s{};
      s.node_off = (int)t.parent.size();
      s.N = (int)tree.size();
      s.leaf_off = (int)t.leaves.size();
      s.term_off = (int)t.terms.size();
      int lev = -1;
      for (int i = 0; i < s.N; ++i) {
        const Node& n = tree[i];
        if (n.depth != lev) {
          if (n.depth != lev + 1) throw std::runtime_error("tables: BFS order violated");
          lev = n.depth;
          if (lev >= kMaxLevels) throw std::runtime_error("tables: tree deeper than kMaxLevels");
          s.lev_off[lev] = i;
        }
        const bool term = n.last_bid == g.liar;
        int row = -1;
        if (n.cb == n.ce && !term) {
          row = s.L++;
          t.leaves.push_back(i);
        }
        if (term) {
          ++s.T;
          t.terms.push_back(i);
        }
        t.parent.push_back(n.parent);
        t.act.push_back(n.last_bid);
        t.cb.push_back(n.cb);
        t.ce.push_back(n.ce);
        t.depth.push_back(n.depth);
        t.leaf_row.push_back(row);
        t.vrow.push_back(row >= 0 ? -1 : i - s.L);
        t.irank.push_back((i == 0 || n.cb != n.ce) ? s.NI++ : -1);
        {
          const int p = std::max(n.parent, 0);  // BFS order: the parent's entries exist already
          const int pir = t.irank[s.node_off + p], pvr = t.vrow[s.node_off + p], pd = t.depth[s.node_off + p];
          t.pack.push_back((pir & 255) | ((pvr & 255) << 8) | ((pd & 1) << 16) | ((t.irank.back() + 1) << 17));
        }
      }
      s.nlev = lev + 1;
      for (int d = s.nlev; d <= kMaxLevels; ++d) s.lev_off[d] = s.N;
      t.shapes.push_back(s);
      t.max_N = std::max(t.max_N, s.N);
      t.max_L = std::max(t.max_L, s.L);
      t.max_T = std::max(t.max_T, s.T);
    }
    return t;
  }
};

}  // namespace rbl
