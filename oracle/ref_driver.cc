// oracle/ref_driver.cc -- TEST INFRASTRUCTURE ONLY.
//
// Puts the UNMODIFIED reference implementation (/root/reference/csrc/liars_dice) behind oracle/orc_api.h so the
// parity tests and the golden-vector generator can drive it from Python.  No reference code is copied: this file
// #includes the reference's subgame_solving.cc as its own translation unit (it is therefore not compiled a second
// time, see oracle/Makefile) with `private` opened, purely so that CFR::regrets / CFR::sum_strategies -- which
// ISubgameSolver does not expose -- can be read back.  Everything else is linked from the reference's objects.
//
// Built only where /root/reference exists; the resulting oracle/_ref/libref_driver.so travels to the GPU box.

#include <array>
#include <cstring>
#include <deque>
#include <functional>
#include <iostream>
#include <memory>
#include <numeric>
#include <optional>
#include <random>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include <torch/script.h>
#include <torch/torch.h>

// Every std/torch header the reference TU needs is already included above (their include guards make the
// re-inclusion below a no-op), so the macro only touches the reference's own declarations.
#define private public
#include "subgame_solving.cc"  // from -I/root/reference/csrc/liars_dice
#undef private

#include "real_net.h"
#include "recursive_solving.h"

#include "orc_api.h"

using namespace liars_dice;

namespace {

PartialPublicState make_state(int last_bid, int player) {
  PartialPublicState s;
  s.last_bid = last_bid;
  s.player_id = player;
  return s;
}

SubgameSolvingParams to_params(const orc_params* p) {
  SubgameSolvingParams q;
  q.num_iters = p->num_iters;
  q.max_depth = p->max_depth;
  q.linear_update = p->linear_update != 0;
  q.use_cfr = p->use_cfr != 0;
  q.optimistic = p->optimistic != 0;
  q.dcfr = p->dcfr != 0;
  q.dcfr_alpha = p->dcfr_alpha;
  q.dcfr_beta = p->dcfr_beta;
  q.dcfr_gamma = p->dcfr_gamma;
  return q;
}

// IValueNet adaptor: routes compute_values to {zeros, a C callback, the synthetic net, a TorchScript file on CPU}
// and add_training_example to a C callback.
class DriverNet : public IValueNet {
 public:
  DriverNet(int mode, int num_actions, int num_hands, orc_net_fn fn, void* user, const char* path,
            orc_example_fn ex_fn, void* ex_user)
      : mode_(mode), A_(num_actions), H_(num_hands), fn_(fn), user_(user), ex_fn_(ex_fn), ex_user_(ex_user) {
    if (mode_ == ORC_NET_TORCHSCRIPT) inner_ = create_torchscript_net(path, "cpu");
  }

  torch::Tensor compute_values(const torch::Tensor queries) override {
    torch::NoGradGuard ng;
    auto q = queries.contiguous();
    const int64_t rows = q.size(0), qs = q.size(1);
    if (mode_ == ORC_NET_TORCHSCRIPT) return inner_->compute_values(q);
    auto out = torch::zeros({rows, (int64_t)H_});
    if (mode_ == ORC_NET_CALLBACK) {
      fn_(user_, q.data_ptr<float>(), rows, qs, out.data_ptr<float>(), H_);
    } else if (mode_ == ORC_NET_SYNTHETIC) {
      orc_synthetic_net(q.data_ptr<float>(), rows, qs, out.data_ptr<float>(), H_, A_);
    }
    return out;
  }

  void add_training_example(const torch::Tensor queries, const torch::Tensor values) override {
    if (!ex_fn_) return;
    auto q = queries.contiguous();
    auto v = values.contiguous();
    ex_fn_(ex_user_, q.data_ptr<float>(), q.numel(), v.data_ptr<float>(), v.numel());
  }

 private:
  int mode_, A_, H_;
  orc_net_fn fn_;
  void* user_;
  orc_example_fn ex_fn_;
  void* ex_user_;
  std::shared_ptr<IValueNet> inner_;
};

std::shared_ptr<IValueNet> make_net(const Game& game, int mode, orc_net_fn fn, void* user, const char* path,
                                    orc_example_fn ex_fn, void* ex_user) {
  if (mode == ORC_NET_NONE) return nullptr;
  return std::make_shared<DriverNet>(mode, game.num_actions(), game.num_hands(), fn, user, path, ex_fn, ex_user);
}

struct SolverHandle {
  Game game;
  std::unique_ptr<ISubgameSolver> solver;
  CFR* cfr = nullptr;  // non-null iff params.use_cfr (the concrete type build_solver returns, :791-800)
  FP* fp = nullptr;
  SolverHandle(int d, int f) : game(d, f) {}
};

void flatten(const TreeStrategy& s, int H, int A, double* out) {
  for (size_t n = 0; n < s.size(); ++n)
    for (int h = 0; h < H; ++h) {
      const auto& row = s[n][h];
      for (int a = 0; a < A; ++a) out[(n * H + h) * A + a] = a < (int)row.size() ? row[a] : 0.0;
    }
}

}  // namespace

extern "C" {

const char* orc_impl_name(void) { return "reference"; }

int orc_num_actions(int dice, int faces) { return Game(dice, faces).num_actions(); }
int orc_num_hands(int dice, int faces) { return Game(dice, faces).num_hands(); }
int orc_num_matches(int dice, int faces, int hand, int face) { return Game(dice, faces).num_matches(hand, face); }
void orc_unpack_action(int dice, int faces, int action, int* quantity, int* face) {
  auto u = Game(dice, faces).unpack_action(action);
  *quantity = u.quantity;
  *face = u.face;
}
void orc_bid_range(int dice, int faces, int last_bid, int* lo, int* hi) {
  auto r = Game(dice, faces).get_bid_range(make_state(last_bid, 0));
  *lo = r.first;
  *hi = r.second;
}

int orc_unroll_tree(int dice, int faces, int root_last_bid, int root_player, int max_depth, int32_t* out,
                    int cap_nodes) {
  Game game(dice, faces);
  auto tree = unroll_tree(game, make_state(root_last_bid, root_player), max_depth);
  const int n = (int)tree.size();
  for (int i = 0; i < n && i < cap_nodes; ++i) {
    out[i * 6 + 0] = tree[i].state.last_bid;
    out[i * 6 + 1] = tree[i].state.player_id;
    out[i * 6 + 2] = tree[i].children_begin;
    out[i * 6 + 3] = tree[i].children_end;
    out[i * 6 + 4] = tree[i].parent;
    out[i * 6 + 5] = tree[i].depth;
  }
  return n;
}

void orc_compute_win_probability(int dice, int faces, int bet, const double* beliefs, double* out) {
  Game game(dice, faces);
  std::vector<double> b(beliefs, beliefs + game.num_hands());
  auto v = compute_win_probability(game, bet, b);
  std::copy(v.begin(), v.end(), out);
}

void orc_get_query(int dice, int faces, int traverser, int last_bid, int player_id, const double* reach0,
                   const double* reach1, float* out) {
  Game game(dice, faces);
  std::vector<double> r0(reach0, reach0 + game.num_hands()), r1(reach1, reach1 + game.num_hands());
  auto q = get_query(game, traverser, make_state(last_bid, player_id), r0, r1);
  std::copy(q.begin(), q.end(), out);
}

void orc_normalize_probabilities_safe(const double* in, int n, double eps, double* out_d, float* out_f) {
  std::vector<double> v(in, in + n);
  if (out_d) normalize_probabilities_safe(v, eps, out_d);
  if (out_f) normalize_probabilities_safe(v, eps, out_f);
}

void* orc_solver_create(int dice, int faces, int root_last_bid, int root_player, const double* beliefs0,
                        const double* beliefs1, const orc_params* params, int net_mode, orc_net_fn net_fn,
                        void* net_user, const char* torchscript_path, orc_example_fn ex_fn, void* ex_user) {
  auto* h = new SolverHandle(dice, faces);
  const int H = h->game.num_hands();
  Pair<std::vector<double>> beliefs;
  beliefs[0].assign(beliefs0, beliefs0 + H);
  beliefs[1].assign(beliefs1, beliefs1 + H);
  auto net = make_net(h->game, net_mode, net_fn, net_user, torchscript_path, ex_fn, ex_user);
  auto p = to_params(params);
  h->solver = build_solver(h->game, make_state(root_last_bid, root_player), beliefs, p, net);
  if (p.use_cfr)
    h->cfr = static_cast<CFR*>(h->solver.get());
  else
    h->fp = static_cast<FP*>(h->solver.get());
  return h;
}

void orc_solver_destroy(void* s) { delete static_cast<SolverHandle*>(s); }
int orc_solver_tree_size(void* s) { return (int)static_cast<SolverHandle*>(s)->solver->get_tree().size(); }
void orc_solver_step(void* s, int traverser) { static_cast<SolverHandle*>(s)->solver->step(traverser); }
void orc_solver_multistep(void* s) { static_cast<SolverHandle*>(s)->solver->multistep(); }

void orc_solver_get(void* s, int which, double* out) {
  auto* h = static_cast<SolverHandle*>(s);
  const int H = h->game.num_hands(), A = h->game.num_actions();
  switch (which) {
    case ORC_GET_AVERAGE:
      flatten(h->solver->get_strategy(), H, A, out);
      break;
    case ORC_GET_LAST:
      flatten(h->solver->get_sampling_strategy(), H, A, out);
      break;
    case ORC_GET_REGRETS:
      if (!h->cfr) throw std::runtime_error("regrets: CFR only");
      flatten(h->cfr->regrets, H, A, out);
      break;
    case ORC_GET_SUM:
      flatten(h->cfr ? h->cfr->sum_strategies : h->fp->sum_strategies, H, A, out);
      break;
    default:
      throw std::runtime_error("orc_solver_get: bad selector");
  }
}

void orc_solver_hand_values(void* s, int player, double* out) {
  auto v = static_cast<SolverHandle*>(s)->solver->get_hand_values(player);
  std::copy(v.begin(), v.end(), out);
}

void orc_solver_update_value_network(void* s) { static_cast<SolverHandle*>(s)->solver->update_value_network(); }

void orc_rl_run(int dice, int faces, double random_action_prob, int sample_leaf, const orc_params* params, int seed,
                int num_games, int net_mode, orc_net_fn net_fn, void* net_user, const char* torchscript_path,
                orc_example_fn ex_fn, void* ex_user) {
  RecursiveSolvingParams rp;
  rp.num_dice = dice;
  rp.num_faces = faces;
  rp.random_action_prob = (float)random_action_prob;
  rp.sample_leaf = sample_leaf != 0;
  rp.subgame_params = to_params(params);
  Game game(dice, faces);
  auto net = make_net(game, net_mode, net_fn, net_user, torchscript_path, ex_fn, ex_user);
  RlRunner runner(rp, net, seed);
  for (int g = 0; g < num_games; ++g) runner.step();
}

void orc_strategy_recursive(int dice, int faces, const orc_params* params, int to_leaf, int net_mode, orc_net_fn net_fn,
                            void* net_user, const char* torchscript_path, double* out) {
  Game game(dice, faces);
  auto net = make_net(game, net_mode, net_fn, net_user, torchscript_path, nullptr, nullptr);
  const auto sp = to_params(params);
  const TreeStrategy s = to_leaf ? compute_strategy_recursive_to_leaf(game, sp, net)
                                 : compute_strategy_recursive(game, sp, net);
  const int H = game.num_hands(), A = game.num_actions();
  for (size_t n = 0; n < s.size(); ++n)
    for (int h = 0; h < H; ++h)
      for (int a = 0; a < A; ++a) out[(n * H + h) * A + a] = s[n].empty() ? 0.0 : s[n][h][a];
}

void orc_strategy_recursive_sampled(int dice, int faces, const orc_params* params, int seed, int root_only, int net_mode,
                                    orc_net_fn net_fn, void* net_user, const char* torchscript_path, double* out) {
  Game game(dice, faces);
  auto net = make_net(game, net_mode, net_fn, net_user, torchscript_path, nullptr, nullptr);
  const TreeStrategy s = compute_sampled_strategy_recursive_to_leaf(game, to_params(params), net, seed, root_only != 0);
  const int H = game.num_hands(), A = game.num_actions();
  for (size_t n = 0; n < s.size(); ++n)
    for (int h = 0; h < H; ++h)
      for (int a = 0; a < A; ++a) out[(n * H + h) * A + a] = s[n].empty() ? 0.0 : s[n][h][a];
}

void orc_compute_exploitability2(int dice, int faces, const double* strategy, double out[2]) {
  Game game(dice, faces);
  const auto tree = unroll_tree(game);
  const int H = game.num_hands(), A = game.num_actions();
  TreeStrategy s;
  init_nd((int)tree.size(), H, A, 0.0, &s);
  for (size_t n = 0; n < tree.size(); ++n)
    for (int h = 0; h < H; ++h)
      for (int a = 0; a < A; ++a) s[n][h][a] = strategy[(n * H + h) * A + a];
  auto e = compute_exploitability2(game, s);
  out[0] = e[0];
  out[1] = e[1];
}

void orc_compute_ev2(int dice, int faces, const double* strategy1, const double* strategy2, double out[2]) {
  Game game(dice, faces);
  const auto tree = unroll_tree(game);
  const int H = game.num_hands(), A = game.num_actions();
  TreeStrategy s[2];
  const double* src[2] = {strategy1, strategy2};
  for (int k = 0; k < 2; ++k) {
    init_nd((int)tree.size(), H, A, 0.0, &s[k]);
    for (size_t n = 0; n < tree.size(); ++n)
      for (int h = 0; h < H; ++h)
        for (int a = 0; a < A; ++a) s[k][n][h][a] = src[k][(n * H + h) * A + a];
  }
  auto e = compute_ev2(game, s[0], s[1]);
  out[0] = e[0];
  out[1] = e[1];
}

void orc_synthetic_net(const float* queries, int64_t rows, int64_t qsize, float* out, int64_t osize,
                       int num_actions) {
  const int A = num_actions;
  const int64_t H = osize;
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
void orc_immediate_regrets(int dice, int faces, const double* strategies, int n_strategies, double* out) {
  Game game(dice, faces);
  const auto tree = unroll_tree(game);
  const int H = game.num_hands(), A = game.num_actions();
  const size_t stride = tree.size() * (size_t)H * A;
  std::vector<TreeStrategy> list(n_strategies);
  for (int k = 0; k < n_strategies; ++k) {
    init_nd((int)tree.size(), H, A, 0.0, &list[k]);
    for (size_t n = 0; n < tree.size(); ++n)
      for (int h = 0; h < H; ++h)
        for (int a = 0; a < A; ++a) list[k][n][h][a] = strategies[(size_t)k * stride + (n * H + h) * A + a];
  }
  const auto r = compute_immediate_regrets(game, list);  // subgame_solving.cc:984-1050, unmodified
  for (size_t n = 0; n < tree.size(); ++n)
    for (int h = 0; h < H; ++h) out[n * H + h] = r[n][h];
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
