/* oracle/orc_api.h -- C API shared by the two CPU oracles.  TEST INFRASTRUCTURE ONLY.
 *
 *   oracle/_ref/libref_driver.so   : the UNMODIFIED reference sources (/root/reference/csrc/liars_dice) behind this API
 *   oracle/_build/liboracle_port.so: our C++ restatement of the same algorithm (oracle/cfr_oracle.cc) behind this API
 *
 * Both export exactly these symbols, so every parity test can be parametrised over {ref, port}; the port is
 * pinned against the ref (tests/test_oracle_pin.py) and against the committed golden vectors (tests/golden/).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load either library.
 *
 * Dense strategy layout is the reference's TreeStrategy: double[N][H][A] (subgame_solving.h:39).
 */
#pragma once
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Value-net double: fills out[rows][osize] from queries[rows][qsize] (IValueNet::compute_values, net_interface.h:28). */
typedef void (*orc_net_fn)(void* user, const float* queries, int64_t rows, int64_t qsize, float* out, int64_t osize);
/* Training-example sink (IValueNet::add_training_example, net_interface.h:31-32). */
typedef void (*orc_example_fn)(void* user, const float* query, int64_t qsize, const float* values, int64_t osize);

/* Mirrors SubgameSolvingParams (subgame_solving.h:43-58). */
typedef struct {
  int32_t num_iters, max_depth, linear_update, use_cfr, optimistic, dcfr;
  double dcfr_alpha, dcfr_beta, dcfr_gamma;
} orc_params;

enum { ORC_NET_ZERO = 0, ORC_NET_CALLBACK = 1, ORC_NET_SYNTHETIC = 2, ORC_NET_TORCHSCRIPT = 3, ORC_NET_NONE = 4 };
enum { ORC_GET_AVERAGE = 0, ORC_GET_LAST = 1, ORC_GET_REGRETS = 2, ORC_GET_SUM = 3 };

const char* orc_impl_name(void); /* "reference" or "port" */

/* ---- game rules (liars_dice.h:46-155) ---- */
int orc_num_actions(int dice, int faces);
int orc_num_hands(int dice, int faces);
int orc_num_matches(int dice, int faces, int hand, int face);
void orc_unpack_action(int dice, int faces, int action, int* quantity, int* face);
void orc_bid_range(int dice, int faces, int last_bid, int* lo, int* hi);

/* ---- tree (tree.h:51-70): 6 ints per node {last_bid, player_id, children_begin, children_end, parent, depth} ---- */
int orc_unroll_tree(int dice, int faces, int root_last_bid, int root_player, int max_depth, int32_t* out, int cap_nodes);

/* ---- pieces with known answers in the reference tests ---- */
void orc_compute_win_probability(int dice, int faces, int bet, const double* beliefs, double* out);
void orc_get_query(int dice, int faces, int traverser, int last_bid, int player_id, const double* reach0,
                   const double* reach1, float* out);
void orc_normalize_probabilities_safe(const double* in, int n, double eps, double* out_d, float* out_f);

/* ---- subgame solver (build_solver, subgame_solving.cc:791-800) ---- */
void* orc_solver_create(int dice, int faces, int root_last_bid, int root_player, const double* beliefs0,
                        const double* beliefs1, const orc_params* params, int net_mode, orc_net_fn net_fn,
                        void* net_user, const char* torchscript_path, orc_example_fn ex_fn, void* ex_user);
void orc_solver_destroy(void* s);
int orc_solver_tree_size(void* s);
void orc_solver_step(void* s, int traverser);
void orc_solver_multistep(void* s);
void orc_solver_get(void* s, int which, double* out); /* dense [N][H][A] */
void orc_solver_hand_values(void* s, int player, double* out);
void orc_solver_update_value_network(void* s);

/* ---- self-play walk (RlRunner::step, recursive_solving.cc:160-182) : num_games games, examples through ex_fn ---- */
void orc_rl_run(int dice, int faces, double random_action_prob, int sample_leaf, const orc_params* params, int seed,
                int num_games, int net_mode, orc_net_fn net_fn, void* net_user, const char* torchscript_path,
                orc_example_fn ex_fn, void* ex_user);

/* ---- full-tree strategy by recursive subgame solving (recursive_solving.cc:47-134, 277-299): out dense [N_full][H][A];
 *      to_leaf = 0: compute_strategy_recursive, 1: compute_strategy_recursive_to_leaf ---- */
void orc_strategy_recursive(int dice, int faces, const orc_params* params, int to_leaf, int net_mode, orc_net_fn net_fn,
                            void* net_user, const char* torchscript_path, double* out);

/* ---- compute_sampled_strategy_recursive_to_leaf (recursive_solving.cc:301-327): every subgame is stopped at its own
 *      iteration, drawn from mt19937(seed) with weights (i even ? i/2+1 : 0), and contributes its SAMPLING strategy;
 *      root_only: subgames below the root are solved to the end of the game (max_depth = 100000) ---- */
void orc_strategy_recursive_sampled(int dice, int faces, const orc_params* params, int seed, int root_only, int net_mode,
                                    orc_net_fn net_fn, void* net_user, const char* torchscript_path, double* out);

/* ---- full-tree evaluation (subgame_solving.cc:802-816): strategy dense [N_full][H][A] ---- */
void orc_compute_exploitability2(int dice, int faces, const double* strategy, double out[2]);

/* ---- compute_ev2 (subgame_solving.cc:931-982): expected value of playing strategy1 against strategy2 from uniform
 *      beliefs, as player 0 (out[0]) and, with the roles swapped and the sign flipped, as player 1 (out[1]) ---- */
void orc_compute_ev2(int dice, int faces, const double* strategy1, const double* strategy2, double out[2]);

/* The synthetic belief net (a test double of ours; elementwise, exactly reproducible in IEEE float):
 *   v[h] = ((0.5f*q[2+A+h] - 0.25f*q[2+A+H+h]) + 0.125f*(q[1]-q[0])) + 0.0625f*q[2 + h % A]                      */
void orc_synthetic_net(const float* queries, int64_t rows, int64_t qsize, float* out, int64_t osize, int num_actions);

/* ---- compute_immediate_regrets (subgame_solving.cc:984-1050): strategies = n_strategies dense [N_full][H][A] arrays back
 *      to back; out [N_full][H] ---- */
void orc_immediate_regrets(int dice, int faces, const double* strategies, int n_strategies, double* out);

/* libstdc++ <random> as the reference uses it: out[3*rounds] = per round uniform_int(0, hi), uniform_real<float>(0,1),
 * discrete(w[0..nw)) from std::mt19937(seed) -- the yardstick for the device-side restatement of those algorithms */
void orc_rng_probe(int seed, int rounds, int hi, const double* w, int nw, double* out);

#ifdef __cplusplus
}
#endif
