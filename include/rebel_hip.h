/* include/rebel_hip.h -- C ABI of librebel_hip.so, the MI355X (gfx950) engine for ReBeL self-play data generation on
 * Liar's Dice.  Plain pointers and sizes only; no torch / pybind types.
 *
 * The reference (facebookresearch/rebel) has no C ABI: its seam is the pybind11 module `cfvpy.rela`
 * (csrc/liars_dice/rela/pybind.cc:119-213) over the C++ interfaces ISubgameSolver / IValueNet / RlRunner.  Every
 * entry point below names the reference interface it stands in for, so that a maintainer can bind it from the
 * reference's pybind layer -- and `rebel_amd/csrc/rela_module.cc` is exactly that binding.
 *
 * Conventions
 *   - every function returning int returns 0 on success, non-zero on failure; rbl_last_error() holds the message
 *     (thread-local).  Reference behaviour "throw std::runtime_error / assert" maps to a non-zero status.
 *   - a `lane` is one independent subgame solver / self-play game (what one reference data-gen thread holds);
 *     lanes [0, B) of an engine are advanced in lock-step by the same kernel launches.
 *   - dense strategy layout on the boundary is the reference's TreeStrategy, double[N][H][A]
 *     (subgame_solving.h:39); device-side layout is edge-indexed.
 *   - host pointers unless the name says `_dev`.
 */
#ifndef REBEL_HIP_H_
#define REBEL_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rbl_engine rbl_engine;     /* one GPU, one game (dice x faces), one set of solver params */
typedef struct rbl_selfplay rbl_selfplay; /* a set of RlRunner lanes on an engine */

/* SubgameSolvingParams (subgame_solving.h:43-58).  use_cfr = 1: CFR / linear CFR / DCFR (the path north_star names);
 * use_cfr = 0: fictitious play incl. linear_update and optimistic (FP, subgame_solving.cc:364-506). */
typedef struct {
  int32_t num_iters, max_depth, linear_update, use_cfr, optimistic, dcfr;
  double dcfr_alpha, dcfr_beta, dcfr_gamma;
} rbl_params;

/* Net2 weights (cfvpy/models.py:64-94): n_layers x [Linear(in,hid) -> LayerNorm(hid)? -> GELU] -> Linear(hid,out).
 * All fp32, row-major [out][in] as torch.nn.Linear stores them.  ln_* may be NULL when use_layer_norm == 0. */
typedef struct {
  int32_t n_layers, n_in, n_hidden, n_out, use_layer_norm;
  const float* const* w;    /* [n_layers]  w[l]: [n_hidden][l==0 ? n_in : n_hidden] */
  const float* const* b;    /* [n_layers]  [n_hidden] */
  const float* const* ln_w; /* [n_layers]  [n_hidden] */
  const float* const* ln_b; /* [n_layers]  [n_hidden] */
  const float* w_out;       /* [n_out][n_hidden] */
  const float* b_out;       /* [n_out] */
  float ln_eps;             /* torch default 1e-5 */
} rbl_mlp_weights;

/* IValueNet::compute_values (net_interface.h:28) as a callback: fill out[rows][n_out] from queries[rows][qsize].
 * With host_buffers=1 the pointers are host memory (engine copies D2H/H2D around the call); with 0 they are device
 * pointers on the engine's device: the engine stream is idle during the call, and the callee must either enqueue its
 * work on `stream` (a hipStream_t) or have completed its writes to `out` before it returns. */
typedef void (*rbl_net_fn)(void* user, const float* queries, int64_t rows, int64_t qsize, float* out, int64_t n_out,
                           void* stream);
/* IValueNet::add_training_example (net_interface.h:31-32), batched: n examples, lane ids for bookkeeping. */
typedef void (*rbl_example_fn)(void* user, int64_t n, const int32_t* lanes, const float* queries, int64_t qsize,
                               const float* values, int64_t n_out);

enum { RBL_GET_AVERAGE = 0, RBL_GET_LAST = 1, RBL_GET_REGRETS = 2, RBL_GET_SUM = 3,
       RBL_GET_SAMPLED = 4, RBL_GET_FINAL = 5 /* rbl_stream_get only: the last sampled repeat, the mean of the repeats */ };

const char* rbl_last_error(void);
int rbl_device_count(void);
const char* rbl_build_info(void); /* "gfx950 hipcc <version> ..." */

/* ---- game rules (liars_dice.h:46-155) and the BFS public tree (tree.h:51-70), host side, for callers/tests ---- */
int rbl_num_actions(int dice, int faces);
int rbl_num_hands(int dice, int faces);
int rbl_query_size(int dice, int faces); /* get_query_size, subgame_solving.cc:100-102 */
/* 6 ints per node {last_bid, player_id, children_begin, children_end, parent, depth}; returns N (fills min(N,cap)). */
int rbl_unroll_tree(int dice, int faces, int root_last_bid, int root_player, int max_depth, int32_t* out, int cap_nodes);

/* ---- engine ---- */
rbl_engine* rbl_engine_create(int device, int dice, int faces, const rbl_params* params, int max_lanes);
void rbl_engine_destroy(rbl_engine* e);
void* rbl_engine_stream(rbl_engine* e); /* hipStream_t all engine work is enqueued on */

/* value net selection (IValueNet implementations: real_net.cc:30-55 zero net; TorchScriptNet :57-87) */
int rbl_engine_set_net_zero(rbl_engine* e);
int rbl_engine_set_net_synthetic(rbl_engine* e); /* test double, elementwise; same formula as oracle/orc_api.h */
int rbl_engine_set_net_mlp(rbl_engine* e, const rbl_mlp_weights* w); /* ModelLocker ctor/updateModel, model_locker.h:56-79 */
/* Arithmetic of the fused MLP forward, applied by the NEXT rbl_engine_set_net_mlp (cfvpy/selfplay.py:42-43, 211:
 * `half_inference` turns the generating replicas into half modules):
 *   0  f32 parity (default): activations and weights as f16 hi + lo pairs, three f16 MFMA products per multiply, f32
 *      accumulation -- <= 1e-5 of the f32 module on O(1) outputs (4e-7 measured)
 *   1  activations rounded to f16 once per layer (round to nearest), weights keep their hi + lo pair: two products
 *   2  activations AND weights rounded to f16: one product -- the reference's half_inference semantics with f32
 *      accumulation, f32 LayerNorm / GELU and one rounding per layer where a half torch module has three
 * Modes 1 and 2 exist for Net2 with LayerNorm and one hidden layer of 256 (the register-resident kernel); anything else is
 * refused by rbl_engine_set_net_mlp. */
int rbl_engine_set_net_precision(rbl_engine* e, int mode);
int rbl_engine_set_net_callback(rbl_engine* e, rbl_net_fn fn, void* user, int host_buffers);

/* standalone batched forward of the current net (ModelLocker::forward, model_locker.h:85-95) */
int rbl_net_forward(rbl_engine* e, const float* queries, int64_t rows, float* out);          /* host in/out */
int rbl_net_forward_dev(rbl_engine* e, const float* queries_dev, int64_t rows, float* out_dev); /* device, async */

/* ---- batched subgame solver: build_solver (subgame_solving.cc:791-800) for B lanes at once ----
 * root_last_bid[b] in [-1, A-1), root_player[b] in {0,1}, beliefs [B][2][H] (NOT re-normalised, as the reference).
 * act_iteration may be NULL; if given, lane b's sigma_last after act_iteration[b] steps is kept for rbl_solver_get_snapshot. */
int rbl_solver_reset(rbl_engine* e, int B, const int32_t* root_last_bid, const int32_t* root_player,
                     const double* beliefs, const int32_t* act_iteration);
int rbl_solver_step(rbl_engine* e, int traverser);  /* ISubgameSolver::step, CFR::step subgame_solving.cc:577-664 */
int rbl_solver_multistep(rbl_engine* e, int n);     /* n steps, traverser = global_iter % 2 (:666-670); n<0: num_iters */
int rbl_solver_sync(rbl_engine* e);                 /* wait for enqueued steps */
int rbl_solver_num_lanes(rbl_engine* e);
int rbl_solver_tree_size(rbl_engine* e, int lane);
int64_t rbl_solver_total_rows(rbl_engine* e);       /* sum over lanes of pseudo-leaves = net rows per step */
/* dense double[N][H][A]; which = RBL_GET_* (get_strategy / get_sampling_strategy :678-688; regrets, sum_strategies) */
int rbl_solver_get(rbl_engine* e, int lane, int which, double* out);
int rbl_solver_get_snapshot(rbl_engine* e, int lane, double* out); /* sigma_last at the lane's act_iteration */
/* ---- evaluation: best response / exploitability on the device ----
 * rbl_solver_set_strategy: overwrite lane's sigma with a dense [N][H][A] strategy (rows for both players);
 * rbl_solver_best_response: BRSolver::compute_br (subgame_solving.cc:316-358) of `traverser` against every lane's
 *   sigma, out[B][H] = root values; pseudo-leaves (depth-limited trees) are valued by the engine's net;
 * rbl_exploitability2: compute_exploitability2 (subgame_solving.cc:802-816) of a full-tree strategy [N_full][H][A]. */
int rbl_solver_set_strategy(rbl_engine* e, int lane, const double* strategy);
int rbl_solver_best_response(rbl_engine* e, int traverser, double* out);
int rbl_exploitability2(int device, int dice, int faces, const double* strategy, double out[2]);
/* compute_ev (subgame_solving.cc:931-973) per lane: root values [B][H] of the traverser following the lane's sigma (the
 * opponent's reach under the same sigma); rbl_ev2: compute_ev2 (:975-982) of two dense full-tree strategies from uniform
 * beliefs -- out[0] = EV of strategy1 as player 0 against strategy2, out[1] = -(EV of strategy2 as player 0 against
 * strategy1), as recursive_eval.cc:372 reports them. */
int rbl_solver_evaluate(rbl_engine* e, int traverser, double* out);
int rbl_ev2(int device, int dice, int faces, const double* strategy1, const double* strategy2, double out[2]);
/* compute_strategy_recursive (to_leaf = 0) / compute_strategy_recursive_to_leaf (1) (recursive_solving.cc:277-299) with
 * the engine's params and net: out = dense full-tree strategy [N_full][H][A], N_full = rbl_unroll_tree(.., -1, 0, 1<<20).
 * The frontier is solved level by level, all subgames of a level as lanes of one launch sequence. */
int rbl_strategy_recursive(rbl_engine* e, int to_leaf, double* out);
/* compute_sampled_strategy_recursive_to_leaf (recursive_solving.cc:301-327; the core of recursive_eval.cc:116-160): as the
 * to-leaf variant, but every subgame is stopped at its own iteration drawn from mt19937(seed) (weights i/2+1 on even i, in
 * the reference's solver-construction order) and contributes its sampling strategy.  root_only != 0: subgames below the
 * root are solved to the end of the game without the net (max_depth = 100000), on a helper engine of the same device. */
int rbl_strategy_recursive_sampled(rbl_engine* e, int seed, int root_only, double* out);
/* The same to-leaf recursion (recursive_solving.cc:76-134, use_sampling_strategy = false) followed by
 * compute_exploitability2 (subgame_solving.cc:802-816, BRSolver::compute_br :316-358) WITHOUT the dense [N][H][A] strategy
 * (241 GB for 2 dice x 6 faces, recursive_eval.cc:269-363): the full-tree strategy stays on the device, edge-indexed
 * [N - 1][H] fp64, written level by level (every subgame of a level = a lane, engine params and net) and consumed in place
 * by two level-synchronous best-response sweeps.  out[2] = the two players' exploitabilities (n_shards == 1).
 * Sharding: the pseudo-leaves of the root subgame (the nodes at depth max_depth) are dealt to the n shards, largest
 * subtree first; shard s follows only its own.  top_values (optional, [2][M][H], M = nodes of depth <= max_depth =
 * rbl_unroll_tree(.., -1, 0, max_depth)) receives the best-response values of those nodes per traverser and top_owner
 * (optional, int32[M]) the owning shard of every non-terminal depth-max_depth node (-1 elsewhere); the caller redoes the
 * sweep over the top levels on the host with each depth-max_depth value taken from its owner; out = NaN if n > 1.
 * stats (optional, double[8]): full-tree nodes, subgames solved, recursion levels, seconds in the recursion, seconds in
 * the two sweeps, bytes of the device-resident strategy, frontier items followed, M. */
int rbl_exploitability_recursive(rbl_engine* e, int shard, int n_shards, double out[2], double* top_values,
                                 int32_t* top_owner, double* stats);
/* The same with the dealt level chosen by the caller.  deal_levels = K >= 1: the frontier PRODUCED by recursion level K - 1
 * (the non-terminal nodes at depth K * max_depth) is dealt to the shards, largest subtree first; the K levels above are
 * solved by every shard.  K = 1 is rbl_exploitability_recursive: a depth-max_depth subtree can be a quarter of the game
 * (2 dice x 6 faces: 8 shards -> 3.7x).  K = 2: every shard redundantly solves the root subgame and its depth-max_depth
 * subgames (277 of 8.4 M at 2 dice x 6 faces), the largest dealt subtree is 1/16 of the game and 8 shards balance.
 * M = rbl_exploitability_top_nodes(dice, faces, max_depth, K) = nodes of depth <= K * max_depth sizes top_values [2][M][H]
 * and top_owner [M].  The shards are independent processes / GPUs; nothing is exchanged between them on the device
 * (recursive_solving.cc:76-134 recursion order is irrelevant to the result: subgames of a level are independent). */
int rbl_exploitability_recursive_deal(rbl_engine* e, int shard, int n_shards, int deal_levels, double out[2],
                                      double* top_values, int32_t* top_owner, double* stats);
int64_t rbl_exploitability_top_nodes(int dice, int faces, int max_depth, int deal_levels); /* -1: error */
/* Shards -> compute_exploitability2's two numbers (subgame_solving.cc:802-816), on the host: BRSolver::compute_br
 * (:326-355) over the top M nodes, every dealt node's value taken from its owner (top_values[owner]), everything else from
 * shard 0; maximum = first child then strictly greater, opponent sums ascending, vector_sum / H.  top_values: n_shards
 * pointers to the shards' [2][M][H] arrays; top_owner: any shard's [M] map (they are identical).  Bit-identical to the
 * unsharded out[2]. */
int rbl_exploitability_combine(int dice, int faces, int max_depth, int deal_levels, int n_shards,
                               const double* const* top_values, const int32_t* top_owner, double out[2]);
/* compute_immediate_regrets (subgame_solving.cc:984-1050; printed by recursive_eval --print_regret[_summary],
 * recursive_eval.cc:28-53): strategies = n_strategies dense full-tree strategies [N_full][H][A] back to back; out[N_full][H] =
 * max over the actions of the regret accumulated over all strategies and both traversers, divided by n_strategies (0 on
 * nodes without children).  Runs as plain CFR regret updates of a full-tree solver on the device. */
int rbl_immediate_regrets(int device, int dice, int faces, const double* strategies, int n_strategies, double* out);
int rbl_solver_hand_values(rbl_engine* e, int lane, int player, double* out); /* get_hand_values :694-696 */
/* update_value_network (:672-676): writes the lane's two training examples, queries[2][Q], values[2][H] */
int rbl_solver_examples(rbl_engine* e, int lane, float* queries, float* values);
/* last queries written for the net, for inspection: out[rows][Q]; rows = rbl_solver_total_rows */
int rbl_solver_get_queries(rbl_engine* e, float* out);
/* developer aid (env RBL_CFR_DBG=1 at engine creation): out[B][16] shader-clock stamps of the last CFR launch's phases */
int rbl_solver_debug_stamps(rbl_engine* e, long long* out);
/* developer aid (env RBL_NET_DBG=1): out[1024][16] shader-clock stamps of the last net forward's first 1024 workgroups */
int rbl_net_debug_stamps(rbl_engine* e, long long* out);

/* ---- full-tree CFR without a dense tabulation (eval_stream.hip): "Solving the game for the full tree" of the reference's
 * evaluation tool (recursive_eval.cc:269-296: build_solver with max_depth = 100000, CFR::step x subgame_iters,
 * compute_exploitability2 of get_strategy() along the way) as level-synchronous sweeps over edge-indexed arrays in HBM
 * (sigma, regrets, sum_strategies [N-1][H], reach and values [N][H]: 58 GB at 2 dice x 6 faces).  CFR only (use_cfr = 1;
 * linear / DCFR discounts as in SubgameSolvingParams).  rbl_stream_step: n x CFR::step(iteration parity);
 * rbl_stream_exploitability: of the average strategy; rbl_stream_get: dense [N][H][A] copies (which = RBL_GET_*) for games
 * that fit (tests).  Errors: rbl_stream_last_error(). ---- */
typedef struct rbl_stream rbl_stream;
rbl_stream* rbl_stream_create(int device, int dice, int faces, const rbl_params* params);
void rbl_stream_destroy(rbl_stream* s);
int64_t rbl_stream_num_nodes(rbl_stream* s);
int rbl_stream_step(rbl_stream* s, int n_steps);
int rbl_stream_exploitability(rbl_stream* s, double out[2]);
int rbl_stream_get(rbl_stream* s, int which, double* out);
const char* rbl_stream_last_error(void);
/* "Recursive solving" of the same tool (recursive_eval.cc:320-388) on the same arrays.  rbl_stream_sampled_add: one repeat =
 * compute_sampled_strategy_recursive_to_leaf(game, params of `e`, net of `e`, seed, root_only = false)
 * (recursive_solving.cc:301-327; every subgame of a recursion level is a lane of `e`, act_iteration drawn per subgame in the
 * reference's construction order), added to summed_strategy / summed_reach (float32 like the reference's tensors,
 * recursive_eval.cc:136-160, 343-349).  rbl_stream_sampled_eval: final_strategy = summed_strategy / (summed_reach + 1e-6),
 * its compute_exploitability2 and compute_ev2(full-tree average strategy of `s`, final_strategy) (:352-371).
 * rbl_stream_sampled_reset: forget the repeats. */
int rbl_stream_sampled_reset(rbl_stream* s);
int rbl_stream_sampled_add(rbl_stream* s, rbl_engine* e, int seed);
/* The same repeat with root_only = true (recursive_solving.cc:318-320, recursive_eval --root_only): the root subgame on a lane
 * of `e`, then every subgame below it solved to the END OF THE GAME (max_depth = 100000, no value net), each stopped at its own
 * act_iteration.  Those subgames (276 trees of up to 4 M nodes at 2 dice x 6 faces) are solved together as a forest on the
 * full tree's edge-indexed arrays: level-synchronous CFR sweeps over the nodes below depth max_depth, every node gated by the
 * stop iteration of its subtree.  Bit-identical to rbl_strategy_recursive_sampled(e, seed, 1) where that fits. */
int rbl_stream_sampled_add_root_only(rbl_stream* s, rbl_engine* e, int seed);
/* report_regrets of the same tool (recursive_eval.cc:28-53 -> compute_immediate_regrets, subgame_solving.cc:984-1050) without
 * the list of dense strategies: regrets accumulate on the device, strategy by strategy.  regrets_add(which): RBL_GET_LAST = the
 * full-tree solver's current sampling strategy (the tool lists it after every even iteration, :285-287); RBL_GET_SAMPLED = the
 * last sampled repeat, rounded to float as the tool's tensor round trip does (:357-358).  regrets_report: first[n_first][H] =
 * immediate regrets of the first nodes (the tool prints 20), sums = {vector_sum over the nodes of depth < `depth`, over the
 * rest}.  Bit-identical to rbl_immediate_regrets on the same strategies. */
int rbl_stream_regrets_reset(rbl_stream* s);
int rbl_stream_regrets_add(rbl_stream* s, int which);
int rbl_stream_regrets_report(rbl_stream* s, int depth, int n_first, double* first, double sums[2]);
int rbl_stream_sampled_eval(rbl_stream* s, double exploitability[2], double ev_of_full[2]);

/* ---- self-play lanes: RlRunner (recursive_solving.h:40-86), one per seed (create_cfr_thread, pybind.cc:36-43) ---- */
rbl_selfplay* rbl_selfplay_create(rbl_engine* e, int n_lanes, const int32_t* seeds, double random_action_prob,
                                  int sample_leaf);
void rbl_selfplay_destroy(rbl_selfplay* sp);
/* Advances every lane by ONE subgame (num_iters CFR steps + sampling, recursive_solving.cc:166-181) and hands the
 * 2*n_lanes training examples to `sink` in lane order.  Returns the number of subgame-CFR-iterations executed
 * (n_lanes * num_iters) or -1 on error.
 * The whole epoch -- RlRunner::step's draws (act_iteration :168-169), the sampling walk (sample_state_to_leaf :192-246 /
 * sample_state_single :248-275), the Bayes updates (:41-44) and the example encoding (subgame_solving.cc:672-676) -- runs
 * as HIP kernels between the CFR launches (selfplay_kernels.hip); the host only receives the examples.  With a callback
 * net (rbl_engine_set_net_callback) or RBL_SELFPLAY_HOST=1 the walk runs on the host instead: same trajectories. */
int64_t rbl_selfplay_advance(rbl_selfplay* sp, rbl_example_fn sink, void* user);
/* 1 if the lanes' walk runs on the device, 0 on the host; the choice is made from the engine's net at the first call of
 * this function or of rbl_selfplay_advance, whichever comes first, and then stays.  -1 on error. */
int rbl_selfplay_on_device(rbl_selfplay* sp);
/* Device pointers to the LAST epoch's examples, queries [2*n_lanes][Q] and values [2*n_lanes][H] f32, valid until the next
 * advance (replaces per-example tensor allocation, subgame_solving.cc:220-226, for a device-resident replay buffer);
 * both null when the walk runs on the host. */
int rbl_selfplay_device_examples(rbl_selfplay* sp, const float** queries_dev, const float** values_dev);
/* Self-test of the device restatement of libstdc++'s <random> (std::mt19937(seed) driving uniform_int_distribution<int>(0,
 * hi), uniform_real_distribution<float>(0,1), discrete_distribution<int>(w, w + nw)): out[3 * rounds] = the draws, in that
 * order per round, produced by the GPU.  tests/ compares them with the host library draw for draw. */
int rbl_selftest_device_rng(int device, int32_t seed, int rounds, int hi, const double* w, int nw, double* out);
int64_t rbl_selfplay_games_finished(rbl_selfplay* sp);
/* Root de-duplication (opt-in extra, REBEL_AMD_ROOT_DEDUP=1 in the environment when the lanes are created; default off; no
 * counterpart in the reference).  RlRunner::step resets a finished game to the root state with uniform beliefs
 * (recursive_solving.cc:160-163) and CFR::step draws nothing, so every lane that is in its root subgame in an epoch computes the
 * same num_iters iterations.  With the option on, one lane per epoch solves the root subgame and the other root lanes sample from
 * its strategy at THEIR act_iteration: trajectories and training examples stay bit-identical per seed, rbl_selfplay_advance then
 * returns the iterations actually EXECUTED, and this call returns the lane-epochs served that way so far (0 when off).
 * (The per-lane solver arrays of a served lane -- rbl_solver_get on the engine underneath -- are NOT maintained in such an epoch:
 * only what the reference's data path observes is, the examples and the next public state.) */
int64_t rbl_selfplay_root_dedup_served(rbl_selfplay* sp);
/* per-lane public state for inspection: last_bid, player_id (liars_dice.h:35-44) */
int rbl_selfplay_state(rbl_selfplay* sp, int lane, int32_t* last_bid, int32_t* player_id);

/* ---- timing of the two dominant kernels since the last reset (HIP events on the engine stream) ---- */
typedef struct {
  double cfr_ms, net_ms;              /* summed duration of the TIMED launches */
  int64_t cfr_launches, net_launches; /* number of timed launches */
  int64_t net_rows;                   /* rows pushed through the net by the timed launches */
  int64_t lane_steps;                 /* subgame-CFR-iterations executed (all launches, timed or not) */
  double cfr_bytes, net_flops;        /* algorithmic bytes / flops of the timed launches (per-lane shapes) */
  /* what the engine actually launched last (reporting; not reset): CFR step kernel 0 generic (cfr_step_kernel), 1 one
   * thread per tree row (cfr_rows_kernel), 2 one wavefront per lane (cfr_wave_kernel), 3 rows kernel with global state
   * (2 dice x 6 faces, RBL_CFR_FLAT=0), 4 element-parallel kernel with sigma in LDS (cfr_flat_kernel, 2 dice x 6 faces);
   * value-net kernel variant (MlpDev::tile: 5 register-resident, 3 feature split; 0 =
   * no MLP net); number of lane parts = streams of the last batch */
  int32_t cfr_kernel, net_kernel, n_streams;
  int32_t net_products; /* f16 MFMA products per multiply of the MLP forward: 3, 2 or 1 (rbl_engine_set_net_precision); 0 = no MLP */
} rbl_kernel_stats;
/* stride = 0: off; n > 0: bracket the CFR and net launches of every n-th iteration with HIP events.  Use an ODD n: the
 * traverser of iteration i is i mod 2 and the two traversers' steps differ in cost (an even stride samples one of them only) */
int rbl_engine_timing(rbl_engine* e, int stride);
int rbl_engine_stats(rbl_engine* e, rbl_kernel_stats* out, int reset);

#ifdef __cplusplus
}
#endif
#endif /* REBEL_HIP_H_ */
