// rebel_amd/csrc/engine.h -- host side of the MI355X engine: device tables, lane state, step scheduling, self-play lanes.
//
// Mirrors, lane-batched, what one reference data-gen thread owns:
//   Engine   ~ build_solver + ISubgameSolver (subgame_solving.h:60-88, subgame_solving.cc:791-800) for B subgames at once
//   SelfPlay ~ RlRunner (recursive_solving.h:40-86, recursive_solving.cc:160-275), one RNG per lane
// The C ABI in include/rebel_hip.h is a thin wrapper over these two classes (capi in engine.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <mutex>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rebel_hip.h"
#include "cfr_kernels.h"
#include "net_kernels.h"
#include "selfplay_kernels.h"
#include "tables.h"

namespace rbl {

#define RBL_HIP_CHECK(expr)                                                                                   \
  do {                                                                                                        \
    hipError_t _e = (expr);                                                                                   \
    if (_e != hipSuccess)                                                                                     \
      throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(_e) + " at " + __FILE__ + ":" + \
                               std::to_string(__LINE__) + " (" #expr ")");                                    \
  } while (0)

template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  void alloc(size_t count) {
    release();
    if (count == 0) count = 1;
    RBL_HIP_CHECK(hipMalloc(&p, count * sizeof(T)));
    n = count;
  }
  void upload(const std::vector<T>& h, hipStream_t s) {
    if (h.size() > n) alloc(h.size());
    if (!h.empty()) RBL_HIP_CHECK(hipMemcpyAsync(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s));
  }
};

enum class NetMode { kZero, kSynthetic, kMlp, kCallback };

class Engine {
 public:
  Engine(int device, int dice, int faces, const rbl_params& params, int max_lanes);
  ~Engine();

  // ---- value net
  void set_net_zero();
  void set_net_synthetic();
  void set_net_mlp(const rbl_mlp_weights& w);
  void set_net_precision(int mode);
  int net_precision() const { return net_precision_; }
  void set_net_callback(rbl_net_fn fn, void* user, bool host_buffers);
  void net_forward_dev(const float* q_dev, int64_t rows, float* out_dev, hipStream_t st = nullptr,
                       const long long* range = nullptr);  // async; `range`: see launch_mlp_forward
  void net_forward_host(const float* q, int64_t rows, float* out);

  // ---- batched solver
  void reset(int B, const int32_t* root_last_bid, const int32_t* root_player, const double* beliefs,
             const int32_t* act_iteration);
  void step(int traverser);
  void multistep(int n);
  void sync();
  int num_lanes() const { return B_; }
  int tree_size(int lane);
  int64_t total_rows() const { return rows_; }
  void get(int lane, int which, double* out);
  void get_snapshot(int lane, double* out);
  void set_strategy(int lane, const double* dense);
  void best_response(int traverser, double* out);
  void evaluate(int traverser, double* out);  // compute_ev: root values of following sigma, per lane [H]
  void hand_values(int lane, int player, double* out);
  void examples(int lane, float* queries, float* values);
  void get_queries(float* out);
  void get_debug(long long* out);
  void get_net_debug(long long* out);  // RBL_NET_DBG=1: phase stamps of the last net forward (first 1024 workgroups)  // RBL_CFR_DBG=1: per-lane phase timestamps of the last CFR launch

  // ---- device-resident epochs (SelfPlay): lane descriptors, row offsets and part boundaries are produced by kernels
  // (selfplay_kernels.hip) on stream(); the host never learns an epoch's row counts before its end
  struct DeviceLanes {
    int *shape, *player, *row, *act;
    double* beliefs;
    const double *snapshot, *root_mean;
    const int* shape_epar;
  };
  DeviceLanes device_lanes() const {
    return DeviceLanes{d_lane_shape_.p, d_lane_player_.p, d_lane_row_.p, d_lane_act_.p, d_beliefs_.p, d_snapshot_.p,
                       d_root_mean_.p, d_shape_epar_.p};
  }
  // read-only view of the solver state for the streaming evaluation (eval_stream.hip); valid until the next reset()
  struct EvalView {
    const double *sums, *sigma, *snapshot, *beliefs;
    const int *lane_shape, *lane_player;
    const ShapeDev* shapes;
    const int *parent, *cb, *ce, *depth, *leaves;
    int num_steps[2];
    bool use_cfr;
  };
  EvalView eval_view() const {
    return EvalView{d_sums_.p,   d_sigma_.p, d_snapshot_.p, d_beliefs_.p, d_lane_shape_.p, d_lane_player_.p, d_shapes_.p, d_parent_.p,
                    d_cb_.p,     d_ce_.p,    d_depth_.p,   d_leaves_.p,     {num_steps_[0], num_steps_[1]}, p_.use_cfr != 0};
  }
  // cfr_rows_kernel<GS> (2 dice x 6 faces) is launched per size-sorted segment of a part: device-resident epochs get the
  // order from sp_order (selfplay_kernels.hip), which needs this buffer; null when the engine does not sort
  int* lane_order_dev() const { return use_order_ ? d_lane_order_.p : nullptr; }
  const ShapeDev* shapes_dev() const { return d_shapes_.p; }
  const int* act_dev() const { return d_act_.p; }
  const int* cb_dev() const { return d_cb_.p; }
  const int* ce_dev() const { return d_ce_.p; }
  const int* depth_dev() const { return d_depth_.p; }
  bool device_epochs_supported() const { return net_mode_ != NetMode::kCallback; }
  int parts_for(int B) const;                                    // lane parts (= streams) a batch of B lanes is split into
  int net_grid_cap(int B) const;                                 // persistent net workgroups for a batch of B lanes (0 = one per CU)
  void part_lanes(int B, int* part_lane /*[kSpMaxParts+1]*/) const;
  void begin_epoch_device(int B, const SpEpochInfo* info_dev);   // descriptors already enqueued on stream(); solver init
  void join_streams();                                           // stream() waits for the other parts' streams
  // root de-duplication of device-resident epochs (selfplay_kernels.h): allocates the per-lane flags and the representative's
  // sigma-per-iteration slab; false (nothing changed) when this engine's step kernels cannot skip lanes
  bool enable_root_dedup();
  bool root_dedup() const { return root_dedup_; }
  int* lane_skip_dev() const { return root_dedup_ ? d_lane_skip_.p : nullptr; }
  const double* snap_all_dev() const { return root_dedup_ ? d_snap_all_.p : nullptr; }
  void end_epoch_device(const SpEpochInfo& info);                // accounting of the timed launches; rows of the epoch

  // bulk read-back used by SelfPlay (edge-indexed, stride Emax*H per lane)
  void read_snapshots(const double** snap, const double** root_mean);  // pinned host copies, valid until the next call

  void timing(int stride);  // 0 = off, n = time the launches of every n-th CFR iteration with HIP events
  void stats(rbl_kernel_stats* out, bool reset);

  const Rules& rules() const { return g_; }
  int device() const { return device_; }
  const rbl_params& params() const { return p_; }
  const ShapeTables& tables() const { return tabs_; }
  hipStream_t stream() const { return stream_; }
  int emax() const { return emax_; }
  int max_lanes() const { return max_lanes_; }
  void write_root_query(int traverser, int last_bid, int player, const double* b0, const double* b1, float* q) const;

 private:
  void construct();
  void ensure_mirror();  // device-resident epochs: refresh the host copies of the lane descriptors on demand
  void release_handles();
  void check_lane(int lane) const;
  void run_net();
  void launch(int mode, int trav, int next_trav, int steps_after, double alpha, double pos, double neg, double strat);
  void expand_dense(int lane, const std::vector<double>& edge, double* out) const;
  void read_lane(const double* dev_base, int lane, std::vector<double>* out);
  struct Timed;
  void time_begin(int kind, hipStream_t st);
  bool time_end(int kind, hipStream_t st);  // false: the sample was dropped (see launch_timing.h)
  void time_abort();                        // forget a sample that time_begin opened and time_end never closed
  struct TimingAbortGuard {                 // scope guard of launch() / run_net(): disarms the slot, drops an open sample
    Engine* e;
    ~TimingAbortGuard() { e->time_abort(); }
  };

  int device_;
  Rules g_;
  rbl_params p_;
  ShapeTables tabs_;
  int max_lanes_, emax_, nmax_;
  hipStream_t stream_ = nullptr, stream2_ = nullptr;
  hipEvent_t ev_ready_ = nullptr;
  hipEvent_t ev_join_[3] = {nullptr, nullptr, nullptr};
  const SpEpochInfo* info_dev_ = nullptr;  // non-null: device-resident epoch (row ranges live on the device)
  bool mirror_valid_ = true;
  long long timed_cfr_[4][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}}, timed_net_[4] = {0, 0, 0, 0};
  hipStream_t stream_x_[2] = {nullptr, nullptr};
  int n_parts_ = 1, max_parts_ = 2, only_part_ = -1, split_min_lanes_ = 1024;
  int part_lane_[5] = {0, 0, 0, 0, 0};
  int64_t part_row_[5] = {0, 0, 0, 0, 0};
  double part_bytes_[4][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
  hipStream_t part_stream(int part) const { return part == 0 ? stream_ : part == 1 ? stream2_ : stream_x_[part - 2]; }

  DevBuf<ShapeDev> d_shapes_;
  DevBuf<int> d_parent_, d_act_, d_cb_, d_ce_, d_depth_, d_leaves_, d_terms_, d_irank_, d_leaf_row_, d_vrow_, d_pack_;
  DevBuf<int8_t> d_matches_, d_wave_tabs_;
  DevBuf<unsigned short> d_wave_epv_;
  DevBuf<int> d_flat_tabs_;
  DevBuf<LaneRec> d_shape_rec_, d_lane_rec_;  // per-shape templates / per-launch-slot records (cfr_kernels.h: LaneRec)
  bool wave_tabs_ok_ = false;                 // the game's trees fit cfr_wave_kernel's byte tables and 15-bit offsets
  bool root_dedup_ = false;
  DevBuf<int> d_lane_skip_;
  DevBuf<double> d_snap_all_;
  void build_lane_records();                  // enqueue lane_rec_kernel on stream_ behind the lane descriptors (and the lane order)
  DevBuf<int> d_shape_epar_;
  DevBuf<int> d_lane_shape_, d_lane_player_, d_lane_row_, d_lane_act_;
  DevBuf<double> d_beliefs_, d_sigma_, d_regrets_, d_sums_, d_snapshot_, d_root_mean_, d_scratch_;
  DevBuf<long long> d_dbg_, d_ndbg_;
  DevBuf<double> d_br_;
  double* h_pinned_ = nullptr;
  DevBuf<float> d_queries_, d_values_, d_mlp_blob_, d_tmp_q_, d_tmp_o_;

  std::vector<int> h_shape_, h_player_, h_row_, h_act_, h_bid_;
  std::vector<double> h_beliefs_;
  int B_ = 0;
  int64_t rows_ = 0;
  bool has_act_ = false;
  int iter_ = 0, num_steps_[2] = {0, 0}, pending_trav_ = -1, num_strategies_ = 0;

  std::mutex net_mutex_;  // weight refresh (another thread) vs the net launch inside step()
  NetMode net_mode_ = NetMode::kZero;
  bool values_zeroed_ = false;
  MlpDev mlp_;
  // split query layout between cfr_wave_kernel and the resident MLP forward (cfr_kernels.h: CfrArgs::q_dyn): on when both are
  // in use; the canonical [rows][Q] buffer is then current only after the init / query-only launches (get_queries rebuilds it)
  bool qsplit_ = false, q_canon_stale_ = false;
  int q_ds_ = 0, q_ss_ = 0, mlp_n_in_true_ = 0;
  int net_precision_ = 0;  // rbl_engine_set_net_precision: applied by the next set_net_mlp
  DevBuf<float> d_qdyn_, d_qstat_, d_tmp_dyn_, d_tmp_stat_;
  void split_part_queries(int part, hipStream_t st);  // canonical rows of a lane part -> dyn / stat rows
  void leave_split_layout();  // net exchanged mid-solve: rebuild the canonical rows the next net reads (holds net_mutex_)
  void enter_split_layout();  // ... or the split rows, from the canonical ones
  rbl_net_fn cb_fn_ = nullptr;
  void* cb_user_ = nullptr;
  bool cb_host_ = true;
  std::vector<float> h_q_, h_v_;

  int block_ = 64;
  bool rows_ok_ = false;  // kModeStep runs on cfr_rows_kernel (one thread per tree row)
  int rows_block_ = 128;
  size_t rows_lds_bytes_ = 0;
  bool wave_ok_ = false;  // kModeStep runs on cfr_wave_kernel (one wavefront per lane)
  size_t wave_lds_bytes_ = 0;
  int n_cus_ = 256, net_grid_env_ = -1;  // CUs of the device; RBL_NET_GRID (-1: automatic)
  bool rows_global_ok_ = false;  // big games: row kernel with sigma / regrets in place in global memory
  size_t rows_global_lds_ = 0;
  int flat_threads_ = 1024;
  bool flat_ok_ = false;         // ... served by cfr_flat_kernel (element-parallel, sigma in LDS) instead
  size_t gs_lds_bytes(const ShapeDev& s) const {
    return flat_ok_ ? cfr_flat_lds_bytes(s.N, s.NI, g_.H, s.L, s.T, g_.faces) : cfr_rows_global_lds_bytes(s.N, s.NI, g_.H, s.L, g_.faces);
  }
  // ... launched per segment of the part's lanes sorted by tree size, each with the LDS request of ITS largest tree: one
  // root-sized lane per CU (124 KB), but two to four of the smaller trees that make up most of a self-play batch
  bool use_order_ = false;
  DevBuf<int> d_lane_order_;
  size_t seg_lds_[4][kSpSegs] = {};
  int seg_threads_[4][kSpSegs] = {};  // cfr_flat_kernel: workgroup size of a segment, by its largest tree
  // a lane's passes are (node, hand) items: small trees on a 1024-thread workgroup leave most of it idle and, at 16 waves,
  // have the CU to themselves; sized to ~12 items per thread instead, eight of them share a CU
  int flat_threads_for(int N) const {
    const int want = N > 200 ? 1024 : (N > 80 ? 512 : (N > 30 ? 256 : 128));
    return std::min(want, flat_threads_);
  }
  void set_segments(const int (*seg_shape)[kSpSegs]);  // LDS request of each launch segment from its head lane's shape
  size_t part_rows_lds_[4] = {0, 0, 0, 0};  // per part: LDS of the largest tree among its lanes (set by reset)
  int part_rows_block_[4] = {128, 128, 128, 128};
  bool rows_fit_ = true;  // size the row kernel's launch to the largest tree of the part
  size_t lds_bytes_ = 0, work_stride_ = 0;
  bool use_lds_ = true;
  bool cfr_dbg_ = false;

  // accounting
  bool timing_ = false;
  int timing_stride_ = 0;
  bool timed_now() const { return timing_ && (iter_ % timing_stride_ == 0); }
  std::vector<hipEvent_t> ev_pool_;
  struct Pending {
    int kind;
    size_t e0, e1;
  };
  // per kind (0 CFR step, 1 net forward): time through events bound to the dispatch packet (gap-free, what rocprofv3
  // reports) while the launch is ONE kernel of a launcher that supports it; otherwise bracket with recorded events
  bool ext_timing_[2] = {true, true};
  bool ext_armed_ = false;
  bool sample_open_ = false;  // between time_begin and time_end
  std::vector<Pending> pending_;
  size_t ev_used_ = 0;
  rbl_kernel_stats stats_{};
  int last_cfr_kernel_ = 0;  // which CFR step kernel the last step launch used (rbl_kernel_stats::cfr_kernel)
  double step_bytes_[2] = {0, 0};  // algorithmic bytes of one CFR step per traverser, summed over lanes
};

// eval_stream.hip: compute_strategy_recursive_to_leaf + compute_exploitability2 with the full-tree strategy kept on the
// device, edge-indexed (no dense [N][H][A] tabulation); see include/rebel_hip.h: rbl_exploitability_recursive
Engine& engine_impl(rbl_engine* e);  // the engine behind a C handle (engine.hip)
void exploitability_recursive(Engine& e, int shard, int n_shards, double* out2, double* top_values, int32_t* top_owner,
                              double* stats, int deal_levels = 1);
int64_t exploitability_top_nodes(const Rules& g, int max_depth, int deal_levels);
void exploitability_combine(const Rules& g, int max_depth, int deal_levels, int n_shards, const double* const* top_values,
                            const int32_t* top_owner, double* out2);

class SelfPlay {
 public:
  SelfPlay(Engine* e, int n_lanes, const int32_t* seeds, double random_action_prob, bool sample_leaf);
  ~SelfPlay();
  SelfPlay(const SelfPlay&) = delete;
  SelfPlay& operator=(const SelfPlay&) = delete;
  int64_t advance(rbl_example_fn sink, void* user);
  int decide_mode();  // fixes host / device walk from the engine's current net (first call), returns 0 / 1
  bool on_device() const { return mode_ == 1; }  // the walk runs as kernels (selfplay_kernels.hip), decided at the first advance()
  // the last epoch's examples as device pointers ([2n][Q], [2n][H]), valid until the next advance(); null in host mode
  void device_examples(const float** q, const float** v) const;
  int64_t games_finished() const { return games_; }
  int64_t lanes_served_by_root_dedup() const { return skipped_; }  // lane-epochs that read the representative's root solve
  void state(int lane, int32_t* last_bid, int32_t* player) const;
  int num_lanes() const { return n_; }

 private:
  void sample_to_leaf(int lane, const double* sigma);
  void sample_single(int lane, const double* sigma);
  void bayes(double* b, const double* sigma, int shape_node_off, int child, int H) const;

  int64_t advance_host(rbl_example_fn sink, void* user);
  int64_t advance_device(rbl_example_fn sink, void* user);
  void init_device();
  SpArgs sp_args() const;

  Engine* e_;
  int n_;
  float rap_;
  bool leaf_;
  int mode_ = -1;  // -1 undecided, 0 host walk (callback nets, RBL_SELFPLAY_HOST=1), 1 device walk
  std::vector<int32_t> seeds_;
  // device walk: RNG, game state and the epoch's examples live on the GPU; the host sees examples + (bid, player)
  DevBuf<uint32_t> d_mt_;
  DevBuf<int> d_mt_idx_, d_bid_, d_player_;
  DevBuf<double> d_sp_beliefs_;
  DevBuf<float> d_ex_q_, d_ex_v_;
  DevBuf<SpEpochInfo> d_info_;
  unsigned char* h_pin_ = nullptr;  // pinned: ex_q, ex_v, bid, player, info
  std::vector<std::mt19937> gen_;
  std::vector<int32_t> bid_, player_, act_;
  std::vector<double> beliefs_;  // [n][2][H]
  std::vector<float> ex_q_, ex_v_;
  std::vector<int32_t> ex_lane_;
  int64_t games_ = 0, skipped_ = 0;
};

}  // namespace rbl
