// rebel_amd/csrc/engine.hip -- Engine / SelfPlay implementation and the C ABI (include/rebel_hip.h).
// Host code only (the kernels live in cfr_kernels.hip / net_kernels.hip); compiled with -ffp-contract=off because the
// few fp64 formulas evaluated here (discounts, root queries, average strategy read-back, Bayes updates while
// sampling) are part of the bit-exactness contract with the reference.
#include "engine.h"
#include "launch_timing.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <random>
#include <thread>

namespace rbl {

thread_local LaunchTimingSlot tl_launch_timing;

namespace {

constexpr double kEps = 1e-80;  // subgame_solving.h:34-36

template <class T>
void normalize_safe(const double* x, int n, double eps, T* out) {  // util.h:68-78
  double sum = 0;
  for (int i = 0; i < n; ++i) sum += x[i] + eps;
  for (int i = 0; i < n; ++i) out[i] = (T)((x[i] + eps) / sum);
}

int env_int(const char* name, int dflt) {
  const char* s = std::getenv(name);
  return s && *s ? std::atoi(s) : dflt;
}

}  // namespace

// =================================================================================================== Engine
Engine::Engine(int device, int dice, int faces, const rbl_params& params, int max_lanes)
    : device_(device), g_(dice, faces), p_(params), max_lanes_(max_lanes) {
  try {
    construct();
  } catch (...) {
    release_handles();  // ~Engine does not run for a throwing constructor: no leaked streams / events / pinned memory
    throw;
  }
}

void Engine::construct() {
  const int device = device_, max_lanes = max_lanes_;
  if (!p_.use_cfr && p_.dcfr) throw std::runtime_error("engine: dcfr needs use_cfr=1");
  if (p_.linear_update && p_.dcfr) throw std::runtime_error("engine: linear_update and dcfr are exclusive (subgame_solving.cc:533)");
  if (p_.max_depth < 0) throw std::runtime_error("engine: max_depth must be >= 0");
  if (max_lanes < 1) throw std::runtime_error("engine: max_lanes must be >= 1");
  int ndev = 0;
  RBL_HIP_CHECK(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) throw std::runtime_error("engine: no such HIP device " + std::to_string(device));
  RBL_HIP_CHECK(hipSetDevice(device_));
  RBL_HIP_CHECK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
  RBL_HIP_CHECK(hipStreamCreateWithFlags(&stream2_, hipStreamNonBlocking));
  for (int i = 0; i < 2; ++i) RBL_HIP_CHECK(hipStreamCreateWithFlags(&stream_x_[i], hipStreamNonBlocking));
  // Lane parts on separate streams: with >= 16384 lanes the launches are long enough that one stream (net forward over
  // all lanes, then the CFR step over all lanes) loses only ~3 % to kernel tails and launch gaps, and every kernel then
  // runs with the GPU to itself (measured durations are the kernels' own); smaller batches gain 5-7 % from two
  // interleaved half-batches whose tails overlap (4096 lanes: 38.6 vs 36.3 M it/s, 8192: 42.4 vs 40.3).
  max_parts_ = std::min(4, std::max(1, env_int("RBL_PARTS", max_lanes_ >= 16384 ? 1 : 2)));
  RBL_HIP_CHECK(hipEventCreateWithFlags(&ev_ready_, hipEventDisableTiming));
  for (int i = 0; i < 3; ++i) RBL_HIP_CHECK(hipEventCreateWithFlags(&ev_join_[i], hipEventDisableTiming));
  split_min_lanes_ = env_int("RBL_SPLIT_MIN_LANES", 1024);

  {  // full trees grow as 2^A: refuse what cannot be tabulated (2 dice x 6 faces full tree = 33.5 M nodes per shape)
    double est = 0;
    const int dmax = std::min(p_.max_depth, g_.A);
    double level = 1;
    for (int d = 0; d <= dmax; ++d) {
      est += level;
      level *= std::max(1, g_.A - 1 - d);
      if (est > 5e6) break;
    }
    const double full = std::ldexp(1.0, g_.A);
    if (std::min(est, full) > 4e6)
      throw std::runtime_error("engine: tree of depth " + std::to_string(p_.max_depth) + " is too large to tabulate (" +
                               std::to_string((long long)std::min(est, full)) + "+ nodes)");
  }
  tabs_ = ShapeTables::build(g_, p_.max_depth);
  nmax_ = tabs_.max_N;
  emax_ = std::max(1, nmax_ - 1);

  d_shapes_.upload(tabs_.shapes, stream_);
  // the one-wavefront CFR kernel reads these (and sigma / values below) with unconditional strided loads: kWavePad
  // elements of slack behind each array keep its overruns inside the allocation
  std::vector<std::vector<int>> keep;  // staging copies must outlive the asynchronous uploads (synchronised below)
  keep.reserve(16);
  auto padded = [&keep](std::vector<int> v) -> const std::vector<int>& {
    v.resize(v.size() + kWavePad, 0);
    keep.push_back(std::move(v));
    return keep.back();
  };
  d_parent_.upload(padded(tabs_.parent), stream_);
  d_act_.upload(padded(tabs_.act), stream_);
  d_cb_.upload(padded(tabs_.cb), stream_);
  d_ce_.upload(padded(tabs_.ce), stream_);
  d_depth_.upload(padded(tabs_.depth), stream_);
  d_leaves_.upload(padded(tabs_.leaves), stream_);
  d_terms_.upload(padded(tabs_.terms), stream_);
  d_irank_.upload(padded(tabs_.irank), stream_);
  d_leaf_row_.upload(padded(tabs_.leaf_row), stream_);
  d_vrow_.upload(padded(tabs_.vrow), stream_);
  d_pack_.upload(padded(tabs_.pack), stream_);
  // Per-shape table blobs in the step kernels' LDS layouts + the per-shape LaneRec templates (cfr_kernels.h): a workgroup reads its
  // 64-byte record and then requests sigma, the net's rows and its tables in ONE burst.
  {
    std::vector<LaneRec> srec(tabs_.shapes.size());
    for (size_t si = 0; si < tabs_.shapes.size(); ++si) {
      const ShapeDev& s = tabs_.shapes[si];
      LaneRec& r = srec[si];
      r = LaneRec{};
      r.N = s.N;
      r.L = s.L;
      r.T = s.T;
      r.NI = s.NI;
      r.nlev = s.nlev;
      r.lo2 = s.lev_off[2];  // (= N for a two-level tree; level 1 always starts at node 1)
      r.act_iter = -1;
      r.node_off = s.node_off;
      r.term_off = s.term_off;
      r.shape = (int)si;
    }
    // cfr_wave_kernel: byte tables (every entry a node id, an action or -1: needs N <= 127) and the 15-bit parent offsets
    wave_tabs_ok_ = tabs_.max_N <= 127 && (size_t)tabs_.max_N * g_.H < 32768;
    std::vector<int8_t> blob;
    std::vector<unsigned short> epv;
    if (wave_tabs_ok_) {
      for (size_t si = 0; si < tabs_.shapes.size(); ++si) {
        const ShapeDev& s = tabs_.shapes[si];
        while (blob.size() % 4) blob.push_back(0);
        srec[si].tab_off = (int)blob.size();
        for (const std::vector<int>* tab : {&tabs_.parent, &tabs_.act, &tabs_.cb, &tabs_.ce})
          for (int n = 0; n < s.N; ++n) blob.push_back((int8_t)(*tab)[s.node_off + n]);
        for (int k = 0; k < s.L; ++k) blob.push_back((int8_t)tabs_.leaves[s.leaf_off + k]);
        for (int k = 0; k < s.T; ++k) blob.push_back((int8_t)tabs_.terms[s.term_off + k]);
        for (int f = 0; f < g_.faces; ++f)  // Game::num_matches (liars_dice.h:83-91), [faces][H]
          for (int h = 0; h < g_.H; ++h) blob.push_back((int8_t)g_.matches(h, f));
        // per edge element (c - 1) * H + h: the LDS offset parent(c) * H + h of its parent's value / reach / row-sum row;
        // bit 15: the child is a terminal (its value lives in its parent's terminal row)
        srec[si].epv_off = (int)epv.size();
        for (int c = 1; c < s.N; ++c)
          for (int h = 0; h < g_.H; ++h)
            epv.push_back((unsigned short)((tabs_.parent[s.node_off + c] * g_.H + h) | (tabs_.act[s.node_off + c] == g_.liar ? 0x8000 : 0)));
      }
    }
    blob.resize(blob.size() + kWavePad, 0);
    epv.resize(epv.size() + kWavePad, 0);
    d_wave_tabs_.upload(blob, stream_);
    d_wave_epv_.upload(epv, stream_);
    // cfr_flat_kernel (2 dice x 6 faces): int tables, the pseudo-leaf / value-row map and the match masks, 16-byte aligned per shape
    std::vector<int> fblob;
    if (cfr_flat_supported(g_.H, g_.A, g_.dice, g_.faces)) {
      for (size_t si = 0; si < tabs_.shapes.size(); ++si) {
        const ShapeDev& s = tabs_.shapes[si];
        while (fblob.size() % 4) fblob.push_back(0);
        srec[si].tab_off = (int)fblob.size();
        for (const std::vector<int>* tab : {&tabs_.parent, &tabs_.act, &tabs_.cb, &tabs_.ce, &tabs_.depth, &tabs_.pack})
          for (int n = 0; n < s.N; ++n) fblob.push_back((*tab)[s.node_off + n]);
        for (int n = 0; n < s.N; ++n) {  // lrow: net row of a pseudo-leaf, else -1 - (value row)
          const int lr = tabs_.leaf_row[s.node_off + n];
          fblob.push_back(lr >= 0 ? lr : -1 - tabs_.vrow[s.node_off + n]);
        }
        for (int k = 0; k < s.L; ++k) fblob.push_back(tabs_.leaves[s.leaf_off + k]);
        for (int k = 0; k < s.T; ++k) fblob.push_back(tabs_.terms[s.term_off + k]);
        if (fblob.size() % 2) fblob.push_back(0);
        for (int f = 0; f < g_.faces; ++f)  // bit h: hand h shows exactly 1 / exactly 2 of the face
          for (int k = 1; k <= 2; ++k) {
            unsigned long long bits = 0;
            for (int h = 0; h < g_.H && h < 64; ++h) bits |= (unsigned long long)(g_.matches(h, f) == k) << h;
            fblob.push_back((int)(unsigned)(bits & 0xffffffffu));
            fblob.push_back((int)(unsigned)(bits >> 32));
          }
      }
    }
    fblob.resize(fblob.size() + kWavePad, 0);
    d_flat_tabs_.upload(fblob, stream_);
    d_shape_rec_.upload(srec, stream_);
    d_lane_rec_.alloc((size_t)max_lanes_);
    RBL_HIP_CHECK(hipStreamSynchronize(stream_));  // the staging vectors go out of scope
  }
  std::vector<int8_t> m((size_t)g_.faces * g_.H + kWavePad);
  for (int f = 0; f < g_.faces; ++f)
    for (int h = 0; h < g_.H; ++h) m[(size_t)f * g_.H + h] = (int8_t)g_.matches(h, f);
  d_matches_.upload(m, stream_);
  std::vector<int> epar(2 * tabs_.shapes.size(), 0);  // edges by the depth parity of their parent (roofline accounting)
  for (size_t si = 0; si < tabs_.shapes.size(); ++si) {
    const ShapeDev& s = tabs_.shapes[si];
    for (int n = 1; n < s.N; ++n) ++epar[2 * si + (tabs_.depth[s.node_off + tabs_.parent[s.node_off + n]] & 1)];
  }
  d_shape_epar_.upload(epar, stream_);

  const size_t L = (size_t)max_lanes_;
  const size_t eh = (size_t)emax_ * g_.H;
  d_lane_shape_.alloc(L);
  d_lane_player_.alloc(L);
  d_lane_row_.alloc(L);
  d_lane_act_.alloc(L);
  d_beliefs_.alloc(L * 2 * g_.H);
  d_sigma_.alloc(L * eh + kWavePad);
  d_regrets_.alloc(L * eh);
  d_sums_.alloc(L * eh);
  d_snapshot_.alloc(L * eh);
  d_root_mean_.alloc(L * 2 * g_.H);
  const size_t max_rows = std::max<size_t>(1, L * tabs_.max_L);
  d_queries_.alloc(max_rows * g_.query_size());
  d_values_.alloc(max_rows * g_.H + kWavePad);
  RBL_HIP_CHECK(hipMemsetAsync(d_values_.p, 0, max_rows * g_.H * sizeof(float), stream_));
  values_zeroed_ = true;

  // working set per lane: LDS if it fits in one CU's 160 KB with room for >= 2 workgroups, else global scratch
  work_stride_ = cfr_work_reals(nmax_, g_.H, tabs_.max_L, tabs_.max_T, g_.dice, g_.faces);
  lds_bytes_ = work_stride_ * sizeof(double);
  const size_t lds_cap = (size_t)env_int("RBL_CFR_LDS_CAP", 80 * 1024);
  use_lds_ = lds_bytes_ <= lds_cap;
  if (!use_lds_) d_scratch_.alloc(L * work_stride_);
  const int nh = nmax_ * g_.H;
  block_ = nh <= 128 ? 64 : (nh <= 384 ? 128 : 256);  // measured on MI355X: 1dx6f (546 pairs) 256 > 128 > 64
  block_ = env_int("RBL_CFR_BLOCK", block_);
  // CFR::step proper runs on the row-per-thread kernel when the game has an instantiation and every shape of this
  // engine fits its LDS layout (including the query rows staged over the dead val/reg/leaf-value bytes)
  rows_ok_ = use_lds_ && env_int("RBL_CFR_ROWS", 1) && cfr_rows_supported(g_.H, g_.A, g_.dice, g_.faces);
  rows_lds_bytes_ = 0;
  for (const ShapeDev& s : tabs_.shapes) {
    rows_lds_bytes_ = std::max(rows_lds_bytes_, cfr_rows_lds_bytes(s.N, s.NI, g_.H, s.L, g_.faces));
    const size_t stage = (size_t)(2 * s.N - 1) * g_.H * 8;
    if ((size_t)s.L * g_.query_size() * 4 > stage) rows_ok_ = false;
  }
  if (rows_lds_bytes_ > std::min<size_t>(lds_cap, 64 * 1024)) rows_ok_ = false;
  {  // one wavefront per lane, element-parallel (cfr_wave_kernel.hip): the default where it has an instantiation
    int max_eh = 0, max_lh = 0;
    wave_lds_bytes_ = 0;
    for (const ShapeDev& s : tabs_.shapes) {
      max_eh = std::max(max_eh, (s.N - 1) * g_.H);
      max_lh = std::max(max_lh, s.L * g_.H);
      if (s.nlev >= 2)
        wave_lds_bytes_ = std::max(wave_lds_bytes_, cfr_wave_lds_bytes(s.N, s.NI, g_.H, s.L, s.T, g_.faces,
                                                                        s.lev_off[s.nlev - 1], s.lev_off[s.nlev - 2]));
    }
    // the kernel's tree model: depth <= 2 and the nodes with children first in BFS order (reach-row rank = node id)
    bool prefix_ok = true;
    for (const ShapeDev& s : tabs_.shapes) {
      if (s.nlev > 3 || s.nlev < 2) prefix_ok = false;  // (nlev < 2: the root itself is a pseudo-leaf, max_depth = 0)
      for (int n = 0; n < s.N; ++n)
        if (tabs_.irank[s.node_off + n] != (n < s.NI ? n : -1)) prefix_ok = false;
      // its value layout: pseudo-leaves only on the deepest level; a deepest-level terminal is the LAST child of its parent
      const int lo_d = s.lev_off[std::max(s.nlev - 1, 0)];
      for (int n = 0; n < s.N; ++n) {
        if (tabs_.leaf_row[s.node_off + n] >= 0 && n < lo_d) prefix_ok = false;
        if (n >= lo_d && tabs_.act[s.node_off + n] == g_.liar && n + 1 != tabs_.ce[s.node_off + tabs_.parent[s.node_off + n]])
          prefix_ok = false;
      }
    }
    wave_lds_bytes_ += (size_t)std::max(0, env_int("RBL_WAVE_LDS_EXTRA", 0));  // developer aid: occupancy experiments
    wave_ok_ = use_lds_ && wave_tabs_ok_ && env_int("RBL_CFR_WAVE", 1) && wave_lds_bytes_ <= 64 * 1024 && prefix_ok &&
               cfr_wave_supported(g_.H, g_.A, g_.dice, g_.faces, max_eh, max_lh, nmax_);
  }
  // big games (2 dice x 6 faces): the row kernel with the strategy arrays in place in global memory
  // ... or, the default, the element-parallel kernel with sigma resident in LDS (cfr_flat_kernel.hip); it needs a root with
  // children in every shape (max_depth >= 1) and 147 KB of LDS at the root of a depth-2 subgame
  flat_ok_ = env_int("RBL_CFR_FLAT", 1) != 0 && cfr_flat_supported(g_.H, g_.A, g_.dice, g_.faces);
  for (const ShapeDev& s : tabs_.shapes)
    if (tabs_.cb[s.node_off] == tabs_.ce[s.node_off] || cfr_flat_lds_bytes(s.N, s.NI, g_.H, s.L, s.T, g_.faces) > 160 * 1024 ||
        s.N > 511 || s.NI > 62 || s.N - s.L > 255 ||  // the kernel's packed table words: 9-bit node ids, 8-bit row indices, 6-bit own reach row (pk_ir)
        s.nlev > 3)                                    // leaves take their reach from the root row: subgames of depth <= 2
      flat_ok_ = false;
  rows_global_lds_ = 0;
  for (const ShapeDev& s : tabs_.shapes) rows_global_lds_ = std::max(rows_global_lds_, gs_lds_bytes(s));
  rows_global_ok_ = !use_lds_ && env_int("RBL_CFR_ROWS", 1) && cfr_rows_global_supported(g_.H, g_.A, g_.dice, g_.faces) &&
                    rows_global_lds_ <= 160 * 1024;
  flat_ok_ = flat_ok_ && rows_global_ok_;
  {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device_) == hipSuccess && v > 0) n_cus_ = v;
    net_grid_env_ = env_int("RBL_NET_GRID", -1);
  }
  flat_threads_ = std::min(1024, std::max(64, env_int("RBL_CFR_FLAT_THREADS", 1024) / 64 * 64));
  use_order_ = rows_global_ok_ && env_int("RBL_GS_SORT", 1) != 0;
  if (use_order_) d_lane_order_.alloc((size_t)max_lanes_);
  for (auto& row : seg_lds_)
    for (size_t& v : row) v = rows_global_lds_;
  for (auto& row : seg_threads_)
    for (int& v : row) v = flat_threads_;
  rows_fit_ = env_int("RBL_CFR_ROWS_FIT", 1) != 0;
  rows_block_ = std::min(128, std::max(64, env_int("RBL_CFR_ROWS_BLOCK", 128)));  // the kernel is built for <= 128 threads
  cfr_dbg_ = env_int("RBL_CFR_DBG", 0) != 0;
  ext_timing_[0] = ext_timing_[1] = env_int("RBL_TIMING_EXT", 1) != 0;  // 0: bracket with recorded events (includes launch gaps)
  if (cfr_dbg_) {
    d_dbg_.alloc(L * 16);
    RBL_HIP_CHECK(hipMemset(d_dbg_.p, 0, L * 16 * sizeof(long long)));
  }
  RBL_HIP_CHECK(hipStreamSynchronize(stream_));
}

Engine::~Engine() { release_handles(); }

void Engine::release_handles() {
  if (!stream_ && !stream2_ && !stream_x_[0] && !stream_x_[1] && !ev_ready_ && !h_pinned_ && ev_pool_.empty() &&
      !ev_join_[0])
    return;
  (void)hipSetDevice(device_);
  if (stream_) (void)hipStreamSynchronize(stream_);
  if (stream2_) (void)hipStreamSynchronize(stream2_);
  for (int i = 0; i < 2; ++i)
    if (stream_x_[i]) {
      (void)hipStreamSynchronize(stream_x_[i]);
      (void)hipStreamDestroy(stream_x_[i]);
    }
  for (auto e : ev_pool_) (void)hipEventDestroy(e);
  if (h_pinned_) (void)hipHostFree(h_pinned_);
  if (ev_ready_) (void)hipEventDestroy(ev_ready_);
  for (int i = 0; i < 3; ++i) {
    if (ev_join_[i]) (void)hipEventDestroy(ev_join_[i]);
    ev_join_[i] = nullptr;
  }
  if (stream_) (void)hipStreamDestroy(stream_);
  if (stream2_) (void)hipStreamDestroy(stream2_);
  for (int i = 0; i < 2; ++i) stream_x_[i] = nullptr;
  ev_pool_.clear();
  h_pinned_ = nullptr;
  ev_ready_ = nullptr;
  stream_ = stream2_ = nullptr;
}

void Engine::check_lane(int lane) const {
  if (lane < 0 || lane >= B_) throw std::runtime_error("engine: lane out of range");
}

// ---------------------------------------------------------------------------------------------- value net
// A net can be exchanged in the middle of a solve (ModelLocker::updateModel, model_locker.h:69-79; tests that reset() before
// they set the net).  The query layout follows the net kind (split rows for the fused MLP, canonical rows otherwise), so the
// layout the NEW net reads is rebuilt here from the one the live solve has been writing.  Callers hold net_mutex_.
void Engine::leave_split_layout() {
  if (!qsplit_) return;
  qsplit_ = false;
  if (rows_ <= 0 || tabs_.max_L == 0 || !q_canon_stale_) return;
  sync();  // the steps since the last init wrote dynamic rows only: the canonical matrix is rebuilt from them
  RBL_HIP_CHECK(hipSetDevice(device_));
  launch_unsplit_queries(d_queries_.p, g_.A, g_.H, d_qdyn_.p, q_ds_, d_qstat_.p, q_ss_, rows_, stream_);
  RBL_HIP_CHECK(hipGetLastError());
  RBL_HIP_CHECK(hipStreamSynchronize(stream_));
  q_canon_stale_ = false;
}

void Engine::enter_split_layout() {  // canonical rows of a live solve -> (dynamic | static) rows; streams are drained
  if (rows_ <= 0 || tabs_.max_L == 0) return;
  for (int part = 0; part < n_parts_; ++part) split_part_queries(part, stream_);
  RBL_HIP_CHECK(hipStreamSynchronize(stream_));
}

void Engine::set_net_zero() {
  std::lock_guard<std::mutex> net_lock(net_mutex_);
  RBL_HIP_CHECK(hipSetDevice(device_));
  leave_split_layout();
  net_mode_ = NetMode::kZero;
  RBL_HIP_CHECK(hipMemsetAsync(d_values_.p, 0, d_values_.n * sizeof(float), stream_));
  values_zeroed_ = true;
}

void Engine::set_net_synthetic() {
  std::lock_guard<std::mutex> net_lock(net_mutex_);
  leave_split_layout();
  net_mode_ = NetMode::kSynthetic;
  values_zeroed_ = false;
}

void Engine::set_net_callback(rbl_net_fn fn, void* user, bool host_buffers) {
  std::lock_guard<std::mutex> net_lock(net_mutex_);
  if (!fn) throw std::runtime_error("set_net_callback: null function");
  leave_split_layout();
  net_mode_ = NetMode::kCallback;
  cb_fn_ = fn;
  cb_user_ = user;
  cb_host_ = host_buffers;
  values_zeroed_ = false;
}

void Engine::set_net_precision(int mode) {
  if (mode < 0 || mode > 2) throw std::runtime_error("set_net_precision: mode must be 0 (f32 parity), 1 or 2 (half_inference)");
  std::lock_guard<std::mutex> net_lock(net_mutex_);
  net_precision_ = mode;
}

void Engine::set_net_mlp(const rbl_mlp_weights& w) {
  std::lock_guard<std::mutex> net_lock(net_mutex_);
  RBL_HIP_CHECK(hipSetDevice(device_));
  if (w.n_in != g_.query_size())
    throw std::runtime_error("set_net_mlp: net input size " + std::to_string(w.n_in) + " != query size " +
                             std::to_string(g_.query_size()));
  if (w.n_out != g_.H)
    throw std::runtime_error("set_net_mlp: net output size " + std::to_string(w.n_out) + " != num_hands " +
                             std::to_string(g_.H));
  const int tile = env_int("RBL_MLP_TILE", 5);  // 5 = persistent register-resident kernel, 3 = feature-split fallback
  // Split query layout (cfr_kernels.h): the one-wavefront CFR kernel writes only what changes per iteration, as contiguous
  // rows; the resident forward reads (dynamic row | static row) as its input, layer 0 packed for that column order.
  const int ds = (1 + 2 * g_.H + 3) & ~3, ss = (1 + g_.A + 3) & ~3;
  const bool split = (wave_ok_ || (flat_ok_ && rows_global_ok_)) && p_.use_cfr && tile == 5 && env_int("RBL_QSPLIT", 1) != 0 &&
                     mlp_resident_supported(w.n_layers, ds + ss, w.n_hidden, w.n_out);
  std::vector<float> w0v;
  std::vector<const float*> wv(w.w, w.w + w.n_layers);
  int n_in_pack = w.n_in;
  if (split) {
    n_in_pack = ds + ss;
    w0v.assign((size_t)w.n_hidden * n_in_pack, 0.f);
    for (int f = 0; f < w.n_hidden; ++f) {
      const float* src = w.w[0] + (size_t)f * w.n_in;
      float* dst = w0v.data() + (size_t)f * n_in_pack;
      dst[0] = src[1];                                                         // traverser flag
      for (int j = 0; j < 2 * g_.H; ++j) dst[1 + j] = src[2 + g_.A + j];       // the two reach vectors
      dst[ds] = src[0];                                                        // player to move
      for (int a2 = 0; a2 < g_.A; ++a2) dst[ds + 1 + a2] = src[2 + a2];        // one-hot last bid
    }
    wv[0] = w0v.data();
  }
  MlpPacked pk = pack_mlp(w.n_layers, n_in_pack, w.n_hidden, w.n_out, w.use_layer_norm, wv.data(), w.b, w.ln_w, w.ln_b,
                          w.w_out, w.b_out, tile);
  // every refusal happens HERE, before the engine is touched: a rejected net leaves the previous one (blob, MlpDev, query
  // layout, net mode) exactly as it was, so the caller can retry with rbl_engine_set_net_precision(e, 0)
  if (net_precision_ != 0 && (pk.tile != 5 || !w.use_layer_norm))
    throw std::runtime_error("set_net_mlp: the half_inference modes need the register-resident kernel and a LayerNorm net "
                             "(hidden layers of 256; rbl_engine_set_net_precision(e, 0) for everything else)");
  // weight refresh (ModelLocker::updateModel, model_locker.h:69-79) happens between launches: no new forward can be
  // enqueued while we hold net_mutex_, and the ones already enqueued on either stream are drained first
  sync();
  d_mlp_blob_.upload(pk.blob, stream_);
  RBL_HIP_CHECK(hipStreamSynchronize(stream_));
  mlp_ = MlpDev{};
  mlp_.n_layers = w.n_layers;
  mlp_.n_in = n_in_pack;
  mlp_n_in_true_ = w.n_in;
  const bool was_split = qsplit_;
  if (was_split && !(split && pk.tile == 5)) leave_split_layout();
  qsplit_ = split && pk.tile == 5;
  q_ds_ = ds;
  q_ss_ = ss;
  if (qsplit_) {
    const size_t max_rows = std::max<size_t>(1, (size_t)max_lanes_ * tabs_.max_L);
    if (d_qdyn_.n < max_rows * ds) d_qdyn_.alloc(max_rows * ds);
    if (d_qstat_.n < max_rows * ss) d_qstat_.alloc(max_rows * ss);
    mlp_.q_stat = d_qstat_.p;
    mlp_.q_dyn_stride = ds;
    mlp_.q_stat_stride = ss;
    if (!was_split) enter_split_layout();  // (a weight refresh of a split net keeps both the rows and q_canon_stale_)
  }
  mlp_.n_hidden = w.n_hidden;
  mlp_.n_out = w.n_out;
  mlp_.products = 3 - net_precision_;
  mlp_.use_ln = env_int("RBL_MLP_DEBUG", 0) == 1 ? 2 : w.use_layer_norm;
  mlp_.tile = pk.tile;
  if (env_int("RBL_NET_DBG", 0)) {
    if (!d_ndbg_.p) d_ndbg_.alloc(1024 * 16);
    mlp_.dbg = d_ndbg_.p;
  }
  mlp_.stagger = env_int("RBL_MLP_STAGGER", 0);
  for (size_t i = 0; i < pk.inv_scale.size() && i < 8; ++i) mlp_.inv_scale[i] = pk.inv_scale[i];
  mlp_.tape_chunks = pk.tape_chunks;
  mlp_.l0_chunks = pk.l0_chunks;
  mlp_.k0_steps = pk.k0_steps;
  mlp_.out_tiles = pk.out_tiles;
  mlp_.ln_eps = w.ln_eps > 0 ? w.ln_eps : 1e-5f;
  const float* base = d_mlp_blob_.p;
  mlp_.tape = base;
  mlp_.w0 = base + pk.off_w0;
  mlp_.wh = base + pk.off_wh;
  mlp_.wo = base + pk.off_wo;
  mlp_.bias = base + pk.off_bias;
  mlp_.ln_w = base + pk.off_lnw;
  mlp_.ln_b = base + pk.off_lnb;
  mlp_.b_out = base + pk.off_bout;
  net_mode_ = NetMode::kMlp;
  values_zeroed_ = false;
}

void Engine::net_forward_dev(const float* q_dev, int64_t rows, float* out_dev, hipStream_t st, const long long* range) {
  if (rows <= 0) return;
  if (range && net_mode_ == NetMode::kCallback)
    throw std::runtime_error("net: a callback net needs host-side row counts (device-resident epochs are not available)");
  if (!st) st = stream_;
  const int Q = g_.query_size(), H = g_.H;
  switch (net_mode_) {
    case NetMode::kZero:
      if (!range) RBL_HIP_CHECK(hipMemsetAsync(out_dev, 0, (size_t)rows * H * sizeof(float), st));
      break;
    case NetMode::kSynthetic:
      launch_synthetic_net(q_dev, rows, Q, out_dev, H, g_.A, st, range);
      break;
    case NetMode::kMlp:
      if (qsplit_) {  // canonical rows from a caller (rbl_net_forward): through temporaries in the layout the packed net reads
        if (range) throw std::runtime_error("net: canonical queries with a device-side range are not used in split mode");
        if (d_tmp_dyn_.n < (size_t)rows * q_ds_) d_tmp_dyn_.alloc((size_t)rows * q_ds_);
        if (d_tmp_stat_.n < (size_t)rows * q_ss_) d_tmp_stat_.alloc((size_t)rows * q_ss_);
        launch_split_queries(q_dev, g_.A, g_.H, d_tmp_dyn_.p, q_ds_, d_tmp_stat_.p, q_ss_, rows, st);
        MlpDev m2 = mlp_;
        m2.q_stat = d_tmp_stat_.p;
        if (net_grid_env_ >= 0) m2.grid_cap = net_grid_env_;  // developer override (RBL_NET_GRID): standalone forwards too
        launch_mlp_forward(m2, d_tmp_dyn_.p, rows, out_dev, st, nullptr);
      } else if (net_grid_env_ >= 0) {
        MlpDev m2 = mlp_;
        m2.grid_cap = net_grid_env_;
        launch_mlp_forward(m2, q_dev, rows, out_dev, st, range);
      } else {
        launch_mlp_forward(mlp_, q_dev, rows, out_dev, st, range);
      }
      break;
    case NetMode::kCallback:
      if (cb_host_) {
        h_q_.resize((size_t)rows * Q);
        h_v_.assign((size_t)rows * H, 0.f);
        RBL_HIP_CHECK(hipMemcpyAsync(h_q_.data(), q_dev, h_q_.size() * sizeof(float), hipMemcpyDeviceToHost, st));
        RBL_HIP_CHECK(hipStreamSynchronize(st));
        cb_fn_(cb_user_, h_q_.data(), rows, Q, h_v_.data(), H, nullptr);
        RBL_HIP_CHECK(hipMemcpyAsync(out_dev, h_v_.data(), h_v_.size() * sizeof(float), hipMemcpyHostToDevice, st));
        RBL_HIP_CHECK(hipStreamSynchronize(st));
      } else {  // device pointers: the engine stream is drained first; the callee returns with `out` complete
        RBL_HIP_CHECK(hipStreamSynchronize(st));
        cb_fn_(cb_user_, q_dev, rows, Q, out_dev, H, (void*)st);
      }
      break;
  }
  RBL_HIP_CHECK(hipGetLastError());
}

void Engine::net_forward_host(const float* q, int64_t rows, float* out) {
  sync();
  RBL_HIP_CHECK(hipSetDevice(device_));
  if (rows <= 0) return;
  const size_t nq = (size_t)rows * g_.query_size(), no = (size_t)rows * g_.H;
  if (d_tmp_q_.n < nq) d_tmp_q_.alloc(nq);
  if (d_tmp_o_.n < no) d_tmp_o_.alloc(no);
  RBL_HIP_CHECK(hipMemcpyAsync(d_tmp_q_.p, q, nq * sizeof(float), hipMemcpyHostToDevice, stream_));
  net_forward_dev(d_tmp_q_.p, rows, d_tmp_o_.p);
  RBL_HIP_CHECK(hipMemcpyAsync(out, d_tmp_o_.p, no * sizeof(float), hipMemcpyDeviceToHost, stream_));
  RBL_HIP_CHECK(hipStreamSynchronize(stream_));
}

// ---------------------------------------------------------------------------------------------- timing
void Engine::timing(int stride) {
  timing_stride_ = stride < 0 ? 0 : stride;
  timing_ = timing_stride_ > 0;
}

void Engine::time_begin(int kind, hipStream_t st) {
  if (!timed_now()) return;
  while (ev_pool_.size() < ev_used_ + 2) {
    hipEvent_t e;
    RBL_HIP_CHECK(hipEventCreate(&e));
    ev_pool_.push_back(e);
  }
  ext_armed_ = ext_timing_[kind];
  if (ext_armed_) {  // the launcher binds the two events to its dispatch packet (launch_timing.h)
    tl_launch_timing.start = ev_pool_[ev_used_];
    tl_launch_timing.stop = ev_pool_[ev_used_ + 1];
    tl_launch_timing.used = 0;
  } else {
    RBL_HIP_CHECK(hipEventRecord(ev_pool_[ev_used_], st));
  }
  pending_.push_back(Pending{kind, ev_used_, ev_used_ + 1});
  ev_used_ += 2;
  sample_open_ = true;
}

// Unwinding between time_begin and time_end (a failed launch: RBL_HIP_CHECK, launch_cfr_flat's LDS attribute error): the open
// sample's event pair was never (both) recorded, so it must not reach stats()' hipEventElapsedTime.  The two events stay
// consumed until the next stats() -- they may already be bound to a dispatch -- only the sample is forgotten.
void Engine::time_abort() {
  tl_launch_timing = LaunchTimingSlot{};
  if (!sample_open_) return;
  sample_open_ = false;
  ext_armed_ = false;
  if (!pending_.empty()) pending_.pop_back();
}

bool Engine::time_end(int kind, hipStream_t st) {
  if (!timed_now()) return false;
  if (ext_armed_) {
    const int used = tl_launch_timing.used;
    tl_launch_timing = LaunchTimingSlot{};
    ext_armed_ = false;
    sample_open_ = false;
    if (used == 1) return true;
    // none or several kernels took the slot (a launcher without support, segmented launches): this sample has no valid
    // event pair; from now on this kind is bracketed with recorded events.  The pair is NOT handed out again before the next
    // stats(): it may be bound to several dispatches that are still in flight.
    ext_timing_[kind] = false;
    pending_.pop_back();
    return false;
  }
  RBL_HIP_CHECK(hipEventRecord(ev_pool_[pending_.back().e1], st));
  sample_open_ = false;
  return true;
}

void Engine::stats(rbl_kernel_stats* out, bool reset) {
  RBL_HIP_CHECK(hipSetDevice(device_));
  sync();
  for (const auto& p : pending_) {
    float ms = 0;
    RBL_HIP_CHECK(hipEventElapsedTime(&ms, ev_pool_[p.e0], ev_pool_[p.e1]));
    (p.kind == 0 ? stats_.cfr_ms : stats_.net_ms) += ms;
  }
  pending_.clear();
  ev_used_ = 0;
  stats_.cfr_kernel = last_cfr_kernel_;
  stats_.net_kernel = net_mode_ == NetMode::kMlp ? mlp_.tile : 0;
  stats_.n_streams = n_parts_;
  stats_.net_products = net_mode_ == NetMode::kMlp ? mlp_.products : 0;
  if (out) *out = stats_;
  if (reset) stats_ = rbl_kernel_stats{};
}

// ---------------------------------------------------------------------------------------------- solver batch
void Engine::reset(int B, const int32_t* root_last_bid, const int32_t* root_player, const double* beliefs,
                   const int32_t* act_iteration) {
  RBL_HIP_CHECK(hipSetDevice(device_));
  sync();
  if (B < 1 || B > max_lanes_) throw std::runtime_error("reset: B must be in [1, max_lanes]");
  const int H = g_.H, Q = g_.query_size();
  h_shape_.resize(B);
  h_player_.resize(B);
  h_row_.resize(B);
  h_bid_.resize(B);
  h_act_.assign(B, -1);
  h_beliefs_.assign(beliefs, beliefs + (size_t)B * 2 * H);
  int64_t rows = 0;
  step_bytes_[0] = step_bytes_[1] = 0;
  for (int b = 0; b < B; ++b) {
    const int rb = root_last_bid[b];
    if (rb < -1 || rb >= g_.A - 1) throw std::runtime_error("reset: root_last_bid out of range (terminal or invalid state)");
    if (root_player[b] != 0 && root_player[b] != 1) throw std::runtime_error("reset: root_player must be 0 or 1");
    h_bid_[b] = rb;
    h_shape_[b] = rb + 1;
    h_player_[b] = root_player[b];
    h_row_[b] = (int)rows;
    const ShapeDev& s = tabs_.shapes[rb + 1];
    rows += s.L;
    if (act_iteration) {
      if (act_iteration[b] < 0) throw std::runtime_error("reset: act_iteration must be >= 0");
      h_act_[b] = act_iteration[b];
    }
    // algorithmic bytes of one step: read sigma over E, RMW regrets and sums + write sigma over E_t,
    // write L queries, read L value rows
    int e_par[2] = {0, 0};
    for (int n = 1; n < s.N; ++n) ++e_par[tabs_.depth[s.node_off + tabs_.parent[s.node_off + n]] & 1];
    for (int t = 0; t < 2; ++t) {
      const int et = e_par[(root_player[b] == t) ? 0 : 1];
      step_bytes_[t] += 8.0 * H * ((s.N - 1) + 5.0 * et) + 4.0 * s.L * (Q + H);
    }
  }
  has_act_ = act_iteration != nullptr;
  info_dev_ = nullptr;
  mirror_valid_ = true;
  B_ = B;
  rows_ = rows;
  iter_ = 0;
  num_steps_[0] = num_steps_[1] = 0;
  d_lane_shape_.upload(h_shape_, stream_);
  d_lane_player_.upload(h_player_, stream_);
  d_lane_row_.upload(h_row_, stream_);
  d_lane_act_.upload(h_act_, stream_);
  d_beliefs_.upload(h_beliefs_, stream_);
  if (root_dedup_)  // a host-described batch solves every lane (the flags belong to device-resident self-play epochs)
    RBL_HIP_CHECK(hipMemsetAsync(d_lane_skip_.p, 0, (size_t)max_lanes_ * sizeof(int), stream_));
  // lane parts (parts_for; one part from 16384 lanes on, see the constructor): independent lane sets on their
  // own streams, rows of a part are contiguous
  n_parts_ = parts_for(B);
  part_lanes(B, part_lane_);
  if (use_order_) {  // lanes of each part by tree size, largest first (= shape id ascending), ties by lane index
    std::vector<int> order(B);
    int seg_shape[4][kSpSegs] = {};
    for (int pt = 0; pt < n_parts_; ++pt) {
      const int l0 = part_lane_[pt], l1 = part_lane_[pt + 1];
      for (int i = l0; i < l1; ++i) order[i] = i;
      std::stable_sort(order.begin() + l0, order.begin() + l1, [&](int x, int y) { return h_shape_[x] < h_shape_[y]; });
      const int ns = sp_segments(l1 - l0);
      for (int k = 0; k < ns && l1 > l0; ++k) {
        const int first = l0 + (int)((long long)(l1 - l0) * k / ns);
        seg_shape[pt][k] = h_shape_[order[std::min(first, l1 - 1)]];
      }
    }
    d_lane_order_.upload(order, stream_);
    RBL_HIP_CHECK(hipStreamSynchronize(stream_));  // `order` goes out of scope
    set_segments(seg_shape);
  }
  for (int pt = 0; pt <= n_parts_; ++pt) part_row_[pt] = pt == n_parts_ ? rows : h_row_[part_lane_[pt]];
  for (int pt = 0; pt < 4; ++pt) {
    part_bytes_[pt][0] = part_bytes_[pt][1] = 0;
    part_rows_lds_[pt] = 0;
    part_rows_block_[pt] = 64;
  }
  for (int b = 0; b < B; ++b) {
    const ShapeDev& s = tabs_.shapes[h_shape_[b]];
    int e_par[2] = {0, 0};
    for (int n = 1; n < s.N; ++n) ++e_par[tabs_.depth[s.node_off + tabs_.parent[s.node_off + n]] & 1];
    int part = 0;
    while (part + 1 < n_parts_ && b >= part_lane_[part + 1]) ++part;
    for (int t = 0; t < 2; ++t) {
      const int et = e_par[(h_player_[b] == t) ? 0 : 1];
      part_bytes_[part][t] += 8.0 * H * ((s.N - 1) + 5.0 * et) + 4.0 * s.L * (Q + H);
    }
    // the row kernel's launch shape follows the largest tree of the part: LDS request (= lanes per CU) and 64 threads when
    // every tree has at most 64 rows.  Uniformly small batches (deep levels of recursive solving, late-game lanes) run
    // 25-30 % faster; a part that holds a root subgame keeps the full shape.  (Sorting the lanes by tree size so that one
    // part gets the small trees was tried: no gain, the kernel is instruction-issue-bound at 8 lanes per CU.)
    if (rows_fit_) {
      part_rows_lds_[part] = std::max(part_rows_lds_[part], cfr_rows_lds_bytes(s.N, s.NI, g_.H, s.L, g_.faces));
      if (s.N > 64) part_rows_block_[part] = rows_block_;
    } else {
      part_rows_lds_[part] = rows_lds_bytes_;
      part_rows_block_[part] = rows_block_;
    }
  }
  build_lane_records();
  RBL_HIP_CHECK(hipEventRecord(ev_ready_, stream_));
  for (int pt = 1; pt < n_parts_; ++pt) RBL_HIP_CHECK(hipStreamWaitEvent(part_stream(pt), ev_ready_, 0));
  num_strategies_ = 0;
  launch(kModeInit, 0, 0, 0, 0, 1, 1, 1);
  pending_trav_ = 0;
}

// Two small lane parts on two streams: the persistent net kernel of one part owns every CU it runs on, so the other part's CFR
// step is time-sliced against it.  With three quarters of the CUs for the net kernel the CFR kernel has a place to run beside it
// (measured, round 5, 4 096 lanes = 2 x 2 048: 1 die x 6 faces 40.5 -> 41.9 M it/s, 1 die x 4 faces 78.8 -> 88.3 M; other caps:
// 160: 41.7 / 87.3, 208: 40.4 / 77.6; at 2 x 4 096 lanes it costs 1-3 %, and the 2 dice x 6 faces kernels, whose root lanes
// need whole CUs, lose with any cap: profiles/r05_net_grid_cap_sweep.txt).  RBL_NET_GRID=n overrides (0 = one per CU).
int Engine::net_grid_cap(int B) const {
  if (net_grid_env_ >= 0) return net_grid_env_;
  const int n = parts_for(B);
  return n >= 2 && wave_ok_ && B / n <= 2048 ? n_cus_ * 3 / 4 : 0;
}

int Engine::parts_for(int B) const {
  int n = 1;
  while (n < max_parts_ && B >= (n + 1) * split_min_lanes_) ++n;
  return n;
}

void Engine::part_lanes(int B, int* part_lane) const {
  const int n = parts_for(B);
  for (int pt = 0; pt <= kSpMaxParts; ++pt) part_lane[pt] = pt <= n ? (int)((int64_t)B * pt / n) : B;
}

// Device-resident epoch: the lane descriptors (shape, root player, first net row, act_iteration, root beliefs) and the
// parts' row boundaries were written by sp_begin / sp_scan, already enqueued on stream_.  Nothing is known on the host
// but the lane count, so launches take the largest shape's configuration and the net kernels read their row range from
// `info_dev`.  No host synchronisation: the previous epoch's kernels are ordered before this one by stream_.
void Engine::begin_epoch_device(int B, const SpEpochInfo* info_dev) {
  RBL_HIP_CHECK(hipSetDevice(device_));
  if (B < 1 || B > max_lanes_) throw std::runtime_error("begin_epoch_device: B must be in [1, max_lanes]");
  if (!device_epochs_supported())
    throw std::runtime_error("begin_epoch_device: a callback net needs host-side row counts");
  info_dev_ = info_dev;
  mirror_valid_ = false;
  has_act_ = true;
  B_ = B;
  rows_ = (int64_t)B * tabs_.max_L;  // upper bound until end_epoch_device
  iter_ = 0;
  num_steps_[0] = num_steps_[1] = 0;
  num_strategies_ = 0;
  n_parts_ = parts_for(B);
  part_lanes(B, part_lane_);
  for (int pt = 0; pt < 4; ++pt) {
    part_bytes_[pt][0] = part_bytes_[pt][1] = 0;
    part_rows_lds_[pt] = rows_lds_bytes_;
    part_rows_block_[pt] = rows_block_;
  }
  for (int pt = 0; pt <= n_parts_; ++pt) part_row_[pt] = (int64_t)part_lane_[pt] * tabs_.max_L;  // bounds only
  if (use_order_) {  // the one thing the host does learn before the epoch: the head shape of each launch segment (sp_order)
    int seg_shape[kSpMaxParts][kSpSegs];
    RBL_HIP_CHECK(hipMemcpyAsync(seg_shape, info_dev->seg_shape, sizeof(seg_shape), hipMemcpyDeviceToHost, stream_));
    RBL_HIP_CHECK(hipStreamSynchronize(stream_));
    set_segments(seg_shape);
  }
  build_lane_records();
  RBL_HIP_CHECK(hipEventRecord(ev_ready_, stream_));
  for (int pt = 1; pt < n_parts_; ++pt) RBL_HIP_CHECK(hipStreamWaitEvent(part_stream(pt), ev_ready_, 0));
  launch(kModeInit, 0, 0, 0, 0, 1, 1, 1);
  pending_trav_ = 0;
}

// The step kernels' per-slot records (cfr_kernels.h: LaneRec) from the epoch's lane descriptors -- uploaded by reset() or written
// by sp_begin / sp_scan / sp_order, in both cases already enqueued on stream_ -- once per epoch, off the host.
void Engine::build_lane_records() {
  if (!wave_ok_ && !flat_ok_) return;
  launch_lane_rec(d_shape_rec_.p, d_lane_shape_.p, d_lane_player_.p, d_lane_row_.p, d_lane_act_.p,
                  use_order_ ? d_lane_order_.p : nullptr, root_dedup_ ? d_lane_skip_.p : nullptr, B_, d_lane_rec_.p, stream_);
  RBL_HIP_CHECK(hipGetLastError());
}

// Root de-duplication (selfplay_kernels.h): only for CFR solvers whose step runs on a kernel that reads the
// per-slot records (the flags live there) -- i.e. every BASELINE configuration.
bool Engine::enable_root_dedup() {
  if (root_dedup_) return true;
  if (!p_.use_cfr || !(wave_ok_ || (flat_ok_ && rows_global_ok_))) return false;
  RBL_HIP_CHECK(hipSetDevice(device_));
  sync();
  d_lane_skip_.alloc((size_t)max_lanes_);
  RBL_HIP_CHECK(hipMemset(d_lane_skip_.p, 0, (size_t)max_lanes_ * sizeof(int)));
  d_snap_all_.alloc((size_t)(p_.num_iters + 1) * emax_ * g_.H);
  root_dedup_ = true;
  return true;
}

void Engine::split_part_queries(int part, hipStream_t st) {
  if (rows_ <= 0 || tabs_.max_L == 0) return;
  const int64_t r0 = part_row_[part], nr = part_row_[part + 1] - r0;
  if (nr <= 0) return;
  if (info_dev_)  // device-resident epoch: the part's real row range lives on the device, nr is the launch bound
    launch_split_queries(d_queries_.p, g_.A, g_.H, d_qdyn_.p, q_ds_, d_qstat_.p, q_ss_, nr, st, info_dev_->part_row + part);
  else
    launch_split_queries(d_queries_.p + r0 * g_.query_size(), g_.A, g_.H, d_qdyn_.p + r0 * q_ds_, q_ds_, d_qstat_.p + r0 * q_ss_,
                         q_ss_, nr, st);
  RBL_HIP_CHECK(hipGetLastError());
  q_canon_stale_ = false;
}

void Engine::set_segments(const int (*seg_shape)[kSpSegs]) {
  for (int pt = 0; pt < 4; ++pt)
    for (int k = 0; k < kSpSegs; ++k) {
      const int sid = pt < n_parts_ ? seg_shape[pt][k] : 0;
      if (sid < 0 || sid >= (int)tabs_.shapes.size()) throw std::runtime_error("engine: bad segment head shape");
      const ShapeDev& sh = tabs_.shapes[sid];
      seg_lds_[pt][k] = gs_lds_bytes(sh);
      seg_threads_[pt][k] = flat_threads_for(sh.N);
    }
}

void Engine::join_streams() {
  RBL_HIP_CHECK(hipSetDevice(device_));
  for (int pt = 1; pt < n_parts_; ++pt) {
    RBL_HIP_CHECK(hipEventRecord(ev_join_[pt - 1], part_stream(pt)));
    RBL_HIP_CHECK(hipStreamWaitEvent(stream_, ev_join_[pt - 1], 0));
  }
}

void Engine::end_epoch_device(const SpEpochInfo& info) {
  rows_ = info.rows;
  const double per_row = net_mode_ == NetMode::kMlp
                             ? 2.0 * ((double)mlp_n_in_true_ * mlp_.n_hidden +
                                      (double)(mlp_.n_layers - 1) * mlp_.n_hidden * mlp_.n_hidden +
                                      (double)mlp_.n_hidden * mlp_.n_out)
                             : 0.0;
  for (int pt = 0; pt < n_parts_; ++pt) {
    part_row_[pt] = info.part_row[pt];
    for (int t = 0; t < 2; ++t) {
      part_bytes_[pt][t] = (double)info.part_bytes[pt][t];
      stats_.cfr_bytes += (double)timed_cfr_[pt][t] * part_bytes_[pt][t];
      timed_cfr_[pt][t] = 0;
    }
    const double nr = (double)(info.part_row[pt + 1] - info.part_row[pt]);
    stats_.net_rows += (int64_t)(timed_net_[pt] * nr);
    stats_.net_flops += timed_net_[pt] * nr * per_row;
    timed_net_[pt] = 0;
  }
  part_row_[n_parts_] = info.rows;
}

void Engine::ensure_mirror() {
  if (mirror_valid_) return;
  sync();
  RBL_HIP_CHECK(hipSetDevice(device_));
  const int H = g_.H;
  h_shape_.resize(B_);
  h_player_.resize(B_);
  h_row_.resize(B_);
  h_act_.resize(B_);
  h_bid_.resize(B_);
  h_beliefs_.resize((size_t)B_ * 2 * H);
  RBL_HIP_CHECK(hipMemcpy(h_shape_.data(), d_lane_shape_.p, B_ * sizeof(int), hipMemcpyDeviceToHost));
  RBL_HIP_CHECK(hipMemcpy(h_player_.data(), d_lane_player_.p, B_ * sizeof(int), hipMemcpyDeviceToHost));
  RBL_HIP_CHECK(hipMemcpy(h_row_.data(), d_lane_row_.p, B_ * sizeof(int), hipMemcpyDeviceToHost));
  RBL_HIP_CHECK(hipMemcpy(h_act_.data(), d_lane_act_.p, B_ * sizeof(int), hipMemcpyDeviceToHost));
  RBL_HIP_CHECK(hipMemcpy(h_beliefs_.data(), d_beliefs_.p, (size_t)B_ * 2 * H * sizeof(double), hipMemcpyDeviceToHost));
  int64_t rows = 0;
  for (int b = 0; b < B_; ++b) {
    h_bid_[b] = h_shape_[b] - 1;
    rows += tabs_.shapes[h_shape_[b]].L;
  }
  rows_ = rows;
  mirror_valid_ = true;
}

void Engine::launch(int mode, int trav, int next_trav, int steps_after, double alpha, double pos, double neg,
                    double strat) {
  CfrArgs a{};
  a.shapes = d_shapes_.p;
  a.parent = d_parent_.p;
  a.act = d_act_.p;
  a.cb = d_cb_.p;
  a.ce = d_ce_.p;
  a.depth = d_depth_.p;
  a.leaves = d_leaves_.p;
  a.terms = d_terms_.p;
  a.irank = d_irank_.p;
  a.leaf_row = d_leaf_row_.p;
  a.vrow = d_vrow_.p;
  a.pack = d_pack_.p;
  a.matches = d_matches_.p;
  a.wave_tabs = d_wave_tabs_.p;
  a.wave_epv = d_wave_epv_.p;
  a.flat_tabs = d_flat_tabs_.p;
  a.lane_rec = d_lane_rec_.p;
  a.H = g_.H;
  a.A = g_.A;
  a.Q = g_.query_size();
  a.faces = g_.faces;
  a.dice = g_.dice;
  a.Emax = emax_;
  a.Nmax = nmax_;
  a.lane_shape = d_lane_shape_.p;
  a.lane_root_player = d_lane_player_.p;
  a.lane_row_off = d_lane_row_.p;
  a.lane_act_iter = has_act_ ? d_lane_act_.p : nullptr;
  a.beliefs = d_beliefs_.p;
  a.sigma = d_sigma_.p;
  a.regrets = d_regrets_.p;
  a.sums = d_sums_.p;
  a.snapshot = d_snapshot_.p;
  a.lane_skip = root_dedup_ ? d_lane_skip_.p : nullptr;
  a.snap_all = root_dedup_ ? d_snap_all_.p : nullptr;
  a.root_mean = d_root_mean_.p;
  a.queries = d_queries_.p;
  // (qsplit_ changes only when the net KIND changes -- set_net_* under net_mutex_, after sync() has drained both streams.  A
  // weight refresh from another thread, ModelLocker::updateModel, keeps the kind and so the layout; switching kinds in the middle
  // of a solve is for the thread that drives the solve: tests and C callers)
  a.q_dyn = qsplit_ ? d_qdyn_.p : nullptr;
  a.q_dyn_stride = q_ds_;
  a.values = d_values_.p;
  a.scratch = d_scratch_.p;
  a.work_stride = work_stride_;
  a.use_lds = use_lds_ ? 1 : 0;
  a.mode = mode;
  a.trav = trav;
  a.next_trav = next_trav;
  a.steps_after = steps_after;
  a.alpha = alpha;
  a.pos = pos;
  a.neg = neg;
  a.strat = strat;
  a.optimistic = p_.optimistic ? 1 : 0;
  a.br_out = d_br_.p;
  a.dbg = cfr_dbg_ ? d_dbg_.p : nullptr;
  TimingAbortGuard abort_open_sample_on_unwind{this};
  for (int part = 0; part < n_parts_; ++part) {
    if (only_part_ >= 0 && part != only_part_) continue;
    const int l0 = part_lane_[part], cnt = part_lane_[part + 1] - l0;
    if (cnt <= 0) continue;
    hipStream_t st = part_stream(part);
    a.lane0 = l0;
    const bool is_step = mode == kModeStep || mode == kModeFpStep;
    if (is_step) time_begin(0, st);
    int which = 0;
    if (mode == kModeStep && wave_ok_ && launch_cfr_wave(a, cnt, wave_lds_bytes_, st)) {
      which = 2;
    } else if (mode == kModeStep && rows_global_ok_ && use_order_) {
      // one launch per size-sorted segment of the part, each with the LDS request of its largest tree
      a.lane_order = d_lane_order_.p;
      const int ns = sp_segments(cnt);
      for (int k = 0; k < ns; ++k) {
        const int p0 = l0 + (int)((long long)cnt * k / ns), p1 = l0 + (int)((long long)cnt * (k + 1) / ns);
        if (p1 <= p0) continue;
        a.lane0 = p0;
        if (flat_ok_) launch_cfr_flat(a, p1 - p0, seg_lds_[part][k], seg_threads_[part][k], st);
        else launch_cfr_rows_global(a, p1 - p0, seg_lds_[part][k], st);
      }
      a.lane_order = nullptr;
      a.lane0 = l0;
      which = flat_ok_ ? 4 : 3;
    } else if (mode == kModeStep && rows_global_ok_ &&
               (flat_ok_ ? launch_cfr_flat(a, cnt, rows_global_lds_, flat_threads_, st) : launch_cfr_rows_global(a, cnt, rows_global_lds_, st))) {
      which = flat_ok_ ? 4 : 3;
    } else if (mode == kModeStep && rows_ok_ && launch_cfr_rows(a, cnt, part_rows_block_[part], part_rows_lds_[part], st)) {
      which = 1;
    } else {
      launch_cfr(a, cnt, block_, lds_bytes_, st);
    }
    if (is_step) last_cfr_kernel_ = which;
    if (qsplit_) {
      if (mode == kModeInit || mode == kModeQueries) split_part_queries(part, st);  // the generic kernel wrote canonical rows
      else if (mode == kModeStep && (which == 2 || which == 4)) q_canon_stale_ = true;  // the wave / flat kernel wrote dynamic rows only
    }
    const bool sampled = is_step && time_end(0, st);
    RBL_HIP_CHECK(hipGetLastError());
    if (is_step) {
      if (sampled) {
        ++stats_.cfr_launches;
        if (info_dev_)
          ++timed_cfr_[part][trav];  // bytes are added at the end of the epoch (end_epoch_device)
        else
          stats_.cfr_bytes += part_bytes_[part][trav];
      }
      stats_.lane_steps += cnt;
    }
  }
}

void Engine::run_net() {
  if (rows_ == 0 || tabs_.max_L == 0) return;
  std::lock_guard<std::mutex> net_lock(net_mutex_);
  if (net_mode_ == NetMode::kZero) {
    if (!values_zeroed_) {
      sync();
      RBL_HIP_CHECK(hipMemsetAsync(d_values_.p, 0, d_values_.n * sizeof(float), stream_));
      RBL_HIP_CHECK(hipStreamSynchronize(stream_));
      values_zeroed_ = true;
    }
    return;
  }
  const int Q = g_.query_size(), H = g_.H;
  // the launch shape of the persistent forward follows the BATCH (net_grid_cap), not the net: it is set on a local copy here, under
  // net_mutex_, so that a weight refresh on another thread (set_net_mlp rebuilds mlp_) never races with it
  MlpDev mlp = mlp_;
  mlp.grid_cap = net_grid_cap(B_);
  TimingAbortGuard abort_open_sample_on_unwind{this};
  for (int part = 0; part < n_parts_; ++part) {
    if (only_part_ >= 0 && part != only_part_) continue;
    const int64_t r0 = part_row_[part], nr = part_row_[part + 1] - r0;
    if (nr <= 0) continue;
    hipStream_t st = part_stream(part);
    const bool timed = net_mode_ == NetMode::kMlp && timed_now();
    if (timed) time_begin(1, st);
    if (qsplit_ && net_mode_ == NetMode::kMlp) {  // split layout: dynamic rows in, static rows beside them
      if (info_dev_) {
        launch_mlp_forward(mlp, d_qdyn_.p, nr, d_values_.p, st, info_dev_->part_row + part);
      } else {
        MlpDev m2 = mlp;
        m2.q_stat = d_qstat_.p + r0 * q_ss_;
        launch_mlp_forward(m2, d_qdyn_.p + r0 * q_ds_, nr, d_values_.p + r0 * H, st, nullptr);
      }
      RBL_HIP_CHECK(hipGetLastError());
      if (timed && time_end(1, st)) {
        ++stats_.net_launches;
        if (info_dev_) {
          ++timed_net_[part];
        } else {
          stats_.net_rows += nr;
          stats_.net_flops += 2.0 * (double)nr *
                              ((double)mlp_n_in_true_ * mlp_.n_hidden +
                               (double)(mlp_.n_layers - 1) * mlp_.n_hidden * mlp_.n_hidden + (double)mlp_.n_hidden * mlp_.n_out);
        }
      }
      continue;
    }
    if (info_dev_) {  // rows [part_row[part], part_row[part + 1]) as written by sp_scan; nr is only the launch bound
      net_forward_dev(d_queries_.p, nr, d_values_.p, st, info_dev_->part_row + part);
      if (timed && time_end(1, st)) {
        ++stats_.net_launches;
        ++timed_net_[part];
      }
      continue;
    }
    net_forward_dev(d_queries_.p + r0 * Q, nr, d_values_.p + r0 * H, st);
    if (timed && time_end(1, st)) {
      ++stats_.net_launches;
      stats_.net_rows += nr;
      stats_.net_flops += 2.0 * (double)nr *
                          ((double)mlp_n_in_true_ * mlp_.n_hidden + (double)(mlp_.n_layers - 1) * mlp_.n_hidden * mlp_.n_hidden +
                           (double)mlp_.n_hidden * mlp_.n_out);
    }
  }
}

void Engine::step(int traverser) {
  RBL_HIP_CHECK(hipSetDevice(device_));
  if (B_ == 0) throw std::runtime_error("step: no lanes (call reset first)");
  if (traverser != 0 && traverser != 1) throw std::runtime_error("step: traverser must be 0 or 1");
  if (pending_trav_ != traverser) {  // queries on the device were encoded for the other traverser: re-encode
    launch(kModeQueries, 0, traverser, 0, 0, 1, 1, 1);
    pending_trav_ = traverser;
  }
  if (!p_.use_cfr) {  // fictitious play (FP::step, subgame_solving.cc:433-476)
    const int nu = num_strategies_ / 2 + 1;
    const double alpha_fp = p_.linear_update ? 2. / (nu + 1) : 1. / nu;
    const double factor = p_.linear_update ? static_cast<double>(nu + 1) / (nu + 2) : 1.0;
    for (int part = 0; part < n_parts_; ++part) {
      only_part_ = part;
      run_net();
      launch(kModeFpStep, traverser, 1 - traverser, iter_ + 1, alpha_fp, 1, 1, factor);
    }
    only_part_ = -1;
    ++num_strategies_;
    ++num_steps_[traverser];
    ++iter_;
    pending_trav_ = 1 - traverser;
    return;
  }
  const int k = num_steps_[traverser];
  // running mean step (subgame_solving.cc:580-590) and discounts (:592-617); "+1": the uniform strategy counts
  const double alpha = p_.linear_update ? 2. / (k + 2) : 1. / (k + 1);
  double pos = 1, neg = 1, strat = 1;
  {
    const double s = k + 1;
    if (p_.linear_update) {
      pos = neg = strat = s / (s + 1);
    } else if (p_.dcfr) {
      pos = p_.dcfr_alpha >= 5 ? 1 : std::pow(s, p_.dcfr_alpha) / (std::pow(s, p_.dcfr_alpha) + 1.);
      neg = p_.dcfr_beta <= -5 ? 0 : std::pow(s, p_.dcfr_beta) / (std::pow(s, p_.dcfr_beta) + 1.);
      strat = std::pow(s / (s + 1), p_.dcfr_gamma);
    }
  }
  for (int part = 0; part < n_parts_; ++part) {  // per stream: net forward, then the CFR step that consumes it
    only_part_ = part;
    run_net();
    launch(kModeStep, traverser, 1 - traverser, iter_ + 1, alpha, pos, neg, strat);
  }
  only_part_ = -1;
  ++num_steps_[traverser];
  ++iter_;
  pending_trav_ = 1 - traverser;
}

void Engine::multistep(int n) {
  if (n < 0) n = p_.num_iters;
  for (int i = 0; i < n; ++i) step(iter_ % 2);  // traverser = iteration parity (subgame_solving.cc:666-670)
}

void Engine::sync() {
  RBL_HIP_CHECK(hipSetDevice(device_));
  RBL_HIP_CHECK(hipStreamSynchronize(stream_));
  RBL_HIP_CHECK(hipStreamSynchronize(stream2_));
  for (int i = 0; i < 2; ++i) RBL_HIP_CHECK(hipStreamSynchronize(stream_x_[i]));
}

int Engine::tree_size(int lane) {
  ensure_mirror();
  check_lane(lane);
  return tabs_.shapes[h_shape_[lane]].N;
}

void Engine::read_lane(const double* dev_base, int lane, std::vector<double>* out) {
  sync();
  RBL_HIP_CHECK(hipSetDevice(device_));
  const size_t eh = (size_t)emax_ * g_.H;
  out->resize(eh);
  RBL_HIP_CHECK(hipMemcpyAsync(out->data(), dev_base + (size_t)lane * eh, eh * sizeof(double), hipMemcpyDeviceToHost,
                               stream_));
  RBL_HIP_CHECK(hipStreamSynchronize(stream_));
}

// edge-indexed [E][H] -> the reference's dense TreeStrategy [N][H][A] (zeros outside the legal range)
void Engine::expand_dense(int lane, const std::vector<double>& edge, double* out) const {
  const ShapeDev& s = tabs_.shapes[h_shape_[lane]];
  const int H = g_.H, A = g_.A;
  std::fill(out, out + (size_t)s.N * H * A, 0.0);
  for (int n = 1; n < s.N; ++n) {
    const int p = tabs_.parent[s.node_off + n], a = tabs_.act[s.node_off + n];
    for (int h = 0; h < H; ++h) out[((size_t)p * H + h) * A + a] = edge[(size_t)(n - 1) * H + h];
  }
}

void Engine::get(int lane, int which, double* out) {
  ensure_mirror();
  check_lane(lane);
  std::vector<double> edge;
  const ShapeDev& s = tabs_.shapes[h_shape_[lane]];
  const int H = g_.H, A = g_.A;
  switch (which) {
    case RBL_GET_LAST:
      read_lane(d_sigma_.p, lane, &edge);
      expand_dense(lane, edge, out);
      return;
    case RBL_GET_REGRETS:
      if (!p_.use_cfr) throw std::runtime_error("get: regrets exist for CFR only");
      read_lane(d_regrets_.p, lane, &edge);
      expand_dense(lane, edge, out);
      return;
    case RBL_GET_SUM:
      read_lane(d_sums_.p, lane, &edge);
      expand_dense(lane, edge, out);
      return;
    case RBL_GET_AVERAGE: {
      if (!p_.use_cfr) {  // FP keeps the average strategy itself on the device (it is what reach / sampling use)
        read_lane(d_sigma_.p, lane, &edge);
        expand_dense(lane, edge, out);
        return;
      }
      // average_strategies = normalised sum_strategies on every node whose mover has stepped (subgame_solving.cc:658-660);
      // rows never touched keep the uniform initialisation (:518-519)
      read_lane(d_sums_.p, lane, &edge);
      expand_dense(lane, edge, out);
      for (int n = 0; n < s.N; ++n) {
        const int c0 = tabs_.cb[s.node_off + n], c1 = tabs_.ce[s.node_off + n];
        if (c0 == c1) continue;
        const int mover = h_player_[lane] ^ (tabs_.depth[s.node_off + n] & 1);
        for (int h = 0; h < H; ++h) {
          double* row = out + ((size_t)n * H + h) * A;
          if (num_steps_[mover] == 0) {
            for (int c = c0; c < c1; ++c) row[tabs_.act[s.node_off + c]] = 1. / (c1 - c0);
          } else {
            double sum = 0;
            for (int a = 0; a < A; ++a) sum += row[a];
            for (int a = 0; a < A; ++a) row[a] = row[a] / sum;
          }
        }
      }
      return;
    }
    default:
      throw std::runtime_error("get: bad selector");
  }
}

// dense [N][H][A] -> the lane's edge-indexed sigma (for best-response / exploitability evaluation of a given strategy)
void Engine::set_strategy(int lane, const double* dense) {
  ensure_mirror();
  check_lane(lane);
  sync();
  const ShapeDev& s = tabs_.shapes[h_shape_[lane]];
  const int H = g_.H, A = g_.A;
  std::vector<double> edge((size_t)emax_ * H, 0.0);
  for (int n = 1; n < s.N; ++n) {
    const int p = tabs_.parent[s.node_off + n], a = tabs_.act[s.node_off + n];
    for (int h = 0; h < H; ++h) edge[(size_t)(n - 1) * H + h] = dense[((size_t)p * H + h) * A + a];
  }
  RBL_HIP_CHECK(hipMemcpyAsync(d_sigma_.p + (size_t)lane * emax_ * H, edge.data(), edge.size() * sizeof(double),
                               hipMemcpyHostToDevice, stream_));
  RBL_HIP_CHECK(hipStreamSynchronize(stream_));
  pending_trav_ = -1;  // the queries on the device no longer match sigma
}

// BRSolver::compute_br (subgame_solving.cc:316-358) for every lane against its current sigma; out [B][H] root values
void Engine::best_response(int traverser, double* out) {
  ensure_mirror();
  RBL_HIP_CHECK(hipSetDevice(device_));
  if (B_ == 0) throw std::runtime_error("best_response: no lanes (call reset first)");
  if (traverser != 0 && traverser != 1) throw std::runtime_error("best_response: traverser must be 0 or 1");
  if (d_br_.n < (size_t)max_lanes_ * g_.H) d_br_.alloc((size_t)max_lanes_ * g_.H);
  if (rows_ > 0) {  // depth-limited tree: leaf values come from the net, for this traverser and this sigma
    launch(kModeQueries, 0, traverser, 0, 0, 1, 1, 1);
    pending_trav_ = traverser;
    run_net();
  }
  launch(kModeBestResponse, traverser, -1, 0, 0, 1, 1, 1);
  sync();
  RBL_HIP_CHECK(hipMemcpy(out, d_br_.p, (size_t)B_ * g_.H * sizeof(double), hipMemcpyDeviceToHost));
}

// compute_ev (subgame_solving.cc:931-973) for every lane: the traverser follows the lane's sigma at its own nodes, the
// opponent's reach is taken under the same sigma (set_strategy a mix of the two strategies to evaluate one against the
// other); pseudo-leaves of depth-limited trees are valued by the engine's net like in best_response.
void Engine::evaluate(int traverser, double* out) {
  ensure_mirror();
  RBL_HIP_CHECK(hipSetDevice(device_));
  if (B_ == 0) throw std::runtime_error("evaluate: no lanes (call reset first)");
  if (traverser != 0 && traverser != 1) throw std::runtime_error("evaluate: traverser must be 0 or 1");
  if (d_br_.n < (size_t)max_lanes_ * g_.H) d_br_.alloc((size_t)max_lanes_ * g_.H);
  if (rows_ > 0) {
    launch(kModeQueries, 0, traverser, 0, 0, 1, 1, 1);
    pending_trav_ = traverser;
    run_net();
  }
  launch(kModeEvaluate, traverser, -1, 0, 0, 1, 1, 1);
  sync();
  RBL_HIP_CHECK(hipMemcpy(out, d_br_.p, (size_t)B_ * g_.H * sizeof(double), hipMemcpyDeviceToHost));
}

void Engine::get_snapshot(int lane, double* out) {
  ensure_mirror();
  check_lane(lane);
  if (!has_act_) throw std::runtime_error("get_snapshot: reset was called without act_iteration");
  if (iter_ < h_act_[lane]) throw std::runtime_error("get_snapshot: lane has not reached its act_iteration yet");
  std::vector<double> edge;
  read_lane(d_snapshot_.p, lane, &edge);
  expand_dense(lane, edge, out);
}

void Engine::hand_values(int lane, int player, double* out) {
  ensure_mirror();
  sync();
  check_lane(lane);
  if (player != 0 && player != 1) throw std::runtime_error("hand_values: player must be 0 or 1");
  RBL_HIP_CHECK(hipSetDevice(device_));
  RBL_HIP_CHECK(hipMemcpyAsync(out, d_root_mean_.p + ((size_t)lane * 2 + player) * g_.H, g_.H * sizeof(double),
                               hipMemcpyDeviceToHost, stream_));
  RBL_HIP_CHECK(hipStreamSynchronize(stream_));
}

void Engine::write_root_query(int traverser, int last_bid, int player, const double* b0, const double* b1,
                              float* q) const {  // write_query_to, subgame_solving.cc:104-123
  int w = 0;
  q[w++] = (float)player;
  q[w++] = (float)traverser;
  for (int a = 0; a < g_.A; ++a) q[w++] = (a == last_bid) ? 1.0f : 0.0f;
  normalize_safe(b0, g_.H, kEps, q + w);
  w += g_.H;
  normalize_safe(b1, g_.H, kEps, q + w);
}

void Engine::examples(int lane, float* queries, float* values) {
  ensure_mirror();
  sync();  // update_value_network, subgame_solving.cc:672-676
  check_lane(lane);
  const int H = g_.H, Q = g_.query_size();
  std::vector<double> rm(2 * H);
  RBL_HIP_CHECK(hipSetDevice(device_));
  RBL_HIP_CHECK(hipMemcpyAsync(rm.data(), d_root_mean_.p + (size_t)lane * 2 * H, 2 * H * sizeof(double),
                               hipMemcpyDeviceToHost, stream_));
  RBL_HIP_CHECK(hipStreamSynchronize(stream_));
  const double* b = h_beliefs_.data() + (size_t)lane * 2 * H;
  for (int t = 0; t < 2; ++t) {
    write_root_query(t, h_bid_[lane], h_player_[lane], b, b + H, queries + (size_t)t * Q);
    for (int h = 0; h < H; ++h) values[t * H + h] = (float)rm[t * H + h];  // double -> float (:224)
  }
}

void Engine::get_net_debug(long long* out) {
  sync();
  if (d_ndbg_.p) RBL_HIP_CHECK(hipMemcpy(out, d_ndbg_.p, 1024 * 16 * sizeof(long long), hipMemcpyDeviceToHost));
}

void Engine::get_debug(long long* out) {
  sync();
  RBL_HIP_CHECK(hipSetDevice(device_));
  RBL_HIP_CHECK(hipStreamSynchronize(stream_));
  if (d_dbg_.p) RBL_HIP_CHECK(hipMemcpy(out, d_dbg_.p, (size_t)B_ * 16 * sizeof(long long), hipMemcpyDeviceToHost));
}

void Engine::get_queries(float* out) {
  ensure_mirror();
  sync();
  RBL_HIP_CHECK(hipSetDevice(device_));
  if (rows_ > 0 && qsplit_ && q_canon_stale_) {  // the steps since the last init wrote dynamic rows only: rebuild the canonical rows
    launch_unsplit_queries(d_queries_.p, g_.A, g_.H, d_qdyn_.p, q_ds_, d_qstat_.p, q_ss_, rows_, stream_);
    RBL_HIP_CHECK(hipGetLastError());
    q_canon_stale_ = false;
  }
  if (rows_ > 0)
    RBL_HIP_CHECK(hipMemcpyAsync(out, d_queries_.p, (size_t)rows_ * g_.query_size() * sizeof(float),
                                 hipMemcpyDeviceToHost, stream_));
  RBL_HIP_CHECK(hipStreamSynchronize(stream_));
}

void Engine::read_snapshots(const double** snap, const double** root_mean) {
  sync();
  RBL_HIP_CHECK(hipSetDevice(device_));
  const size_t eh = (size_t)emax_ * g_.H;
  const size_t n_snap = (size_t)max_lanes_ * eh, n_rm = (size_t)max_lanes_ * 2 * g_.H;
  if (!h_pinned_) {  // page-locked staging: the per-epoch read-back (17.7 MB at 4096 lanes of 1dx6f) runs at PCIe speed
    RBL_HIP_CHECK(hipHostMalloc((void**)&h_pinned_, (n_snap + n_rm) * sizeof(double), hipHostMallocDefault));
  }
  RBL_HIP_CHECK(hipMemcpyAsync(h_pinned_, d_snapshot_.p, (size_t)B_ * eh * sizeof(double), hipMemcpyDeviceToHost, stream_));
  RBL_HIP_CHECK(hipMemcpyAsync(h_pinned_ + n_snap, d_root_mean_.p, (size_t)B_ * 2 * g_.H * sizeof(double),
                               hipMemcpyDeviceToHost, stream_));
  RBL_HIP_CHECK(hipStreamSynchronize(stream_));
  *snap = h_pinned_;
  *root_mean = h_pinned_ + n_snap;
}

// =================================================================================================== evaluation
// Full-tree strategy by recursive subgame solving (compute_strategy_recursive / _to_leaf, recursive_solving.cc:47-134,
// 277-299).  The reference recurses depth-first, one subgame at a time; a subgame depends only on the beliefs handed
// down to it, so here the frontier is solved level by level with every subgame of a level as one lane of the engine.
void strategy_recursive(Engine& e, bool to_leaf, double* out) {
  const Rules& g = e.rules();
  const ShapeTables& tb = e.tables();
  const int H = g.H, A = g.A;
  const std::vector<Node> full = unroll_tree(g, -1, 0, 1000000);
  std::fill(out, out + full.size() * (size_t)H * A, 0.0);
  struct Item {
    int node;
    std::vector<double> b;  // [2][H]
  };
  std::vector<Item> frontier;
  frontier.push_back(Item{0, std::vector<double>(2 * (size_t)H, 1.0 / H)});  // get_initial_beliefs
  std::vector<double> dense;
  while (!frontier.empty()) {
    std::vector<Item> next;
    for (size_t base = 0; base < frontier.size(); base += e.max_lanes()) {
      const int B = (int)std::min<size_t>(e.max_lanes(), frontier.size() - base);
      std::vector<int32_t> bid(B), pl(B);
      std::vector<double> bel((size_t)B * 2 * H);
      for (int i = 0; i < B; ++i) {
        const Item& it = frontier[base + i];
        bid[i] = full[it.node].last_bid;
        pl[i] = full[it.node].player;
        std::copy(it.b.begin(), it.b.end(), bel.begin() + (size_t)i * 2 * H);
      }
      e.reset(B, bid.data(), pl.data(), bel.data(), nullptr);
      e.multistep(-1);
      for (int i = 0; i < B; ++i) {
        const Item& it = frontier[base + i];
        const ShapeDev& sh = tb.shapes[bid[i] + 1];
        dense.resize((size_t)sh.N * H * A);
        e.get(i, RBL_GET_AVERAGE, dense.data());
        if (!to_leaf) {  // :47-74: keep the root row, hand Bayes-updated beliefs to every child
          const Node& nd = full[it.node];
          std::copy(dense.begin(), dense.begin() + (size_t)H * A, out + (size_t)it.node * H * A);
          int lo, hi;
          g.bid_range(nd.last_bid, &lo, &hi);
          for (int c = nd.cb; c < nd.ce; ++c) {
            if (full[c].last_bid == g.liar) continue;
            Item ch{c, it.b};
            const int action = c - nd.cb + lo;
            double* nb = ch.b.data() + (size_t)nd.player * H;
            for (int h = 0; h < H; ++h) nb[h] *= dense[(size_t)h * A + action];
            normalize_safe(nb, H, kEps, nb);
            next.push_back(std::move(ch));
          }
        } else {  // :76-134 (use_sampling_strategy = false): copy the whole partial tree, recurse at its open leaves
          struct Q {
            int f, p;
            std::vector<double> r;
          };
          std::vector<Q> queue;
          queue.push_back(Q{it.node, 0, it.b});
          for (size_t qi = 0; qi < queue.size(); ++qi) {
            Q cur = queue[qi];
            std::copy(dense.begin() + (size_t)cur.p * H * A, dense.begin() + (size_t)(cur.p + 1) * H * A,
                      out + (size_t)cur.f * H * A);
            const Node& fn = full[cur.f];
            const int pc0 = tb.cb[sh.node_off + cur.p], pc1 = tb.ce[sh.node_off + cur.p];
            int lo, hi;
            g.bid_range(fn.last_bid, &lo, &hi);
            for (int k = 0; k < pc1 - pc0; ++k) {
              Q ch{fn.cb + k, pc0 + k, cur.r};
              double* r = ch.r.data() + (size_t)fn.player * H;
              for (int h = 0; h < H; ++h) r[h] *= dense[((size_t)cur.p * H + h) * A + lo + k];
              queue.push_back(std::move(ch));
            }
            if (pc0 == pc1 && fn.cb != fn.ce) {
              Item nr{cur.f, cur.r};
              normalize_safe(nr.b.data(), H, kEps, nr.b.data());
              normalize_safe(nr.b.data() + H, H, kEps, nr.b.data() + H);
              next.push_back(std::move(nr));
            }
          }
        }
      }
    }
    frontier.swap(next);
  }
}

// compute_sampled_strategy_recursive_to_leaf (recursive_solving.cc:301-327): like the to-leaf recursion above, but every
// subgame is stopped at its own iteration -- one draw from mt19937(seed) per subgame, weights (i even ? i/2+1 : 0), in the
// order in which the reference's depth-first recursion builds its solvers -- and contributes its SAMPLING strategy
// (last_strategies at that iteration; it also propagates the beliefs).  The draw order only depends on the tree, so all
// draws are made up front; then the frontier is solved level by level with one lane per subgame, each lane snapshotting
// its strategy at its own act_iteration.  root_only: subgames below the root are solved to the end of the game
// (max_depth = 100000, no value net) on a helper engine.
void strategy_recursive_sampled(Engine& e, int seed, bool root_only, double* out) {
  const Rules& g = e.rules();
  const int H = g.H, A = g.A;
  const std::vector<Node> full = unroll_tree(g, -1, 0, 1000000);
  std::fill(out, out + full.size() * (size_t)H * A, 0.0);
  std::unique_ptr<Engine> deep;  // root_only: full-depth subgames
  if (root_only) {
    rbl_params dp = e.params();
    dp.max_depth = 100000;
    deep.reset(new Engine(e.device(), g.dice, g.faces, dp, e.max_lanes()));
    deep->set_net_zero();
  }
  auto engine_of = [&](int node) -> Engine& { return root_only && node != 0 ? *deep : e; };

  // ---- 1. act_iteration of every subgame root, in the reference's solver-construction order
  std::vector<int> act(full.size(), -1);
  {
    std::mt19937 gen(seed);
    std::vector<double> w;
    for (int i = 0; i < e.params().num_iters; ++i) w.push_back(i % 2 ? 0.0 : (i / 2. + 1));
    std::function<void(int)> visit = [&](int node) {
      if (full[node].last_bid == g.liar) return;
      std::discrete_distribution<int> dist(w.begin(), w.end());
      act[node] = dist(gen);
      const ShapeTables& tb = engine_of(node).tables();
      const ShapeDev& sh = tb.shapes[full[node].last_bid + 1];
      std::vector<std::pair<int, int>> queue{{node, 0}};  // (full node, partial node), BFS
      for (size_t qi = 0; qi < queue.size(); ++qi) {
        const int f = queue[qi].first, p = queue[qi].second;
        const int pc0 = tb.cb[sh.node_off + p], pc1 = tb.ce[sh.node_off + p];
        for (int k = 0; k < pc1 - pc0; ++k) queue.push_back({full[f].cb + k, pc0 + k});
        if (pc0 == pc1 && full[f].cb != full[f].ce) visit(f);
      }
    };
    visit(0);
  }

  // ---- 2. level by level
  struct Item {
    int node;
    std::vector<double> b;  // [2][H]
  };
  std::vector<Item> frontier;
  frontier.push_back(Item{0, std::vector<double>(2 * (size_t)H, 1.0 / H)});
  std::vector<double> dense;
  while (!frontier.empty()) {
    std::vector<Item> next;
    Engine& en = engine_of(frontier[0].node);  // a level is either the root alone or entirely below it
    const ShapeTables& tb = en.tables();
    for (size_t base = 0; base < frontier.size(); base += en.max_lanes()) {
      const int B = (int)std::min<size_t>(en.max_lanes(), frontier.size() - base);
      std::vector<int32_t> bid(B), pl(B), ai(B);
      std::vector<double> bel((size_t)B * 2 * H);
      int steps = 0;
      for (int i = 0; i < B; ++i) {
        const Item& it = frontier[base + i];
        bid[i] = full[it.node].last_bid;
        pl[i] = full[it.node].player;
        ai[i] = act[it.node];
        steps = std::max(steps, ai[i]);
        std::copy(it.b.begin(), it.b.end(), bel.begin() + (size_t)i * 2 * H);
      }
      en.reset(B, bid.data(), pl.data(), bel.data(), ai.data());
      en.multistep(steps);
      for (int i = 0; i < B; ++i) {
        const Item& it = frontier[base + i];
        const ShapeDev& sh = tb.shapes[bid[i] + 1];
        dense.resize((size_t)sh.N * H * A);
        en.get_snapshot(i, dense.data());
        struct Q {
          int f, p;
          std::vector<double> r;
        };
        std::vector<Q> queue;
        queue.push_back(Q{it.node, 0, it.b});
        for (size_t qi = 0; qi < queue.size(); ++qi) {
          Q cur = queue[qi];
          std::copy(dense.begin() + (size_t)cur.p * H * A, dense.begin() + (size_t)(cur.p + 1) * H * A,
                    out + (size_t)cur.f * H * A);
          const Node& fn = full[cur.f];
          const int pc0 = tb.cb[sh.node_off + cur.p], pc1 = tb.ce[sh.node_off + cur.p];
          int lo, hi;
          g.bid_range(fn.last_bid, &lo, &hi);
          for (int k = 0; k < pc1 - pc0; ++k) {
            Q ch{fn.cb + k, pc0 + k, cur.r};
            double* r = ch.r.data() + (size_t)fn.player * H;
            for (int h = 0; h < H; ++h) r[h] *= dense[((size_t)cur.p * H + h) * A + lo + k];
            queue.push_back(std::move(ch));
          }
          if (pc0 == pc1 && fn.cb != fn.ce) {
            Item nr{cur.f, cur.r};
            normalize_safe(nr.b.data(), H, kEps, nr.b.data());
            normalize_safe(nr.b.data() + H, H, kEps, nr.b.data() + H);
            next.push_back(std::move(nr));
          }
        }
      }
    }
    frontier.swap(next);
  }
}

// =================================================================================================== SelfPlay
SelfPlay::SelfPlay(Engine* e, int n_lanes, const int32_t* seeds, double random_action_prob, bool sample_leaf)
    : e_(e), n_(n_lanes), rap_((float)random_action_prob), leaf_(sample_leaf) {
  if (n_lanes < 1 || n_lanes > e->max_lanes()) throw std::runtime_error("selfplay: n_lanes must be in [1, max_lanes]");
  const Rules& g = e->rules();
  seeds_.assign(seeds, seeds + n_);
  for (int i = 0; i < n_; ++i) gen_.emplace_back(seeds[i]);  // RlRunner: std::mt19937 gen_(seed)
  bid_.assign(n_, g.liar);  // "terminal": the first advance() starts a fresh game on every lane
  player_.assign(n_, 0);
  act_.assign(n_, 0);
  beliefs_.assign((size_t)n_ * 2 * g.H, 0.0);
}

void SelfPlay::state(int lane, int32_t* last_bid, int32_t* player) const {
  if (lane < 0 || lane >= n_) throw std::runtime_error("selfplay: lane out of range");
  *last_bid = bid_[lane];
  *player = player_[lane];
}

// beliefs[h] *= sigma[node][h][action]; normalize_beliefs_inplace (recursive_solving.cc:41-44)
void SelfPlay::bayes(double* b, const double* sigma, int, int child, int H) const {
  for (int h = 0; h < H; ++h) b[h] *= sigma[(size_t)(child - 1) * H + h];
  normalize_safe(b, H, kEps, b);
}

void SelfPlay::sample_to_leaf(int lane, const double* sigma) {  // recursive_solving.cc:192-246
  const Rules& g = e_->rules();
  const ShapeTables& tb = e_->tables();
  const ShapeDev& s = tb.shapes[bid_[lane] + 1];
  const int H = g.H, A = g.A;
  std::mt19937& gen = gen_[lane];
  double* bel = beliefs_.data() + (size_t)lane * 2 * H;
  const int root_player = player_[lane];
  std::vector<int> path;  // child node ids
  {
    int n = 0;
    const int br_sampler = std::uniform_int_distribution<>(0, 1)(gen);
    std::vector<double> sb(bel, bel + 2 * H);
    std::vector<double> row(A);
    while (tb.cb[s.node_off + n] != tb.ce[s.node_off + n]) {
      const float eps = std::uniform_real_distribution<float>(0, 1)(gen);
      const int mover = root_player ^ (tb.depth[s.node_off + n] & 1);
      const int c0 = tb.cb[s.node_off + n], c1 = tb.ce[s.node_off + n];
      const int lo = tb.act[s.node_off + c0];
      int action;
      if (mover == br_sampler && eps < rap_) {
        std::uniform_int_distribution<> dis(lo, lo + (c1 - c0) - 1);
        action = dis(gen);
      } else {
        std::discrete_distribution<> hd(sb.begin() + mover * H, sb.begin() + (mover + 1) * H);
        const int hand = hd(gen);
        std::fill(row.begin(), row.end(), 0.0);
        for (int c = c0; c < c1; ++c) row[tb.act[s.node_off + c]] = sigma[(size_t)(c - 1) * H + hand];
        std::discrete_distribution<> ad(row.begin(), row.end());
        action = ad(gen);
      }
      const int child = c0 + action - lo;
      bayes(sb.data() + mover * H, sigma, 0, child, H);
      path.push_back(child);
      n = child;
    }
  }
  for (int child : path) {  // second pass on the lane's real beliefs (:235-245)
    bayes(bel + player_[lane] * H, sigma, 0, child, H);
    bid_[lane] = tb.act[s.node_off + child];
    player_[lane] = 1 - player_[lane];
  }
}

void SelfPlay::sample_single(int lane, const double* sigma) {  // recursive_solving.cc:248-275
  const Rules& g = e_->rules();
  const ShapeTables& tb = e_->tables();
  const ShapeDev& s = tb.shapes[bid_[lane] + 1];
  const int H = g.H, A = g.A;
  std::mt19937& gen = gen_[lane];
  double* bel = beliefs_.data() + (size_t)lane * 2 * H;
  const int br_sampler = std::uniform_int_distribution<>(0, 1)(gen);
  const float eps = std::uniform_real_distribution<float>(0, 1)(gen);
  int lo, hi;
  g.bid_range(bid_[lane], &lo, &hi);
  const int pl = player_[lane];
  int action;
  if (pl == br_sampler && eps < rap_) {
    std::uniform_int_distribution<> dis(lo, hi - 1);
    action = dis(gen);
  } else {
    std::discrete_distribution<> hd(bel + pl * H, bel + (pl + 1) * H);
    const int hand = hd(gen);
    std::vector<double> row(A, 0.0);
    const int c0 = tb.cb[s.node_off], c1 = tb.ce[s.node_off];
    for (int c = c0; c < c1; ++c) row[tb.act[s.node_off + c]] = sigma[(size_t)(c - 1) * H + hand];
    std::discrete_distribution<> ad(row.begin(), row.end());
    action = ad(gen);
  }
  const int c0 = tb.cb[s.node_off];
  if (c0 == tb.ce[s.node_off]) throw std::runtime_error("selfplay: max_depth=0 subgame has no actions to sample");
  bayes(bel + pl * H, sigma, 0, c0 + action - lo, H);
  bid_[lane] = action;  // Game::act, liars_dice.h:121-129
  player_[lane] = 1 - pl;
}

SelfPlay::~SelfPlay() {
  if (h_pin_) (void)hipHostFree(h_pin_);
}

void SelfPlay::device_examples(const float** q, const float** v) const {
  *q = mode_ == 1 ? d_ex_q_.p : nullptr;
  *v = mode_ == 1 ? d_ex_v_.p : nullptr;
}

// The walk runs on the device unless the value net is a host/device callback (the net launch then needs host-side row
// counts every iteration) or RBL_SELFPLAY_HOST=1 asks for the host walk (A/B: both produce identical trajectories,
// tests/test_selfplay_parity.py).  Decided once, at the first epoch: the two modes keep separate RNG states.
int SelfPlay::decide_mode() {
  if (mode_ < 0) {
    const Rules& g = e_->rules();
    const bool ok = e_->device_epochs_supported() && !env_int("RBL_SELFPLAY_HOST", 0) && g.H <= 64 && g.A <= 64 &&
                    (leaf_ || e_->params().max_depth >= 1);
    mode_ = ok ? 1 : 0;
    if (mode_ == 1) init_device();
  }
  return mode_;
}

int64_t SelfPlay::advance(rbl_example_fn sink, void* user) {
  decide_mode();
  if (mode_ == 1) {
    if (!e_->device_epochs_supported())
      throw std::runtime_error("selfplay: the value net became a callback net after device self-play started; create "
                               "the lanes after setting the net, or run with RBL_SELFPLAY_HOST=1");
    return advance_device(sink, user);
  }
  return advance_host(sink, user);
}

void SelfPlay::init_device() {
  const Rules& g = e_->rules();
  const int H = g.H, Q = g.query_size();
  RBL_HIP_CHECK(hipSetDevice(e_->device()));
  hipStream_t st = e_->stream();
  std::vector<uint32_t> mt((size_t)624 * n_), one(624);
  for (int i = 0; i < n_; ++i) {
    mt19937_seed_state((uint32_t)seeds_[i], one.data());  // std::mt19937(seed): seed taken modulo 2^32
    for (int k = 0; k < 624; ++k) mt[(size_t)k * n_ + i] = one[k];
  }
  // the staging vectors are named locals that live until the stream synchronisation below: an asynchronous copy from
  // pageable memory is host-synchronous in today's runtime, but the HIP API does not promise that
  const std::vector<int> mt_idx(n_, 624), bid0(bid_.begin(), bid_.end()), player0(player_.begin(), player_.end());
  d_mt_.upload(mt, st);
  d_mt_idx_.upload(mt_idx, st);
  d_bid_.upload(bid0, st);
  d_player_.upload(player0, st);
  d_sp_beliefs_.upload(beliefs_, st);
  d_ex_q_.alloc((size_t)2 * n_ * Q);
  d_ex_v_.alloc((size_t)2 * n_ * H);
  d_info_.alloc(1);
  RBL_HIP_CHECK(hipMemsetAsync(d_info_.p, 0, sizeof(SpEpochInfo), st));
  const size_t bytes = (size_t)2 * n_ * (Q + H) * sizeof(float) + (size_t)2 * n_ * sizeof(int) + sizeof(SpEpochInfo);
  RBL_HIP_CHECK(hipHostMalloc((void**)&h_pin_, bytes, hipHostMallocDefault));
  RBL_HIP_CHECK(hipStreamSynchronize(st));  // the staging vectors above go out of scope
  ex_lane_.resize((size_t)2 * n_);
  for (int i = 0; i < n_; ++i) ex_lane_[2 * i] = ex_lane_[2 * i + 1] = i;
  // REBEL_AMD_ROOT_DEDUP=1 (default off; a labelled extra, never the measured headline): the root subgame is solved once per epoch
  if (env_int("REBEL_AMD_ROOT_DEDUP", 0) != 0 && !e_->enable_root_dedup())
    std::fprintf(stderr, "rebel_amd: REBEL_AMD_ROOT_DEDUP=1 ignored: this engine's CFR step does not run on a record-driven kernel\n");
}

SpArgs SelfPlay::sp_args() const {
  const Rules& g = e_->rules();
  const Engine::DeviceLanes dl = e_->device_lanes();
  SpArgs a{};
  a.shapes = e_->shapes_dev();
  a.act = e_->act_dev();
  a.cb = e_->cb_dev();
  a.ce = e_->ce_dev();
  a.depth = e_->depth_dev();
  a.shape_epar = dl.shape_epar;
  a.H = g.H;
  a.A = g.A;
  a.Q = g.query_size();
  a.liar = g.liar;
  a.Emax = e_->emax();
  a.num_iters = e_->params().num_iters;
  a.n = n_;
  a.sample_leaf = leaf_ ? 1 : 0;
  a.rap = rap_;
  a.mt = d_mt_.p;
  a.mt_idx = d_mt_idx_.p;
  a.bid = d_bid_.p;
  a.player = d_player_.p;
  a.beliefs = d_sp_beliefs_.p;
  a.lane_shape = dl.shape;
  a.lane_player = dl.player;
  a.lane_row = dl.row;
  a.lane_act = dl.act;
  a.eng_beliefs = dl.beliefs;
  a.snapshot = dl.snapshot;
  a.root_mean = dl.root_mean;
  a.ex_q = d_ex_q_.p;
  a.ex_v = d_ex_v_.p;
  a.info = d_info_.p;
  a.dedup = e_->root_dedup() ? 1 : 0;
  a.lane_skip = e_->lane_skip_dev();
  a.snap_all = e_->snap_all_dev();
  a.lane_order = e_->lane_order_dev();
  a.n_parts = e_->parts_for(n_);
  e_->part_lanes(n_, a.part_lane);
  return a;
}

// One epoch without the host in the loop: [sp_begin, sp_scan] -> solver init -> num_iters x (net, CFR step) on the
// engine's streams -> [sp_end]; the only read-back is the epoch's examples (2n x (Q + H) floats) and 8 bytes of game
// state per lane, after which the next epoch's launches follow immediately.
int64_t SelfPlay::advance_device(rbl_example_fn sink, void* user) {
  const Rules& g = e_->rules();
  const int H = g.H, Q = g.query_size();
  const int num_iters = e_->params().num_iters;
  RBL_HIP_CHECK(hipSetDevice(e_->device()));
  hipStream_t st = e_->stream();
  const SpArgs a = sp_args();
  launch_sp_begin(a, st);
  launch_sp_scan(a, st);
  if (a.lane_order) launch_sp_order(a, st);
  RBL_HIP_CHECK(hipGetLastError());
  e_->begin_epoch_device(n_, d_info_.p);
  e_->multistep(num_iters);
  e_->join_streams();
  launch_sp_end(a, st);
  RBL_HIP_CHECK(hipGetLastError());
  float* hq = reinterpret_cast<float*>(h_pin_);
  float* hv = hq + (size_t)2 * n_ * Q;
  int* hb = reinterpret_cast<int*>(hv + (size_t)2 * n_ * H);
  int* hp = hb + n_;
  SpEpochInfo* hi = reinterpret_cast<SpEpochInfo*>(hp + n_);
  if (sink) {  // without a sink the examples stay on the device (rbl_selfplay_device_examples)
    RBL_HIP_CHECK(hipMemcpyAsync(hq, d_ex_q_.p, (size_t)2 * n_ * Q * sizeof(float), hipMemcpyDeviceToHost, st));
    RBL_HIP_CHECK(hipMemcpyAsync(hv, d_ex_v_.p, (size_t)2 * n_ * H * sizeof(float), hipMemcpyDeviceToHost, st));
  }
  RBL_HIP_CHECK(hipMemcpyAsync(hb, d_bid_.p, (size_t)n_ * sizeof(int), hipMemcpyDeviceToHost, st));
  RBL_HIP_CHECK(hipMemcpyAsync(hp, d_player_.p, (size_t)n_ * sizeof(int), hipMemcpyDeviceToHost, st));
  RBL_HIP_CHECK(hipMemcpyAsync(hi, d_info_.p, sizeof(SpEpochInfo), hipMemcpyDeviceToHost, st));
  RBL_HIP_CHECK(hipStreamSynchronize(st));
  e_->end_epoch_device(*hi);
  games_ = (int64_t)hi->games;
  skipped_ += hi->skipped;
  std::copy(hb, hb + n_, bid_.begin());
  std::copy(hp, hp + n_, player_.begin());
  if (sink) sink(user, (int64_t)2 * n_, ex_lane_.data(), hq, Q, hv, H);
  // subgame-CFR-iterations EXECUTED this epoch: with root de-duplication the served root lanes ran none
  return (int64_t)(n_ - hi->skipped) * num_iters;
}

int64_t SelfPlay::advance_host(rbl_example_fn sink, void* user) {
  const Rules& g = e_->rules();
  const int H = g.H, Q = g.query_size();
  const int num_iters = e_->params().num_iters;
  // RlRunner::step (recursive_solving.cc:160-182), one subgame of it per lane
  for (int i = 0; i < n_; ++i) {
    if (bid_[i] == g.liar) {  // previous game over: state_ = root, beliefs_ uniform (:161-163)
      bid_[i] = -1;
      player_[i] = 0;
      double* b = beliefs_.data() + (size_t)i * 2 * H;
      for (int k = 0; k < 2 * H; ++k) b[k] = 1.0 / H;
    }
    act_[i] = std::uniform_int_distribution<>(0, num_iters)(gen_[i]);  // inclusive (:168-169)
  }
  std::vector<int32_t> root_bid(bid_), root_player(player_);
  std::vector<double> root_beliefs(beliefs_);
  const bool trace = env_int("RBL_DEBUG_EPOCH", 0) != 0;  // host-side phase times of one epoch on stderr
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return std::chrono::duration<double, std::milli>(b - a).count();
  };
  const auto t0 = now();
  e_->reset(n_, bid_.data(), player_.data(), beliefs_.data(), act_.data());
  const auto t1 = now();
  e_->multistep(num_iters);
  if (trace) e_->sync();
  const auto t2 = now();
  const double *snap_p = nullptr, *rmean_p = nullptr;
  e_->read_snapshots(&snap_p, &rmean_p);
  const auto t3 = now();
  const size_t eh = (size_t)e_->emax() * H;
  ex_q_.resize((size_t)2 * n_ * Q);
  ex_v_.resize((size_t)2 * n_ * H);
  ex_lane_.resize((size_t)2 * n_);
  // The sampling walk of a lane touches only that lane's RNG, state and beliefs: lanes are spread over a few host
  // threads (the walk of 4096 lanes is ~4 ms on one core, 3 % of an epoch during which the GPU would sit idle).
  auto walk = [&](int lo, int hi, int64_t* finished) {
    for (int i = lo; i < hi; ++i) {
      const double* sigma = snap_p + (size_t)i * eh;
      if (leaf_)
        sample_to_leaf(i, sigma);
      else
        sample_single(i, sigma);
      if (bid_[i] == g.liar) ++*finished;
      const double* rb = root_beliefs.data() + (size_t)i * 2 * H;
      for (int t = 0; t < 2; ++t) {  // update_value_network (subgame_solving.cc:672-676)
        const size_t k = (size_t)2 * i + t;
        e_->write_root_query(t, root_bid[i], root_player[i], rb, rb + H, ex_q_.data() + k * Q);
        for (int h = 0; h < H; ++h) ex_v_[k * H + h] = (float)rmean_p[((size_t)i * 2 + t) * H + h];
        ex_lane_[k] = i;
      }
    }
  };
  {
    const int want = env_int("RBL_HOST_THREADS", 8);
    const int nt = std::max(1, std::min({want, n_ / 256, (int)std::thread::hardware_concurrency()}));
    std::vector<int64_t> done(nt, 0);
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t)
      pool.emplace_back(walk, (int)((int64_t)n_ * t / nt), (int)((int64_t)n_ * (t + 1) / nt), &done[t]);
    walk(0, (int)((int64_t)n_ / nt), &done[0]);
    for (auto& th : pool) th.join();
    for (int64_t d : done) games_ += d;
  }
  const auto t4 = now();
  if (sink) sink(user, (int64_t)2 * n_, ex_lane_.data(), ex_q_.data(), Q, ex_v_.data(), H);
  if (trace)
    std::fprintf(stderr, "[rbl] epoch: reset %.2f ms, %d iterations %.2f ms, snapshot read-back %.2f ms, sampling walk %.2f ms, sink %.2f ms\n",
                 ms(t0, t1), num_iters, ms(t1, t2), ms(t2, t3), ms(t3, t4), ms(t4, now()));
  return (int64_t)n_ * num_iters;
}

}  // namespace rbl

// =================================================================================================== C ABI
struct rbl_engine {
  rbl::Engine impl;
  rbl_engine(int device, int dice, int faces, const rbl_params& p, int max_lanes)
      : impl(device, dice, faces, p, max_lanes) {}
};
namespace rbl {
Engine& engine_impl(rbl_engine* e) { return e->impl; }
}  // namespace rbl
struct rbl_selfplay {
  rbl::SelfPlay impl;
  rbl_selfplay(rbl::Engine* e, int n, const int32_t* seeds, double rap, bool leaf) : impl(e, n, seeds, rap, leaf) {}
};

namespace {
thread_local std::string g_err;
rbl::Engine& need(rbl_engine* e) {
  if (!e) throw std::runtime_error("null engine handle");
  return e->impl;
}
rbl::SelfPlay& need(rbl_selfplay* s) {
  if (!s) throw std::runtime_error("null self-play handle");
  return s->impl;
}
template <class F>
int guard(F&& f) {
  try {
    f();
    return 0;
  } catch (const std::exception& ex) {
    g_err = ex.what();
    return 1;
  } catch (...) {
    g_err = "unknown error";
    return 1;
  }
}
}  // namespace

extern "C" {

const char* rbl_last_error(void) { return g_err.c_str(); }

int rbl_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char* rbl_build_info(void) { return "librebel_hip gfx950 (HIP " __VERSION__ ")"; }

int rbl_num_actions(int dice, int faces) { return 1 + 2 * dice * faces; }
int rbl_num_hands(int dice, int faces) {
  int h = 1;
  for (int i = 0; i < dice; ++i) h *= faces;
  return h;
}
int rbl_query_size(int dice, int faces) { return 2 + rbl_num_actions(dice, faces) + 2 * rbl_num_hands(dice, faces); }

int rbl_unroll_tree(int dice, int faces, int root_last_bid, int root_player, int max_depth, int32_t* out, int cap_nodes) {
  int n = -1;
  guard([&] {
    rbl::Rules g(dice, faces);
    auto t = rbl::unroll_tree(g, root_last_bid, root_player, max_depth);
    n = (int)t.size();
    for (int i = 0; i < n && i < cap_nodes; ++i) {
      out[i * 6 + 0] = t[i].last_bid;
      out[i * 6 + 1] = t[i].player;
      out[i * 6 + 2] = t[i].cb;
      out[i * 6 + 3] = t[i].ce;
      out[i * 6 + 4] = t[i].parent;
      out[i * 6 + 5] = t[i].depth;
    }
  });
  return n;
}

rbl_engine* rbl_engine_create(int device, int dice, int faces, const rbl_params* params, int max_lanes) {
  rbl_engine* e = nullptr;
  guard([&] {
    if (!params) throw std::runtime_error("rbl_engine_create: params is null");
    e = new rbl_engine(device, dice, faces, *params, max_lanes);
  });
  return e;
}
void rbl_engine_destroy(rbl_engine* e) { delete e; }
void* rbl_engine_stream(rbl_engine* e) { return e ? (void*)e->impl.stream() : nullptr; }

int rbl_engine_set_net_zero(rbl_engine* e) { return guard([&] { need(e).set_net_zero(); }); }
int rbl_engine_set_net_synthetic(rbl_engine* e) { return guard([&] { need(e).set_net_synthetic(); }); }
int rbl_engine_set_net_mlp(rbl_engine* e, const rbl_mlp_weights* w) {
  return guard([&] {
    if (!w) throw std::runtime_error("rbl_engine_set_net_mlp: weights is null");
    need(e).set_net_mlp(*w);
  });
}
int rbl_engine_set_net_precision(rbl_engine* e, int mode) {
  return guard([&] { need(e).set_net_precision(mode); });
}
int rbl_engine_set_net_callback(rbl_engine* e, rbl_net_fn fn, void* user, int host_buffers) {
  return guard([&] { need(e).set_net_callback(fn, user, host_buffers != 0); });
}
int rbl_net_forward(rbl_engine* e, const float* queries, int64_t rows, float* out) {
  return guard([&] { need(e).net_forward_host(queries, rows, out); });
}
int rbl_net_forward_dev(rbl_engine* e, const float* queries_dev, int64_t rows, float* out_dev) {
  return guard([&] { need(e).net_forward_dev(queries_dev, rows, out_dev); });
}

int rbl_solver_reset(rbl_engine* e, int B, const int32_t* root_last_bid, const int32_t* root_player,
                     const double* beliefs, const int32_t* act_iteration) {
  return guard([&] { need(e).reset(B, root_last_bid, root_player, beliefs, act_iteration); });
}
int rbl_solver_step(rbl_engine* e, int traverser) { return guard([&] { need(e).step(traverser); }); }
int rbl_solver_multistep(rbl_engine* e, int n) { return guard([&] { need(e).multistep(n); }); }
int rbl_solver_sync(rbl_engine* e) { return guard([&] { need(e).sync(); }); }
int rbl_solver_num_lanes(rbl_engine* e) { return e ? e->impl.num_lanes() : -1; }
int rbl_solver_tree_size(rbl_engine* e, int lane) {
  int n = -1;
  guard([&] { n = need(e).tree_size(lane); });
  return n;
}
int64_t rbl_solver_total_rows(rbl_engine* e) { return e ? e->impl.total_rows() : -1; }
int rbl_solver_get(rbl_engine* e, int lane, int which, double* out) {
  return guard([&] { need(e).get(lane, which, out); });
}
int rbl_solver_set_strategy(rbl_engine* e, int lane, const double* strategy) {
  return guard([&] { need(e).set_strategy(lane, strategy); });
}
int rbl_solver_best_response(rbl_engine* e, int traverser, double* out) {
  return guard([&] { need(e).best_response(traverser, out); });
}
int rbl_strategy_recursive(rbl_engine* e, int to_leaf, double* out) {
  return guard([&] { rbl::strategy_recursive(need(e), to_leaf != 0, out); });
}

int rbl_strategy_recursive_sampled(rbl_engine* e, int seed, int root_only, double* out) {
  return guard([&] { rbl::strategy_recursive_sampled(need(e), seed, root_only != 0, out); });
}
int rbl_exploitability2(int device, int dice, int faces, const double* strategy, double out[2]) {
  return guard([&] {  // compute_exploitability2 (subgame_solving.cc:802-816): two full-tree BR sweeps, uniform beliefs
    rbl_params p{};
    p.num_iters = 1;
    p.max_depth = 1000000;
    p.use_cfr = 1;
    rbl::Engine e(device, dice, faces, p, 1);
    const int H = e.rules().H;
    std::vector<double> b(2 * H, 1. / H), v(H);
    const int32_t rb = -1, rp = 0;
    e.reset(1, &rb, &rp, b.data(), nullptr);
    e.set_strategy(0, strategy);
    for (int t = 0; t < 2; ++t) {
      e.best_response(t, v.data());
      double s = 0;
      for (int h = 0; h < H; ++h) s += v[h];  // vector_sum (util.h:87-90)
      out[t] = s / H;
    }
  });
}
int rbl_exploitability_recursive(rbl_engine* e, int shard, int n_shards, double out[2], double* top_values,
                                 int32_t* top_owner, double* stats) {
  return guard([&] { rbl::exploitability_recursive(need(e), shard, n_shards, out, top_values, top_owner, stats, 1); });
}
int rbl_exploitability_recursive_deal(rbl_engine* e, int shard, int n_shards, int deal_levels, double out[2],
                                      double* top_values, int32_t* top_owner, double* stats) {
  return guard([&] { rbl::exploitability_recursive(need(e), shard, n_shards, out, top_values, top_owner, stats, deal_levels); });
}
int64_t rbl_exploitability_top_nodes(int dice, int faces, int max_depth, int deal_levels) {
  int64_t n = -1;
  guard([&] { n = rbl::exploitability_top_nodes(rbl::Rules(dice, faces), max_depth, deal_levels); });
  return n;
}
int rbl_exploitability_combine(int dice, int faces, int max_depth, int deal_levels, int n_shards,
                               const double* const* top_values, const int32_t* top_owner, double out[2]) {
  return guard([&] {
    if (!out) throw std::runtime_error("rbl_exploitability_combine: null output");
    rbl::exploitability_combine(rbl::Rules(dice, faces), max_depth, deal_levels, n_shards, top_values, top_owner, out);
  });
}
int rbl_solver_evaluate(rbl_engine* e, int traverser, double* out) {
  return guard([&] { need(e).evaluate(traverser, out); });
}
int rbl_immediate_regrets(int device, int dice, int faces, const double* strategies, int n_strategies, double* out) {
  return guard([&] {  // compute_immediate_regrets (subgame_solving.cc:984-1050) on the full tree
    // For every strategy and traverser the reference sweeps the tree bottom-up under that strategy and accumulates
    // regrets[node][hand][action] += value(child) - value(node): exactly the regret update of one plain CFR step (no
    // discount) with sigma = the strategy.  So: a full-tree solver with use_cfr and no linear / DCFR weighting, sigma reset
    // to the strategy before each traverser's step (the step's regret matching overwrites the traverser's rows), regrets
    // left to accumulate on the device across all strategies; the maximum over the actions is taken at the end.
    if (!strategies || !out || n_strategies < 1) throw std::runtime_error("rbl_immediate_regrets: bad arguments");
    rbl_params p{};
    p.num_iters = 2 * n_strategies;
    p.max_depth = 1000000;
    p.use_cfr = 1;
    rbl::Engine e(device, dice, faces, p, 1);
    e.set_net_zero();
    const rbl::Rules& g = e.rules();
    const int H = g.H, A = g.A;
    const std::vector<rbl::Node> full = rbl::unroll_tree(g, -1, 0, 1000000);
    const size_t stride = full.size() * (size_t)H * A;
    std::vector<double> b(2 * H, 1. / H), reg(stride);
    const int32_t rb = -1, rp = 0;
    e.reset(1, &rb, &rp, b.data(), nullptr);
    for (int k = 0; k < n_strategies; ++k)
      for (int t = 0; t < 2; ++t) {
        e.set_strategy(0, strategies + (size_t)k * stride);
        e.step(t);
      }
    e.get(0, RBL_GET_REGRETS, reg.data());
    for (size_t n = 0; n < full.size(); ++n)
      for (int h = 0; h < H; ++h) {
        double best = 0.0;
        if (full[n].cb != full[n].ce) {
          // std::max_element over ALL actions of the dense row: illegal actions hold 0 (subgame_solving.cc:1041-1044)
          const double* r = reg.data() + (n * H + h) * A;
          best = r[0];
          for (int a = 1; a < A; ++a) best = r[a] > best ? r[a] : best;
          best /= n_strategies;
        }
        out[n * H + h] = best;
      }
  });
}
int rbl_ev2(int device, int dice, int faces, const double* strategy1, const double* strategy2, double out[2]) {
  return guard([&] {  // compute_ev2 (subgame_solving.cc:975-982): player 0 follows one strategy, player 1 the other
    rbl_params p{};
    p.num_iters = 1;
    p.max_depth = 1000000;
    p.use_cfr = 1;
    rbl::Engine e(device, dice, faces, p, 1);
    const rbl::Rules& g = e.rules();
    const int H = g.H, A = g.A;
    const std::vector<rbl::Node> full = rbl::unroll_tree(g, -1, 0, 1000000);
    std::vector<double> b(2 * H, 1. / H), v(H), mix(full.size() * (size_t)H * A);
    const int32_t rb = -1, rp = 0;
    e.reset(1, &rb, &rp, b.data(), nullptr);
    const double* s[2] = {strategy1, strategy2};
    for (int k = 0; k < 2; ++k) {
      for (size_t n = 0; n < full.size(); ++n) {
        const double* src = (full[n].player == 0 ? s[k] : s[1 - k]) + n * (size_t)H * A;
        std::copy(src, src + (size_t)H * A, mix.begin() + n * (size_t)H * A);
      }
      e.set_strategy(0, mix.data());
      e.evaluate(0, v.data());
      double sum = 0;
      for (int h = 0; h < H; ++h) sum += v[h];  // vector_sum (util.h:87-90)
      out[k] = k == 0 ? sum / H : -(sum / H);
    }
  });
}
int rbl_solver_get_snapshot(rbl_engine* e, int lane, double* out) {
  return guard([&] { need(e).get_snapshot(lane, out); });
}
int rbl_solver_hand_values(rbl_engine* e, int lane, int player, double* out) {
  return guard([&] { need(e).hand_values(lane, player, out); });
}
int rbl_solver_examples(rbl_engine* e, int lane, float* queries, float* values) {
  return guard([&] { need(e).examples(lane, queries, values); });
}
int rbl_solver_get_queries(rbl_engine* e, float* out) { return guard([&] { need(e).get_queries(out); }); }
int rbl_solver_debug_stamps(rbl_engine* e, long long* out) { return guard([&] { need(e).get_debug(out); }); }
int rbl_net_debug_stamps(rbl_engine* e, long long* out) { return guard([&] { need(e).get_net_debug(out); }); }

rbl_selfplay* rbl_selfplay_create(rbl_engine* e, int n_lanes, const int32_t* seeds, double random_action_prob,
                                  int sample_leaf) {
  rbl_selfplay* sp = nullptr;
  guard([&] { sp = new rbl_selfplay(&need(e), n_lanes, seeds, random_action_prob, sample_leaf != 0); });
  return sp;
}
void rbl_selfplay_destroy(rbl_selfplay* sp) { delete sp; }
int64_t rbl_selfplay_advance(rbl_selfplay* sp, rbl_example_fn sink, void* user) {
  int64_t n = -1;
  guard([&] { n = need(sp).advance(sink, user); });
  return n;
}
int rbl_selfplay_on_device(rbl_selfplay* sp) {
  int m = -1;
  guard([&] { m = need(sp).decide_mode(); });
  return m;
}
int rbl_selfplay_device_examples(rbl_selfplay* sp, const float** queries_dev, const float** values_dev) {
  return guard([&] {
    if (!queries_dev || !values_dev) throw std::runtime_error("rbl_selfplay_device_examples: null output pointer");
    need(sp).device_examples(queries_dev, values_dev);
  });
}
int rbl_selftest_device_rng(int device, int32_t seed, int rounds, int hi, const double* w, int nw, double* out) {
  return guard([&] {
    if (rounds < 1 || nw < 1 || !w || !out) throw std::runtime_error("rbl_selftest_device_rng: bad arguments");
    RBL_HIP_CHECK(hipSetDevice(device));
    std::vector<uint32_t> st(624);
    rbl::mt19937_seed_state((uint32_t)seed, st.data());
    rbl::DevBuf<uint32_t> d_mt;
    rbl::DevBuf<int> d_idx;
    rbl::DevBuf<double> d_w, d_out;
    d_mt.upload(st, nullptr);
    d_idx.upload(std::vector<int>(1, 624), nullptr);
    d_w.upload(std::vector<double>(w, w + nw), nullptr);
    d_out.alloc((size_t)3 * rounds);
    RBL_HIP_CHECK(hipDeviceSynchronize());
    rbl::launch_sp_rng_probe(d_mt.p, d_idx.p, 1, 0, rounds, hi, d_w.p, nw, d_out.p, nullptr);
    RBL_HIP_CHECK(hipGetLastError());
    RBL_HIP_CHECK(hipMemcpy(out, d_out.p, (size_t)3 * rounds * sizeof(double), hipMemcpyDeviceToHost));
  });
}
int64_t rbl_selfplay_games_finished(rbl_selfplay* sp) { return sp ? sp->impl.games_finished() : -1; }
int64_t rbl_selfplay_root_dedup_served(rbl_selfplay* sp) { return sp ? sp->impl.lanes_served_by_root_dedup() : -1; }
int rbl_selfplay_state(rbl_selfplay* sp, int lane, int32_t* last_bid, int32_t* player_id) {
  return guard([&] { need(sp).state(lane, last_bid, player_id); });
}

int rbl_engine_timing(rbl_engine* e, int stride) { return guard([&] { need(e).timing(stride); }); }
int rbl_engine_stats(rbl_engine* e, rbl_kernel_stats* out, int reset) {
  return guard([&] { need(e).stats(out, reset != 0); });
}

}  // extern "C"
