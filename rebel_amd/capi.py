"""ctypes binding of the C ABI in include/rebel_hip.h (rebel_amd/librebel_hip.so).

This is the thinnest possible host mirror: the parity tests (tests/ -m gpu) and bench.py call the HIP path through it,
so what they exercise is exactly the exported C symbols.  Loading fails loudly when the library is missing -- there is
no CPU fallback anywhere in this package.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("REBEL_HIP_LIB") or os.path.join(HERE, "librebel_hip.so")  # override: A/B builds

GET_AVERAGE, GET_LAST, GET_REGRETS, GET_SUM = 0, 1, 2, 3
GET_SAMPLED, GET_FINAL = 4, 5  # StreamSolver.get only

NET_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_float), C.c_int64, C.c_int64, C.POINTER(C.c_float), C.c_int64,
                     C.c_void_p)
EXAMPLE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_float), C.c_int64,
                         C.POINTER(C.c_float), C.c_int64)

# every symbol include/rebel_hip.h declares (tests/test_capi_symbols.py checks the header against this list and the .so)
SYMBOLS = [
    "rbl_last_error", "rbl_device_count", "rbl_build_info", "rbl_num_actions", "rbl_num_hands", "rbl_query_size",
    "rbl_unroll_tree", "rbl_engine_create", "rbl_engine_destroy", "rbl_engine_stream", "rbl_engine_set_net_zero",
    "rbl_engine_set_net_synthetic", "rbl_engine_set_net_mlp", "rbl_engine_set_net_precision", "rbl_engine_set_net_callback", "rbl_net_forward",
    "rbl_net_forward_dev", "rbl_solver_reset", "rbl_solver_step", "rbl_solver_multistep", "rbl_solver_sync",
    "rbl_solver_num_lanes", "rbl_solver_tree_size", "rbl_solver_total_rows", "rbl_solver_get",
    "rbl_solver_get_snapshot", "rbl_solver_set_strategy", "rbl_solver_best_response", "rbl_exploitability2", "rbl_ev2", "rbl_immediate_regrets", "rbl_solver_evaluate", "rbl_strategy_recursive", "rbl_strategy_recursive_sampled", "rbl_exploitability_recursive", "rbl_exploitability_recursive_deal", "rbl_exploitability_top_nodes", "rbl_exploitability_combine", "rbl_stream_create", "rbl_stream_destroy", "rbl_stream_num_nodes", "rbl_stream_step", "rbl_stream_exploitability", "rbl_stream_get", "rbl_stream_last_error", "rbl_stream_sampled_reset", "rbl_stream_sampled_add", "rbl_stream_sampled_add_root_only", "rbl_stream_regrets_reset", "rbl_stream_regrets_add", "rbl_stream_regrets_report", "rbl_stream_sampled_eval", "rbl_solver_hand_values", "rbl_solver_examples", "rbl_solver_get_queries", "rbl_solver_debug_stamps", "rbl_net_debug_stamps",
    "rbl_selfplay_create", "rbl_selfplay_destroy", "rbl_selfplay_advance", "rbl_selfplay_games_finished",
    "rbl_selfplay_root_dedup_served",
    "rbl_selfplay_state", "rbl_selfplay_on_device", "rbl_selfplay_device_examples", "rbl_selftest_device_rng",
    "rbl_engine_timing", "rbl_engine_stats",
]


class Params(C.Structure):
    """SubgameSolvingParams (/root/reference/csrc/liars_dice/subgame_solving.h:43-58)."""
    _fields_ = [("num_iters", C.c_int32), ("max_depth", C.c_int32), ("linear_update", C.c_int32),
                ("use_cfr", C.c_int32), ("optimistic", C.c_int32), ("dcfr", C.c_int32), ("dcfr_alpha", C.c_double),
                ("dcfr_beta", C.c_double), ("dcfr_gamma", C.c_double)]


class MlpWeights(C.Structure):
    _fields_ = [("n_layers", C.c_int32), ("n_in", C.c_int32), ("n_hidden", C.c_int32), ("n_out", C.c_int32),
                ("use_layer_norm", C.c_int32), ("w", C.POINTER(C.POINTER(C.c_float))),
                ("b", C.POINTER(C.POINTER(C.c_float))), ("ln_w", C.POINTER(C.POINTER(C.c_float))),
                ("ln_b", C.POINTER(C.POINTER(C.c_float))), ("w_out", C.POINTER(C.c_float)),
                ("b_out", C.POINTER(C.c_float)), ("ln_eps", C.c_float)]


class KernelStats(C.Structure):
    _fields_ = [("cfr_ms", C.c_double), ("net_ms", C.c_double), ("cfr_launches", C.c_int64),
                ("net_launches", C.c_int64), ("net_rows", C.c_int64), ("lane_steps", C.c_int64),
                ("cfr_bytes", C.c_double), ("net_flops", C.c_double), ("cfr_kernel", C.c_int32), ("net_kernel", C.c_int32),
                ("n_streams", C.c_int32), ("net_products", C.c_int32)]


CFR_KERNEL_NAMES = {0: "cfr_step_kernel (generic)", 1: "cfr_rows_kernel (one thread per tree row)",
                    2: "cfr_wave_kernel (one wavefront per lane)", 3: "cfr_rows_kernel<GS> (global state, 2dx6f)",
                    4: "cfr_flat_kernel (element-parallel, sigma in LDS, 2dx6f)"}
NET_KERNEL_NAMES = {0: "none", 3: "mlp_fsplit_forward_kernel (feature split)",
                    5: "mlp_resident_kernel (f16x2-split MFMA, register-resident weights)"}


def make_params(num_iters=10, max_depth=2, linear_update=False, use_cfr=False, optimistic=False, dcfr=False,
                dcfr_alpha=0.0, dcfr_beta=0.0, dcfr_gamma=0.0):
    return Params(num_iters, max_depth, int(linear_update), int(use_cfr), int(optimistic), int(dcfr), dcfr_alpha,
                  dcfr_beta, dcfr_gamma)


_lib = None


def lib():
    """Loads librebel_hip.so (once).  Raises if it has not been built: `make -C rebel_amd/csrc lib`."""
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own libamdhip64.so.7 (same SONAME as /opt/rocm's).  Whichever is loaded first serves the whole
    # process; if librebel_hip.so pulls in /opt/rocm's copy and torch is imported afterwards, torch finds "No HIP GPUs".
    # Importing torch first (when it is installed) makes the order deterministic; C callers are unaffected.
    import sys
    if "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing -- build it with `make -C rebel_amd/csrc lib` "
                           "(or __graft_entry__.build()); rebel_amd has no CPU fallback")
    L = C.CDLL(LIB_PATH)
    vp, i32p, dp, fp = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_float)
    sig = {
        "rbl_last_error": (C.c_char_p, []),
        "rbl_device_count": (C.c_int, []),
        "rbl_build_info": (C.c_char_p, []),
        "rbl_num_actions": (C.c_int, [C.c_int, C.c_int]),
        "rbl_num_hands": (C.c_int, [C.c_int, C.c_int]),
        "rbl_query_size": (C.c_int, [C.c_int, C.c_int]),
        "rbl_unroll_tree": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, i32p, C.c_int]),
        "rbl_engine_create": (vp, [C.c_int, C.c_int, C.c_int, C.POINTER(Params), C.c_int]),
        "rbl_engine_destroy": (None, [vp]),
        "rbl_engine_stream": (vp, [vp]),
        "rbl_engine_set_net_zero": (C.c_int, [vp]),
        "rbl_engine_set_net_synthetic": (C.c_int, [vp]),
        "rbl_engine_set_net_mlp": (C.c_int, [vp, C.POINTER(MlpWeights)]),
        "rbl_engine_set_net_callback": (C.c_int, [vp, NET_FN, vp, C.c_int]),
        "rbl_engine_set_net_precision": (C.c_int, [vp, C.c_int]),
        "rbl_net_forward": (C.c_int, [vp, fp, C.c_int64, fp]),
        "rbl_net_forward_dev": (C.c_int, [vp, vp, C.c_int64, vp]),
        "rbl_solver_reset": (C.c_int, [vp, C.c_int, i32p, i32p, dp, i32p]),
        "rbl_solver_step": (C.c_int, [vp, C.c_int]),
        "rbl_solver_multistep": (C.c_int, [vp, C.c_int]),
        "rbl_solver_sync": (C.c_int, [vp]),
        "rbl_solver_num_lanes": (C.c_int, [vp]),
        "rbl_solver_tree_size": (C.c_int, [vp, C.c_int]),
        "rbl_solver_total_rows": (C.c_int64, [vp]),
        "rbl_solver_get": (C.c_int, [vp, C.c_int, C.c_int, dp]),
        "rbl_solver_get_snapshot": (C.c_int, [vp, C.c_int, dp]),
        "rbl_solver_set_strategy": (C.c_int, [vp, C.c_int, dp]),
        "rbl_solver_best_response": (C.c_int, [vp, C.c_int, dp]),
        "rbl_exploitability2": (C.c_int, [C.c_int, C.c_int, C.c_int, dp, dp]),
        "rbl_strategy_recursive": (C.c_int, [vp, C.c_int, dp]),
        "rbl_solver_evaluate": (C.c_int, [vp, C.c_int, dp]),
        "rbl_ev2": (C.c_int, [C.c_int, C.c_int, C.c_int, dp, dp, dp]),
        "rbl_immediate_regrets": (C.c_int, [C.c_int, C.c_int, C.c_int, dp, C.c_int, dp]),
        "rbl_strategy_recursive_sampled": (C.c_int, [vp, C.c_int, C.c_int, dp]),
        "rbl_solver_hand_values": (C.c_int, [vp, C.c_int, C.c_int, dp]),
        "rbl_solver_examples": (C.c_int, [vp, C.c_int, fp, fp]),
        "rbl_solver_get_queries": (C.c_int, [vp, fp]),
        "rbl_solver_debug_stamps": (C.c_int, [vp, C.POINTER(C.c_longlong)]),
        "rbl_net_debug_stamps": (C.c_int, [vp, C.POINTER(C.c_longlong)]),
        "rbl_exploitability_recursive": (C.c_int, [vp, C.c_int, C.c_int, dp, dp, i32p, dp]),
        "rbl_exploitability_recursive_deal": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, dp, dp, i32p, dp]),
        "rbl_exploitability_top_nodes": (C.c_int64, [C.c_int, C.c_int, C.c_int, C.c_int]),
        "rbl_exploitability_combine": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(dp), i32p, dp]),
        "rbl_stream_create": (vp, [C.c_int, C.c_int, C.c_int, C.POINTER(Params)]),
        "rbl_stream_destroy": (None, [vp]),
        "rbl_stream_num_nodes": (C.c_int64, [vp]),
        "rbl_stream_step": (C.c_int, [vp, C.c_int]),
        "rbl_stream_exploitability": (C.c_int, [vp, dp]),
        "rbl_stream_get": (C.c_int, [vp, C.c_int, dp]),
        "rbl_stream_last_error": (C.c_char_p, []),
        "rbl_stream_sampled_reset": (C.c_int, [vp]),
        "rbl_stream_sampled_add": (C.c_int, [vp, vp, C.c_int]),
        "rbl_stream_sampled_add_root_only": (C.c_int, [vp, vp, C.c_int]),
        "rbl_stream_regrets_reset": (C.c_int, [vp]),
        "rbl_stream_regrets_add": (C.c_int, [vp, C.c_int]),
        "rbl_stream_regrets_report": (C.c_int, [vp, C.c_int, C.c_int, dp, dp]),
        "rbl_stream_sampled_eval": (C.c_int, [vp, dp, dp]),
        "rbl_selfplay_create": (vp, [vp, C.c_int, i32p, C.c_double, C.c_int]),
        "rbl_selfplay_destroy": (None, [vp]),
        "rbl_selfplay_advance": (C.c_int64, [vp, EXAMPLE_FN, vp]),
        "rbl_selfplay_games_finished": (C.c_int64, [vp]),
        "rbl_selfplay_root_dedup_served": (C.c_int64, [vp]),
        "rbl_selfplay_state": (C.c_int, [vp, C.c_int, i32p, i32p]),
        "rbl_selfplay_on_device": (C.c_int, [vp]),
        "rbl_selfplay_device_examples": (C.c_int, [vp, C.POINTER(fp), C.POINTER(fp)]),
        "rbl_selftest_device_rng": (C.c_int, [C.c_int, C.c_int32, C.c_int, C.c_int, dp, C.c_int, dp]),
        "rbl_engine_timing": (C.c_int, [vp, C.c_int]),
        "rbl_engine_stats": (C.c_int, [vp, C.POINTER(KernelStats), C.c_int]),
    }
    assert sorted(sig) == sorted(SYMBOLS)
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


class RebelError(RuntimeError):
    pass


def _check(status):
    if status != 0:
        raise RebelError(lib().rbl_last_error().decode())


def _i32(a):
    return np.ascontiguousarray(a, np.int32)


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def device_count():
    return lib().rbl_device_count()


def unroll_tree(d, f, root_last_bid=-1, root_player=0, max_depth=2):
    L = lib()
    n = L.rbl_unroll_tree(d, f, root_last_bid, root_player, max_depth, None, 0)
    out = np.zeros((n, 6), np.int32)
    L.rbl_unroll_tree(d, f, root_last_bid, root_player, max_depth, _p(out, C.c_int32), n)
    return out


def exploitability2(dice, faces, strategy, device=0):
    """compute_exploitability2 (subgame_solving.cc:802-816) of a dense full-tree strategy, on the GPU."""
    s = np.ascontiguousarray(strategy, np.float64)
    out = np.zeros(2)
    _check(lib().rbl_exploitability2(device, dice, faces, _p(s, C.c_double), _p(out, C.c_double)))
    return out


def ev2(dice, faces, strategy1, strategy2, device=0):
    """compute_ev2 (subgame_solving.cc:975-982) of two dense full-tree strategies, on the GPU."""
    a = np.ascontiguousarray(strategy1, np.float64)
    b = np.ascontiguousarray(strategy2, np.float64)
    out = np.zeros(2)
    _check(lib().rbl_ev2(device, dice, faces, _p(a, C.c_double), _p(b, C.c_double), _p(out, C.c_double)))
    return out


def combine_exploitability(dice, faces, max_depth, tops, deal_levels=1):
    """Shards of rbl_exploitability_recursive[_deal] -> the two exploitabilities: rbl_exploitability_combine (host code of the
    C ABI; BRSolver::compute_br, subgame_solving.cc:326-355, over the nodes of depth <= deal_levels * max_depth with every
    dealt node's value taken from its owner).  tops[s] = (top_values, top_owner) of shard s."""
    L = lib()
    vals = [np.ascontiguousarray(t[0], np.float64) for t in tops]
    owner = np.ascontiguousarray(tops[0][1], np.int32)
    M = L.rbl_exploitability_top_nodes(dice, faces, max_depth, deal_levels)
    if M < 0:
        raise RebelError(L.rbl_last_error().decode())
    if any(v.shape[:2] != (2, M) for v in vals) or owner.shape != (M,):
        raise ValueError(f"combine_exploitability: top arrays must be [2][{M}][H] / [{M}]")
    ptrs = (C.POINTER(C.c_double) * len(vals))(*[_p(v, C.c_double) for v in vals])
    out = np.zeros(2)
    _check(L.rbl_exploitability_combine(dice, faces, max_depth, deal_levels, len(vals), ptrs, _p(owner, C.c_int32),
                                        _p(out, C.c_double)))
    return out


def immediate_regrets(dice, faces, strategies, device=0):
    """compute_immediate_regrets (subgame_solving.cc:984-1050): strategies float64[K, N_full, H, A] -> float64[N_full, H]."""
    s = np.ascontiguousarray(strategies, np.float64)
    K, N, H = s.shape[0], s.shape[1], s.shape[2]
    out = np.zeros((N, H))
    _check(lib().rbl_immediate_regrets(device, dice, faces, _p(s, C.c_double), K, _p(out, C.c_double)))
    return out


class Engine:
    """One GPU x one game x one SubgameSolvingParams; B lanes advanced in lock-step (see include/rebel_hip.h)."""

    def __init__(self, dice, faces, params, max_lanes=1, device=0):
        L = self.L = lib()
        self.dice, self.faces = dice, faces
        self.A, self.H, self.Q = L.rbl_num_actions(dice, faces), L.rbl_num_hands(dice, faces), L.rbl_query_size(dice, faces)
        self.params = params
        self.h = L.rbl_engine_create(device, dice, faces, C.byref(params), max_lanes)
        if not self.h:
            raise RebelError(L.rbl_last_error().decode())
        self._keep = []

    def close(self):
        if getattr(self, "h", None):
            self.L.rbl_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- value net
    def set_net_zero(self):
        _check(self.L.rbl_engine_set_net_zero(self.h))

    def set_net_synthetic(self):
        _check(self.L.rbl_engine_set_net_synthetic(self.h))

    def set_net_callback(self, fn):
        """fn(queries float32[rows, Q]) -> float32[rows, H], called on host buffers."""
        H = self.H

        def _cb(_u, q, rows, qs, out, osz, _stream):
            qa = np.ctypeslib.as_array(q, (rows, qs))
            res = np.ascontiguousarray(fn(qa.copy()), np.float32)
            assert res.shape == (rows, osz) and osz == H, (res.shape, rows, osz)
            np.ctypeslib.as_array(out, (rows, osz))[...] = res

        cb = NET_FN(_cb)
        self._keep.append(cb)
        _check(self.L.rbl_engine_set_net_callback(self.h, cb, None, 1))

    def set_net_precision(self, mode):
        """0 f32 parity (default), 1 half activations, 2 half activations and weights; applied by the next set_net_mlp."""
        _check(self.L.rbl_engine_set_net_precision(self.h, int(mode)))

    def set_net_mlp(self, layers, ln, w_out, b_out, ln_eps=1e-5):
        """layers: [(W [hid,in], b [hid])...]; ln: [(g [hid], beta [hid])...] or None; torch Linear layout."""
        f32 = lambda a: np.ascontiguousarray(a, np.float32)
        ws = [f32(w) for w, _ in layers]
        bs = [f32(b) for _, b in layers]
        n = len(layers)
        PP = C.POINTER(C.c_float) * n
        fpp = lambda arrs: PP(*[_p(a, C.c_float) for a in arrs])
        keep = [ws, bs]
        mw = MlpWeights()
        mw.n_layers, mw.n_in, mw.n_hidden = n, ws[0].shape[1], ws[0].shape[0]
        w_out, b_out = f32(w_out), f32(b_out)
        mw.n_out = w_out.shape[0]
        mw.use_layer_norm = int(ln is not None)
        wp, bp = fpp(ws), fpp(bs)
        mw.w, mw.b = C.cast(wp, C.POINTER(C.POINTER(C.c_float))), C.cast(bp, C.POINTER(C.POINTER(C.c_float)))
        keep += [wp, bp, w_out, b_out]
        if ln is not None:
            gs, os_ = [f32(g) for g, _ in ln], [f32(o) for _, o in ln]
            gp, op = fpp(gs), fpp(os_)
            mw.ln_w, mw.ln_b = C.cast(gp, C.POINTER(C.POINTER(C.c_float))), C.cast(op, C.POINTER(C.POINTER(C.c_float)))
            keep += [gs, os_, gp, op]
        mw.w_out, mw.b_out, mw.ln_eps = _p(w_out, C.c_float), _p(b_out, C.c_float), ln_eps
        _check(self.L.rbl_engine_set_net_mlp(self.h, C.byref(mw)))

    def net_forward(self, queries):
        q = np.ascontiguousarray(queries, np.float32)
        out = np.zeros((q.shape[0], self.H), np.float32)
        _check(self.L.rbl_net_forward(self.h, _p(q, C.c_float), q.shape[0], _p(out, C.c_float)))
        return out

    # ---- batched solver
    def reset(self, root_last_bid, root_player, beliefs, act_iteration=None):
        rb, rp = _i32(root_last_bid), _i32(root_player)
        B = len(rb)
        b = np.ascontiguousarray(beliefs, np.float64).reshape(B, 2, self.H)
        act = _i32(act_iteration) if act_iteration is not None else None
        _check(self.L.rbl_solver_reset(self.h, B, _p(rb, C.c_int32), _p(rp, C.c_int32), _p(b, C.c_double),
                                       _p(act, C.c_int32) if act is not None else None))
        self.B = B

    def step(self, traverser):
        _check(self.L.rbl_solver_step(self.h, traverser))

    def multistep(self, n=-1):
        _check(self.L.rbl_solver_multistep(self.h, n))

    def sync(self):
        _check(self.L.rbl_solver_sync(self.h))

    def tree_size(self, lane):
        return self.L.rbl_solver_tree_size(self.h, lane)

    def total_rows(self):
        return self.L.rbl_solver_total_rows(self.h)

    def get(self, lane, which):
        out = np.zeros((self.tree_size(lane), self.H, self.A))
        _check(self.L.rbl_solver_get(self.h, lane, which, _p(out, C.c_double)))
        return out

    def get_snapshot(self, lane):
        out = np.zeros((self.tree_size(lane), self.H, self.A))
        _check(self.L.rbl_solver_get_snapshot(self.h, lane, _p(out, C.c_double)))
        return out

    def set_strategy(self, lane, dense):
        d = np.ascontiguousarray(dense, np.float64)
        assert d.shape == (self.tree_size(lane), self.H, self.A), d.shape
        _check(self.L.rbl_solver_set_strategy(self.h, lane, _p(d, C.c_double)))

    def best_response(self, traverser):
        out = np.zeros((self.B, self.H))
        _check(self.L.rbl_solver_best_response(self.h, traverser, _p(out, C.c_double)))
        return out

    def evaluate(self, traverser):
        """compute_ev per lane: root values [B][H] of the traverser following the lane's current sigma."""
        out = np.zeros((self.B, self.H))
        _check(self.L.rbl_solver_evaluate(self.h, traverser, _p(out, C.c_double)))
        return out

    def strategy_recursive(self, to_leaf=False):
        """compute_strategy_recursive(_to_leaf) with this engine's params and net -> dense [N_full][H][A]."""
        n = len(unroll_tree(self.dice, self.faces, -1, 0, 1000000))
        out = np.zeros((n, self.H, self.A))
        _check(self.L.rbl_strategy_recursive(self.h, int(to_leaf), _p(out, C.c_double)))
        return out

    def strategy_recursive_sampled(self, seed, root_only=False):
        """compute_sampled_strategy_recursive_to_leaf(seed, root_only) with this engine's params and net."""
        n = len(unroll_tree(self.dice, self.faces, -1, 0, 1000000))
        out = np.zeros((n, self.H, self.A))
        _check(self.L.rbl_strategy_recursive_sampled(self.h, int(seed), int(root_only), _p(out, C.c_double)))
        return out

    def exploitability_recursive(self, shard=0, n_shards=1, max_depth=None, deal_levels=1):
        """rbl_exploitability_recursive_deal: (exploitabilities[2], (top_values[2][M][H], top_owner[M]), stats dict); the
        full-tree strategy never leaves the device.  max_depth = the engine's params.max_depth (sizes the top levels);
        deal_levels: which recursion level's frontier the shards share out (include/rebel_hip.h)."""
        if max_depth is None:
            max_depth = self.params.max_depth
        M = self.L.rbl_exploitability_top_nodes(self.dice, self.faces, max_depth, deal_levels)
        if M < 0:
            raise RebelError(self.L.rbl_last_error().decode())
        out = np.zeros(2)
        tv = np.zeros((2, M, self.H))
        own = np.full(M, -1, np.int32)
        st = np.zeros(8)
        _check(self.L.rbl_exploitability_recursive_deal(self.h, int(shard), int(n_shards), int(deal_levels), _p(out, C.c_double),
                                                        _p(tv, C.c_double), _p(own, C.c_int32), _p(st, C.c_double)))
        keys = ("nodes", "subgames", "levels", "solve_s", "sweep_s", "strategy_bytes", "frontier_items", "top_nodes")
        return out, (tv, own), dict(zip(keys, st))

    def hand_values(self, lane, player):
        out = np.zeros(self.H)
        _check(self.L.rbl_solver_hand_values(self.h, lane, player, _p(out, C.c_double)))
        return out

    def examples(self, lane):
        q, v = np.zeros((2, self.Q), np.float32), np.zeros((2, self.H), np.float32)
        _check(self.L.rbl_solver_examples(self.h, lane, _p(q, C.c_float), _p(v, C.c_float)))
        return q, v

    def queries(self):
        out = np.zeros((self.total_rows(), self.Q), np.float32)
        _check(self.L.rbl_solver_get_queries(self.h, _p(out, C.c_float)))
        return out

    def debug_stamps(self):
        out = np.zeros((self.B, 16), np.int64)
        _check(self.L.rbl_solver_debug_stamps(self.h, _p(out, C.c_longlong)))
        return out

    def net_debug_stamps(self):
        out = np.zeros((1024, 16), np.int64)
        _check(self.L.rbl_net_debug_stamps(self.h, _p(out, C.c_longlong)))
        return out

    # ---- accounting
    def timing(self, stride=1):
        """0 = off; n = bracket the kernels of every n-th CFR iteration with HIP events (engine stream)."""
        _check(self.L.rbl_engine_timing(self.h, int(stride)))

    def stats(self, reset=False):
        s = KernelStats()
        _check(self.L.rbl_engine_stats(self.h, C.byref(s), int(reset)))
        return {k: getattr(s, k) for k, _ in KernelStats._fields_}


class StreamSolver:
    """Full-tree CFR with edge-indexed state in HBM (rbl_stream_*): the reference tool's "Solving the game for the full
    tree" at sizes whose dense TreeStrategy does not fit."""

    def __init__(self, dice, faces, params, device=0):
        self.L = lib()
        self.dice, self.faces = dice, faces
        self.A, self.H = self.L.rbl_num_actions(dice, faces), self.L.rbl_num_hands(dice, faces)
        self.h = self.L.rbl_stream_create(device, dice, faces, C.byref(params))
        if not self.h:
            raise RebelError((self.L.rbl_stream_last_error() or b"rbl_stream_create failed").decode())
        self.nodes = int(self.L.rbl_stream_num_nodes(self.h))

    def _ck(self, status):
        if status != 0:
            raise RebelError((self.L.rbl_stream_last_error() or b"?").decode())

    def step(self, n=1):
        self._ck(self.L.rbl_stream_step(self.h, int(n)))

    def exploitability(self):
        out = np.zeros(2)
        self._ck(self.L.rbl_stream_exploitability(self.h, _p(out, C.c_double)))
        return out

    def get(self, which):
        out = np.zeros((self.nodes, self.H, self.A))
        self._ck(self.L.rbl_stream_get(self.h, int(which), _p(out, C.c_double)))
        return out

    def sampled_reset(self):
        self._ck(self.L.rbl_stream_sampled_reset(self.h))

    def regrets_reset(self):
        self._ck(self.L.rbl_stream_regrets_reset(self.h))

    def regrets_add(self, which):
        """One more strategy of report_regrets' list: GET_LAST (the full-tree solver's sampling strategy) or GET_SAMPLED."""
        self._ck(self.L.rbl_stream_regrets_add(self.h, int(which)))

    def regrets_report(self, depth, n_first=20):
        """-> (immediate regrets [min(n_first, N)][H], (sum over depth < `depth`, sum over the rest))."""
        nf = int(min(n_first, self.nodes))
        first, sums = np.zeros((nf, self.H)), np.zeros(2)
        self._ck(self.L.rbl_stream_regrets_report(self.h, int(depth), nf, _p(first, C.c_double), _p(sums, C.c_double)))
        return first, (float(sums[0]), float(sums[1]))

    def sampled_add(self, engine, seed, root_only=False):
        """One repeat of compute_sampled_strategy_recursive_to_leaf(seed, root_only) on `engine`'s lanes (root_only: the subgames
        below the root subgame solved to the end of the game, as a forest on this solver's arrays)."""
        fn = self.L.rbl_stream_sampled_add_root_only if root_only else self.L.rbl_stream_sampled_add
        self._ck(fn(self.h, engine.h, int(seed)))

    def sampled_eval(self):
        """(exploitability2 of the mean of the repeats, compute_ev2(full-tree average, mean of the repeats))."""
        ex, ev = np.zeros(2), np.zeros(2)
        self._ck(self.L.rbl_stream_sampled_eval(self.h, _p(ex, C.c_double), _p(ev, C.c_double)))
        return ex, ev

    def close(self):
        if self.h:
            self.L.rbl_stream_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def device_rng_draws(seed, rounds, hi, weights, device=0):
    """The GPU's restatement of libstdc++ <random> (selfplay_kernels.hip): per round uniform_int(0, hi), canonical float,
    discrete(weights) from mt19937(seed) -> float64[rounds, 3]."""
    w = np.ascontiguousarray(weights, np.float64)
    out = np.zeros((rounds, 3))
    _check(lib().rbl_selftest_device_rng(device, seed, rounds, hi, _p(w, C.c_double), len(w), _p(out, C.c_double)))
    return out


class SelfPlay:
    """RlRunner lanes (recursive_solving.h:40-86) on an Engine; advance() = one subgame per lane."""

    def __init__(self, engine, seeds, random_action_prob=0.25, sample_leaf=True):
        self.e = engine
        s = _i32(seeds)
        self.n = len(s)
        self.h = engine.L.rbl_selfplay_create(engine.h, self.n, _p(s, C.c_int32), float(random_action_prob),
                                              int(sample_leaf))
        if not self.h:
            raise RebelError(engine.L.rbl_last_error().decode())

    def advance(self, collect=True):
        """-> (subgame-CFR-iterations executed, lanes int32[2n], queries f32[2n,Q], values f32[2n,H])"""
        got = []

        def _sink(_u, n, lanes, q, qs, v, vs):
            if collect:
                got.append((np.ctypeslib.as_array(lanes, (n,)).copy(), np.ctypeslib.as_array(q, (n, qs)).copy(),
                            np.ctypeslib.as_array(v, (n, vs)).copy()))

        cb = EXAMPLE_FN(_sink)
        n = self.e.L.rbl_selfplay_advance(self.h, cb, None)
        if n < 0:
            raise RebelError(self.e.L.rbl_last_error().decode())
        return (n,) + (got[0] if got else (None, None, None))

    def games_finished(self):
        return self.e.L.rbl_selfplay_games_finished(self.h)

    def root_dedup_served(self):
        """lane-epochs served by the epoch's representative root solve so far (REBEL_AMD_ROOT_DEDUP=1; 0 when off)"""
        return self.e.L.rbl_selfplay_root_dedup_served(self.h)

    def on_device(self):
        """1: the sampling walk runs as HIP kernels; 0: on the host (callback net / RBL_SELFPLAY_HOST=1).  Decides on first use."""
        return self.e.L.rbl_selfplay_on_device(self.h)

    def state(self, lane):
        a, b = C.c_int32(), C.c_int32()
        _check(self.e.L.rbl_selfplay_state(self.h, lane, C.byref(a), C.byref(b)))
        return a.value, b.value

    def close(self):
        if getattr(self, "h", None):
            self.e.L.rbl_selfplay_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
