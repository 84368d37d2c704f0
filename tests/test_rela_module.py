"""CPU tests of the pybind11 drop-in `rebel_amd.rela` (no GPU needed): the Python-visible surface of the reference's
`cfvpy.rela` (rela/pybind.cc:119-213) and the replay buffer's semantics, checked against the REFERENCE module itself
(oracle/_ref/rela*.so, compiled unmodified) wherever it is available: same seeds + same contents => same batches.
"""
import glob
import importlib.util
import os
import threading
import time

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ours():
    import rebel_amd.rela as m

    return m


@pytest.fixture(scope="module")
def ref_rela():
    paths = glob.glob(os.path.join(ROOT, "oracle", "_ref", "rela*.so"))
    if not paths:
        pytest.skip("reference rela module not built (oracle/_ref)")
    spec = importlib.util.spec_from_file_location("rela", paths[0])
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


SURFACE = ["Context", "DataThreadLoop", "ModelLocker", "RecursiveSolvingParams", "SubgameSolvingParams", "ThreadLoop",
           "ValuePrioritizedReplay", "ValueTransition", "compute_exploitability_fp", "compute_exploitability_with_net",
           "compute_stats_with_net", "create_cfr_thread"]


def test_surface_matches_reference(ours, ref_rela):
    pub = lambda m: sorted(n for n in dir(m) if not n.startswith("_"))
    assert pub(ours) == pub(ref_rela) == sorted(SURFACE)
    for cls in ("SubgameSolvingParams", "RecursiveSolvingParams", "ValuePrioritizedReplay", "Context", "ModelLocker",
                "ValueTransition"):
        a = sorted(n for n in dir(getattr(ours, cls)) if not n.startswith("_"))
        b = sorted(n for n in dir(getattr(ref_rela, cls)) if not n.startswith("_"))
        assert a == b, cls


def test_every_signature_matches_the_reference_module(ours, ref_rela):
    """Beyond names: the pybind11 signature line of every public function, constructor and method (argument names, order, types,
    return type) equals the compiled reference module's, module prefix aside -- so every positional and every keyword call the
    unmodified trainer makes (cfvpy/selfplay.py:182-260: `ValuePrioritizedReplay(capacity=..., seed=..., alpha=..., beta=...,
    prefetch=..., use_priority=..., compressed_values=...)`, `create_cfr_thread(model_locker, replay, cfg, seed)` ...) binds the
    same way.  The one exception is DataThreadLoop's constructor: the reference exposes it with an argument type it never binds
    (`rela::CVNetBufferConnector`), so it cannot be called from Python there either; ours has none."""
    def sigs(m, prefix):
        out = {}
        for n in dir(m):
            if n.startswith("_"):
                continue
            o = getattr(m, n)
            members = [(n, o)] if not isinstance(o, type) else [
                (f"{n}.{k}", getattr(o, k)) for k in dir(o) if k == "__init__" or not k.startswith("_")]
            for name, f in members:
                doc = (getattr(f, "__doc__", None) or "").strip().splitlines()
                out[name] = doc[0].replace(prefix, "rela.") if doc else ""
        return out

    a, b = sigs(ours, "rebel_amd.rela."), sigs(ref_rela, "rela.")
    assert set(a) == set(b)
    for name in sorted(a):
        if name == "DataThreadLoop.__init__":
            assert "CVNetBufferConnector" in b[name]
            continue
        assert a[name] == b[name], name
    assert "capacity" in a["ValuePrioritizedReplay.__init__"] and "model_locker" in a["create_cfr_thread"]


def test_every_call_of_the_unmodified_trainer_binds(ours):
    """tests/golden/trainer_rela_calls.json lists every call cfvpy/selfplay.py and cfvpy/utils.py make into `cfvpy.rela` (taken from
    their syntax trees by tests/golden/make_trainer_calls.py: callee, positional count, keyword names incl. the keys of the
    `replay_params` dictionary splatted into the replay's constructor).  Each must bind against this module's signature: enough
    parameters for the positionals, every keyword a parameter name not already taken -- the static half of "the unmodified
    trainer drives the module" (the dynamic half, the same call sequence on a GPU, is tests/test_rela_gpu.py)."""
    import json
    import re

    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "trainer_rela_calls.json")))
    assert len(fx["calls"]) >= 30

    def params(callee):
        obj = ours
        for part in callee.split("."):
            obj = getattr(obj, part)
        f = obj.__init__ if isinstance(obj, type) else obj
        sig = f.__doc__.strip().splitlines()[0]
        inner = sig[sig.index("(") + 1:sig.rindex(") ->")]
        names = [a.split(":")[0].strip() for a in re.split(r",\s*(?![^\[]*\])", inner) if a.strip()]
        return names[1:] if names and names[0] == "self" else names

    seen = set()
    for c in fx["calls"]:
        names = params(c["callee"])
        kw = c["keywords"] + c["star_kwargs_keys"]
        assert c["n_positional"] <= len(names), c
        free = names[c["n_positional"]:]
        assert all(k in free for k in kw) and len(set(kw)) == len(kw), (c, names)
        if c["callee"] in ("ValuePrioritizedReplay", "create_cfr_thread", "ModelLocker", "ValuePrioritizedReplay.load"):
            assert c["n_positional"] + len(kw) == len(names), (c, names)  # no defaults in the reference's bindings: all given
        seen.add(c["callee"])
    assert {"ModelLocker", "ValuePrioritizedReplay", "create_cfr_thread", "Context.push_env_thread", "Context.start",
            "Context.terminate", "ModelLocker.update_model", "ValuePrioritizedReplay.sample", "ValuePrioritizedReplay.pop_until",
            "ValuePrioritizedReplay.load", "ValuePrioritizedReplay.save", "compute_stats_with_net", "RecursiveSolvingParams"} <= seen
    # cfvpy/utils.py subclasses rela.Context (TimedContext): the class must be subclassable from Python
    assert [b["base"] for b in fx["subclasses"]] == ["Context"]

    class Sub(ours.Context):
        pass

    assert isinstance(Sub(), ours.Context)


def test_param_defaults_and_nested_setattr(ours):
    sp = ours.SubgameSolvingParams()  # subgame_solving.h:43-58
    assert (sp.num_iters, sp.max_depth, sp.linear_update, sp.use_cfr, sp.optimistic, sp.dcfr) == (10, 2, False, False,
                                                                                                 False, False)
    assert (sp.dcfr_alpha, sp.dcfr_beta, sp.dcfr_gamma) == (0, 0, 0)
    rp = ours.RecursiveSolvingParams()  # recursive_solving.h:31-38
    assert rp.random_action_prob == 1.0 and rp.sample_leaf is False
    rp.subgame_params.num_iters = 1024  # selfplay.py:604-605 relies on the nested reference sticking
    rp.subgame_params.use_cfr = True
    assert rp.subgame_params.num_iters == 1024 and rp.subgame_params.use_cfr
    assert not hasattr(rp, "no_such_key")  # create_mdp_config raises on unknown keys (selfplay.py:599-603)


def _fill(replay, n, Q=27, H=6, seed=0, weights=None):
    g = torch.Generator().manual_seed(seed)
    q = torch.rand(n, Q, generator=g)
    v = torch.rand(n, H, generator=g)
    w = torch.ones(n) if weights is None else weights
    replay.push([q, v, w])
    return q, v, w


@pytest.mark.parametrize("use_priority", [False, True])
def test_replay_sampling_identical_to_reference(ours, ref_rela, use_priority):
    kw = dict(capacity=400, seed=77, alpha=0.7, beta=0.4, prefetch=0, use_priority=use_priority, compressed_values=False)
    a, b = ours.ValuePrioritizedReplay(**kw), ref_rela.ValuePrioritizedReplay(**kw)
    w = torch.rand(450, generator=torch.Generator().manual_seed(5)) + 0.1
    for r in (a, b):
        _fill(r, 450, weights=w)
    assert a.size() == b.size() == 450 and a.num_add() == b.num_add() == 450
    for it in range(6):
        (ta, wa), (tb, wb) = a.sample(32, "cpu"), b.sample(32, "cpu")
        assert torch.equal(ta.query, tb.query) and torch.equal(ta.values, tb.values), it
        assert torch.allclose(wa, wb, rtol=1e-6, atol=0), it
        assert a.size() == b.size()  # trimmed to capacity after the first sample (prioritized_replay.h:429-433)
        pr = torch.rand(32, generator=torch.Generator().manual_seed(it)) + 0.05
        a.update_priority(pr if use_priority else torch.zeros(0))
        b.update_priority(pr if use_priority else torch.zeros(0))
    assert a.size() == 400


def test_replay_requires_priority_update(ours):
    r = ours.ValuePrioritizedReplay(capacity=64, seed=1, alpha=1.0, beta=0.4, prefetch=0, use_priority=True,
                                    compressed_values=False)
    _fill(r, 64)
    r.sample(8, "cpu")
    with pytest.raises(RuntimeError):
        r.sample(8, "cpu")  # the reference asserts here (prioritized_replay.h:265-271)


def test_replay_file_format_interchange(ours, ref_rela, tmp_path):
    kw = dict(capacity=64, seed=3, alpha=1.0, beta=0.4, prefetch=0, use_priority=False, compressed_values=False)
    a, b = ours.ValuePrioritizedReplay(**kw), ref_rela.ValuePrioritizedReplay(**kw)
    for r in (a, b):
        _fill(r, 50, Q=19, H=4)
    pa, pb = str(tmp_path / "a.bin"), str(tmp_path / "b.bin")
    a.save(pa)
    b.save(pb)
    assert open(pa, "rb").read() == open(pb, "rb").read()  # [int qn][int vn][qn f32][vn f32] per record (types.cc:87-111)
    c, d = ours.ValuePrioritizedReplay(**kw), ref_rela.ValuePrioritizedReplay(**kw)
    c.load(pb, 1.0, -1, 2)
    d.load(pa, 1.0, -1, 2)
    assert c.size() == d.size() == 25
    (tc, _), (td, _) = c.sample(16, "cpu"), d.sample(16, "cpu")
    assert torch.equal(tc.query, td.query) and torch.equal(tc.values, td.values)


def test_replay_extract_push_roundtrip(ours, ref_rela):
    kw = dict(capacity=64, seed=3, alpha=0.5, beta=0.4, prefetch=0, use_priority=True, compressed_values=False)
    a, b = ours.ValuePrioritizedReplay(**kw), ref_rela.ValuePrioritizedReplay(**kw)
    w = torch.linspace(0.5, 2.0, 40)
    for r in (a, b):
        _fill(r, 40, weights=w)
    ea, eb = a.extract(), b.extract()
    assert a.size() == b.size() == 0
    for x, y in zip(ea, eb):
        assert torch.allclose(x, y, rtol=1e-6, atol=0)
    assert torch.allclose(ea[2], w, rtol=1e-5)  # priorities come back un-exponentiated (prioritized_replay.h:342)
    a.push(ea)
    assert a.size() == 40


def test_replay_add_blocks_until_sampled(ours):
    """1.25x ring: producers stall when it is full until a sample trims it to `capacity` (prioritized_replay.h:64-65)."""
    r = ours.ValuePrioritizedReplay(capacity=8, seed=1, alpha=1.0, beta=0.4, prefetch=0, use_priority=False,
                                    compressed_values=False)
    _fill(r, 10, Q=5, H=2)
    done = threading.Event()

    def producer():
        _fill(r, 1, Q=5, H=2)
        done.set()

    t = threading.Thread(target=producer, daemon=True)
    t.start()
    time.sleep(0.3)
    assert not done.is_set() and r.size() == 10
    r.sample(4, "cpu")
    assert done.wait(5.0)
    assert r.size() == 9 and r.num_add() == 11


def test_prefetch_returns_batches(ours):
    r = ours.ValuePrioritizedReplay(capacity=128, seed=1, alpha=1.0, beta=0.4, prefetch=3, use_priority=False,
                                    compressed_values=False)
    q, v, _ = _fill(r, 100)
    rows = {tuple(x.tolist()) for x in q}
    for _ in range(5):
        t, w = r.sample(16, "cpu")
        assert t.query.shape == (16, 27) and t.values.shape == (16, 6) and w.shape == (16,)
        assert all(tuple(x.tolist()) in rows for x in t.query)


def test_context_is_subclassable_and_typed(ours):
    class Timed(ours.Context):  # cfvpy/utils.py:73-95
        def __init__(self):
            super().__init__()
            self.started = 0

        def start(self):
            super().start()
            self.started += 1

    c = Timed()
    assert c.terminated()  # nothing pushed
    with pytest.raises(TypeError):
        c.push_env_thread(object())
    c.start()
    assert c.started == 1
    c.terminate()


def test_model_locker_refuses_cpu_device(ours):
    from rebel_amd.models import Net2

    m = torch.jit.script(Net2(num_faces=4, num_dice=1, n_hidden=64, use_layer_norm=True, n_layers=2))
    # the message names the reference README command it breaks and the replacement
    with pytest.raises(RuntimeError, match=r"cpu_gen_threads=60.*cpu_gen_threads=0 selfplay.threads_per_gpu=1000.*cuda"):
        ours.ModelLocker([m], "cpu")


def test_eval_helpers_need_a_gpu(ours):
    """The evaluation helpers run on the device too: without one they raise, they never fall back to a CPU path."""
    from rebel_amd import capi

    if capi.device_count() > 0:
        pytest.skip("a GPU is present")
    p = ours.RecursiveSolvingParams()
    p.num_dice, p.num_faces = 1, 3
    p.subgame_params.use_cfr = True
    with pytest.raises(RuntimeError, match="HIP|device"):
        ours.compute_exploitability_fp(p)


def test_blocks_larger_than_the_free_space_are_appended_in_chunks(ours):
    """A block of 2 x lanes examples against a small buffer: the producer only ever asks for ring - capacity slots at a
    time, which sample()'s trim always frees; contents arrive complete and in order."""
    r = ours.ValuePrioritizedReplay(capacity=32, seed=3, alpha=1.0, beta=0.4, prefetch=0, use_priority=False,
                                    compressed_values=False)
    n = 200  # five times the 40-slot ring
    q = torch.arange(n * 3, dtype=torch.float32).reshape(n, 3)
    v = torch.arange(n * 2, dtype=torch.float32).reshape(n, 2) + 0.5
    seen = []
    done = threading.Event()

    def producer():
        r.push([q, v, torch.ones(n)])
        done.set()

    th = threading.Thread(target=producer)
    th.start()
    t0 = time.time()
    while not done.is_set():
        if r.size() > 0:
            b, _ = r.sample(4, "cpu")
            seen.append(b.query[:, 0].clone())
        assert time.time() - t0 < 60, "producer stalled"
        time.sleep(0.001)
    th.join()
    assert r.num_add() == n
    tail = r.extract()
    assert torch.equal(tail[0], q[n - tail[0].shape[0]:]) and torch.equal(tail[1], v[n - tail[1].shape[0]:])
    assert all((s % 3 == 0).all() for s in seen)  # every sampled row is a whole row of the source


def test_num_add_does_not_wrap_at_2_to_31(ours):
    """One MI355X adds ~94 k examples/s: an `int` counter (the reference's type, prioritized_replay.h:496) turns negative
    after 6.3 h on one generating GPU, 54 min on seven, and the unmodified trainer's throttle
    `num_add() * train_gen_ratio >= train_size * (epoch + 1)` (cfvpy/selfplay.py:391-404) then never opens again.  The
    counter is 64-bit here: seeded just under 2^31 through the test hook, three more blocks keep it growing."""
    r = ours.ValuePrioritizedReplay(capacity=64, seed=1, alpha=1.0, beta=0.4, prefetch=0, use_priority=False,
                                    compressed_values=False)
    start = 2 ** 31 - 10
    r._set_num_add_for_test(start)
    seen = [r.num_add()]
    for _ in range(3):
        r.push([torch.zeros(8, 3), torch.zeros(8, 2), torch.ones(8)])
        seen.append(r.num_add())
    assert seen == [start, start + 8, start + 16, start + 24] and seen[-1] > 2 ** 31
    ratio, train_size, epoch = 4, 25600, 10 ** 5  # the gate of selfplay.py:391-404 with liars_sp.yaml's train_gen_ratio
    assert r.num_add() * ratio >= train_size * (epoch + 1)
    assert r.size() == 24


def test_context_plans_one_engine_per_model_locker_device(ours, monkeypatch):
    """The in-process multi-GPU topology (cfvpy/selfplay.py:187-252: one ModelLocker per generating GPU, threads_per_gpu
    create_cfr_thread calls each, seeds rank*1000+i): Context.start() builds one worker = one engine + one driver thread
    per (ModelLocker, replay, config) group, on the locker's device, holding exactly that group's lanes.  Host logic:
    checked here without a GPU through the plan the context would start."""
    from rebel_amd.models import Net2

    m = [torch.jit.script(Net2(num_faces=4, num_dice=1, n_hidden=256, use_layer_norm=True, n_layers=2)) for _ in range(3)]
    lockers = [ours.ModelLocker([m[k]], f"cuda:{k + 1}") for k in range(3)]  # cuda:0 trains, cuda:1.. generate (:193)
    replay = ours.ValuePrioritizedReplay(capacity=1024, seed=1, alpha=1.0, beta=0.4, prefetch=0, use_priority=False,
                                         compressed_values=False)
    cfg = ours.RecursiveSolvingParams()
    cfg.num_dice, cfg.num_faces, cfg.random_action_prob, cfg.sample_leaf = 1, 4, 0.25, True
    cfg.subgame_params.num_iters, cfg.subgame_params.use_cfr = 64, True
    cfg2 = ours.RecursiveSolvingParams()
    cfg2.num_dice, cfg2.num_faces, cfg2.random_action_prob, cfg2.sample_leaf = 1, 4, 0.25, True
    cfg2.subgame_params.num_iters, cfg2.subgame_params.use_cfr = 128, True
    ctx = ours.Context()
    rank, per_gpu = 2, 5
    seed = 0
    for k, locker in enumerate(lockers):
        for _ in range(per_gpu):
            ctx.push_env_thread(ours.create_cfr_thread(locker, replay, cfg, rank * 1000 + seed))
            seed += 1
    ctx.push_env_thread(ours.create_cfr_thread(lockers[0], replay, cfg2, 7))  # other solver settings: its own engine
    plan = ctx._plan()
    assert [(dev, idx, n) for dev, idx, n, _ in plan] == [("cuda:1", 1, 5), ("cuda:2", 2, 5), ("cuda:3", 3, 5),
                                                         ("cuda:1", 1, 1)]
    assert [s for _, _, _, s in plan][:3] == [list(range(2000 + 5 * k, 2005 + 5 * k)) for k in range(3)]
    assert plan[3][3] == [7]
    monkeypatch.setenv("REBEL_AMD_LANES_PER_THREAD", "3")  # lanes per create_cfr_thread call, seed stride 1000003
    plan = ctx._plan()
    assert plan[0][2] == 5 and plan[0][3][:4] == [2000, 2000 + 1000003, 2000 + 2 * 1000003, 2001]
    # ... or lanes per ModelLocker, spread over its calls (the first n % calls take one more): 12 lanes over 5 calls = 3 3 2 2 2
    monkeypatch.delenv("REBEL_AMD_LANES_PER_THREAD")
    monkeypatch.setenv("REBEL_AMD_LANES_PER_GPU", "12")
    plan = ctx._plan()
    assert plan[0][2] == 5 and len(plan[0][3]) == 12
    assert plan[0][3] == [2000, 2000 + 1000003, 2000 + 2 * 1000003, 2001, 2001 + 1000003, 2001 + 2 * 1000003,
                          2002, 2002 + 1000003, 2003, 2003 + 1000003, 2004, 2004 + 1000003]
    monkeypatch.setenv("REBEL_AMD_LANES_PER_GPU", "3")  # fewer lanes than calls on the first three lockers: refused
    with pytest.raises(Exception, match="REBEL_AMD_LANES_PER_GPU"):
        ctx._plan()
    monkeypatch.delenv("REBEL_AMD_LANES_PER_GPU")


def test_call_1001_on_one_locker_is_refused_without_an_explicit_lane_layout(ours, monkeypatch):
    """cfvpy/selfplay.py:250 seeds generator i of rank r with r*1000 + i: the 1001st create_cfr_thread call on one ModelLocker
    would replay another rank's game draw for draw.  Context.start() refuses it (round 5: a stderr warning) before any engine
    is built -- so this needs no GPU -- unless the caller states its lane layout through REBEL_AMD_LANES_PER_GPU / _THREAD."""
    from rebel_amd.models import Net2

    monkeypatch.delenv("REBEL_AMD_LANES_PER_THREAD", raising=False)
    monkeypatch.delenv("REBEL_AMD_LANES_PER_GPU", raising=False)
    m = torch.jit.script(Net2(num_faces=4, num_dice=1, n_hidden=256, use_layer_norm=True, n_layers=2))
    locker = ours.ModelLocker([m], "cuda:0")
    replay = ours.ValuePrioritizedReplay(capacity=1024, seed=1, alpha=1.0, beta=0.4, prefetch=0, use_priority=False,
                                         compressed_values=False)
    cfg = ours.RecursiveSolvingParams()
    cfg.num_dice, cfg.num_faces = 1, 4
    cfg.subgame_params.num_iters, cfg.subgame_params.use_cfr = 8, True
    ctx = ours.Context()
    for i in range(1001):
        ctx.push_env_thread(ours.create_cfr_thread(locker, replay, cfg, i))
    with pytest.raises(Exception, match="1001 create_cfr_thread calls share one ModelLocker"):
        ctx.start()
    assert len(ctx._plan()[0][3]) == 1001  # the plan is still inspectable; nothing was started
