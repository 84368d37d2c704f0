"""GPU parity tests for the self-play walk (RlRunner, recursive_solving.cc:160-275) through the C ABI.

Per lane (= per seed) the emitted training examples -- root queries and root value means of every subgame of every
game, in order -- must equal the reference's bit for bit: that pins the node indices chosen by the sampler (they
determine the next root and beliefs), act_iteration handling, belief propagation and the CFR results together.
"""
import numpy as np
import pytest

from tests import golden_util as G
from tests.cases import RL_CASES

pytestmark = pytest.mark.gpu


def _run_lanes(c, seeds, games, expect_device=1):
    from rebel_amd import capi

    e = capi.Engine(c["d"], c["f"], capi.make_params(**c["p"]), max_lanes=len(seeds))
    e.set_net_synthetic() if c["net"] == "synthetic" else e.set_net_zero()
    sp = capi.SelfPlay(e, seeds, random_action_prob=c["rap"], sample_leaf=c["leaf"])
    per_lane = [[] for _ in seeds]
    done_at = [None] * len(seeds)  # number of examples when the lane finished `games` games
    finished = [0] * len(seeds)
    while any(d is None for d in done_at):
        _, lanes, q, v = sp.advance()
        assert sp.on_device() == expect_device  # the walk itself ran as HIP kernels (or on the host when asked to)
        for k in range(len(lanes)):
            per_lane[lanes[k]].append((q[k], v[k]))
        for i in range(len(seeds)):
            if sp.state(i)[0] == e.A - 1 and done_at[i] is None:
                finished[i] += 1
                if finished[i] == games:
                    done_at[i] = len(per_lane[i])
    return [pl[:n] for pl, n in zip(per_lane, done_at)]


@pytest.mark.parametrize("name", sorted(RL_CASES))
def test_selfplay_vs_golden(name):
    c = RL_CASES[name]
    (ex,) = _run_lanes(c, [c["seed"]], c["games"])
    g = G.load("rl_cases.npz")
    gq, gv = g[f"{name}/queries"], g[f"{name}/values"]
    assert len(ex) == len(gq)
    assert np.array_equal(np.stack([q for q, _ in ex]), gq)
    assert np.array_equal(np.stack([v for _, v in ex]), gv)


def test_selfplay_many_lanes_vs_oracle(port):
    """64 lanes with different seeds in lock-step: every lane reproduces the oracle's run for its own seed."""
    from oracle import orc

    c = dict(d=1, f=6, p=dict(num_iters=64, max_depth=2, linear_update=True, use_cfr=True), rap=0.25, leaf=True,
             net="synthetic")
    seeds = list(range(100, 164))
    games = 3
    lanes = _run_lanes(c, seeds, games)
    for seed, ex in zip(seeds, lanes):
        ref = port.rl_run(c["d"], c["f"], orc.make_params(**c["p"]), seed, games, random_action_prob=c["rap"],
                          sample_leaf=c["leaf"], net=orc.NET_SYNTHETIC)
        assert len(ex) == len(ref), seed
        for (q, v), (rq, rv) in zip(ex, ref):
            assert np.array_equal(q, rq) and np.array_equal(v, rv), seed


def test_selfplay_threaded_host_walk_vs_oracle(port):
    """1536 lanes: enough for the host-side sampling walk to be spread over several threads (engine.hip, SelfPlay::advance)
    and for two streams; a sample of lanes still reproduces the oracle's run for its own seed, games counted correctly."""
    from oracle import orc

    c = dict(d=1, f=6, p=dict(num_iters=24, max_depth=2, linear_update=True, use_cfr=True), rap=0.25, leaf=True,
             net="synthetic")
    seeds = list(range(5000, 5000 + 1536))
    games = 2
    lanes = _run_lanes(c, seeds, games)
    for i in list(range(0, len(seeds), 97)) + [255, 256, 767, 768, 1535]:
        ref = port.rl_run(c["d"], c["f"], orc.make_params(**c["p"]), seeds[i], games, random_action_prob=c["rap"],
                          sample_leaf=c["leaf"], net=orc.NET_SYNTHETIC)
        ex = lanes[i]
        assert len(ex) == len(ref), seeds[i]
        for (q, v), (rq, rv) in zip(ex, ref):
            assert np.array_equal(q, rq) and np.array_equal(v, rv), seeds[i]


def test_selfplay_2d6f_global_scratch_path(port):
    """2 dice x 6 faces (H = 36, root subgame N = 325): the lane working set does not fit LDS, so the kernel runs on its
    per-lane global scratch slab; trajectories still equal the oracle's bit for bit."""
    from oracle import orc

    c = dict(d=2, f=6, p=dict(num_iters=12, max_depth=2, linear_update=True, use_cfr=True), rap=0.25, leaf=True,
             net="synthetic")
    seeds = [3, 4, 5]
    lanes = _run_lanes(c, seeds, 2)
    for seed, ex in zip(seeds, lanes):
        ref = port.rl_run(c["d"], c["f"], orc.make_params(**c["p"]), seed, 2, random_action_prob=c["rap"],
                          sample_leaf=c["leaf"], net=orc.NET_SYNTHETIC)
        assert len(ex) == len(ref), seed
        for (q, v), (rq, rv) in zip(ex, ref):
            assert np.array_equal(q, rq) and np.array_equal(v, rv), seed


def test_selfplay_2d6f_size_sorted_launches_change_nothing(monkeypatch, port):
    """2 dice x 6 faces runs its CFR kernel per size-sorted SEGMENT of a lane part (device counting sort per epoch, one LDS
    request per segment; >= 1024 lanes per part = two or more segments): the example streams equal those of the unsorted
    launch order (RBL_GS_SORT=0) bit for bit, and sampled lanes equal the oracle's."""
    from oracle import orc

    c = dict(d=2, f=6, p=dict(num_iters=6, max_depth=2, linear_update=True, use_cfr=True), rap=0.25, leaf=True,
             net="synthetic")
    seeds = list(range(2000, 2000 + 2304))  # two parts of 1 152 lanes = two segments each
    sorted_runs = _run_lanes(c, seeds, 1)
    monkeypatch.setenv("RBL_GS_SORT", "0")
    plain = _run_lanes(c, seeds, 1)
    monkeypatch.delenv("RBL_GS_SORT")
    for a, b in zip(sorted_runs, plain):
        assert len(a) == len(b)
        for (q, v), (rq, rv) in zip(a, b):
            assert np.array_equal(q, rq) and np.array_equal(v, rv)
    for i in (0, 700, 1151, 1152, 2303):
        ref = port.rl_run(c["d"], c["f"], orc.make_params(**c["p"]), seeds[i], 1, random_action_prob=c["rap"],
                          sample_leaf=c["leaf"], net=orc.NET_SYNTHETIC)
        assert len(sorted_runs[i]) == len(ref), i
        for (q, v), (rq, rv) in zip(sorted_runs[i], ref):
            assert np.array_equal(q, rq) and np.array_equal(v, rv), i


def test_selfplay_fictitious_play_lanes(port):
    """use_cfr=false is the pybind default (subgame_solving.h:48): self-play with the FP solver (optimistic, linear)
    reproduces the oracle per seed as well."""
    from oracle import orc

    c = dict(d=1, f=5, p=dict(num_iters=48, max_depth=2, linear_update=True, optimistic=True, use_cfr=False), rap=0.25,
             leaf=True, net="synthetic")
    seeds = [21, 22, 23, 24]
    lanes = _run_lanes(c, seeds, 3)
    for seed, ex in zip(seeds, lanes):
        ref = port.rl_run(c["d"], c["f"], orc.make_params(**c["p"]), seed, 3, random_action_prob=c["rap"],
                          sample_leaf=c["leaf"], net=orc.NET_SYNTHETIC)
        assert len(ex) == len(ref), seed
        for (q, v), (rq, rv) in zip(ex, ref):
            assert np.array_equal(q, rq) and np.array_equal(v, rv), seed


@pytest.mark.parametrize("seed,hi,weights", [
    (0, 1, [0.5, 0.5]), (7, 1024, [0.1, 0.2, 0.7]), (123456789, 12, [1e-80, 0.3, 0.0, 0.69, 1e-9, 0.01]),
    (-5, 5, [3.0, 1.0, 0.0, 0.0, 2.5, 9.25, 1e-300]), (2**31 - 1, 2**31 - 2, [1.0]), (42, 0, [0.0, 0.0, 1.0])])
def test_device_rng_matches_libstdcxx_draw_for_draw(seed, hi, weights, port):
    """The device restatement of std::mt19937 + uniform_int / uniform_real<float> / discrete distributions
    (selfplay_kernels.hip) against the host library the reference links: 3 x 1500 draws cross several state twists
    (624 words each) and the Lemire rejection branch; a single weight draws nothing (libstdc++ returns 0)."""
    from rebel_amd import capi

    got = capi.device_rng_draws(seed, 1500, hi, weights)
    ref = port.rng_probe(seed, 1500, hi, weights)
    assert np.array_equal(got, ref)


def test_device_walk_equals_host_walk(monkeypatch):
    """Same seeds through the device walk and the legacy host walk (RBL_SELFPLAY_HOST=1, the path callback nets use):
    identical example streams, sample_leaf on and off."""
    for leaf in (True, False):
        c = dict(d=1, f=6, p=dict(num_iters=40, max_depth=2, linear_update=True, use_cfr=True), rap=0.35, leaf=leaf,
                 net="synthetic")
        seeds = list(range(900, 900 + 48))
        dev = _run_lanes(c, seeds, 3)
        monkeypatch.setenv("RBL_SELFPLAY_HOST", "1")
        host = _run_lanes(c, seeds, 3, expect_device=0)
        monkeypatch.delenv("RBL_SELFPLAY_HOST")
        for a, b in zip(dev, host):
            assert len(a) == len(b)
            for (q, v), (rq, rv) in zip(a, b):
                assert np.array_equal(q, rq) and np.array_equal(v, rv)


def test_selfplay_games_counter_and_callback_net_falls_back_to_host(port):
    """games_finished counts terminal states on the device; a callback net (teacher-forced oracle values) keeps the host
    walk and still reproduces the oracle."""
    from oracle import orc
    from rebel_amd import capi

    p = dict(num_iters=16, max_depth=2, linear_update=True, use_cfr=True)
    e = capi.Engine(1, 4, capi.make_params(**p), max_lanes=32)
    e.set_net_synthetic()
    sp = capi.SelfPlay(e, list(range(32)), random_action_prob=0.25, sample_leaf=True)
    finished = 0
    for _ in range(12):
        sp.advance()
        finished += sum(1 for i in range(32) if sp.state(i)[0] == e.A - 1)
    assert sp.on_device() == 1 and sp.games_finished() == finished > 0
    e2 = capi.Engine(1, 4, capi.make_params(**p), max_lanes=4)
    e2.set_net_callback(lambda q: port.synthetic_net(q, e2.A, e2.H))
    sp2 = capi.SelfPlay(e2, [11, 12, 13, 14], random_action_prob=0.25, sample_leaf=True)
    got = [[] for _ in range(4)]
    for _ in range(6):
        _, lanes, q, v = sp2.advance()
        for k in range(len(lanes)):
            got[lanes[k]].append((q[k], v[k]))
    assert sp2.on_device() == 0
    for i, seed in enumerate([11, 12, 13, 14]):
        ref = port.rl_run(1, 4, orc.make_params(**p), seed, 6, random_action_prob=0.25, sample_leaf=True,
                          net=orc.NET_SYNTHETIC)
        for (q, v), (rq, rv) in zip(got[i], ref):
            assert np.array_equal(q, rq) and np.array_equal(v, rv)


def test_selfplay_at_bench_size_vs_oracle(port):
    """The bench's own configuration -- 1dx6f, 1024 iterations per subgame, 16 384 lanes on one stream, device walk -- with
    the synthetic net in place of Net2 so that parity is exact: after two epochs a sample of lanes equals the oracle's
    example stream for its seed bit for bit, every emitted belief vector is a probability vector, and the games counter
    equals the number of lanes that reached a terminal state."""
    from oracle import orc
    from rebel_amd import capi

    d, f, iters, B = 1, 6, 1024, 16384
    p = dict(num_iters=iters, max_depth=2, linear_update=True, use_cfr=True)
    e = capi.Engine(d, f, capi.make_params(**p), max_lanes=B)
    e.set_net_synthetic()
    seeds = list(range(B))
    sp = capi.SelfPlay(e, seeds, random_action_prob=0.25, sample_leaf=True)
    got, finished = [[] for _ in range(B)], 0
    sample = list(range(0, B, 1371)) + [B - 1]
    for _ in range(2):
        n, lanes, q, v = sp.advance()
        assert n == B * iters and sp.on_device() == 1
        assert np.isfinite(q).all() and np.isfinite(v).all()
        A, H = e.A, e.H
        assert np.allclose(q[:, 2 + A:2 + A + H].sum(1), 1, atol=1e-5) and np.allclose(q[:, 2 + A + H:].sum(1), 1, atol=1e-5)
        assert set(np.unique(q[:, :2 + A])) <= {0.0, 1.0} and (q[:, 2:2 + A].sum(1) <= 1).all()
        for i in sample:
            got[i] += [(q[2 * i], v[2 * i]), (q[2 * i + 1], v[2 * i + 1])]
        finished += sum(1 for i in range(0, B, 97) if sp.state(i)[0] == e.A - 1)
    assert sp.games_finished() >= finished > 0
    for i in sample:
        ref = port.rl_run(d, f, orc.make_params(**p), seeds[i], 2, random_action_prob=0.25, sample_leaf=True,
                          net=orc.NET_SYNTHETIC)
        assert len(ref) >= 4
        for (q, v), (rq, rv) in zip(got[i], ref[:4]):
            assert np.array_equal(q, rq) and np.array_equal(v, rv), i


def test_selfplay_full_depth_subgames_need_no_net(port):
    """max_depth beyond the game's length: every subgame is the whole remaining game, there are no pseudo-leaves and no
    net rows at all (the device epoch then runs CFR launches only); trajectories still equal the oracle's."""
    from oracle import orc

    c = dict(d=1, f=4, p=dict(num_iters=40, max_depth=100, linear_update=True, use_cfr=True), rap=0.25, leaf=True,
             net="zero")
    seeds = [61, 62, 63, 64, 65]
    lanes = _run_lanes(c, seeds, 3)
    for seed, ex in zip(seeds, lanes):
        ref = port.rl_run(c["d"], c["f"], orc.make_params(**c["p"]), seed, 3, random_action_prob=c["rap"],
                          sample_leaf=c["leaf"], net=orc.NET_ZERO)
        assert len(ex) == len(ref), seed
        for (q, v), (rq, rv) in zip(ex, ref):
            assert np.array_equal(q, rq) and np.array_equal(v, rv), seed


def _dedup_streams(monkeypatch, dedup, d, f, iters, B, epochs, net):
    """Example stream, public states and counters of `epochs` epochs of B lanes (seeds 0..B-1), root de-duplication on / off."""
    from rebel_amd import capi

    if dedup:
        monkeypatch.setenv("REBEL_AMD_ROOT_DEDUP", "1")
    else:
        monkeypatch.delenv("REBEL_AMD_ROOT_DEDUP", raising=False)
    e = capi.Engine(d, f, capi.make_params(num_iters=iters, max_depth=2, linear_update=True, use_cfr=True), max_lanes=B)
    if net == "synthetic":
        e.set_net_synthetic()
    else:
        import torch

        from rebel_amd.models import Net2, mlp_weights_from_state_dict

        torch.manual_seed(7)
        m = Net2(num_faces=f, num_dice=d, n_hidden=256, use_layer_norm=True, n_layers=2).eval()
        with torch.no_grad():
            m.output.weight *= 30  # outputs of the size a trained net produces
            m.output.bias *= 30
        e.set_net_mlp(*mlp_weights_from_state_dict(m.state_dict()))
    sp = capi.SelfPlay(e, list(range(B)), random_action_prob=0.25, sample_leaf=True)
    out, executed = [], []
    for _ in range(epochs):
        n, lanes, q, v = sp.advance()
        assert sp.on_device() == 1
        executed.append(n)
        out.append((q.copy(), v.copy(), np.array([sp.state(i) for i in range(0, B, max(1, B // 257))])))
    served, games = sp.root_dedup_served(), sp.games_finished()
    sp.close()
    e.close()
    return out, executed, served, games


@pytest.mark.parametrize("d,f,iters,B,epochs,net", [(1, 6, 1024, 16384, 3, "synthetic"),  # the bench's shape, one stream
                                                    (1, 6, 128, 4096, 4, "mlp"),          # the MFMA forward, two lane parts
                                                    (2, 3, 64, 2048, 3, "synthetic"),
                                                    (2, 6, 48, 1024, 3, "synthetic")])    # cfr_flat_kernel + size-sorted launches
def test_root_dedup_keeps_every_stream_bit_identical(monkeypatch, port, d, f, iters, B, epochs, net):
    """REBEL_AMD_ROOT_DEDUP=1 (opt-in extra, default off).  Every lane in its root subgame computes the same
    num_iters iterations (recursive_solving.cc:160-163: reset to the root with uniform beliefs; CFR::step draws nothing), so with
    the option on ONE lane per epoch solves the root and the others sample from its sigma at their own act_iteration.  The bar:
    the example stream of EVERY lane and its public states are bit-identical to the mode-off run, epoch by epoch -- also through
    the MFMA forward, whose rows do not depend on their position in the batch -- while the iterations executed drop by exactly the
    served lane-epochs; and a sample of lanes still equals the oracle."""
    from oracle import orc

    off, ex_off, served_off, games_off = _dedup_streams(monkeypatch, False, d, f, iters, B, epochs, net)
    on, ex_on, served_on, games_on = _dedup_streams(monkeypatch, True, d, f, iters, B, epochs, net)
    assert served_off == 0 and ex_off == [B * iters] * epochs
    assert ex_on[0] == iters  # first epoch: every lane is at the root, one representative executes
    assert sum(ex_on) == (B * epochs - served_on) * iters and served_on >= B - 1
    assert games_on == games_off > 0
    for k, ((q0, v0, s0), (q1, v1, s1)) in enumerate(zip(off, on)):
        assert np.array_equal(q0, q1) and np.array_equal(v0, v1) and np.array_equal(s0, s1), k
    roots = [(int((q[0::2, 2:2 + 2 * d * f + 1].sum(1) == 0).sum())) for q, _, _ in on]
    print(f"root dedup {d}dx{f}f x {B} lanes: root lanes per epoch {roots}, executed lane-iterations per epoch "
          f"{[n // iters for n in ex_on]} of {B}")
    assert [B - n // iters for n in ex_on] == [max(0, r - 1) for r in roots]  # all root lanes but the representative are served
    if net == "synthetic":
        for i in list(range(0, B, max(1, B // 6))) + [B - 1]:
            ref = port.rl_run(d, f, orc.make_params(num_iters=iters, max_depth=2, linear_update=True, use_cfr=True), i, 2,
                              random_action_prob=0.25, sample_leaf=True, net=orc.NET_SYNTHETIC)
            got = [(q[2 * i + t], v[2 * i + t]) for q, v, _ in on for t in (0, 1)]
            for (q, v), (rq, rv) in zip(got, ref):
                assert np.array_equal(q, rq) and np.array_equal(v, rv), i
