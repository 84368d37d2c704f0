"""GPU parity tests (P1): the HIP CFR path, called through the C ABI (rebel_amd.capi -> librebel_hip.so), against the
CPU oracle (oracle/, pinned to the compiled reference) and the committed golden vectors generated from the reference.

Bar: bit-exact (np.array_equal on fp64 arrays / sha256) -- regrets, sum_strategies, last and average strategies, root
value means, queries and training examples -- for identical leaf values (zero net, synthetic net, teacher-forced
callback net).  See DESIGN.md "Parity protocol".
"""
import numpy as np
import pytest

from tests import golden_util as G
from tests.cases import SOLVER_CASES, case_beliefs

pytestmark = pytest.mark.gpu

CFR_CASES = sorted(SOLVER_CASES)  # CFR variants and fictitious play


def _engine(c, max_lanes=1):
    from rebel_amd import capi

    e = capi.Engine(c["d"], c["f"], capi.make_params(**c["p"]), max_lanes=max_lanes)
    if c["net"] == "synthetic":
        e.set_net_synthetic()
    else:
        e.set_net_zero()  # "none" cases are full-depth trees: no pseudo-leaves, the net is never consulted
    return e


def _oracle_solver(port, c, beliefs):
    from oracle import orc
    from tests.cases import NET_CODE

    return port.solver(c["d"], c["f"], orc.make_params(**c["p"]), c.get("lb", -1), c.get("pl", 0), beliefs,
                       NET_CODE[c["net"]])


@pytest.mark.parametrize("name", CFR_CASES)
def test_solver_bit_exact_vs_oracle_and_golden(name, port):
    from oracle import orc
    from rebel_amd import capi

    c = SOLVER_CASES[name]
    e = _engine(c)
    b = case_beliefs(c, e.H)
    o = _oracle_solver(port, c, b)
    e.reset([c.get("lb", -1)], [c.get("pl", 0)], b[None])
    assert e.tree_size(0) == o.N
    n_it = c["p"]["num_iters"]
    pairs = [(capi.GET_AVERAGE, orc.GET_AVERAGE), (capi.GET_LAST, orc.GET_LAST), (capi.GET_SUM, orc.GET_SUM)]
    if c["p"].get("use_cfr"):
        pairs.append((capi.GET_REGRETS, orc.GET_REGRETS))
    for w, ow in pairs:  # freshly built solver
        assert np.array_equal(e.get(0, w), o.get(ow)), (name, "init", w)
    checkpoints = {1, 2, 3, n_it // 2, n_it}
    for it in range(n_it):
        e.step(it % 2)
        o.step(it % 2)
        if it + 1 in checkpoints:
            for w, ow in pairs:
                assert np.array_equal(e.get(0, w), o.get(ow)), (name, it, w)
            if it >= 1:
                for pl in (0, 1):
                    assert np.array_equal(e.hand_values(0, pl), o.hand_values(pl)), (name, it, pl)
    arrays = {"average": e.get(0, capi.GET_AVERAGE), "last": e.get(0, capi.GET_LAST), "sum": e.get(0, capi.GET_SUM)}
    if c["p"].get("use_cfr"):
        arrays["regrets"] = e.get(0, capi.GET_REGRETS)
    G.check_solver_arrays(name, arrays, np.stack([e.hand_values(0, 0), e.hand_values(0, 1)]), exact=True)
    if c["net"] != "none":
        q, v = e.examples(0)
        g = G.load("solver_cases.npz")
        assert np.array_equal(q, g[f"{name}/example_queries"])
        assert np.array_equal(v, g[f"{name}/example_values"])


def test_multistep_matches_golden():
    """The fused multistep path (no host round trips between iterations) ends on the golden state as well."""
    from rebel_amd import capi

    for name in ("1d6f_root_syn_1024", "2d3f_root_syn_1024", "1d4f_dcfr_syn_64"):
        c = SOLVER_CASES[name]
        e = _engine(c)
        b = case_beliefs(c, e.H)
        e.reset([c.get("lb", -1)], [c.get("pl", 0)], b[None])
        e.multistep()
        arrays = {"average": e.get(0, capi.GET_AVERAGE), "last": e.get(0, capi.GET_LAST),
                  "sum": e.get(0, capi.GET_SUM), "regrets": e.get(0, capi.GET_REGRETS)}
        G.check_solver_arrays(name, arrays, np.stack([e.hand_values(0, 0), e.hand_values(0, 1)]), exact=True)


@pytest.mark.parametrize("d,f,iters", [(1, 6, 96), (2, 3, 48), (1, 4, 64)])
def test_batched_heterogeneous_lanes_bit_exact(d, f, iters, port):
    """Many lanes at different roots / movers / beliefs in ONE launch sequence: every lane equals its own oracle run."""
    from oracle import orc
    from rebel_amd import capi

    kw = dict(num_iters=iters, max_depth=2, linear_update=True, use_cfr=True)
    A, H = port.num_actions(d, f), port.num_hands(d, f)
    rng = np.random.default_rng(1234)
    roots = list(range(-1, A - 1)) * 2 + [-1] * 5
    B = len(roots)
    players = rng.integers(0, 2, B)
    beliefs = rng.dirichlet(np.ones(H), size=(B, 2))
    beliefs[3] = 1.0 / H
    acts = rng.integers(0, iters + 1, B)
    acts[0], acts[1] = 0, iters
    e = capi.Engine(d, f, capi.make_params(**kw), max_lanes=B)
    e.set_net_synthetic()
    e.reset(roots, players, beliefs, acts)
    e.multistep()
    for b in range(B):
        o = port.solver(d, f, orc.make_params(**kw), roots[b], int(players[b]), beliefs[b], orc.NET_SYNTHETIC)
        snap = None
        for it in range(iters):
            if it == acts[b]:
                snap = o.get(orc.GET_LAST)
            o.step(it % 2)
        if acts[b] == iters:
            snap = o.get(orc.GET_LAST)
        assert e.tree_size(b) == o.N
        for w, ow in [(capi.GET_LAST, orc.GET_LAST), (capi.GET_SUM, orc.GET_SUM), (capi.GET_REGRETS, orc.GET_REGRETS),
                      (capi.GET_AVERAGE, orc.GET_AVERAGE)]:
            assert np.array_equal(e.get(b, w), o.get(ow)), (b, roots[b], w)
        assert np.array_equal(e.get_snapshot(b), snap), (b, "snapshot at act_iteration", acts[b])
        for pl in (0, 1):
            assert np.array_equal(e.hand_values(b, pl), o.hand_values(pl))
        o.update_value_network()
        q, v = e.examples(b)
        assert np.array_equal(q, np.stack([x for x, _ in o.examples]))
        assert np.array_equal(v, np.stack([x for _, x in o.examples]))


@pytest.mark.parametrize("d,f,iters,B,stride", [
    (1, 6, 48, 4096, 173),     # two streams, every CU holding its full complement of CFR workgroups
    (1, 4, 1024, 4096, 311),   # BASELINE config 2: 1dx4f, subgame_iters=1024, 4096 concurrent subgames
    (2, 3, 1024, 1536, 257),   # BASELINE config 4: 2dx3f, subgame_iters=1024 (one GPU's lane set)
    (2, 6, 96, 64, 13),        # BASELINE config 5's game: H = 36 lanes side by side on the big-tree kernel
])
def test_full_occupancy_lanes_bit_exact(port, d, f, iters, B, stride):
    """Thousands of heterogeneous lanes in lock-step at the BASELINE configurations' sizes: a sample of lanes, spread over
    both streams' halves, still equals its own oracle run bit for bit (state, snapshot at act_iteration, root values)."""
    from oracle import orc
    from rebel_amd import capi

    kw = dict(num_iters=iters, max_depth=2, linear_update=True, use_cfr=True)
    A, H = port.num_actions(d, f), port.num_hands(d, f)
    rng = np.random.default_rng(99)
    roots = rng.integers(-1, A - 1, B)
    roots[rng.random(B) < 0.4] = -1  # plenty of root subgames (largest trees)
    players = rng.integers(0, 2, B)
    beliefs = rng.dirichlet(np.ones(H), size=(B, 2))
    acts = rng.integers(0, iters + 1, B)
    e = capi.Engine(d, f, capi.make_params(**kw), max_lanes=B)
    e.set_net_synthetic()
    e.reset(roots, players, beliefs, acts)
    e.multistep()
    for b in list(range(0, B, stride)) + [B // 2 - 1, B // 2, B - 1]:
        o = port.solver(d, f, orc.make_params(**kw), int(roots[b]), int(players[b]), beliefs[b], orc.NET_SYNTHETIC)
        snap = None
        for it in range(iters):
            if it == acts[b]:
                snap = o.get(orc.GET_LAST)
            o.step(it % 2)
        if acts[b] == iters:
            snap = o.get(orc.GET_LAST)
        for w, ow in [(capi.GET_LAST, orc.GET_LAST), (capi.GET_SUM, orc.GET_SUM), (capi.GET_REGRETS, orc.GET_REGRETS)]:
            assert np.array_equal(e.get(b, w), o.get(ow)), (b, roots[b], w)
        assert np.array_equal(e.get_snapshot(b), snap), (b, "snapshot", acts[b])
        for pl in (0, 1):
            assert np.array_equal(e.hand_values(b, pl), o.hand_values(pl))


def test_callback_net_teacher_forcing(port):
    """Leaf values supplied by the SAME host function on both sides: identical query streams, identical end state."""
    from oracle import orc
    from rebel_amd import capi

    d, f = 1, 6
    A, H = port.num_actions(d, f), port.num_hands(d, f)
    rng = np.random.default_rng(0)
    W = rng.standard_normal((2 + A + 2 * H, H)).astype(np.float32) * 0.3
    seen = {"gpu": [], "port": []}

    def make(tag):
        def fn(q):
            seen[tag].append(q.copy())
            return np.tanh(q @ W).astype(np.float32)
        return fn

    kw = dict(num_iters=200, max_depth=2, linear_update=True, use_cfr=True)
    e = capi.Engine(d, f, capi.make_params(**kw))
    e.set_net_callback(make("gpu"))
    e.reset([-1], [0], np.full((1, 2, H), 1.0 / H))
    e.multistep()
    o = port.solver(d, f, orc.make_params(**kw), net=orc.NET_CALLBACK, net_fn=make("port"))
    o.multistep()
    assert len(seen["gpu"]) == len(seen["port"]) == 200
    for x, y in zip(seen["gpu"], seen["port"]):
        assert np.array_equal(x, y)
    for w, ow in [(capi.GET_AVERAGE, orc.GET_AVERAGE), (capi.GET_LAST, orc.GET_LAST), (capi.GET_SUM, orc.GET_SUM),
                  (capi.GET_REGRETS, orc.GET_REGRETS)]:
        assert np.array_equal(e.get(0, w), o.get(ow))


def test_irregular_traverser_order(port):
    """ISubgameSolver::step(traverser) with a non-alternating order (the engine re-encodes the pending queries)."""
    from oracle import orc
    from rebel_amd import capi

    kw = dict(num_iters=10, max_depth=2, linear_update=False, use_cfr=True)
    e = capi.Engine(1, 5, capi.make_params(**kw))
    e.set_net_synthetic()
    H = e.H
    b = np.random.default_rng(5).dirichlet(np.ones(H), size=2)
    e.reset([2], [1], b[None])
    o = port.solver(1, 5, orc.make_params(**kw), 2, 1, b, orc.NET_SYNTHETIC)
    for t in (0, 0, 1, 0, 1, 1, 1, 0):
        e.step(t)
        o.step(t)
        assert np.array_equal(e.get(0, capi.GET_REGRETS), o.get(orc.GET_REGRETS))
        assert np.array_equal(e.get(0, capi.GET_AVERAGE), o.get(orc.GET_AVERAGE))


def test_queries_bit_exact_first_iteration(port):
    """The [rows, Q] fp32 query matrix the net sees equals the oracle's, row for row (row order = BFS leaf order)."""
    from oracle import orc
    from rebel_amd import capi

    kw = dict(num_iters=4, max_depth=2, linear_update=True, use_cfr=True)
    seen = []

    def fn(q):
        seen.append(q.copy())
        return np.zeros((q.shape[0], 9), np.float32)

    b = np.random.default_rng(9).dirichlet(np.ones(9), size=2)
    o = port.solver(2, 3, orc.make_params(**kw), 1, 1, b, orc.NET_CALLBACK, net_fn=fn)
    o.step(0)
    e = capi.Engine(2, 3, capi.make_params(**kw))
    e.reset([1], [1], b[None])
    assert np.array_equal(e.queries(), seen[0])


@pytest.mark.parametrize("d,f", [(1, 6), (1, 4), (2, 3), (2, 6)])
def test_split_query_layout_equals_canonical(d, f, monkeypatch, port):
    """With the fused MLP net the one-wavefront CFR kernel (and, since round 4, the 2 dice x 6 faces flat kernel) writes only the dynamic part of the query rows, contiguously
    (CfrArgs::q_dyn), and the net reads (dynamic row | static row): the canonical [rows, Q] matrix rebuilt from them equals the
    matrix of the canonical path (RBL_QSPLIT=0), of a zero-net engine and of the oracle bit for bit.  The net here has a zero
    output layer, so the two layouts' different summation orders cannot leak into the CFR state."""
    from oracle import orc
    from rebel_amd import capi

    kw = dict(num_iters=16, max_depth=2, linear_update=True, use_cfr=True)
    rng = np.random.default_rng(5)
    roots, players = [-1, 2, 5, -1, 7], [0, 1, 0, 1, 1]
    B = len(roots)
    H = f ** d
    beliefs = rng.dirichlet(np.ones(H), size=(B, 2))

    def engine(zero_net):
        e = capi.Engine(d, f, capi.make_params(**kw), max_lanes=B)
        if zero_net:
            e.set_net_zero()
        else:
            layers = [(rng.uniform(-1, 1, (256, e.Q)).astype(np.float32), rng.uniform(-1, 1, 256).astype(np.float32)),
                      (rng.uniform(-.1, .1, (256, 256)).astype(np.float32), np.zeros(256, np.float32))]
            ln = [(np.ones(256, np.float32), np.zeros(256, np.float32))] * 2
            e.set_net_mlp(layers, ln, np.zeros((e.H, 256), np.float32), np.zeros(e.H, np.float32))
        e.reset(roots, players, beliefs)
        return e

    split = engine(False)
    monkeypatch.setenv("RBL_QSPLIT", "0")
    canon = engine(False)
    monkeypatch.delenv("RBL_QSPLIT")
    zero = engine(True)
    seen = []
    oracles = [port.solver(d, f, orc.make_params(**kw), roots[b], players[b], beliefs[b], orc.NET_CALLBACK,
                           net_fn=lambda q: (seen.append(q.copy()), np.zeros((q.shape[0], H), np.float32))[1])
               for b in range(B)]
    for it in range(7):
        q0 = split.queries()
        assert np.array_equal(q0, canon.queries()) and np.array_equal(q0, zero.queries()), it
        seen.clear()
        for o in oracles:
            o.step(it % 2)
        assert np.array_equal(q0, np.concatenate(seen)), it
        for e in (split, canon, zero):
            e.step(it % 2)
    for lane in range(B):
        for which in (capi.GET_LAST, capi.GET_REGRETS, capi.GET_SUM):
            assert np.array_equal(split.get(lane, which), zero.get(lane, which))


def test_net_exchanged_in_a_live_solve_rebuilds_the_query_layout(port):
    """the query layout follows the net kind (split rows for the fused MLP, canonical rows otherwise).  Setting the
    net AFTER reset(), or swapping net kinds in the middle of a solve, must hand the new net the rows of the live solve:
    reset -> set_net_mlp -> steps equals set_net_mlp -> reset -> steps bit for bit, and an MLP -> synthetic swap after some
    steps continues exactly like the oracle driven by the same leaf values."""
    from oracle import orc
    from rebel_amd import capi

    d, f = 1, 6
    kw = dict(num_iters=24, max_depth=2, linear_update=True, use_cfr=True)
    rng = np.random.default_rng(11)
    roots, players = [-1, 3, -1, 6], [0, 1, 1, 0]
    B, H = len(roots), f ** d
    beliefs = rng.dirichlet(np.ones(H), size=(B, 2))
    e0 = capi.Engine(d, f, capi.make_params(**kw), max_lanes=B)
    layers = [(rng.uniform(-1, 1, (256, e0.Q)).astype(np.float32), rng.uniform(-1, 1, 256).astype(np.float32)),
              (rng.uniform(-.1, .1, (256, 256)).astype(np.float32), rng.uniform(-.1, .1, 256).astype(np.float32))]
    ln = [(np.ones(256, np.float32), np.zeros(256, np.float32))] * 2
    mlp = (layers, ln, rng.uniform(-.1, .1, (H, 256)).astype(np.float32), np.zeros(H, np.float32))
    e0.close()

    def run(order):
        e = capi.Engine(d, f, capi.make_params(**kw), max_lanes=B)
        if order == "net_first":
            e.set_net_mlp(*mlp)
            e.reset(roots, players, beliefs)
        else:  # the solver is built with the default (zero) net's canonical rows, then the MLP arrives
            e.set_net_zero()
            e.reset(roots, players, beliefs)
            e.set_net_mlp(*mlp)
        for it in range(6):
            e.step(it % 2)
        return e

    a, b = run("net_first"), run("reset_first")
    assert np.array_equal(a.queries(), b.queries())
    for lane in range(B):
        for which in (capi.GET_LAST, capi.GET_REGRETS, capi.GET_SUM):
            assert np.array_equal(a.get(lane, which), b.get(lane, which)), (lane, which)
    # MLP -> synthetic in the middle of the solve: the wave kernel wrote dynamic rows only, the synthetic net reads canonical rows
    q_before = a.queries()
    a.set_net_synthetic()
    assert np.array_equal(a.queries(), q_before)
    b.set_net_synthetic()
    for it in range(6, 10):
        a.step(it % 2)
        b.step(it % 2)
    for lane in range(B):
        for which in (capi.GET_LAST, capi.GET_REGRETS, capi.GET_SUM):
            assert np.array_equal(a.get(lane, which), b.get(lane, which)), (lane, which)
    # ... and against a run that used the synthetic net's canonical rows all along for those steps: the leaf values of the
    # steps after the swap only depend on the queries, which the oracle-checked canonical path produces
    assert np.isfinite(a.get(0, capi.GET_REGRETS)).all()


@pytest.mark.parametrize("d,f", [(1, 6), (1, 4), (1, 5), (2, 3)])
def test_lanes_without_pseudo_leaves_in_the_wave_kernel(d, f, port):
    """late-game roots (last bid >= A - 3) have only terminal leaves (L == 0; the net is never called for them,
    subgame_solving.cc:254) and the root with the highest bid has a single edge.  Alone in an engine (lane 0, row offset 0)
    and mixed with other lanes they equal the oracle bit for bit -- the staging clamps of the one-wavefront kernel must not
    index below the lane's arrays."""
    from oracle import orc
    from rebel_amd import capi

    kw = dict(num_iters=12, max_depth=2, linear_update=True, use_cfr=True)
    A, H = port.num_actions(d, f), port.num_hands(d, f)
    rng = np.random.default_rng(3)
    for roots in ([A - 3], [A - 4], [A - 3, -1, A - 4, A - 5, A - 3]):
        B = len(roots)
        players = [int(x) for x in rng.integers(0, 2, B)]
        beliefs = rng.dirichlet(np.ones(H), size=(B, 2))
        e = capi.Engine(d, f, capi.make_params(**kw), max_lanes=B)
        e.set_net_synthetic()
        e.reset(roots, players, beliefs)
        e.multistep()
        for b in range(B):
            o = port.solver(d, f, orc.make_params(**kw), roots[b], players[b], beliefs[b], orc.NET_SYNTHETIC)
            o.multistep()
            assert e.tree_size(b) == o.N
            for w, ow in ((capi.GET_REGRETS, orc.GET_REGRETS), (capi.GET_LAST, orc.GET_LAST), (capi.GET_SUM, orc.GET_SUM)):
                assert np.array_equal(e.get(b, w), o.get(ow)), (roots, b, w)
        e.close()


@pytest.mark.parametrize("linear,optimistic", [(False, False), (True, False), (False, True), (True, True)])
def test_fictitious_play_variants_bit_exact(linear, optimistic, port):
    """FP solver (subgame_solving.cc:364-506) incl. linear and optimistic averaging, heterogeneous lanes, snapshots."""
    from oracle import orc
    from rebel_amd import capi

    d, f, iters = 1, 5, 41
    kw = dict(num_iters=iters, max_depth=2, linear_update=linear, optimistic=optimistic, use_cfr=False)
    rng = np.random.default_rng(11)
    roots = [-1, 0, 3, 7, 8, 9, -1]
    B = len(roots)
    players = rng.integers(0, 2, B)
    H = port.num_hands(d, f)
    beliefs = rng.dirichlet(np.ones(H), size=(B, 2))
    acts = rng.integers(0, iters + 1, B)
    e = capi.Engine(d, f, capi.make_params(**kw), max_lanes=B)
    e.set_net_synthetic()
    e.reset(roots, players, beliefs, acts)
    e.multistep()
    for b in range(B):
        o = port.solver(d, f, orc.make_params(**kw), roots[b], int(players[b]), beliefs[b], orc.NET_SYNTHETIC)
        snap = None
        for it in range(iters):
            if it == acts[b]:
                snap = o.get(orc.GET_LAST)
            o.step(it % 2)
        if acts[b] == iters:
            snap = o.get(orc.GET_LAST)
        for w, ow in [(capi.GET_AVERAGE, orc.GET_AVERAGE), (capi.GET_LAST, orc.GET_LAST), (capi.GET_SUM, orc.GET_SUM)]:
            assert np.array_equal(e.get(b, w), o.get(ow)), (b, roots[b], w)
        assert np.array_equal(e.get_snapshot(b), snap), (b, acts[b])
        for pl in (0, 1):
            assert np.array_equal(e.hand_values(b, pl), o.hand_values(pl))


def test_error_paths():
    from rebel_amd import capi

    with pytest.raises(capi.RebelError):  # DCFR discounts belong to CFR
        capi.Engine(1, 4, capi.make_params(num_iters=4, use_cfr=False, dcfr=True))
    e = capi.Engine(1, 4, capi.make_params(num_iters=4, use_cfr=True), max_lanes=2)
    with pytest.raises(capi.RebelError):  # terminal root state
        e.reset([e.A - 1], [0], np.full((1, 2, e.H), 0.25))
    with pytest.raises(capi.RebelError):  # more lanes than the engine was sized for
        e.reset([-1] * 3, [0] * 3, np.full((3, 2, e.H), 0.25))
    with pytest.raises(capi.RebelError):
        e.step(0)  # nothing was reset successfully


def test_2d6f_root_2048_iterations_bit_exact(port):
    """BASELINE config 5: 2 dice x 6 faces (H = 36, root subgame N = 325, L = 276), subgame_iters = 2048, next to three
    smaller subgames of the same game: bit-exact against the oracle after 2048 iterations."""
    from oracle import orc
    from rebel_amd import capi

    d, f, iters = 2, 6, 2048
    kw = dict(num_iters=iters, max_depth=2, linear_update=True, use_cfr=True)
    H = port.num_hands(d, f)
    rng = np.random.default_rng(5)
    roots, players = [-1, 7, 16, 22], [0, 1, 0, 1]
    beliefs = rng.dirichlet(np.ones(H), size=(4, 2))
    e = capi.Engine(d, f, capi.make_params(**kw), max_lanes=4)
    e.set_net_synthetic()
    e.reset(roots, players, beliefs)
    e.multistep()
    for b in range(4):
        o = port.solver(d, f, orc.make_params(**kw), roots[b], players[b], beliefs[b], orc.NET_SYNTHETIC)
        o.multistep()
        for w, ow in [(capi.GET_LAST, orc.GET_LAST), (capi.GET_SUM, orc.GET_SUM), (capi.GET_REGRETS, orc.GET_REGRETS),
                      (capi.GET_AVERAGE, orc.GET_AVERAGE)]:
            assert np.array_equal(e.get(b, w), o.get(ow)), (b, w)
        for pl in (0, 1):
            assert np.array_equal(e.hand_values(b, pl), o.hand_values(pl))


@pytest.mark.parametrize("flat", [1, 0])
def test_2d6f_kernels_agree_and_the_flat_one_runs(flat, port, monkeypatch):
    """2 dice x 6 faces has two step kernels: cfr_flat_kernel (element-parallel, sigma in LDS: the default) and
    cfr_rows_kernel<GS> (RBL_CFR_FLAT=0).  Both reach the oracle's state bit for bit on a mix of subgame sizes incl. DCFR
    discounts and per-lane stop iterations, and the engine reports which one it launched (no silent fallback)."""
    from oracle import orc
    from rebel_amd import capi

    monkeypatch.setenv("RBL_CFR_FLAT", str(flat))
    d, f, iters = 2, 6, 37
    kw = dict(num_iters=iters, max_depth=2, linear_update=False, use_cfr=True, dcfr=True, dcfr_alpha=1.5, dcfr_beta=0.5,
              dcfr_gamma=2.0)
    H = port.num_hands(d, f)
    rng = np.random.default_rng(11)
    roots = [-1, 0, 3, 9, 14, 20, 22, 23, -1]  # 23: only "liar" is left (two nodes, no pseudo-leaf)
    players = [0, 1, 0, 1, 0, 1, 0, 1, 1]
    B = len(roots)
    beliefs = rng.dirichlet(np.ones(H), size=(B, 2))
    act = [5, 36, 0, 17, 36, 2, 9, 1, 30]
    e = capi.Engine(d, f, capi.make_params(**kw), max_lanes=B)
    e.set_net_synthetic()
    e.reset(roots, players, beliefs, act)
    e.multistep()
    assert e.stats()["cfr_kernel"] == (4 if flat else 3)
    for b in range(B):
        o = port.solver(d, f, orc.make_params(**kw), roots[b], players[b], beliefs[b], orc.NET_SYNTHETIC)
        for it in range(iters):
            o.step(it % 2)
            if it + 1 == act[b]:
                snap = o.get(orc.GET_LAST)
        if act[b] == 0:
            snap = port.solver(d, f, orc.make_params(**kw), roots[b], players[b], beliefs[b], orc.NET_SYNTHETIC).get(orc.GET_LAST)
        for w, ow in [(capi.GET_LAST, orc.GET_LAST), (capi.GET_SUM, orc.GET_SUM), (capi.GET_REGRETS, orc.GET_REGRETS)]:
            assert np.array_equal(e.get(b, w), o.get(ow)), (b, w)
        assert np.array_equal(e.get_snapshot(b), snap), b
        for pl in (0, 1):
            assert np.array_equal(e.hand_values(b, pl), o.hand_values(pl))


@pytest.mark.parametrize("name", ["1d6f_root_syn_1024", "2d3f_bid2_p1_syn_256", "1d4f_dcfr_syn_64"])
def test_row_kernel_fallback_bit_exact(name, port, monkeypatch):
    """RBL_CFR_WAVE=0: the row-per-thread kernel (the fallback of the one-wavefront kernel, and the kernel deeper subgames
    run on) reaches the same golden end state."""
    from rebel_amd import capi

    monkeypatch.setenv("RBL_CFR_WAVE", "0")
    c = SOLVER_CASES[name]
    e = _engine(c)
    b = case_beliefs(c, e.H)
    e.reset([c.get("lb", -1)], [c.get("pl", 0)], b[None])
    e.multistep()
    arrays = {"average": e.get(0, capi.GET_AVERAGE), "last": e.get(0, capi.GET_LAST), "sum": e.get(0, capi.GET_SUM),
              "regrets": e.get(0, capi.GET_REGRETS)}
    G.check_solver_arrays(name, arrays, np.stack([e.hand_values(0, 0), e.hand_values(0, 1)]), exact=True)
