"""Pins the port oracle (oracle/cfr_oracle.cc) to the compiled, unmodified reference (oracle/_ref): bit-for-bit on
every solver array and on self-play trajectories.  Skipped where the compiled reference is not available (then the
committed golden vectors, generated from it, take over: tests/test_golden.py).
"""
import numpy as np
import pytest

from oracle import orc
from tests.cases import NET_CODE, RL_CASES, SOLVER_CASES, case_beliefs


@pytest.mark.parametrize("name", sorted(SOLVER_CASES))
def test_solver_bit_exact(name, port, ref):
    c = SOLVER_CASES[name]
    p = orc.make_params(**c["p"])
    H = port.num_hands(c["d"], c["f"])
    b = case_beliefs(c, H)
    a = ref.solver(c["d"], c["f"], p, c.get("lb", -1), c.get("pl", 0), b, NET_CODE[c["net"]])
    o = port.solver(c["d"], c["f"], p, c.get("lb", -1), c.get("pl", 0), b, NET_CODE[c["net"]])
    assert a.N == o.N
    checkpoints = {1, 2, 3, p.num_iters // 2, p.num_iters}
    which = [orc.GET_AVERAGE, orc.GET_LAST, orc.GET_SUM] + ([orc.GET_REGRETS] if p.use_cfr else [])
    for it in range(p.num_iters):
        a.step(it % 2)
        o.step(it % 2)
        if it + 1 in checkpoints:
            for w in which:
                assert np.array_equal(a.get(w), o.get(w)), (name, it, w)
            if it >= 1:
                for pl in (0, 1):
                    assert np.array_equal(a.hand_values(pl), o.hand_values(pl))
    if c["net"] == "none":
        return  # the reference dereferences its (null) net in update_value_network
    a.update_value_network()
    o.update_value_network()
    assert len(a.examples) == len(o.examples) == 2
    for (qa, va), (qo, vo) in zip(a.examples, o.examples):
        assert np.array_equal(qa, qo) and np.array_equal(va, vo)


@pytest.mark.parametrize("name", sorted(RL_CASES))
def test_selfplay_trajectories_bit_exact(name, port, ref):
    c = RL_CASES[name]
    p = orc.make_params(**c["p"])
    kw = dict(random_action_prob=c["rap"], sample_leaf=c["leaf"], net=NET_CODE[c["net"]])
    a = ref.rl_run(c["d"], c["f"], p, c["seed"], c["games"], **kw)
    o = port.rl_run(c["d"], c["f"], p, c["seed"], c["games"], **kw)
    assert len(a) == len(o) and len(a) >= 2 * c["games"]
    for (qa, va), (qo, vo) in zip(a, o):
        assert np.array_equal(qa, qo) and np.array_equal(va, vo)


def test_callback_net_teacher_forcing(port, ref):
    """Both oracles driven by the SAME python callback net see identical queries and end bit-identical."""
    d, f = 1, 6
    A, H = port.num_actions(d, f), port.num_hands(d, f)
    rng = np.random.default_rng(0)
    W = rng.standard_normal((2 + A + 2 * H, H)).astype(np.float32) * 0.3
    seen = {"ref": [], "port": []}

    def make(tag):
        def fn(q):
            seen[tag].append(q.copy())
            return np.tanh(q @ W).astype(np.float32)
        return fn

    p = orc.make_params(num_iters=200, max_depth=2, linear_update=True, use_cfr=True)
    a = ref.solver(d, f, p, net=orc.NET_CALLBACK, net_fn=make("ref"))
    o = port.solver(d, f, p, net=orc.NET_CALLBACK, net_fn=make("port"))
    a.multistep()
    o.multistep()
    assert len(seen["ref"]) == len(seen["port"]) == 200
    for x, y in zip(seen["ref"], seen["port"]):
        assert np.array_equal(x, y)
    for w in (orc.GET_AVERAGE, orc.GET_LAST, orc.GET_SUM, orc.GET_REGRETS):
        assert np.array_equal(a.get(w), o.get(w))


def test_exploitability_matches(port, ref):
    p = orc.make_params(num_iters=128, max_depth=100, linear_update=True, use_cfr=True)
    s = ref.solver(1, 4, p, net=orc.NET_NONE)
    s.multistep()
    strat = s.get(orc.GET_AVERAGE)
    assert np.array_equal(ref.exploitability2(1, 4, strat), port.exploitability2(1, 4, strat))


@pytest.mark.parametrize("to_leaf", [False, True])
@pytest.mark.parametrize("d,f,depth,iters", [(1, 3, 2, 24), (1, 4, 2, 16), (1, 4, 3, 10)])
def test_recursive_strategy_bit_exact(d, f, depth, iters, to_leaf, port, ref):
    """compute_strategy_recursive(_to_leaf) (recursive_solving.cc:277-299): the port's restatement vs the reference,
    synthetic net, full-tree dense strategy bit for bit."""
    p = orc.make_params(num_iters=iters, max_depth=depth, linear_update=True, use_cfr=True)
    a = ref.strategy_recursive(d, f, p, to_leaf=to_leaf, net=orc.NET_SYNTHETIC)
    o = port.strategy_recursive(d, f, p, to_leaf=to_leaf, net=orc.NET_SYNTHETIC)
    assert a.shape == o.shape and np.array_equal(a, o)
    assert np.array_equal(ref.exploitability2(d, f, a), port.exploitability2(d, f, o))


@pytest.mark.parametrize("d,f,depth,iters,seed,root_only", [(1, 4, 2, 16, 0, False), (1, 4, 2, 33, 7, False),
                                                            (1, 4, 1, 16, 3, True), (1, 5, 2, 24, 1, False)])
def test_sampled_recursive_strategy_port_vs_reference(d, f, depth, iters, seed, root_only, ref, port):
    """compute_sampled_strategy_recursive_to_leaf (recursive_solving.cc:301-327): the port's restatement (draw order,
    per-subgame num_iters, sampling strategy for output and belief propagation, root_only depth) vs the reference."""
    from oracle import orc

    p = orc.make_params(num_iters=iters, max_depth=depth, linear_update=True, use_cfr=True)
    net = orc.NET_ZERO if root_only else orc.NET_SYNTHETIC
    a = ref.strategy_recursive_sampled(d, f, p, seed, root_only, net=net)
    o = port.strategy_recursive_sampled(d, f, p, seed, root_only, net=net)
    assert np.array_equal(a, o)


@pytest.mark.parametrize("d,f", [(1, 3), (1, 4), (2, 2)])
def test_ev2_port_vs_reference(d, f, ref, port):
    """compute_ev2 (subgame_solving.cc:931-982): the port's restatement vs the reference, two unrelated strategies."""
    from oracle import orc

    s1 = port.solver(d, f, orc.make_params(num_iters=20, max_depth=100000, linear_update=True, use_cfr=True))
    s1.multistep()
    s2 = port.solver(d, f, orc.make_params(num_iters=5, max_depth=100000, linear_update=False, use_cfr=True))
    s2.multistep()
    a, b = s1.get(orc.GET_AVERAGE), s2.get(orc.GET_LAST)
    assert np.array_equal(ref.ev2(d, f, a, b), port.ev2(d, f, a, b))
    assert np.array_equal(ref.ev2(d, f, b, a), port.ev2(d, f, b, a))


def test_immediate_regrets_port_equals_reference(port, ref):
    """compute_immediate_regrets (subgame_solving.cc:984-1050): the restatement against the compiled reference on strategy
    lists produced by recursive solving (1dx3f and 1dx4f full trees)."""
    for d, f, k in ((1, 3, 3), (1, 4, 2)):
        S = np.stack([ref.strategy_recursive(d, f, orc.make_params(num_iters=16 + 8 * i, max_depth=2, linear_update=True,
                                                                      use_cfr=True), to_leaf=True) for i in range(k)])
        a, b = ref.immediate_regrets(d, f, S), port.immediate_regrets(d, f, S)
        assert np.array_equal(a, b) and np.abs(a).max() > 0.01
