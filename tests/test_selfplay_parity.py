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


def _run_lanes(c, seeds, games):
    from rebel_amd import capi

    e = capi.Engine(c["d"], c["f"], capi.make_params(**c["p"]), max_lanes=len(seeds))
    e.set_net_synthetic() if c["net"] == "synthetic" else e.set_net_zero()
    sp = capi.SelfPlay(e, seeds, random_action_prob=c["rap"], sample_leaf=c["leaf"])
    per_lane = [[] for _ in seeds]
    done_at = [None] * len(seeds)  # number of examples when the lane finished `games` games
    finished = [0] * len(seeds)
    while any(d is None for d in done_at):
        _, lanes, q, v = sp.advance()
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
