"""The port oracle against the committed golden vectors generated from the compiled reference
(tests/golden/make_golden.py).  This is what pins the oracle on machines without /root/reference (the GPU box)."""
import numpy as np
import pytest

from oracle import orc
from tests import golden_util as G
from tests.cases import NET_CODE, RL_CASES, SOLVER_CASES, case_beliefs

NAMES = {orc.GET_AVERAGE: "average", orc.GET_LAST: "last", orc.GET_SUM: "sum", orc.GET_REGRETS: "regrets"}


@pytest.mark.parametrize("name", sorted(SOLVER_CASES))
def test_port_solver_vs_golden(name, port):
    c = SOLVER_CASES[name]
    p = orc.make_params(**c["p"])
    H = port.num_hands(c["d"], c["f"])
    s = port.solver(c["d"], c["f"], p, c.get("lb", -1), c.get("pl", 0), case_beliefs(c, H), NET_CODE[c["net"]])
    s.multistep()
    arrays = {n: s.get(w) for w, n in NAMES.items() if p.use_cfr or w != orc.GET_REGRETS}
    G.check_solver_arrays(name, arrays, np.stack([s.hand_values(0), s.hand_values(1)]), exact=True)
    assert int(G.load("solver_cases.npz")[f"{name}/tree_size"]) == s.N
    if c["net"] != "none":
        s.update_value_network()
        g = G.load("solver_cases.npz")
        assert np.array_equal(np.stack([q for q, _ in s.examples]), g[f"{name}/example_queries"])
        assert np.array_equal(np.stack([v for _, v in s.examples]), g[f"{name}/example_values"])


@pytest.mark.parametrize("name", sorted(RL_CASES))
def test_port_selfplay_vs_golden(name, port):
    c = RL_CASES[name]
    p = orc.make_params(**c["p"])
    ex = port.rl_run(c["d"], c["f"], p, c["seed"], c["games"], random_action_prob=c["rap"], sample_leaf=c["leaf"],
                     net=NET_CODE[c["net"]])
    g = G.load("rl_cases.npz")
    assert np.array_equal(np.stack([q for q, _ in ex]), g[f"{name}/queries"])
    assert np.array_equal(np.stack([v for _, v in ex]), g[f"{name}/values"])


def test_survey_probe_values():
    """Known answers recorded in SURVEY.md 8(c) from the compiled reference (12 printed digits)."""
    g = G.load("solver_cases.npz")
    np.testing.assert_allclose(g["1d4f_full_cfr_128/hand_values"][0],
                               [0.097053648032, 0.085496154555, 0.235013004993, -0.227521076657], rtol=0, atol=5e-13)
    np.testing.assert_allclose(g["1d6f_root_zero_1024/hand_values"][0],
                               [-7.121469e-6, -7.055479e-6, -6.818820e-6, -6.444485e-6, -5.722579e-6, 1.0709475e-5],
                               rtol=0, atol=5e-12)
