"""The reference's own gtest known answers, restated against the oracle API (both the compiled reference and our
port must satisfy them).  Sources: /root/reference/csrc/liars_dice/{liars_dice_test.cc,tree_test.cc,
subgame_solving_test.cc,recursive_solving_test.cc} -- cited per test.
"""
import numpy as np

from oracle import orc


# ---------------------------------------------------------------- liars_dice_test.cc:46-121 (2 dice x 6 faces)
def test_unpack_action(any_oracle):
    o = any_oracle
    assert o.unpack_action(2, 6, 0) == (1, 0)
    assert o.unpack_action(2, 6, 1) == (1, 1)
    assert o.unpack_action(2, 6, 6) == (2, 0)


def test_bid_ranges(any_oracle):
    o = any_oracle
    A = o.num_actions(2, 6)
    assert A == 4 * 6 + 1
    assert o.bid_range(2, 6, -1) == (0, 4 * 6)          # root: liar not allowed
    assert o.bid_range(2, 6, 0) == (1, 4 * 6 + 1)
    assert o.bid_range(2, 6, 11) == (12, 4 * 6 + 1)
    assert o.bid_range(2, 6, A - 1) == (4 * 6 + 1, 4 * 6 + 1)  # after liar: empty


def test_player_alternates(any_oracle):
    t = any_oracle.unroll_tree(2, 6, -1, 0, 3)
    for n in t[1:]:
        assert n[1] == 1 - t[n[4]][1]


def test_num_matches(any_oracle):
    o = any_oracle
    H = o.num_hands(2, 6)
    assert [o.num_matches(2, 6, 0, f) for f in range(6)] == [2, 0, 0, 0, 0, 0]
    assert [o.num_matches(2, 6, H - 1, f) for f in range(6)] == [2, 2, 2, 2, 2, 2]  # two wild sixes
    assert [o.num_matches(2, 6, 0 * 6 + 5, f) for f in range(6)] == [2, 1, 1, 1, 1, 1]


# ---------------------------------------------------------------- tree_test.cc
def _children(t, i):
    return list(range(t[i][2], t[i][3]))


def test_unroll_full_1d2f(any_oracle):  # tree_test.cc:20-34
    t = any_oracle.unroll_tree(1, 2, -1, 0, 1 + any_oracle.num_actions(1, 2))
    assert len(t) == 31
    assert _children(t, 0) == [1, 2, 3, 4]
    assert _children(t, 1) == [5, 6, 7, 8]
    assert _children(t, 2) == [9, 10, 11]
    assert _children(t, 15) == [25, 26]
    assert _children(t, 16) == [27]
    assert _children(t, 25) == [30]


def test_unroll_depths_2d6f(any_oracle):  # tree_test.cc:36-105
    o = any_oracle
    t0 = o.unroll_tree(2, 6, 22, 0, 0)
    assert len(t0) == 1 and t0[0][4] == -1 and _children(t0, 0) == [] and t0[0][0] == 22
    t1 = o.unroll_tree(2, 6, 22, 0, 1)
    assert len(t1) == 3 and _children(t1, 0) == [1, 2] and t1[1][4] == 0 and t1[2][4] == 0
    t2 = o.unroll_tree(2, 6, 22, 0, 2)
    assert len(t2) == 4 and _children(t2, 0) == [1, 2] and t2[3][4] == 1
    t3 = o.unroll_tree(2, 6, 21, 0, 2)
    assert len(t3) == 7
    assert _children(t3, 0) == [1, 2, 3] and _children(t3, 1) == [4, 5] and _children(t3, 2) == [6]


def test_tree_is_breadth_first(any_oracle):  # tree_test.cc:107-125
    o = any_oracle
    full = o.unroll_tree(1, 5, -1, 0, 1 + o.num_actions(1, 5))
    for depth in range(0, 20):
        sub = o.unroll_tree(1, 5, -1, 0, depth)
        for i, n in enumerate(sub):
            assert n[0] == full[i][0] and n[1] == full[i][1]
            if n[3] > n[2]:
                assert (n[2], n[3], n[4]) == (full[i][2], full[i][3], full[i][4])


# ---------------------------------------------------------------- subgame_solving_test.cc:48-104
def _true_matches(dice, faces, hand, face):
    m = 0
    for _ in range(dice):
        d = hand % faces
        m += d == face or d == faces - 1
        hand //= faces
    return m


def _check_terminal_eval(o, dice, faces):
    H, A = o.num_hands(dice, faces), o.num_actions(dice, faces)
    for ophand in range(H):
        beliefs = np.zeros(H)
        beliefs[ophand] = 1
        for bet in range(A - 1):
            qty, face = o.unpack_action(dice, faces, bet)
            values = o.win_probability(dice, faces, bet, beliefs)
            for my in range(H):
                m = _true_matches(dice, faces, my, face) + _true_matches(dice, faces, ophand, face)
                assert values[my] == (1.0 if m >= qty else 0.0)


def test_terminal_eval_1d6f(any_oracle):
    _check_terminal_eval(any_oracle, 1, 6)


def test_terminal_eval_2d3f(any_oracle):
    _check_terminal_eval(any_oracle, 2, 3)


def _exploitability(o, d, f, params, net=orc.NET_NONE):
    s = o.solver(d, f, params, net=net)
    s.multistep()
    e = o.exploitability2(d, f, s.get(orc.GET_AVERAGE))
    return (e[0] + e[1]) / 2.0


def test_fp_one_die_one_face(any_oracle):  # :106-142
    for linear in (False, True):
        v = _exploitability(any_oracle, 1, 1, orc.make_params(num_iters=3500, max_depth=100, linear_update=linear))
        assert 0.0 <= v < 1e-3


def test_fp_one_die_two_faces(any_oracle):  # :144-160
    v = _exploitability(any_oracle, 1, 2, orc.make_params(num_iters=10000, max_depth=1000))
    assert 0.0 <= v < 1e-3


def test_cfr_one_die_two_faces(any_oracle):  # :162-179
    p = orc.make_params(num_iters=180, max_depth=1000, linear_update=True, use_cfr=True)
    v = _exploitability(any_oracle, 1, 2, p)
    assert 0.0 <= v < 1e-3


def test_fp_one_die_three_faces_linear(any_oracle):  # :210-225
    v = _exploitability(any_oracle, 1, 3, orc.make_params(num_iters=1 << 12, max_depth=1000, linear_update=True))
    assert 0.0 <= v < 2e-3


def test_query_roundtrip(any_oracle):  # :267-296 (decode side = deserialize_query, subgame_solving.cc:910-929)
    o = any_oracle
    d, f = 1, 3
    H, A = o.num_hands(d, f), o.num_actions(d, f)
    b1 = np.arange(H, dtype=np.float64)
    b2 = b1 + 0.5
    b1, b2 = b1 / b1.sum(), b2 / b2.sum()
    tree = o.unroll_tree(d, f, -1, 0, 1 + A)
    for traverser in (0, 1):
        for n in tree:
            if n[0] == A - 1:
                continue
            q = o.get_query(d, f, traverser, int(n[0]), int(n[1]), b1, b2)
            assert int(q[0] + 0.5) == n[1] and int(q[1] + 0.5) == traverser
            hot = np.nonzero(q[2:2 + A] > 0.5)[0]
            assert (hot.tolist() == [n[0]]) if n[0] >= 0 else (len(hot) == 0)
            np.testing.assert_allclose(q[2 + A:2 + A + H], b1, atol=1e-6)
            np.testing.assert_allclose(q[2 + A + H:], b2, atol=1e-6)


def test_prob_normalization_tiny(any_oracle):  # :298-310
    probs = [2.93185e-81, 3.00956e-81, 3.17805e-81, 8.80785e-81]
    od, of = any_oracle.normalize_safe(probs, 1e-80)
    assert abs(od.sum() - 1.0) < 1e-10
    assert abs(float(np.sum(of.astype(np.float64))) - 1.0) < 1e-6


# ---------------------------------------------------------------- recursive_solving_test.cc:37-68
def test_rl_runner_zero_net(any_oracle):
    p = orc.make_params(num_iters=100, max_depth=1, linear_update=True)  # FP, single-state sampling (pybind default)
    ex = any_oracle.rl_run(1, 3, p, seed=0, num_games=10, random_action_prob=1.0, sample_leaf=False)
    assert len(ex) >= 20 and len(ex) % 2 == 0
    p = orc.make_params(num_iters=100, max_depth=2, linear_update=True)
    ex = any_oracle.rl_run(1, 3, p, seed=0, num_games=10, random_action_prob=1.0, sample_leaf=True)
    assert len(ex) >= 20 and len(ex) % 2 == 0
