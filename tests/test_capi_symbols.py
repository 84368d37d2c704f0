"""CPU checks of the C-ABI boundary: librebel_hip.so loads without a GPU, exports every symbol include/rebel_hip.h
declares, host-only entry points (rules, BFS tree) work and match the oracle, compute entry points fail loudly without
a device, and nothing under rebel_amd/ reaches into oracle/ (the oracle is test infrastructure)."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "rebel_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rbl_[a-z_0-9]+)\s*\(", src)))


def test_header_and_binding_agree():
    from rebel_amd import capi

    assert _header_symbols() == sorted(capi.SYMBOLS)


def test_library_exports_every_declared_symbol():
    from rebel_amd import capi

    L = capi.lib()
    for name in _header_symbols():
        assert hasattr(L, name), name
    assert b"gfx950" in L.rbl_build_info()


def test_host_rules_and_tree_match_oracle(port):
    from rebel_amd import capi

    L = capi.lib()
    for d, f in [(1, 4), (1, 6), (2, 3), (2, 6), (1, 1), (3, 2)]:
        assert L.rbl_num_actions(d, f) == port.num_actions(d, f)
        assert L.rbl_num_hands(d, f) == port.num_hands(d, f)
        assert L.rbl_query_size(d, f) == 2 + port.num_actions(d, f) + 2 * port.num_hands(d, f)
    for d, f, rb, pl, depth in [(1, 2, -1, 0, 100), (2, 6, 22, 0, 2), (2, 6, 21, 1, 3), (1, 6, -1, 0, 2),
                                (1, 6, 5, 1, 2), (1, 4, -1, 0, 100), (2, 3, 11, 0, 1), (1, 6, 3, 0, 0)]:
        assert np.array_equal(capi.unroll_tree(d, f, rb, pl, depth), port.unroll_tree(d, f, rb, pl, depth))


def test_tree_known_answers():
    """tree_test.cc:20-34: the 1 die x 2 faces full tree has 31 nodes with these children lists."""
    from rebel_amd import capi

    t = capi.unroll_tree(1, 2, -1, 0, 100)
    ch = lambda i: list(range(t[i][2], t[i][3]))
    assert len(t) == 31
    assert ch(0) == [1, 2, 3, 4] and ch(1) == [5, 6, 7, 8] and ch(2) == [9, 10, 11]
    assert ch(15) == [25, 26] and ch(16) == [27] and ch(25) == [30]


def test_compute_fails_loudly_without_device():
    from rebel_amd import capi

    if capi.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(capi.RebelError, match="HIP|device"):
        capi.Engine(1, 4, capi.make_params(num_iters=4, use_cfr=True))


def test_product_never_touches_the_oracle():
    """No import / include / link / dlopen of anything under oracle/ from the product tree (comments may cite it)."""
    pat = re.compile(r"(from|import)\s+oracle\b|#include\s*[\"<][^\n]*(orc_api|oracle|cfr_oracle)|liboracle|libref_driver|oracle/_|dlopen|CDLL\([^)]*orac")
    for base, _, files in os.walk(os.path.join(ROOT, "rebel_amd")):
        if "_build" in base or "__pycache__" in base:
            continue
        for fn in files:
            if fn.endswith((".py", ".cc", ".h", ".hip", ".cpp")) or fn == "Makefile":
                lines = open(os.path.join(base, fn), errors="ignore").read().splitlines()
                cmt = ("#",) if fn.endswith(".py") or fn == "Makefile" else ("//", "*", "/*")
                code = [l for l in lines if not l.strip().startswith(cmt)]
                hits = [l for l in code if pat.search(l)]
                assert not hits, (fn, hits[:3])


def _combine_reference(dice, faces, levels, tops):
    """The recombination sweep written out in numpy (the round-3 host implementation): BRSolver::compute_br,
    subgame_solving.cc:326-355, over the top of the tree with every dealt node's value taken from its owner."""
    from rebel_amd import capi

    tree = capi.unroll_tree(dice, faces, -1, 0, levels)
    owner = tops[0][1]
    H = tops[0][0].shape[2]
    out = np.zeros(2)
    for t in range(2):
        val = np.array(tops[0][0][t], dtype=np.float64, copy=True)
        for n in range(len(tree) - 1, -1, -1):
            last_bid, player, cb, ce = (int(x) for x in tree[n][:4])
            if cb == ce:
                if owner[n] >= 0:
                    val[n] = tops[owner[n]][0][t][n]
                continue
            v = val[cb].copy()
            for c in range(cb + 1, ce):
                v = np.where(val[c] > v, val[c], v) if player == t else v + val[c]
            val[n] = v
        s = 0.0
        for x in val[0]:
            s += x
        out[t] = s / H
    return out


@pytest.mark.parametrize("dice,faces,depth,deal", [(1, 4, 2, 1), (1, 4, 2, 2), (1, 6, 2, 2), (2, 3, 2, 2), (1, 4, 3, 1),
                                                   (1, 4, 2, 8), (2, 6, 2, 2)])
def test_exploitability_combine_is_host_code_of_the_c_abi(dice, faces, depth, deal):
    """rbl_exploitability_combine / rbl_exploitability_top_nodes (the recombination of the shards existed
    only in Python): host-only entry points, so they are exercised here without a GPU -- against the numpy sweep, on random
    shard values and a random owner map."""
    from rebel_amd import capi

    L = capi.lib()
    M = L.rbl_exploitability_top_nodes(dice, faces, depth, deal)
    tree = capi.unroll_tree(dice, faces, -1, 0, depth * deal)
    assert M == len(tree)
    H = L.rbl_num_hands(dice, faces)
    rng = np.random.default_rng(dice * 100 + faces * 10 + deal)
    n_shards = 5
    childless = np.array([int(r[2]) == int(r[3]) for r in tree])
    liar = L.rbl_num_actions(dice, faces) - 1
    dealt = childless & np.array([int(r[0]) != liar for r in tree])
    owner = np.where(dealt, rng.integers(0, n_shards, M), -1).astype(np.int32)
    tops = [(rng.normal(size=(2, M, H)), owner) for _ in range(n_shards)]
    got = capi.combine_exploitability(dice, faces, depth, tops, deal_levels=deal)
    assert np.array_equal(got, _combine_reference(dice, faces, depth * deal, tops))
    with pytest.raises(Exception):
        capi.combine_exploitability(dice, faces, depth, [(tops[0][0][:, :-1], owner[:-1])], deal_levels=deal)
