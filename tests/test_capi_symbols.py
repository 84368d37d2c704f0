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
