"""Loads the committed golden vectors (tests/golden/*.npz, generated from the compiled reference by
tests/golden/make_golden.py) and checks a solver implementation against them."""
import hashlib
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ARRAY_NAMES = ("average", "last", "sum", "regrets")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


_cache = {}


def load(name):
    if name not in _cache:
        _cache[name] = np.load(os.path.join(GOLDEN_DIR, name), allow_pickle=False)
    return _cache[name]


def check_solver_arrays(case_name, arrays, hand_values, exact=True, atol=0.0):
    """arrays: dict name -> dense double[N][H][A]; hand_values: [2][H].  exact => bit-for-bit (sha256)."""
    g = load("solver_cases.npz")
    hv = g[f"{case_name}/hand_values"]
    if exact:
        assert np.array_equal(hv, hand_values), (case_name, "hand_values", np.abs(hv - hand_values).max())
    else:
        np.testing.assert_allclose(hand_values, hv, rtol=0, atol=atol)
    for n in ARRAY_NAMES:
        key = f"{case_name}/{n}_sha256"
        if key not in g.files or n not in arrays:
            continue
        if exact:
            if f"{case_name}/{n}" in g.files:  # small case: give a useful diff on failure
                ref = g[f"{case_name}/{n}"]
                assert np.array_equal(ref, arrays[n]), (case_name, n, np.abs(ref - arrays[n]).max())
            assert sha(arrays[n]) == str(g[key]), (case_name, n, "sha256 mismatch")
        elif f"{case_name}/{n}" in g.files:
            np.testing.assert_allclose(arrays[n], g[f"{case_name}/{n}"], rtol=0, atol=atol)
