"""Shared parity-case table: the same seeded cases drive (a) the golden-vector generator (reference -> fixtures),
(b) the port-vs-reference pin, (c) the HIP-vs-oracle parity tests.  Plain data, no imports from oracle/ or the product.
"""
import numpy as np

# name -> dict(dice, faces, params kwargs, root_last_bid, root_player, beliefs seed (None = uniform), net)
SOLVER_CASES = {
    "1d4f_root_zero_128": dict(d=1, f=4, p=dict(num_iters=128, max_depth=2, linear_update=True, use_cfr=True),
                               net="zero"),
    "1d4f_root_syn_128": dict(d=1, f=4, p=dict(num_iters=128, max_depth=2, linear_update=True, use_cfr=True),
                              net="synthetic"),
    "1d6f_root_zero_1024": dict(d=1, f=6, p=dict(num_iters=1024, max_depth=2, linear_update=True, use_cfr=True),
                                net="zero"),
    "1d6f_root_syn_1024": dict(d=1, f=6, p=dict(num_iters=1024, max_depth=2, linear_update=True, use_cfr=True),
                               net="synthetic"),
    "1d6f_bid6_p1_syn_300": dict(d=1, f=6, p=dict(num_iters=300, max_depth=2, linear_update=True, use_cfr=True),
                                 lb=6, pl=1, bseed=3, net="synthetic"),
    "1d6f_bid10_syn_64": dict(d=1, f=6, p=dict(num_iters=64, max_depth=2, linear_update=True, use_cfr=True),
                              lb=10, pl=0, bseed=4, net="synthetic"),  # terminal-adjacent: no pseudo-leaves
    "1d6f_bid11_syn_16": dict(d=1, f=6, p=dict(num_iters=16, max_depth=2, linear_update=True, use_cfr=True),
                              lb=11, pl=1, bseed=5, net="synthetic"),  # only the liar call is left: N=2
    "2d3f_bid2_p1_syn_256": dict(d=2, f=3, p=dict(num_iters=256, max_depth=2, linear_update=True, use_cfr=True),
                                 lb=2, pl=1, bseed=1, net="synthetic"),
    "2d3f_root_syn_1024": dict(d=2, f=3, p=dict(num_iters=1024, max_depth=2, linear_update=True, use_cfr=True),
                               net="synthetic"),
    "2d6f_root_syn_48": dict(d=2, f=6, p=dict(num_iters=48, max_depth=2, linear_update=True, use_cfr=True),
                             bseed=2, net="synthetic"),
    "1d4f_full_cfr_128": dict(d=1, f=4, p=dict(num_iters=128, max_depth=100, linear_update=True, use_cfr=True),
                              net="none"),
    "1d4f_dcfr_syn_64": dict(d=1, f=4, p=dict(num_iters=64, max_depth=2, use_cfr=True, dcfr=True, dcfr_alpha=1.5,
                                              dcfr_beta=0.5, dcfr_gamma=2.0), net="synthetic"),
    "1d4f_depth3_plain_syn_64": dict(d=1, f=4, p=dict(num_iters=64, max_depth=3, use_cfr=True), net="synthetic"),
    "1d5f_depth1_syn_33": dict(d=1, f=5, p=dict(num_iters=33, max_depth=1, linear_update=True, use_cfr=True),
                               bseed=7, net="synthetic"),
    "1d4f_fp_linear_syn_64": dict(d=1, f=4, p=dict(num_iters=64, max_depth=2, linear_update=True, use_cfr=False),
                                  net="synthetic"),
}

# name -> dict(d, f, params, seed, games, random_action_prob, sample_leaf, net)
RL_CASES = {
    "rl_1d4f_syn_seed7": dict(d=1, f=4, p=dict(num_iters=128, max_depth=2, linear_update=True, use_cfr=True), seed=7,
                              games=20, rap=0.25, leaf=True, net="synthetic"),
    "rl_1d4f_zero_seed7": dict(d=1, f=4, p=dict(num_iters=128, max_depth=2, linear_update=True, use_cfr=True), seed=7,
                               games=20, rap=0.25, leaf=True, net="zero"),
    "rl_1d6f_syn_seed0": dict(d=1, f=6, p=dict(num_iters=1024, max_depth=2, linear_update=True, use_cfr=True), seed=0,
                              games=6, rap=0.25, leaf=True, net="synthetic"),
    "rl_2d3f_syn_seed5": dict(d=2, f=3, p=dict(num_iters=200, max_depth=2, linear_update=True, use_cfr=True), seed=5,
                              games=8, rap=0.25, leaf=True, net="synthetic"),
    "rl_1d6f_single_seed11": dict(d=1, f=6, p=dict(num_iters=100, max_depth=2, linear_update=True, use_cfr=True),
                                  seed=11, games=8, rap=0.5, leaf=False, net="synthetic"),
    "rl_1d5f_depth3_seed5": dict(d=1, f=5, p=dict(num_iters=64, max_depth=3, linear_update=True, use_cfr=True),
                                 seed=5, games=8, rap=0.25, leaf=True, net="synthetic"),
    "rl_1d4f_depth1_seed2": dict(d=1, f=4, p=dict(num_iters=50, max_depth=1, linear_update=True, use_cfr=True),
                                 seed=2, games=10, rap=1.0, leaf=True, net="synthetic"),
}

NET_CODE = {"zero": 0, "synthetic": 2, "none": 4}


def case_beliefs(case, H):
    """Initial beliefs for a case: uniform, or Dirichlet(1) per player from default_rng(bseed)."""
    bseed = case.get("bseed")
    if bseed is None:
        return np.full((2, H), 1.0 / H)
    return np.random.default_rng(bseed).dirichlet(np.ones(H), size=2)
