"""What each tree-size class of a 2 dice x 6 faces batch costs a CFR step (would a one-wavefront kernel for the small
trees pay?).  The same 2 048-lane bench-like mix as scripts/probe_cfr_phases_2d6f.py (30 % roots, the rest spread over the bids),
timed (a) whole, (b) WITHOUT its small trees (N <= 64) -- the most a free small-tree kernel could give --, (c) the small trees
alone, (d) roots alone, (e) without the roots.  Wall-clock per CFR step over 200 steps, synthetic elementwise net (its launch is
part of every figure), size-sorted launches as in the bench.  usage: probe_2d6f_class_share.py [steps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rebel_amd import capi  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
B = 2048
rng = np.random.default_rng(0)
probe = capi.Engine(2, 6, capi.make_params(num_iters=1024, max_depth=2, linear_update=True, use_cfr=True), max_lanes=B)
bids = np.where(rng.random(B) < 0.3, -1, rng.integers(0, probe.A - 2, B)).astype(np.int32)
bel = rng.dirichlet(np.ones(probe.H), size=(B, 2))
probe.set_net_synthetic()
probe.reset(bids, np.zeros(B, np.int32), bel)
sizes = np.array([probe.tree_size(i) for i in range(B)])
probe.close()


def timed(sel, label):
    n = int(sel.sum())
    if n == 0:
        return
    e = capi.Engine(2, 6, capi.make_params(num_iters=1024, max_depth=2, linear_update=True, use_cfr=True), max_lanes=n)
    e.set_net_synthetic()
    e.reset(bids[sel], np.zeros(n, np.int32), bel[sel])
    e.multistep(20)
    e.sync()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        e.multistep(steps)
        e.sync()
        best = min(best, (time.perf_counter() - t0) / steps * 1e6)
    print(f"{label:46s} {n:5d} lanes  {best:8.1f} us per step")
    e.close()
    return best


small, mid, root = sizes <= 64, (sizes > 64) & (sizes <= 160), sizes > 160
print(f"classes: N <= 64: {small.sum()} lanes (median N {int(np.median(sizes[small]))}), 64 < N <= 160: {mid.sum()} "
      f"({int(np.median(sizes[mid]))}), N > 160: {root.sum()} ({int(np.median(sizes[root]))})")
full = timed(np.ones(B, bool), "(a) the whole mix")
no_small = timed(~small, "(b) without the small trees")
timed(small, "(c) the small trees alone")
timed(root, "(d) the root-sized trees alone")
timed(~root, "(e) without the root-sized trees")
print(f"upper bound of a free small-tree kernel: {full - no_small:.1f} us of {full:.1f} us per step = {(full - no_small) / full * 100:.1f} %")
