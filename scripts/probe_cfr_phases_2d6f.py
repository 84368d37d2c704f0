"""Developer aid: phase stamps of the CFR row kernel (GS variant) on a mix of 2 dice x 6 faces subgames (RBL_CFR_DBG=1).
usage: probe_cfr_phases_2d6f.py [steps] [lanes]"""
import os
import sys
import time

import numpy as np

os.environ["RBL_CFR_DBG"] = "1"
sys.path.insert(0, '.')
from rebel_amd import capi  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 9
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
e = capi.Engine(2, 6, capi.make_params(num_iters=1024, max_depth=2, linear_update=True, use_cfr=True), max_lanes=B)
e.set_net_synthetic()
rng = np.random.default_rng(0)
# self-play visits public states along games: a third of the lanes at the root, the rest spread over the bids (measured mix
# of the bench: ~30 % root-sized trees)
bids = np.where(rng.random(B) < (2.0 if os.environ.get('RBL_PROBE_ROOT') else 0.3), -1, rng.integers(0, e.A - 2, B)).astype(np.int32)
e.reset(bids, (bids + 1) % 2 * 0, rng.dirichlet(np.ones(e.H), size=(B, 2)))
e.multistep(steps)
e.sync()
d = e.debug_stamps()
names = ["staged", "reach", "-", "-", "bottom-up", "new reach", "write-back", "queries"]
dt = np.diff(d[:B, :9], axis=1)
sizes = np.array([e.tree_size(i) for i in range(B)])
for lo, hi in ((0, 64), (64, 160), (160, 400)):
    sel = (sizes > lo) & (sizes <= hi)
    if not sel.any():
        continue
    print(f"trees of {lo} < N <= {hi}: {sel.sum()} lanes, median N {int(np.median(sizes[sel]))}; per-phase shader-clock cycles (clock64 = s_memtime) (median, p90):")
    for i, n in enumerate(names):
        if n != "-":
            print(f"  {n:12s} {np.median(dt[sel, i]):8.0f} {np.percentile(dt[sel, i], 90):8.0f}")
    tot = d[:B, 8][sel] - d[:B, 0][sel]
    print("  total        %8.0f %8.0f" % (np.median(tot), np.percentile(tot, 90)))
print("launch span (max end - min start), shader-clock cycles:", d[:B, 8].max() - d[:B, 0].min())
os.environ["RBL_CFR_DBG"] = "0"
t0 = time.perf_counter()
e.multistep(40)
e.sync()
print(f"wall per step incl. synthetic net: {(time.perf_counter() - t0) / 40 * 1e6:.1f} us for {B} lanes")
