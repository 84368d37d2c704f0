"""Developer aid: phase stamps of the one-wavefront CFR kernel on root lanes (RBL_PROBE_DICE / RBL_PROBE_FACES pick the game).
usage: probe_cfr_phases.py [steps] [lanes]"""
import os, sys, numpy as np
os.environ["RBL_CFR_DBG"] = "1"
sys.path.insert(0, '.')
from rebel_amd import capi
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
e = capi.Engine(int(os.environ.get('RBL_PROBE_DICE', 1)), int(os.environ.get('RBL_PROBE_FACES', 6)), capi.make_params(num_iters=1024, max_depth=2, linear_update=True, use_cfr=True), max_lanes=B)
e.set_net_synthetic()
e.reset([-1]*B, [0]*B, np.full((B, 2, e.H), 1.0/e.H))
e.multistep(int(sys.argv[1]) if len(sys.argv) > 1 else 9); e.sync()
d = e.debug_stamps()
names = ["staged", "reach", "leaf scalers", "leaf values", "bottom-up", "new reach", "write-back", "queries"]
dt = np.diff(d[:, :9], axis=1)
print("per-phase shader-clock cycles (median over lanes, p90):")
for i, n in enumerate(names):
    print(f"  {n:14s} {np.median(dt[:, i]):9.0f} {np.percentile(dt[:, i], 90):9.0f}")
tot = d[:, 8] - d[:, 0]
print("  total          %9.0f %9.0f" % (np.median(tot), np.percentile(tot, 90)))
print("launch span (max end - min start) cycles:", d[:, 8].max() - d[:, 0].min())
