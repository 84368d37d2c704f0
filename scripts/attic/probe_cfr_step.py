"""CFR step time alone (synthetic elementwise net) on B lanes of 1dx6f: root lanes and a self-play-like shape mix."""
import os, sys, time
import numpy as np
sys.path.insert(0, '.')
from rebel_amd import capi
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
d, f = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1, 6)
p = capi.make_params(num_iters=1024, max_depth=2, linear_update=True, use_cfr=True)
e = capi.Engine(d, f, p, max_lanes=B)
e.set_net_synthetic()
rng = np.random.default_rng(0)
for name, roots in (("root", [-1] * B), ("mix", np.where(rng.random(B) < 0.5, -1, rng.integers(0, e.A - 2, B)).tolist())):
    e.reset(roots, [0] * B, np.full((B, 2, e.H), 1.0 / e.H))
    e.multistep(16); e.sync()
    t0 = time.perf_counter(); e.multistep(200); e.sync(); dt = time.perf_counter() - t0
    print(f"WAVE={os.environ.get('RBL_CFR_WAVE','1')} PARTS={os.environ.get('RBL_PARTS','2')} {d}dx{f}f {name:4s} lanes={B}: {dt/200*1e6:7.1f} us per iteration (CFR step + synthetic net)")
