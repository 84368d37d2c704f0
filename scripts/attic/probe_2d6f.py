"""2 dice x 6 faces: CFR step time on root lanes (synthetic elementwise net), row kernel (global-state variant) vs the generic
scratch-slab kernel (RBL_CFR_ROWS=0); and an end-to-end self-play figure with the real net."""
import os, sys, time
import numpy as np
sys.path.insert(0, '.')
from rebel_amd import capi
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
p = capi.make_params(num_iters=2048, max_depth=2, linear_update=True, use_cfr=True)
e = capi.Engine(2, 6, p, max_lanes=B)
e.set_net_synthetic()
e.reset([-1] * B, [0] * B, np.full((B, 2, e.H), 1.0 / e.H))
e.multistep(8); e.sync()
t0 = time.perf_counter(); e.multistep(64); e.sync(); dt = time.perf_counter() - t0
print(f"ROWS={os.environ.get('RBL_CFR_ROWS','1')} lanes={B}: {dt/64*1e6:.1f} us per iteration (CFR step + synthetic net), {B*64/dt/1e6:.2f} M lane-steps/s")
