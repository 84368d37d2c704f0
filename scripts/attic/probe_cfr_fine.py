"""Developer aid: raw stamps of the flat CFR kernel (RBL_CFR_DBG=1).  Build cfr_flat_kernel.hip with -DRBL_FINE=1 (stamps
inside the reach phase: rows, pseudo-leaves, terminals, barrier per level) or -DRBL_FINE=2 (inside the bottom-up sweep and the
query phase) and link it into a scratch copy of librebel_hip.so; the product build has 9 stamps (scripts/probe_cfr_phases_2d6f.py).
STEPS=n picks the traverser parity of the last step."""
import os
import sys

import numpy as np

os.environ["RBL_CFR_DBG"] = "1"
sys.path.insert(0, '.')
from rebel_amd import capi  # noqa: E402

B = 512
e = capi.Engine(2, 6, capi.make_params(num_iters=1024, max_depth=2, linear_update=True, use_cfr=True), max_lanes=B)
e.set_net_synthetic()
rng = np.random.default_rng(0)
bids = np.where(np.arange(B) % 2 == 0, -1, 18).astype(np.int32)
e.reset(bids, bids * 0, rng.dirichlet(np.ones(e.H), size=(B, 2)))
e.multistep(int(os.environ.get("STEPS", "9")))
e.sync()
d = e.debug_stamps()[:B]
for name, sel in (("root (N=325)", bids == -1), ("bid 18", bids == 18)):
    dd = np.diff(d[sel], axis=1)
    print(name, "N", e.tree_size(int(np.nonzero(sel)[0][0])), "median diffs:", [int(np.median(dd[:, i])) for i in range(15)])
