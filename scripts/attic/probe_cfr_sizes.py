"""CFR step time (synthetic net, 4096 lanes) for uniform tree sizes and for a half big / half small mix.
RBL_CFR_ROWS_FIT=0 disables the size-sorted parts with fitted launch shapes."""
import sys, time, numpy as np
sys.path.insert(0, '.')
from rebel_amd import capi
B = 4096
cases = [("all root", [-1] * B), ("all bid 2", [2] * B), ("all bid 5", [5] * B), ("all bid 8", [8] * B),
         ("half root / half bid 5, interleaved", [-1, 5] * (B // 2)), ("1/4 root, 3/4 bid 6", ([-1] + [6] * 3) * (B // 4))]
for name, bids in cases:
    e = capi.Engine(1, 6, capi.make_params(num_iters=1024, max_depth=2, linear_update=True, use_cfr=True), max_lanes=B)
    e.set_net_synthetic()
    e.reset(bids, [0] * B, np.full((B, 2, e.H), 1.0 / e.H))
    e.multistep(32); e.sync()
    t = time.time(); e.multistep(256); e.sync(); dt = time.time() - t
    print(f"{name:40s} mean N {np.mean([e.tree_size(i) for i in range(0, B, 97)]):5.1f}  {dt / 256 * 1e6:6.1f} us per iteration")
