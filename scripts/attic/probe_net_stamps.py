"""Developer aid: phase stamps of the pipelined net kernel (library built with -DRBL_PIPE_STAMPS, RBL_NET_DBG=1)."""
import os
import sys

import numpy as np

sys.path.insert(0, '.')
os.environ["RBL_NET_DBG"] = "1"
os.environ["RBL_MLP_TILE"] = sys.argv[2] if len(sys.argv) > 2 else "6"
os.environ["RBL_MLP_STAGGER"] = sys.argv[3] if len(sys.argv) > 3 else "0"
from rebel_amd import capi  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 589824
g = np.load('tests/golden/net2_1d6f.npz')
layers = [(g["body__0__weight"], g["body__0__bias"]), (g["body__4__weight"], g["body__4__bias"])]
ln = [(g["body__1__weight"], g["body__1__bias"]), (g["body__5__weight"], g["body__5__bias"])]
e = capi.Engine(1, 6, capi.make_params(num_iters=4, use_cfr=True))
e.set_net_mlp(layers, ln, g["output__weight"], g["output__bias"])
q = np.tile(g["queries"], (rows // len(g["queries"]) + 1, 1))[:rows]
for _ in range(3):
    e.net_forward(q)
st = e.net_debug_stamps()
names = ["chunksA", "tailA", "barrier1", "chunksB", "tailB", "barrier2"]
for w in range(2):
    d = np.diff(st[:256, w * 8:w * 8 + 7], axis=1)
    print(f"wave {4 * w}: median cycles over 256 workgroups:", {n: int(np.median(d[:, i])) for i, n in enumerate(names)},
          "total", int(np.median(st[:256, w * 8 + 6] - st[:256, w * 8])))
