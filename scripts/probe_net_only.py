"""Runs only the value-net forward on a fixed batch (rows from argv, default 270336) a few times -- for rocprofv3 PMC runs."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rebel_amd import capi
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 270336
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'net2_1d6f.npz'))
layers = [(g["body__0__weight"], g["body__0__bias"]), (g["body__4__weight"], g["body__4__bias"])]
ln = [(g["body__1__weight"], g["body__1__bias"]), (g["body__5__weight"], g["body__5__bias"])]
e = capi.Engine(1, 6, capi.make_params(num_iters=4, use_cfr=True))
e.set_net_mlp(layers, ln, g["output__weight"], g["output__bias"])
q = np.tile(g["queries"], (rows // len(g["queries"]) + 1, 1))[:rows]
for _ in range(reps):
    t = time.time(); y = e.net_forward(q); print('forward incl. copies ms', (time.time() - t) * 1e3)
