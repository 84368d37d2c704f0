import os, sys, numpy as np
os.environ["RBL_NET_DBG"] = "1"
sys.path.insert(0, '.')
from rebel_amd import capi
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 270336
g = np.load('tests/golden/net2_1d6f.npz')
layers = [(g["body__0__weight"], g["body__0__bias"]), (g["body__4__weight"], g["body__4__bias"])]
ln = [(g["body__1__weight"], g["body__1__bias"]), (g["body__5__weight"], g["body__5__bias"])]
e = capi.Engine(1, 6, capi.make_params(num_iters=4, use_cfr=True))
e.set_net_mlp(layers, ln, g["output__weight"], g["output__bias"])
q = np.tile(g["queries"], (rows // len(g["queries"]) + 1, 1))[:rows]
for _ in range(3): e.net_forward(q)
d = e.net_debug_stamps()
n = min(1024, (rows + 63) // 64)
d = d[:n]
d = d[d[:, 0] != 0]  # persistent kernels launch fewer workgroups than there are 64-row groups
n = len(d)
names = ["stage queries", "L0 gemm", "L0 y write", "L0 epilogue", "H gemm", "H y write", "H epilogue", "output"]
dt = np.diff(d[:, :9], axis=1)
print(f"rows={rows}: per-phase cycles over the first {n} workgroups (median, p90)")
for i, nm in enumerate(names): print(f"  {nm:14s} {np.median(dt[:, i]):8.0f} {np.percentile(dt[:, i], 90):8.0f}")
if d.shape[1] > 11 and d[:, 9:12].any():
    print("  output detail (from hidden-epilogue end): mfma done %d, partial barrier %d, stores issued %d, end %d" % tuple(np.median(d[:, k] - d[:, 7]) for k in (9, 10, 11, 8)))
if d.shape[1] > 12 and d[:, 12].any():
    ng = -(-rows // 64); per = np.maximum(1, (ng - np.arange(n) + n - 1) // max(n, 1)) if ng > n else np.ones(n)
    per = np.array([len(range(b, ng, min(ng, 256))) for b in range(n)])
    print("  steady state: %.0f cycles per 64-row group (whole workgroup / groups, median over workgroups)" % np.median((d[:, 12] - d[:, 0]) / per))
print("  total          %8.0f %8.0f" % (np.median(d[:, 8] - d[:, 0]), np.percentile(d[:, 8] - d[:, 0], 90)))
