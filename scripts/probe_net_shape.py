"""Developer aid: the resident value-net forward (tile 5) at another game's shape -- time per launch on device buffers,
max |error| against float64 on a sample, and (RBL_NET_DBG=1) the phase stamps of the first group of each workgroup.
usage: probe_net_shape.py DICE FACES [rows] [reps] [n_layers]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, '.')
os.environ.setdefault("RBL_NET_DBG", "1")
import torch  # noqa: E402  (device buffers only)

from rebel_amd import capi  # noqa: E402

dice, faces = int(sys.argv[1]), int(sys.argv[2])
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 229376
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
n_layers = int(sys.argv[5]) if len(sys.argv) > 5 else 2


def np_net(q, layers, ln, w_out, b_out, eps=1e-5):
    from scipy.special import erf
    x = q.astype(np.float64)
    for i, (w, b) in enumerate(layers):
        x = x @ w.astype(np.float64).T + b
        g, o = ln[i]
        mu = x.mean(-1, keepdims=True)
        var = ((x - mu) ** 2).mean(-1, keepdims=True)
        x = (x - mu) / np.sqrt(var + eps) * g + o
        x = 0.5 * x * (1 + erf(x / np.sqrt(2)))
    return x @ w_out.astype(np.float64).T + b_out


e = capi.Engine(dice, faces, capi.make_params(num_iters=4, use_cfr=True, max_depth=2))
Q, H, hid = e.Q, e.H, 256
rng = np.random.default_rng(7)
layers = [(rng.uniform(-1, 1, (hid, Q)).astype(np.float32) / np.sqrt(Q), rng.uniform(-0.1, 0.1, hid).astype(np.float32))]
for _ in range(n_layers - 1):
    layers.append((rng.uniform(-1, 1, (hid, hid)).astype(np.float32) / np.sqrt(hid), rng.uniform(-0.1, 0.1, hid).astype(np.float32)))
ln = [(rng.uniform(0.5, 1.5, hid).astype(np.float32), rng.uniform(-0.2, 0.2, hid).astype(np.float32)) for _ in range(n_layers)]
w_out = rng.uniform(-1, 1, (H, hid)).astype(np.float32) / np.sqrt(hid)
b_out = rng.uniform(-0.1, 0.1, H).astype(np.float32)
q = np.zeros((rows, Q), np.float32)
q[:, 0] = rng.integers(0, 2, rows)
q[:, 1] = rng.integers(0, 2, rows)
q[np.arange(rows), 2 + rng.integers(0, e.A, rows)] = 1
q[:, 2 + e.A:2 + e.A + H] = rng.dirichlet(np.ones(H), rows)
q[:, 2 + e.A + H:] = rng.dirichlet(np.ones(H), rows)
sample = np.concatenate([np.arange(0, min(rows, 2048)), np.arange(max(0, rows - 2048), rows), rng.integers(0, rows, 4096)])
ref = np_net(q[sample], layers, ln, w_out, b_out)
qd = torch.from_numpy(q).cuda()
od = torch.empty((rows, H), dtype=torch.float32, device='cuda')
e.set_net_mlp(layers, ln, w_out, b_out)
torch.cuda.synchronize()
capi._check(e.L.rbl_net_forward_dev(e.h, qd.data_ptr(), rows, od.data_ptr()))
e.sync()
err = np.abs(od.cpu().numpy()[sample] - ref).max()
for _ in range(3):
    capi._check(e.L.rbl_net_forward_dev(e.h, qd.data_ptr(), rows, od.data_ptr()))
e.sync()
batches = []
for _b in range(6):
    t0 = time.perf_counter()
    for _ in range(reps):
        capi._check(e.L.rbl_net_forward_dev(e.h, qd.data_ptr(), rows, od.data_ptr()))
    e.sync()
    batches.append((time.perf_counter() - t0) / reps * 1e6)
us = min(batches)
print(f"{dice}d x {faces}f n_layers {n_layers} kernel {e.stats()['net_kernel']}: n_in {Q} n_out {H} rows {rows}  max|err| {err:.3e}  {us:.1f} us/launch  {us * 1e3 / rows:.4f} ns/row "
      f"(batches {' '.join(f'{b:.0f}' for b in batches)})", flush=True)
if os.environ.get("RBL_NET_DBG") == "1":
    st = e.net_debug_stamps()
    n = min(256, (rows + 63) // 64)
    names = ["stage", "L0 gemm", "-", "L0 epilogue", "hidden gemm", "-", "hidden epilogue", "output"]
    if n_layers == 3:  # resident kernel, two hidden layers: the second layer's stretch has no stamp of its own
        names[5] = "hidden-1 epilogue + hidden-2 gemm (ring)"
    d = np.diff(st[:n, :9], axis=1)
    print("first group of each workgroup, median cycles:", {nm: int(np.median(d[:, i])) for i, nm in enumerate(names) if nm != "-"},
          "sum", int(np.median(st[:n, 8] - st[:n, 0])))
    groups = np.array([len(range(b, (rows + 63) // 64, 256)) for b in range(n)])
    print("steady state: cycles per group", int(np.median((st[:n, 12] - st[:n, 0]) / groups)))
