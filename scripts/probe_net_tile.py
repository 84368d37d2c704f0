"""Value-net forward variants side by side on one GPU: max |error| against the float64 restatement of Net2 on a sample
of rows, and the time per launch on device-resident buffers (host clock around `reps` asynchronous launches + one sync).
usage: probe_net_tile.py [rows] [reps] [variant ...]   variant = TILE[:STAGGER], e.g. 5 6 6:2"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import torch  # noqa: E402  (device buffers only)

from rebel_amd import capi  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 589824
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
variants = sys.argv[3:] or ["5", "6", "6:2"]


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


e = capi.Engine(1, 6, capi.make_params(num_iters=4, use_cfr=True))
Q, H, hid = e.Q, e.H, 256
rng = np.random.default_rng(7)
layers = [(rng.uniform(-1, 1, (hid, Q)).astype(np.float32) / np.sqrt(Q), rng.uniform(-0.1, 0.1, hid).astype(np.float32)),
          (rng.uniform(-1, 1, (hid, hid)).astype(np.float32) / np.sqrt(hid), rng.uniform(-0.1, 0.1, hid).astype(np.float32))]
ln = [(rng.uniform(0.5, 1.5, hid).astype(np.float32), rng.uniform(-0.2, 0.2, hid).astype(np.float32)) for _ in range(2)]
w_out = rng.uniform(-1, 1, (H, hid)).astype(np.float32) / np.sqrt(hid)
b_out = rng.uniform(-0.1, 0.1, H).astype(np.float32)
q = np.zeros((rows, Q), np.float32)
q[:, 0] = rng.integers(0, 2, rows)
q[:, 1] = rng.integers(0, 2, rows)
q[np.arange(rows), 2 + rng.integers(0, e.A, rows)] = 1
q[:, 2 + e.A:2 + e.A + H] = rng.dirichlet(np.ones(H), rows)
q[:, 2 + e.A + H:] = rng.dirichlet(np.ones(H), rows)
sample = np.concatenate([np.arange(0, min(rows, 4096)), np.arange(max(0, rows - 4096), rows),
                         rng.integers(0, rows, 8192)])
ref = np_net(q[sample], layers, ln, w_out, b_out)
qd = torch.from_numpy(q).cuda()
od = torch.empty((rows, H), dtype=torch.float32, device='cuda')
torch.cuda.synchronize()
for v in variants:
    tile, _, stag = v.partition(':')
    os.environ["RBL_MLP_TILE"] = tile
    os.environ["RBL_MLP_STAGGER"] = stag or "0"
    e.set_net_mlp(layers, ln, w_out, b_out)
    od.zero_()
    torch.cuda.synchronize()
    capi._check(e.L.rbl_net_forward_dev(e.h, qd.data_ptr(), rows, od.data_ptr()))
    e.sync()
    y = od.cpu().numpy()
    err = np.abs(y[sample] - ref).max()
    for _ in range(3):
        capi._check(e.L.rbl_net_forward_dev(e.h, qd.data_ptr(), rows, od.data_ptr()))
    e.sync()
    batches = []
    for _b in range(8):
        t0 = time.perf_counter()
        for _ in range(reps):
            capi._check(e.L.rbl_net_forward_dev(e.h, qd.data_ptr(), rows, od.data_ptr()))
        e.sync()
        batches.append((time.perf_counter() - t0) / reps * 1e6)
    us = min(batches)
    print("   batches (us/launch):", " ".join(f"{b:.0f}" for b in batches))
    flop = rows * 2 * (Q * 256 + 256 * 256 + 256 * H)
    print(f"variant {v:>4}: rows {rows}  max|err| {err:.3e}  {us:9.1f} us/launch  {us * 1e3 / rows:.4f} ns/row  "
          f"{flop / us * 1e-6:7.1f} TFLOP/s algorithmic  (cycles per 64-row group and CU at 2.4 GHz: {us * 2400 / (rows / 64 / 256):.0f})",
          flush=True)
