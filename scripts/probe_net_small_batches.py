"""The floor of small value-net launches ("or phase stamps under profiles/ that show the floor of a
1.8-groups-per-CU launch").  The resident forward at 1 die x 4 faces' shape (config 1) for row counts around the 30 k rows of a
2 048-lane part: wall time per launch on device buffers (standalone entry point: split-layout conversion kernel + forward) with the
grid capped at 192 workgroups (what the engine gives the net kernel when two small parts interleave) and uncapped, and the
RBL_NET_DBG stamps of the workgroups: cycles of the FIRST group (cold start: weights into registers / LDS, first rows' round trip)
and per group over the whole workgroup.  A third argument "fine" sweeps 1.75 ... 3.5 groups per workgroup in small steps (where
the cost of a partial last round shows).  usage: probe_net_small_batches.py [dice faces [fine]]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RBL_NET_DBG"] = "1"
import torch  # noqa: E402  (device buffers only)

from rebel_amd import capi  # noqa: E402
from rebel_amd.models import Net2, mlp_weights_from_state_dict  # noqa: E402

dice, faces = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1, 4)
torch.manual_seed(0)
net = Net2(num_faces=faces, num_dice=dice, n_hidden=256, use_layer_norm=True, n_layers=2).eval()
for cap in (192, 0):
    os.environ["RBL_NET_GRID"] = str(cap)
    e = capi.Engine(dice, faces, capi.make_params(num_iters=4, use_cfr=True, max_depth=2))
    e.set_net_mlp(*mlp_weights_from_state_dict(net.state_dict()))
    grid = cap or 256
    print(f"== {dice}d x {faces}f, grid cap {grid} workgroups (64-row groups, one persistent workgroup per CU)")
    print("  groups  per-WG  rows     us/launch   first group (cycles)  cycles/group over the workgroup   workgroup total")
    fine = len(sys.argv) > 3 and sys.argv[3] == "fine"
    sweep = (1.75, 2.0, 2.1, 2.25, 2.46, 2.6, 2.75, 2.9, 3.0, 3.1, 3.25, 3.5) if fine else (0.5, 1.0, 1.8, 2.0, 2.46, 3.0, 4.0, 8.0, 36.0)
    for per_wg in sweep:
        groups = max(1, int(round(per_wg * grid)))
        rows = groups * 64
        rng = np.random.default_rng(1)
        q = rng.random((rows, e.Q), dtype=np.float32)
        qd = torch.from_numpy(q).cuda()
        od = torch.empty((rows, e.H), dtype=torch.float32, device="cuda")
        for _ in range(5):
            capi._check(e.L.rbl_net_forward_dev(e.h, qd.data_ptr(), rows, od.data_ptr()))
        e.sync()
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(50):
                capi._check(e.L.rbl_net_forward_dev(e.h, qd.data_ptr(), rows, od.data_ptr()))
            e.sync()
            best = min(best, (time.perf_counter() - t0) / 50 * 1e6)
        st = e.net_debug_stamps()
        n = min(grid, groups, 1024)
        per = np.array([len(range(b, groups, grid)) for b in range(n)])
        first = np.median(st[:n, 8] - st[:n, 0])
        total = np.median(st[:n, 12] - st[:n, 0])
        print(f"  {groups:6d}  {groups / grid:6.2f}  {rows:7d}  {best:9.1f}   {first:12.0f}          {np.median((st[:n, 12] - st[:n, 0]) / per):12.0f}                  {total:10.0f}")
    e.close()
