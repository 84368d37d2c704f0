"""Developer aid: the tree-size mix of a 2 dice x 6 faces self-play batch, epoch by epoch (what the size-sorted launch segments of
cfr_flat_kernel see).  usage: probe_2d6f_mix.py [epochs] [lanes] [iters]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from rebel_amd import capi  # noqa: E402
from rebel_amd.models import Net2, mlp_weights_from_state_dict  # noqa: E402

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 6
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
e = capi.Engine(2, 6, capi.make_params(num_iters=iters, max_depth=2, linear_update=True, use_cfr=True), max_lanes=B)
torch.manual_seed(0)
net = Net2(num_faces=6, num_dice=2, n_hidden=256, use_layer_norm=True, n_layers=2).eval()
e.set_net_mlp(*mlp_weights_from_state_dict(net.state_dict()))
sp = capi.SelfPlay(e, list(range(B)))
edges = [0, 30, 80, 160, 200, 260, 324, 400]
for ep in range(epochs):
    sizes = np.array([e.tree_size(i) for i in range(B)])
    hist, _ = np.histogram(sizes, bins=edges)
    print(f"epoch {ep}: N<=30 {hist[0]}  31-80 {hist[1]}  81-160 {hist[2]}  161-200 {hist[3]}  201-260 {hist[4]}  261-324 {hist[5]}  "
          f"325 {hist[6]}   (sorted halves of each 1024-lane part: head sizes "
          f"{[int(np.sort(sizes[p * B // 2:(p + 1) * B // 2])[::-1][k]) for p in range(2) for k in (0, B // 4)]})", flush=True)
    sp.advance()
st = e.stats()
print({k: st[k] for k in ("cfr_kernel", "net_kernel", "n_streams")})
