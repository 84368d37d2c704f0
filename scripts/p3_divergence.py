"""P3: fused end-to-end run with the REAL value net -- GPU engine (f16x2-split MFMA forward) vs the
CPU oracle driven by the same Net2 evaluated by torch on CPU.  The two net forwards differ by ~1e-7; CFR amplifies that.
Prints max |delta| of root value means and of sigma_last at power-of-two iteration counts (1d x 6f root subgame)."""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from oracle import orc
from rebel_amd import capi
from rebel_amd.models import Net2, mlp_weights_from_state_dict

d, f = 1, 6
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0   # output-layer scale (1 = default init, outputs ~3e-3)
torch.manual_seed(0)
net = Net2(num_faces=f, num_dice=d, n_hidden=256, use_layer_norm=True, n_layers=2).eval()
with torch.no_grad():
    net.output.weight *= scale
    net.output.bias *= scale

def fn(q):
    with torch.no_grad():
        return net(torch.from_numpy(q)).numpy()

kw = dict(num_iters=1024, max_depth=2, linear_update=True, use_cfr=True)
port = orc.Oracle("port")
o = port.solver(d, f, orc.make_params(**kw), net=orc.NET_CALLBACK, net_fn=fn)
e = capi.Engine(d, f, capi.make_params(**kw))
e.set_net_mlp(*mlp_weights_from_state_dict(net.state_dict()))
H = e.H
e.reset([-1], [0], np.full((1, 2, H), 1.0 / H))
print(f"output scale x{scale:g}: iters | max|d root_mean| | max|d sigma_last| | max|d regrets|/max|regrets|")
marks = {1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024}
for it in range(1024):
    o.step(it % 2); e.step(it % 2)
    if it + 1 in marks and it >= 1:
        dm = max(np.abs(e.hand_values(0, p) - o.hand_values(p)).max() for p in (0, 1))
        ds = np.abs(e.get(0, capi.GET_LAST) - o.get(orc.GET_LAST)).max()
        R = o.get(orc.GET_REGRETS)
        dr = np.abs(e.get(0, capi.GET_REGRETS) - R).max() / max(1e-300, np.abs(R).max())
        print(f"{it + 1:6d} | {dm:.2e} | {ds:.2e} | {dr:.2e}")
