"""Divergence of GPU (MFMA net) vs oracle + torch-CPU Net2 on one 1dx6f root subgame, by net output scale."""
import sys
import numpy as np
sys.path.insert(0, '.')
from tests.test_p3_real_net import _net, _torch_fn
from oracle import orc
from rebel_amd import capi
from rebel_amd.models import mlp_weights_from_state_dict
port = orc.Oracle("port")
d, f = 1, 6
for scale in (1.0, 10.0, 30.0):
    net = _net(d, f, scale)
    kw = dict(num_iters=256, max_depth=2, linear_update=True, use_cfr=True)
    e = capi.Engine(d, f, capi.make_params(**kw), max_lanes=1)
    e.set_net_mlp(*mlp_weights_from_state_dict(net.state_dict()))
    e.reset([-1], [0], np.full((1, 2, e.H), 1.0 / e.H))
    o = port.solver(d, f, orc.make_params(**kw), net=orc.NET_CALLBACK, net_fn=_torch_fn(net))
    for it in range(256):
        e.step(it % 2); o.step(it % 2)
        if it + 1 in (8, 16, 32, 64, 128, 256):
            dl = np.abs(e.get(0, capi.GET_LAST) - o.get(orc.GET_LAST)).max()
            da = np.abs(e.get(0, capi.GET_AVERAGE) - o.get(orc.GET_AVERAGE)).max()
            dv = max(np.abs(e.hand_values(0, pl) - o.hand_values(pl)).max() for pl in (0, 1))
            print(f"scale {scale:5.1f} it {it+1:4d}  last {dl:.2e}  avg {da:.2e}  values {dv:.2e}", flush=True)
