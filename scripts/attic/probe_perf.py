import time, numpy as np, sys
sys.path.insert(0, '.')
from rebel_amd import capi
g = np.load('tests/golden/net2_1d6f.npz')
layers = [(g["body__0__weight"], g["body__0__bias"]), (g["body__4__weight"], g["body__4__bias"])]
ln = [(g["body__1__weight"], g["body__1__bias"]), (g["body__5__weight"], g["body__5__bias"])]
for B in (256, 4096):
    e = capi.Engine(1, 6, capi.make_params(num_iters=1024, max_depth=2, linear_update=True, use_cfr=True), max_lanes=B)
    e.set_net_mlp(layers, ln, g["output__weight"], g["output__bias"])
    H = e.H
    e.reset([-1]*B, [0]*B, np.full((B,2,H), 1.0/H))
    e.multistep(32); e.sync()
    e.timing(True); e.stats(reset=True)
    t=time.time(); e.multistep(128); e.sync(); dt=time.time()-t
    s = e.stats(reset=True)
    print(B, 'wall ms/iter', dt/128*1e3, 'it/s', B*128/dt, s)
    print('  cfr us/launch', s['cfr_ms']/s['cfr_launches']*1e3, 'GB/s', s['cfr_bytes']/s['cfr_ms']/1e6,
          ' net us/launch', s['net_ms']/s['net_launches']*1e3, 'TF/s', s['net_flops']/s['net_ms']/1e9)
    e.timing(False)
    t=time.time(); e.multistep(256); e.sync(); dt=time.time()-t
    print('  untimed: ms/iter', dt/256*1e3, 'it/s', B*256/dt)
