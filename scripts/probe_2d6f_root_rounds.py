"""Developer aid (round 6): CFR step time of n ROOT lanes of 2 dice x 6 faces (one 1 024-thread workgroup with 147 KB of LDS each = one
per CU), n around the multiples of the 256 CUs: the round quantisation of the size-sorted launches.  Kernel time = HIP events bound to
the dispatch; wall = per step incl. the synthetic net launch.  usage: probe_2d6f_root_rounds.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, ".")
from rebel_amd import capi
steps = 100
for parts in (1, 2):
    os.environ["RBL_PARTS"] = str(parts)
    for n in (128, 256, 257, 512, 768, 1024, 1070):
        e = capi.Engine(2, 6, capi.make_params(num_iters=1024, max_depth=2, linear_update=True, use_cfr=True), max_lanes=n)
        e.set_net_synthetic()
        rng = np.random.default_rng(0)
        e.reset(np.full(n, -1, np.int32), np.zeros(n, np.int32), rng.dirichlet(np.ones(e.H), size=(n, 2)))
        e.multistep(10); e.sync()
        e.stats(reset=True); e.timing(1)
        t0 = time.perf_counter(); e.multistep(steps); e.sync(); dt = (time.perf_counter() - t0) / steps * 1e6
        st = e.stats(reset=True); e.timing(0)
        print(f"parts {parts} roots {n:5d}: wall {dt:7.1f} us/step; cfr launches {st['cfr_launches']} avg {st['cfr_ms']/max(1,st['cfr_launches'])*1e3:7.1f} us  streams {st['n_streams']}", flush=True)
        e.close()
