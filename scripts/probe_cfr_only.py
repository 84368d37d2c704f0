"""Runs only CFR steps (synthetic elementwise net) on root lanes (argv[2], default 4096) -- for rocprofv3 PMC runs of the CFR
step kernel.  usage: probe_cfr_only.py [steps] [lanes]"""
import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rebel_amd import capi
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
e = capi.Engine(int(os.environ.get('RBL_PROBE_DICE', 1)), int(os.environ.get('RBL_PROBE_FACES', 6)), capi.make_params(num_iters=1024, max_depth=2, linear_update=True, use_cfr=True), max_lanes=B)
e.set_net_synthetic()
e.reset([-1]*B, [0]*B, np.full((B, 2, e.H), 1.0/e.H))
e.multistep(int(sys.argv[1]) if len(sys.argv) > 1 else 6); e.sync()
