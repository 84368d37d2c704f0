"""Developer aid: run under `rocprofv3 --pmc WRITE_SIZE|FETCH_SIZE --kernel-trace` -- 512 root lanes of 2 dice x 6 faces,
20 CFR steps with the synthetic net; per step the flat kernel should write 3 x E_t x 288 B + L x 396 B per lane
(E_t = 24 / 300 alternating, L = 276) and read E x 288 + 2 x E_t x 288 + ~2 x L x 144 B."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from rebel_amd import capi  # noqa: E402

B = 512
e = capi.Engine(2, 6, capi.make_params(num_iters=1024, max_depth=2, linear_update=True, use_cfr=True), max_lanes=B)
e.set_net_synthetic()
rng = np.random.default_rng(0)
e.reset([-1] * B, [0] * B, rng.dirichlet(np.ones(e.H), size=(B, 2)))
e.multistep(20)
e.sync()
print("expected MB per launch: writes", [round(B * (3 * et * 288 + 276 * 396) / 1e6, 1) for et in (24, 300)],
      "reads", [round(B * (324 * 288 + 2 * et * 288 + 2 * 276 * 144) / 1e6, 1) for et in (24, 300)])
