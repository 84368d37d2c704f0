#!/bin/bash
# round-5 GPU session A: n_layers = 3 on the resident kernel -- parity, then rows/s of the build variants vs the tile-3 kernel
mkdir -p gpurun_out; exec > gpurun_out/r5_a.log 2>&1
export RBL_NET_DBG=1
timeout 300 python -m pytest tests/test_net_parity.py -x -q -m gpu 2>&1 | tail -3
RBL_NET_DBG=0 timeout 120 python scripts/probe_net_shape.py 1 6 270336 60 2 > /dev/null
echo "--- default lib, n_layers 3"
timeout 120 python scripts/probe_net_shape.py 1 6 270336 40 3
for v in t4e2 t4e2pf1 t6e3; do
  echo "--- variant $v, n_layers 3"
  REBEL_HIP_LIB=scratch_alt/librebel_hip_$v.so timeout 120 python scripts/probe_net_shape.py 1 6 270336 40 3
done
echo "--- 2d3f n_layers 3 (K0C = 2)"
timeout 120 python scripts/probe_net_shape.py 2 3 270336 40 3
echo "--- default lib, n_layers 2"
timeout 120 python scripts/probe_net_shape.py 1 6 270336 40 2
timeout 120 python scripts/probe_net_shape.py 1 4 270336 40 2
timeout 120 python scripts/probe_net_shape.py 2 3 270336 40 2
timeout 120 python scripts/probe_net_shape.py 2 6 229376 40 2
echo "--- bench (headline only)"
timeout 300 python bench.py --no-extra-legs --no-cpu-baseline --steps 6 --warmup 2 | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k: d[k] for k in ('value', 'ms_per_step')}, {k: d['roofline'][k] for k in ('frac', 'avg_launch_us', 'ns_per_row')}, {k: d['roofline_cfr'][k] for k in ('frac', 'avg_launch_us')})"
