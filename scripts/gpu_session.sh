#!/bin/bash
mkdir -p gpurun_out; exec > gpurun_out/session.log 2>&1
timeout 600 python -m pytest tests/test_net_parity.py -x -q -m gpu 2>&1 | tail -3
export RBL_NET_DBG=1
for v in base new base new; do
  echo "--- $v"
  if [ $v = base ]; then export REBEL_HIP_LIB=scratch_alt/librebel_hip_base.so; else unset REBEL_HIP_LIB; fi
  timeout 120 python scripts/probe_net_shape.py 1 6 270336 40 2 | grep -v amdgpu
done
unset REBEL_HIP_LIB
timeout 120 python scripts/probe_net_shape.py 1 6 270336 40 3 | grep -v amdgpu
timeout 120 python scripts/probe_net_shape.py 1 4 270336 40 2 | grep -v amdgpu
export RBL_NET_DBG=0
for rep in 1 2; do
for v in base new; do
  echo "--- bench $v (rep $rep)"
  if [ $v = base ]; then export REBEL_HIP_LIB=scratch_alt/librebel_hip_base.so; else unset REBEL_HIP_LIB; fi
  timeout 200 python bench.py --no-extra-legs --no-cpu-baseline --no-configs --steps 4 --warmup 2 | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k: d[k] for k in ('value', 'ms_per_step')}, 'net', {k: round(d['roofline'][k], 4) for k in ('frac', 'avg_launch_us', 'ns_per_row')}, 'cfr', {k: round(d['roofline_cfr'][k], 4) for k in ('frac', 'avg_launch_us')}, d['power'])"
done
done
