#!/bin/bash
mkdir -p gpurun_out; exec > gpurun_out/session.log 2>&1
timeout 900 python -m pytest tests/test_cfr_parity.py tests/test_selfplay_parity.py tests/test_golden.py tests/test_eval_parity.py -x -q -m gpu -k "2d6f or 2d or flat or golden or six" 2>&1 | tail -4
for rep in 1 2; do
for v in base new; do
  echo "--- bench 2d6f $v (rep $rep)"
  if [ $v = base ]; then export REBEL_HIP_LIB=scratch_alt/librebel_hip_base.so; else unset REBEL_HIP_LIB; fi
  timeout 300 python bench.py --dice 2 --faces 6 --lanes 2048 --iters 2048 --no-extra-legs --no-cpu-baseline --no-configs --steps 4 --warmup 2 | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k: d[k] for k in ('value', 'ms_per_step')}, 'net', {k: round(d['roofline'][k], 4) for k in ('frac', 'avg_launch_us')}, 'cfr', {k: round(d['roofline_cfr'][k], 4) for k in ('frac', 'avg_launch_us')})"
done
done
unset REBEL_HIP_LIB
echo "--- phases new"
timeout 300 python scripts/probe_cfr_phases_2d6f.py 9 2048 | grep -v amdgpu
