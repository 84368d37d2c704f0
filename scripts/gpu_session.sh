#!/bin/bash
mkdir -p gpurun_out; exec > gpurun_out/session.log 2>&1
for extra in 0 256 512 1296 1536 1792 2816; do
  echo "--- RBL_WAVE_LDS_EXTRA=$extra (request $((8704 + extra)) B)"
  RBL_WAVE_LDS_EXTRA=$extra timeout 200 python bench.py --no-extra-legs --no-cpu-baseline --no-configs --steps 2 --warmup 1 | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('cfr', {k: round(d['roofline_cfr'][k], 4) for k in ('frac', 'avg_launch_us')})"
done
