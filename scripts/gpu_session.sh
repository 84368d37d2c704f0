#!/bin/bash
mkdir -p gpurun_out; exec > gpurun_out/session.log 2>&1
STEPS=4 WARMUP=2 BENCH_ARGS="--dice 2 --faces 6 --lanes 2048 --iters 2048 --no-configs" timeout 1200 bash scripts/collect_profiles.sh r05_2d6f
cat gpurun_out/prof_r05_2d6f/r05_2d6f_kernel_stats_timed_epochs.csv
cat gpurun_out/prof_r05_2d6f/r05_2d6f_kernel_stats.csv
cat gpurun_out/prof_r05_2d6f/bench_under_rocprof.json | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d[k] for k in ('value', 'ms_per_step', 'streams')}, 'net', {k: round(d['roofline'][k], 4) for k in ('frac', 'avg_launch_us')}, 'cfr', {k: round(d['roofline_cfr'][k], 4) for k in ('frac', 'avg_launch_us')})"
