#!/bin/bash
# round-5 evidence: GPU suite, rocprofv3 kernel trace + PMC passes on the driver's command shape, then the driver's bench line
mkdir -p gpurun_out; exec > gpurun_out/session.log 2>&1
(timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05_gputest.log 2>&1; echo rc=$? >> gpurun_out/r05_gputest.log); tail -3 gpurun_out/r05_gputest.log
timeout 900 bash scripts/collect_profiles.sh r05
cp gpurun_out/prof_r05/r05_kernel_stats.csv gpurun_out/prof_r05/r05_kernel_stats_timed_epochs.csv gpurun_out/prof_r05/r05_pmc_traffic.json profiles/ 2>/dev/null
cp gpurun_out/prof_r05/bench_under_rocprof.json profiles/r05_bench_under_rocprof.json 2>/dev/null
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err; echo bench rc=$?
tail -c 600 gpurun_out/r05_bench.json
