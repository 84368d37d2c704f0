#!/bin/bash
# One session on ONE box: the round's evidence set on the final code (run through gpurun from the repo root).
#   bash scripts/collect_round_evidence.sh r06
TAG=${1:-r06}
R=$(pwd); O=$R/gpurun_out/evidence_$TAG; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "Warning\|warnings.warn" | tail -12 > $O/${TAG}_gputest.log
# the driver's own command shape, every leg
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
# rocprofv3 --kernel-trace --stats + the two PMC traffic passes of the same command (headline leg only)
timeout 1200 bash scripts/collect_profiles.sh $TAG > $O/collect_$TAG.log 2>&1
cp gpurun_out/prof_$TAG/${TAG}_* gpurun_out/prof_$TAG/bench_under_rocprof.json $O/ 2>/dev/null
mv $O/bench_under_rocprof.json $O/${TAG}_bench_under_rocprof.json 2>/dev/null
# ... and of BASELINE config 4 (2 dice x 6 faces @2048 x 2 048 lanes), warmed until its root / small-tree mix has settled
BENCH_ARGS="--dice 2 --faces 6 --iters 2048 --lanes 2048" STEPS=8 WARMUP=16 timeout 1200 bash scripts/collect_profiles.sh ${TAG}_2d6f > $O/collect_${TAG}_2d6f.log 2>&1
cp gpurun_out/prof_${TAG}_2d6f/${TAG}_2d6f_* $O/ 2>/dev/null
cp gpurun_out/prof_${TAG}_2d6f/bench_under_rocprof.json $O/${TAG}_2d6f_bench_under_rocprof.json 2>/dev/null
# instruction / busy counters of the two hot kernels
timeout 900 bash scripts/pmc_hot_kernels.sh $TAG > $O/pmc_hot_$TAG.log 2>&1
cp gpurun_out/pmc_$TAG/*summary* $O/ 2>/dev/null
rm -rf gpurun_out/prof_$TAG gpurun_out/prof_${TAG}_2d6f
ls -la $O
