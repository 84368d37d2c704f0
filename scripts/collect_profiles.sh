#!/bin/bash
# Round profile collection on the GPU box (run through gpurun from the repo root):
#   1. rocprofv3 --kernel-trace --stats over the default bench command (short window)  -> per-kernel durations
#   2. rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (TCC slots: they do not fit one pass;
#      MI355X_MICROARCH.md "rocprofv3 PMC slots")                                      -> HBM-side bytes per launch
# Summaries are written by scripts/pmc_summary.py; copy what should be judged from gpurun_out/ into profiles/.
TAG=${1:-r02}
R=$(pwd)
O=$R/gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# the driver's command shape (BENCH_rNN.json: bench.py --gpus 1 --steps 20 --warmup 5), minus the legs that are not the
# headline (CPU baseline, comparison legs): the timed epochs see the same lanes, seeds and subgame mix as the driver's
CMD="python3 $R/bench.py --gpus 1 --steps ${STEPS:-20} --warmup ${WARMUP:-5} --no-cpu-baseline --no-extra-legs ${BENCH_ARGS:-}"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- $CMD > $O/kt.log 2>&1
grep '^{' $O/kt.log | tail -1 > $O/bench_under_rocprof.json
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o fetch -- $CMD > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o write -- $CMD > $O/write.log 2>&1
cd $R
python3 scripts/pmc_summary.py $O $TAG
# keep the summaries, drop the raw traces (tens of MB per pass: gpurun merges at most 64 MiB back)
cp $O/kt/kt_kernel_stats.csv $O/${TAG}_rocprofv3_kernel_stats.csv 2>/dev/null
if [ -z "$KEEP_RAW" ]; then rm -rf $O/kt $O/fetch $O/write; fi
ls -la $O | head -30
