#!/bin/bash
# round-5 evidence refresh: GPU suite + the driver's bench line on the final code
mkdir -p gpurun_out; exec > gpurun_out/session.log 2>&1
(timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05_gputest.log 2>&1; echo rc=$? >> gpurun_out/r05_gputest.log); tail -3 gpurun_out/r05_gputest.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err; echo bench rc=$?
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
