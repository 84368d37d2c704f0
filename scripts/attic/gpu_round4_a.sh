#!/bin/bash
# first GPU pass of round 4: suite, the driver's bench command, event-method A/B, power/clock trace
O=gpurun_out/r04a; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/rc.txt
tail -5 $O/pytest.log
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/rc.txt
RBL_TIMING_EXT=0 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs > $O/bench_recorded_events.json 2> $O/bench_recorded_events.err
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs > $O/bench_ext_events.json 2> $O/bench_ext_events.err
timeout 600 python scripts/power_trace.py 6 > $O/power_trace.txt 2> $O/power_trace.err; echo "power rc=$?" | tee -a $O/rc.txt
tail -3 $O/power_trace.err
python bench.py --gpus 2 > $O/gpus2.out 2> $O/gpus2.err; echo "gpus2 rc=$? (expected non-zero)" | tee -a $O/rc.txt
