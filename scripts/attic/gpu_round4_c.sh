#!/bin/bash
# third GPU pass of round 4: suite (half_inference modes), energy attribution, bench with the half_inference leg
O=gpurun_out/r04c; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/rc.txt
tail -5 $O/pytest.log; grep "half_inference " $O/pytest.log | head
timeout 900 python -m pytest tests/test_net_parity.py -m gpu -q -s -k half_inference_modes > $O/half_modes.log 2>&1; grep "half_inference " $O/half_modes.log
bash scripts/power_attribution.sh 5 > $O/power_attribution.log 2>&1; echo "power_attribution rc=$?" | tee -a $O/rc.txt
timeout 900 python bench.py --gpus 1 --steps 10 --warmup 5 --no-cpu-baseline --no-configs > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/rc.txt
