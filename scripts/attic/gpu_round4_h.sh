#!/bin/bash
O=gpurun_out/r04h; mkdir -p $O
timeout 900 python -m pytest tests/test_net_parity.py tests/test_golden.py tests/test_p3_real_net.py -m gpu -x -q > $O/pytest_net.log 2>&1; echo "pytest net rc=$?" | tee -a $O/rc.txt; tail -3 $O/pytest_net.log
RBL_MLP_STAGGER=5 python scripts/probe_net_phases.py 589824 2>&1 | tail -12
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-configs > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/rc.txt
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04h/bench.json'))
print("value", d["value"], "net us", d["roofline"]["avg_launch_us"], d["roofline"]["frac"], "cfr us", d["roofline_cfr"]["avg_launch_us"], d["roofline_cfr"]["frac"], "power", d.get("power"))
print("half", d["half_inference"]["value"], d["half_inference"]["net"]["avg_launch_us"], "4096", d["lanes_4096"]["value"], "2str", d["two_streams"]["value"])
PY
