#!/bin/bash
# fourth GPU pass of round 4: suite (forest solve, regret reports, half modes, 16 384-lane P3 sample), bench for the VALU diet
O=gpurun_out/r04d; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/rc.txt
tail -15 $O/pytest.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-configs > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/rc.txt
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04d/bench.json'))
print("value", d["value"], "net us", d["roofline"]["avg_launch_us"], "cfr us", d["roofline_cfr"]["avg_launch_us"], "power", d.get("power"))
print("half", d["half_inference"]["value"], d["half_inference"]["net"]["avg_launch_us"], "4096", d["lanes_4096"]["value"], "2str", d["two_streams"]["value"])
PY
