#!/bin/bash
# final GPU pass of round 4: the driver's sequence (suite, smoke, bench) on the final code + the torchrun form of the bench
O=gpurun_out/r04j; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/rc.txt; tail -2 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/rc.txt; tail -1 $O/smoke.log
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/rc.txt
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04j/bench.json'))
print("value", d["value"], "net us", d["roofline"]["avg_launch_us"], d["roofline"]["frac"], d["roofline"].get("rocprof",{}).get("frac"), "cfr us", d["roofline_cfr"]["avg_launch_us"], d["roofline_cfr"]["frac"], "power", d.get("power"))
print("half", d["half_inference"]["value"], "4096", d["lanes_4096"]["value"], "2str", d["two_streams"]["value"], "cpu", d["cpu_baseline"]["value"])
for c in d["configs"]: print(c["baseline_config"], c["value"], c["net"]["frac"], c["cfr"]["frac"], c.get("cpu_reference",{}).get("value"))
PY
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 > $O/bench_torchrun.json 2> $O/bench_torchrun.err; echo "torchrun rc=$?" | tee -a $O/rc.txt
python -c "
import json; d=json.loads([l for l in open('gpurun_out/r04j/bench_torchrun.json') if l.startswith('{')][-1]); print('torchrun n_gpus', d['n_gpus'], d['value'], d.get('per_gpu'))"
