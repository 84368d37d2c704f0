#!/bin/bash
# sixth GPU pass of round 4: suite + smoke on the final kernels, the driver's bench command, the evaluation tool with
# --root_only and the regret reports at 2 dice x 6 faces
O=gpurun_out/r04g; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/rc.txt
tail -3 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/rc.txt; tail -1 $O/smoke.log
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/rc.txt
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04g/bench.json'))
print("value", d["value"], "net us", d["roofline"]["avg_launch_us"], d["roofline"]["frac"], "cfr us", d["roofline_cfr"]["avg_launch_us"], d["roofline_cfr"]["frac"], "power", d.get("power"))
print("half", d["half_inference"]["value"], "4096", d["lanes_4096"]["value"], "2str", d["two_streams"]["value"], "cpu", d["cpu_baseline"]["value"])
for c in d["configs"]: print(c["baseline_config"], c["value"], c["net"]["frac"], c["cfr"]["frac"], c.get("cpu_reference",{}).get("value"))
PY
timeout 900 python scripts/recursive_eval.py --num_dice 2 --num_faces 6 --subgame_iters 32 --mdp_depth 2 --num_repeats 2 --net zero --cfr --stream --root_only --print_regret_summary --max_lanes 8192 > $O/recursive_eval_2d6f_root_only.txt 2> $O/recursive_eval_2d6f_root_only.err; echo "tool rc=$?" | tee -a $O/rc.txt
tail -8 $O/recursive_eval_2d6f_root_only.txt | cut -c1-300
