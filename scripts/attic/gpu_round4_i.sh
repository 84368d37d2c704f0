#!/bin/bash
# seventh GPU pass of round 4: P3 with 36 data sets per generator; profiles of the final kernels; 2d x 6f kernel stats
O=gpurun_out/r04i; mkdir -p $O
timeout 3000 python tests/p3_policy_iteration.py --sets 36 --work /tmp/p3pi > $O/p3_policy_iteration.json 2> $O/p3.err; echo "p3 rc=$?" | tee -a $O/rc.txt
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04i/p3_policy_iteration.json'))
print(json.dumps({k:v for k,v in d["summary"].items() if k!="reference_sets_from_cache"})); print(json.dumps(d.get("statistics")))
PY
bash scripts/collect_profiles.sh r04 > $O/collect_r04.log 2>&1; tail -2 $O/collect_r04.log
STEPS=5 WARMUP=3 BENCH_ARGS="--dice 2 --faces 6 --iters 2048 --lanes 2048" bash scripts/collect_profiles.sh r04_2d6f > $O/collect_r04_2d6f.log 2>&1; tail -2 $O/collect_r04_2d6f.log
