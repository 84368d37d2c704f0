#!/bin/bash
# P3 with the SAME generator seeds on both sides: 32 lanes with seeds k*100000 + i against the reference's 32 threads
O=gpurun_out/r04k; mkdir -p $O
timeout 3000 python tests/p3_policy_iteration.py --sets 36 --lanes 32 --work /tmp/p3pi > $O/p3_policy_iteration_paired.json 2> $O/p3.err; echo "p3 rc=$?" | tee -a $O/rc.txt
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04k/p3_policy_iteration_paired.json'))
print(json.dumps({k:v for k,v in d["summary"].items() if k!="reference_sets_from_cache"})); print(json.dumps(d.get("statistics")))
PY
