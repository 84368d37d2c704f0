#!/bin/bash
# fifth GPU pass of round 4: P3 "what the data is for" with 12 data sets per generator (reference sets from tests/_p3_cache)
O=gpurun_out/r04e; mkdir -p $O
timeout 2400 python tests/p3_policy_iteration.py --sets 12 --work /tmp/p3pi > $O/p3_policy_iteration.json 2> $O/p3.err; echo "p3 rc=$?" | tee -a $O/rc.txt
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04e/p3_policy_iteration.json'))
print(json.dumps(d.get("summary"),indent=0)); print(json.dumps(d.get("statistics"),indent=0))
PY
