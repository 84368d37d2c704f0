#!/bin/bash
# PMC instruction / stall counters of the CFR step kernel (wave kernel vs row kernel), 4096 root lanes, one stream
R=$(pwd); O=$R/gpurun_out/pmc_cfr; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for w in 1 0; do
  RBL_CFR_WAVE=$w RBL_PARTS=1 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/w$w -o p -- python3 $R/scripts/probe_cfr_only.py 12 > $O/w$w.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for w in (1, 0):
    f = glob.glob(f'gpurun_out/pmc_cfr/w{w}/**/*counter_collection.csv', recursive=True)
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        if 'cfr_' in r['Kernel_Name'] and 'step_kernel' not in r['Kernel_Name']:
            d[r['Kernel_Name'].split('(')[0][-40:]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in d.items():
        print(f"WAVE={w}", k, {c: round(sum(x) / len(x)) for c, x in v.items()}, "launches", len(next(iter(v.values()))))
PY
