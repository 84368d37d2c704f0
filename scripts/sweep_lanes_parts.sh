#!/bin/bash
# lanes x parts sweep of the end-to-end bench (no CPU baseline), one JSON summary line each
for lanes in 4096 8192 16384; do
  for parts in 1 2; do
    RBL_PARTS=$parts python3 bench.py --no-cpu-baseline --lanes $lanes --steps 3 --warmup 2 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('lanes=$lanes parts=$parts value=%.2fM ms=%.1f net_us=%.1f rows=%.0f net_frac=%.4f cfr_us=%.1f cfr_frac=%.4f' % (d['value']/1e6, d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['rows_per_launch'], d['roofline']['frac'], d['roofline_cfr']['avg_launch_us'], d['roofline_cfr']['frac']))"
  done
done
