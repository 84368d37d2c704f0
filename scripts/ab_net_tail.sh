#!/bin/bash
# Same-box A/B of the persistent forward's 32-row tail items (RBL_NET_TAIL=0/1; automatic = on for small interleaved parts).
# usage (GPU box): bash scripts/ab_net_tail.sh > gpurun_out/ab_net_tail.txt 2>&1
run() {
  env RBL_NET_TAIL=$T python3 bench.py --no-cpu-baseline --no-extra-legs "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('tail=$T', '$*', '-> value=%.2fM it/s ms/step=%.1f net_us=%.1f cfr_us=%.1f' % (d['value']/1e6, d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline_cfr']['avg_launch_us']))"
}
for rep in 1 2; do
  for T in 0 1; do
    run --dice 1 --faces 4 --iters 1024 --lanes 4096 --steps 10 --warmup 4
    run --dice 1 --faces 5 --iters 1024 --lanes 4096 --steps 10 --warmup 4
    run --dice 1 --faces 6 --iters 1024 --lanes 4096 --steps 10 --warmup 4
    run --dice 2 --faces 3 --iters 1024 --lanes 4096 --steps 6 --warmup 3
    run --dice 1 --faces 6 --iters 1024 --lanes 16384 --steps 6 --warmup 3
  done
done
