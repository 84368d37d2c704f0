#!/bin/bash
# The other BASELINE.json configurations on one GPU (not the headline): one summary line each.
run() {
  python3 bench.py --no-cpu-baseline --no-extra-legs "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$*', '-> value=%.2fM it/s games/s=%.0f ms/step=%.1f net_us=%.1f (frac %.3f) cfr_us=%.1f (frac %.3f)' % (d['value']/1e6, d['games_per_s'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['roofline_cfr']['avg_launch_us'], d['roofline_cfr']['frac']))"
}
run --dice 1 --faces 4 --iters 1024 --lanes 4096 --steps 6 --warmup 3
run --dice 1 --faces 4 --iters 1024 --lanes 16384 --steps 6 --warmup 3
run --dice 2 --faces 3 --iters 1024 --lanes 16384 --steps 4 --warmup 3
run --dice 2 --faces 6 --iters 2048 --lanes 1024 --steps 2 --warmup 1
run --dice 2 --faces 6 --iters 2048 --lanes 2048 --steps 2 --warmup 1
run --dice 2 --faces 6 --iters 2048 --lanes 4096 --steps 2 --warmup 1
run --dice 1 --faces 6 --iters 1024 --lanes 4096 --steps 10 --warmup 4
