#!/bin/bash
mkdir -p gpurun_out; exec > gpurun_out/session.log 2>&1
timeout 900 python -m pytest tests/test_p3_real_net.py tests/test_selfplay_parity.py tests/test_net_parity.py -x -q -m gpu 2>&1 | tail -3
run() {
  echo -n "$1: "; shift
  env "$@" | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value'] / 1e6, 2), 'M', round(d['ms_per_step'], 2), 'ms', 'net', round(d['roofline']['avg_launch_us'], 1), 'cfr', round(d['roofline_cfr']['avg_launch_us'], 1))"
}
B="timeout 300 python bench.py --no-extra-legs --no-cpu-baseline --no-configs"
for rep in 1 2; do
run "1d6f 4096 auto" X=1 $B --lanes 4096 --steps 8 --warmup 2
run "1d6f 4096 grid 0" RBL_NET_GRID=0 $B --lanes 4096 --steps 8 --warmup 2
run "1d4f 4096 auto" X=1 $B --faces 4 --lanes 4096 --steps 8 --warmup 2
run "1d4f 4096 grid 0" RBL_NET_GRID=0 $B --faces 4 --lanes 4096 --steps 8 --warmup 2
done
run "1d5f 4096 auto" X=1 $B --faces 5 --lanes 4096 --steps 8 --warmup 2
run "1d5f 4096 grid 0" RBL_NET_GRID=0 $B --faces 5 --lanes 4096 --steps 8 --warmup 2
run "2d3f 4096 auto" X=1 $B --dice 2 --faces 3 --lanes 4096 --steps 8 --warmup 2
run "2d3f 4096 grid 0" RBL_NET_GRID=0 $B --dice 2 --faces 3 --lanes 4096 --steps 8 --warmup 2
run "1d6f 2048 auto" X=1 $B --lanes 2048 --steps 8 --warmup 2
run "1d6f 2048 grid 0" RBL_NET_GRID=0 $B --lanes 2048 --steps 8 --warmup 2
