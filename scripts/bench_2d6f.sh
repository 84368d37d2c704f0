for L in 2048 4096; do python3 bench.py --dice 2 --faces 6 --iters 2048 --lanes $L --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs 2>/dev/null | python3 -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('lanes', b['config']['lanes_per_gpu'], 'value', b['value'], 'ms/step', b['ms_per_step'], 'net us', b['roofline']['avg_launch_us'], 'cfr us', b['roofline_cfr']['avg_launch_us'], 'streams', b['streams'])"; done
