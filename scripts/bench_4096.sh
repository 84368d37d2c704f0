for P in 1 2; do RBL_PARTS=$P python3 bench.py --lanes 4096 --steps 10 --warmup 4 --no-cpu-baseline --no-extra-legs 2>/dev/null | python3 -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1])
it=b['ms_per_step']*1e3/1024
print('parts', b['streams'], 'value %.2fM' % (b['value']/1e6), 'us/iter %.1f' % it, 'net us %.1f rows %.0f' % (b['roofline']['avg_launch_us'], b['roofline']['rows_per_launch']), 'cfr us %.1f' % b['roofline_cfr']['avg_launch_us'])"; done
