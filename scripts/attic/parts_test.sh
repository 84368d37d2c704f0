for cfg in "1 4 1024 4096" "1 6 1024 4096" "2 6 2048 2048" "2 3 1024 4096"; do
  set -- $cfg
  for p in 1 2; do
    v=$(RBL_PARTS=$p python bench.py --dice $1 --faces $2 --iters $3 --lanes $4 --steps 8 --warmup 3 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), round(d['roofline']['avg_launch_us'],1), round(d['roofline_cfr']['avg_launch_us'],1), d['streams'])")
    echo "cfg $cfg parts $p -> $v"
  done
done
