# launch-boundary knobs of the HIP runtime against the headline (same box, back to back).
# CAUTION: when this was run (end of round 4) the first line came back (gap 1.5 us: nothing to recover) and one of the knobs below
# hung the process until gpurun's limit -- every run is wrapped in `timeout` now; there is no reason to run it again.
run() { timeout 150 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); it=d['ms_per_step']/1024*1e3; print(round(d['value']/1e6,2), 'iter us', round(it,1), 'kernels', round(d['roofline']['avg_launch_us']+d['roofline_cfr']['avg_launch_us'],1), 'gap', round(it-d['roofline']['avg_launch_us']-d['roofline_cfr']['avg_launch_us'],1))"; }
echo "default: $(run)"
echo "ROC_SYSTEM_SCOPE_SIGNAL=0: $(ROC_SYSTEM_SCOPE_SIGNAL=0 run)"
echo "HIP_FORCE_DEV_KERNARG=1: $(HIP_FORCE_DEV_KERNARG=1 run)"
echo "HIP_FORCE_DEV_KERNARG=0: $(HIP_FORCE_DEV_KERNARG=0 run)"
echo "AMD_OPT_FLUSH=0: $(AMD_OPT_FLUSH=0 run)"
echo "ROC_ACTIVE_WAIT_TIMEOUT=100: $(ROC_ACTIVE_WAIT_TIMEOUT=100 run)"
echo "default again: $(run)"
