#!/bin/bash
# Stress variants of the first GPU test, trying to reproduce the round-1 driver-side SIGABRT in rbl_engine_create.
O=gpurun_out/r02b; mkdir -p $O
T="tests/test_cfr_parity.py::test_solver_bit_exact_vs_oracle_and_golden"
run() { # name, env...
  local name=$1; shift
  local fails=0
  for i in $(seq 1 8); do
    env "$@" timeout 120 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider -k "1d4f_dcfr_syn_64 or 1d4f_depth3" > $O/$name.$i.log 2>&1
    rc=$?
    if [ $rc -ne 0 ]; then fails=$((fails+1)); echo "$name run $i rc=$rc"; tail -5 $O/$name.$i.log; else rm -f $O/$name.$i.log; fi
  done
  echo "== $name: $fails/8 failed"
}
run plain A=1
run mcheck MALLOC_CHECK_=3 MALLOC_PERTURB_=165
# concurrent SMI sampler like the driver's GPU-busy sampler
( while true; do rocm-smi --showuse --showmemuse --json > /dev/null 2>&1; amd-smi metric --json > /dev/null 2>&1; sleep 0.2; done ) &
SMI=$!
run smi A=1
kill $SMI
# a sitecustomize hook that dumps /proc/self/maps at exit (what the driver's native-.so recorder plausibly does)
mkdir -p /tmp/_pyhook && cat > /tmp/_pyhook/sitecustomize.py <<'PY'
import atexit, os
def _dump():
    try:
        open('/tmp/_pyhook/maps.%d' % os.getpid(), 'w').write(open('/proc/self/maps').read())
    except Exception:
        pass
atexit.register(_dump)
PY
run hook PYTHONPATH=/tmp/_pyhook
run serialize AMD_SERIALIZE_KERNEL=3 HIP_LAUNCH_BLOCKING=1
env | grep -i "hip\|hsa\|rocm\|amd\|gpu" > $O/env.txt
rocminfo | grep -i "gfx\|compute unit\|Marketing" | head -20 > $O/rocminfo.txt
