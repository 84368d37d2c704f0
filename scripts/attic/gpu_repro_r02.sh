#!/bin/bash
# Reproduces the driver's exact GPU test command on a fresh box and, if it aborts, captures a native backtrace.
mkdir -p gpurun_out/r02
python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider > gpurun_out/r02/pytest_gpu.log 2>&1
rc=$?
echo "pytest rc=$rc" | tee gpurun_out/r02/pytest_rc.txt
tail -5 gpurun_out/r02/pytest_gpu.log
if [ $rc -ne 0 ]; then
  AMD_LOG_LEVEL=3 timeout 300 python3 -m pytest tests/test_cfr_parity.py -x -q -m gpu -p no:cacheprovider > gpurun_out/r02/pytest_amdlog.log 2>&1
  tail -60 gpurun_out/r02/pytest_amdlog.log
  timeout 600 /opt/rocm/bin/rocgdb -batch -ex run -ex bt -ex "info sharedlibrary" --args python3 -m pytest tests/test_cfr_parity.py -x -q -m gpu -p no:cacheprovider > gpurun_out/r02/gdb.log 2>&1
  grep -n -A40 "SIGABRT\|Aborted" gpurun_out/r02/gdb.log | head -120
fi
