#!/bin/bash
O=$PWD/gpurun_out/r02c; mkdir -p $O
run() { # name, dir, env...
  local name=$1; shift; local dir=$1; shift
  local fails=0
  for i in 1 2 3; do
    ( cd $dir && env "$@" timeout 120 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider -k "1d4f_dcfr_syn_64 or 1d4f_depth3" > $O/$name.$i.log 2>&1 )
    rc=$?
    if [ $rc -ne 0 ]; then fails=$((fails+1)); echo "$name run $i rc=$rc"; tail -5 $O/$name.$i.log; else rm -f $O/$name.$i.log; fi
  done
  echo "== $name: $fails/3 failed"
}
ls -la /root/repo | head -3
run symlink_cwd /root/repo A=1
run no_ldpath /root/repo -u LD_LIBRARY_PATH
run no_hipvis /root/repo -u HIP_VISIBLE_DEVICES
run no_both /root/repo -u HIP_VISIBLE_DEVICES -u LD_LIBRARY_PATH -u ROCM_PATH -u HIP_PLATFORM
run clean_env /root/repo -i PATH=/usr/local/bin:/usr/bin:/bin HOME=/root
run rocr_vis /root/repo -u HIP_VISIBLE_DEVICES ROCR_VISIBLE_DEVICES=0
