#!/bin/bash
mkdir -p gpurun_out; exec > gpurun_out/session.log 2>&1
export RBL_NET_DBG=1
for rep in 1 2; do
for v in default late2 early; do
  if [ $v = default ]; then unset REBEL_HIP_LIB; else export REBEL_HIP_LIB=scratch_alt/librebel_hip_$v.so; fi
  echo "--- $v"
  timeout 120 python scripts/probe_net_shape.py 2 3 270336 40 2 | grep -v amdgpu
  timeout 120 python scripts/probe_net_shape.py 2 6 229376 40 2 | grep -v amdgpu
done
done
