#!/bin/bash
# Developer aid: A/B builds of librebel_hip.so that differ in net_resident_kernel.hip's -D knobs only, into scratch_alt/
# (git-ignored, travels to the GPU box).  Select one at run time with REBEL_HIP_LIB=scratch_alt/librebel_hip_<name>.so.
# usage: build_net_variants.sh name1 "-DFLAG..." [name2 "-DFLAG..." ...]
set -e
cd "$(dirname "$0")/../rebel_amd/csrc"
make -s lib
mkdir -p ../../scratch_alt
OBJS=$(ls _build/*.o | grep -v "net_resident_kernel.o\|rela_module.o")
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  (
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -Wno-unused-result $flags \
        -c net_resident_kernel.hip -o ../../scratch_alt/net_resident_$name.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scratch_alt/librebel_hip_$name.so $OBJS \
        ../../scratch_alt/net_resident_$name.o
    echo "built scratch_alt/librebel_hip_$name.so ($flags)"
  ) &
done
wait
