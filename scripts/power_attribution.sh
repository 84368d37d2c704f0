#!/bin/bash
# Energy attribution of the value-net forward (round 4): the product library next to two knock-out builds of
# net_resident_kernel.hip, each driven by scripts/power_trace.py (socket power + gfx clock at ~10 ms, us per launch).
#   base      the product kernel
#   ko_mfma   no matrix instructions (operands kept alive: weight registers, B fragments from LDS)
#   ko_epi    no LayerNorm scale / GELU / split in the epilogue
# energy per launch = mean power x time per launch; idle power is what the first phase of every run shows.
# usage (GPU box, repo root): bash scripts/power_attribution.sh [seconds per phase]
set -e
R=$(pwd); S=$R/rebel_amd/csrc/scratch_libs; mkdir -p $S gpurun_out/r04_power
cd rebel_amd/csrc
for v in ko_mfma ko_epi "ko_mfma ko_epi"; do
  n=$(echo $v | tr ' ' '_'); flags=""
  for f in $v; do flags="$flags -DRBL_$(echo $f | tr a-z A-Z)"; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -Wno-unused-result $flags -c net_resident_kernel.hip -o $S/nrk_$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $S/librebel_hip_$n.so $(ls _build/*.o | grep -v -e net_resident_kernel.o -e rela_module.o) $S/nrk_$n.o
done
cd $R
export RBL_QSPLIT=0 POWER_TRACE_TILES="5" POWER_TRACE_NET_ONLY=1
python scripts/power_trace.py ${1:-5} > gpurun_out/r04_power/base.txt 2> gpurun_out/r04_power/base.err
for n in ko_mfma ko_epi ko_mfma_ko_epi; do
  REBEL_HIP_LIB=$S/librebel_hip_$n.so python scripts/power_trace.py ${1:-5} > gpurun_out/r04_power/$n.txt 2> gpurun_out/r04_power/$n.err
done
tail -n 2 gpurun_out/r04_power/*.txt
