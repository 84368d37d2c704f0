#!/bin/bash
# Builds and runs scripts/micro/mfma32_fillers.hip on the GPU box; output -> gpurun_out/micro_mfma32_fillers.txt
set -e
mkdir -p gpurun_out /tmp/micro
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize scripts/micro/mfma32_fillers.hip -o /tmp/micro/mfma32_fillers
/tmp/micro/mfma32_fillers "$@" | tee gpurun_out/micro_mfma32_fillers${1}.txt
