#!/bin/bash
# developer build of librebel_hip.so with the pipelined net kernel's phase stamps compiled in (extra flags: "$@")
set -e
cd /root/repo/rebel_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -Wno-unused-result -fno-slp-vectorize "$@" -c net_pipe_kernel.hip -o _build/net_pipe_kernel.o -save-temps=obj 2>&1 | grep -v "warning: argument unused" || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../librebel_hip.so _build/cfr_kernels.o _build/cfr_rows_kernel.o _build/cfr_wave_kernel.o _build/selfplay_kernels.o _build/net_kernels.o _build/net_resident_kernel.o _build/net_pipe_kernel.o _build/engine.o
grep -E "; (NumVgprs|ScratchSize)" _build/net_pipe_kernel-hip-amdgcn-amd-amdhsa-gfx950.s | paste - - 
