#!/bin/bash
# developer aid: registers / scratch of every instantiation of the resident net kernel, compile only
# usage: scripts/cc_resident.sh [extra hipcc flags, e.g. -DRBL_KTAIL4=5]
cd "$(dirname "$0")/../rebel_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -Wno-unused-result -ffp-contract=off "$@" -Rpass-analysis=kernel-resource-usage -c net_resident_kernel.hip -o /tmp/nrk_x.o 2>&1 | grep -E "Name:|VGPRs:|ScratchSize" | sed -E 's/.*(Name: [^ ]*|VGPRs: [0-9]*|ScratchSize \[bytes\/lane\]: [0-9]*).*/\1/' | tr '\n' ' ' | sed 's/Name:/\nName:/g' | sed -E 's/_ZN3rbl12_GLOBAL__N_119mlp_resident_kernelILi([0-9])ELb([01])ELi([0-9])ELi([0-9])E\w*/K0C=\1 LN=\2 NOT=\3 PROD=\4/'
echo
