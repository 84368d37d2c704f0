#!/bin/bash
mkdir -p gpurun_out; exec > gpurun_out/r5_b.log 2>&1
timeout 300 python -m pytest tests/test_net_parity.py -x -q -m gpu -k "class_default or fallback" 2>&1 | tail -3
echo "--- 2d6f CFR phases (flat kernel), bench-like mix"
timeout 300 python scripts/probe_cfr_phases_2d6f.py 9 2048
echo "--- all root lanes"
RBL_PROBE_ROOT=1 timeout 300 python scripts/probe_cfr_phases_2d6f.py 9 512
