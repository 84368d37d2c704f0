#!/bin/bash
mkdir -p gpurun_out; exec > gpurun_out/session.log 2>&1
date
timeout 1700 python tests/p3_policy_iteration.py --dice 1 --faces 6 --sets 36 --lanes 32 --threads 32 --ref-cache tests/_p3_cache_1d6f --train-device cuda --work /tmp/p3pi > gpurun_out/p3_1d6f_paired.json 2> gpurun_out/p3_1d6f.err
echo rc=$?
date
tail -c 1500 gpurun_out/p3_1d6f_paired.json
tail -5 gpurun_out/p3_1d6f.err
