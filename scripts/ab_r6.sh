#!/bin/bash
# Round-6 same-box A/B: the library in the tree against scratch_alt/librebel_hip_r5.so (round-5 HEAD, built by hand from
# `git archive 2d04b4d rebel_amd/csrc include`; not tracked).  Phase stamps of the two CFR step kernels and short bench legs.
# usage (GPU box): bash scripts/ab_r6.sh > gpurun_out/ab_r6.txt 2>&1
OLD=$PWD/scratch_alt/librebel_hip_r5.so
one() {  # label, env assignment for the library
  echo "########## $1"
  echo "=== cfr_wave_kernel phase stamps, 1 die x 6 faces, 16384 root lanes"
  env $2 python3 scripts/probe_cfr_phases.py 9 16384
  echo "=== cfr_wave_kernel phase stamps, 2 dice x 3 faces, 16384 root lanes"
  env $2 RBL_PROBE_DICE=2 RBL_PROBE_FACES=3 python3 scripts/probe_cfr_phases.py 9 16384
  echo "=== cfr_flat_kernel phase stamps, 2 dice x 6 faces, 2048 lanes (bench-like mix)"
  env $2 python3 scripts/probe_cfr_phases_2d6f.py 9 2048
  echo "=== bench legs"
  run() {
    env $LIBENV python3 bench.py --no-cpu-baseline --no-extra-legs "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$*', '-> value=%.2fM it/s ms/step=%.1f net_us=%.1f (frac %.3f) cfr_us=%.1f (frac %.3f)' % (d['value']/1e6, d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['roofline_cfr']['avg_launch_us'], d['roofline_cfr']['frac']))"
  }
  LIBENV=$2
  run --dice 1 --faces 6 --iters 1024 --lanes 16384 --steps 6 --warmup 3
  run --dice 1 --faces 6 --iters 1024 --lanes 4096 --steps 10 --warmup 4
  run --dice 1 --faces 4 --iters 1024 --lanes 4096 --steps 10 --warmup 4
  run --dice 2 --faces 3 --iters 1024 --lanes 16384 --steps 4 --warmup 3
  run --dice 2 --faces 6 --iters 2048 --lanes 2048 --steps 6 --warmup 8
}
one "round 6 (tree)" "RBL_AB=new"
[ -f "$OLD" ] && one "round 5 (scratch_alt/librebel_hip_r5.so)" "REBEL_HIP_LIB=$OLD"
one "round 6 (tree), again" "RBL_AB=new"
