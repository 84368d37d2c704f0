#!/bin/bash
# second GPU pass of round 4: energy attribution of the net forward; config 5's exploitability check sharded at the second
# recursion level (8 shards back to back on one GPU) and once at its stated size (subgame_iters = 2048)
O=gpurun_out/r04b; mkdir -p $O
bash scripts/power_attribution.sh 5 > $O/power_attribution.log 2>&1; echo "power_attribution rc=$?" | tee -a $O/rc.txt
E="python scripts/exploitability_stream.py --dice 2 --faces 6 --depth 2 --lanes 8192"
$E --iters 32 --json $O/expl_i32_unsharded.json > $O/expl_i32_unsharded.txt 2>&1
for s in 0 1 2 3 4 5 6 7; do
  $E --iters 32 --shard $s --n_shards 8 --deal_levels 2 --json $O/expl_i32_shard$s.json >> $O/expl_i32_shards.txt 2>&1
done
$E --combine $O/expl_i32_shard?.json > $O/expl_i32_combined.txt 2>&1
cat $O/expl_i32_unsharded.txt $O/expl_i32_combined.txt | cut -c1-600
rm -f $O/expl_i32_shard?.json $O/expl_i32_unsharded.json   # 15 MB of top values each: the text lines are the record
timeout 1200 $E --iters 2048 > $O/expl_i2048.txt 2>&1; echo "expl2048 rc=$?" | tee -a $O/rc.txt
cut -c1-600 $O/expl_i2048.txt
