#!/bin/bash
# usage: isa_blocks.sh file.s first_line last_line  -> per basic block: instruction-class counts; then the class string
# of the block with the most MFMAs (M mfma, v valu, a accvgpr, r ds read, w ds write, L/S scratch load/store, W waitcnt,
# n s_nop, B barrier, g global, s other salu)
S=$1; A=$2; B=$3
awk -v A=$A -v B=$B 'NR>=A && NR<=B { if ($0 ~ /^\.LBB/) {if (m>0) print start": "name" mfma="m" valu="v" ds="d" scratch="sc" lines="NR-start; name=$1; start=NR; m=0; v=0; d=0; sc=0} if ($1 ~ /^v_mfma/) m++; else if ($1 ~ /^v_/) v++; if ($1 ~ /^ds_/) d++; if ($1 ~ /^scratch_/) sc++; }' $S | sort -t= -k2 -n -r | head -${4:-6}
L=$(awk -v A=$A -v B=$B 'NR>=A && NR<=B { if ($0 ~ /^\.LBB/) {if (m>best) {best=m; bs=start; be=NR} start=NR; m=0} if ($1 ~ /^v_mfma/) m++; } END {print bs" "be}' $S)
set -- $L
sed -n $1,$2p $S | grep -v "^\s*;" | awk '{print $1}' | grep -E "^(v_|s_|ds_|scratch_|global_|buffer_)" | awk '{ if ($1 ~ /^v_mfma/) c="M"; else if ($1 ~ /^v_accvgpr/) c="a"; else if ($1 ~ /^v_/) c="v"; else if ($1 ~ /^ds_read|^ds_load/) c="r"; else if ($1 ~ /^ds_/) c="w"; else if ($1 ~ /^scratch_load/) c="L"; else if ($1 ~ /^scratch_store/) c="S"; else if ($1 ~ /^s_waitcnt/) c="W"; else if ($1 ~ /^s_nop/) c="n"; else if ($1 ~ /^s_barrier/) c="B"; else if ($1 ~ /^global/) c="g"; else c="s"; printf "%s", c } END {print ""}' | fold -w 150
