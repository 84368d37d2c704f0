#!/bin/bash
# PMC counters of the two hot kernels (north_star: "MFMA utilisation on the net forward against chip peak", "achieved HBM GB/s
# on the CFR sweep" is scripts/collect_profiles.sh): instruction counts, busy / wait cycles per launch.  Counter passes only use
# --kernel-trace (gpurun refuses --pmc together with the trace domains).   usage (GPU box, repo root): bash scripts/pmc_hot_kernels.sh [tag]
TAG=${1:-r04}
R=$(pwd); O=$R/gpurun_out/pmc_$TAG; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P1="GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE"
rocprofv3 --pmc $P1 --kernel-trace --output-format csv -d $O/net1 -o p -- python3 $R/scripts/probe_net_only.py 589824 4 > $O/net1.log 2>&1
rocprofv3 --pmc $P2 --kernel-trace --output-format csv -d $O/net2 -o p -- python3 $R/scripts/probe_net_only.py 589824 4 > $O/net2.log 2>&1
RBL_PARTS=1 rocprofv3 --pmc $P1 --kernel-trace --output-format csv -d $O/cfr1 -o p -- python3 $R/scripts/probe_cfr_only.py 12 16384 > $O/cfr1.log 2>&1
RBL_PARTS=1 rocprofv3 --pmc $P2 --kernel-trace --output-format csv -d $O/cfr2 -o p -- python3 $R/scripts/probe_cfr_only.py 12 16384 > $O/cfr2.log 2>&1
cd $R
python3 - "$O" "$TAG" <<'PY'
import collections, csv, glob, sys
base, tag = sys.argv[1], sys.argv[2]
out = [f"# rocprofv3 PMC summaries, {tag} (per-launch means over the launches of each run).  Units: SQ_WAVE_CYCLES / SQ_WAIT_* / "
       "SQ_ACTIVE_INST_* are quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, GRBM_GUI_ACTIVE are cycles "
       "(GUI_ACTIVE summed over the 8 XCDs).  Commands: scripts/pmc_hot_kernels.sh"]
for sub, key, title in (("net1", "mlp_resident_kernel", "value-net forward, 589 824 canonical rows (through the split kernel), pass 1"),
                        ("net2", "mlp_resident_kernel", "value-net forward, pass 2"),
                        ("cfr1", "cfr_wave_kernel", "CFR step kernel, 16 384 root lanes, one stream, pass 1"),
                        ("cfr2", "cfr_wave_kernel", "CFR step kernel, pass 2")):
    f = glob.glob(f"{base}/{sub}/**/*counter_collection.csv", recursive=True)
    if not f:
        out.append(f"## {title}: no counter file")
        continue
    d = collections.defaultdict(list)
    name = None
    for r in csv.DictReader(open(f[0])):
        if key in r["Kernel_Name"]:
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void rbl::", "").split("(")[0]
            d[r["Counter_Name"]].append(float(r["Counter_Value"]))
    out.append(f"\n## {name}: {title}   [{len(next(iter(d.values()), []))} launches]")
    m = {c: sum(v) / len(v) for c, v in d.items()}
    for c in sorted(m):
        out.append(f"{c:<34}{m[c]:>16.0f}")
    if "GRBM_GUI_ACTIVE" in m and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
        cyc = m["GRBM_GUI_ACTIVE"] / 8
        simd = cyc * 256 * 4
        out.append(f"  -> kernel cycles {cyc:.0f} (GUI_ACTIVE / 8 XCDs); MFMA busy {m['SQ_VALU_MFMA_BUSY_CYCLES'] / simd:.3f} of SIMD-cycles; "
                   f"VALU active {m['SQ_ACTIVE_INST_VALU'] * 4 / simd:.3f}; any instruction active {m['SQ_ACTIVE_INST_ANY'] * 4 / simd:.3f} (per wave slot: "
                   f"{m['SQ_ACTIVE_INST_ANY'] / m['SQ_WAVE_CYCLES']:.3f} of wave-cycles); waiting {m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES']:.3f}, issue-stalled "
                   f"{m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES']:.3f}")
open(f"{base}/{tag}_pmc_summary.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
if [ -z "$KEEP_RAW" ]; then rm -rf $O/net1 $O/net2 $O/cfr1 $O/cfr2; fi
