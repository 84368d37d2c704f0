"""Summarises a scripts/collect_profiles.sh run: per-kernel stats of the kernel trace and the HBM-side traffic per launch
from the FETCH_SIZE / WRITE_SIZE passes.   usage: python scripts/pmc_summary.py <gpurun_out/prof_TAG> <TAG>

FETCH_SIZE / WRITE_SIZE are in KB.  On gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes (MI355X_MICROARCH.md,
"HBM"): it is doubled here, as the guide prescribes for wide coalesced streaming reads (what both kernels issue);
WRITE_SIZE is taken as reported.  Infinity-Cache hits are counted by both, so these are memory-side (fabric) bytes."""
import collections
import csv
import glob
import json
import os
import statistics
import sys

base, tag = sys.argv[1], sys.argv[2]


def find(sub, suffix):
    hits = glob.glob(os.path.join(base, sub, "**", "*" + suffix), recursive=True)
    return hits[0] if hits else None


def short(name):
    for key in ("mlp_resident_kernel", "mlp_pipe_kernel", "mlp_fsplit_forward_kernel", "cfr_rows_kernel", "cfr_wave_kernel", "cfr_flat_kernel", "split_queries_kernel", "unsplit_queries_kernel", "cfr_big_kernel",
                "cfr_step_kernel", "sp_begin_kernel", "sp_scan_kernel", "sp_end_kernel", "synthetic_net_kernel"):
        if key in name:
            return name[name.index(key):].split("(")[0]
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:80]


out = {"tag": tag, "kernels": {}, "note": "durations from the --kernel-trace pass (ns); fetch/write from the separate "
       "--pmc passes (kernels serialised by the counter collection), KB -> bytes, FETCH_SIZE x2 (gfx950)"}
try:
    _bj = json.load(open(os.path.join(base, "bench_under_rocprof.json")))
    out["lanes"] = _bj["config"]["lanes_per_gpu"]
    out["steps_warmup"] = [_bj["steps"], _bj["warmup"]]
except Exception:
    out["lanes"] = None
kt = find("kt", "kernel_trace.csv")
if kt:
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(kt)):
        d[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    tot = sum(sum(v) for v in d.values())
    rows = sorted(d.items(), key=lambda kv: -sum(kv[1]))
    with open(os.path.join(base, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for n, v in rows:
            w.writerow([n, len(v), sum(v), round(sum(v) / len(v), 1), round(100 * sum(v) / tot, 2), min(v), max(v),
                        round(statistics.pstdev(v), 1)])
            out["kernels"].setdefault(n, {}).update(calls=len(v), avg_ns=sum(v) / len(v))
# the bench times only the epochs after its warm-up: the same average restricted to those launches (the first
# warmup / (warmup + steps) of a hot kernel's launches belong to the untimed, root-heavy warm-up epochs)
try:
    bj = json.load(open(os.path.join(base, "bench_under_rocprof.json")))
    wu, st = int(bj["warmup"]), int(bj["steps"])
    d2 = collections.defaultdict(list)
    for r in csv.DictReader(open(kt)):
        d2[short(r["Kernel_Name"])].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    lines = ["kernel,launches_all,avg_ns_all,launches_timed_epochs,avg_ns_timed_epochs,bench_hip_event_avg_ns"]
    for key, field in (("mlp_resident_kernel", "roofline"), ("cfr_wave_kernel", "roofline_cfr"), ("cfr_rows_kernel", "roofline_cfr")):
        for n, v in d2.items():
            if key in n and len(v) % (wu + st) == 0 and len(v) >= 64:
                v.sort()
                timed = [x[1] for x in v[len(v) * wu // (wu + st):]]
                lines.append(f"\"{n}\",{len(v)},{sum(x[1] for x in v) / len(v):.1f},{len(timed)},{sum(timed) / len(timed):.1f},"
                             f"{bj[field]['avg_launch_us'] * 1000:.1f}")
    open(os.path.join(base, f"{tag}_kernel_stats_timed_epochs.csv"), "w").write("\n".join(lines) + "\n")
except Exception as ex:  # summary only
    print("timed-epoch stats skipped:", ex)
for sub, cname, scale in (("fetch", "FETCH_SIZE", 2.0), ("write", "WRITE_SIZE", 1.0)):
    cc = find(sub, "counter_collection.csv")
    if not cc:
        continue
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(cc)):
        if r["Counter_Name"] == cname:
            d[short(r["Kernel_Name"])].append((int(r.get("Dispatch_Id", 0)), float(r["Counter_Value"]) * 1024.0 * scale))
    try:
        bj = json.load(open(os.path.join(base, "bench_under_rocprof.json")))
        wu, st = int(bj["warmup"]), int(bj["steps"])
    except Exception:
        wu, st = 0, 1
    for n, dv in d.items():
        dv.sort()
        v = [x[1] for x in dv]
        # the init / query-only launches of cfr_step_kernel and the first launches (cold caches) are in there too: median
        rec = {"mean": sum(v) / len(v), "median": statistics.median(v), "n": len(v)}
        if len(v) % (wu + st) == 0 and len(v) >= 64:  # a per-iteration kernel: the launches of the bench's TIMED epochs only
            timed = v[len(v) * wu // (wu + st):]
            rec["timed_epochs"] = {"mean": sum(timed) / len(timed), "median": statistics.median(timed), "n": len(timed)}
        out["kernels"].setdefault(n, {})[cname.lower() + "_bytes_per_launch"] = rec
json.dump(out, open(os.path.join(base, f"{tag}_pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
