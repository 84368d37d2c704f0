"""rocprofv3 (ROCm 7.2) writes a rocpd sqlite database; this prints / saves the per-kernel summary of its `kernels` view in
the layout of `rocprofv3 --stats` CSV output.   usage: python scripts/rocpd_kernel_stats.py <results.db> [out.csv]"""
import collections, csv, sqlite3, statistics, sys

con = sqlite3.connect(sys.argv[1])
d = collections.defaultdict(list)
for name, dur in con.execute("select name, (end - start) from kernels"):
    d[name].append(dur)
tot = sum(sum(v) for v in d.values())
rows = [(n, len(v), sum(v), round(sum(v) / len(v), 3), round(100 * sum(v) / tot, 2), min(v), max(v), round(statistics.pstdev(v), 3))
        for n, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))]
out = open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout
w = csv.writer(out, quoting=csv.QUOTE_NONNUMERIC)
w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
w.writerows(rows)
