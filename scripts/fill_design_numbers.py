"""Fills the @@...@@ placeholders of DESIGN.md section 4 (round-4 block) from profiles/r04_bench.json, so that the text
quotes exactly what the committed bench line says.  usage: python scripts/fill_design_numbers.py"""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.load(open(os.path.join(ROOT, "profiles", "r04_bench.json")))
rn, rc = d["roofline"], d["roofline_cfr"]


def traffic(r):
    t = r.get("traffic_detail")
    if not t:
        return "--"
    return f"{t['read'] / 1e6:.1f} MB read (FETCH×2) + {t['written'] / 1e6:.1f} MB written = {r['traffic'] / 1e6:.1f} MB"


def cfg(c):
    ref = c.get("cpu_reference", {})
    return (f"**{c['value'] / 1e6:.2f} M it/s** | {c['ms_per_step']:.1f} | {c['net']['avg_launch_us']:.1f} µs, "
            f"{c['net']['rows_per_launch'] / 1e3:.0f} k rows, frac **{c['net']['frac']:.3f}** | {c['cfr']['avg_launch_us']:.1f} µs, frac "
            f"**{c['cfr']['frac']:.3f}** | {ref.get('value', 0) / 1e3:.0f} k it/s ({ref.get('cores', '?')} threads) | "
            f"{c.get('speedup_vs_cpu_reference', 0):.0f}×")


cpu = d["cpu_baseline"]
per = dict(re.findall(r"(\d+): (\d+)/s", cpu["sample"]))
c = {x["baseline_config"]: x for x in d["configs"]}
rep = {
    "VALUE": f"{d['value'] / 1e6:.1f}", "GAMES": f"{d['games_per_s'] / 1e3:.1f}", "MS": f"{d['ms_per_step']:.1f}",
    "POWER": f"{d.get('power', {}).get('mean_socket_power_w', 0):.0f}", "CLK": f"{d.get('power', {}).get('mean_gfxclk_mhz', 0):.0f}",
    "NETUS": f"{rn['avg_launch_us']:.1f}", "NETTF": f"{rn['achieved']:.0f}", "NETFRAC": f"{rn['frac']:.3f}", "NETTRAFFIC": traffic(rn),
    "CFRUS": f"{rc['avg_launch_us']:.1f}", "CFRGB": f"{rc['achieved'] / 1e3:.2f}", "CFRFRAC": f"{rc['frac']:.3f}", "CFRTRAFFIC": traffic(rc),
    "C1": cfg(c[1]), "C3": cfg(c[3]), "C4": cfg(c[4]),
    "TWOSTR": f"{d['two_streams']['value'] / 1e6:.1f}", "L4096": f"{d['lanes_4096']['value'] / 1e6:.1f}",
    "L4096R": f"{d['lanes_4096']['value'] / d['value']:.2f}", "HALF": f"{d['half_inference']['value'] / 1e6:.1f}",
    "HALFUS": f"{d['half_inference']['net']['avg_launch_us']:.1f}", "HALFFRAC": f"{d['half_inference']['net']['frac']:.3f}",
    "CPU": f"{cpu['value'] / 1e3:.1f}", "CPU32": f"{int(per.get('32', 0)) / 1e3:.1f}", "CPU60": f"{int(per.get('60', 0)) / 1e3:.1f}",
    "CPU256": f"{int(per.get('256', 0)) / 1e3:.1f}", "RATIO": f"{d['speedup_vs_cpu_baseline']:.0f}",
}
p = os.path.join(ROOT, "DESIGN.md")
s = open(p).read()
missing = [k for k in rep if f"@@{k}@@" not in s]
for k, v in rep.items():
    s = s.replace(f"@@{k}@@", v)
left = re.findall(r"@@\w+@@", s)
open(p, "w").write(s)
print("filled", len(rep) - len(missing), "placeholders; not found:", missing, "; left:", left)
