"""Static instruction mix of one kernel instantiation, attributed to source lines (hipcc -gline-tables-only -S): which lines
of a .hip file the VALU / SALU / LDS / VMEM / MFMA instructions of a kernel come from.  Static counts: unrolled loops count
once per copy, rolled loops once -- a map of where the code is, not a profile.
usage: python scripts/isa_by_source.py rebel_amd/csrc/cfr_wave_kernel.hip 'cfr_wave_kernelILi6ELi13ELi1ELi6E' [-ffp-contract=off] [--top 25]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

src, key = sys.argv[1], sys.argv[2]
extra = [x for x in sys.argv[3:] if x.startswith("-") and not x.startswith("--top")]
top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 25
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with tempfile.TemporaryDirectory() as tmp:
    out = os.path.join(tmp, "k.s")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{root}/include", "-Wno-unused-result",
           "-gline-tables-only", "-S", "-o", out, src, "--cuda-device-only", *extra]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    L = open(out).read().split("\n")
start = [i for i, l in enumerate(L) if key in l and l.rstrip().endswith(":") or (key in l and ": ;" in l)][0]
end = [i for i, l in enumerate(L) if "s_endpgm" in l and i > start][0]


def cat(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_"):
        return "salu"
    return "valu" if op.startswith("v_") else "other"


cur, per, total, ops = None, collections.defaultdict(collections.Counter), collections.Counter(), collections.Counter()
for l in L[start:end]:
    m = re.match(r"\s*\.loc\s+\d+\s+(\d+)\s+\d+", l)
    if m:
        cur = int(m.group(1))
        continue
    t = l.split(";")[0].strip()
    if not t or t.startswith(".") or t.endswith(":"):
        continue
    c = cat(t.split()[0])
    per[cur][c] += 1
    total[c] += 1
    ops[t.split()[0]] += 1
text = open(src).read().split("\n")
print(f"# {src} :: {key}: {sum(total.values())} instructions, {dict(total)}")
print("# most frequent opcodes:", ", ".join(f"{op} {n}" for op, n in ops.most_common(18)))
print("# source lines by instruction count (static):")
for ln, c in sorted(per.items(), key=lambda kv: -sum(kv[1].values()))[:top]:
    code = text[ln - 1].strip()[:100] if ln and 0 < ln <= len(text) else "(no line info)"
    print(f"{sum(c.values()):5d}  valu {c['valu']:4d} salu {c['salu']:4d} lds {c['lds']:3d} vmem {c['vmem']:3d} mfma {c['mfma']:3d}  line {ln}: {code}")
