"""Developer aid: approximate VGPR liveness over the main loop of a kernel in hipcc's -S output (straight-line
approximation: branches inside the loop body are ignored, the body is walked backwards twice so loop-carried values count).
usage: vgpr_liveness.py file.s kernel_name_substring [first_loop_label]"""
import re
import sys

src = open(sys.argv[1]).read().split("\n")
name = sys.argv[2]
start = next(i for i, l in enumerate(src) if l.startswith("_Z") and name in l and l.rstrip().endswith(tuple(":")) or (l.startswith("_Z") and name in l and ":" in l))
end = next(i for i in range(start, len(src)) if "s_endpgm" in src[i])
body = src[start:end]
# the main loop: from the first label that is the target of a backward branch with the largest span
labels = {l.split(":")[0]: i for i, l in enumerate(body) if l.startswith(".LBB")}
best = (0, 0, 0)
for i, l in enumerate(body):
    m = re.search(r"s_cbranch_\w+\s+(\.LBB\w+)|s_branch\s+(\.LBB\w+)", l)
    if m:
        t = m.group(1) or m.group(2)
        if t in labels and labels[t] < i and i - labels[t] > best[0]:
            best = (i - labels[t], labels[t], i)
_, lo, hi = best
loop = body[lo:hi + 1]
print(f"kernel lines {start}-{end}, main loop lines {lo}-{hi} of the kernel ({hi - lo} lines)")


def regs(tok):
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", tok):
        out.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", tok):
        out.add(int(a))
    return out


STORE = ("ds_write", "ds_store", "global_store", "scratch_store", "flat_store", "buffer_store", "v_cmp", "v_readlane", "v_readfirstlane",
         "s_", "ds_bpermute_noret", "global_atomic", "ds_add")
RMW = ("v_writelane", "v_permlane", "v_fma_mixhi", "v_mfma", "v_fmac", "v_pk_fmac", "v_dot")
ins = []
for l in loop:
    l = l.split(";")[0].strip()
    if not l or l.startswith(".") or l.endswith(":"):
        ins.append((l, set(), set()))
        continue
    op, _, rest = l.partition(" ")
    ops = [o.strip() for o in rest.split(",")]
    if op.startswith(STORE):
        d, u = set(), set().union(*[regs(o) for o in ops]) if ops else set()
    else:
        d = regs(ops[0]) if ops else set()
        u = set().union(*[regs(o) for o in ops[1:]]) if len(ops) > 1 else set()
        if op.startswith(RMW) and not (op.startswith("v_mfma") and ops[-1].strip() == "0"):
            if op.startswith("v_mfma"):
                pass  # acc operand is listed among the sources already
            else:
                u |= d
    ins.append((l, d, u))
live = set()
prof = [0] * len(ins)
for _ in range(2):
    for i in range(len(ins) - 1, -1, -1):
        _, d, u = ins[i]
        live = (live - d) | u
        prof[i] = len(live)
print("max live", max(prof), "at loop line", prof.index(max(prof)))
marks = [i for i, (l, _, _) in enumerate(ins) if l.startswith("s_barrier") or "sched_barrier" in loop[i]]
step = max(1, len(ins) // 60)
for i in range(0, len(ins), step):
    seg = prof[i:i + step]
    tag = "B" if any(ins[k][0].startswith("s_barrier") for k in range(i, min(len(ins), i + step))) else " "
    m = sum(1 for k in range(i, min(len(ins), i + step)) if ins[k][0].startswith("v_mfma"))
    sc = sum(1 for k in range(i, min(len(ins), i + step)) if ins[k][0].startswith("scratch_"))
    print(f"{i:5d} {tag} live max {max(seg):3d} min {min(seg):3d}  mfma {m:3d} scratch {sc}")

if len(sys.argv) > 3:
    at = int(sys.argv[3])
    live = set()
    for rnd in range(2):
        for i in range(len(ins) - 1, -1, -1):
            _, d, u = ins[i]
            live = (live - d) | u
            if rnd == 1 and i == at:
                s = sorted(live)
                rng, st = [], None
                for r in s:
                    if st is None:
                        st = pr = r
                    elif r == pr + 1:
                        pr = r
                    else:
                        rng.append((st, pr)); st = pr = r
                rng.append((st, pr))
                print("live at", at, ":", " ".join(f"v{a}" if a == b else f"v[{a}:{b}]" for a, b in rng), "=", len(s))
                dead = sorted(set(range(256)) - live)
                print("not live:", dead)
