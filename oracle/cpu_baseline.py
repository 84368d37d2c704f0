#!/usr/bin/env python3
"""CPU baseline = the UNMODIFIED reference data-generation path, timed on this host's cores.  TEST INFRASTRUCTURE ONLY
(used by bench.py's `cpu_baseline` leg; never imported by rebel_amd/).

Drives oracle/_ref/rela*.so -- the reference's own pybind11 module compiled from /root/reference/csrc/liars_dice by
oracle/Makefile -- exactly the way cfvpy/selfplay.py:187-252 does for `cpu_gen_threads`: one TorchScript Net2 replica
+ one ModelLocker("cpu") per generator thread, T x create_cfr_thread(locker, replay, cfg, seed=i), Context.start(),
and measures replay.num_add() over a fixed window (cfvpy/selfplay.py:285-293).  One subgame = 2 examples = num_iters
subgame-CFR-iterations.

Run as a subprocess:  python oracle/cpu_baseline.py --dice 1 --faces 6 --iters 1024 --threads 8 --seconds 20
Prints one JSON line.
"""
import argparse
import glob
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def run_once(a, T):
    """One timed window with T generator threads, in a fresh interpreter (threads only stop between games)."""
    import subprocess

    cmd = [sys.executable, os.path.abspath(__file__), "--dice", str(a.dice), "--faces", str(a.faces), "--iters",
           str(a.iters), "--threads", str(T), "--seconds", str(a.seconds), "--warmup", str(a.warmup), "--single"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=a.seconds + a.warmup + 240)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not lines:
        return {"error": (r.stderr or "no output")[-400:]}
    return json.loads(lines[-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dice", type=int, default=1)
    ap.add_argument("--faces", type=int, default=6)
    ap.add_argument("--iters", type=int, default=1024)
    ap.add_argument("--threads", type=int, default=0,
                    help="generator threads; 0 = sweep {16, 32, 60, os.cpu_count()} and report the best (60 is the "
                         "README's cpu_gen_threads setting, os.cpu_count() what BASELINE.md section 3 plans; the reference "
                         "path stops scaling well before either)")
    ap.add_argument("--seconds", type=float, default=120.0, help="total window, split evenly over the thread counts")
    ap.add_argument("--warmup", type=float, default=3.0)
    ap.add_argument("--single", action="store_true", help=argparse.SUPPRESS)
    a = ap.parse_args()
    if not glob.glob(os.path.join(HERE, "_ref", "rela*.so")):
        print(json.dumps({"error": "oracle/_ref/rela*.so not built (needs /root/reference: make -C oracle ref)"}))
        return 2
    if not a.single:
        cores = os.cpu_count() or 1
        counts = [a.threads] if a.threads else sorted({min(c, cores) for c in (16, 32, 60, cores)})
        a.seconds = a.seconds / len(counts)
        runs = [run_once(a, T) for T in counts]
        good = [r for r in runs if "value" in r]
        if not good:
            print(json.dumps({"error": str(runs)}))
            return 1
        best = max(good, key=lambda r: r["value"])
        best["sample"] += "; best of threads " + ", ".join(f"{r['threads']}: {r['value']:.0f}/s" for r in good)
        print(json.dumps(best), flush=True)
        return 0
    import torch

    torch.set_num_threads(1)  # one intra-op thread per generator thread (5x effect)
    sys.path.insert(0, os.path.join(HERE, "_ref"))
    sys.path.insert(0, ROOT)
    import rela  # the reference's module
    from rebel_amd.models import Net2  # same keys/init as the reference's class (checked in tests)

    T = a.threads
    torch.manual_seed(0)
    net = Net2(num_faces=a.faces, num_dice=a.dice, n_hidden=256, use_layer_norm=True, n_layers=2).eval()
    import io

    buf = io.BytesIO()  # script once, load T replicas (one per generator thread, as selfplay.py:187-221 does)
    torch.jit.save(torch.jit.script(net), buf)
    models, lockers = [], []
    for _ in range(T):
        buf.seek(0)
        m = torch.jit.load(buf)
        models.append(m)
        lockers.append(rela.ModelLocker([m], "cpu"))
    replay = rela.ValuePrioritizedReplay(capacity=2 ** 20, seed=10001, alpha=1.0, beta=0.4, prefetch=3,
                                         use_priority=False, compressed_values=False)
    cfg = rela.RecursiveSolvingParams()
    cfg.num_dice, cfg.num_faces = a.dice, a.faces
    cfg.random_action_prob, cfg.sample_leaf = 0.25, True
    sp = cfg.subgame_params
    sp.num_iters, sp.max_depth, sp.linear_update, sp.use_cfr = a.iters, 2, True, True
    ctx = rela.Context()
    for i in range(T):
        ctx.push_env_thread(rela.create_cfr_thread(lockers[i], replay, cfg, i))
    ctx.start()
    time.sleep(a.warmup)
    n0, t0 = replay.num_add(), time.time()
    time.sleep(a.seconds)
    n1, t1 = replay.num_add(), time.time()
    subgames = (n1 - n0) / 2
    out = {"value": subgames * a.iters / (t1 - t0), "unit": "subgame-CFR-iterations/s", "cores": T,
           "host_cores": os.cpu_count(), "threads": T, "kind": "reference", "subgames_per_s": subgames / (t1 - t0),
           "sample": f"{a.dice}dx{a.faces}f, {a.iters} iters/subgame, {T} reference gen threads x {a.seconds:.0f}s window, "
                     f"Net2(256x2,LN) TorchScript on CPU, torch intra-op threads=1"}
    print(json.dumps(out), flush=True)
    os._exit(0)  # generator threads only honour terminate() between games; do not wait for them


if __name__ == "__main__":
    sys.exit(main())
