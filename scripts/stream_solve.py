"""The reference tool's first section -- "Solving the game for the full tree" (recursive_eval.cc:269-296) -- with the solver
state edge-indexed in HBM (rbl_stream_*, eval_stream.hip): linear CFR on the whole game, exploitability of the average
strategy at iterations 2^k and at the end, in the tool's output format.  Works where the dense TreeStrategy does not fit.
usage: stream_solve.py --dice 2 --faces 6 --iters 64 [--no_linear] [--json out.json]"""
import argparse
import json
import sys
import time

sys.path.insert(0, '.')
from rebel_amd import capi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dice", type=int, default=2)
ap.add_argument("--faces", type=int, default=6)
ap.add_argument("--iters", type=int, default=64)
ap.add_argument("--no_linear", action="store_true")
ap.add_argument("--json", default="")
a = ap.parse_args()
t0 = time.perf_counter()
s = capi.StreamSolver(a.dice, a.faces, capi.make_params(num_iters=a.iters, max_depth=100000, use_cfr=True,
                                                         linear_update=not a.no_linear))
t_build = time.perf_counter() - t0
print(f"num_dice={a.dice} num_faces={a.faces}")
print(f"Tree has {s.nodes} nodes; solver state on the device: {6 * 8 * s.H * s.nodes / 1e9:.1f} GB (built in {t_build:.1f} s)")
trace, t_step, t_expl = [], 0.0, 0.0
for it in range(a.iters):
    t1 = time.perf_counter()
    s.step(1)
    t_step += time.perf_counter() - t1
    if ((it + 1) & it) == 0 or it + 1 == a.iters:
        t1 = time.perf_counter()
        ex = s.exploitability()
        t_expl += time.perf_counter() - t1
        trace.append((it + 1, float(ex[0]), float(ex[1])))
        print("Iter=%8d exploitabilities=(%.3e, %.3e) sum=%.3e" % (it + 1, ex[0], ex[1], (ex[0] + ex[1]) / 2), flush=True)
ex = trace[-1]
print(f"Full FP exploitability: {(ex[1] + ex[2]) / 2:.6f} ({ex[1]:.6f},{ex[2]:.6f})")
res = dict(dice=a.dice, faces=a.faces, iters=a.iters, nodes=s.nodes, ms_per_step=t_step / a.iters * 1e3,
           ms_per_exploitability=t_expl / len(trace) * 1e3, trace=trace,
           # one step reads / writes (per edge x hand, 8 bytes): sigma r (reach) + sigma r, regrets rw, sigma w (values) + sigma r,
           # sums rw (sums); per node x hand: reach w + r, values w + r, reach w: ~9 edge passes + 5 node passes
           approx_gb_per_step=(9 + 5) * 8 * s.H * s.nodes / 1e9)
res["approx_tb_per_s"] = res["approx_gb_per_step"] / res["ms_per_step"]
print(json.dumps(res))
if a.json:
    json.dump(res, open(a.json, "w"))
