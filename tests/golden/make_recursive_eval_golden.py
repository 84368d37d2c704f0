"""Generates tests/golden/recursive_eval_1d4f.json (+ recursive_eval_net_1d4f.npz): stdout of the UNMODIFIED reference
tool oracle/_ref/recursive_eval (csrc/liars_dice/recursive_eval.cc, built by oracle/Makefile `ref`) for
  (a) --net zero --cfr                       (full-tree solve only: the tool cannot repeat with the zero net), and
  (b) --net <TorchScript Net2, seed 1234> --mdp_depth 2 --num_repeats 4 --cfr
  (c) --net zero --cfr --print_regret --print_regret_summary   (the regret report of the full-tree section; round 4: made in
      the build container -- it needs no GPU -- and merged into the json: `make_recursive_eval_golden.py --only-regrets`)
  (d) --net zero --cfr --dcfr 1.5 0 2        (round 6, `--only-dcfr`: discounted CFR, made in the build container like (c))
  (e)-(g) --net zero --repeat_oracle_net [--eval_oracle_values_iters 8] (fictitious play) on 1 die x 3 / 4 faces (round 6, `--only-oracle`)
on 1 die x 4 faces.  tests/test_eval_parity.py::test_recursive_eval_tool_vs_reference_binary runs scripts/recursive_eval.py
with the same arguments and compares the XXX / YYY lines (scripts/eval_all.py:100-104 parses them).
The reference tool loads a TorchScript net on "cuda" first (recursive_eval.cc:316, real_net.cc:130-132), so (b) needs a
GPU: it was produced on the MI355X box by `gpurun -- python tests/golden/make_recursive_eval_golden.py gpurun_out` (the
prebuilt oracle/_ref/recursive_eval travels with the snapshot) and the two files were copied from gpurun_out/ into
tests/golden/.  usage: make_recursive_eval_golden.py [output dir, default tests/golden]"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rebel_amd.models import Net2  # noqa: E402  (same state_dict keys as cfvpy/models.py:64-94)

BIN = os.path.join(ROOT, "oracle", "_ref", "recursive_eval")
COMMON = ["--num_dice", "1", "--num_faces", "4", "--cfr", "--mdp_depth", "2"]


def run(args):
    with tempfile.TemporaryDirectory() as tmp:  # the tool writes strategy.*.txt into its cwd
        out = subprocess.run([BIN] + args, cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, check=True).stdout
    lines = out.splitlines()
    xxx = json.loads([l for l in lines if l.startswith("XXX ")][0][4:])
    yyy = json.loads([l for l in lines if l.startswith("YYY ")][0][4:])
    return dict(args=args, xxx=xxx, yyy=yyy, stdout=lines)


def main():
    if "--only-regrets" in sys.argv:  # merge case (c) into the committed json without touching the GPU-made cases
        path = os.path.join(ROOT, "tests", "golden", "recursive_eval_1d4f.json")
        golden = json.load(open(path))
        golden["zero_regrets"] = run(COMMON + ["--subgame_iters", "64", "--net", "zero", "--print_regret", "--print_regret_summary"])
        json.dump(golden, open(path, "w"), indent=1)
        print("\n".join(golden["zero_regrets"]["stdout"][-10:]))
        return
    if "--only-dcfr" in sys.argv:  # round 6: case (d) --dcfr 1.5 0 2 (recursive_eval.cc:248-253), full-tree section, no GPU needed
        path = os.path.join(ROOT, "tests", "golden", "recursive_eval_1d4f.json")
        golden = json.load(open(path))
        golden["zero_dcfr"] = run(COMMON + ["--dcfr", "1.5", "0", "2", "--subgame_iters", "256", "--net", "zero"])
        json.dump(golden, open(path, "w"), indent=1)
        print("\n".join(golden["zero_dcfr"]["stdout"][-8:]))
        return
    if "--only-oracle" in sys.argv:  # round 6: cases (e), (f) --repeat_oracle_net [--eval_oracle_values_iters 8] (recursive_eval.cc:
        # 325-334: the value net is a full solve of the queried subgame; no TorchScript, so no GPU needed), 1 die x 3 faces
        path = os.path.join(ROOT, "tests", "golden", "recursive_eval_1d4f.json")
        golden = json.load(open(path))
        # (fictitious play: with --cfr the reference's oracle trips `sum >= kAlmostZero` in util.h:29 on every game tried)
        small = ["--num_dice", "1", "--num_faces", "3", "--mdp_depth", "2", "--subgame_iters", "16", "--num_repeats", "2",
                 "--num_threads", "1", "--net", "zero", "--repeat_oracle_net"]
        golden["oracle_1d3f"] = run(list(small))
        golden["oracle_1d3f_iters8"] = run(small + ["--eval_oracle_values_iters", "8"])
        golden["oracle_1d4f"] = run(small[:3] + ["4"] + small[4:])
        json.dump(golden, open(path, "w"), indent=1)
        for k in ("oracle_1d3f", "oracle_1d3f_iters8", "oracle_1d4f"):
            print(k, golden[k]["xxx"], golden[k]["yyy"])
        return
    out_dir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    torch.manual_seed(1234)
    net = Net2(num_faces=4, num_dice=1, n_hidden=256, use_layer_norm=True, n_layers=2)
    with torch.no_grad():  # outputs of the size a trained net produces (the 0.01 initial scale would hide the net)
        net.output.weight.data *= 30
        net.output.bias.data *= 30
    sd = {k: v.detach().numpy() for k, v in net.state_dict().items()}
    np.savez(os.path.join(out_dir, "recursive_eval_net_1d4f.npz"), **sd)
    with tempfile.TemporaryDirectory() as tmp:
        pt = os.path.join(tmp, "net.pt")
        torch.jit.script(net).save(pt)
        golden = dict(
            zero=run(COMMON + ["--subgame_iters", "256", "--net", "zero"]),
            # report_regrets of the full-tree section (recursive_eval.cc:28-53, 285-306): needs no GPU
            zero_regrets=run(COMMON + ["--subgame_iters", "64", "--net", "zero", "--print_regret", "--print_regret_summary"]),
            net=run(COMMON + ["--subgame_iters", "32", "--num_repeats", "4", "--num_threads", "1", "--net", pt]))
    golden["net"]["args"][-1] = "tests/golden/recursive_eval_net_1d4f.npz"
    golden["net"]["xxx"]["net"] = golden["net"]["yyy"]["net"] = "tests/golden/recursive_eval_net_1d4f.npz"
    with open(os.path.join(out_dir, "recursive_eval_1d4f.json"), "w") as f:
        json.dump(golden, f, indent=1)
    print(json.dumps({k: dict(xxx=v["xxx"], yyy=v["yyy"]) for k, v in golden.items()}, indent=1))


if __name__ == "__main__":
    main()
