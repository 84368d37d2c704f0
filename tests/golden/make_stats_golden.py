"""Generates tests/golden/stats_with_net.json: the three numbers of the UNMODIFIED reference's
`rela.compute_stats_with_net(params, model_path)` (csrc/liars_dice/rela/pybind.cc:57-84: exploitability of the to-leaf recursive
strategy, eval_net MSE with beliefs from the net's strategy, eval_net MSE with beliefs from the full-tree solution;
stats.cc:44-153), from the reference's own pybind module compiled by oracle/Makefile (oracle/_ref/rela*.so), for the TorchScript
Net2 of tests/golden/recursive_eval_net_1d4f.npz (1 die x 4 faces) and a seed-77 Net2 on 1 die x 5 faces.
The reference loads a TorchScript net on "cuda" (real_net.cc:130-132), so this needs a GPU: it was run on the MI355X box by
`gpurun -- python tests/golden/make_stats_golden.py gpurun_out` (the prebuilt oracle/_ref module travels with the snapshot) and
the file was copied from gpurun_out/ into tests/golden/.  It also prints this repo's numbers beside the reference's.
usage: make_stats_golden.py [output dir, default tests/golden]"""
import glob
import importlib.util
import json
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rebel_amd.models import Net2  # noqa: E402  (same state_dict keys as cfvpy/models.py:64-94)


def ref_module():
    path = glob.glob(os.path.join(ROOT, "oracle", "_ref", "rela*.so"))[0]
    spec = importlib.util.spec_from_file_location("rela", path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def params(mod, d, f, iters, cfr):
    cfg = mod.RecursiveSolvingParams()
    cfg.num_dice, cfg.num_faces = d, f
    sp = cfg.subgame_params
    sp.num_iters, sp.max_depth, sp.linear_update, sp.use_cfr = iters, 2, True, cfr
    return cfg


def nets():
    sd = dict(np.load(os.path.join(ROOT, "tests", "golden", "recursive_eval_net_1d4f.npz")))
    a = Net2(num_faces=4, num_dice=1, n_hidden=256, use_layer_norm=True, n_layers=2)
    a.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    torch.manual_seed(77)
    b = Net2(num_faces=5, num_dice=1, n_hidden=256, use_layer_norm=True, n_layers=2)
    with torch.no_grad():  # outputs of the size a trained net produces
        b.output.weight.data *= 30
        b.output.bias.data *= 30
    return {"1d4f_cfr32": (a, 1, 4, 32, True), "1d4f_fp16": (a, 1, 4, 16, False), "1d5f_cfr16_seed77": (b, 1, 5, 16, True)}


def main():
    out_dir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    ref = ref_module()
    import rebel_amd.rela as ours

    golden = {}
    with tempfile.TemporaryDirectory() as tmp:
        cwd = os.getcwd()
        os.chdir(tmp)  # the reference writes strategy dumps into its cwd
        try:
            for name, (net, d, f, iters, cfr) in nets().items():
                pt = os.path.join(tmp, name + ".pt")
                torch.jit.script(net.eval()).save(pt)
                want = [float(x) for x in ref.compute_stats_with_net(params(ref, d, f, iters, cfr), pt)]
                got = [float(x) for x in ours.compute_stats_with_net(params(ours, d, f, iters, cfr), pt)]
                golden[name] = dict(num_dice=d, num_faces=f, num_iters=iters, use_cfr=cfr, reference=want)
                print(name, "reference", want, "ours", got, file=sys.stderr)
        finally:
            os.chdir(cwd)
    with open(os.path.join(out_dir, "stats_with_net.json"), "w") as fh:
        json.dump(golden, fh, indent=1)
    print(json.dumps(golden))


if __name__ == "__main__":
    main()
