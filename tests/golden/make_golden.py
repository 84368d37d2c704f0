#!/usr/bin/env python3
"""Generates the committed golden vectors FROM THE COMPILED, UNMODIFIED REFERENCE (oracle/_ref/libref_driver.so,
built by `make -C oracle ref` where /root/reference exists).  Run from the repo root:

    python tests/golden/make_golden.py

Outputs (small, committed):
    tests/golden/solver_cases.npz   per case in tests/cases.py:SOLVER_CASES: root hand values, sha256 of every dense
                                    solver array (average / last / sum / regrets, double[N][H][A]) and -- for the
                                    small cases -- the arrays themselves
    tests/golden/rl_cases.npz       per case in RL_CASES: the emitted (query, values) example sequence of RlRunner
    tests/golden/net2_1d6f.npz      Net2(1 die x 6 faces, n_hidden=256, n_layers=2, layer norm; torch.manual_seed(0))
                                    weights + a batch of reference-produced queries + torch-CPU outputs (P2 parity)

The GPU box has no /root/reference; the parity tests there read only these files.
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import orc  # noqa: E402
from tests.cases import NET_CODE, RL_CASES, SOLVER_CASES, case_beliefs  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
FULL_ARRAY_LIMIT = 12000  # doubles per array kept verbatim


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    ref = orc.Oracle("ref")
    assert ref.impl_name() == "reference"
    names = {orc.GET_AVERAGE: "average", orc.GET_LAST: "last", orc.GET_SUM: "sum", orc.GET_REGRETS: "regrets"}

    out = {}
    for name, c in sorted(SOLVER_CASES.items()):
        p = orc.make_params(**c["p"])
        H = ref.num_hands(c["d"], c["f"])
        s = ref.solver(c["d"], c["f"], p, c.get("lb", -1), c.get("pl", 0), case_beliefs(c, H), NET_CODE[c["net"]])
        s.multistep()
        out[f"{name}/hand_values"] = np.stack([s.hand_values(0), s.hand_values(1)])
        out[f"{name}/tree_size"] = np.int64(s.N)
        for w, wn in names.items():
            if w == orc.GET_REGRETS and not p.use_cfr:
                continue
            arr = s.get(w)
            out[f"{name}/{wn}_sha256"] = np.array(sha(arr))
            if arr.size <= FULL_ARRAY_LIMIT:
                out[f"{name}/{wn}"] = arr
        if c["net"] != "none":
            s.update_value_network()
            out[f"{name}/example_queries"] = np.stack([q for q, _ in s.examples])
            out[f"{name}/example_values"] = np.stack([v for _, v in s.examples])
        print(name, "N=%d" % s.N, out[f"{name}/hand_values"][0][:3])
    np.savez_compressed(os.path.join(OUT, "solver_cases.npz"), **out)

    out = {}
    for name, c in sorted(RL_CASES.items()):
        p = orc.make_params(**c["p"])
        ex = ref.rl_run(c["d"], c["f"], p, c["seed"], c["games"], random_action_prob=c["rap"], sample_leaf=c["leaf"],
                        net=NET_CODE[c["net"]])
        out[f"{name}/queries"] = np.stack([q for q, _ in ex])
        out[f"{name}/values"] = np.stack([v for _, v in ex])
        print(name, len(ex), "examples")
    np.savez_compressed(os.path.join(OUT, "rl_cases.npz"), **out)

    # ---- value-net forward golden (P2): the reference's Net2 definition imported from /root/reference (python)
    import torch

    sys.path.insert(0, "/root/reference")
    from cfvpy.models import Net2  # the reference's own model class

    torch.manual_seed(0)
    net = Net2(num_faces=6, num_dice=1, n_hidden=256, use_layer_norm=True, n_layers=2).eval()
    queries = []
    p = orc.make_params(num_iters=40, max_depth=2, linear_update=True, use_cfr=True)
    with torch.no_grad():
        def fn(q):
            queries.append(q.copy())
            return net(torch.from_numpy(q)).numpy()
        ref.rl_run(1, 6, p, 3, 2, net=orc.NET_CALLBACK, net_fn=fn)
        q = np.concatenate(queries)[:: max(1, len(queries) // 40)][:2048]
        y = net(torch.from_numpy(q)).numpy()
    sd = {k.replace(".", "__"): v.numpy() for k, v in net.state_dict().items()}
    np.savez_compressed(os.path.join(OUT, "net2_1d6f.npz"), queries=q, outputs=y, **sd)
    print("net2 golden:", q.shape, y.shape, list(sd))


if __name__ == "__main__":
    main()
