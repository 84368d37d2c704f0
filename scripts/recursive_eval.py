#!/usr/bin/env python3
"""GPU counterpart of the reference's `recursive_eval` tool (csrc/liars_dice/recursive_eval.cc:193-426), on the C ABI.

    python scripts/recursive_eval.py --num_dice 1 --num_faces 4 --subgame_iters 1024 --mdp_depth 2 --num_repeats 64 \
        --net zero|<Net2 state dict .pt/.npz> [--cfr] [--dcfr ALPHA BETA GAMMA] [--no_linear] [--root_only] [--seed0 0]

What it reproduces:
  * "Solving the game for the full tree": the full-tree solver with exploitability printed at iterations 2^k and at the
    end (recursive_eval.cc:269-296);
  * "Recursive solving": `num_repeats` sampled strategies (compute_sampled_strategy_recursive_to_leaf, seeds 0..R-1,
    recursive_solving.cc:301-327 -- on the GPU every subgame of a tree level is a lane, rbl_strategy_recursive_sampled),
    averaged with the reach of the acting player as weights (recursive_eval.cc:136-160, 336-363; float32 like the
    reference's tensors), exploitability at 2^k repeats and at the end;
  * the machine-readable line `XXX {"net":..., "full_tree":..., "repeated toleaf N":...}` that scripts/eval_all.py:100-104
    greps for.
  * the `YYY {...}` line: EV of the full-tree strategy against each evaluated strategy (compute_ev2, rbl_ev2).
  * `--print_regret` / `--print_regret_summary` (recursive_eval.cc:28-53): immediate regrets of the list of sampled
    strategies (compute_immediate_regrets, subgame_solving.cc:984-1050 -> rbl_immediate_regrets), CFR runs only, as in
    the reference (:354-357).
  * `--repeat_oracle_net [--eval_oracle_values_iters N]` (recursive_eval.cc:232, 245-247, 325-334): the value net of the repeated
    solves is an oracle -- a full-depth solve of every queried subgame (class OracleNet: the rows of a query batch are lanes
    of one full-depth engine).  Dense mode only.
Not reproduced: strategy dumps (strategy.*.txt).

`--stream`: the same tool with every full-tree array edge-indexed in HBM (rbl_stream_*, eval_stream.hip) instead of dense
[N][H][A] host arrays -- the only way to run it at 2 dice x 6 faces (33.5 M nodes: 241 GB dense, 9.7 GB edge-indexed per
strategy).  CFR only (with or without --root_only, with or without the regret reports); the numbers are bit-identical to the dense path where both run
(tests/test_eval_parity.py::test_stream_sampled_repeats_bit_exact).
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def reach_of_actor(tree, strategy, H):
    """stats.reach_probabilities[player(node)][node] (stats.cc / subgame_solving.cc:54-78): own-action reach of the
    player acting at the node, from uniform initial beliefs.  tree rows (rbl_unroll_tree): (last_bid, player, children_begin,
    children_end, parent, depth)."""
    N = len(tree)
    reach = np.zeros((2, N, H))
    reach[:, 0, :] = 1.0 / H
    for n in range(N):
        _, player, cb, ce, _, _ = tree[n]
        if cb == ce:
            continue
        lo = tree[cb][0]  # first legal action = last_bid of the first child
        for c in range(cb, ce):
            a = lo + (c - cb)
            for p in (0, 1):
                reach[p, c] = reach[p, n] * strategy[n, :, a] if p == player else reach[p, n]
    players = tree[:, 1]
    return reach[players, np.arange(N)]  # [N][H]


def load_net(eng, path):
    if path == "zero":
        eng.set_net_zero()
        return
    from rebel_amd.models import mlp_weights_from_state_dict
    import torch
    if path.endswith(".npz"):
        sd = {k: torch.from_numpy(v) for k, v in np.load(path).items()}
    else:
        sd = torch.load(path, map_location="cpu")
    eng.set_net_mlp(*mlp_weights_from_state_dict(sd))


class OracleNet:
    """create_oracle_value_predictor (real_net.cc:89-128; recursive_eval.cc:325-334, `--repeat_oracle_net`): the "value net" is a
    full-depth solve of the queried subgame -- deserialize_query (subgame_solving.cc:911-929), build_solver(game, state, beliefs,
    params with max_depth = 100000[, num_iters = --eval_oracle_values_iters]), multistep, get_hand_values(traverser).  The
    reference solves the rows of a query batch one after the other on the host; here every row of a batch is a LANE of one
    full-depth engine on the device."""

    def __init__(self, capi, d, f, params, device, max_lanes):
        self.capi, self.d, self.f, self.params, self.device = capi, d, f, params, device
        self.eng, self.cap, self.want = None, 0, max_lanes
        self.A, self.H = 2 * d * f + 1, f ** d
        self.calls = self.rows = 0

    def __call__(self, q):
        rows, A, H = q.shape[0], self.A, self.H
        out = np.zeros((rows, H), np.float32)
        if self.eng is None:
            self.cap = max(64, min(self.want, 1 << 14))
            self.eng = self.capi.Engine(self.d, self.f, self.params, max_lanes=self.cap, device=self.device)
            self.eng.set_net_zero()
        self.calls += 1
        self.rows += rows
        for r0 in range(0, rows, self.cap):
            qq = q[r0:r0 + self.cap]
            n = qq.shape[0]
            player = (qq[:, 0] + 0.5).astype(np.int32)
            trav = (qq[:, 1] + 0.5).astype(np.int32)
            hot = qq[:, 2:2 + A] > 0.5
            bid = np.where(hot.any(1), A - 1 - np.argmax(hot[:, ::-1], axis=1), -1).astype(np.int32)  # the LAST hot action (:918-922)
            beliefs = np.stack([qq[:, 2 + A:2 + A + H], qq[:, 2 + A + H:2 + A + 2 * H]], axis=1).astype(np.float64)
            self.eng.reset(bid, player, beliefs)
            self.eng.multistep()
            for i in range(n):
                out[r0 + i] = self.eng.hand_values(i, int(trav[i]))  # row_values -> float32 tensor (real_net.cc:99)
        return out


def main_stream(a):
    import time

    from rebel_amd import capi

    assert a.cfr, "--stream: CFR solvers only"
    assert not a.repeat_oracle_net, "--stream: the oracle net runs in dense mode only"
    d, f = a.num_dice, a.num_faces
    base = solver_params(a)
    t0 = time.perf_counter()
    s = capi.StreamSolver(d, f, capi.make_params(max_depth=100000, **base), device=a.device)
    print(f"num_dice={d} num_faces={f}")
    print(f"Tree of depth {s.A} has {s.nodes} nodes")
    print("##############################################\n##### Solving the game for the full tree #####\n"
          "##############################################")
    want_regrets = a.print_regret or a.print_regret_summary

    def regret_report(first, sums):  # report_regrets, recursive_eval.cc:28-53, from the regrets accumulated on the device
        out = ""
        if a.print_regret:
            out += "\tRegrets: " + "".join(" ".join("%.6f" % x for x in row) + " | " for row in first) + "\n"
        if a.print_regret_summary:
            out += "\tRegrets (depth<=%d)/rest: %.6f/%.6f" % (a.mdp_depth, sums[0], sums[1])
        return out

    if want_regrets:
        s.regrets_reset()
    for it in range(a.subgame_iters):
        s.step(1)
        if it % 2 == 0 and want_regrets:
            s.regrets_add(capi.GET_LAST)  # the tool lists the sampling strategy after every even iteration (:285-287)
        if ((it + 1) & it) == 0 or it + 1 == a.subgame_iters:
            ex = s.exploitability()
            print("Iter=%8d exploitabilities=(%.3e, %.3e) sum=%.3e" % (it + 1, ex[0], ex[1], (ex[0] + ex[1]) / 2), flush=True)
    print(f"Full FP exploitability: {(ex[0] + ex[1]) / 2:.6f} ({ex[0]:.6f},{ex[1]:.6f})")
    print(regret_report(*s.regrets_report(a.mdp_depth)) if want_regrets else "")
    t_full = time.perf_counter() - t0
    results = [("net", a.net), ("full_tree", "%.6f" % ((ex[0] + ex[1]) / 2))]
    results_ev = [("net", a.net), ("full_tree", "%.6f" % 0.0)]  # compute_ev2 of a strategy against itself: (x - x) / 2
    t_rep = 0.0
    if a.net:
        assert a.mdp_depth > 0, "--mdp_depth is required with --net"
        print("##############################################\n##### Recursive solving                      #\n"
              "##############################################")
        eng = capi.Engine(d, f, capi.make_params(max_depth=a.mdp_depth, **base), max_lanes=a.max_lanes, device=a.device)
        load_net(eng, a.net)
        t1 = time.perf_counter()
        if want_regrets:
            s.regrets_reset()  # a new list for this section (:343)
        for sid in range(max(a.num_repeats, 0)):
            s.sampled_add(eng, sid, root_only=a.root_only)
            if want_regrets:
                s.regrets_add(capi.GET_SAMPLED)
            if ((sid + 1) & sid) == 0 or sid + 1 == a.num_repeats:
                ex, ev = s.sampled_eval()
                print("%5d: %.6f (%.6f,%.6f)\tEV of full: %.6f (%.6f,%.6f)" % (sid + 1, (ex[0] + ex[1]) / 2, ex[0], ex[1],
                                                                          (ev[0] + ev[1]) / 2, ev[0], ev[1])
                      + (regret_report(*s.regrets_report(a.mdp_depth)) if want_regrets else ""), flush=True)
                results.append((f"repeated toleaf {sid + 1}", "%.6f" % ((ex[0] + ex[1]) / 2)))
                results_ev.append((f"repeated toleaf {sid + 1}", "%.6f" % ((ev[0] + ev[1]) / 2)))
        t_rep = time.perf_counter() - t1
    for name, val in results[1:]:
        print(f" {name} {val}")
    print("XXX " + json.dumps(dict(results), separators=(", ", ":")))
    print("YYY " + json.dumps(dict(results_ev), separators=(", ", ":")))
    print(f"# stream: full-tree solve {t_full:.1f} s, {max(a.num_repeats, 0)} repeats {t_rep:.1f} s")


def solver_params(a):
    """base_params of the tool (recursive_eval.cc:211-253, :272-273): --dcfr sets the three exponents and takes linear_update off;
    without --cfr the reference builds FP, which never reads the dcfr fields (subgame_solving.cc:791-800) -- the engine refuses
    dcfr without use_cfr, so they are passed on only with --cfr."""
    base = dict(num_iters=a.subgame_iters, linear_update=not a.no_linear and a.dcfr is None, use_cfr=a.cfr, optimistic=a.optimistic)
    if a.dcfr is not None and a.cfr:
        base.update(dcfr=True, dcfr_alpha=a.dcfr[0], dcfr_beta=a.dcfr[1], dcfr_gamma=a.dcfr[2])
    return base


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--num_dice", type=int, default=1)
    ap.add_argument("--num_faces", type=int, default=4)
    ap.add_argument("--subgame_iters", type=int, default=1024)
    ap.add_argument("--mdp_depth", type=int, default=-1)
    ap.add_argument("--num_repeats", type=int, default=-1)
    ap.add_argument("--net", default="")
    ap.add_argument("--root_only", action="store_true")
    ap.add_argument("--no_linear", action="store_true")
    ap.add_argument("--optimistic", action="store_true")
    ap.add_argument("--cfr", action="store_true")
    # recursive_eval.cc:248-253: discounted CFR with (alpha, beta, gamma); it switches linear averaging off (:273)
    ap.add_argument("--dcfr", type=float, nargs=3, metavar=("ALPHA", "BETA", "GAMMA"), default=None)
    # recursive_eval.cc:232, 245-247: the value net of the repeated solves is an ORACLE (a full solve of the queried subgame, with
    # --eval_oracle_values_iters iterations when given); --net must still be non-empty, as in the reference (:313)
    ap.add_argument("--repeat_oracle_net", action="store_true")
    ap.add_argument("--eval_oracle_values_iters", type=int, default=-1)
    ap.add_argument("--print_regret", action="store_true")
    ap.add_argument("--print_regret_summary", action="store_true")
    ap.add_argument("--num_threads", type=int, default=10)  # accepted for command-line compatibility; lanes replace threads
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--max_lanes", type=int, default=4096)
    ap.add_argument("--stream", action="store_true")
    a = ap.parse_args()
    if a.stream:
        return main_stream(a)

    from rebel_amd import capi

    d, f = a.num_dice, a.num_faces
    base = solver_params(a)
    tree = capi.unroll_tree(d, f, -1, 0, 1000000)
    print(f"num_dice={d} num_faces={f}")
    print(f"Tree of depth {int(tree[:, 5].max())} has {len(tree)} nodes")
    print("##############################################\n##### Solving the game for the full tree #####\n"
          "##############################################")
    full = capi.Engine(d, f, capi.make_params(max_depth=100000, **base), max_lanes=1, device=a.device)
    full.set_net_zero()
    H = full.H
    want_regrets = a.print_regret or a.print_regret_summary

    def regret_lines(reg, total_nodes):  # report_regrets, recursive_eval.cc:28-53 (the tool prints fixed, 6 decimals: :208)
        out = ""
        if a.print_regret:
            out += "\tRegrets: " + "".join(" ".join("%.6f" % x for x in reg[n]) + " | " for n in range(min(20, total_nodes))) + "\n"
        return out

    full.reset([-1], [0], np.full((1, 2, H), 1.0 / H))
    full_list = []  # the sampling strategy after every even iteration (:285-287); only kept when a report is asked for
    for it in range(a.subgame_iters):
        full.step(it % 2)
        if it % 2 == 0 and a.cfr and want_regrets:
            full_list.append(full.get(0, capi.GET_LAST))
        if ((it + 1) & it) == 0 or it + 1 == a.subgame_iters:
            ex = capi.exploitability2(d, f, full.get(0, capi.GET_AVERAGE), a.device)
            print("Iter=%8d exploitabilities=(%.3e, %.3e) sum=%.3e" % (it + 1, ex[0], ex[1], (ex[0] + ex[1]) / 2))
    full_strategy = full.get(0, capi.GET_AVERAGE)
    ex = capi.exploitability2(d, f, full_strategy, a.device)
    print(f"Full FP exploitability: {(ex[0] + ex[1]) / 2:.6f} ({ex[0]:.6f},{ex[1]:.6f})")
    if a.cfr:  # report_regrets + "\n" (:302-306): an empty line when no report is asked for
        line = ""
        if full_list:
            reg = capi.immediate_regrets(d, f, np.stack(full_list), a.device)
            line = regret_lines(reg, len(tree))
            if a.print_regret_summary:
                top = sum(reg[n].sum() for n in range(len(tree)) if tree[n][5] < a.mdp_depth)
                rest = sum(reg[n].sum() for n in range(len(tree)) if tree[n][5] >= a.mdp_depth)
                line += "\tRegrets (depth<=%d)/rest: %.6f/%.6f" % (a.mdp_depth, top, rest)
        print(line)
    results = [("net", a.net), ("full_tree", "%.6f" % ((ex[0] + ex[1]) / 2))]
    ev = capi.ev2(d, f, full_strategy, full_strategy, a.device)
    results_ev = [("net", a.net), ("full_tree", "%.6f" % ((ev[0] + ev[1]) / 2))]

    if a.net:
        assert a.mdp_depth > 0, "--mdp_depth is required with --net"
        print("##############################################\n##### Recursive solving                      #\n"
              "##############################################")
        eng = capi.Engine(d, f, capi.make_params(max_depth=a.mdp_depth, **base), max_lanes=a.max_lanes, device=a.device)
        if a.repeat_oracle_net:
            oracle_base = dict(base)
            if a.eval_oracle_values_iters > 0:
                oracle_base["num_iters"] = a.eval_oracle_values_iters
            oracle = OracleNet(capi, d, f, capi.make_params(max_depth=100000, **oracle_base), a.device, a.max_lanes * 64)
            eng.set_net_callback(oracle)
        else:
            load_net(eng, a.net)
        summed = reach_sum = None
        strategy_list = []

        def regret_report():  # report_regrets, recursive_eval.cc:28-53
            if not strategy_list or not (a.print_regret or a.print_regret_summary):
                return ""
            reg = capi.immediate_regrets(d, f, np.stack(strategy_list), a.device)
            out = regret_lines(reg, len(tree))
            if a.print_regret_summary:
                top = sum(reg[n].sum() for n in range(len(tree)) if tree[n][5] < a.mdp_depth)
                rest = sum(reg[n].sum() for n in range(len(tree)) if tree[n][5] >= a.mdp_depth)
                out += "\tRegrets (depth<=%d)/rest: %.6f/%.6f" % (a.mdp_depth, top, rest)
            return out

        for sid in range(max(a.num_repeats, 0)):
            s64 = eng.strategy_recursive_sampled(sid, a.root_only)
            if a.cfr:
                strategy_list.append(s64.astype(np.float32).astype(np.float64))  # tensor_to_tree_strategy of a float tensor
            s = s64.astype(np.float32)
            w = reach_of_actor(tree, s64, H).astype(np.float32)[:, :, None]  # from the fp64 strategy, then rounded (:143-152)
            summed = s * w if summed is None else summed + s * w
            reach_sum = w if reach_sum is None else reach_sum + w
            if ((sid + 1) & sid) == 0 or sid + 1 == a.num_repeats:
                final = (summed / (reach_sum + np.float32(1e-6))).astype(np.float64)
                ex = capi.exploitability2(d, f, final, a.device)
                ev = capi.ev2(d, f, full_strategy, final, a.device)  # compute_ev2(game, full_strategy, final_strategy), :370
                print("%5d: %.6f (%.6f,%.6f)\tEV of full: %.6f (%.6f,%.6f)" % (sid + 1, (ex[0] + ex[1]) / 2, ex[0], ex[1],
                                                                          (ev[0] + ev[1]) / 2, ev[0], ev[1]) + regret_report())
                tag = "repeated oracle toleaf" if a.repeat_oracle_net else "repeated toleaf"  # recursive_eval.cc:377-379
                results.append((f"{tag} {sid + 1}", "%.6f" % ((ex[0] + ex[1]) / 2)))
                results_ev.append((f"{tag} {sid + 1}", "%.6f" % ((ev[0] + ev[1]) / 2)))
    for name, val in results[1:]:
        print(f" {name} {val}")
    print("XXX " + json.dumps(dict(results), separators=(", ", ":")))
    print("YYY " + json.dumps(dict(results_ev), separators=(", ", ":")))


if __name__ == "__main__":
    main()
