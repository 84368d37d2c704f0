"""GPU parity for the evaluation row: best response / exploitability sweeps on the device against the
oracle (BRSolver::compute_br, compute_exploitability2; subgame_solving.cc:316-358, 802-816) -- bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("d,f,iters", [(1, 4, 128), (1, 5, 40), (2, 2, 60), (1, 6, 16)])
def test_exploitability2_bit_exact(d, f, iters, port):
    from oracle import orc
    from rebel_amd import capi

    p = orc.make_params(num_iters=iters, max_depth=100, linear_update=True, use_cfr=True)
    s = port.solver(d, f, p, net=orc.NET_NONE)
    s.multistep()
    for which in (orc.GET_AVERAGE, orc.GET_LAST):
        strat = s.get(which)
        assert np.array_equal(capi.exploitability2(d, f, strat), port.exploitability2(d, f, strat))


def test_exploitability_of_gpu_solved_strategy_converges():
    """subgame_solving_test.cc:162-179: linear CFR on 1 die x 2 faces, 180 iterations -> exploitability in [0, 1e-3),
    with solve AND evaluation on the device."""
    from rebel_amd import capi

    e = capi.Engine(1, 2, capi.make_params(num_iters=180, max_depth=100, linear_update=True, use_cfr=True))
    e.reset([-1], [0], np.full((1, 2, e.H), 1.0 / e.H))
    e.multistep()
    expl = capi.exploitability2(1, 2, e.get(0, capi.GET_AVERAGE))
    total = (expl[0] + expl[1]) / 2
    assert 0 <= total < 1e-3, expl


def test_best_response_on_depth_limited_lanes(port):
    """BR with net-valued pseudo-leaves, several lanes at once.  FP::step(t) IS compute_br(t, average strategy)
    (subgame_solving.cc:433-476), so the oracle's first fictitious-play step yields the reference BR root values
    against the uniform strategy a freshly reset lane holds."""
    from oracle import orc
    from rebel_amd import capi

    kw = dict(num_iters=30, max_depth=2, linear_update=True, use_cfr=True)
    e = capi.Engine(1, 6, capi.make_params(**kw), max_lanes=3)
    e.set_net_synthetic()
    rng = np.random.default_rng(3)
    roots, players = [-1, 4, 9], [0, 1, 0]
    beliefs = rng.dirichlet(np.ones(e.H), size=(3, 2))
    e.reset(roots, players, beliefs)  # sigma = uniform after reset
    for t in (0, 1):
        got = e.best_response(t)
        for b in range(3):
            o = port.solver(1, 6, orc.make_params(num_iters=2, max_depth=2, use_cfr=False), roots[b], players[b],
                            beliefs[b], orc.NET_SYNTHETIC)
            o.step(t)
            assert np.array_equal(got[b], o.hand_values(t)), (t, b)


@pytest.mark.parametrize("to_leaf", [False, True])
@pytest.mark.parametrize("d,f,depth,iters", [(1, 4, 2, 32), (1, 4, 3, 10), (1, 5, 2, 12), (2, 2, 2, 20)])
def test_recursive_strategy_bit_exact(d, f, depth, iters, to_leaf, port):
    """compute_strategy_recursive(_to_leaf) level-batched on the device == the oracle's depth-first recursion."""
    from oracle import orc
    from rebel_amd import capi

    kw = dict(num_iters=iters, max_depth=depth, linear_update=True, use_cfr=True)
    e = capi.Engine(d, f, capi.make_params(**kw), max_lanes=64)  # smaller than the widest level: exercises chunking
    e.set_net_synthetic()
    got = e.strategy_recursive(to_leaf=to_leaf)
    want = port.strategy_recursive(d, f, orc.make_params(**kw), to_leaf=to_leaf, net=orc.NET_SYNTHETIC)
    assert got.shape == want.shape and np.array_equal(got, want)
    assert np.array_equal(capi.exploitability2(d, f, got), port.exploitability2(d, f, want))


@pytest.mark.parametrize("d,f,depth,iters,use_cfr,lanes", [
    (1, 4, 2, 32, True, 64), (1, 4, 1, 10, True, 7), (1, 5, 2, 12, True, 48), (2, 2, 2, 20, True, 64), (1, 6, 2, 6, True, 512),
    (1, 4, 2, 1, True, 64), (1, 4, 2, 16, False, 64), (1, 4, 3, 8, True, 32), (2, 3, 2, 8, True, 512)])
def test_streaming_exploitability_bit_exact(d, f, depth, iters, use_cfr, lanes, port):
    """rbl_exploitability_recursive (eval_stream.hip): the to-leaf recursion + compute_exploitability2 with
    the full-tree strategy device-resident and edge-indexed == the oracle's depth-first recursion + dense BR sweep, and ==
    the engine's own dense path; sharded by root action and recombined on the host it is the same number again."""
    from oracle import orc
    from rebel_amd import capi

    kw = dict(num_iters=iters, max_depth=depth, linear_update=True, use_cfr=use_cfr)
    e = capi.Engine(d, f, capi.make_params(**kw), max_lanes=lanes)
    e.set_net_synthetic()
    got, top, stats = e.exploitability_recursive()
    want_strategy = port.strategy_recursive(d, f, orc.make_params(**kw), to_leaf=True, net=orc.NET_SYNTHETIC)
    want = port.exploitability2(d, f, want_strategy)
    assert np.array_equal(got, want), (got, want)
    assert np.array_equal(got, capi.exploitability2(d, f, e.strategy_recursive(to_leaf=True)))
    assert stats["nodes"] == want_strategy.shape[0] and stats["subgames"] >= 1
    assert stats["strategy_bytes"] == 8 * e.H * (stats["nodes"] - 1)
    assert np.array_equal(capi.combine_exploitability(d, f, depth, [top]), got)
    for n_shards in (2, 3, 8):
        parts = [e.exploitability_recursive(s, n_shards) for s in range(n_shards)]
        assert all(np.isnan(p[0]).all() for p in parts)
        assert all(np.array_equal(p[1][1], top[1] if n_shards == 1 else parts[0][1][1]) for p in parts)  # one owner map
        assert sum(p[2]["subgames"] for p in parts) == stats["subgames"] + (n_shards - 1)  # every shard solves the root
        assert np.array_equal(capi.combine_exploitability(d, f, depth, [p[1] for p in parts]), got)
    # dealing the frontier of the SECOND recursion level: every shard redundantly solves the root subgame
    # and its pseudo-leaves' subgames, the rest is shared out in pieces of at most 1/16 of the game; same number, bit for bit
    one = e.exploitability_recursive(0, 1, deal_levels=2)
    assert np.array_equal(one[0], got) and np.array_equal(capi.combine_exploitability(d, f, depth, [one[1]], deal_levels=2), got)
    redundant = None
    for n_shards in (2, 3, 8):
        parts = [e.exploitability_recursive(s, n_shards, deal_levels=2) for s in range(n_shards)]
        assert all(np.array_equal(p[1][1], parts[0][1][1]) for p in parts)
        assert np.array_equal(capi.combine_exploitability(d, f, depth, [p[1] for p in parts], deal_levels=2), got), n_shards
        # the levels above the dealt one are solved by every shard, everything below by exactly one
        extra = sum(p[2]["subgames"] for p in parts) - stats["subgames"]
        assert extra % (n_shards - 1) == 0
        redundant = extra // (n_shards - 1) if redundant is None else redundant
        assert extra // (n_shards - 1) == redundant and 1 <= redundant <= stats["subgames"]


@pytest.mark.parametrize("d,f,depth,iters,seed,root_only,use_cfr", [
    (1, 4, 2, 32, 0, False, True), (1, 4, 2, 33, 5, False, True), (1, 5, 2, 16, 1, False, True), (1, 4, 1, 12, 2, True, True),
    (1, 4, 2, 24, 3, True, True), (2, 2, 2, 20, 4, False, True), (1, 4, 2, 16, 6, False, False)])
def test_sampled_recursive_strategy_bit_exact(d, f, depth, iters, seed, root_only, use_cfr, port):
    """compute_sampled_strategy_recursive_to_leaf (recursive_solving.cc:301-327, the core of recursive_eval): per-subgame
    act iterations drawn up front in the reference's order, level-batched lanes snapshotting at their own iteration."""
    from oracle import orc
    from rebel_amd import capi

    kw = dict(num_iters=iters, max_depth=depth, linear_update=True, use_cfr=use_cfr)
    e = capi.Engine(d, f, capi.make_params(**kw), max_lanes=48)
    e.set_net_synthetic()
    got = e.strategy_recursive_sampled(seed, root_only)
    want = port.strategy_recursive_sampled(d, f, orc.make_params(**kw), seed, root_only, net=orc.NET_SYNTHETIC)
    assert got.shape == want.shape and np.array_equal(got, want)
    # a different seed stops the subgames elsewhere
    assert not np.array_equal(got, e.strategy_recursive_sampled(seed + 1, root_only))


@pytest.mark.parametrize("d,f", [(1, 4), (1, 5), (2, 2)])
def test_ev2_bit_exact(d, f, port):
    """compute_ev2 (subgame_solving.cc:931-982; the 'EV of full' numbers of recursive_eval): policy evaluation as a mode of
    the CFR kernel, one strategy against another, bit-exact against the oracle."""
    from oracle import orc
    from rebel_amd import capi

    s1 = port.solver(d, f, orc.make_params(num_iters=24, max_depth=100000, linear_update=True, use_cfr=True))
    s1.multistep()
    s2 = port.solver(d, f, orc.make_params(num_iters=7, max_depth=100000, linear_update=False, use_cfr=True))
    s2.multistep()
    a, b = s1.get(orc.GET_AVERAGE), s2.get(orc.GET_LAST)
    assert np.array_equal(capi.ev2(d, f, a, b), port.ev2(d, f, a, b))
    assert np.array_equal(capi.ev2(d, f, b, a), port.ev2(d, f, b, a))
    # a strategy against itself: zero-sum
    e = capi.ev2(d, f, a, a)
    assert abs(e[0] + e[1]) < 1e-12  # out[1] is minus the same expectation


def test_recursive_eval_tool(tmp_path):
    """scripts/recursive_eval.py (the reference's recursive_eval CLI on the C ABI): runs, prints the XXX json line the
    reference's eval_all.py parses; with a subgame depth that covers the whole game the sampled strategies are last
    strategies of the full-tree solver at random even iterations, and their reach-weighted average must beat a single one."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "recursive_eval.py"), "--num_dice", "1", "--num_faces",
                        "4", "--subgame_iters", "64", "--mdp_depth", "100", "--num_repeats", "8", "--net", "zero", "--cfr",
                        "--print_regret_summary"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("XXX ")][-1]
    d = json.loads(line[4:])
    assert d["net"] == "zero" and float(d["full_tree"]) < 0.05
    assert float(d["repeated toleaf 8"]) < float(d["repeated toleaf 1"])
    assert "Iter=      64" in r.stdout
    # report_regrets after the full-tree solve (recursive_eval.cc:302-306) and at 1, 2, 4, 8 repeats (:374-377)
    assert r.stdout.count("Regrets (depth<=100)/rest: ") == 5
    ev = json.loads([l for l in r.stdout.splitlines() if l.startswith("YYY ")][-1][4:])
    assert set(ev) == set(d) and abs(float(ev["full_tree"])) < 1e-6  # EV of the full-tree strategy against itself


@pytest.mark.parametrize("d,f,iters,variant", [(1, 4, 64, "linear"), (1, 5, 33, "plain"), (2, 2, 40, "linear"), (1, 6, 17, "dcfr"),
                                               (1, 4, 1, "linear")])
def test_stream_solver_bit_exact(d, f, iters, variant, port):
    """rbl_stream_*: full-tree CFR as level-synchronous sweeps over edge-indexed arrays (the reference tool's "Solving the
    game for the full tree", recursive_eval.cc:269-296) == the oracle's full-depth CFR solver: regrets, sum / last / average
    strategies and the exploitability trace, bit for bit."""
    from oracle import orc
    from rebel_amd import capi

    kw = dict(num_iters=iters, max_depth=100000, use_cfr=True, linear_update=variant == "linear")
    if variant == "dcfr":
        kw.update(dcfr=True, dcfr_alpha=1.5, dcfr_beta=0.5, dcfr_gamma=2.0)
    s = capi.StreamSolver(d, f, capi.make_params(**kw))
    o = port.solver(d, f, orc.make_params(**kw), net=orc.NET_NONE)
    assert s.nodes == o.N
    for it in range(iters):
        s.step(1)
        o.step(it % 2)
        if ((it + 1) & it) == 0 or it + 1 == iters:
            assert np.array_equal(s.exploitability(), port.exploitability2(d, f, o.get(orc.GET_AVERAGE))), it
    for ours, theirs in ((capi.GET_REGRETS, orc.GET_REGRETS), (capi.GET_SUM, orc.GET_SUM), (capi.GET_LAST, orc.GET_LAST),
                         (capi.GET_AVERAGE, orc.GET_AVERAGE)):
        assert np.array_equal(s.get(ours), o.get(theirs)), ours


@pytest.mark.parametrize("root_only", [False, True])
@pytest.mark.parametrize("d,f,iters,depth,repeats", [(1, 4, 32, 2, 3), (1, 5, 17, 1, 2), (2, 2, 24, 3, 2), (1, 6, 20, 2, 2)])
def test_stream_sampled_repeats_bit_exact(d, f, iters, depth, repeats, root_only):
    """rbl_stream_sampled_*: the tool's "Recursive solving" section on edge-indexed device arrays.  Every repeat equals
    rbl_strategy_recursive_sampled (itself pinned to the oracle in test_recursive_parity.py), the float32 reach-weighted
    mean equals the reference's tensor arithmetic (recursive_eval.cc:136-160, 343-353) restated in numpy, exploitability and
    compute_ev2 equal the dense kernels' -- all bit for bit."""
    import importlib.util
    import os

    from rebel_amd import capi

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("recursive_eval_tool", os.path.join(root, "scripts", "recursive_eval.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)

    base = dict(num_iters=iters, use_cfr=True, linear_update=True)
    s = capi.StreamSolver(d, f, capi.make_params(max_depth=100000, **base))
    s.step(iters)
    full = s.get(capi.GET_AVERAGE)
    eng = capi.Engine(d, f, capi.make_params(max_depth=depth, **base), max_lanes=64)
    eng.set_net_zero()
    tree = capi.unroll_tree(d, f, -1, 0, 1000000)
    summed = reach = None
    for seed in range(repeats):
        s.sampled_add(eng, seed, root_only=root_only)  # root_only: the subgames below the root as a forest on the stream arrays
        want = eng.strategy_recursive_sampled(seed, root_only)
        assert np.array_equal(s.get(capi.GET_SAMPLED), want), seed
        w = tool.reach_of_actor(tree, want, s.H).astype(np.float32)[:, :, None]
        s32 = want.astype(np.float32)
        summed = s32 * w if summed is None else summed + s32 * w
        reach = w if reach is None else reach + w
        final = (summed / (reach + np.float32(1e-6))).astype(np.float64)
        assert np.array_equal(s.get(capi.GET_FINAL), final), seed
    ex, ev = s.sampled_eval()
    assert np.array_equal(ex, capi.exploitability2(d, f, final))
    assert np.array_equal(ev, capi.ev2(d, f, full, final))
    s.sampled_reset()
    s.sampled_add(eng, 1, root_only=root_only)
    assert np.array_equal(s.get(capi.GET_SAMPLED), eng.strategy_recursive_sampled(1, root_only))


def test_recursive_eval_tool_stream_mode_equals_dense_mode():
    """scripts/recursive_eval.py --stream (every full-tree array edge-indexed on the device) prints the same trace, XXX and
    YYY lines as the dense mode, which is pinned to the reference binary below."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args = ["--num_dice", "1", "--num_faces", "4", "--subgame_iters", "32", "--mdp_depth", "2", "--num_repeats", "4", "--net", "zero",
            "--cfr"]
    outs = []
    for extra in ([], ["--stream"]):
        r = subprocess.run([sys.executable, os.path.join(root, "scripts", "recursive_eval.py")] + args + extra, cwd=root,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:]
        outs.append([l for l in r.stdout.splitlines() if l.startswith(("Iter=", "XXX ", "YYY ", "Full FP")) or l[:5].strip().isdigit() and l[5:6] == ":"])
    assert outs[0] == outs[1] and len(outs[0]) >= 6 + 1 + 3 + 2, outs


def test_recursive_eval_tool_vs_reference_binary():
    """scripts/recursive_eval.py against the UNMODIFIED reference tool (oracle/_ref/recursive_eval = recursive_eval.cc built by
    oracle/Makefile; golden stdout in tests/golden/recursive_eval_1d4f.json, made by make_recursive_eval_golden.py): the
    XXX / YYY lines scripts/eval_all.py:100-104 parses.  Zero net (pure CFR, no value net anywhere): identical strings.
    TorchScript Net2 with O(0.3) outputs, 4 sampled repeats of 32 iterations: the 6-decimal numbers to 2e-5 (measured 7e-6; the MFMA
    forward and torch's differ by ~1e-7 per call, which CFR amplifies: DESIGN.md section 5, P3)."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    golden = json.load(open(os.path.join(root, "tests", "golden", "recursive_eval_1d4f.json")))
    # round 6: + discounted CFR (--dcfr) and the oracle-net mode (--repeat_oracle_net [--eval_oracle_values_iters]): no value net
    # anywhere, so the strings are identical
    for case, tol in (("zero", 0.0), ("zero_dcfr", 0.0), ("oracle_1d3f", 0.0), ("oracle_1d3f_iters8", 0.0), ("oracle_1d4f", 0.0),
                      ("net", 2e-5)):  # measured on MI355X: max 7e-6
        g = golden[case]
        out = subprocess.run([sys.executable, os.path.join(root, "scripts", "recursive_eval.py")] + g["args"], cwd=root,
                             stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        assert out.returncode == 0, out.stdout[-2000:]
        lines = out.stdout.splitlines()
        for tag in ("XXX", "YYY"):
            got = json.loads([l for l in lines if l.startswith(tag + " ")][0][4:])
            want = g[tag.lower()]
            assert list(got) == list(want), (got, want)
            for k in want:
                if k == "net":
                    continue
                if tol == 0.0:
                    assert got[k] == want[k], (case, tag, k, got[k], want[k])
                else:
                    print(f"[recursive_eval vs reference] {case} {tag} {k}: ours {got[k]} reference {want[k]}")
                    assert abs(float(got[k]) - float(want[k])) <= tol, (case, tag, k, got[k], want[k])
        if case in ("zero", "zero_dcfr"):  # the exploitability trace of the full-tree solve, line for line
            want_iter = [l for l in g["stdout"] if l.startswith("Iter=") or l.startswith("Full FP")]
            got_iter = [l for l in lines if l.startswith("Iter=") or l.startswith("Full FP")]
            assert got_iter == want_iter


@pytest.mark.parametrize("stream", [False, True])
def test_recursive_eval_regret_reports_vs_reference_binary(stream):
    """--print_regret / --print_regret_summary (report_regrets, recursive_eval.cc:28-53 over the sampling strategies of the
    full-tree solve, :285-306): the two report lines of scripts/recursive_eval.py -- dense mode (rbl_immediate_regrets on the
    list of strategies) and --stream mode (regrets accumulated on the device, rbl_stream_regrets_*: the only form that exists
    at 2 dice x 6 faces) -- equal the UNMODIFIED reference tool's, character for character (golden made by
    tests/golden/make_recursive_eval_golden.py --only-regrets from oracle/_ref/recursive_eval)."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    g = json.load(open(os.path.join(root, "tests", "golden", "recursive_eval_1d4f.json")))["zero_regrets"]
    out = subprocess.run([sys.executable, os.path.join(root, "scripts", "recursive_eval.py")] + g["args"] + (["--stream"] if stream else []),
                         cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:]
    want = [l for l in g["stdout"] if l.startswith(("Iter=", "Full FP", "\tRegrets"))]
    got = [l for l in out.stdout.splitlines() if l.startswith(("Iter=", "Full FP", "\tRegrets"))]
    assert len(want) == 10 and got == want, (got[-2:], want[-2:])


def test_stream_regret_reports_equal_immediate_regrets():
    """rbl_stream_regrets_* on sampled repeats (float-rounded strategies, recursive_eval.cc:357-358) == rbl_immediate_regrets on
    the same list, bit for bit; root_only repeats included."""
    from rebel_amd import capi

    d, f, iters, depth = 1, 5, 12, 2
    base = dict(num_iters=iters, use_cfr=True, linear_update=True)
    s = capi.StreamSolver(d, f, capi.make_params(max_depth=100000, **base))
    eng = capi.Engine(d, f, capi.make_params(max_depth=depth, **base), max_lanes=64)
    eng.set_net_synthetic()
    tree = capi.unroll_tree(d, f, -1, 0, 1000000)
    for root_only in (False, True):
        s.sampled_reset()
        s.regrets_reset()
        lst = []
        for seed in range(3):
            s.sampled_add(eng, seed, root_only=root_only)
            s.regrets_add(capi.GET_SAMPLED)
            lst.append(s.get(capi.GET_SAMPLED).astype(np.float32).astype(np.float64))
            want = capi.immediate_regrets(d, f, np.stack(lst))
            first, sums = s.regrets_report(depth, n_first=len(tree))
            assert np.array_equal(first, want), (root_only, seed)
            top = sum(want[n].sum() for n in range(len(tree)) if tree[n][5] < depth)
            rest = sum(want[n].sum() for n in range(len(tree)) if tree[n][5] >= depth)
            assert sums == (top, rest)


def test_streaming_exploitability_2d6f_full_tree():
    """BASELINE configs[4] at full size: 2 dice x 6 faces, 33 554 431 nodes, strategy edge-indexed on the device (9.7 GB);
    two iterations per subgame keep it a few seconds.  Sanity of the numbers + two shards recombined == unsharded."""
    from rebel_amd import capi

    e = capi.Engine(2, 6, capi.make_params(num_iters=2, max_depth=2, linear_update=True, use_cfr=True), max_lanes=8192)
    e.set_net_synthetic()
    got, top, stats = e.exploitability_recursive()
    assert stats["nodes"] == 2 ** 25 - 1 and stats["subgames"] == 2 ** 23 and stats["levels"] == 13
    assert np.isfinite(got).all() and -1.0 <= got.min() and got.max() <= 1.0 and got.sum() >= 0
    parts = [e.exploitability_recursive(s, 2) for s in range(2)]
    assert np.array_equal(capi.combine_exploitability(2, 6, 2, [p[1] for p in parts]), got)


def test_immediate_regrets_bit_exact(port):
    """rbl_immediate_regrets (compute_immediate_regrets, subgame_solving.cc:984-1050; recursive_eval's --print_regret
    reports): plain-CFR regret updates of a full-tree solver on the device, against the oracle (pinned to the compiled
    reference in tests/test_oracle_pin.py)."""
    from oracle import orc
    from rebel_amd import capi

    for d, f, k in ((1, 3, 3), (1, 4, 4), (2, 2, 2)):
        S = np.stack([port.strategy_recursive(d, f, orc.make_params(num_iters=12 + 9 * i, max_depth=2, linear_update=True,
                                                                       use_cfr=True), to_leaf=True) for i in range(k)])
        got, want = capi.immediate_regrets(d, f, S), port.immediate_regrets(d, f, S)
        assert np.array_equal(got, want), (d, f, np.abs(got - want).max())
        assert np.abs(want).max() > 0.01
