"""The committed evidence under profiles/ is self-consistent and is what bench.py's readers expect (CPU only): the bench line
of the round, the rocprofv3 kernel-trace summary it quotes in `rocprof`, the PMC traffic it quotes in `traffic`.  Guards the
file formats (kernel names with commas, quoted CSV) and the arithmetic a reader would redo."""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _newest_bench():
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_bench.json")))
    assert paths, "no profiles/rNN_bench.json"
    return json.load(open(paths[-1])), os.path.basename(paths[-1])


def test_bench_line_arithmetic():
    d, name = _newest_bench()
    lanes, iters = d["config"]["lanes_per_gpu"], d["config"]["subgame_iters"]
    units = lanes * iters * d["steps"] * d["n_gpus"]
    assert abs(d["value"] - units / (d["ms_per_step"] * 1e-3 * d["steps"])) < 1e-6 * d["value"], name
    for key in ("roofline", "roofline_cfr"):
        r = d[key]
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
        work = r.get("algorithmic_flops_per_launch", r.get("algorithmic_bytes_per_launch"))
        scale = 1e12 if key == "roofline" else 1e9
        assert abs(r["achieved"] - work / scale / (r["avg_launch_us"] * 1e-6)) < 1e-6 * r["achieved"]
    # the two kernels' dispatch intervals of an iteration fit into the iteration
    per_iter_us = d["roofline"]["avg_launch_us"] + d["roofline_cfr"]["avg_launch_us"]
    assert per_iter_us * iters * 1e-3 <= d["ms_per_step"] * 1.001
    assert d["cpu_baseline"]["kind"] == "reference" and d["cpu_baseline"]["value"] > 0
    if "configs" in d:  # round 4 on: the other BASELINE configurations ride in the same line
        ids = [c["baseline_config"] for c in d["configs"]]
        assert ids == sorted(ids) and {1, 3, 4} <= set(ids) <= {0, 1, 3, 4}, ids  # round 5 adds configs[0]
        for c in d["configs"]:
            # configs 1, 3, 4 carry their kernel fractions and the reference's CPU rate; only config 0 (added in round 5: the
            # reference's own plumbing case) may lack them
            if c["baseline_config"] != 0 or "net" in c:
                assert 0 < c["net"]["frac"] < 1 and 0 < c["cfr"]["frac"] < 1
            if c["baseline_config"] != 0:
                assert c["cpu_reference"] is not None
            assert c["value"] > 0 and (c.get("cpu_reference") is None or c["cpu_reference"]["value"] > 0)


def test_bench_line_quotes_the_committed_profiles():
    d, name = _newest_bench()
    sys.path.insert(0, ROOT)
    import bench

    for key, kern in (("roofline", ("mlp_resident_kernel",)), ("roofline_cfr", ("cfr_wave_kernel",))):
        r = d[key]
        # round 6 on: what the line quotes from committed profiles sits in ONE sub-object, apart from the live figures, and the
        # live `traffic` is null; rounds <= 5 had the same fields beside `frac`
        fcp = r.get("from_committed_profile", r)
        if "from_committed_profile" in r:
            assert r["traffic"] is None and "rocprof" not in r and "traffic_detail" not in r and "NOT measured in this run" in fcp["note"]
        if "traffic_detail" in fcp:
            src = os.path.join(ROOT, fcp["traffic_detail"]["source"])
            assert os.path.exists(src), src
            k = json.load(open(src))["kernels"][fcp["traffic_detail"]["kernel"]]
            rd = k["fetch_size_bytes_per_launch"]["timed_epochs"]["mean"]
            wr = k["write_size_bytes_per_launch"]["timed_epochs"]["mean"]
            assert abs(fcp["traffic"] - (rd + wr)) < 1e-6 * fcp["traffic"]
        if "rocprof" in fcp:
            src = os.path.join(ROOT, fcp["rocprof"]["source"])
            assert os.path.exists(src), src
            want_us = fcp["rocprof"]["avg_launch_us"]
            assert any(abs(float(line.rsplit(",", 5)[4]) / 1e3 - want_us) < 1e-6 and fcp["rocprof"]["kernel"] in line
                       for line in open(src).read().splitlines()[1:])
            work = r.get("algorithmic_flops_per_launch", r.get("algorithmic_bytes_per_launch")) / (1e12 if key == "roofline" else 1e9)
            assert abs(fcp["rocprof"]["frac"] - work / (want_us * 1e-6) / r["peak"]) < 1e-9
        # the readers find the newest committed summaries of the driver's command shape
        # (the steps / warm-up shape is whatever the newest committed PMC summary says it profiled: no literals here)
        shape = tuple(json.load(open(os.path.join(ROOT, bench.pmc_traffic(kern)["source"])))["steps_warmup"])
        got = bench.rocprof_timed_epochs(kern, shape)
        assert got and any(k in got["kernel"] for k in kern) and got["avg_launch_us"] > 0
        assert bench.pmc_traffic(kern)["bytes"] > 0
