"""GPU checks of bench.py itself: the JSON contract of the default (single-process) path on a small workload, and the
`--force-dist` path, which initialises the RCCL process group (backend "nccl") for one rank and runs the barrier /
all-reduce bookkeeping the driver's N = 2, 4, 8 launches rely on -- so that path is known-good before an 8-GPU node
runs it."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env=None):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--lanes", "2560", "--iters", "48",
           "--no-cpu-baseline"] + extra
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def test_bench_json_contract_small():
    d = _run([])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "roofline_cfr"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["lanes_per_gpu"] == 2560 and d["selfplay_walk"] == "device kernels"
    assert d["streams"] == 2  # what the engine ran (two lane parts from 2 048 lanes on), reported by rbl_engine_stats
    assert "cfr_wave_kernel" in d["roofline_cfr"]["kernel"] and "mlp_resident_kernel" in d["roofline"]["kernel"]
    for key in ("roofline", "roofline_cfr"):
        r = d[key]
        assert r["achieved"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
        assert r["standalone"]["achieved"] > 0  # the stand-alone leg ran
    assert abs(d["value"] - 2 * 2560 * 48 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]


def test_bench_force_dist_single_rank_rccl():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    d = _run(["--force-dist", "--no-extra-legs"], env)
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["scaling"] == "weak"
