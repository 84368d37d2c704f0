"""bench.py's multi-rank path on CPU: two gloo ranks run bench.main() end to end -- `--gpus 2`,
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*, process-group set-up, per-rank lane seeds and device, barrier-bracketed timed
region, reduce_job (MAX time, SUM work), rank 0's JSON line -- against a counting engine (tests/bench_stub_driver.py).
An 8-GPU node runs exactly this code with RCCL and the real engine."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_on_gloo(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    lanes, iters, steps = 64, 8, 3
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2", OMP_NUM_THREADS="1",
               BENCH_STUB_SEEDS=str(tmp_path / "seeds"))
    cmd = [sys.executable, os.path.join(ROOT, "tests", "bench_stub_driver.py"), "--gpus", "2", "--steps", str(steps), "--warmup",
           "1", "--lanes", str(lanes), "--iters", str(iters), "--no-cpu-baseline"]
    procs = [subprocess.Popen(cmd, env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True, cwd=ROOT) for r in range(2)]
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    lines0 = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    lines1 = [l for l in outs[1][0].splitlines() if l.startswith("{")]
    assert len(lines0) == 1 and not lines1  # exactly one JSON line, from rank 0
    res = json.loads(lines0[0])
    assert res["n_gpus"] == 2 and res["steps"] == steps and res["warmup"] == 1 and res["scaling"] == "weak"
    assert res["config"]["lanes_per_gpu"] == lanes and res["config"]["parallelism"] == "independent lane sets x2"
    assert len(res["config"]["workload"]) <= 120 and f"{lanes} lanes/GPU" in res["config"]["workload"]
    # whole-job work over the SLOWEST rank's time: units = ranks x lanes x iters x steps; rank 1 sleeps twice as long
    units = 2 * lanes * iters * steps
    assert abs(res["value"] * res["ms_per_step"] * 1e-3 * steps - units) < 1e-6 * units
    assert res["ms_per_step"] >= 20.0  # rank 1's 0.02 s per step, not rank 0's 0.01 s
    assert abs(res["games_per_s"] * res["ms_per_step"] * 1e-3 * steps - 2 * 3 * steps) < 1e-6
    assert "cpu_baseline" not in res and "lanes_4096" not in res  # single-rank legs are skipped in a multi-rank job
    seeds = [json.load(open(str(tmp_path / "seeds") + f".{r}")) for r in range(2)]
    assert seeds[0]["seeds"] == list(range(lanes)) and seeds[1]["seeds"] == list(range(lanes, 2 * lanes))
    assert [s_["device"] for s_ in seeds] == [0, 1]  # one engine per LOCAL_RANK
