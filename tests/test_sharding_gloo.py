"""N>1 path on CPU: two `gloo` ranks run bench.py's rank bookkeeping (disjoint lane seeds per rank, max-over-ranks
time, whole-job sum of units).  There is no data-path collective to test -- lanes are independent -- so this pins the
sharding arithmetic the 8-GPU driver run relies on."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, json
    import torch, torch.distributed as dist
    sys.path.insert(0, %r)
    from rebel_amd.sharding import lane_seeds, reduce_job
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    seeds = lane_seeds(rank, 16)
    dt, units, games = reduce_job(dist, world, dt=1.0 + rank, units=1000.0 * (rank + 1), games=3.0, device="cpu")
    all_seeds = [None] * world
    dist.all_gather_object(all_seeds, seeds)
    if rank == 0:
        print(json.dumps({"dt": dt, "units": units, "games": games, "seeds": all_seeds}))
    dist.barrier()
    dist.destroy_process_group()
""") % ROOT


def test_two_rank_sharding(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2", OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    import json

    res = json.loads([l for l in outs[0][0].splitlines() if l.startswith("{")][-1])
    assert res["dt"] == 2.0          # MAX over ranks
    assert res["units"] == 3000.0    # whole-job sum
    assert res["games"] == 6.0
    a, b = res["seeds"]
    assert len(a) == len(b) == 16 and not set(a) & set(b)  # disjoint lane sets, no collective on the data path
