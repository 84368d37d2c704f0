"""Test double for tests/test_bench_dist_plumbing.py: runs bench.py's main() -- argument parsing, rank environment, process
group, lane seeds, barrier + timed region, reduce_job, the JSON line -- on CPU ranks, with the GPU pieces replaced:
torch.cuda (availability / set_device / synchronize), the "nccl" backend (gloo instead) and rebel_amd.capi (an engine that
counts instead of computing).  Nothing here is reachable from the product or from bench.py itself."""
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

torch.cuda.is_available = lambda: True
torch.cuda.set_device = lambda d: None
torch.cuda.device_count = lambda: int(os.environ.get("BENCH_STUB_GPUS", 8))
torch.cuda.synchronize = lambda *a, **k: None
_init = dist.init_process_group


def _init_gloo(backend=None, **kw):
    assert backend == "nccl", "bench.py must ask for RCCL (backend 'nccl')"
    kw.pop("device_id", None)
    return _init(backend="gloo", **kw)


dist.init_process_group = _init_gloo

stub = types.ModuleType("rebel_amd.capi")
stub.NET_KERNEL_NAMES = {5: "stub net"}
stub.CFR_KERNEL_NAMES = {2: "stub cfr"}
stub.make_params = lambda **kw: types.SimpleNamespace(**kw)


class Engine:
    def __init__(self, dice, faces, params, max_lanes=1, device=0):
        self.params, self.lanes, self.device = params, max_lanes, device

    def set_net_mlp(self, *a):
        pass

    def set_net_precision(self, mode):
        pass

    def sync(self):
        pass

    def timing(self, stride):
        pass

    def stats(self, reset=False):
        return dict(cfr_ms=1.0, net_ms=1.0, cfr_launches=1, net_launches=1, net_rows=64, lane_steps=0, cfr_bytes=1e6,
                    net_flops=1e9, cfr_kernel=2, net_kernel=5, n_streams=1, net_products=3)

    def close(self):
        pass


class SelfPlay:
    def __init__(self, engine, seeds, random_action_prob=0.25, sample_leaf=True):
        self.e, self.seeds, self.games = engine, list(seeds), 0
        with open(os.environ["BENCH_STUB_SEEDS"] + f".{os.environ.get('RANK', 0)}", "w") as f:
            json.dump({"seeds": self.seeds, "device": engine.device}, f)

    def advance(self, collect=True):
        time.sleep(0.01 * (1 + int(os.environ.get("RANK", 0))))  # ranks differ: the job's time is the slowest rank's
        self.games += 3
        n = self.e.lanes
        return n * self.e.params.num_iters, np.arange(2 * n), None, None

    def games_finished(self):
        return self.games

    def on_device(self):
        return 1

    def close(self):
        pass


stub.Engine, stub.SelfPlay = Engine, SelfPlay
import rebel_amd  # noqa: E402

sys.modules["rebel_amd.capi"] = stub
rebel_amd.capi = stub

import bench  # noqa: E402

if __name__ == "__main__":
    bench.main()
