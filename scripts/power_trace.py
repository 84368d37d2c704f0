"""Socket power and shader clock while the hot kernels run (VERDICT r3 next #2b: "a power/clock trace of tile 5 vs tile 6").

A sampler thread reads the SMU's gpu_metrics through amdsmi every ~10 ms (average / current socket power, current gfx clock
of every XCD, hot-spot temperature) while the main thread keeps one workload's launches queued back to back for a few seconds:
  idle | net forward, register-resident kernel (RBL_MLP_TILE=5; POWER_TRACE_TILES="5 3" adds the feature-split kernel -- the
  tile-6 rows of profiles/r04_power_trace_tile5_vs_tile6.txt were taken at commit 6a4194a, before the software-pipelined
  kernel left the product) | CFR step kernel alone (synthetic net) | the self-play mix of bench.py (net -> cfr per iteration)
Prints one JSON object per phase: mean / p10 / p90 of power and clock over the phase's samples, us per launch (host clock
over the whole phase), and for the net phases the cycles per 64-row group and CU that follow from the measured clock.
usage: python scripts/power_trace.py [seconds per phase = 6] [rows = 589824]"""
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from rebel_amd import capi  # noqa: E402

SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
ROWS = int(sys.argv[2]) if len(sys.argv) > 2 else 589824


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.samples, self.stop, self.err = [], False, None
        try:
            import amdsmi

            amdsmi.amdsmi_init()
            self.smi, self.h = amdsmi, amdsmi.amdsmi_get_processor_handles()[0]
            self.read()  # fail here, not in the thread
        except Exception as ex:  # noqa: BLE001
            self.smi, self.err = None, repr(ex)

    def read(self):
        m = self.smi.amdsmi_get_gpu_metrics_info(self.h)
        clks = [c for c in (m.get("current_gfxclks") or []) if isinstance(c, (int, float)) and 0 < c < 10000]
        clk = float(np.mean(clks)) if clks else float(m.get("current_gfxclk") or 0)
        pw = m.get("current_socket_power")
        if not isinstance(pw, (int, float)) or pw <= 0 or pw > 5000:
            pw = m.get("average_socket_power")
        return time.perf_counter(), float(pw or 0), clk, float(m.get("temperature_hotspot") or 0), \
            float(max(clks)) if clks else clk, float(min(clks)) if clks else clk

    def run(self):
        while not self.stop and self.smi:
            try:
                self.samples.append(self.read())
            except Exception as ex:  # noqa: BLE001
                self.err = repr(ex)
                return
            time.sleep(0.008)

    def window(self, t0, t1):
        s = np.array([x for x in self.samples if t0 + 0.5 <= x[0] <= t1], float)  # skip the ramp of the first half second
        if len(s) == 0:
            return {"samples": 0}
        q = lambda c, p: float(np.percentile(s[:, c], p))  # noqa: E731
        return {"samples": len(s), "power_w": {"mean": float(s[:, 1].mean()), "p10": q(1, 10), "p90": q(1, 90)},
                "gfxclk_mhz": {"mean": float(s[:, 2].mean()), "p10": q(2, 10), "p90": q(2, 90),
                               "max_xcd_mean": float(s[:, 4].mean()), "min_xcd_mean": float(s[:, 5].mean())},
                "hotspot_c": float(s[:, 3].mean())}


def phase(name, sampler, body, extra=None):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < SECONDS:
        n += body()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    out = {"phase": name, "seconds": t1 - t0, "launches": n, "us_per_launch": (t1 - t0) / max(1, n) * 1e6}
    out.update(sampler.window(t0, t1))
    if extra:
        out.update(extra(out))
    print(json.dumps(out), flush=True)
    return out


def main():
    sampler = Sampler()
    if sampler.err:
        print(json.dumps({"sampler_error": sampler.err}), flush=True)
    sampler.start()
    try:
        caps = sampler.smi.amdsmi_get_power_cap_info(sampler.h) if sampler.smi else None
        print(json.dumps({"power_cap_info": {k: (v if isinstance(v, (int, float, str)) else str(v)) for k, v in (caps or {}).items()}}),
              flush=True)
    except Exception as ex:  # noqa: BLE001
        print(json.dumps({"power_cap_error": repr(ex)}), flush=True)
    phase("idle", sampler, lambda: (time.sleep(0.05), 0)[1])

    e = capi.Engine(1, 6, capi.make_params(num_iters=4, use_cfr=True))
    Q, H, hid = e.Q, e.H, 256
    rng = np.random.default_rng(7)
    layers = [(rng.uniform(-1, 1, (hid, Q)).astype(np.float32) / np.sqrt(Q), rng.uniform(-0.1, 0.1, hid).astype(np.float32)),
              (rng.uniform(-1, 1, (hid, hid)).astype(np.float32) / np.sqrt(hid), rng.uniform(-0.1, 0.1, hid).astype(np.float32))]
    ln = [(rng.uniform(0.5, 1.5, hid).astype(np.float32), rng.uniform(-0.2, 0.2, hid).astype(np.float32)) for _ in range(2)]
    w_out = rng.uniform(-1, 1, (H, hid)).astype(np.float32) / np.sqrt(hid)
    b_out = rng.uniform(-0.1, 0.1, H).astype(np.float32)
    q = np.zeros((ROWS, Q), np.float32)
    q[:, 0] = rng.integers(0, 2, ROWS)
    q[:, 1] = rng.integers(0, 2, ROWS)
    q[np.arange(ROWS), 2 + rng.integers(0, e.A, ROWS)] = 1
    q[:, 2 + e.A:2 + e.A + H] = rng.dirichlet(np.ones(H), ROWS)
    q[:, 2 + e.A + H:] = rng.dirichlet(np.ones(H), ROWS)
    qd = torch.from_numpy(q).cuda()
    od = torch.empty((ROWS, H), dtype=torch.float32, device="cuda")

    def net_body():
        for _ in range(20):
            capi._check(e.L.rbl_net_forward_dev(e.h, qd.data_ptr(), ROWS, od.data_ptr()))
        e.sync()
        return 20

    def net_extra(o):
        clk = o.get("gfxclk_mhz", {}).get("mean")
        groups_per_cu = ROWS / 64 / 256
        return {"ns_per_row": o["us_per_launch"] * 1e3 / ROWS,
                "cycles_per_64row_group_at_measured_clock": o["us_per_launch"] * clk / groups_per_cu if clk else None,
                "algorithmic_tflops": ROWS * 2 * (Q * 256 + 256 * 256 + 256 * H) / o["us_per_launch"] * 1e-6}

    for tile in os.environ.get("POWER_TRACE_TILES", "5").split():
        os.environ["RBL_MLP_TILE"] = tile
        e.set_net_mlp(layers, ln, w_out, b_out)
        net_body()
        phase(f"net forward, RBL_MLP_TILE={tile} ({ROWS} rows, " + ("canonical rows" if os.environ.get("RBL_QSPLIT") == "0" else "canonical rows through the split kernel") + ")", sampler, net_body, net_extra)
    os.environ["RBL_MLP_TILE"] = "5"
    e.close()
    if os.environ.get("POWER_TRACE_NET_ONLY"):
        sampler.stop = True
        return

    B = 16384
    ec = capi.Engine(1, 6, capi.make_params(num_iters=1024, max_depth=2, linear_update=True, use_cfr=True), max_lanes=B)
    ec.set_net_synthetic()
    ec.reset([-1] * B, [0] * B, np.full((B, 2, ec.H), 1.0 / ec.H))

    def cfr_body():
        ec.multistep(200)
        ec.sync()
        return 200

    phase("cfr_wave_kernel + synthetic elementwise net, 16 384 root lanes (per iteration)", sampler, cfr_body)
    ec.close()

    from rebel_amd.models import Net2, mlp_weights_from_state_dict

    torch.manual_seed(0)
    net = Net2(num_faces=6, num_dice=1, n_hidden=256, use_layer_norm=True, n_layers=2).eval()
    es = capi.Engine(1, 6, capi.make_params(num_iters=1024, max_depth=2, linear_update=True, use_cfr=True), max_lanes=B)
    es.set_net_mlp(*mlp_weights_from_state_dict(net.state_dict()))
    sp = capi.SelfPlay(es, list(range(B)), random_action_prob=0.25, sample_leaf=True)
    for _ in range(3):
        sp.advance(collect=False)

    def mix_body():
        sp.advance(collect=False)
        return 1024

    phase("self-play mix of bench.py: net -> cfr per iteration, 16 384 lanes (per iteration)", sampler, mix_body)
    sampler.stop = True


if __name__ == "__main__":
    main()
