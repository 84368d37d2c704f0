#!/usr/bin/env python3
"""bench.py -- the headline benchmark: self-play data generation for ReBeL on Liar's Dice, MI355X engine.

    python bench.py --gpus N --steps K --warmup W

`--gpus N` is honoured either way it is launched: under `python -m torch.distributed.run --nproc-per-node N ...` (the rank
variables are in the environment; WORLD_SIZE must equal N) or bare -- then bench.py starts the N ranks itself (re-executing
under torch.distributed.run on 127.0.0.1) and fails loudly when fewer than N GPUs are visible.

Metric (BASELINE.json): subgame CFR iterations/sec, whole job, 1 die x 6 faces @ 1024 iterations per subgame.
A "step" is one pass of the hot path over one batch: EVERY lane plays one subgame of its self-play game end to end --
RlRunner::step's draws and solver construction (device kernels), 1024 x [batched value-net forward on MFMA + CFR step
kernel], sigma snapshot at the lane's act_iteration, the sampling walk to the next public state and the Bayes updates
(device kernels, libstdc++ <random> restated draw for draw), 2 training examples per lane handed to the example sink
(the only read-back).  One step = lanes x 1024 subgame-CFR-iterations.  Work is sharded across GPUs as independent
lane sets (seeds rank*lanes+i), no data-path collective: "scaling": "weak".

The JSON line also carries
  roofline      dominant kernel (the fused MLP forward; MFMA-bound): algorithmic FLOP per launch / mean launch duration,
                measured live over the timed region on every 7th iteration (an ODD stride: both traversers are sampled).
                Durations are the kernels' own begin -> end intervals: the two HIP events are bound to the dispatch packet
                (hipExtLaunchKernelGGL; rebel_amd/csrc/launch_timing.h), i.e. the timestamps rocprofv3 --kernel-trace
                reports, without the launch gap a recorded event pair includes.  At the default 16 384 lanes the engine runs
                ONE stream -- net(all lanes) -> cfr(all lanes) per iteration -- so a kernel has the GPU to itself; below
                16 384 lanes it interleaves two half-batches on two streams ("in-mix" durations)
  roofline_cfr  the CFR step kernel (HBM-bound): algorithmic bytes per launch / mean launch duration
  traffic       HBM-side bytes per launch from the committed rocprofv3 PMC passes (profiles/*_pmc_traffic.json: separate
                --pmc FETCH_SIZE / WRITE_SIZE runs of this same command, which cannot be taken inside a timed run; FETCH x2
                per the gfx950 note of the guide)
  configs       the other BASELINE.json configurations on the same engine (N = 1 only): configs[0] 1dx4f @128 (the reference's
                plumbing case: its CPU path with ONE generator thread), 1dx4f @1024 x 4 096 lanes, 2dx3f @1024 x 16 384 lanes,
                2dx6f @2048 x 2 048 lanes -- value, ms_per_step, both kernels' roofline fractions and the reference path's
                rate on this box's host cores
  per_gpu       (N > 1) every rank's own rate and roofline fractions, and the ranks the RCCL process group saw
  rela_boundary the SAME metric measured the reference's way, through the drop-in boundary (cfvpy/selfplay.py:187-252, 285-293;
                gen_benchmark.cc:146-153): scripted Net2 in a rela.ModelLocker, rela.ValuePrioritizedReplay(capacity 2 000 000),
                1 000 x create_cfr_thread (REBEL_AMD_LANES_PER_GPU spreads the lanes over them), Context.start(); rate = delta replay.num_add() / 2 x subgame_iters /
                wall-clock between epoch boundaries, WHILE a consumer thread calls replay.sample(512, "cuda:0") in a loop and
                locker.update_model(net) every 2 s
  cpu_baseline  the UNMODIFIED reference path (oracle/_ref/rela*.so) timed on this box's host cores for >= 30 s per
                generator-thread count in {16, 32, 60 (README), os.cpu_count()} -- a reported baseline, not a target.
"""
import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_F16_PEAK_TFLOPS = 2500.0  # /opt/skills/guides/MI355X_MICROARCH.md: dense f16/bf16 MFMA peak (the net kernel's pipe)
MFMA_F32_PEAK_TFLOPS = 157.3   # same guide: f32-input MFMA peak (= f32 vector peak), quoted for context
HBM_PEAK_GBPS = 8000.0         # same guide: HBM3E spec peak

# BASELINE.json configs[0], [1], [3], [4] as single-GPU legs of the `configs` block (configs[2] is the headline):
# (index, dice, faces, subgame_iters, lanes).  configs[0] is the reference's own plumbing case -- 1 die x 4 faces, 128
# iterations, cpu_gen_threads = 1 on its CPU path: its `cpu_reference` is timed with ONE
# generator thread; the engine leg beside it runs the same game and iteration count on 4 096 lanes.
OTHER_CONFIGS = [(0, 1, 4, 128, 4096), (1, 1, 4, 1024, 4096), (3, 2, 3, 1024, 16384), (4, 2, 6, 2048, 2048)]


def cpu_baseline(dice, faces, iters, seconds, threads=0):
    """The UNMODIFIED reference path (oracle/_ref) timed on this box's host cores -> dict, or None + the reason when the
    compiled reference is not there.  Nothing is substituted for it: a line either carries the reference's own rate or
    `cpu_baseline: null` with `cpu_baseline_error` (round 4 fell back to the port oracle with a synthetic net, which
    silently changed what the key meant)."""
    cmd = [sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), "--dice", str(dice), "--faces", str(faces),
           "--iters", str(iters), "--seconds", str(seconds), "--threads", str(threads)]
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=seconds + 240)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
        out = json.loads(line)
        if "error" in out:
            raise RuntimeError(out["error"])
        return {k: out[k] for k in ("value", "unit", "cores", "host_cores", "kind", "sample")}, None
    except Exception as ex:
        return None, f"reference CPU path not measured: {ex}"


class PowerSampler:
    """Socket power and gfx clock of GPU `index` during a timed region (amdsmi gpu_metrics, ~10 ms): the value-net forward runs
    AT the socket's power cap (profiles/r04_net_energy_attribution.txt), so the line carries the power it was measured at."""

    def __init__(self, index):
        import threading

        self.samples, self.stop, self.smi = [], False, None
        try:
            import amdsmi

            amdsmi.amdsmi_init()
            self.h = amdsmi.amdsmi_get_processor_handles()[index]
            self.smi = amdsmi
            self.cap_w = float(amdsmi.amdsmi_get_power_cap_info(self.h)["power_cap"]) / 1e6
            self._read()
        except Exception:  # no SMI access on this box: the block is simply absent
            self.smi = None
        self.t = threading.Thread(target=self._run, daemon=True)

    def _read(self):
        m = self.smi.amdsmi_get_gpu_metrics_info(self.h)
        clks = [c for c in (m.get("current_gfxclks") or []) if isinstance(c, (int, float)) and 0 < c < 10000]
        pw = m.get("current_socket_power")
        if not isinstance(pw, (int, float)) or not 0 < pw < 5000:
            pw = m.get("average_socket_power")
        return time.perf_counter(), float(pw or 0), (sum(clks) / len(clks)) if clks else float(m.get("current_gfxclk") or 0)

    def _run(self):
        while not self.stop and self.smi:
            try:
                self.samples.append(self._read())
            except Exception:
                return
            time.sleep(0.01)

    def start(self):
        if self.smi:
            self.t.start()

    def summary(self, t0, t1):
        self.stop = True
        s = [x for x in self.samples if t0 <= x[0] <= t1]
        if not s:
            return None
        return {"mean_socket_power_w": sum(x[1] for x in s) / len(s), "mean_gfxclk_mhz": sum(x[2] for x in s) / len(s),
                "power_cap_w": self.cap_w, "samples": len(s), "source": "amdsmi gpu_metrics at ~10 ms over the timed region"}


def rela_boundary_leg(dice, faces, iters, lanes, device_index, epochs, consumer):
    """The metric the reference's way, through the drop-in boundary: exactly cfvpy/selfplay.py:187-252 (initialize_datagen) and
    :285-293 / gen_benchmark.cc:146-153 (rate = replay.num_add() / wall-clock), with a trainer-shaped consumer beside the
    generators -- replay.sample(512, device) in a loop (selfplay.py:411 per train iteration) and ModelLocker.update_model(net)
    every 2 s (selfplay.py: network_sync_epochs).  num_add() is polled at ~0.3 ms: it moves once per epoch (2 x lanes examples
    appended as one block), so the time stamps of its changes ARE the epoch boundaries and the rate is exact over whole epochs --
    the fixed 8 s window of rounds 3-4 (scripts/probe_rela_throughput.py) cut an epoch in two at either end (+-4 %)."""
    import threading

    import torch

    import rebel_amd.rela as rela
    from rebel_amd.models import Net2

    # at most 1 000 create_cfr_thread calls per ModelLocker (selfplay.py:250 seeds rank*1000 + i: call #1001 would replay the next
    # rank's game, and Context.start() refuses it); REBEL_AMD_LANES_PER_GPU spreads the lanes over them (16 384 = 384 x 17 + 616 x 16)
    threads = max(1, min(1000, lanes))
    dev = f"cuda:{device_index}"
    prev = os.environ.get("REBEL_AMD_LANES_PER_GPU")
    os.environ["REBEL_AMD_LANES_PER_GPU"] = str(lanes)
    ctx = None
    try:
        torch.manual_seed(0)
        net = Net2(num_faces=faces, num_dice=dice, n_hidden=256, use_layer_norm=True, n_layers=2).eval()
        ref_model = torch.jit.script(Net2(num_faces=faces, num_dice=dice, n_hidden=256, use_layer_norm=True, n_layers=2).to(dev)).eval()
        ref_model.load_state_dict(net.state_dict())
        locker = rela.ModelLocker([ref_model], dev)
        # liars_sp.yaml's replay block (capacity 2 000 000, alpha 1, beta 1, prefetch 8, use_priority false)
        replay = rela.ValuePrioritizedReplay(capacity=2000000, seed=10001, alpha=1.0, beta=1.0, prefetch=8, use_priority=False,
                                             compressed_values=False)
        cfg = rela.RecursiveSolvingParams()
        cfg.num_dice, cfg.num_faces, cfg.random_action_prob, cfg.sample_leaf = dice, faces, 0.25, True
        sp = cfg.subgame_params
        sp.num_iters, sp.max_depth, sp.linear_update, sp.use_cfr = iters, 2, True, True
        ctx = rela.Context()
        for i in range(threads):
            ctx.push_env_thread(rela.create_cfr_thread(locker, replay, cfg, i))
        stop = threading.Event()
        seen = {"samples": 0, "updates": 0, "error": None}

        def consume():
            last = time.perf_counter()
            try:
                while not stop.is_set():
                    if replay.size() < 1024:  # burn-in, selfplay.py:314-327
                        time.sleep(0.005)
                        continue
                    batch, _ = replay.sample(512, dev)
                    assert batch.query.shape[0] == 512
                    seen["samples"] += 1
                    if time.perf_counter() - last >= 2.0:
                        locker.update_model(net)
                        seen["updates"] += 1
                        last = time.perf_counter()
            except Exception as ex:  # reported in the line, never swallowed
                seen["error"] = repr(ex)

        th = threading.Thread(target=consume, daemon=True)
        ctx.start()
        if consumer:
            th.start()
        lanes_run = lanes
        stamps, last_n, t_begin = [], 0, time.perf_counter()
        skip = 2  # the first epochs carry engine creation and the cold start
        while len(stamps) < skip + epochs + 1:
            n = replay.num_add()
            if n != last_n:
                stamps.append((time.perf_counter(), n))
                last_n = n
            if time.perf_counter() - t_begin > 600:
                raise RuntimeError(f"rela leg: only {len(stamps)} epochs in 600 s")
            if ctx.terminated():
                raise RuntimeError("rela leg: the generators stopped")
            time.sleep(0.0003)
        t_samples0 = seen["samples"]
        stop.set()
        if consumer:
            th.join(30)
        (t0, n0), (t1, n1) = stamps[skip], stamps[-1]
        per_epoch = sorted((b[0] - a[0]) * 1e3 for a, b in zip(stamps[skip:-1], stamps[skip + 1:]))
        out = {"value": (n1 - n0) / 2 * iters / (t1 - t0), "unit": "subgame-CFR-iterations/s", "examples_per_s": (n1 - n0) / (t1 - t0),
               "epochs": len(stamps) - 1 - skip, "ms_per_epoch_median": per_epoch[len(per_epoch) // 2],
               "ms_per_epoch_max": per_epoch[-1], "examples_per_epoch": (n1 - n0) // (len(stamps) - 1 - skip),
               "lanes": lanes_run, "create_cfr_thread_calls": threads,
               "lanes_per_thread": sorted({lanes // threads, -(-lanes // threads)}),
               "replay": {"capacity": 2000000, "use_priority": False, "prefetch": 8, "storage": replay._storage_device()},
               "consumer": ({"sample_calls_per_s": t_samples0 / max(1e-9, stamps[-1][0] - stamps[0][0]), "batch": 512,
                             "update_model_calls": seen["updates"], "error": seen["error"]} if consumer else None)}
        return out
    finally:
        if ctx is not None:
            ctx.terminate()
            t_end = time.time()
            while not ctx.terminated() and time.time() - t_end < 60:
                time.sleep(0.01)
            del ctx
        if prev is None:
            os.environ.pop("REBEL_AMD_LANES_PER_GPU", None)
        else:
            os.environ["REBEL_AMD_LANES_PER_GPU"] = prev


def spawn_ranks(n_gpus, share_gpu=False):
    """`--gpus N` without a rank environment: start the N ranks ourselves, one process per GPU (torch.distributed.run)."""
    import torch

    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if share_gpu and have >= 1:
        have = n_gpus  # test-only: every rank on GPU 0 (see --share-gpu)
    if have < n_gpus:
        sys.stderr.write(f"bench.py: --gpus {n_gpus} was asked for but {have} GPU(s) are visible on this box; refusing to "
                         f"measure fewer GPUs than the line would claim (n_gpus)\n")
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(sys.argv[0])] + sys.argv[1:]
    return subprocess.call(cmd, env=dict(os.environ, OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "1")))


def rocprof_timed_epochs(kernel_key, steps_warmup):
    """Per-launch duration of `kernel_key` over the timed epochs of the newest committed rocprofv3 --kernel-trace run of this
    command (scripts/collect_profiles.sh -> profiles/rNN_kernel_stats_timed_epochs.csv), if it has this run's steps / warm-up."""
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_kernel_stats_timed_epochs.csv")), reverse=True):
        tag = os.path.basename(path).split("_")[0]
        try:
            meta = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic.json")))
            if meta.get("steps_warmup") != list(steps_warmup):
                continue
            for line in open(path).read().splitlines()[1:]:
                # kernel,launches_all,avg_ns_all,launches_timed_epochs,avg_ns_timed_epochs,bench_event_avg_ns -- the kernel
                # name is a template instantiation with commas of its own (quoted from round 4 on)
                name, _, _, n_timed, avg_timed, _ = line.rsplit(",", 5)
                if any(k in name for k in kernel_key):
                    return {"avg_launch_us": float(avg_timed) / 1e3, "launches": int(n_timed), "kernel": name.strip('"'),
                            "lanes_profiled": meta.get("lanes"), "source": os.path.relpath(path, ROOT)}
        except Exception:
            continue
    return None


def pmc_traffic(kernel_key):
    """HBM-side bytes per launch of `kernel_key` from the newest committed PMC summary (scripts/collect_profiles.sh)."""
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        for name, k in d.get("kernels", {}).items():
            if any(key in name for key in kernel_key) and "fetch_size_bytes_per_launch" in k and \
                    "write_size_bytes_per_launch" in k:
                # mean over the launches of the profiled run's TIMED epochs where the summary has it (same command
                # shape as this run: same lanes, seeds, warm-up), else the median over all launches
                rd, wr = (x.get("timed_epochs", {}).get("mean", x["median"]) for x in
                          (k["fetch_size_bytes_per_launch"], k["write_size_bytes_per_launch"]))
                return {"bytes": rd + wr, "read": rd, "written": wr,
                        "restricted_to_timed_epochs": "timed_epochs" in k["fetch_size_bytes_per_launch"],
                        "profiled_steps_warmup": d.get("steps_warmup"),
                        "source": os.path.relpath(path, ROOT), "kernel": name,
                        "lanes_profiled": d.get("lanes"), "note": d.get("note")}
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--lanes", type=int, default=int(os.environ.get("BENCH_LANES", 16384)))
    ap.add_argument("--iters", type=int, default=int(os.environ.get("BENCH_ITERS", 1024)))
    ap.add_argument("--dice", type=int, default=1)
    ap.add_argument("--faces", type=int, default=6)
    ap.add_argument("--cpu-seconds", type=float, default=float(os.environ.get("BENCH_CPU_SECONDS", 120)),
                    help="total CPU-baseline window, split over the generator-thread counts (>= 30 s each by default)")
    ap.add_argument("--config-cpu-seconds", type=float, default=float(os.environ.get("BENCH_CONFIG_CPU_SECONDS", 15)),
                    help="reference-path window per entry of the `configs` block (at the headline's best thread count)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="skip the comparison legs and the `configs` block (profiling runs)")
    ap.add_argument("--no-configs", action="store_true", help="skip the `configs` block only")
    ap.add_argument("--force-dist", action="store_true", help="initialise the RCCL process group even for one rank")
    ap.add_argument("--share-gpu", action="store_true",
                    help="TEST ONLY (a box with one GPU): every rank runs its engine on GPU 0 and the bookkeeping goes over gloo "
                         "(RCCL refuses two ranks on one device).  The line says so; it is never a scaling number")
    ap.add_argument("--dump-examples", default=None,
                    help="TEST ONLY: directory; every rank saves the timed epochs' training examples as rank<r>.npz")
    ap.add_argument("--rela-epochs", type=int, default=int(os.environ.get("BENCH_RELA_EPOCHS", 12)),
                    help="timed epochs of the rela_boundary leg (0 = skip the leg)")
    a = ap.parse_args()
    if a.gpus < 1:
        ap.error("--gpus must be >= 1")

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(spawn_ranks(a.gpus, a.share_gpu))

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != a.gpus:
        sys.stderr.write(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {a.gpus} (or without a "
                         f"rank environment, then bench.py starts the ranks itself)\n")
        sys.exit(2)

    import numpy as np
    import torch  # first: librebel_hip.so then binds to the HIP runtime torch already loaded (same SONAME)
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs a GPU: rebel_amd has no CPU fallback")
    if a.share_gpu:
        local_rank = 0  # every rank's engine on GPU 0
    if torch.cuda.device_count() <= local_rank:
        raise RuntimeError(f"bench.py: rank {rank} wants GPU {local_rank} but {torch.cuda.device_count()} are visible")
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or a.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if a.share_gpu:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from rebel_amd import capi
    from rebel_amd.models import Net2, mlp_weights_from_state_dict
    from rebel_amd.sharding import gather_ranks, lane_seeds, reduce_job

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def make_net(dice, faces, half=False):
        torch.manual_seed(0)  # same random-init net on every rank (weights are read-only shared state)
        net = Net2(num_faces=faces, num_dice=dice, n_hidden=256, use_layer_norm=True, n_layers=2).eval()
        if half:  # what the trainer's half_inference hands the generators: model.half() (cfvpy/selfplay.py:42-43)
            net = net.half()
        return mlp_weights_from_state_dict({k: v.float() for k, v in net.state_dict().items()})

    power = {}

    leg_roots = []  # root-subgame lanes of each timed epoch of the LAST run_leg call (the epoch mix of a batch oscillates at first)

    def run_leg(game, lanes, warmup, steps, timing_stride, sync_ranks, precision=0, sample_power=False, dump=None):
        """`warmup` untimed + `steps` timed epochs on a fresh engine -> (seconds, units, games, examples, kernel stats)."""
        dice, faces, iters = game
        n_act = 2 * dice * faces + 1
        del leg_roots[:]
        params = capi.make_params(num_iters=iters, max_depth=2, linear_update=True, use_cfr=True)
        eng = capi.Engine(dice, faces, params, max_lanes=lanes, device=local_rank)
        eng.set_net_precision(precision)
        eng.set_net_mlp(*make_net(dice, faces, half=precision != 0))
        sp = capi.SelfPlay(eng, lane_seeds(rank, lanes), random_action_prob=0.25, sample_leaf=True)
        for _ in range(warmup):
            sp.advance(collect=False)
        eng.sync()
        eng.stats(reset=True)
        eng.timing(timing_stride)
        n_ex, games0, units = 0, sp.games_finished(), 0
        sampler = PowerSampler(local_rank) if sample_power else None
        if sampler:
            sampler.start()
        if sync_ranks:
            barrier()
        t0 = time.perf_counter()
        kept = []
        for _ in range(steps):
            n, lanes_, q, v = sp.advance(collect=True)  # examples land in host arrays = the replay push hand-off
            units += n
            n_ex += len(lanes_)
            # lanes whose subgame started at the root state: the one-hot last bid of their example is empty (write_query_to)
            if getattr(q, "ndim", 0) == 2:
                leg_roots.append(int((q[0::2, 2:2 + n_act].sum(axis=1) == 0).sum()))
            if dump:
                kept.append((q, v))
        eng.sync()
        if sync_ranks:
            barrier()
        dt = time.perf_counter() - t0
        if sampler:
            power["headline"] = sampler.summary(t0, t0 + dt)
        st = eng.stats(reset=True)
        eng.timing(0)
        if dump:
            os.makedirs(dump, exist_ok=True)
            np.savez(os.path.join(dump, f"rank{rank}.npz"), q=np.stack([k[0] for k in kept]), v=np.stack([k[1] for k in kept]),
                     seeds=np.asarray(lane_seeds(rank, lanes)))
        games = sp.games_finished() - games0
        on_device = sp.on_device()
        sp.close()
        eng.close()
        return dt, units, games, n_ex, st, on_device

    def kernel_figures(st):
        net_t = st["net_ms"] / max(1, st["net_launches"]) * 1e-3
        cfr_t = st["cfr_ms"] / max(1, st["cfr_launches"]) * 1e-3
        net_tf = st["net_flops"] / max(1, st["net_launches"]) / net_t / 1e12 if net_t > 0 else 0.0
        cfr_gb = st["cfr_bytes"] / max(1, st["cfr_launches"]) / cfr_t / 1e9 if cfr_t > 0 else 0.0
        return net_t, cfr_t, net_tf, cfr_gb

    headline = (a.dice, a.faces, a.iters)
    dt, units, games, n_examples, st, walk_on_device = run_leg(headline, a.lanes, a.warmup, a.steps, 7, True, sample_power=rank == 0,
                                                                  dump=a.dump_examples)
    net_t, cfr_t, net_tf, cfr_gb = kernel_figures(st)
    headline_roots = list(leg_roots)

    dt_max, units_all, games_all = reduce_job(dist, world if not a.force_dist else max(world, 2), dt, float(units),
                                              float(games)) if use_dist else (dt, float(units), float(games))
    per_rank = gather_ranks(dist, world, [float(rank), float(local_rank), units / dt, dt, net_tf / MFMA_F16_PEAK_TFLOPS,
                                          cfr_gb / HBM_PEAK_GBPS, cfr_gb, net_t * 1e6, cfr_t * 1e6]) if use_dist else None

    two_streams = lanes4096 = None
    leg_errors = {}  # an extra leg that fails must not cost the headline its line: the failure goes INTO the line instead

    def extra_leg(name, *args, **kw):
        try:
            return run_leg(*args, **kw)
        except Exception as ex:
            leg_errors[name] = repr(ex)
            return None

    parts_env = os.environ.get("RBL_PARTS")
    streams = int(st["n_streams"])  # what the engine ran (rbl_engine_stats), not what the environment suggests
    if world == 1 and not a.no_extra_legs and streams == 1 and not parts_env:
        # what two interleaved half-batches would give at this lane count (kernel tails overlap; per-kernel timings are
        # then contended, which is why the headline leg runs one stream)
        os.environ["RBL_PARTS"] = "2"
        r_ = extra_leg("two_streams", headline, a.lanes, a.warmup, a.steps, 0, False)
        del os.environ["RBL_PARTS"]
        if r_:
            two_streams = {"value": r_[1] / r_[0], "note": "RBL_PARTS=2, same warm-up and timed epochs as the headline leg"}
    if world == 1 and not a.no_extra_legs and a.lanes != 4096:
        r_ = extra_leg("lanes_4096", headline, 4096, a.warmup, a.steps, 0, False)
        if r_:
            lanes4096 = {"value": r_[1] / r_[0], "note": "same engine at 4096 lanes (BASELINE config 2's lane count; two "
                         "streams), same warm-up and timed epochs as the headline leg"}

    half_leg = None
    if world == 1 and not a.no_extra_legs:
        # LABELLED EXTRA LEG, never the headline: the same workload with a half model (the trainer's `half_inference`) in the
        # one-product arithmetic a half module selects (rbl_engine_set_net_precision 2: f16 activations x f16 weights, f32
        # accumulation, f32 LayerNorm / GELU -- at least as accurate as the half torch module, tests/test_net_parity.py)
        r_ = extra_leg("half_inference", headline, a.lanes, a.warmup, a.steps, 7, False, precision=2)
        if r_:
            half_leg = (r_[0], r_[1], r_[4])

    dedup_leg = None
    if world == 1 and not a.no_extra_legs:
        # LABELLED EXTRA LEG, never the headline: REBEL_AMD_ROOT_DEDUP=1.  Every lane in its ROOT subgame computes the
        # same thing (RlRunner::step resets to the root with uniform beliefs, recursive_solving.cc:160-163; CFR::step draws nothing),
        # so one lane per epoch solves the root and the other root lanes sample from its sigma at their own act_iteration: the
        # example stream per seed is bit-identical (tests/test_selfplay_parity.py::test_root_dedup_*), the iterations EXECUTED drop.
        # `value`, `roofline` and `rela_boundary` above keep executing every lane's iterations, as the reference's threads do.
        os.environ["REBEL_AMD_ROOT_DEDUP"] = "1"
        try:
            r_ = extra_leg("root_dedup", headline, a.lanes, a.warmup, a.steps, 7, False)
            if r_:
                dedup_leg = (r_[0], r_[1], r_[2], r_[3], r_[4], list(leg_roots))
        finally:
            del os.environ["REBEL_AMD_ROOT_DEDUP"]

    rela_leg = None
    if world == 1 and rank == 0 and not a.no_extra_legs and a.rela_epochs > 0:
        try:
            rela_leg = rela_boundary_leg(a.dice, a.faces, a.iters, a.lanes, local_rank, a.rela_epochs, consumer=True)
            rela_leg["without_consumer"] = {k: v for k, v in rela_boundary_leg(a.dice, a.faces, a.iters, a.lanes, local_rank,
                                                                               max(4, a.rela_epochs // 2), consumer=False).items()
                                            if k in ("value", "unit", "epochs", "ms_per_epoch_median")}
        except Exception as ex:  # the leg must not cost the headline its line; the failure is in the line instead
            rela_leg = {"value": None, "error": repr(ex)}

    def roofline_blocks(game, st, streams_):
        """The `roofline` / `roofline_cfr` objects of one leg from its kernel stats."""
        dice, faces, _ = game
        n_t, c_t, n_tf, c_gb = kernel_figures(st)
        H_, A_ = faces ** dice, 2 * dice * faces + 1
        Q_ = 2 + A_ + 2 * H_
        prods = st.get("net_products", 3) or 3
        issued_ratio = prods * (-(-Q_ // 32) * 32 * 256 + 256 * 256 + 256 * -(-H_ // 16) * 16) / (Q_ * 256 + 256 * 256 + 256 * H_)
        how = "HIP events bound to the dispatch packet (hipExtLaunchKernelGGL): the kernel's own begin -> end interval, as " \
              "rocprofv3 --kernel-trace reports it; "
        how += "one stream, net(all lanes) -> cfr(all lanes) per iteration: the kernel has the GPU to itself" if streams_ == 1 \
            else "in-mix: the other stream's kernel is active beside it"
        rows = st["net_rows"] / max(1, st["net_launches"])
        net = {"kernel": capi.NET_KERNEL_NAMES.get(st["net_kernel"], str(st["net_kernel"])), "bound": "mfma", "achieved": n_tf,
               "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": n_tf / MFMA_F16_PEAK_TFLOPS, "traffic": None,
               "avg_launch_us": n_t * 1e6, "timed_launches": st["net_launches"], "rows_per_launch": rows,
               "ns_per_row": n_t * 1e9 / max(1.0, rows),
               "algorithmic_flops_per_launch": st["net_flops"] / max(1, st["net_launches"]), "measured": how,
               "issued_mfma_tflops": n_tf * issued_ratio, "vs_f32_mfma_peak": n_tf / MFMA_F32_PEAK_TFLOPS}
        cfr = {"kernel": capi.CFR_KERNEL_NAMES.get(st["cfr_kernel"], str(st["cfr_kernel"])), "bound": "hbm", "achieved": c_gb,
               "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": c_gb / HBM_PEAK_GBPS, "traffic": None,
               "avg_launch_us": c_t * 1e6, "timed_launches": st["cfr_launches"],
               "algorithmic_bytes_per_launch": st["cfr_bytes"] / max(1, st["cfr_launches"]),
               "measured": how + ("" if st["cfr_kernel"] != 4 else "; this kernel is launched once per size-sorted segment of the "
                                  "lanes, so a step is bracketed with recorded events (launch gaps between the segments included)")}
        return net, cfr

    configs = []
    if world == 1 and not a.no_extra_legs and not a.no_configs and headline == (1, 6, 1024):
        # the other BASELINE.json configurations on this engine, one GPU; short legs (3 warm-up + 5 timed epochs) -- except 2 dice x 6
        # faces, whose root / small-tree mix oscillates for the first epochs of a batch (all lanes start at the root: 2 048, 105,
        # 1 673, 419, 1 431 ... roots, scripts/probe_2d6f_mix.py; a damped oscillation that is within +-60 of its level of ~985 after
        # a dozen epochs) and decides its rate -- a step there is ROUNDS of root workgroups, one per CU, so 1 024 roots are four
        # rounds and 1 025 five: 16 warm-up + 8 timed epochs there, and every leg prints its per-epoch root count so
        # that two runs can be compared
        for idx, d_, f_, it_, ln_ in OTHER_CONFIGS:
            cw, cs = (16, 8) if (d_, f_) == (2, 6) else (min(3, max(1, a.warmup)), min(5, max(2, a.steps)))
            if a.steps < 3:  # test-sized runs stay short
                cw, cs = min(cw, 3), min(cs, 2)
            r_ = extra_leg(f"configs[{idx}]", (d_, f_, it_), ln_, cw, cs, 7, False)
            if not r_:
                continue
            cdt, cunits, cgames, _, cst, _ = r_
            croots = list(leg_roots)
            cnet, ccfr = roofline_blocks((d_, f_, it_), cst, int(cst["n_streams"]))
            keep = ("kernel", "achieved", "unit", "frac", "avg_launch_us", "timed_launches")
            configs.append({"baseline_config": idx, "workload": f"{d_}dx{f_}f self-play, {ln_} lanes, subgame_iters={it_}, depth 2",
                            "value": cunits / cdt, "unit": "subgame-CFR-iterations/s", "ms_per_step": cdt / cs * 1e3,
                            "steps": cs, "warmup": cw, "games_per_s": cgames / cdt, "streams": int(cst["n_streams"]),
                            "roots_per_epoch": croots,
                            "net": {k: cnet[k] for k in keep + ("rows_per_launch", "ns_per_row")},
                            "cfr": {k: ccfr[k] for k in keep + ("algorithmic_bytes_per_launch",)}})

    if rank == 0:
        rl_net, rl_cfr = roofline_blocks(headline, st, streams)
        out = {
            "metric": "subgame CFR iters/sec (whole job), 1dx6f @1024 iters; self-play games/sec in games_per_s",
            "value": units_all / dt_max,
            "unit": "subgame-CFR-iterations/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": dt_max / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64 CFR state; value net f32 in/out, GEMMs as f16x2-split MFMA (3 f16 products per multiply, "
                     "f32 accumulate, 4e-7 max error vs float64), f32 LayerNorm/GELU",
            "data": "synthetic (self-play from the root state, random-init Net2 seed 0, lane seeds rank*lanes+i)",
            # short on purpose: the driver keeps the first ~120 characters (the lane count has to survive the cut)
            "config": {"workload": f"{a.dice}dx{a.faces}f self-play, {a.lanes} lanes/GPU, subgame_iters={a.iters}, depth 2, "
                                   f"linear CFR, Net2 256x2+LN on MFMA",
                       "detail": "RlRunner::step per lane: sample_leaf, random_action_prob=0.25; every lane one subgame per step",
                       "lanes_per_gpu": a.lanes, "subgame_iters": a.iters, "parallelism": f"independent lane sets x{world}",
                       "lanes_note": "16 384 lanes since round 2 (BENCH_r01 ran 4 096: see lanes_4096 for that size)"},
            "games_per_s": games_all / dt_max,
            "examples_per_s": n_examples * world / dt_max,
            # achieved = ALGORITHMIC flops 2*rows*(Q*256 + 256*256 + 256*H) per launch / mean launch time; the kernel issues
            # 3 f16 MFMA products per multiply on padded tiles (K 27->32, H 6->16): 3.16x this on the matrix pipe.  On
            # gfx950 the f16 MFMA pipe and f32 FMA-class VALU work do not overlap (scripts/micro/), and the LayerNorm +
            # erf-GELU epilogue on 512 activations per row costs about as many issue cycles as the MFMAs: see DESIGN.md
            "roofline": rl_net,
            "roofline_cfr": rl_cfr,
            "streams": streams,
            "selfplay_walk": "device kernels" if walk_on_device == 1 else "host",
            "roots_per_epoch": headline_roots,  # rank 0's lanes in their root subgame, per timed epoch (identical work: DESIGN 7)
        }
        for key, kern in (("roofline", ("mlp_resident_kernel",)), ("roofline_cfr", ("cfr_wave_kernel", "cfr_rows_kernel"))):
            # Everything in this object is READ FROM FILES COMMITTED UNDER profiles/ (an earlier rocprofv3 session of this same
            # command on another box of the pool), never measured by this run: PMC counter passes serialise the kernels and cannot
            # ride in a timed run.  It is kept apart from the live figures (achieved / frac / avg_launch_us above are this run's HIP
            # events); `traffic` beside them stays null in the live line.
            fcp = {}
            # the committed PMC passes were taken on the headline configuration: other games carry none
            tr = pmc_traffic(kern) if headline == (1, 6, 1024) else None
            if tr:  # scaled by the lanes the passes were taken at
                scale = (a.lanes / tr["lanes_profiled"]) if tr.get("lanes_profiled") else 1.0
                fcp["traffic"] = tr["bytes"] * scale
                fcp["traffic_unit"] = "bytes per launch (FETCH_SIZE x2 + WRITE_SIZE, separate --pmc passes, gfx950 corrections)"
                fcp["traffic_detail"] = tr
                work = out[key].get("algorithmic_bytes_per_launch")
                if work:
                    fcp["traffic_over_algorithmic"] = fcp["traffic"] / work
            rp = rocprof_timed_epochs(kern, (a.steps, a.warmup)) if headline == (1, 6, 1024) else None
            if rp and rp.get("lanes_profiled") == a.lanes:
                # the same launches (same seeds, lanes, warm-up, timed epochs) as seen by rocprofv3 --kernel-trace in the committed
                # profile: the algorithmic work per launch is identical, so a frac follows from that run's duration
                work = out[key]["algorithmic_flops_per_launch"] / 1e12 if key == "roofline" else \
                    out[key]["algorithmic_bytes_per_launch"] / 1e9
                rp["achieved"] = work / (rp["avg_launch_us"] * 1e-6)
                rp["frac"] = rp["achieved"] / out[key]["peak"]
                fcp["rocprof"] = rp
            if fcp:
                fcp["note"] = "NOT measured in this run: read from the committed profiles named in `source`"
                out[key]["from_committed_profile"] = fcp
        if power.get("headline"):
            # the dominant kernel is socket-power-bound (84 % of the step at ~1.39 kW of a 1.4 kW cap): see DESIGN.md 3.4
            out["power"] = power["headline"]
        if per_rank:
            if a.share_gpu:
                out["share_gpu"] = "TEST MODE: every rank ran on GPU 0 (gloo bookkeeping); not a scaling measurement"
            # north_star: "1/2/4/8-GPU throughput reported as absolute numbers and as fraction of HBM roofline" -- the job's CFR sweep
            # against N x 8 TB/s and its net forward against N x the f16 MFMA peak (sums of the ranks' live kernel figures)
            n_r = len(per_rank)
            out["job"] = {"n_gpus": n_r, "value": out["value"], "value_per_gpu_mean": sum(r[2] for r in per_rank) / n_r,
                          "value_per_gpu_min": min(r[2] for r in per_rank), "value_per_gpu_max": max(r[2] for r in per_rank),
                          "cfr_gbps_sum": sum(r[6] for r in per_rank), "hbm_peak_gbps_sum": HBM_PEAK_GBPS * n_r,
                          "cfr_frac_of_n_x_hbm_roofline": sum(r[6] for r in per_rank) / (HBM_PEAK_GBPS * n_r),
                          "net_tflops_sum": sum(r[4] for r in per_rank) * MFMA_F16_PEAK_TFLOPS,
                          "net_frac_of_n_x_mfma_peak": sum(r[4] for r in per_rank) / n_r,
                          "slowest_rank_seconds": max(r[3] for r in per_rank), "fastest_rank_seconds": min(r[3] for r in per_rank)}
            out["per_gpu"] = {"ranks_seen_by_rccl": dist.get_world_size(), "backend": dist.get_backend(),
                              "ranks": [{"rank": int(r[0]), "gpu": int(r[1]), "value": r[2], "seconds": r[3], "net_frac_mfma": r[4],
                                         "cfr_frac_hbm": r[5], "cfr_gbps": r[6], "net_launch_us": r[7], "cfr_launch_us": r[8]}
                                        for r in per_rank]}
        if half_leg:
            hdt, hunits, hst = half_leg
            hnet, hcfr = roofline_blocks(headline, hst, int(hst["n_streams"]))
            out["half_inference"] = {
                "label": "EXTRA LEG, not the headline: half model (cfvpy/selfplay.py half_inference), value-net GEMMs as ONE f16 "
                         "MFMA product per multiply (f16 activations and weights, f32 accumulate, f32 LayerNorm/GELU)",
                "value": hunits / hdt, "unit": "subgame-CFR-iterations/s", "ms_per_step": hdt / a.steps * 1e3,
                "net": {k: hnet[k] for k in ("kernel", "achieved", "unit", "frac", "avg_launch_us", "ns_per_row", "issued_mfma_tflops")},
                "cfr": {k: hcfr[k] for k in ("kernel", "achieved", "unit", "frac", "avg_launch_us")},
                "net_products": hst.get("net_products")}
        if dedup_leg:
            ddt, dunits, dgames, dex, dst, droots = dedup_leg
            dnet, dcfr = roofline_blocks(headline, dst, int(dst["n_streams"]))
            out["root_dedup"] = {
                "label": "EXTRA LEG, not the headline and not the metric: REBEL_AMD_ROOT_DEDUP=1 (opt-in).  The root subgame is solved "
                         "by ONE lane per epoch; the other root lanes read its strategy at their own act_iteration.  Training "
                         "examples and trajectories per seed are bit-identical to the headline run; only executed work drops",
                "games_per_s": dgames / ddt, "examples_per_s": dex / ddt, "executed_iterations_per_s": dunits / ddt,
                "ms_per_step": ddt / a.steps * 1e3,
                "examples_per_s_vs_headline": (dex / ddt) / (n_examples / dt),
                "lane_epochs_served_by_the_representative": 1.0 - dunits / (a.lanes * a.iters * a.steps),
                "roots_per_epoch": droots,
                "net": {k: dnet[k] for k in ("kernel", "frac", "avg_launch_us", "rows_per_launch", "ns_per_row")},
                "cfr": {k: dcfr[k] for k in ("kernel", "frac", "avg_launch_us", "algorithmic_bytes_per_launch")}}
        if lanes4096:
            out["lanes_4096"] = lanes4096
        if two_streams:
            out["two_streams"] = two_streams
        if rela_leg is not None:
            out["rela_boundary"] = rela_leg
            if rela_leg.get("value"):
                rela_leg["ratio_to_value"] = rela_leg["value"] / out["value"]
                nc = rela_leg.get("without_consumer")
                if nc and nc.get("value"):
                    nc["ratio_to_value"] = nc["value"] / out["value"]
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"], err = cpu_baseline(a.dice, a.faces, a.iters, a.cpu_seconds)
            if err:
                out["cpu_baseline_error"] = err
            elif out["cpu_baseline"].get("value"):
                out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
            best_threads = out["cpu_baseline"].get("cores", 16) if out["cpu_baseline"] else 16
            for c in configs:  # the reference path on the same configuration, at the headline's best generator-thread count
                i_, d_, f_, it_ = next((i, d, f, it) for i, d, f, it, _ in OTHER_CONFIGS if i == c["baseline_config"])
                # configs[0] is quoted at cpu_gen_threads = 1 (BASELINE.json): one generator thread there
                c["cpu_reference"], err = cpu_baseline(d_, f_, it_, a.config_cpu_seconds, threads=1 if i_ == 0 else best_threads)
                if err:
                    c["cpu_reference_error"] = err
                elif c["cpu_reference"].get("value"):
                    c["speedup_vs_cpu_reference"] = c["value"] / c["cpu_reference"]["value"]
        if configs:
            out["configs"] = configs
        if leg_errors:
            out["extra_leg_errors"] = leg_errors
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
