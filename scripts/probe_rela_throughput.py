"""Throughput of the drop-in path: cfvpy.rela call sequence (ModelLocker + replay + create_cfr_thread x lanes + Context), as
the reference trainer would drive it; counts replay.num_add() like selfplay.py:285-293 does."""
import sys, time
sys.path.insert(0, '.')
import torch
import rebel_amd.rela as rela
from rebel_amd.models import Net2

d, f, iters = 1, 6, 1024
lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
torch.manual_seed(0)
net = Net2(num_faces=f, num_dice=d, n_hidden=256, use_layer_norm=True, n_layers=2)
ref_model = torch.jit.script(net.to("cuda:0")).eval()
locker = rela.ModelLocker([ref_model], "cuda:0")
replay = rela.ValuePrioritizedReplay(capacity=2 ** 20, seed=10001, alpha=1.0, beta=0.4, prefetch=3, use_priority=True,
                                     compressed_values=False)
cfg = rela.RecursiveSolvingParams()
cfg.num_dice, cfg.num_faces, cfg.random_action_prob, cfg.sample_leaf = d, f, 0.25, True
sp = cfg.subgame_params
sp.num_iters, sp.max_depth, sp.linear_update, sp.use_cfr = iters, 2, True, True
ctx = rela.Context()
for i in range(lanes):
    ctx.push_env_thread(rela.create_cfr_thread(locker, replay, cfg, i))
ctx.start()
t0, n0 = None, None
while True:
    time.sleep(0.25)
    n = replay.num_add()
    if t0 is None and n >= 2 * lanes:  # first epoch done: start the clock
        t0, n0 = time.time(), n
    if t0 is not None and time.time() - t0 > (float(sys.argv[2]) if len(sys.argv) > 2 else 8):
        dt = time.time() - t0
        print(f"lanes {lanes}: {(n - n0) / 2 * iters / dt / 1e6:.2f} M subgame-CFR-iterations/s through rela ({(n - n0) / dt:.0f} examples/s); "
              f"replay ring on {replay._storage_device()}")
        break
ctx.terminate()
