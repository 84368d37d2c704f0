"""Multi-GPU layout of the data-generation job: games are independent (each lane owns its state, RNG and solver --
recursive_solving.h:73-85 in the reference), so GPUs get disjoint lane/seed ranges and there is NO data-path
collective; the only cross-rank traffic is the scalar bookkeeping of a benchmark (max time, summed work)."""


def lane_seeds(rank, lanes_per_gpu):
    """Disjoint seeds per rank.  (The reference's convention rank*1000+i, selfplay.py:250, collides above 1000 lanes.)"""
    return [rank * lanes_per_gpu + i for i in range(lanes_per_gpu)]


def reduce_job(dist, world, dt, units, games, device=None):
    """-> (max over ranks of dt, sum of units, sum of games).  The scalars travel on the process group's own device: GPU
    memory for RCCL ("nccl"), host memory for gloo (the CPU tests)."""
    if world <= 1:
        return dt, units, games
    import torch

    if device is None:
        device = "cuda" if dist.get_backend() == "nccl" else "cpu"

    tot = torch.tensor([dt, units, games], dtype=torch.float64, device=device)
    mx = tot.clone()
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    return mx[0].item(), tot[1].item(), tot[2].item()


def gather_ranks(dist, world, row, device=None):
    """Every rank's `row` (a list of floats) on every rank, in rank order: the per-GPU figures of a multi-GPU bench line.
    Bookkeeping only, like reduce_job -- there is no data-path collective."""
    if world <= 1 and not dist.is_initialized():
        return [list(row)]
    import torch

    if device is None:
        device = "cuda" if dist.get_backend() == "nccl" else "cpu"
    mine = torch.tensor(row, dtype=torch.float64, device=device)
    every = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(every, mine)
    return [t.tolist() for t in every]
