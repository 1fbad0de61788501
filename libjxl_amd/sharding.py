"""Multi-GPU sharding for the decode path: frames (or, for very large frames, their independent 256x256 AC groups)
are independent units, so they are dealt round-robin to ranks with no data-path collective. The only
communication is the timing/throughput reduction of the benchmark."""


def shard_units(num_units, rank, world_size):
    """Indices of the units (frames) decoded by `rank`; round-robin so that every rank gets the same count +-1."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    return list(range(rank, num_units, world_size))


def aggregate(local_units, local_seconds, dist=None):
    """(total units over all ranks, max seconds over ranks). `dist` is torch.distributed or None (single process)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return int(local_units), float(local_seconds)
    import torch
    device = "cuda" if dist.get_backend() == "nccl" else "cpu"
    u = torch.tensor([float(local_units)], dtype=torch.float64, device=device)
    t = torch.tensor([float(local_seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(round(u.item())), float(t.item())
