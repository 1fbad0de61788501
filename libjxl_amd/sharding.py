"""Multi-GPU sharding for the decode path: frames (or, for very large frames, their independent 256x256 AC groups)
are independent units, so they are dealt round-robin to ranks with no data-path collective. The only
communication is the timing/throughput reduction of the benchmark."""


def shard_units(num_units, rank, world_size):
    """Indices of the units (frames) decoded by `rank`; round-robin so that every rank gets the same count +-1."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    return list(range(rank, num_units, world_size))


def band_of(num_group_rows, rank, world_size):
    """Rows of 256x256 groups [begin, end) of one frame that `rank` produces when a single large frame is split over
    the ranks (16K: 64 group rows -> 8 bands of 8). Contiguous bands, sizes differ by at most one row; ranks beyond the
    number of rows get an empty band (begin == end). The band decode needs no exchange between ranks: every rank also
    decodes the group row above and below its band (include/jxl_amd_hip.h, JxlHipFrameDesc)."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    base, extra = divmod(num_group_rows, world_size)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def aggregate(local_units, local_seconds, dist=None):
    """(total units over all ranks, max seconds over ranks). `dist` is torch.distributed or None (single process)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(local_units), float(local_seconds)
    import torch
    device = "cuda" if dist.get_backend() == "nccl" else "cpu"
    u = torch.tensor([float(local_units)], dtype=torch.float64, device=device)
    t = torch.tensor([float(local_seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return u.item(), float(t.item())
