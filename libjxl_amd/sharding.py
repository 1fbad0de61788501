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
    number of rows get an empty band (begin == end). Two ways to give a band the rows its filters read beyond it: every
    rank also decodes the group row above and below its band (the default: no exchange at all), or, with the context
    option "band_halo", the few rows are exchanged after the transform stage (exchange_halos below)."""
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


def exchange_halos(rank, world_size, pack, unpack, send, recv):
    """The halo exchange of one frame set split into bands (option "band_halo"; include/jxl_amd_hip.h jxlhip_halo_*): rank r
    owns band r; after its transform stage it needs the last rows of band r - 1 and the first rows of band r + 1.

    pack(side) -> a block (side 0 = this band's first rows, 1 = its last rows); unpack(side, block) (side 0 = the rows just
    above this band, 1 = just below); send(block, peer) / recv(peer) -> block move a block between neighbouring ranks
    (torch.distributed send / recv over RCCL between GPUs, a device-to-device copy inside one process). Ranks whose band
    is empty take no part; the schedule is deadlock-free for blocking transports: even ranks send first, odd ranks
    receive first, one neighbour at a time."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    steps = []
    for peer, my_side in ((rank - 1, 0), (rank + 1, 1)):  # my first rows go up, my last rows go down
        if 0 <= peer < world_size:
            steps.append((peer, my_side))
    for peer, my_side in steps:
        if rank % 2 == 0:
            send(pack(my_side), peer)
            unpack(my_side, recv(peer))
        else:
            block = recv(peer)
            send(pack(my_side), peer)
            unpack(my_side, block)
