"""Stream -> rank assignment and the only cross-rank exchange of the benchmark.

Independent input streams shard one set per GPU; there is no data-path collective (no RCCL traffic while
decoding).  Only the final frame count (SUM) and the wall time (MAX) are reduced, as the bench contract asks."""


def streams_of_rank(n_streams, rank, world):
    """stream i -> GPU i mod world (the reference hard-codes device 0, nv_dec.cpp:209)."""
    return [i for i in range(n_streams) if i % world == rank]


def reduce_result(dist, frames, seconds, device="cuda"):
    """Returns (total frames over all ranks, max seconds over ranks)."""
    if dist is None:
        return int(frames), float(seconds)
    import torch
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    f = torch.tensor([float(frames)], dtype=torch.float64, device=device)
    dist.all_reduce(f, op=dist.ReduceOp.SUM)
    return int(round(f.item())), float(t.item())
