"""Frame-batch sharding across the GPUs of one node: every unit of work (frame i's ORB, pair
(i-1, i)'s BF match and GICP) is independent, so a batch is cut into contiguous blocks by frame index, one block
per rank, with NO collective on the data path.  torch.distributed (RCCL on GPUs, gloo in the CPU tests) is used
only for the barrier, the max-over-ranks timing and for gathering small POD results."""


def shard_range(n_items, rank, world):
    """Contiguous block [begin, end) of `n_items` owned by `rank` (sizes differ by at most one)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, rem = divmod(n_items, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def pair_halo(begin):
    """Pair (i-1, i) is owned by the owner of frame i; the first pair of a block needs frame begin-1 too
    (1-frame halo, re-extracted locally rather than moved between devices)."""
    return max(begin - 1, 0)


def gather_counts(local_count, dist):
    """Sum of per-rank processed units (all_reduce on a tiny tensor)."""
    import torch
    t = torch.tensor([int(local_count)], dtype=torch.int64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())
