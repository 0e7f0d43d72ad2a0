"""Pair-level sharding across the GPUs of one node.

Image pairs are independent problems, so the path shards with no data-path collective: rank r
owns a contiguous block of pairs, runs the one-workgroup-per-pair kernel on its own GPU and only
the packed per-pair results (9 doubles + 16 int32 stats + n mask bytes) are gathered over
RCCL/xGMI (backend "nccl" on ROCm; "gloo" in the CPU tests).  Per-pair seeds are a function of
the global pair id only, so results do not depend on the shard count.
"""
import numpy as np


def shard_range(n_pairs, rank, world):
    """Contiguous block [lo, hi) of pair ids owned by `rank` (first n_pairs % world ranks get one more)."""
    base, rem = divmod(n_pairs, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def pair_seed(pair_id, base_seed=1):
    """RANSAC seed of a pair: depends on the global pair id only (Knuth multiplicative hash)."""
    return (int(base_seed) + 2654435761 * (int(pair_id) + 1)) & 0x7FFFFFFF


def pair_seeds(lo, hi, base_seed=1):
    return np.array([pair_seed(p, base_seed) for p in range(lo, hi)], dtype=np.uint32)


def gather_results(models, stats, masks, n_per_pair, n_pairs_total, group=None):
    """All-gather the per-pair results of every rank (torch tensors on the rank's device).

    models [P_r, 9] float64, stats [P_r, 16] int32, masks [P_r * n_per_pair] uint8 for equally sized
    pairs.  Ranks may own different numbers of pairs (padded to the maximum for the collective).
    Returns (models [P,9], stats [P,16], masks [P, n_per_pair]) in global pair order.
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return models, stats, masks.view(-1, n_per_pair)
    pmax = (n_pairs_total + world - 1) // world
    dev = models.device
    rec = 72 + 64 + n_per_pair                                  # bytes per pair
    packed = torch.zeros((pmax, rec), dtype=torch.uint8, device=dev)
    p_r = models.shape[0]
    packed[:p_r, :72] = models.contiguous().view(torch.uint8).view(p_r, 72)
    packed[:p_r, 72:136] = stats.contiguous().view(torch.uint8).view(p_r, 64)
    packed[:p_r, 136:] = masks.view(p_r, n_per_pair)
    out = torch.empty((world, pmax, rec), dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(out.view(-1), packed.view(-1), group=group)
    rows = []
    for r in range(world):
        lo, hi = shard_range(n_pairs_total, r, world)
        rows.append(out[r, :hi - lo])
    allp = torch.cat(rows, 0)
    m = allp[:, :72].contiguous().view(torch.float64).view(-1, 9)
    s = allp[:, 72:136].contiguous().view(torch.int32).view(-1, 16)
    return m, s, allp[:, 136:]
