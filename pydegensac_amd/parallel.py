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


def gather_results(models, stats, masks, n_per_pair, n_pairs_total, group=None, always_collective=False):
    """All-gather the per-pair results of every rank (torch tensors on the rank's device).

    models [P_r, 9] float64, stats [P_r, 16] int32, masks [sum of the rank's pair sizes] uint8.
    `n_per_pair`: an int (every pair has that many correspondences) or the sequence of ALL n_pairs_total pair sizes
    (ragged batch, the C-ABI's offsets form).  Ranks may own different numbers of pairs / bytes: equal shards take one
    all_gather_into_tensor of the packed records; ragged shards gather the fixed 136 B per pair (padded by at most one
    pair) and ship each rank's masks at their own size (one broadcast per rank), never padded to the largest shard.
    Returns (models [P,9], stats [P,16], masks) in global pair order, where masks is
    [P, n] for equal sizes and a flat [sum of all sizes] uint8 tensor for a ragged batch (pair p at
    offsets[p]:offsets[p+1] with offsets = cumsum of the sizes).
    always_collective: run the all-gather even in a one-rank group (exercises the RCCL path on a single GPU).
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    ragged = not isinstance(n_per_pair, (int, np.integer))
    counts = np.asarray(n_per_pair, dtype=np.int64).ravel() if ragged else np.full(n_pairs_total, int(n_per_pair), np.int64)
    if len(counts) != n_pairs_total:
        raise ValueError("need one size per pair")
    if world == 1 and not (always_collective and dist.is_initialized()):
        return models, stats, (masks if ragged else masks.view(-1, int(n_per_pair)))
    dev = models.device
    rng = [shard_range(n_pairs_total, r, world) for r in range(world)]
    mbytes = [int(counts[lo:hi].sum()) for lo, hi in rng]
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    p_r = models.shape[0]
    if p_r != rng[rank][1] - rng[rank][0] or masks.numel() != mbytes[rank]:
        raise ValueError("this rank's tensors do not match its shard of the batch")
    rec = [(hi - lo) * 136 + mb for (lo, hi), mb in zip(rng, mbytes)]      # 72 B model + 64 B stats per pair, then the masks
    if len({hi - lo for lo, hi in rng}) == 1 and len(set(mbytes)) == 1:
        # every rank holds the same number of pairs AND of mask bytes (C4: equal pairs, equal sizes): ONE collective over the
        # packed records.  Equal record sizes alone are not enough: 2 pairs of 100 and 1 pair of 336 both pack to 472 bytes,
        # and the slices below use this rank's own pair count for every rank's record.
        cap = rec[0]
        packed = torch.empty(cap, dtype=torch.uint8, device=dev)
        packed[:p_r * 72] = models.contiguous().view(torch.uint8).view(-1)
        packed[p_r * 72:p_r * 136] = stats.contiguous().view(torch.uint8).view(-1)
        packed[p_r * 136:] = masks.view(-1)
        out = torch.empty((world, cap), dtype=torch.uint8, device=dev)
        dist.all_gather_into_tensor(out.view(-1), packed, group=group)
        ms = [out[r, :p_r * 72] for r in range(world)]; ss = [out[r, p_r * 72:p_r * 136] for r in range(world)]
        ks = [out[r, p_r * 136:] for r in range(world)]
    else:
        # ragged shards: the fixed-size part (136 B per pair; shards differ by at most one pair) goes through one padded
        # all-gather, the masks travel at their own size: every rank knows every shard's byte count from the global size
        # list, so rank r's masks are one broadcast of exactly mbytes[r] bytes (no padding to the largest shard: a batch
        # with one 50 000-correspondence pair does not make every rank ship that pair's size)
        pmax = max(hi - lo for lo, hi in rng)
        fixed = torch.zeros(pmax * 136, dtype=torch.uint8, device=dev)
        fixed[:p_r * 72] = models.contiguous().view(torch.uint8).view(-1)
        fixed[pmax * 72:pmax * 72 + p_r * 64] = stats.contiguous().view(torch.uint8).view(-1)
        out = torch.empty((world, pmax * 136), dtype=torch.uint8, device=dev)
        dist.all_gather_into_tensor(out.view(-1), fixed, group=group)
        ks = [masks.contiguous().view(-1) if r == rank else torch.empty(mbytes[r], dtype=torch.uint8, device=dev) for r in range(world)]
        pending = []
        for r in range(world):
            if mbytes[r] > 0:
                src = dist.get_global_rank(group, r) if group is not None else r
                pending.append(dist.broadcast(ks[r], src=src, group=group, async_op=True))
        for w in pending:
            w.wait()
        ms = [out[r, :(rng[r][1] - rng[r][0]) * 72] for r in range(world)]
        ss = [out[r, pmax * 72:pmax * 72 + (rng[r][1] - rng[r][0]) * 64] for r in range(world)]
    m = torch.cat(ms).contiguous().view(torch.float64).view(-1, 9)
    st = torch.cat(ss).contiguous().view(torch.int32).view(-1, 16)
    k = torch.cat(ks)
    return m, st, (k if ragged else k.view(-1, int(n_per_pair)))
