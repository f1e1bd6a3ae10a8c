"""Sharding of a batch of series over ranks (one rank per GPU) and the timing reduction.

Series blocks are independent (one reference compressor instance per block,
G/GorillaCompressor.java:13-26), so the only multi-GPU logic is: who codes which
series, and how the per-rank results and timings are put together.  No data moves
between ranks on the codec path.
"""
import numpy as np


def shard_bounds(pt_off, world):
    """Cuts series [0, S) into `world` contiguous ranges with nearly equal point counts.

    pt_off: uint64[S+1] CSR offsets (or an int S for unit-weight series).
    Returns int64[world+1] series indices b with b[0]=0, b[world]=S, non-decreasing:
    rank r owns series [b[r], b[r+1]).
    """
    if np.isscalar(pt_off):
        S = int(pt_off)
        return np.array([(S * r) // world for r in range(world + 1)], dtype=np.int64)
    off = np.asarray(pt_off, dtype=np.uint64).astype(np.int64)
    S = len(off) - 1
    total = int(off[-1])
    # the series whose start offset is nearest at or above each ideal cut
    ideal = np.array([(total * r) // world for r in range(world + 1)], dtype=np.int64)
    b = np.searchsorted(off, ideal, side="left").astype(np.int64)
    b[0], b[-1] = 0, S
    return np.minimum(np.maximum.accumulate(b), S)


def shard_of(rank, world, block_ts, ts, val, pt_off):
    """The slice of a host batch that `rank` codes: (block_ts, ts, val, pt_off rebased to 0)."""
    b = shard_bounds(pt_off, world)
    s0, s1 = int(b[rank]), int(b[rank + 1])
    p0, p1 = int(pt_off[s0]), int(pt_off[s1])
    off = (np.asarray(pt_off[s0:s1 + 1], dtype=np.uint64) - np.uint64(p0)).astype(np.uint64)
    return block_ts[s0:s1], ts[p0:p1], val[p0:p1], off


def max_over_ranks(seconds, dist=None, device=None):
    """Job time = slowest rank.  `dist` is torch.distributed (initialised) or None for one rank."""
    if dist is None:
        return float(seconds)
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(x, dist=None, device=None):
    """Whole-job count (points, bytes) from per-rank counts."""
    if dist is None:
        return float(x)
    import torch
    t = torch.tensor([float(x)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def concat_images(words_per_rank, off_per_rank):
    """Host-side merge of per-rank compact outputs into one (words, off) pair, in rank order."""
    words = np.concatenate([np.asarray(w, dtype=np.uint64) for w in words_per_rank]) if words_per_rank else np.zeros(0, np.uint64)
    offs = [np.zeros(1, dtype=np.uint64)]
    base = np.uint64(0)
    for o in off_per_rank:
        o = np.asarray(o, dtype=np.uint64)
        offs.append(o[1:] + base)
        base = base + o[-1]
    return words, np.concatenate(offs)
