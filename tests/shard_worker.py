"""Worker of the world_size>1 tests: run under torch.distributed.run.

Every rank takes its shard of one seeded ragged batch, codes it (with the HIP library when
--backend gpu, else with the C oracle standing in for the device), and rank 0 checks that the
shards put back together equal the whole batch coded at once by the oracle.
"""
import argparse
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "gorilla-tsc_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="cpu")
    ap.add_argument("--out", required=True)
    args = ap.parse_args()
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()

    import pyoracle as O
    from pygtsc import shard, synth
    b = synth.ragged_mix(seed=11, nseries=301, pmax=400)
    blk, ts, val, off = shard.shard_of(rank, world, b.block_ts, b.ts, b.val, b.pt_off)
    S = len(blk)
    if args.backend == "gpu":
        import pygtsc as G
        ctx = G.Context(0)
        words, woff = ctx.encode_host(blk, ts, val, pt_off=off)
        ts2, val2, npts, st = ctx.decode_host(words, woff, pt_off=off)
        ctx.close()
    else:
        out, ooff, nw, st = O.encode_batch(blk, ts, val, off, nthreads=1)
        # compact the slots the way gtsc_compact_device does
        woff = np.zeros(S + 1, dtype=np.uint64)
        woff[1:] = np.cumsum(nw)
        words = np.concatenate([out[int(ooff[s]):int(ooff[s]) + int(nw[s])] for s in range(S)]) if S else np.zeros(0, np.uint64)
        ts2, val2, npts, st2 = O.decode_batch(words, woff, nw, off, nthreads=1)
    assert np.array_equal(ts2, ts) and np.array_equal(val2, val), "rank %d: round trip" % rank

    # timing reduction: the job time is the slowest rank's
    tmax = shard.max_over_ranks(1.0 + rank, dist)
    npoints = shard.sum_over_ranks(len(ts), dist)
    gathered = [None] * world
    dist.all_gather_object(gathered, (words, woff))
    ok = True
    if rank == 0:
        allw, alloff = shard.concat_images([g[0] for g in gathered], [g[1] for g in gathered])
        ref, roff, rnw, _ = O.encode_batch(b.block_ts, b.ts, b.val, b.pt_off, nthreads=2)
        refw = np.concatenate([ref[int(roff[s]):int(roff[s]) + int(rnw[s])] for s in range(b.nseries)])
        ok = bool(np.array_equal(allw, refw)) and int(alloff[-1]) == len(refw) and len(alloff) == b.nseries + 1
        ok = ok and tmax == float(world) and int(npoints) == b.npoints
        json.dump({"ok": ok, "world": world, "tmax": tmax, "npoints": int(npoints), "words": int(len(allw))}, open(args.out, "w"))
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
