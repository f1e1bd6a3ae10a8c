#!/usr/bin/env python3
"""How long one long block takes the compressor alone in its wave and among 63 equally long ones
(diagnostic): the chain of one block is the floor for batches of very unequal blocks."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gorilla-tsc_amd")]
import numpy as np, torch
import pygtsc as G
from pygtsc import synth

def run(lens, label):
    S = len(lens)
    pt_off = np.zeros(S + 1, dtype=np.uint64); np.cumsum(lens, out=pt_off[1:])
    tot = int(pt_off[-1])
    rng = np.random.default_rng(1)
    ts = np.concatenate([synth.BLOCK_TS + 1000 * np.arange(1, n + 1, dtype=np.int64) for n in lens])
    val = (100.0 + np.cumsum(rng.integers(-1, 2, tot) * (rng.random(tot) < 0.1))).view(np.uint64)
    words = np.array([G.max_words(int(n)) for n in lens], dtype=np.uint64)
    off = np.zeros(S + 1, dtype=np.uint64); np.cumsum(words, out=off[1:])
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()
    d_block = d(np.full(S, synth.BLOCK_TS, dtype=np.int64)); d_ts = d(ts); d_val = d(val); d_po = d(pt_off); d_oo = d(off)
    d_out = torch.zeros(int(off[-1]), dtype=torch.int64, device="cuda"); d_nw = torch.zeros(S, dtype=torch.int64, device="cuda")
    d_ts2 = torch.zeros_like(d_ts); d_val2 = torch.zeros_like(d_val); d_np = torch.zeros(S, dtype=torch.int64, device="cuda")
    ctx = G.Context(0)
    torch.cuda.synchronize()
    for k in range(3):
        t0 = time.perf_counter()
        ctx.encode_device(d_block, d_ts, d_val, d_out, d_nw, pt_off=d_po, out_off=d_oo); ctx.sync()
        t1 = time.perf_counter()
        ctx.decode_device(d_out, d_ts2, d_val2, d_np, S, in_off=d_oo, in_nwords=d_nw, pt_off=d_po); ctx.sync()
        t2 = time.perf_counter()
    n = int(max(lens))
    print("%-34s encode %.2f ms (%.0f ns/point of the longest block), decode %.2f ms (%.0f ns/point)" % (
        label, (t1 - t0) * 1e3, (t1 - t0) * 1e9 / n, (t2 - t1) * 1e3, (t2 - t1) * 1e9 / n))
    assert torch.equal(d_ts2, d_ts)
    ctx.close()

N = 100000
run([N] * 64, "64 blocks of 100,000 points")
run([N] + [10] * 63, "one of 100,000 among 63 of 10")
run([N] + list(range(1000, 64000, 1000)), "one of 100,000 among 1,000..63,000")
