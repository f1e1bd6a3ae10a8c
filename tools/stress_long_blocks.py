#!/usr/bin/env python3
"""Randomised parity sweep of the long-block decompressors on the GPU (diagnostic): batches of a few blocks of 1,000 to
60,000 points whose character changes along the block (constant stretches, gauges, counters, random walks, jittered or
regular timestamps, zero timestamps), both formats, capacities exact / short / generous, images cut short, with zeroed
tails and flipped bits -- every count, status word and point compared with the CPU oracle.  Which kernel takes the
blocks follows from the environment (GTSC_WG_DECODE_MAX, GTSC_WG_DENSE_BITS, GTSC_LONG_DECODE_MAX, ...).
usage: python tools/stress_long_blocks.py [seconds] [first seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gorilla-tsc_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import torch  # noqa: F401

import pygtsc as G
import pyoracle as O
from pygtsc import synth
from test_gpu_parity import check_decode_against_oracle, check_encode_against_oracle

FMTS = [(G.FMT_V2, O.V2), (G.FMT_V1, O.V1)]


def values_of(kind, n, rng):
    if kind == 0:
        return np.full(n, rng.uniform(0, 1000))
    if kind == 1:
        return 500.0 + np.cumsum(rng.integers(-1, 2, n) * (rng.random(n) < rng.choice([0.02, 0.1, 0.5])))
    if kind == 2:
        return (rng.integers(0, 1000) + np.cumsum(rng.integers(0, 5, size=n))).astype(np.float64)
    if kind == 3:
        return 100.0 + np.cumsum(rng.standard_normal(n))
    if kind == 4:
        return rng.integers(0, 2 ** 63, n).astype(np.uint64).view(np.float64)
    return np.arange(n, dtype=np.float64) * rng.choice([1.0, 0.5, 3.0])


def make_block(rng, n):
    """Values and timestamps in stretches of different character."""
    vals, deltas = [], []
    left = n
    while left > 0:
        m = int(min(left, rng.integers(1, max(2, n // rng.integers(1, 6)) + 1)))
        vals.append(values_of(int(rng.integers(0, 6)), m, rng))
        tk = rng.integers(0, 4)
        if tk == 0:
            d = np.full(m, 1000)
        elif tk == 1:
            d = 1000 + rng.integers(-40, 41, m)
        elif tk == 2:
            d = 1000 + (rng.random(m) < 0.02) * rng.integers(1, 3000, m)
        else:
            d = rng.integers(1, 100000, m)
        deltas.append(d.astype(np.int64))
        left -= m
    return np.concatenate(vals), np.concatenate(deltas)


def make_batch(rng):
    S = int(rng.integers(1, 7))
    lens = rng.integers(1000, rng.choice([3000, 12000, 60000]), S)
    pt_off = np.zeros(S + 1, dtype=np.uint64)
    np.cumsum(lens, out=pt_off[1:])
    tot = int(pt_off[-1])
    ts = np.empty(tot, dtype=np.int64)
    val = np.empty(tot, dtype=np.float64)
    block = np.empty(S, dtype=np.int64)
    for s in range(S):
        a, n = int(pt_off[s]), int(lens[s])
        v, d = make_block(rng, n)
        # now and then a block whose timestamps pass through zero (the next point is read as a "first" one again)
        block[s] = -int(d[: rng.integers(1, n)].sum()) if rng.random() < 0.05 else synth.BLOCK_TS
        ts[a:a + n] = block[s] + np.cumsum(d)
        val[a:a + n] = v
    return synth.Batch(block, ts, np.ascontiguousarray(val).view(np.uint64), pt_off)


def run_case(ctx, seed):
    """One random batch through compression and decompression in both formats, with good and damaged images and
    three kinds of capacities; returns (cases, points, dispatch bits seen)."""
    rng = np.random.default_rng(seed)
    b = make_batch(rng)
    S = b.nseries
    lens = np.diff(b.pt_off.astype(np.int64))
    seen = cases = 0
    for gfmt, ofmt in FMTS:
        out, off, nw, d_out, _ = check_encode_against_oracle(ctx, b, gfmt, ofmt)
        caps = [b.pt_off]
        short = np.maximum(lens - rng.integers(0, 3000, S), 1)
        generous = lens + rng.integers(0, 3000, S)
        for c in (short, generous):
            o = np.zeros(S + 1, dtype=np.uint64)
            np.cumsum(c, out=o[1:])
            caps.append(o)
        for cap_off in caps:
            check_decode_against_oracle(ctx, out, off, nw, cap_off, gfmt, ofmt)
            seen |= ctx.last_dispatch()
        cut = nw.copy()
        cut[::2] = cut[::2] * rng.integers(1, 100) // 100
        check_decode_against_oracle(ctx, out, off, cut, b.pt_off, gfmt, ofmt)
        zeroed = out.copy()
        for s in range(S):
            a, n = int(off[s]), int(nw[s])
            zeroed[a + int(n * rng.random()): a + n] = 0
        check_decode_against_oracle(ctx, zeroed, off, nw, b.pt_off, gfmt, ofmt)
        flipped = out.copy()
        total = int(off[-1])
        hits = rng.integers(0, total, int(rng.integers(1, 6)))
        flipped[hits] ^= np.uint64(1) << rng.integers(0, 64, len(hits)).astype(np.uint64)
        check_decode_against_oracle(ctx, flipped, off, nw, caps[2], gfmt, ofmt)
        cases += 6
    return cases, cases * int(b.pt_off[-1]), seen


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    ctx = G.Context(0)
    t0 = time.time()
    cases = points = seen = 0
    while time.time() - t0 < seconds:
        try:
            c, p, k = run_case(ctx, seed)
        except AssertionError:
            print("FAILED at seed %d" % seed, flush=True)
            raise
        cases += c
        points += p
        seen |= k
        seed += 1
        if seed % 20 == 0:
            print("... seed %d, %d cases, %.1f M points, kernels seen 0x%x" % (seed, cases, points / 1e6, seen), flush=True)
    print("long blocks ok: %d cases (seeds up to %d), %.1f M points, kernels seen 0x%x" % (cases, seed - 1, points / 1e6, seen))
    ctx.close()


if __name__ == "__main__":
    main()
