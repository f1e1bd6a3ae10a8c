#!/usr/bin/env python3
"""Randomised parity sweep on the GPU (diagnostic, not part of the test suite): many ragged and
uniform batches, both formats, arrays that start at every 8-byte phase of a 128-byte line,
tight and loose slots, truncated images -- every result compared with the CPU oracle.
usage: python tools/stress_parity.py [seconds] [first seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gorilla-tsc_amd"), os.path.join(ROOT, "oracle")]
import numpy as np
import torch

import pygtsc as G
import pyoracle as O
from pygtsc import synth

FMTS = [(G.FMT_V2, O.V2), (G.FMT_V1, O.V1)]


def dev_at_phase(a, phase, fill=0):
    """A device tensor holding `a` that starts `phase` 8-byte elements into a fresh allocation."""
    a = np.ascontiguousarray(a)
    if a.dtype == np.uint64:
        a = a.view(np.int64)
    t = torch.full((len(a) + phase + 16,), fill, dtype=torch.int64, device="cuda")
    t[phase:phase + len(a)] = torch.from_numpy(a).cuda()
    return t[phase:phase + len(a)] if len(a) else t[phase:phase], t


def make_batch(rng):
    kind = rng.integers(0, 5)
    if kind == 0:
        return synth.ragged_mix(seed=int(rng.integers(1, 1 << 30)), nseries=int(rng.integers(6, 700)), pmax=int(rng.integers(2, 400)))
    if kind == 1:
        S, P = int(rng.integers(1, 600)), int(rng.integers(1, 260))
        b = synth.gauge_integer(S=S, P=P, seed=int(rng.integers(1, 1 << 30)))
        if rng.integers(0, 2):
            # late samples in some rows: intervals that change and come back inside runs of unchanged values
            every = int(rng.choice([2, 5, 20, 100, 1000]))
            late = (rng.random(b.ts.size) < 1.0 / every) & np.repeat(rng.random(S) < 0.5, P)
            ts = b.ts + np.where(late, rng.integers(1, 3000, size=b.ts.size), 0)
            b = synth.Batch(b.block_ts, ts, b.val, b.pt_off)
        return b
    if kind == 2:
        S, P = int(rng.integers(1, 300)), int(rng.integers(1, 200))
        return synth.adversarial(S=S, P=P, seed=int(rng.integers(1, 1 << 30)))
    if kind == 3:
        S, P = int(rng.integers(1, 400)), int(rng.integers(1, 300))
        return synth.gauge_drift(S=S, P=P, seed=int(rng.integers(1, 1 << 30)))
    return synth.skewed(total_points=int(rng.integers(2000, 60000)), seed=int(rng.integers(1, 1 << 30)), pmin=1, pmax=int(rng.integers(50, 5000)))


def one_case(ctx, seed):
    rng = np.random.default_rng(seed)
    b = make_batch(rng)
    S = b.nseries
    gfmt, ofmt = FMTS[int(rng.integers(0, 2))]
    lens = np.diff(b.pt_off.astype(np.int64))
    P = b.uniform_npoints()
    uniform = P is not None and P > 0 and bool(rng.integers(0, 2))
    ph = [int(x) for x in rng.integers(0, 16, size=6)]
    d_block, _k0 = dev_at_phase(b.block_ts, ph[0])
    d_ts, _k1 = dev_at_phase(b.ts, ph[1])
    d_val, _k2 = dev_at_phase(b.val, ph[2])
    d_nw = torch.zeros(S, dtype=torch.int64, device="cuda")
    d_st = torch.zeros(S, dtype=torch.int32, device="cuda")
    if uniform:
        slot = G.max_words(P) + int(rng.integers(0, 3))
        off = np.arange(S + 1, dtype=np.uint64) * np.uint64(slot)
    else:
        bits = 64 + np.where(lens > 0, 91 + 114 * (lens - 1), 0) + 37
        words = bits // 64 + 2 + rng.integers(0, 3, size=S)
        off = np.zeros(S + 1, dtype=np.uint64)
        np.cumsum(words, out=off[1:])
    d_out, _k3 = dev_at_phase(np.full(int(off[-1]), -1, dtype=np.int64), ph[3], fill=-1)
    torch.cuda.synchronize()
    if uniform:
        ctx.encode_device(d_block, d_ts, d_val, d_out, d_nw, uniform_npoints=P, uniform_slot_words=int(off[1]), status=d_st, fmt=gfmt)
    else:
        d_po = torch.from_numpy(b.pt_off.view(np.int64)).cuda()
        d_oo = torch.from_numpy(off.view(np.int64)).cuda()
        ctx.encode_device(d_block, d_ts, d_val, d_out, d_nw, pt_off=d_po, out_off=d_oo, status=d_st, fmt=gfmt)
    ctx.sync()
    out = d_out.cpu().numpy().view(np.uint64)
    nw = d_nw.cpu().numpy().view(np.uint64)
    ref_out, _, ref_nw, ref_st = O.encode_batch(b.block_ts, b.ts, b.val, b.pt_off, out_off=off, fmt=ofmt, nthreads=8)
    assert np.array_equal(nw, ref_nw), "image lengths"
    assert not d_st.cpu().numpy().any() and not ref_st.any()
    for s in range(S):
        a = int(off[s])
        assert np.array_equal(out[a:a + int(nw[s])], ref_out[a:a + int(nw[s])]), ("words of series", s)
        assert np.all(out[a + int(nw[s]):int(off[s + 1])] == np.uint64(2 ** 64 - 1)), ("slack of series", s)
    # decode: whole images, or with some of them cut short
    in_nw = nw.copy()
    if rng.integers(0, 3) == 0 and S:
        cut = rng.random(S) < 0.3
        in_nw[cut] = (in_nw[cut] * rng.random(int(cut.sum()))).astype(np.uint64)
    caps = lens + rng.integers(0, 3, size=S) if rng.integers(0, 2) else np.maximum(lens - rng.integers(0, 4, size=S), 0)
    if uniform and rng.integers(0, 2):
        caps = np.full(S, int(caps.max()) if S else 0)
        ucap = int(caps[0]) if S else 0
    else:
        ucap = None
    cap_off = np.zeros(S + 1, dtype=np.uint64)
    np.cumsum(caps, out=cap_off[1:])
    total = int(cap_off[-1])
    if total == 0:
        return b.npoints  # nothing to decode into
    d_ts2, _k4 = dev_at_phase(np.full(total, -7, dtype=np.int64), ph[4], fill=-7)
    d_val2, _k5 = dev_at_phase(np.full(total, -7, dtype=np.int64), ph[5], fill=-7)
    d_np = torch.zeros(S, dtype=torch.int64, device="cuda")
    d_st2 = torch.zeros(S, dtype=torch.int32, device="cuda")
    d_inw = torch.from_numpy(in_nw.view(np.int64)).cuda()
    torch.cuda.synchronize()
    kw = dict(in_nwords=d_inw, status=d_st2, fmt=gfmt)
    if uniform:
        kw["uniform_slot_words"] = int(off[1])
    else:
        kw["in_off"] = torch.from_numpy(off.view(np.int64)).cuda()
    if ucap is not None and ucap > 0:
        kw["uniform_capacity"] = ucap
    else:
        kw["pt_off"] = torch.from_numpy(cap_off.view(np.int64)).cuda()
    ctx.decode_device(d_out, d_ts2, d_val2, d_np, S, **kw)
    try:
        ctx.sync()
    except G.GtscError:
        pass
    rts, rval, rnp, rst = O.decode_batch(out, off, in_nw, cap_off, fmt=ofmt)
    npts = d_np.cpu().numpy().view(np.uint64)
    st = d_st2.cpu().numpy().view(np.uint32)
    assert np.array_equal(npts, rnp) and np.array_equal(st, rst), "decode counts/status"
    gts = d_ts2.cpu().numpy()
    gval = d_val2.cpu().numpy().view(np.uint64)
    for s in range(S):
        a0, a1 = int(cap_off[s]), int(cap_off[s + 1])
        n = min(int(npts[s]), a1 - a0)
        assert np.array_equal(gts[a0:a0 + n], rts[a0:a0 + n]) and np.array_equal(gval[a0:a0 + n], rval[a0:a0 + n]), ("points of series", s)
        assert np.all(gts[a0 + n:a1] == -7), ("slack of series", s)
    # nothing outside the arrays was touched
    for keep, phase, n, fill in ((_k3, ph[3], int(off[-1]), -1), (_k4, ph[4], total, -7), (_k5, ph[5], total, -7)):
        h = keep.cpu().numpy()
        assert np.all(h[:phase] == fill) and np.all(h[phase + n:] == fill), "guard words"
    return b.npoints


def main():
    BUDGET = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
    SEED0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    assert torch.cuda.is_available()
    ctx = G.Context(0)
    t0 = time.time()
    seed, pts, last = SEED0, 0, t0
    while time.time() - t0 < BUDGET:
        try:
            pts += one_case(ctx, seed)
        except Exception:
            print("FAILED at seed", seed, flush=True)
            raise
        seed += 1
        if time.time() - last > 30:
            last = time.time()
            print("%d cases, %.1f M points, %.0f s" % (seed - SEED0, pts / 1e6, last - t0), flush=True)
    print("stress ok: %d cases (seeds %d..%d), %.1f M points" % (seed - SEED0, SEED0, seed - 1, pts / 1e6), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
