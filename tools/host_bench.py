#!/usr/bin/env python3
"""Host-to-host rate of gtsc_encode_host / gtsc_decode_host (PCIe copies included).

Inputs and outputs live in pinned host memory from gtsc_host_alloc; the library stages
them through the GPU in chunks on two streams.  Workloads: "skewed" (log-uniform series
lengths, mixed value classes) and "gauge" (uniform 120-point series).
usage: python tools/host_bench.py [skewed|gauge] [total_points_log2]
"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gorilla-tsc_amd"))
import numpy as np  # noqa: E402

import pygtsc as G  # noqa: E402
from pygtsc import synth  # noqa: E402


def pinned(n, dtype):
    """numpy view over n items of pinned host memory"""
    nbytes = max(int(n), 1) * np.dtype(dtype).itemsize
    p = G.lib().gtsc_host_alloc(nbytes)
    if not p:
        raise MemoryError("gtsc_host_alloc(%d)" % nbytes)
    buf = (C.c_uint8 * nbytes).from_address(p)
    a = np.frombuffer(buf, dtype=dtype, count=int(n))
    return a, p


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "skewed"
    lg = int(sys.argv[2]) if len(sys.argv) > 2 else 26
    if kind == "skewed":
        b = synth.skewed(total_points=1 << lg)
    else:
        b = synth.gauge_integer(S=(1 << lg) // 120, P=120)
    S, N = b.nseries, b.npoints
    L = G.lib()
    ctx = G.Context(0)
    lens = np.diff(b.pt_off.astype(np.int64))
    mw = {int(p): G.max_words(int(p)) for p in np.unique(lens)}
    cap = int(sum(mw[int(p)] for p in lens))
    ptrs = []

    def pin_copy(a, dtype):
        v, p = pinned(len(a), dtype)
        v[:] = a
        ptrs.append(p)
        return v

    blk = pin_copy(b.block_ts, np.int64)
    ts = pin_copy(b.ts, np.int64)
    val = pin_copy(b.val, np.uint64)
    off = pin_copy(b.pt_off, np.uint64)
    out, p = pinned(cap, np.uint64); ptrs.append(p)
    out_off, p = pinned(S + 1, np.uint64); ptrs.append(p)
    ts2, p = pinned(N, np.int64); ptrs.append(p)
    val2, p = pinned(N, np.uint64); ptrs.append(p)
    npts, p = pinned(S, np.uint64); ptrs.append(p)
    st, p = pinned(S, np.uint32); ptrs.append(p)
    hp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731

    def enc():
        rc = L.gtsc_encode_host(ctx._h, G.FMT_V2, S, hp(blk), hp(ts), hp(val), hp(off), 0, hp(out), cap, hp(out_off))
        assert rc == 0, L.gtsc_last_error()

    def dec():
        rc = L.gtsc_decode_host(ctx._h, G.FMT_V2, S, hp(out), hp(out_off), hp(ts2), hp(val2), hp(off), 0, hp(npts), hp(st))
        assert rc == 0, L.gtsc_last_error()

    enc(); dec()
    assert np.array_equal(ts2, b.ts) and np.array_equal(val2, b.val), "round trip failed"
    best_e = best_d = 1e30
    for _ in range(3):
        t0 = time.perf_counter(); enc(); t1 = time.perf_counter(); dec(); t2 = time.perf_counter()
        best_e, best_d = min(best_e, t1 - t0), min(best_d, t2 - t1)
    raw = 16.0 * N
    print(json.dumps({
        "workload": kind, "series": S, "points": N, "raw_MB": round(raw / 1e6, 1),
        "compressed_MB": round(8.0 * int(out_off[S]) / 1e6, 1),
        "encode_host_GBps": round(raw / best_e / 1e9, 2), "decode_host_GBps": round(raw / best_d / 1e9, 2),
        "encode_ms": round(best_e * 1e3, 2), "decode_ms": round(best_d * 1e3, 2),
    }))
    for p in ptrs:
        L.gtsc_host_free(p)
    ctx.close()


if __name__ == "__main__":
    main()
