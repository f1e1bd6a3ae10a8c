#!/usr/bin/env python3
"""Kernel time of one long block per value class of the skewed workload (diagnostic)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gorilla-tsc_amd")]
import numpy as np, torch
import pygtsc as G
from pygtsc import synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
rng = np.random.default_rng(3)
classes = {
    "constant": np.full(N, 123.456),
    "counter": (rng.integers(0, 1000) + np.cumsum(rng.integers(0, 5, size=N))).astype(np.float64),
    "random_walk": 100.0 + np.cumsum(rng.standard_normal(N)),
    "readme": np.arange(N, dtype=np.float64),   # G/benchmark/EncodingBenchmark.java:60-69: value = (double) i
    "drift": np.round((rng.random() * 1000 + np.cumsum(rng.standard_normal(N) * 0.05)) * 100) / 100,
    "gauge": 500.0 + np.cumsum(rng.integers(-1, 2, N) * (rng.random(N) < 0.1)),
    "random_bits": rng.integers(0, 2 ** 63, N).astype(np.uint64).view(np.float64),
}
# the same values at scrape times that are a few ms off the cadence (timestamp codes of 9 to 16 bits instead of '0')
classes["jitter_gauge"] = classes["gauge"]
classes["jitter_walk"] = classes["random_walk"]
classes["jitter_counter"] = classes["counter"]
ctx = G.Context(0)
for name, v in classes.items():
    ts = synth.BLOCK_TS + 1000 * np.arange(1, N + 1, dtype=np.int64)
    if name.startswith("jitter_"):
        ts = ts + rng.integers(-40, 41, N)
    val = np.ascontiguousarray(v).view(np.uint64)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()
    d_block, d_ts, d_val = d(np.array([synth.BLOCK_TS], dtype=np.int64)), d(ts), d(val)
    slot = G.max_words(N)
    d_out = torch.zeros(slot, dtype=torch.int64, device="cuda"); d_nw = torch.zeros(1, dtype=torch.int64, device="cuda")
    d_ts2 = torch.zeros_like(d_ts); d_val2 = torch.zeros_like(d_val); d_np = torch.zeros(1, dtype=torch.int64, device="cuda")
    po = d(np.array([0, N], dtype=np.uint64)); so = d(np.array([0, slot], dtype=np.uint64))
    torch.cuda.synchronize()
    for k in range(3):
        t0 = time.perf_counter()
        ctx.encode_device(d_block, d_ts, d_val, d_out, d_nw, pt_off=po, out_off=so); ctx.sync()
        t1 = time.perf_counter()
        ctx.decode_device(d_out, d_ts2, d_val2, d_np, 1, in_off=so, in_nwords=d_nw, pt_off=po); ctx.sync()
        t2 = time.perf_counter()
    assert torch.equal(d_ts2, d_ts) and torch.equal(d_val2, d_val)
    print("%-12s %6.1f bits/pt  encode %7.3f ms  decode %7.3f ms (%.0f ns/pt)" % (
        name, 64.0 * int(d_nw[0]) / N, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t2 - t1) * 1e9 / N))
ctx.close()
