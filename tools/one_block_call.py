#!/usr/bin/env python3
"""Device-resident decompress calls of one uniform batch of few long blocks, for a kernel trace of a lone call:
usage: one_block_call.py [random_walk|readme|gauge] [S] [P] [calls]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gorilla-tsc_amd")]
import numpy as np, torch
import pygtsc as G
from pygtsc import synth

kind = sys.argv[1] if len(sys.argv) > 1 else "random_walk"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1
P = int(sys.argv[3]) if len(sys.argv) > 3 else 7200
calls = int(sys.argv[4]) if len(sys.argv) > 4 else 20
b = {"random_walk": lambda: synth.random_walk(S=S, P=P), "readme": lambda: synth.readme_workload(S=S, P=P),
     "gauge": lambda: synth.gauge_integer(S=S, P=P)}[kind]()
d = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()
ctx = G.Context(0)
d_block, d_ts, d_val = d(b.block_ts), d(b.ts), d(b.val)
slot = G.max_words(P)
d_out = torch.zeros(S * slot, dtype=torch.int64, device="cuda"); d_nw = torch.zeros(S, dtype=torch.int64, device="cuda")
d_ts2, d_val2 = torch.zeros_like(d_ts), torch.zeros_like(d_val); d_np = torch.zeros(S, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
ctx.encode_device(d_block, d_ts, d_val, d_out, d_nw, uniform_npoints=P, uniform_slot_words=slot); ctx.sync()
best = 1e9
for _ in range(calls):
    t0 = time.perf_counter()
    ctx.decode_device(d_out, d_ts2, d_val2, d_np, S, uniform_slot_words=slot, in_nwords=d_nw, uniform_capacity=P)
    ctx.sync()
    best = min(best, time.perf_counter() - t0)
assert torch.equal(d_ts2, d_ts) and torch.equal(d_val2, d_val)
print("%s %d x %d: decompress call %.1f us (best of %d)" % (kind, S, P, best * 1e6, calls))
ctx.close()
