#!/usr/bin/env python3
"""Per-step kernel times of the default workload (diagnostic): shows the steps the GPU takes to reach steady clocks."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gorilla-tsc_amd")]
import torch, numpy as np
import pygtsc as G, bench
S, P = 1_000_000, 120
dev = torch.device("cuda", 0)
block, ts, val = bench.make_workload("gauge", S, P, dev, 0x5EED0002)
slot = G.max_words(P)
out = torch.zeros(S * slot, dtype=torch.int64, device=dev); nw = torch.zeros(S, dtype=torch.int64, device=dev)
ts2 = torch.zeros_like(ts); val2 = torch.zeros_like(val); npts = torch.zeros(S, dtype=torch.int64, device=dev)
ctx = G.Context(0); stream = torch.cuda.Stream(device=dev); ctx.set_stream(stream.cuda_stream)
N = 60
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(N)]
torch.cuda.synchronize()
with torch.cuda.stream(stream):
    for k in range(N):
        ev[k][0].record(stream)
        ctx.encode_device(block, ts, val, out, nw, uniform_npoints=P, uniform_slot_words=slot)
        ev[k][1].record(stream)
        ctx.decode_device(out, ts2, val2, npts, S, uniform_slot_words=slot, in_nwords=nw, uniform_capacity=P)
        ev[k][2].record(stream)
ctx.sync(); torch.cuda.synchronize()
print("enc", " ".join("%.3f" % e[0].elapsed_time(e[1]) for e in ev))
print("dec", " ".join("%.3f" % e[1].elapsed_time(e[2]) for e in ev))
