#!/usr/bin/env python3
"""Does the decompressor run differently when it follows itself than when it follows the compressor (as in
bench.py's step)?  Times the decode kernel of the default workload with HIP events in both settings.
usage: python tools/decode_alone.py [series] [points]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gorilla-tsc_amd"))
sys.path.insert(0, ROOT)
import numpy as np, torch
import pygtsc as G
import bench
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
P = int(sys.argv[2]) if len(sys.argv) > 2 else 120
dev = torch.device("cuda")
block, ts, val = bench.make_workload("gauge", S, P, dev, 1)
slot = G.max_words(P)
out = torch.zeros(S * slot, dtype=torch.int64, device=dev)
nw = torch.zeros(S, dtype=torch.int64, device=dev)
ts2, val2 = torch.zeros_like(ts), torch.zeros_like(val)
npts = torch.zeros(S, dtype=torch.int64, device=dev)
ctx = G.Context(0)
stream = torch.cuda.Stream(device=dev)
ctx.set_stream(stream.cuda_stream)
enc = lambda: ctx.encode_device(block, ts, val, out, nw, uniform_npoints=P, uniform_slot_words=slot)
dec = lambda: ctx.decode_device(out, ts2, val2, npts, S, uniform_slot_words=slot, in_nwords=nw, uniform_capacity=P)
def timed(seq, reps=40):
    """seq: list of (fn, timed?) per step; returns mean ms of the timed ones"""
    with torch.cuda.stream(stream):
        for _ in range(10):
            for f, _t in seq: f()
    ctx.sync()
    ev = []
    with torch.cuda.stream(stream):
        for _ in range(reps):
            for f, t in seq:
                if t:
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record(stream); f(); b.record(stream); ev.append((a, b))
                else:
                    f()
    ctx.sync(); torch.cuda.synchronize()
    return float(np.mean([a.elapsed_time(b) for a, b in ev]))
torch.cuda.synchronize()
enc(); ctx.sync()
print("decode after encode : %.4f ms" % timed([(enc, False), (dec, True)]))
print("decode after decode : %.4f ms" % timed([(dec, True)]))
print("encode after decode : %.4f ms" % timed([(dec, False), (enc, True)]))
print("encode after encode : %.4f ms" % timed([(enc, True)]))
fill = torch.empty(1 << 28, dtype=torch.int64, device=dev)  # 2 GiB
def wipe():
    with torch.cuda.stream(stream):
        fill.fill_(1)
print("decode after a 2 GiB fill : %.4f ms" % timed([(wipe, False), (dec, True)], reps=20))
print("decode after encode, again: %.4f ms" % timed([(enc, False), (dec, True)]))
ctx.close()
