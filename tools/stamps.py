#!/usr/bin/env python3
"""Reads the per-wave cycle stamps a -DGTSC_STAMP build of libgtsc leaves in the slot tails
(diagnostic only).  usage: GTSC_LIB=.../libgtsc_stamp.so python tools/stamps.py [P]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gorilla-tsc_amd"))
sys.path.insert(0, ROOT)
import numpy as np, torch
import pygtsc as G
import bench
P = int(sys.argv[1]) if len(sys.argv) > 1 else 120
WORKLOAD = sys.argv[2] if len(sys.argv) > 2 else "gauge"
S = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
block, ts, val = bench.make_workload(WORKLOAD, S, P, torch.device("cuda"), 1)
slot = G.max_words(P)
out = torch.zeros(S * slot, dtype=torch.int64, device="cuda")
nw = torch.zeros(S, dtype=torch.int64, device="cuda")
ctx = G.Context(0)
torch.cuda.synchronize()
for _ in range(2):
    ctx.encode_device(block, ts, val, out, nw, uniform_npoints=P, uniform_slot_words=slot)
ctx.sync()
d = out.view(S, slot)[::64, -4:].cpu().numpy().astype(np.float64)
names = ["stage(wait loads+ds_write)", "fetch(issue)", "compute", "final flush"]
tot = d.sum(1).mean()
for i, n in enumerate(names):
    print("%-28s mean %9.0f cycles/wave  (%.1f%%)" % (n, d[:, i].mean(), 100 * d[:, i].mean() / tot))
print("total %.0f cycles/wave = %.1f us at 100 MHz ticks?" % (tot, tot / 100))
# ---- decode stamps: lane 0 of each wave leaves {compute, flush} cycles in out_npoints[s+1], [s+2]
ts2 = torch.zeros_like(ts); val2 = torch.zeros_like(val)
npts = torch.zeros(S, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
for _ in range(2):
    ctx.decode_device(out, ts2, val2, npts, S, uniform_slot_words=slot, in_nwords=nw, uniform_capacity=P)
ctx.sync()
d = npts.view(-1, 64)[:, 1:8].cpu().numpy().astype(np.float64)
tot = d[:, :3].sum(1).mean()
for i, n in enumerate(["decode compute", "decode flush", "decode prologue", "  entry->fill loads issued", "  issued->ring written", "  entry->kernel arguments there", "  of compute: waiting at landings (16-word ring)"]):
    print("%-28s mean %9.0f cycles/wave  (%.1f%%)" % (n, d[:, i].mean(), 100 * d[:, i].mean() / tot))
