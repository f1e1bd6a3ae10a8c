#!/usr/bin/env python3
"""One 100,000-point block of a value class of tools/block_class_time.py through the library named by GTSC_LIB
(the -DGTSC_WG_STAMP build prints the cycle stamps of the workgroup-per-block decompressor).  usage: wg_stamp_one.py class"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gorilla-tsc_amd")]
import numpy as np, torch
import pygtsc as G
from pygtsc import synth

N = 100000
rng = np.random.default_rng(3)
classes = {
    "constant": np.full(N, 123.456),
    "counter": (rng.integers(0, 1000) + np.cumsum(rng.integers(0, 5, size=N))).astype(np.float64),
}
name = sys.argv[1]
v = classes[name.replace("jitter_", "")]
ts = synth.BLOCK_TS + 1000 * np.arange(1, N + 1, dtype=np.int64)
if name.startswith("jitter_"):
    ts = ts + rng.integers(-40, 41, N)
d = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()
ctx = G.Context(0)
d_block, d_ts, d_val = d(np.array([synth.BLOCK_TS], dtype=np.int64)), d(ts), d(np.ascontiguousarray(v).view(np.uint64))
slot = G.max_words(N)
d_out = torch.zeros(slot, dtype=torch.int64, device="cuda"); d_nw = torch.zeros(1, dtype=torch.int64, device="cuda")
d_ts2 = torch.zeros_like(d_ts); d_val2 = torch.zeros_like(d_val); d_np = torch.zeros(1, dtype=torch.int64, device="cuda")
po = d(np.array([0, N], dtype=np.uint64)); so = d(np.array([0, slot], dtype=np.uint64))
ctx.encode_device(d_block, d_ts, d_val, d_out, d_nw, pt_off=po, out_off=so); ctx.sync()
ctx.decode_device(d_out, d_ts2, d_val2, d_np, 1, in_off=so, in_nwords=d_nw, pt_off=po); ctx.sync()
assert torch.equal(d_ts2, d_ts) and torch.equal(d_val2, d_val)
ctx.close()
