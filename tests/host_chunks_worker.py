"""Run with GTSC_CHUNK_POINTS set small: the host entry points then cut a small ragged batch into
many chunks (alternating between the two staging sets), and the result must not change."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "gorilla-tsc_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402

import pygtsc as G  # noqa: E402
import pyoracle as O  # noqa: E402
from pygtsc import synth  # noqa: E402

assert int(os.environ["GTSC_CHUNK_POINTS"]) < 10000
b = synth.ragged_mix(seed=9, nseries=1200, pmax=400)  # includes empty series and series longer than a chunk's average
ctx = G.Context(0)
for gfmt, ofmt in ((G.FMT_V2, O.V2), (G.FMT_V1, O.V1)):
    words, off = ctx.encode_host(b.block_ts, b.ts, b.val, pt_off=b.pt_off, fmt=gfmt)
    ref_out, ref_off, ref_nw, _ = O.encode_batch(b.block_ts, b.ts, b.val, b.pt_off, fmt=ofmt, nthreads=4)
    ref = np.concatenate([ref_out[int(ref_off[s]):int(ref_off[s]) + int(ref_nw[s])] for s in range(b.nseries)])
    assert len(words) == len(ref) and np.array_equal(words, ref), "compressed words differ"
    ts, val, npts, st = ctx.decode_host(words, off, pt_off=b.pt_off, fmt=gfmt)
    assert not st.any() and np.array_equal(ts, b.ts) and np.array_equal(val, b.val)
    assert np.array_equal(npts.astype(np.int64), np.diff(b.pt_off.astype(np.int64)))
u = synth.gauge_integer(S=700, P=120)
words, off = ctx.encode_host(u.block_ts, u.ts, u.val, uniform_npoints=120)
ts, val, npts, st = ctx.decode_host(words, off, uniform_capacity=120)
assert np.array_equal(ts, u.ts) and np.array_equal(val, u.val) and np.all(npts == 120)
ctx.close()
print("chunks ok")
