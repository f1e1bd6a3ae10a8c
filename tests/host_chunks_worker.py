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

mode = sys.argv[1] if len(sys.argv) > 1 else "ragged"

if mode == "fail":
    # GTSC_TEST_FAIL_CHUNK: the pipeline fails where it stages that chunk; nothing may be in flight afterwards
    b = synth.ragged_mix(seed=9, nseries=1200, pmax=400)
    ctx = G.Context(0)
    good_words, good_off = None, None
    for call in ("encode", "decode"):
        try:
            if call == "encode":
                ctx.encode_host(b.block_ts, b.ts, b.val, pt_off=b.pt_off)
            else:
                ref_out, ref_off, ref_nw, _ = O.encode_batch(b.block_ts, b.ts, b.val, b.pt_off, nthreads=4)
                words = np.concatenate([ref_out[int(ref_off[s]):int(ref_off[s]) + int(ref_nw[s])] for s in range(b.nseries)])
                off = np.zeros(b.nseries + 1, dtype=np.uint64)
                np.cumsum(ref_nw, out=off[1:])
                ctx.decode_host(words, off, pt_off=b.pt_off)
            raise SystemExit("the injected failure did not surface in " + call)
        except G.GtscError as e:
            assert e.code == G.ERR_HIP and "injected" in str(e), str(e)
        assert ctx.host_pipeline_idle(), call + ": copies still in flight after the error"
    ctx.close()
    print("failure ok")
    raise SystemExit(0)

if mode in ("skewed", "skewed_many"):
    # the skewed-length configuration: 10 to 100,000 points per series, constant / counter / random-walk series --
    # or ("skewed_many") 10 to 20,000 points, so that a chunk holds several hundred series: more than the device has
    # compute units, which is where the workgroup-per-block decompressor lists its blocks by size and runs on a
    # stream of its own beside the other two
    if mode == "skewed":
        b = synth.skewed(total_points=1 << 21, seed=0x5EED0005, pmin=10, pmax=100000)
    else:
        b = synth.skewed(total_points=3 << 20, seed=0x5EED0006, pmin=10, pmax=20000)
    ctx = G.Context(0)
    for gfmt, ofmt in ((G.FMT_V2, O.V2), (G.FMT_V1, O.V1)):
        words, off, nbits = ctx.encode_host(b.block_ts, b.ts, b.val, pt_off=b.pt_off, fmt=gfmt, want_nbits=True)
        ref_out, ref_off, ref_nw, _ = O.encode_batch(b.block_ts, b.ts, b.val, b.pt_off, fmt=ofmt, nthreads=8)
        expect_off = np.zeros(b.nseries + 1, dtype=np.uint64)
        np.cumsum(ref_nw, out=expect_off[1:])
        assert np.array_equal(off, expect_off), "image lengths differ"
        assert np.array_equal((nbits // np.uint64(64)) + np.uint64(2), ref_nw), "bit lengths differ"
        ref = np.concatenate([ref_out[int(ref_off[s]):int(ref_off[s]) + int(ref_nw[s])] for s in range(b.nseries)])
        assert np.array_equal(words, ref), "compressed words differ"
        ts, val, npts, st = ctx.decode_host(words, off, pt_off=b.pt_off, fmt=gfmt)
        rts, rval, rnp, rst = O.decode_batch(words, off, None, b.pt_off, fmt=ofmt)
        assert np.array_equal(st, rst) and np.array_equal(npts, rnp) and not st.any()
        assert np.array_equal(ts, rts) and np.array_equal(val, rval)
        assert np.array_equal(ts, b.ts) and np.array_equal(val, b.val)
    ctx.close()
    print("chunks ok")
    raise SystemExit(0)

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
