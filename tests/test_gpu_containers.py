"""Snapshots of open blocks and ByteBuffer byte images on the GPU, against the oracle
(T/EncodeGorillaTest.java:483-500 testCopyFlush; G/ByteBufferBitOutput.java:42-103)."""
import collections

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import pyoracle as O  # noqa: E402
import pygtsc as G  # noqa: E402
from pygtsc import synth  # noqa: E402

FMTS = [(G.FMT_V2, O.V2), (G.FMT_V1, O.V1)]


@pytest.fixture(scope="module")
def ctx():
    c = G.Context(0)
    yield c
    c.close()


# The library's calls are asynchronous on its own stream.  A tensor made in an argument list dies
# when the call returns, and torch's caching allocator hands its memory to the next allocation
# while the kernel may still be reading it: the helper keeps the most recent uploads alive.
_KEEP = collections.deque(maxlen=64)


def dev(a):
    a = np.ascontiguousarray(a)
    t = torch.from_numpy(a.view(np.int64) if a.dtype == np.uint64 else a).cuda()
    _KEEP.append(t)
    return t


def batch():
    b = synth.ragged_mix(seed=21, nseries=150, pmax=180)
    # word-boundary cases: a few constant series of growing length walk the stream end across a word edge
    return b


def encode(ctx, b, gfmt, options):
    S = b.nseries
    lens = np.diff(b.pt_off.astype(np.int64))
    slot = np.zeros(S + 1, dtype=np.uint64)
    np.cumsum([G.max_words(int(p)) for p in lens], out=slot[1:])
    d_out = torch.full((int(slot[-1]),), -1, dtype=torch.int64, device="cuda")
    d_nw = torch.zeros(S, dtype=torch.int64, device="cuda")
    d_nb = torch.zeros(S, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    ctx.encode_device(dev(b.block_ts), dev(b.ts), dev(b.val), d_out, d_nw, pt_off=dev(b.pt_off), out_off=dev(slot),
                      fmt=gfmt, options=options, out_nbits=d_nb)
    ctx.sync()
    return d_out, d_nw, d_nb, slot


@pytest.mark.parametrize("gfmt,ofmt", FMTS)
def test_snapshot_of_open_blocks(ctx, gfmt, ofmt):
    b = batch()
    d_out, d_nw, d_nb, slot = encode(ctx, b, gfmt, G.ENC_NO_CLOSE)
    out, nw, nb = d_out.cpu().numpy().view(np.uint64), d_nw.cpu().numpy(), d_nb.cpu().numpy()
    for s in range(b.nseries):
        t, v = b.series(s)
        e = O.Encoder(int(b.block_ts[s]), ofmt)
        for ti, vi in zip(t.tolist(), v.tolist()):
            e.add(ti, vi)
        ref = e.words()  # not closed
        assert int(nb[s]) == e.nbits
        assert int(nw[s]) == len(ref)
        assert np.array_equal(out[int(slot[s]):int(slot[s]) + len(ref)], ref), s


@pytest.mark.parametrize("gfmt,ofmt", FMTS)
def test_byte_images_and_back(ctx, gfmt, ofmt):
    b = batch()
    S = b.nseries
    d_out, d_nw, d_nb, slot = encode(ctx, b, gfmt, 0)
    nb = d_nb.cpu().numpy()
    boff = np.zeros(S + 1, dtype=np.uint64)
    np.cumsum(nb // 8 + 1, out=boff[1:])  # back to back, at odd byte offsets
    d_bytes = torch.full((int(boff[-1]) + 8,), 0x5A, dtype=torch.uint8, device="cuda")
    d_nby = torch.zeros(S, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    ctx.words_to_bytes_device(d_out, d_nb, d_bytes, d_nby, word_off=dev(slot), byte_off=dev(boff))
    ctx.sync()
    got = d_bytes.cpu().numpy()
    nby = d_nby.cpu().numpy()
    for s in range(S):
        t, v = b.series(s)
        e = O.Encoder(int(b.block_ts[s]), ofmt)
        for ti, vi in zip(t.tolist(), v.tolist()):
            e.add(ti, vi)
        e.close()
        ref = np.frombuffer(e.bytes(), dtype=np.uint8)
        assert int(nb[s]) == e.nbits and int(nby[s]) == len(ref)
        assert np.array_equal(got[int(boff[s]):int(boff[s + 1])], ref), s
    assert np.all(got[int(boff[-1]):] == 0x5A)  # nothing written past the last image
    # bytes -> word images -> decode gives the points back
    woff = np.zeros(S + 1, dtype=np.uint64)
    np.cumsum((nby + 7) // 8 + 1, out=woff[1:])
    d_w = torch.full((int(woff[-1]),), -1, dtype=torch.int64, device="cuda")
    d_nw2 = torch.zeros(S, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    ctx.bytes_to_words_device(d_bytes, d_w, d_nw2, nbytes=d_nby, byte_off=dev(boff), word_off=dev(woff))
    d_ts = torch.zeros(b.npoints, dtype=torch.int64, device="cuda")
    d_val = torch.zeros(b.npoints, dtype=torch.int64, device="cuda")
    d_np = torch.zeros(S, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    ctx.decode_device(d_w, d_ts, d_val, d_np, S, in_off=dev(woff), in_nwords=d_nw2, pt_off=dev(b.pt_off), fmt=gfmt)
    ctx.sync()
    assert np.array_equal(d_np.cpu().numpy(), np.diff(b.pt_off.astype(np.int64)))
    rts, rval, rnp, rst = O.decode_batch(d_w.cpu().numpy().view(np.uint64), woff, d_nw2.cpu().numpy().view(np.uint64),
                                         b.pt_off, fmt=ofmt)
    assert np.array_equal(d_ts.cpu().numpy(), rts) and np.array_equal(d_val.cpu().numpy().view(np.uint64), rval)
    if gfmt == G.FMT_V2:
        assert np.array_equal(d_ts.cpu().numpy(), b.ts) and np.array_equal(d_val.cpu().numpy().view(np.uint64), b.val)


def dfcm_decode_against_oracle(ctx, words, off, nw, pt_off, size):
    """GPU decode with DifferentialFCM(size) against the oracle's decoder on the same images."""
    S = len(off) - 1
    total = int(pt_off[-1])
    d_ts = torch.full((max(total, 1),), -7, dtype=torch.int64, device="cuda")
    d_val = torch.full((max(total, 1),), -7, dtype=torch.int64, device="cuda")
    d_np = torch.zeros(S, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    ctx.decode_device(dev(words), d_ts, d_val, d_np, S, in_off=dev(off), in_nwords=dev(nw), pt_off=dev(pt_off))
    ctx.sync()
    gts, gval, gnp = d_ts.cpu().numpy(), d_val.cpu().numpy().view(np.uint64), d_np.cpu().numpy()
    for s in range(S):
        a, n = int(off[s]), int(nw[s])
        rts, rval, rc = O.decode_block(words[a:a + n], O.V2, dfcm_size=size)
        assert rc == O.END and int(gnp[s]) == len(rts), s
        p0 = int(pt_off[s])
        assert np.array_equal(gts[p0:p0 + len(rts)], rts), s
        assert np.array_equal(gval[p0:p0 + len(rts)], rval), s
    return gts, gval


@pytest.mark.parametrize("size", [1, 5, 128, 1024])
def test_dfcm_predictor(ctx, size):
    """GorillaCompressor(ts, out, new DifferentialFCM(size)) per block (G/GorillaCompressor.java:34-37),
    checked word for word against the oracle's DFCM.  Decoding is checked against the oracle's
    decoder, which follows the reference in NOT updating the predictor when a value code is the
    single bit '0' (G/ValueDecompressor.java:53): after such a point the reference's decoder and
    compressor disagree about the predictor's state, so DFCM blocks with a correctly predicted
    value do not round-trip in the reference either.  The GPU reproduces that, bit for bit."""
    b = synth.ragged_mix(seed=31, nseries=130, pmax=260)
    S = b.nseries
    lens = np.diff(b.pt_off.astype(np.int64))
    slot = np.zeros(S + 1, dtype=np.uint64)
    np.cumsum([G.max_words(int(p)) for p in lens], out=slot[1:])
    d_out = torch.full((int(slot[-1]),), -1, dtype=torch.int64, device="cuda")
    d_nw = torch.zeros(S, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    ctx.set_predictor(G.PRED_DFCM, size)
    try:
        ctx.encode_device(dev(b.block_ts), dev(b.ts), dev(b.val), d_out, d_nw, pt_off=dev(b.pt_off), out_off=dev(slot))
        ctx.sync()
        out, nw = d_out.cpu().numpy().view(np.uint64), d_nw.cpu().numpy().view(np.uint64)
        differs_from_last_value = False
        for s in range(S):
            t, v = b.series(s)
            ref = O.encode_block(int(b.block_ts[s]), t, v, O.V2, dfcm_size=size)
            assert int(nw[s]) == len(ref) and np.array_equal(out[int(slot[s]):int(slot[s]) + len(ref)], ref), s
            plain = O.encode_block(int(b.block_ts[s]), t, v, O.V2)
            differs_from_last_value |= len(plain) != len(ref) or not np.array_equal(plain, ref)
        assert differs_from_last_value  # the predictor did take part
        gts, gval = dfcm_decode_against_oracle(ctx, out, slot, nw, b.pt_off, size)
        assert np.array_equal(gts, b.ts)  # timestamps do not depend on the predictor
        # host entry points use the same predictor
        words, off = ctx.encode_host(b.block_ts, b.ts, b.val, pt_off=b.pt_off)
        ts, val, npts, st = ctx.decode_host(words, off, pt_off=b.pt_off)
        assert np.array_equal(ts, gts) and np.array_equal(val, gval)
        s0 = 7
        ref = O.encode_block(int(b.block_ts[s0]), *b.series(s0), O.V2, dfcm_size=size)
        assert np.array_equal(words[int(off[s0]):int(off[s0 + 1])], ref)
        with pytest.raises(G.GtscError):  # the 1.x classes take no predictor
            ctx.encode_device(dev(b.block_ts), dev(b.ts), dev(b.val), d_out, d_nw, pt_off=dev(b.pt_off), out_off=dev(slot),
                              fmt=G.FMT_V1)
    finally:
        ctx.set_predictor(G.PRED_LAST)


def test_dfcm_decode_follows_the_reference_decoder(ctx):
    """Two shapes that tell a decoder that updates its predictor at every point from the
    reference's, which does not on a '0' value code: (a) linear long series (every difference
    repeats, so DFCM predicts exactly from the third point on), (b) series that repeat a
    predicted value after the hash has moved.  The expected output is the oracle decoder's,
    which is not the input: [0,10,20,30,40,50] with DifferentialFCM(8) decodes to
    [0,10,20,20,20,20]."""
    B = synth.BLOCK_TS
    rng = np.random.default_rng(4)
    series = [np.array([0, 10, 20, 30, 40, 50], dtype=np.uint64)]
    for k in range(40):  # (a) linear, various steps and lengths
        n = 3 + 5 * k
        series.append((np.uint64(1000 * k) + np.uint64(7 + k) * np.arange(n, dtype=np.uint64)))
    for k in range(40):  # (b) irregular stretches between linear ones: the hash moves, then predictions hit again
        parts = []
        v = int(rng.integers(0, 1 << 40))
        for _ in range(6):
            step = int(rng.integers(1, 1 << 20))
            n = int(rng.integers(3, 30))
            parts.append(v + step * np.arange(n, dtype=np.int64))
            v = int(parts[-1][-1]) + int(rng.integers(1, 1 << 30))
            parts.append(rng.integers(0, 1 << 52, int(rng.integers(1, 6))))
        series.append(np.concatenate(parts).astype(np.uint64))
    lens = [len(v) for v in series]
    pt_off = np.zeros(len(series) + 1, dtype=np.uint64)
    np.cumsum(lens, out=pt_off[1:])
    ts = np.concatenate([B + 1000 * np.arange(1, n + 1, dtype=np.int64) for n in lens])
    val = np.concatenate(series)
    S = len(series)
    for size in (8, 1024):
        ctx.set_predictor(G.PRED_DFCM, size)
        try:
            words, off = ctx.encode_host(np.full(S, B, dtype=np.int64), ts, val, pt_off=pt_off)
            nw = np.diff(off.astype(np.int64)).astype(np.uint64)
            for s in range(S):
                ref = O.encode_block(B, ts[int(pt_off[s]):int(pt_off[s + 1])], series[s], O.V2, dfcm_size=size)
                assert np.array_equal(words[int(off[s]):int(off[s + 1])], ref), s
            gts, gval = dfcm_decode_against_oracle(ctx, words, off, nw, pt_off, size)
            if size == 8:
                assert gval[:6].tolist() == [0, 10, 20, 20, 20, 20]
            assert not np.array_equal(gval, val)  # the reference's decoder does not give the input back here
        finally:
            ctx.set_predictor(G.PRED_LAST)


def test_host_snapshots_and_bit_lengths(ctx):
    """gtsc_encode_host_ex: open-block snapshots (testCopyFlush, T/EncodeGorillaTest.java:483-500)
    and stream bit lengths from host memory, against the oracle; the ByteBuffer image is the first
    floor(N/8)+1 bytes of the big-endian words."""
    b = batch()
    for gfmt, ofmt in FMTS:
        for options in (0, G.ENC_NO_CLOSE):
            words, off, nbits = ctx.encode_host(b.block_ts, b.ts, b.val, pt_off=b.pt_off, fmt=gfmt, options=options,
                                                want_nbits=True)
            for s in range(b.nseries):
                t, v = b.series(s)
                e = O.Encoder(int(b.block_ts[s]), ofmt)
                for ti, vi in zip(t.tolist(), v.tolist()):
                    e.add(ti, vi)
                if not options:
                    e.close()
                img = words[int(off[s]):int(off[s + 1])]
                assert int(nbits[s]) == e.nbits and np.array_equal(img, e.words()), (s, options)
                if not options:
                    as_bytes = img.astype(">u8").tobytes()[: int(nbits[s]) // 8 + 1]
                    assert as_bytes == e.bytes(), s
