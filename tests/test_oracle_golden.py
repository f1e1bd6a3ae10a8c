"""CPU suite: the oracle against the golden fixtures and against the independent
bit-string restatement.  Compressed-byte parity is unpinned by a running
reference (no JVM); what these tests pin is (1) agreement of two differently
shaped restatements with each other and with hand-derived vectors and (2) the
reference's own assertion, decode(encode(x)) == x, on its own test inputs
(reference: src/test/java/.../EncodeGorillaTest.java, EncodeTest.java)."""
import hashlib
import struct

import numpy as np
import pytest

import pyoracle as O
import spec_codec as SP

FMT = {"v2": O.V2, "v1": O.V1}


def d2u(x):
    return struct.unpack("<Q", struct.pack("<d", float(x)))[0]


def test_known_answers(golden):
    for name, c in golden["answers"].items():
        e = O.Encoder(c["block_ts"], FMT[c["format"]])
        for t, v in zip(c["ts"], c["val_bits"]):
            e.add(t, v)
        if c["closed"]:
            e.close()
        w = e.words()
        assert ["%016x" % int(x) for x in w] == c["words_hex"], name
        assert e.nbits == c["nbits"], name
        assert e.bytes().hex() == c["bytes_hex"], name
        be = b"".join(struct.pack(">Q", int(x)) for x in w)
        assert hashlib.sha256(be).hexdigest() == c["sha256_be"], name
        # image length rule: floor(N/64) + 2 when closed, + 1 for a snapshot
        assert len(w) == c["nbits"] // 64 + (2 if c["closed"] else 1), name
        assert len(c["bytes_hex"]) // 2 == c["nbits"] // 8 + (1 if c["closed"] else 0), name


def test_reference_inputs_round_trip(golden):
    """What the reference's tests assert, on their own data, for both formats."""
    inp = golden["inputs"]
    sets = [(inp["willItBlend"]["block_ts"], inp["willItBlend"]["ts"], inp["willItBlend"]["val_bits"])]
    for k in ("simple", "similarFloats"):
        sets.append((inp[k]["block_ts"], [inp[k]["block_ts"] + o for o in inp[k]["offsets"]],
                     [d2u(x) for x in inp[k]["doubles"]]))
    for block, ts, vals in sets:
        for fmt in (O.V2, O.V1):
            w = O.encode_block(block, ts, vals, fmt=fmt)
            t, v, rc = O.decode_block(w, fmt=fmt)
            assert rc == O.END
            assert t.tolist() == list(ts) and v.tolist() == list(vals)


def test_empty_block_and_snapshot():
    # testEmptyBlock: close() without points -> first readPair() is null
    w = O.encode_block(1500400800000, [], [])
    t, v, rc = O.decode_block(w)
    assert rc == O.END and len(t) == 0
    # testCopyFlush: an unclosed snapshot decodes its points (the reference reads exactly 2 pairs)
    B = 1500400800000
    w = O.encode_block(B, [B + 1, B + 2], [d2u(1.0), d2u(1.0)], close=False)
    d = O.lib().gor_dec_new(O.V2, w.ctypes.data, len(w), 0)
    import ctypes as C
    t, v = C.c_int64(), C.c_uint64()
    for k in (1, 2):
        assert O.lib().gor_dec_next(d, C.byref(t), C.byref(v)) == O.OK
        assert t.value == B + k and v.value == d2u(1.0)
    O.lib().gor_dec_free(d)


@pytest.mark.parametrize("fmt", ["v2", "v1"])
def test_oracle_matches_bitstring_restatement_fuzz(fmt):
    rng = np.random.default_rng(1234)
    for it in range(120):
        n = int(rng.integers(0, 90))
        block = int(1500400800000 + rng.integers(0, 100) * 3600000)
        kind = it % 4
        if kind == 0:
            deltas = rng.integers(1, 5000, size=n)
        elif kind == 1:
            deltas = np.full(n, 60000)
        elif kind == 2:
            deltas = rng.choice([1, 2, 63, 64, 65, 66, 255, 256, 257, 2047, 2048, 2049, 2050, 100000, 999], size=n)
        else:
            deltas = rng.integers(1, 2 ** 26, size=n)
        ts = (block + np.cumsum(deltas)).tolist() if n else []
        if it % 3 == 0:
            vals = rng.integers(0, 2 ** 64, size=n, dtype=np.uint64).tolist()
        elif it % 3 == 1:
            vals = (100.0 + np.cumsum(rng.standard_normal(n))).view(np.uint64).tolist() if n else []
        else:
            vals = rng.integers(0, 2 ** 20, size=n).tolist()
        w = O.encode_block(block, ts, vals, fmt=FMT[fmt])
        bits = SP.encode_bits(block, list(zip(ts, vals)), fmt)
        assert [int(x) for x in w] == SP.words_image(bits)
        t, v, rc = O.decode_block(w, fmt=FMT[fmt])
        assert rc == O.END and t.tolist() == ts and v.tolist() == vals
        assert SP.decode_words([int(x) for x in w], fmt) == list(zip(ts, vals))


def test_first_timestamp_zero_quirk():
    """A first timestamp of exactly 0 leaves storedTimestamp == 0, so the next point
    is written as a 'first' point again (reference: GorillaCompressor.java:50,65).
    The decoder has the same test (GorillaDecompressor.java:54) and undoes it."""
    pts = [(0, 5), (1000, 6), (2000, 7)]
    for fmt in ("v2", "v1"):
        w = O.encode_block(0, [p[0] for p in pts], [p[1] for p in pts], fmt=FMT[fmt])
        bits = SP.encode_bits(0, pts, fmt)
        assert [int(x) for x in w] == SP.words_image(bits)
        assert bits[64:64 + 27] == "0" * 27 and len(bits) >= 64 + 2 * 91
        t, v, rc = O.decode_block(w, fmt=FMT[fmt])
        assert rc == O.END and list(zip(t.tolist(), v.tolist())) == pts


def test_int_wrap_of_v2_deltas():
    """v2 keeps deltas as Java ints; the stream is still defined when they wrap
    (reference: GorillaCompressor.java:74,102-103)."""
    B = 1500400800000
    pts = [(B + 5, 1), (B + 5 + 2 ** 31 + 7, 2), (B + 5 + 2 ** 33, 3)]
    w = O.encode_block(B, [p[0] for p in pts], [p[1] for p in pts])
    assert [int(x) for x in w] == SP.words_image(SP.encode_bits(B, pts, "v2"))


def test_truncated_and_malformed_streams():
    B = 1500400800000
    ts = [B + 1000 * (i + 1) for i in range(50)]
    vals = (100.0 + np.arange(50) * 0.37).view(np.uint64).tolist()
    w = O.encode_block(B, ts, vals)
    # dropping the trailing zero word: the reader's eager load runs off the array
    t, v, rc = O.decode_block(w[:-1])
    assert rc in (O.END, O.ERR_TRUNCATED)
    t, v, rc = O.decode_block(w[: len(w) // 2])
    assert rc == O.ERR_TRUNCATED and 0 < len(t) < 50 and t.tolist() == ts[: len(t)]
    t, v, rc = O.decode_block(w[:0])
    assert rc == O.ERR_TRUNCATED
    # '10' (reuse window) before any window was defined
    bits = format(B, "064b") + format(1000, "027b") + format(vals[0], "064b") + "0" + "10" + "1" * 40
    t, v, rc = O.decode_block(np.array(SP.words_image(bits), dtype=np.uint64))
    assert rc == O.ERR_MALFORMED and len(t) == 1


def test_dfcm_predictor_round_trip(golden):
    """DifferentialFCM on both sides (reference: EncodeGorillaTest.java:550-589)."""
    wb = golden["inputs"]["willItBlend"]
    w = O.encode_block(wb["block_ts"], wb["ts"], wb["val_bits"], dfcm_size=1024)
    be = b"".join(struct.pack(">Q", int(x)) for x in w)
    assert hashlib.sha256(be).hexdigest() == "71fe5ddae135f49adc99b9e7a50f34df97a86a137bf7efbb10d727baf0b14221"
    t, v, rc = O.decode_block(w, dfcm_size=1024)
    assert rc == O.END and v.tolist() == wb["val_bits"] and t.tolist() == wb["ts"]


def test_large_blocks_round_trip():
    """testEncodeLargeAmountOfData / testLongEncoding shapes: 100k doubles i*rand, 10k random longs."""
    rng = np.random.default_rng(99)
    B = 1500400800000
    n = 100000
    ts = B + 60 * np.arange(1, n + 1, dtype=np.int64)
    vals = (np.arange(n) * rng.random(n)).view(np.uint64)
    for fmt in (O.V2, O.V1):
        out, off, nw, st = O.encode_batch(np.array([B]), ts, vals, np.array([0, n], dtype=np.uint64), fmt=fmt)
        assert st[0] == 0
        t, v, npts, st2 = O.decode_batch(out, off, nw, np.array([0, n], dtype=np.uint64), fmt=fmt)
        assert npts[0] == n and st2[0] == 0 and np.array_equal(t, ts) and np.array_equal(v, vals)
    lv = rng.integers(0, 2 ** 31 - 1, size=10000).astype(np.uint64)
    lt = B + 60 * np.arange(1, 10001, dtype=np.int64)
    w = O.encode_block(B, lt.tolist(), lv.tolist())
    t, v, rc = O.decode_block(w)
    assert rc == O.END and np.array_equal(v, lv) and np.array_equal(t, lt)


def test_batch_matches_streaming_and_threads():
    from pygtsc import synth
    b = synth.ragged_mix(seed=3, nseries=64, pmax=200)
    o1 = O.encode_batch(b.block_ts, b.ts, b.val, b.pt_off, nthreads=1)
    o4 = O.encode_batch(b.block_ts, b.ts, b.val, b.pt_off, nthreads=4)
    for x, y in zip(o1, o4):
        assert np.array_equal(x, y)
    out, off, nw, st = o1
    assert not st.any()
    for s in range(b.nseries):
        t, v = b.series(s)
        w = O.encode_block(int(b.block_ts[s]), t.tolist(), v.tolist())
        assert np.array_equal(out[int(off[s]): int(off[s]) + int(nw[s])], w)
        assert int(nw[s]) <= O.max_words(len(t))
    t, v, npts, st = O.decode_batch(out, off, nw, b.pt_off, nthreads=3)
    assert not st.any() and np.array_equal(t, b.ts) and np.array_equal(v, b.val)
    assert np.array_equal(npts, np.diff(b.pt_off.astype(np.int64)).astype(np.uint64))


def test_worst_case_bound_holds():
    """114 bits per point (36-bit timestamp code + 78-bit new-window value code) bounds
    every block; the densest steady state is 36 + 66 (window (0,0) is reused forever)."""
    B = 1500400800000
    n = 200
    ts, t = [], B + 1
    for i in range(n):
        ts.append(t)
        t += 5000 if i % 2 else 50000
    vals, v = [], 0x8000000000000001
    for i in range(n):
        vals.append(v)
        v ^= 0x8000000000000001 | (i << 1)
    w = O.encode_block(B, ts, vals)
    bits = SP.encode_bits(B, list(zip(ts, vals)), "v2")
    assert len(bits) == 64 + 91 + (36 + 78) + (36 + 66) * (n - 2) + 37
    assert len(w) == len(bits) // 64 + 2 <= O.max_words(n)
    assert O.max_words(0) == 3 and O.max_words(1) == (64 + 91 + 37) // 64 + 2
