"""GPU parity suite: the HIP path, called through the C ABI (ctypes), against the
CPU oracle on the same inputs -- bit-exact on compressed words, decoded
timestamps, decoded value bits, counts and status words.  torch is used only
to own device memory."""
import collections
import struct

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
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    c = G.Context(0)
    yield c
    c.close()


# The library's calls are asynchronous on its own stream.  A tensor made in an argument list dies
# when the call returns, and torch's caching allocator hands its memory to the next allocation
# while the kernel may still be reading it: the helper keeps the most recent uploads alive.
_KEEP = collections.deque(maxlen=64)


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a).view(np.int64) if a.dtype == np.uint64 else np.ascontiguousarray(a)).cuda()
    _KEEP.append(t)
    return t


def host_u64(t):
    return t.cpu().numpy().view(np.uint64)


def slot_offsets(b, tight=None):
    lens = np.diff(b.pt_off.astype(np.int64))
    bits = 64 + np.where(lens > 0, 91 + 114 * (lens - 1), 0) + 37
    words = bits // 64 + 2 if tight is None else np.asarray(tight, dtype=np.int64)
    off = np.zeros(len(lens) + 1, dtype=np.uint64)
    np.cumsum(words, out=off[1:])
    return off


def gpu_encode(ctx, b, gfmt, uniform=False, out_off=None, want_status=True):
    S = b.nseries
    d_block, d_ts, d_val = dev(b.block_ts), dev(b.ts), dev(b.val)
    d_nw = torch.zeros(S, dtype=torch.int64, device="cuda")
    d_st = torch.full((S,), 77, dtype=torch.int32, device="cuda") if want_status else None
    if uniform:
        P = b.uniform_npoints()
        assert P is not None
        slot = G.max_words(P)
        out_off = np.arange(S + 1, dtype=np.uint64) * np.uint64(slot)
        d_out = torch.full((S * slot,), -1, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        ctx.encode_device(d_block, d_ts, d_val, d_out, d_nw, uniform_npoints=P, uniform_slot_words=slot, status=d_st,
                          fmt=gfmt)
    else:
        if out_off is None:
            out_off = slot_offsets(b)
        d_out = torch.full((int(out_off[-1]),), -1, dtype=torch.int64, device="cuda")
        d_po, d_oo = dev(b.pt_off), dev(out_off)
        torch.cuda.synchronize()
        ctx.encode_device(d_block, d_ts, d_val, d_out, d_nw, pt_off=d_po, out_off=d_oo, status=d_st, fmt=gfmt)
    try:
        ctx.sync()
        rc = 0
    except G.GtscError as e:
        rc = e.code
    st = d_st.cpu().numpy().view(np.uint32) if want_status else None
    return host_u64(d_out), out_off, host_u64(d_nw), st, rc, d_out, d_nw


def check_encode_against_oracle(ctx, b, gfmt, ofmt, uniform=False):
    out, off, nw, st, rc, d_out, d_nw = gpu_encode(ctx, b, gfmt, uniform=uniform)
    assert rc == 0 and not st.any()
    ref_out, ref_off, ref_nw, ref_st = O.encode_batch(b.block_ts, b.ts, b.val, b.pt_off, out_off=off, fmt=ofmt,
                                                      nthreads=8)
    assert np.array_equal(nw, ref_nw)
    # compare exactly the image words of every series
    idx = np.repeat(off[:-1].astype(np.int64), nw.astype(np.int64)) + (
        np.arange(int(nw.sum())) - np.repeat(np.cumsum(nw.astype(np.int64)) - nw.astype(np.int64), nw.astype(np.int64)))
    assert np.array_equal(out[idx], ref_out[idx])
    return out, off, nw, d_out, d_nw


def gpu_decode(ctx, words, in_off, in_nw, pt_off, gfmt, uniform_cap=None, d_words=None):
    S = len(in_off) - 1
    d_w = dev(words) if d_words is None else d_words
    d_io = dev(in_off)
    d_inw = dev(in_nw) if in_nw is not None else None
    total = int(pt_off[-1]) if uniform_cap is None else S * uniform_cap
    d_ts = torch.full((max(total, 1),), -7, dtype=torch.int64, device="cuda")
    d_val = torch.full((max(total, 1),), -7, dtype=torch.int64, device="cuda")
    d_np = torch.zeros(S, dtype=torch.int64, device="cuda")
    d_st = torch.full((S,), 77, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    if uniform_cap is None:
        ctx.decode_device(d_w, d_ts, d_val, d_np, S, in_off=d_io, in_nwords=d_inw, pt_off=dev(pt_off), status=d_st, fmt=gfmt)
    else:
        ctx.decode_device(d_w, d_ts, d_val, d_np, S, in_off=d_io, in_nwords=d_inw, uniform_capacity=uniform_cap,
                          status=d_st, fmt=gfmt)
    try:
        ctx.sync()
        rc = 0
    except G.GtscError as e:
        rc = e.code
    return d_ts.cpu().numpy()[:total], host_u64(d_val)[:total], host_u64(d_np), d_st.cpu().numpy().view(np.uint32), rc


# --------------------------------------------------------------------------
def test_known_answer_vectors(ctx, golden):
    for gfmt, name in ((G.FMT_V2, "v2"), (G.FMT_V1, "v1")):
        cases = [(k, c) for k, c in golden["answers"].items() if c["format"] == name and c["closed"]]
        lens = [len(c["ts"]) for _, c in cases]
        pt_off = np.zeros(len(cases) + 1, dtype=np.uint64)
        np.cumsum(lens, out=pt_off[1:])
        b = synth.Batch(np.array([c["block_ts"] for _, c in cases], dtype=np.int64),
                        np.array([t for _, c in cases for t in c["ts"]], dtype=np.int64),
                        np.array([v for _, c in cases for v in c["val_bits"]], dtype=np.uint64), pt_off)
        out, off, nw, st, rc, _, _ = gpu_encode(ctx, b, gfmt)
        assert rc == 0 and not st.any()
        for i, (k, c) in enumerate(cases):
            img = out[int(off[i]): int(off[i]) + int(nw[i])]
            assert ["%016x" % int(x) for x in img] == c["words_hex"], k


@pytest.mark.parametrize("gfmt,ofmt", FMTS)
def test_reference_test_inputs_round_trip(ctx, golden, gfmt, ofmt):
    inp = golden["inputs"]
    d2u = lambda x: struct.unpack("<Q", struct.pack("<d", float(x)))[0]
    sets = [(inp["willItBlend"]["block_ts"], inp["willItBlend"]["ts"], inp["willItBlend"]["val_bits"])]
    for k in ("simple", "similarFloats"):
        sets.append((inp[k]["block_ts"], [inp[k]["block_ts"] + o for o in inp[k]["offsets"]],
                     [d2u(x) for x in inp[k]["doubles"]]))
    sets.append((1500400800000, [], []))  # testEmptyBlock
    pt_off = np.zeros(len(sets) + 1, dtype=np.uint64)
    np.cumsum([len(s[1]) for s in sets], out=pt_off[1:])
    b = synth.Batch(np.array([s[0] for s in sets], dtype=np.int64), np.array([t for s in sets for t in s[1]], dtype=np.int64),
                    np.array([v for s in sets for v in s[2]], dtype=np.uint64), pt_off)
    out, off, nw, _, _ = check_encode_against_oracle(ctx, b, gfmt, ofmt)
    ts, val, npts, st, rc = gpu_decode(ctx, out, off, nw, b.pt_off, gfmt)
    assert rc == 0 and not st.any()
    assert np.array_equal(ts, b.ts) and np.array_equal(val, b.val)
    assert np.array_equal(npts.astype(np.int64), np.diff(b.pt_off.astype(np.int64)))


@pytest.mark.parametrize("gfmt,ofmt", FMTS)
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_ragged_batches_match_oracle(ctx, gfmt, ofmt, seed):
    b = synth.ragged_mix(seed=seed, nseries=1000, pmax=400)
    out, off, nw, _, _ = check_encode_against_oracle(ctx, b, gfmt, ofmt)
    ts, val, npts, st, rc = gpu_decode(ctx, out, off, nw, b.pt_off, gfmt)
    rts, rval, rnp, rst = O.decode_batch(out, off, nw, b.pt_off, fmt=ofmt, nthreads=8)
    assert rc == 0 and np.array_equal(st, rst) and np.array_equal(npts, rnp)
    assert np.array_equal(ts, rts) and np.array_equal(val, rval)
    assert np.array_equal(ts, b.ts) and np.array_equal(val, b.val)


GENS = [
    ("random_walk", lambda: synth.random_walk(S=96, P=7200)),
    ("quantised_walk", lambda: synth.quantised_walk(S=96, P=7200)),
    ("gauge_integer", lambda: synth.gauge_integer(S=20000, P=120)),
    ("gauge_drift", lambda: synth.gauge_drift(S=20000, P=120)),
    ("adversarial", lambda: synth.adversarial(S=130, P=3000)),
    ("readme", lambda: synth.readme_workload(S=3, P=100000)),
    ("skewed", lambda: synth.skewed(total_points=1 << 20)),
    ("word_edges", lambda: synth.random_walk(S=300, P=63)),
]


@pytest.mark.parametrize("name,gen", GENS, ids=[g[0] for g in GENS])
def test_benchmark_shapes_match_oracle(ctx, name, gen):
    b = gen()
    for gfmt, ofmt in FMTS:
        uniform = b.uniform_npoints() is not None
        out, off, nw, d_out, d_nw = check_encode_against_oracle(ctx, b, gfmt, ofmt, uniform=uniform)
        cap = b.uniform_npoints() if uniform else None
        ts, val, npts, st, rc = gpu_decode(ctx, out, off, nw, b.pt_off, gfmt, uniform_cap=cap, d_words=d_out)
        assert rc == 0 and not st.any()
        assert np.array_equal(ts, b.ts) and np.array_equal(val, b.val)
        assert np.array_equal(npts.astype(np.int64), np.diff(b.pt_off.astype(np.int64)))


def test_uniform_and_csr_layouts_agree(ctx):
    b = synth.gauge_drift(S=5000, P=120)
    o1 = gpu_encode(ctx, b, G.FMT_V2, uniform=True)
    o2 = gpu_encode(ctx, b, G.FMT_V2, uniform=False)
    assert np.array_equal(o1[2], o2[2])
    nw = o1[2].astype(np.int64)
    for s in (0, 1, 77, 4999):
        a = o1[0][int(o1[1][s]): int(o1[1][s]) + nw[s]]
        c = o2[0][int(o2[1][s]): int(o2[1][s]) + nw[s]]
        assert np.array_equal(a, c)


@pytest.mark.parametrize("P,S", [(1, 70), (7, 1), (7, 300), (100, 65), (121, 1030), (123, 2500), (136, 129), (200, 64)])
def test_uniform_layouts_of_any_length(ctx, P, S):
    """Uniform layouts whose series start at every phase of a 128-byte line (the waves then take
    every 2nd, 4th, 8th or 16th series) and whose count is not a multiple of a wave's share."""
    b = synth.gauge_drift(S=S, P=P, seed=1000 + P)
    for gfmt, ofmt in FMTS:
        out, off, nw, st, rc, d_out, d_nw = gpu_encode(ctx, b, gfmt, uniform=True)
        assert rc == 0 and not st.any()
        ref_out, ref_off, ref_nw, _ = O.encode_batch(b.block_ts, b.ts, b.val, b.pt_off, out_off=off, fmt=ofmt, nthreads=4)
        assert np.array_equal(nw, ref_nw)
        for s in range(S):
            a0, n = int(off[s]), int(nw[s])
            assert np.array_equal(out[a0:a0 + n], ref_out[a0:a0 + n]), (P, S, s)
        ts, val, npts, st2, rc2 = gpu_decode(ctx, out, off, nw, b.pt_off, gfmt, uniform_cap=P, d_words=d_out)
        assert rc2 == 0 and not st2.any() and np.all(npts == P)
        assert np.array_equal(ts, b.ts) and np.array_equal(val, b.val)


def test_compaction(ctx):
    b = synth.ragged_mix(seed=11, nseries=3000, pmax=150)
    out, off, nw, d_out, d_nw = check_encode_against_oracle(ctx, b, G.FMT_V2, O.V2)
    S = b.nseries
    total = int(nw.sum())
    d_c = torch.full((total,), -1, dtype=torch.int64, device="cuda")
    d_co = torch.full((S + 1,), -1, dtype=torch.int64, device="cuda")
    ctx.compact_device(d_out, d_nw, d_c, d_co, slot_off=dev(off))
    ctx.sync()
    co = host_u64(d_co)
    expect = np.zeros(S + 1, dtype=np.uint64)
    np.cumsum(nw, out=expect[1:])
    assert np.array_equal(co, expect)
    c = host_u64(d_c)
    for s in range(0, S, 37):
        assert np.array_equal(c[int(co[s]): int(co[s + 1])], out[int(off[s]): int(off[s]) + int(nw[s])])
    # the compact form decodes as well (slot == image)
    ts, val, npts, st, rc = gpu_decode(ctx, c, co, None, b.pt_off, G.FMT_V2)
    assert rc == 0 and np.array_equal(ts, b.ts) and np.array_equal(val, b.val)


def test_slot_overflow_is_reported(ctx):
    b = synth.random_walk(S=200, P=500)
    lens = np.full(200, 100, dtype=np.int64)  # far too small for 500 random-walk points
    lens[::2] = G.max_words(500)
    off = slot_offsets(b, tight=lens)
    out, off, nw, st, rc, _, _ = gpu_encode(ctx, b, G.FMT_V2, out_off=off)
    assert rc == G.ERR_CAPACITY
    assert np.all(st[1::2] == G.ST_OVERFLOW) and not st[::2].any()
    ref_out, _, ref_nw, ref_st = O.encode_batch(b.block_ts, b.ts, b.val, b.pt_off, out_off=off)
    assert np.array_equal(nw, ref_nw) and np.array_equal(st, ref_st)
    # neighbours of an overflowing series are untouched
    for s in range(0, 200, 2):
        a = int(off[s])
        assert np.array_equal(out[a: a + int(nw[s])], ref_out[a: a + int(nw[s])])


def test_compaction_after_slot_overflow(ctx):
    """Compaction enqueued right after an encode that overflowed some slots, without a sync in
    between: an overflowed series reports the length it would have had, and the copy must not
    read past its slot (into the neighbour's image or past the end of the array).  Offsets follow
    the reported lengths; series that fit are copied whole; the sync reports the overflow."""
    b = synth.random_walk(S=200, P=500)
    lens = np.full(200, 100, dtype=np.int64)
    lens[::2] = G.max_words(500)
    lens[-1] = 100  # the last series overflows too: an unclamped copy would read past the end of out_words
    off = slot_offsets(b, tight=lens)
    S = b.nseries
    d_out = torch.full((int(off[-1]),), -1, dtype=torch.int64, device="cuda")
    d_nw = torch.zeros(S, dtype=torch.int64, device="cuda")
    d_off = dev(off)
    ref_out, _, ref_nw, ref_st = O.encode_batch(b.block_ts, b.ts, b.val, b.pt_off, out_off=off)
    total = int(ref_nw.sum())
    d_c = torch.full((total,), -1, dtype=torch.int64, device="cuda")
    d_co = torch.full((S + 1,), -1, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    ctx.encode_device(dev(b.block_ts), dev(b.ts), dev(b.val), d_out, d_nw, pt_off=dev(b.pt_off), out_off=d_off)
    ctx.compact_device(d_out, d_nw, d_c, d_co, slot_off=d_off)
    with pytest.raises(G.GtscError) as ei:
        ctx.sync()
    assert ei.value.code == G.ERR_CAPACITY
    nw, co, c = host_u64(d_nw), host_u64(d_co), host_u64(d_c)
    assert np.array_equal(nw, ref_nw)
    expect = np.zeros(S + 1, dtype=np.uint64)
    np.cumsum(nw, out=expect[1:])
    assert np.array_equal(co, expect)
    fill = np.uint64(0xFFFFFFFFFFFFFFFF)
    for s in range(S):
        a, n, slot = int(co[s]), int(nw[s]), int(lens[s])
        m = min(n, slot)
        assert np.array_equal(c[a:a + m], ref_out[int(off[s]):int(off[s]) + m]), s
        assert np.all(c[a + m:a + n] == fill), s  # beyond the slot nothing is copied


def test_decode_capacity_overflow_and_bad_streams(ctx):
    b = synth.random_walk(S=64, P=300)
    out, off, nw, _, _ = check_encode_against_oracle(ctx, b, G.FMT_V2, O.V2)
    # (1) capacity smaller than the series
    ts, val, npts, st, rc = gpu_decode(ctx, out, off, nw, b.pt_off, G.FMT_V2, uniform_cap=100)
    assert rc == G.ERR_CAPACITY and np.all(st == G.ST_OVERFLOW) and np.all(npts == 300)
    assert np.array_equal(ts.reshape(64, 100), b.ts.reshape(64, 300)[:, :100])
    # (2) truncated images: same counts and status as the oracle
    short = nw.copy()
    short[::3] = short[::3] // 2
    short[1::3] -= 1
    short[5] = 0
    ts, val, npts, st, rc = gpu_decode(ctx, out, off, short, b.pt_off, G.FMT_V2)
    rts, rval, rnp, rst = O.decode_batch(out, off, short, b.pt_off)
    assert rc == G.ERR_STREAM
    assert np.array_equal(st, rst) and np.array_equal(npts, rnp)
    for s in range(64):
        a, n = int(b.pt_off[s]), int(npts[s])
        assert np.array_equal(ts[a:a + n], rts[a:a + n]) and np.array_equal(val[a:a + n], rval[a:a + n])
    # (3) window reuse before any window: malformed
    import spec_codec as SP
    B = 1500400800000
    bits = format(B, "064b") + format(1000, "027b") + format(12345, "064b") + "0" + "10" + "1" * 40
    w = np.array(SP.words_image(bits), dtype=np.uint64)
    o = np.array([0, len(w)], dtype=np.uint64)
    ts, val, npts, st, rc = gpu_decode(ctx, w, o, None, np.array([0, 10], dtype=np.uint64), G.FMT_V2)
    rts, rval, rnp, rst = O.decode_batch(w, o, None, np.array([0, 10], dtype=np.uint64))
    assert rc == G.ERR_STREAM and st[0] == rst[0] == G.ST_MALFORMED and npts[0] == rnp[0] == 1


def check_decode_against_oracle(ctx, words, off, nw, cap_off, gfmt, ofmt, uniform_cap=None):
    ts, val, npts, st, rc = gpu_decode(ctx, words, off, nw, cap_off, gfmt, uniform_cap=uniform_cap)
    rts, rval, rnp, rst = O.decode_batch(words, off, nw, cap_off, fmt=ofmt)
    assert np.array_equal(st, rst) and np.array_equal(npts, rnp)
    assert (rc != 0) == bool(rst.any())
    for s in range(len(off) - 1):
        a0, a1 = int(cap_off[s]), int(cap_off[s + 1])
        n = min(int(npts[s]), a1 - a0)
        assert np.array_equal(ts[a0:a0 + n], rts[a0:a0 + n]) and np.array_equal(val[a0:a0 + n], rval[a0:a0 + n]), s
        assert np.all(ts[a0 + n:a1] == -7) and np.all(val[a0 + n:a1] == np.uint64(-7 & (2 ** 64 - 1))), s  # fill pattern untouched
    return st


@pytest.mark.parametrize("gfmt,ofmt", FMTS)
@pytest.mark.parametrize("seed", [1, 2])
def test_garbage_images_decode_like_the_oracle(ctx, gfmt, ofmt, seed):
    """Images that no compressor wrote -- random words, runs of zeros and ones, any declared
    length -- must give the oracle's counts, status words and points, and leave every byte beyond
    a row's points alone (uniform and per-series layouts, capacity smaller than many streams)."""
    rng = np.random.default_rng(seed)
    S, W, CAP = 3000, 24, 96
    words = rng.integers(0, 2 ** 63, S * W, dtype=np.int64).view(np.uint64) * np.uint64(2) + rng.integers(0, 2, S * W).astype(np.uint64)
    words[rng.random(S * W) < 0.3] = 0
    words[rng.random(S * W) < 0.1] = np.uint64(0xFFFFFFFFFFFFFFFF)
    off = np.arange(S + 1, dtype=np.uint64) * np.uint64(W)
    nw = rng.integers(0, W + 1, S).astype(np.uint64)
    cap_off = np.arange(S + 1, dtype=np.uint64) * np.uint64(CAP)
    st = check_decode_against_oracle(ctx, words, off, nw, cap_off, gfmt, ofmt, uniform_cap=CAP)
    assert len(np.unique(st)) >= 4  # the case mix really covers the status words
    # per-series layout: ragged slots and capacities
    lens = rng.integers(0, 40, S)
    roff = np.zeros(S + 1, dtype=np.uint64)
    np.cumsum(lens, out=roff[1:])
    rwords = words[: int(roff[-1])] if int(roff[-1]) <= len(words) else np.resize(words, int(roff[-1]))
    caps = rng.integers(0, 150, S)
    rcap = np.zeros(S + 1, dtype=np.uint64)
    np.cumsum(caps, out=rcap[1:])
    check_decode_against_oracle(ctx, rwords, roff, None, rcap, gfmt, ofmt)


@pytest.mark.parametrize("gfmt,ofmt", FMTS)
def test_damaged_images_decode_like_the_oracle(ctx, gfmt, ofmt):
    """Valid images with a few bits flipped: whatever the damage turns the stream into, the GPU
    decoder and the oracle read the same points out of it."""
    rng = np.random.default_rng(77)
    b = synth.ragged_mix(seed=8, nseries=1500, pmax=200)
    out, off, nw, _, _ = check_encode_against_oracle(ctx, b, gfmt, ofmt)
    bad = out.copy()
    total = int(off[-1])
    hits = rng.integers(0, total, total // 40)
    bad[hits] ^= np.uint64(1) << rng.integers(0, 64, len(hits)).astype(np.uint64)
    check_decode_against_oracle(ctx, bad, off, nw, b.pt_off, gfmt, ofmt)


@pytest.mark.parametrize("gfmt,ofmt", FMTS)
def test_quirks_first_ts_zero_and_int_wrap(ctx, gfmt, ofmt):
    B = 1500400800000
    series = [
        (0, [0, 1000, 2000, 3000], [5, 6, 7, 7]),                       # storedTimestamp == 0 after the first add
        (B, [B + 5, B + 5 + 2 ** 31 + 7, B + 5 + 2 ** 33], [1, 2, 3]),  # int overflow of the v2 delta
        (B, [B + 2 ** 27 + 5, B + 2 ** 27 + 6], [9, 9]),                # first delta beyond 27 bits (masked)
        (B, [B - 5, B + 6], [9, 10]),                                   # negative first delta
    ]
    pt_off = np.zeros(len(series) + 1, dtype=np.uint64)
    np.cumsum([len(s[1]) for s in series], out=pt_off[1:])
    b = synth.Batch(np.array([s[0] for s in series], dtype=np.int64),
                    np.array([t for s in series for t in s[1]], dtype=np.int64),
                    np.array([v for s in series for v in s[2]], dtype=np.uint64), pt_off)
    out, off, nw, _, _ = check_encode_against_oracle(ctx, b, gfmt, ofmt)
    # decoders agree with each other even where the stream does not round-trip
    cap = np.arange(len(series) + 1, dtype=np.uint64) * np.uint64(8)
    ts, val, npts, st, rc = gpu_decode(ctx, out, off, nw, cap, gfmt)
    rts, rval, rnp, rst = O.decode_batch(out, off, nw, cap, fmt=ofmt)
    assert np.array_equal(st, rst) and np.array_equal(npts, rnp)
    for s in range(len(series)):
        a, n = int(cap[s]), int(min(npts[s], 8))
        assert np.array_equal(ts[a:a + n], rts[a:a + n]) and np.array_equal(val[a:a + n], rval[a:a + n])


@pytest.mark.parametrize("gfmt,ofmt", FMTS)
def test_timestamp_high_half_changes_inside_a_chunk(ctx, gfmt, ofmt):
    """Timestamps that cross multiples of 2^32 at every position of a 16-point chunk, jump back
    and forth over one, and series whose high half changes at nearly every point: the decoder
    keeps one high half per row and chunk and must still return every timestamp exactly."""
    rng = np.random.default_rng(99)
    series = []
    for k in range(70):  # crossing after k points, steps of 1 s
        edge = (350 + k) << 32
        P = 40 + (k % 37)
        first = edge - 1000 * (k % 33) - 500
        series.append((first - 1000, first + 1000 * np.arange(P, dtype=np.int64)))
    for k in range(40):  # see-saw over an edge with steps near +-2^31
        edge = (400 + k) << 32
        P = 30 + k
        step = np.where(np.arange(P) % 2 == 0, 2 ** 31 - 1 - k, -(2 ** 31) + 1 + 3 * k).astype(np.int64)
        t = edge - 2 ** 30 + np.cumsum(step)
        series.append((int(t[0]) - 5, t))
    for k in range(30):  # random walk with steps up to +-2^31-1
        P = 20 + 5 * k
        t = ((500 + k) << 32) + np.cumsum(rng.integers(-(2 ** 31) + 1, 2 ** 31 - 1, P))
        series.append((int(t[0]) - 100, t.astype(np.int64)))
    pt_off = np.zeros(len(series) + 1, dtype=np.uint64)
    np.cumsum([len(s[1]) for s in series], out=pt_off[1:])
    ts = np.concatenate([s[1] for s in series]).astype(np.int64)
    val = rng.integers(0, 2 ** 63, len(ts), dtype=np.int64).view(np.uint64)
    val[::3] = val[1::3][: len(val[::3])] if len(val[1::3]) >= len(val[::3]) else val[::3]
    b = synth.Batch(np.array([s[0] for s in series], dtype=np.int64), ts, val, pt_off)
    out, off, nw, _, _ = check_encode_against_oracle(ctx, b, gfmt, ofmt)
    gts, gval, npts, st, rc = gpu_decode(ctx, out, off, nw, b.pt_off, gfmt)
    rts, rval, rnp, rst = O.decode_batch(out, off, nw, b.pt_off, fmt=ofmt)
    assert np.array_equal(st, rst) and np.array_equal(npts, rnp)
    assert np.array_equal(gts, rts) and np.array_equal(gval, rval)
    # the regular-cadence group round-trips (the others overflow the format's int delta-of-delta)
    n_reg = int(b.pt_off[70])
    assert np.array_equal(gts[:n_reg], b.ts[:n_reg]) and np.array_equal(gval[:n_reg], b.val[:n_reg])
    # a smaller capacity than the series has: nothing may be written past a row's slot
    cap_off = (b.pt_off // np.uint64(2)).astype(np.uint64)
    gts2, gval2, npts2, st2, rc2 = gpu_decode(ctx, out, off, nw, cap_off, gfmt)
    rts2, rval2, rnp2, rst2 = O.decode_batch(out, off, nw, cap_off, fmt=ofmt)
    assert np.array_equal(npts2, rnp2) and np.array_equal(st2, rst2)
    for s in range(len(series)):
        a0, a1 = int(cap_off[s]), int(cap_off[s + 1])
        n = min(int(npts2[s]), a1 - a0)
        assert np.array_equal(gts2[a0:a0 + n], rts2[a0:a0 + n]) and np.array_equal(gval2[a0:a0 + n], rval2[a0:a0 + n])
        assert np.all(gts2[a0 + n:a1] == -7)  # untouched fill pattern


def test_host_entry_points(ctx):
    b = synth.ragged_mix(seed=5, nseries=2500, pmax=300)
    for gfmt, ofmt in FMTS:
        words, off = ctx.encode_host(b.block_ts, b.ts, b.val, pt_off=b.pt_off, fmt=gfmt)
        ref_out, ref_off, ref_nw, _ = O.encode_batch(b.block_ts, b.ts, b.val, b.pt_off, fmt=ofmt, nthreads=8)
        expect_off = np.zeros(b.nseries + 1, dtype=np.uint64)
        np.cumsum(ref_nw, out=expect_off[1:])
        assert np.array_equal(off, expect_off)
        for s in range(0, b.nseries, 13):
            assert np.array_equal(words[int(off[s]): int(off[s + 1])],
                                  ref_out[int(ref_off[s]): int(ref_off[s]) + int(ref_nw[s])])
        ts, val, npts, st = ctx.decode_host(words, off, pt_off=b.pt_off, fmt=gfmt)
        assert not st.any() and np.array_equal(ts, b.ts) and np.array_equal(val, b.val)
    # uniform layout and too small an output buffer
    u = synth.gauge_integer(S=3000, P=120)
    words, off = ctx.encode_host(u.block_ts, u.ts, u.val, uniform_npoints=120)
    ts, val, npts, st = ctx.decode_host(words, off, uniform_capacity=120)
    assert np.array_equal(ts, u.ts) and np.array_equal(val, u.val) and np.all(npts == 120)
    with pytest.raises(G.GtscError) as ei:
        ctx.encode_host(u.block_ts, u.ts, u.val, uniform_npoints=120, capacity_words=1000)
    assert ei.value.code == G.ERR_CAPACITY


def test_host_entry_points_many_chunks():
    """The chunked two-stream pipeline with chunks of a few thousand points (own process: the chunk
    size is read once per process)."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, GTSC_CHUNK_POINTS="3000")
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_chunks_worker.py")],
                       env=env, capture_output=True, text=True, timeout=170)
    assert r.returncode == 0 and "chunks ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def test_full_size_1m_series_round_trip(ctx):
    """The headline configuration of the benchmark: 1M series x 120 points.  Too big for a word-by-word oracle
    comparison in test time, so: (a) decode(encode(x)) == x on the device for all
    120M points, (b) image lengths within the bound, (c) a strided sample of series
    checked against the oracle byte for byte."""
    S, P = 1_000_000, 120
    g = torch.Generator(device="cuda").manual_seed(0x5EED0002)
    start = torch.randint(0, 1001, (S, 1), device="cuda", generator=g).double()
    r = torch.rand((S, P), device="cuda", generator=g)
    step = torch.where(r < 0.05, -1.0, torch.where(r < 0.10, 1.0, 0.0)).double()
    step[:, 0] = 0
    v = (start + torch.cumsum(step, dim=1)).contiguous()
    d_val = v.view(torch.int64).reshape(-1)
    d_ts = (synth.BLOCK_TS + 60000 * torch.arange(1, P + 1, device="cuda", dtype=torch.int64)).repeat(S, 1).reshape(-1).contiguous()
    d_block = torch.full((S,), synth.BLOCK_TS, dtype=torch.int64, device="cuda")
    slot = G.max_words(P)
    d_out = torch.zeros(S * slot, dtype=torch.int64, device="cuda")
    d_nw = torch.zeros(S, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    ctx.encode_device(d_block, d_ts, d_val, d_out, d_nw, uniform_npoints=P, uniform_slot_words=slot)
    ctx.sync()
    assert int(d_nw.max()) <= slot and int(d_nw.min()) >= 3
    d_ts2 = torch.zeros_like(d_ts)
    d_val2 = torch.zeros_like(d_val)
    d_np = torch.zeros(S, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()  # the context has its own stream: torch's fills must land first
    ctx.decode_device(d_out, d_ts2, d_val2, d_np, S, uniform_slot_words=slot, in_nwords=d_nw, uniform_capacity=P)
    ctx.sync()
    assert bool((d_np == P).all())
    assert torch.equal(d_ts2, d_ts) and torch.equal(d_val2, d_val)
    # sample against the oracle
    idx = torch.arange(0, S, 997, device="cuda")
    hb = synth.Batch(np.full(len(idx), synth.BLOCK_TS, dtype=np.int64),
                     d_ts.view(S, P)[idx].reshape(-1).cpu().numpy(),
                     d_val.view(S, P)[idx].reshape(-1).cpu().numpy().view(np.uint64),
                     np.arange(len(idx) + 1, dtype=np.uint64) * np.uint64(P))
    ref_out, ref_off, ref_nw, _ = O.encode_batch(hb.block_ts, hb.ts, hb.val, hb.pt_off, nthreads=8)
    got = host_u64(d_out.view(S, slot)[idx])
    got_nw = host_u64(d_nw[idx])
    assert np.array_equal(got_nw, ref_nw)
    for i in range(len(idx)):
        n = int(ref_nw[i])
        assert np.array_equal(got[i, :n], ref_out[int(ref_off[i]): int(ref_off[i]) + n])


def test_randomised_shapes_layouts_and_alignments(ctx):
    """A slice of the sweep in tools/stress_parity.py: ragged and uniform batches of both formats in
    arrays that start at any 8-byte phase of a line (timestamps and values on different phases
    included, which selects the compressor's form for split phases), tight and loose slots,
    truncated images, capacities above and below the series' lengths; compressed words, decoded
    points, counts, status words and the guard words around every array against the oracle."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "stress_parity", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "stress_parity.py"))
    sp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sp)
    for seed in range(1000, 1400):
        sp.one_case(ctx, seed)


@pytest.mark.parametrize("gfmt,ofmt", FMTS)
def test_zero_timestamps_inside_full_chunks(ctx, gfmt, ofmt):
    """storedTimestamp == 0 makes the reference treat the next point as a first one again.  Here
    the zeros sit in the middle of long series, at even and odd positions, in a uniform batch
    whose chunks are all full: the kernels' unguarded paths have to notice them."""
    S, P = 128, 96
    rng = np.random.default_rng(5)
    ts = 1000 * np.arange(1, P + 1, dtype=np.int64)[None, :].repeat(S, 0) + rng.integers(0, 50, size=(S, P))
    for s in range(0, S, 3):
        ts[s, 5 + (s % 80)] = 0
    ts[7, 20:24] = 0
    ts[9, 0] = 0
    val = synth.gauge_integer(S=S, P=P, seed=3).val
    b = synth.Batch(np.zeros(S, dtype=np.int64), np.ascontiguousarray(ts.reshape(-1)), val,
                    np.arange(S + 1, dtype=np.uint64) * np.uint64(P))
    for uniform in (True, False):
        out, off, nw, _, _ = check_encode_against_oracle(ctx, b, gfmt, ofmt, uniform=uniform)
        gts, gval, npts, st, rc = gpu_decode(ctx, out, off, nw, b.pt_off, gfmt)
        rts, rval, rnp, rst = O.decode_batch(out, off, nw, b.pt_off, fmt=ofmt)
        assert np.array_equal(st, rst) and np.array_equal(npts, rnp)
        for s in range(S):
            a, n = int(b.pt_off[s]), int(min(npts[s], P))
            assert np.array_equal(gts[a:a + n], rts[a:a + n]) and np.array_equal(gval[a:a + n], rval[a:a + n])


def _run_skip_batch(S=256, P=200, seed=11):
    """Gauges (one value in ten changes, regular cadence: the streams the decompressor skips runs in) with rows of
    every kind that must drop a run-skipping attempt, or bend it, mixed into each wave of 64."""
    rng = np.random.default_rng(seed)
    g = synth.gauge_integer(S=S, P=P, seed=seed)
    ts = g.ts.reshape(S, P).copy()
    val = g.val.reshape(S, P).copy()
    block = g.block_ts.copy()
    step = np.arange(1, P + 1, dtype=np.int64)
    for w in range(0, S, 64):
        block[w + 3] = (350 << 32) - 60000 * 100            # the high half of the timestamps moves at point 100
        ts[w + 3] = block[w + 3] + 60000 * step
        val[w + 5] = val[w + 5, 0]                           # a constant: nothing but runs
        block[w + 7] = 1000                                  # timestamps below 2^32
        ts[w + 7] = 1000 + step
        ts[w + 9, 77:] += 5                                  # one irregular interval
        block[w + 11] = -3000                                # negative timestamps that reach zero at point 50
        ts[w + 11] = -3000 + 60 * step
        block[w + 13] = 1 << 40                              # timestamps that run backwards
        ts[w + 13] = (1 << 40) - 60000 * step
        block[w + 15] = 0                                    # a delta of 2^33
        ts[w + 15] = (1 << 33) * step
        val[w + 17] = rng.integers(0, 1 << 63, size=P, dtype=np.int64).view(np.uint64)  # a dense row among sparse ones
        val[w + 19, 60:] = rng.integers(0, 1 << 63, dtype=np.int64)                      # a long run, then all 64 bits change
        val[w + 21, ::7] ^= np.uint64(0x8000000000000001)     # payloads of 64 bits every seventh point
        ts[w + 23, 120] = ts[w + 23, 119]                     # a repeated timestamp (delta 0 from there)
        ts[w + 23, 121:] = ts[w + 23, 119]
        for r_, every in ((25, 40), (27, 16), (29, 5), (31, 2)):  # late samples: the interval changes and comes back
            late = rng.random(P) < 1.0 / every
            ts[w + r_] += np.where(late, rng.integers(1, 3000, size=P), 0)
        ts[w + 33] += np.cumsum(rng.integers(-30000, 30000, size=P))  # a new interval at every point, some of them negative
        ts[w + 35, 15::16] += 7                                # the last slot of a chunk, every chunk
        ts[w + 37, 16::16] += 7                                # the first slot
        ts[w + 39, 14::16] += 7; ts[w + 39, 15::16] += 9       # the last two
    return synth.Batch(block, np.ascontiguousarray(ts.reshape(-1)), np.ascontiguousarray(val.reshape(-1)), g.pt_off)


@pytest.mark.parametrize("gfmt,ofmt", FMTS)
def test_run_skipping_decoder_matches_oracle(ctx, gfmt, ofmt):
    """Sparse streams are decoded a run of unchanged points at a time (decode_tiled_kernel).  Whole waves of gauges
    with the awkward rows of _run_skip_batch among them: points, counts and status as the oracle's decoder gives
    them, with capacities at, above and below the block length, images cut short, and images of zero words."""
    b = _run_skip_batch()
    S, P = b.nseries, b.uniform_npoints()
    out, off, nw, d_out, d_nw = check_encode_against_oracle(ctx, b, gfmt, ofmt, uniform=True)
    for cap in (P, P + 3, P - 5, 16 * (P // 16)):
        cap_off = np.arange(S + 1, dtype=np.uint64) * np.uint64(cap)
        gts, gval, npts, st, rc = gpu_decode(ctx, out, off, nw, cap_off, gfmt, uniform_cap=cap)
        rts, rval, rnp, rst = O.decode_batch(out, off, nw, cap_off, fmt=ofmt, nthreads=8)
        assert np.array_equal(st, rst) and np.array_equal(npts, rnp), cap
        for s in range(S):
            a, n = s * cap, int(min(npts[s], cap))
            assert np.array_equal(gts[a:a + n], rts[a:a + n]) and np.array_equal(gval[a:a + n], rval[a:a + n]), (cap, s)
            assert np.all(gts[a + n:a + cap] == -7) and np.all(gval[a + n:a + cap] == np.uint64(2 ** 64 - 7)), (cap, s)
    # damaged input: images cut short anywhere, and images whose words from some point on are zero
    # (an endless run of unchanged points as far as the bits say)
    rng = np.random.default_rng(5)
    cut = nw.copy()
    some = rng.random(S) < 0.25
    cut[some] = (cut[some] * rng.random(int(some.sum()))).astype(np.uint64)
    zeroed = out.copy()
    for s in np.flatnonzero(rng.random(S) < 0.25):
        a = int(off[s]) + int(rng.integers(1, int(nw[s])))
        zeroed[a:int(off[s + 1])] = 0
    cap_off = np.arange(S + 1, dtype=np.uint64) * np.uint64(P)
    for words, in_nw in ((out, cut), (zeroed, nw), (zeroed, np.full(S, int(off[1]), dtype=np.uint64))):
        gts, gval, npts, st, rc = gpu_decode(ctx, words, off, in_nw, cap_off, gfmt, uniform_cap=P)
        rts, rval, rnp, rst = O.decode_batch(words, off, in_nw, cap_off, fmt=ofmt, nthreads=8)
        assert np.array_equal(st, rst) and np.array_equal(npts, rnp)
        for s in range(S):
            a, n = s * P, int(min(npts[s], P))
            assert np.array_equal(gts[a:a + n], rts[a:a + n]) and np.array_equal(gval[a:a + n], rval[a:a + n]), s
            assert np.all(gts[a + n:a + P] == -7), s


def test_lock_step_decoder_in_its_own_process():
    """GTSC_RUN_SKIP=0 keeps the 8-word form of the lane-per-series decompressor on its lock-step loop for every
    chunk (the loop that run skipping falls back to); read once per process, so a child runs the sweep and the
    tests of sparse streams with it."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GTSC_RUN_SKIP="0")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_parity.py"), "6", "12000"], env=env,
                       capture_output=True, text=True, timeout=170)
    assert r.returncode == 0 and "stress ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider",
                        "-k", "run_skipping or zero_timestamps or high_half or garbage"],
                       env=env, capture_output=True, text=True, timeout=300, cwd=root)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_lane_per_series_kernels_in_their_own_process():
    """The plain lane-per-series kernels (both formats) are what GTSC_KERNEL=direct selects; the choice is read
    once per process, so the sweep runs in a child process for a few seconds."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GTSC_KERNEL="direct")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_parity.py"), "8", "7000"], env=env,
                       capture_output=True, text=True, timeout=170)
    assert r.returncode == 0 and "stress ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def test_decode_ring_choice_is_explicit(ctx):
    """gtsc_set_decode_ring pins the form of the lane-per-series decompressor (8- or 16-word input rings):
    same points, counts and status either way, on sparse and on dense streams; with 0 both forms are launched and
    the batch's density, measured on the device, says which of them works."""
    try:
        for b in (synth.gauge_integer(S=3000, P=120), synth.adversarial(S=200, P=700)):
            out, off, nw, _, _ = check_encode_against_oracle(ctx, b, G.FMT_V2, O.V2, uniform=True)
            got = []
            for ring in (8, 16, 0):
                ctx.set_decode_ring(ring)
                ts, val, npts, st, rc = gpu_decode(ctx, out, off, nw, b.pt_off, G.FMT_V2, uniform_cap=b.uniform_npoints())
                assert rc == 0 and not st.any()
                assert np.array_equal(ts, b.ts) and np.array_equal(val, b.val)
                k = ctx.last_dispatch()
                assert bool(k & G.DISPATCH_LANE_BOTH_RINGS) == (ring == 0) and bool(k & G.DISPATCH_LANE_RING16) == (ring == 16)
        with pytest.raises(G.GtscError):
            ctx.set_decode_ring(12)
    finally:
        ctx.set_decode_ring(8)


def test_wave_per_block_kernels_forced_in_their_own_process():
    """The wave-per-block kernels normally take blocks of 1,024 points and more (the decompressor only in
    batches of few such blocks; the compressor's workgroup-per-block form batches of up to 384 blocks).  GTSC_LONG_POINTS=0 sends every block to them, whatever its length,
    GTSC_LONG_DECODE_MAX(_DENSE) lift the batch limits and GTSC_WG_DECODE_MAX=0 keeps the workgroup-per-block
    decompressor out of it; the variables preset the tunables of every context created in a process, so a child process runs
    the randomised sweep and the parity tests that hold the odd cases (benchmark shapes incl. the 100,000-point
    blocks and skewed lengths, garbage and damaged images, quirks, zero timestamps, slot overflow, capacities)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GTSC_LONG_POINTS="0", GTSC_LONG_DECODE_MAX="1000000000", GTSC_LONG_DECODE_MAX_DENSE="1000000000",
               GTSC_WG_DECODE_MAX="0", GTSC_WG_DECODE_VLONG_MAX="0")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_parity.py"), "8", "9000"], env=env,
                       capture_output=True, text=True, timeout=170)
    assert r.returncode == 0 and "stress ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
    # batches of up to 384 blocks are compressed by a workgroup of waves per block; GTSC_COOP_MAX=0 leaves
    # them to the wave-per-block kernel as well
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_parity.py"), "5", "7000"], env=dict(env, GTSC_COOP_MAX="0"),
                       capture_output=True, text=True, timeout=170)
    assert r.returncode == 0 and "stress ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider",
                        "-k", "benchmark_shapes or garbage or damaged or quirks or zero_timestamps or high_half or overflow "
                              "or known_answer or reference_test_inputs or ragged_batches or uniform_layouts"],
                       env=env, capture_output=True, text=True, timeout=300, cwd=root)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
