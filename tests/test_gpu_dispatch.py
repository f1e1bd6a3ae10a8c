"""GPU tests of how the library chooses its kernels: the choice must follow from the batch alone (never from what a
context coded before), every kernel a batch can be sent to must give the oracle's output on that batch, and the
tunables only move blocks between kernels."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import pyoracle as O  # noqa: E402
import pygtsc as G  # noqa: E402
from pygtsc import synth  # noqa: E402

from test_gpu_parity import FMTS, check_decode_against_oracle, check_encode_against_oracle, gpu_decode  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HUGE = 1 << 62


def gauge_with_late_samples(S, P, seed, one_in=50):
    """Integer gauges whose samples are now and then a few milliseconds late (timestamp codes other than '0')."""
    b = synth.gauge_integer(S=S, P=P, seed=seed)
    rng = np.random.default_rng(seed)
    late = (rng.random(S * P) < 1.0 / one_in) * rng.integers(1, 40, S * P)
    b.ts = b.ts + late.astype(np.int64)
    return b


def damage(words, off, nw, rng):
    """Three kinds of broken images from good ones: cut short, tails zeroed, single bits flipped."""
    cut = nw.copy()
    cut[::2] = cut[::2] * 2 // 3
    cut[1::4] -= 1
    zeroed = words.copy()
    for s in range(0, len(nw), 3):
        a, n = int(off[s]), int(nw[s])
        zeroed[a + n // 2: a + n] = 0
    flipped = words.copy()
    total = int(off[-1])
    hits = rng.integers(0, total, max(total // 500, 8))
    flipped[hits] ^= np.uint64(1) << rng.integers(0, 64, len(hits)).astype(np.uint64)
    return cut, zeroed, flipped


@pytest.mark.parametrize("ring", [8, 16])
def test_lane_per_series_kernels_on_long_rows(ring):
    """What batches of very many long blocks run on -- the lane-per-series kernels on 7,200-point and 100,000-point rows --
    compared with the oracle at that shape: a context whose "long" threshold is out of reach sends every block to them.
    Dense rows, jittered timestamps, gauges with late samples; both ring forms; images cut short, with zeroed tails
    and flipped bits; capacities below the length."""
    ctx = G.Context(0)
    try:
        ctx.set_tunable(G.TUNE_LONG_POINTS, HUGE)
        ctx.set_decode_ring(ring)
        rng = np.random.default_rng(5)
        batches = [synth.random_walk(S=70, P=7200, seed=11), synth.adversarial(S=66, P=7200, seed=12),
                   gauge_with_late_samples(130, 7200, 13), synth.readme_workload(S=2, P=100000),
                   synth.random_walk(S=3, P=100000, seed=14)]
        for b in batches:
            P = b.uniform_npoints()
            for gfmt, ofmt in FMTS:
                out, off, nw, d_out, _ = check_encode_against_oracle(ctx, b, gfmt, ofmt, uniform=True)
                assert ctx.last_dispatch() & (G.DISPATCH_WAVE | G.DISPATCH_COOP | G.DISPATCH_WG) == 0
                ts, val, npts, st, rc = gpu_decode(ctx, out, off, nw, b.pt_off, gfmt, uniform_cap=P, d_words=d_out)
                k = ctx.last_dispatch()
                assert k & G.DISPATCH_LANE_TILED and not k & (G.DISPATCH_WAVE | G.DISPATCH_WG)
                assert bool(k & G.DISPATCH_LANE_RING16) == (ring == 16)
                assert rc == 0 and not st.any() and np.array_equal(ts, b.ts) and np.array_equal(val, b.val)
                if P > 10000 and gfmt == G.FMT_V1:
                    continue  # (the oracle's decoder over the damaged 100,000-point rows once is enough)
                cut, zeroed, flipped = damage(out, off, nw, rng)
                cap_off = np.arange(b.nseries + 1, dtype=np.uint64) * np.uint64(P)
                check_decode_against_oracle(ctx, out, off, cut, cap_off, gfmt, ofmt, uniform_cap=P)
                check_decode_against_oracle(ctx, zeroed, off, nw, cap_off, gfmt, ofmt, uniform_cap=P)
                check_decode_against_oracle(ctx, flipped, off, nw, cap_off, gfmt, ofmt, uniform_cap=P)
                small = P * 3 // 4 + 5
                small_off = np.arange(b.nseries + 1, dtype=np.uint64) * np.uint64(small)
                check_decode_against_oracle(ctx, out, off, nw, small_off, gfmt, ofmt, uniform_cap=small)
    finally:
        ctx.close()


def decode_and_report(ctx, words, off, nw, pt_off, uniform_cap=None):
    ts, val, npts, st, rc = gpu_decode(ctx, words, off, nw, pt_off, G.FMT_V2, uniform_cap=uniform_cap)
    return (ts, val, npts, st, rc), ctx.last_dispatch(with_plan=True)


def test_dispatch_follows_from_the_batch_alone():
    """The same batch takes the same kernels on a fresh context and on one that has coded unrelated data before
    (dense, sparse, long, short; with syncs in between), and the device makes the same plan of it."""
    seasoned = G.Context(0)
    try:
        for hist in (synth.adversarial(S=300, P=2000), synth.gauge_integer(S=3000, P=120), synth.gauge_integer(S=700, P=3000, seed=2)):
            out, off, nw, d_out, _ = check_encode_against_oracle(seasoned, hist, G.FMT_V2, O.V2, uniform=True)
            gpu_decode(seasoned, out, off, nw, hist.pt_off, G.FMT_V2, uniform_cap=hist.uniform_npoints(), d_words=d_out)
        skew = synth.skewed(total_points=150000, seed=3, pmin=10, pmax=30000)
        strayed = synth.random_walk(S=40, P=3000, seed=16)   # dense values at scrape times a few ms off the cadence
        strayed.ts = strayed.ts + np.random.default_rng(16).integers(-40, 41, strayed.ts.size)
        cases = [
            ("short uniform", synth.gauge_integer(S=3000, P=120, seed=5), True, G.DISPATCH_LANE_TILED, G.DISPATCH_WAVE | G.DISPATCH_WG),
            ("few long dense", synth.random_walk(S=40, P=3000, seed=6), True, G.DISPATCH_WAVE | G.DISPATCH_PLAN, 0),
            ("few long sparse", synth.gauge_integer(S=40, P=3000, seed=7), True, G.DISPATCH_WG | G.DISPATCH_PLAN, 0),
            ("few long dense, timestamps off the cadence", strayed, True, G.DISPATCH_WG | G.DISPATCH_PLAN, 0),
            ("equal-sized points, window changes at powers of two", synth.readme_workload(S=3, P=50000), True, G.DISPATCH_WG | G.DISPATCH_PLAN, 0),
            ("skewed lengths", skew, False, G.DISPATCH_PLAN | G.DISPATCH_WG | G.DISPATCH_WAVE | G.DISPATCH_LANE_TILED, 0),
        ]
        for name, b, uniform, must, must_not in cases:
            fresh = G.Context(0)
            try:
                enc = []
                for c in (fresh, seasoned):
                    out, off, nw, d_out, _ = check_encode_against_oracle(c, b, G.FMT_V2, O.V2, uniform=uniform)
                    enc.append(c.last_dispatch())
                assert enc[0] == enc[1], name
                cap = b.uniform_npoints() if uniform else None
                (r0, (k0, p0)) = decode_and_report(fresh, out, off, nw, b.pt_off, cap)
                (r1, (k1, p1)) = decode_and_report(seasoned, out, off, nw, b.pt_off, cap)
                assert (k0, p0) == (k1, p1), (name, k0, k1, p0, p1)
                assert k0 & must == must and k0 & must_not == 0, (name, k0)
                for r in (r0, r1):
                    assert r[4] == 0 and not r[3].any() and np.array_equal(r[0], b.ts) and np.array_equal(r[1], b.val), name
                if k0 & G.DISPATCH_PLAN:
                    lens = np.diff(b.pt_off.astype(np.int64))
                    assert p0[0] == int((lens >= 1024).sum()) and p0[1] == int((lens >= 16384).sum()), name
                    assert p0[2] == int(nw[lens >= 1024].sum()) and p0[3] == int(lens[lens >= 1024].sum()), name
            finally:
                fresh.close()
    finally:
        seasoned.close()


def test_tunables_move_blocks_between_kernels_not_results():
    """Every long block through each of the three decompressors in turn, by tunables on one context: the same
    points, counts and status words each time (good images and damaged ones), and the dispatch says who ran."""
    ctx = G.Context(0)
    try:
        rng = np.random.default_rng(9)
        b = synth.skewed(total_points=400000, seed=8, pmin=10, pmax=60000)
        for gfmt, ofmt in FMTS:
            out, off, nw, _, _ = check_encode_against_oracle(ctx, b, gfmt, ofmt)
            cut, zeroed, flipped = damage(out, off, nw, rng)
            settings = {
                "lanes": {G.TUNE_LONG_POINTS: HUGE},
                "waves": {G.TUNE_WG_DECODE_MAX: 0, G.TUNE_WG_DECODE_VLONG_MAX: 0, G.TUNE_WAVE_DECODE_MAX: HUGE, G.TUNE_WAVE_DECODE_MAX_DENSE: HUGE},
                "workgroups": {G.TUNE_WG_DECODE_MAX: HUGE, G.TUNE_WG_DENSE_BITS: 1 << 30},
                "workgroups, every block": {G.TUNE_LONG_POINTS: 0, G.TUNE_WG_DECODE_MAX: HUGE, G.TUNE_WG_DENSE_BITS: 1 << 30,
                                            G.TUNE_WAVE_DECODE_MAX: HUGE, G.TUNE_WAVE_DECODE_MAX_DENSE: HUGE},
            }
            expect = {"lanes": G.DISPATCH_LANE_TILED, "waves": G.DISPATCH_WAVE, "workgroups": G.DISPATCH_WG, "workgroups, every block": G.DISPATCH_WG}
            for name, tun in settings.items():
                before = {k: ctx.get_tunable(k) for k in tun}
                try:
                    for k, v in tun.items():
                        ctx.set_tunable(k, v)
                    check_decode_against_oracle(ctx, out, off, nw, b.pt_off, gfmt, ofmt)
                    assert ctx.last_dispatch() & expect[name], name
                    check_decode_against_oracle(ctx, out, off, cut, b.pt_off, gfmt, ofmt)
                    check_decode_against_oracle(ctx, zeroed, off, nw, b.pt_off, gfmt, ofmt)
                    check_decode_against_oracle(ctx, flipped, off, nw, b.pt_off, gfmt, ofmt)
                finally:
                    for k, v in before.items():
                        ctx.set_tunable(k, v)
        with pytest.raises(G.GtscError):
            ctx.set_tunable(999, 1)
        with pytest.raises(G.GtscError):
            ctx.set_tunable(G.TUNE_DECODE_RING, 12)
    finally:
        ctx.close()


@pytest.mark.parametrize("kernel", ["by the batch", "workgroups", "waves", "lanes"])
def test_long_blocks_whose_character_changes_along_the_block(kernel):
    """A slice of tools/stress_long_blocks.py: a few blocks of 1,000 to 60,000 points made of stretches of different
    kinds (constant, gauge, counter, random walk, random bits; regular, jittered and wild timestamps; timestamps that
    pass through zero), both formats, capacities exact, short and generous, images cut short, with zeroed tails and
    flipped bits -- with the kernels the batch gets by itself and with every long block sent to each decompressor in
    turn.  (The whole sweep has been run for 16,000 cases through the workgroup-per-block kernel and 7,000 and 6,000
    through the other two.)"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import stress_long_blocks as SL
    tunables = {"by the batch": {},
                "workgroups": {G.TUNE_WG_DECODE_MAX: HUGE, G.TUNE_WG_DENSE_BITS: 1 << 30},
                "waves": {G.TUNE_WG_DECODE_MAX: 0, G.TUNE_WG_DECODE_VLONG_MAX: 0, G.TUNE_WAVE_DECODE_MAX: HUGE, G.TUNE_WAVE_DECODE_MAX_DENSE: HUGE},
                "lanes": {G.TUNE_WG_DECODE_MAX: 0, G.TUNE_WG_DECODE_VLONG_MAX: 0, G.TUNE_WAVE_DECODE_MAX: 0, G.TUNE_WAVE_DECODE_MAX_DENSE: 0}}[kernel]
    must = {"by the batch": 0, "workgroups": G.DISPATCH_WG, "waves": G.DISPATCH_WAVE, "lanes": G.DISPATCH_LANE_TILED}[kernel]
    ctx = G.Context(0)
    try:
        for which, value in tunables.items():
            ctx.set_tunable(which, value)
        for seed in (13, 77, 5003, 5100):   # (13: the batch on which a race between the walker and the table makers showed)
            _, _, seen = SL.run_case(ctx, seed)
            assert seen & must == must
    finally:
        ctx.close()


def test_workgroup_per_block_decoder_draws_its_blocks_largest_first():
    """Batches of more series than the device has compute units: the workgroup-per-block decompressor lists its blocks
    by size class on the device and its workgroups draw them largest first -- here with every block sent to it.  Skewed
    lengths (about 700 series, every class of the list in use), and 2,400 series of one size (more than a class holds:
    the workgroups fall back to going through the series by number).  Good and damaged images against the oracle."""
    ctx = G.Context(0)
    try:
        for k, v in {G.TUNE_LONG_POINTS: 0, G.TUNE_WG_DECODE_MAX: HUGE, G.TUNE_WG_DENSE_BITS: 1 << 30,
                     G.TUNE_WAVE_DECODE_MAX: HUGE, G.TUNE_WAVE_DECODE_MAX_DENSE: HUGE}.items():
            ctx.set_tunable(k, v)
        rng = np.random.default_rng(21)
        skewed = synth.skewed(total_points=1500000, seed=31, pmin=10, pmax=20000)
        same = synth.random_walk(S=2400, P=300, seed=32)
        assert skewed.nseries > 400
        for b, fmts in ((skewed, FMTS), (same, FMTS[:1])):
            for gfmt, ofmt in fmts:
                out, off, nw, _, _ = check_encode_against_oracle(ctx, b, gfmt, ofmt)
                check_decode_against_oracle(ctx, out, off, nw, b.pt_off, gfmt, ofmt)
                assert ctx.last_dispatch() & G.DISPATCH_WG
                cut, zeroed, flipped = damage(out, off, nw, rng)
                check_decode_against_oracle(ctx, out, off, cut, b.pt_off, gfmt, ofmt)
                check_decode_against_oracle(ctx, zeroed, off, nw, b.pt_off, gfmt, ofmt)
                check_decode_against_oracle(ctx, flipped, off, nw, b.pt_off, gfmt, ofmt)
    finally:
        ctx.close()


def test_sparse_hint_sends_many_long_blocks_to_the_lane_per_series_compressor():
    """GTSC_TUNE_SPARSE_HINT (with the block count from which it applies set to zero): the same batch compressed a
    wavefront per block and a lane per block gives the oracle's words both ways, in both formats."""
    ctx = G.Context(0)
    try:
        b = synth.gauge_integer(S=700, P=3000, seed=21)
        for gfmt, ofmt in FMTS:
            check_encode_against_oracle(ctx, b, gfmt, ofmt, uniform=True)
            assert ctx.last_dispatch() & G.DISPATCH_WAVE and not ctx.last_dispatch() & G.DISPATCH_LANE_TILED
            ctx.set_tunable(G.TUNE_SPARSE_HINT, 1)
            ctx.set_tunable(G.TUNE_SPARSE_ENCODE_MIN, 0)
            out, off, nw, d_out, _ = check_encode_against_oracle(ctx, b, gfmt, ofmt, uniform=True)
            assert ctx.last_dispatch() & G.DISPATCH_LANE_TILED and not ctx.last_dispatch() & G.DISPATCH_WAVE
            ts, val, npts, st, rc = gpu_decode(ctx, out, off, nw, b.pt_off, gfmt, uniform_cap=3000, d_words=d_out)
            assert rc == 0 and not st.any() and np.array_equal(ts, b.ts) and np.array_equal(val, b.val)
            ctx.set_tunable(G.TUNE_SPARSE_HINT, 0)
    finally:
        ctx.close()


def test_workgroup_per_block_decoder_forced_in_its_own_process():
    """The workgroup-per-block decompressor normally takes the sparse long blocks of batches of few.  With GTSC_LONG_POINTS=0,
    GTSC_WG_DECODE_MAX lifted and every image counted as sparse it takes every block, whatever its length: a child
    process (the variables preset the tunables of every context created in it) runs the randomised sweep and the
    parity tests that hold the odd cases."""
    env = dict(os.environ, GTSC_LONG_POINTS="0", GTSC_WG_DECODE_MAX="1000000000", GTSC_WG_DENSE_BITS="1000000000",
               GTSC_LONG_DECODE_MAX="1000000000", GTSC_LONG_DECODE_MAX_DENSE="1000000000")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_parity.py"), "8", "15000"], env=env,
                       capture_output=True, text=True, timeout=170)
    assert r.returncode == 0 and "stress ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider",
                        "-k", "benchmark_shapes or garbage or damaged or quirks or zero_timestamps or high_half or overflow "
                              "or known_answer or reference_test_inputs or ragged_batches or uniform_layouts"],
                       env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_skewed_lengths_through_the_host_entry_points():
    """Series of 10 to 100,000 points (constant, counter and random-walk series) through gtsc_encode_host_ex and
    gtsc_decode_host in many chunks, against the oracle, both formats (own process: the chunk size is read once)."""
    env = dict(os.environ, GTSC_CHUNK_POINTS="300000")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "host_chunks_worker.py"), "skewed"],
                       env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0 and "chunks ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def test_host_decode_with_the_workgroup_kernel_on_a_stream_of_its_own():
    """Chunks of several hundred series of 10 to 20,000 points through the host entry points, against the oracle, both
    formats: every chunk has more series than the device has compute units and blocks for all three decompressors, so
    the workgroup-per-block one draws its blocks from the device-built list and runs beside the other two."""
    env = dict(os.environ, GTSC_CHUNK_POINTS="1000000")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "host_chunks_worker.py"), "skewed_many"],
                       env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0 and "chunks ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def test_host_entry_points_leave_nothing_in_flight_after_an_error():
    """A failure half way through the chunked pipeline (injected where the third chunk is staged): the call returns the
    error and none of its copies is still under way, so the caller may free its arrays."""
    env = dict(os.environ, GTSC_CHUNK_POINTS="3000", GTSC_TEST_FAIL_CHUNK="2")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "host_chunks_worker.py"), "fail"],
                       env=env, capture_output=True, text=True, timeout=170)
    assert r.returncode == 0 and "failure ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def test_one_batch_over_two_contexts():
    """gtsc_encode_host_multi / gtsc_decode_host_multi with two contexts on device 0 standing in for two devices
    (this box has one GPU): words, offsets, bit lengths and decoded points equal the single-context call's and the
    oracle's, for a ragged batch of skewed lengths and for a uniform one, both formats."""
    a, b2 = G.Context(0), G.Context(0)
    try:
        sk = synth.skewed(total_points=300000, seed=17, pmin=10, pmax=40000)
        un = synth.gauge_integer(S=5001, P=120, seed=18)
        for gfmt, ofmt in FMTS:
            for b, uniform in ((sk, False), (un, True)):
                kw = dict(uniform_npoints=120) if uniform else dict(pt_off=b.pt_off)
                words, off, nbits = G.encode_host_multi([a, b2], b.block_ts, b.ts, b.val, fmt=gfmt, **kw)
                w1, o1, n1 = a.encode_host(b.block_ts, b.ts, b.val, fmt=gfmt, want_nbits=True, **kw)
                assert np.array_equal(off, o1) and np.array_equal(nbits, n1) and np.array_equal(words, w1)
                ref_out, ref_off, ref_nw, _ = O.encode_batch(b.block_ts, b.ts, b.val, b.pt_off, fmt=ofmt, nthreads=8)
                ref = np.concatenate([ref_out[int(ref_off[s]):int(ref_off[s]) + int(ref_nw[s])] for s in range(b.nseries)])
                assert np.array_equal(words, ref), "compressed words differ from the oracle's"
                kd = dict(uniform_capacity=120) if uniform else dict(pt_off=b.pt_off)
                ts, val, npts, st = G.decode_host_multi([a, b2], words, off, fmt=gfmt, **kd)
                assert not st.any() and np.array_equal(ts, b.ts) and np.array_equal(val, b.val)
                assert np.array_equal(npts.astype(np.int64), np.diff(b.pt_off.astype(np.int64)))
        # a damaged image in the second shard: its status and the call's return code come through
        bad = words.copy()
        bad[int(off[-2]) + 1:] = 0
        ts, val, npts, st = G.decode_host_multi([a, b2], bad, off, uniform_capacity=120, fmt=FMTS[-1][0], check=False)
        rts, rval, rnp, rst = O.decode_batch(bad, off, None, un.pt_off, fmt=FMTS[-1][1])
        assert np.array_equal(st, rst) and np.array_equal(npts, rnp)
    finally:
        a.close()
        b2.close()
