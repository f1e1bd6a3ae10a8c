#!/usr/bin/env python3
"""Benchmark of the device-resident Gorilla codec on MI355X.

One "step" = compress one batch of series with the HIP encode kernel, then
decompress the result with the HIP decode kernel, inputs and outputs resident
in HBM.  Default workload: 1,000,000 series x 120 points (2 h at 60 s cadence)
of gauge-like slowly varying doubles per GPU.  With --gpus N > 1 (launched by
torch.distributed.run) every rank codes its own shard of series of the same
size on its own GPU: series are independent, so there is no collective on the
data path; the only communication is the barrier / MAX of the timing.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "gorilla-tsc_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import pygtsc as G  # noqa: E402
from pygtsc import shard, synth  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak


def make_workload(name, S, P, device, seed):
    """Synthetic series generated on the device.  Returns (block_ts, ts, val) int64 tensors."""
    g = torch.Generator(device=device).manual_seed(seed)
    if name == "gauge":
        # integer-valued gauge: start U[0,1000], +-1 with probability 0.1 per step, 60 s cadence
        cadence = 60000
        start = torch.randint(0, 1001, (S, 1), device=device, generator=g).double()
        r = torch.rand((S, P), device=device, generator=g)
        step = torch.where(r < 0.05, -1.0, torch.where(r < 0.10, 1.0, 0.0)).double()
        step[:, 0] = 0
        v = start + torch.cumsum(step, dim=1)
        del r, step
    elif name.startswith("gauge_p"):
        # the same gauge with another share of changing values: gauge_p30 = 30 % of the steps move it
        cadence = 60000
        pr = int(name[7:]) / 100.0
        start = torch.randint(0, 1001, (S, 1), device=device, generator=g).double()
        r = torch.rand((S, P), device=device, generator=g)
        step = torch.where(r < pr / 2, -1.0, torch.where(r < pr, 1.0, 0.0)).double()
        step[:, 0] = 0
        v = start + torch.cumsum(step, dim=1)
        del r, step
    elif name.startswith("gauge_jit"):
        # the gauge with one timestamp in N a millisecond late: gauge_jit1000
        cadence = 60000
        start = torch.randint(0, 1001, (S, 1), device=device, generator=g).double()
        r = torch.rand((S, P), device=device, generator=g)
        step = torch.where(r < 0.05, -1.0, torch.where(r < 0.10, 1.0, 0.0)).double()
        step[:, 0] = 0
        v = start + torch.cumsum(step, dim=1)
        del r, step
    elif name == "gauge_drift":
        # slow drift rounded to 2 decimals (about 33 bits/point)
        cadence = 60000
        start = torch.rand((S, 1), device=device, generator=g, dtype=torch.float64) * 1000
        v = start + torch.cumsum(torch.randn((S, P), device=device, generator=g, dtype=torch.float64) * 0.05, dim=1)
        v = torch.round(v * 100) / 100
    elif name == "counter":
        # integer-valued counter that grows by 0..4 per step (one step in five leaves it unchanged)
        cadence = 1000
        start = torch.randint(0, 1000, (S, 1), device=device, generator=g).double()
        v = start + torch.cumsum(torch.randint(0, 5, (S, P), device=device, generator=g).double(), dim=1)
    elif name == "random_walk":
        cadence = 1000
        v = 100.0 + torch.cumsum(torch.randn((S, P), device=device, generator=g, dtype=torch.float64), dim=1)
    elif name == "readme":
        # the reference benchmark's own data: value = (double) i (G/benchmark/EncodingBenchmark.java:60-69)
        cadence = 60
        v = torch.arange(P, device=device, dtype=torch.float64).repeat(S, 1)
    elif name == "adversarial":
        cadence = 1000
        v = torch.randint(-2 ** 63, 2 ** 63 - 1, (S, P), device=device, generator=g, dtype=torch.int64).view(torch.float64)
    else:
        raise SystemExit("unknown workload %r" % name)
    val = v.contiguous().view(torch.int64).reshape(-1)
    ts_row = synth.BLOCK_TS + cadence * torch.arange(1, P + 1, device=device, dtype=torch.int64)
    if name == "adversarial":
        jit = torch.randint(-400, 400, (S, P), device=device, generator=g, dtype=torch.int64)
        ts = (ts_row[None, :] + jit).reshape(-1).contiguous()
    elif name.startswith("gauge_jit"):
        late = (torch.rand((S, P), device=device, generator=g) < 1.0 / int(name[9:])).to(torch.int64)
        ts = (ts_row[None, :] + late).reshape(-1).contiguous()
    else:
        ts = ts_row.repeat(S, 1).reshape(-1).contiguous()
    block = torch.full((S,), synth.BLOCK_TS, dtype=torch.int64, device=device)
    return block, ts, val


def cpu_baseline(block, ts, val, P, sample_series, threads):
    """The C oracle (a port of the reference algorithm), one codec instance per series,
    series split over `threads` POSIX threads, on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as O
    S = min(sample_series, block.numel())
    b = block[:S].cpu().numpy()
    t = ts[: S * P].cpu().numpy()
    v = val[: S * P].cpu().numpy().view(np.uint64)
    pt_off = np.arange(S + 1, dtype=np.uint64) * np.uint64(P)
    best_e = best_d = 1e30
    for _ in range(2):
        t0 = time.perf_counter()
        out, off, nw, st = O.encode_batch(b, t, v, pt_off, nthreads=threads)
        t1 = time.perf_counter()
        dt, dv, npts, st2 = O.decode_batch(out, off, nw, pt_off, nthreads=threads)
        t2 = time.perf_counter()
        best_e, best_d = min(best_e, t1 - t0), min(best_d, t2 - t1)
    assert np.array_equal(dt, t) and np.array_equal(dv, v)
    raw = 16.0 * S * P
    # one thread on a sixteenth of the sample, for comparison with single-thread figures of other codecs
    S1 = max(S // 16, 1)
    t0 = time.perf_counter()
    o1, f1, n1, _ = O.encode_batch(b[:S1], t[: S1 * P], v[: S1 * P], pt_off[: S1 + 1], nthreads=1)
    t1 = time.perf_counter()
    O.decode_batch(o1, f1, n1, pt_off[: S1 + 1], nthreads=1)
    t2 = time.perf_counter()
    return {
        "value": round(2 * raw / (best_e + best_d) / 1e9, 3),
        "unit": "GB/s",
        "cores": threads,
        "kind": "port",
        "sample": "%d series x %d points (%.0f MB raw), compress+decompress, best of 2" % (S, P, raw / 1e6),
        "compress_GBps": round(raw / best_e / 1e9, 3),
        "decompress_GBps": round(raw / best_d / 1e9, 3),
        "one_thread_compress_Mpoints_per_s": round(S1 * P / (t1 - t0) / 1e6, 1),
        "one_thread_decompress_Mpoints_per_s": round(S1 * P / (t2 - t1) / 1e6, 1),
    }


def pmc_traffic(workload, S, P, kernel):
    """(HBM bytes per launch of `kernel`, where the figure comes from): the rocprofv3 PMC passes (FETCH_SIZE
    with the gfx950 correction + WRITE_SIZE) that tools/profile.sh collected for this same workload, or
    (None, None).  Counters cannot be read from inside the timed process, so the figure is carried in a file."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        t = json.load(open(path))
    except (OSError, ValueError):
        return None, None
    if (t.get("workload"), t.get("series"), t.get("points")) != (workload, S, P):
        return None, None
    return t.get(kernel + "_bytes"), "profiles/traffic.json (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE passes of %s)" % t.get("taken", "an earlier run")


EXTRA_WORKLOADS = [
    # (key, workload, series, points): the dense and long-block configurations, reported beside the headline
    ("gauge_drift_1Mx120", "gauge_drift", 1_000_000, 120),
    ("random_walk_262144x7200", "random_walk", 262_144, 7200),
    ("adversarial_65536x7200", "adversarial", 65_536, 7200),
    ("gauge_16668x7200", "gauge", 16_668, 7200),
    ("gauge_262144x7200", "gauge", 262_144, 7200),
]


def other_workloads(measure, ctx):
    """The configurations the headline does not cover, each compress + decompress device-resident like the
    headline (5 timed steps after 3), plus batches of one or a few long blocks and the host-to-host rates."""
    out = {}
    for key, wl, S, P in EXTRA_WORKLOADS:
        try:
            m = measure(wl, S, P, 5, 3, 0x5EED0100)
        except Exception as e:  # an extra must not take the headline down with it
            out[key] = {"error": repr(e)[:200]}
            torch.cuda.empty_cache()
            continue
        slow = max(m["enc_ms"], m["dec_ms"])
        out[key] = {
            "compress_GBps": round(m["raw_bytes"] / m["enc_ms"] / 1e6, 1), "decompress_GBps": round(m["raw_bytes"] / m["dec_ms"] / 1e6, 1),
            "raw_only_frac_of_hbm_peak": {"compress": round(m["raw_bytes"] / m["enc_ms"] / 1e6 / HBM_PEAK_GBPS, 4),
                                          "decompress": round(m["raw_bytes"] / m["dec_ms"] / 1e6 / HBM_PEAK_GBPS, 4)},
            "roofline_frac": round(m["algo_bytes"] / slow / 1e6 / HBM_PEAK_GBPS, 4),
            "encode_ms": round(m["enc_ms"], 4), "decode_ms": round(m["dec_ms"], 4),
            "encode_ms_spread": m["enc_spread"], "decode_ms_spread": m["dec_spread"],
            "compressed_bits_per_point": round(64.0 * m["total_words"] / (S * P), 3),
            "tunables_set_by_this_caller": m["tunables"], "kernels_dispatched": m["kernels"],
        }
        torch.cuda.empty_cache()
    # batches too small to fill the GPU: the reference's own block, and the reference benchmark's data
    # (G/benchmark/EncodingBenchmark.java:49,60-69: 100,000 points of (double) i) as one block and as 1,024
    for key, make in (("one_block_7200_random_walk", lambda: synth.random_walk(S=1, P=7200)),
                      ("readme_1x100000", lambda: synth.readme_workload(S=1, P=100000)),
                      ("readme_1024x100000", lambda: synth.readme_workload(S=1024, P=100000))):
        try:
            out[key] = few_blocks(ctx, make())
        except Exception as e:
            out[key] = {"error": repr(e)[:200]}
        torch.cuda.empty_cache()
    return out


def few_blocks(ctx, b, cpu_reps=5):
    """A batch too small to fill the GPU -- one 2-hour block of 7,200 random-walk doubles (the reference's own
    use), or the reference benchmark's 100,000-point blocks.  GPU: wall time of a device-resident compress and
    decompress call incl. launch and sync; CPU: the C oracle, one thread."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as O
    S, P = b.block_ts.size, int(b.pt_off[1] - b.pt_off[0])
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()  # noqa: E731
    d_block, d_ts, d_val = d(b.block_ts), d(b.ts), d(b.val)
    slot = G.max_words(P)
    d_out = torch.zeros(S * slot, dtype=torch.int64, device="cuda")
    d_nw = torch.zeros(S, dtype=torch.int64, device="cuda")
    d_ts2, d_val2 = torch.zeros_like(d_ts), torch.zeros_like(d_val)
    d_np = torch.zeros(S, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    te = td = 1e9
    for _ in range(20):
        t0 = time.perf_counter()
        ctx.encode_device(d_block, d_ts, d_val, d_out, d_nw, uniform_npoints=P, uniform_slot_words=slot)
        ctx.sync()
        t1 = time.perf_counter()
        ctx.decode_device(d_out, d_ts2, d_val2, d_np, S, uniform_slot_words=slot, in_nwords=d_nw, uniform_capacity=P)
        ctx.sync()
        t2 = time.perf_counter()
        te, td = min(te, t1 - t0), min(td, t2 - t1)
    assert torch.equal(d_ts2, d_ts) and torch.equal(d_val2, d_val)
    # the CPU side on one block of the batch (they are all alike in these batches)
    one = synth.Batch(b.block_ts[:1], b.ts[:P], b.val[:P], b.pt_off[:2])
    ce = cd = 1e9
    for _ in range(cpu_reps):
        t0 = time.perf_counter()
        w, off, nw, _ = O.encode_batch(one.block_ts, one.ts, one.val, one.pt_off, nthreads=1)
        t1 = time.perf_counter()
        O.decode_batch(w, off, nw, one.pt_off, nthreads=1)
        t2 = time.perf_counter()
        ce, cd = min(ce, t1 - t0), min(cd, t2 - t1)
    assert np.array_equal(d_out.cpu().numpy().view(np.uint64)[: int(nw[0])], w[: int(nw[0])])
    return {"blocks": S, "points_per_block": P,
            "gpu_compress_us": round(te * 1e6, 1), "gpu_decompress_us": round(td * 1e6, 1),
            "cpu_port_one_thread_compress_us_per_block": round(ce * 1e6, 1),
            "cpu_port_one_thread_decompress_us_per_block": round(cd * 1e6, 1),
            "note": "GPU figures are whole calls over all blocks (launch + sync), best of 20; CPU one block, best of %d" % cpu_reps}


def host_to_host():
    """gtsc_encode_host / gtsc_decode_host on pinned host buffers, PCIe copies included (never the headline value):
    2^26 points of uniform gauges and of skewed lengths (10 to 100,000 points, constant / counter / random-walk
    series).  Run by tools/host_bench.py in a child process of its own: the pipeline's three streams plus the
    context's are what the runtime's four hardware queues carry side by side, and this process holds torch's
    streams as well."""
    import subprocess
    res = {}
    for kind in ("gauge", "skewed"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "host_bench.py"), kind, "26"], capture_output=True, text=True,
                           timeout=300)
        line = [x for x in r.stdout.splitlines() if x.startswith("{")]
        if r.returncode != 0 or not line:
            res[kind] = {"error": (r.stderr or r.stdout)[-200:]}
            continue
        j = json.loads(line[-1])
        res[kind] = {"series": j["series"], "points": j["points"], "compress_GBps": j["encode_host_GBps"],
                     "decompress_GBps": j["decode_host_GBps"], "compressed_MB": j["compressed_MB"]}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10,
                    help="untimed steps first (the GPU takes about eight of these steps to reach its steady clocks)")
    ap.add_argument("--workload", default="gauge")
    ap.add_argument("--series", type=int, default=1_000_000)
    ap.add_argument("--points", type=int, default=120)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-series", type=int, default=1_000_000)
    ap.add_argument("--slot-words", type=int, default=0, help="words per output slot (default: the worst case for --points)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL; gloo to rehearse "
                    "the multi-rank path on a box with fewer GPUs than ranks)")
    ap.add_argument("--no-extras", action="store_true", help="skip the other workloads (dense streams, long blocks, one block, "
                    "host to host) that the default single-GPU run reports under other_workloads")
    ap.add_argument("--no-verify", action="store_true", help="skip the round-trip check (kernel experiments only)")
    ap.add_argument("--val-phase", type=int, default=0, help="start the value array this many 8-byte words into its allocation "
                    "(diagnostic: an odd number puts it on the other 16-byte phase than the timestamps)")
    ap.add_argument("--out-phase", type=int, default=0, help="start the decoded value array this many 8-byte words into its allocation "
                    "(diagnostic: where the two output streams fall relative to each other)")
    ap.add_argument("--format", default="v2", choices=["v2", "v1"], help="v2 = GorillaCompressor (the default), v1 = Compressor")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    default_variant = args.format == "v2" and not args.val_phase and not args.slot_words and not args.out_phase
    extras = (world == 1 and not args.no_extras and default_variant
              and (args.workload, args.series, args.points) == ("gauge", 1_000_000, 120))
    # The host-to-host figures come from child processes that run before this process touches the GPU: two
    # processes with GPU queues of their own get in each other's way (the copy-out rate halves).
    h2h = None
    if extras:
        try:
            h2h = host_to_host()
        except Exception as e:
            h2h = {"error": repr(e)[:200]}
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d needs torch.distributed.run with %d ranks (WORLD_SIZE=%d)" % (args.gpus, args.gpus, world))
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    ndev = torch.cuda.device_count()
    if local_rank >= ndev and args.backend == "nccl":
        raise SystemExit("rank %d has no GPU (%d visible)" % (local_rank, ndev))
    dev_index = local_rank % ndev  # more ranks than GPUs only in a gloo rehearsal
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(args.backend)
    red_device = device if args.backend == "nccl" else None  # where the timing reduction's tensor lives

    S, P = args.series, args.points
    ctx = G.Context(dev_index)
    stream = torch.cuda.Stream(device=device)
    ctx.set_stream(stream.cuda_stream)
    fmt = G.FMT_V2 if args.format == "v2" else G.FMT_V1

    def measure(workload, S, P, steps, warmup, seed, timed_by_host=False, keep_inputs=False):
        """Compress + decompress `steps` times after `warmup` untimed steps; HIP-event time per kernel
        (events on the stream the library launches on).  Returns a dict of what was measured."""
        block, ts, val = make_workload(workload, S, P, device, seed)
        if args.val_phase:
            big = torch.empty(val.numel() + args.val_phase, dtype=val.dtype, device=device)
            big[args.val_phase:].copy_(val)
            val = big[args.val_phase:]
        slot = args.slot_words or G.max_words(P)
        out = torch.zeros(S * slot, dtype=torch.int64, device=device)
        nw = torch.zeros(S, dtype=torch.int64, device=device)
        ts2 = torch.zeros_like(ts)
        if args.out_phase:
            big2 = torch.zeros(val.numel() + args.out_phase, dtype=val.dtype, device=device)
            val2 = big2[args.out_phase:]
        else:
            val2 = torch.zeros_like(val)
        npts = torch.zeros(S, dtype=torch.int64, device=device)

        ctx_dispatch = {}

        def encode():
            ctx.encode_device(block, ts, val, out, nw, uniform_npoints=P, uniform_slot_words=slot, fmt=fmt)
            ctx_dispatch["encode"] = ctx.last_dispatch()

        def decode():
            ctx.decode_device(out, ts2, val2, npts, S, uniform_slot_words=slot, in_nwords=nw, uniform_capacity=P, fmt=fmt)
            ctx_dispatch["decode"] = ctx.last_dispatch()

        # The library decides from the batch alone; what a caller knows about its data it says through tunables.  This
        # caller compresses the batch once, looks at the density, and says: dense streams (the lane-per-series
        # decompressor's 16-word rings), or sparse ones (the lane-per-series compressor for many long blocks).
        torch.cuda.synchronize()
        with torch.cuda.stream(stream):
            encode()
        try:
            ctx.sync()
        except G.GtscError:
            if not args.no_verify:
                raise
        bits_per_point = 64.0 * int(nw.sum()) / max(S * P, 1)
        tun = {"decode_ring": 16 if bits_per_point > 40 else 8, "sparse_hint": 1 if bits_per_point <= 16 else 0}
        if os.environ.get("GTSC_BENCH_RING"):   # (measurements: what a caller who leaves the ring to the device gets)
            tun["decode_ring"] = int(os.environ["GTSC_BENCH_RING"])
        ctx.set_decode_ring(tun["decode_ring"])
        ctx.set_tunable(G.TUNE_SPARSE_HINT, tun["sparse_hint"])
        with torch.cuda.stream(stream):
            for _ in range(warmup):
                encode()
                decode()
        try:
            ctx.sync()
        except G.GtscError:
            if not args.no_verify:
                raise
        # parity guard on the very buffers that get timed
        if not args.no_verify:
            assert torch.equal(ts2, ts) and torch.equal(val2, val) and bool((npts == P).all()), "round trip failed"
        total_words = int(nw.sum())
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(steps)]
        if timed_by_host and dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(stream):
            for k in range(steps):
                ev[k][0].record(stream)
                encode()
                ev[k][1].record(stream)
                decode()
                ev[k][2].record(stream)
        try:
            ctx.sync()
        except G.GtscError:
            if not args.no_verify:
                raise
        torch.cuda.synchronize()
        if timed_by_host and dist:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        enc_all = [e[0].elapsed_time(e[1]) for e in ev]
        dec_all = [e[1].elapsed_time(e[2]) for e in ev]
        enc_ms, dec_ms = float(np.mean(enc_all)), float(np.mean(dec_all))
        spread = lambda x: {"min": round(float(np.min(x)), 4), "median": round(float(np.median(x)), 4), "max": round(float(np.max(x)), 4)}  # noqa: E731
        raw = 16.0 * S * P
        algo = raw + 8.0 * total_words + 16.0 * S  # raw points + compressed words + per-series block_ts and length
        r = {"elapsed": elapsed, "enc_ms": enc_ms, "dec_ms": dec_ms, "raw_bytes": raw, "algo_bytes": algo, "total_words": total_words,
             "enc_spread": spread(enc_all), "dec_spread": spread(dec_all), "tunables": tun,
             "kernels": {"encode": ctx_dispatch["encode"], "decode": ctx_dispatch["decode"]}}
        if keep_inputs:
            r["inputs"] = (block, ts, val)
        return r

    m = measure(args.workload, S, P, args.steps, args.warmup, 0x5EED0002 + rank, timed_by_host=True, keep_inputs=True)
    block, ts, val = m.pop("inputs")
    elapsed = shard.max_over_ranks(m["elapsed"], dist, red_device)
    enc_ms, dec_ms, raw_bytes, algo_bytes, total_words = m["enc_ms"], m["dec_ms"], m["raw_bytes"], m["algo_bytes"], m["total_words"]
    dom, dom_ms = ("encode", enc_ms) if enc_ms >= dec_ms else ("decode", dec_ms)
    achieved = algo_bytes / (dom_ms * 1e-3) / 1e9
    traffic, traffic_source = pmc_traffic(args.workload, S, P, dom) if default_variant else (None, None)

    result = {
        "metric": "device-resident Gorilla compress/decompress GB/s of raw (ts,double) points",
        "value": round(2 * raw_bytes * args.steps * world / elapsed / 1e9, 2),
        "unit": "GB/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {
            "workload": "%s: %d series x %d points per GPU, %s, uniform layout; step = compress + decompress" % (
                args.workload, S, P, args.format + " format"),
            "series_per_gpu": S, "points_per_series": P,
            "compressed_bits_per_point": round(64.0 * total_words / (S * P), 3),
        },
        "compress_GBps_per_gpu": round(raw_bytes / (enc_ms * 1e-3) / 1e9, 2),
        "decompress_GBps_per_gpu": round(raw_bytes / (dec_ms * 1e-3) / 1e9, 2),
        "raw_only_frac_of_hbm_peak": {
            "compress": round(raw_bytes / (enc_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
            "decompress": round(raw_bytes / (dec_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
        },
        "roofline": {
            "bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": traffic_source,
            "algorithmic_bytes_per_launch": algo_bytes, "avg_launch_ms": round(dom_ms, 4),
            "encode_ms": round(enc_ms, 4), "decode_ms": round(dec_ms, 4),
            # per-launch spread of the HIP-event times over the timed steps (avg_launch_ms is their mean)
            "encode_ms_spread": m["enc_spread"], "decode_ms_spread": m["dec_spread"],
            "tunables_set_by_this_caller": m["tunables"], "kernels_dispatched": m["kernels"],
        },
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        threads = min(16, os.cpu_count() or 1)
        result["cpu_baseline"] = cpu_baseline(block, ts, val, P, args.cpu_sample_series, threads)
    if extras:
        del block, ts, val
        torch.cuda.empty_cache()
        result["other_workloads"] = other_workloads(measure, ctx)
        result["other_workloads"]["host_to_host"] = h2h
    if rank == 0:
        print(json.dumps(result), flush=True)
    ctx.close()
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
