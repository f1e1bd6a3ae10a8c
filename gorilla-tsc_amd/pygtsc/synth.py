"""Synthetic series generators for tests and benchmarks (host side, numpy).

Every generator returns a Batch: block_ts[S], ts[sum P], val[sum P] (raw IEEE-754
bit patterns as uint64, the form Double.doubleToRawLongBits gives the
reference's addValue) and pt_off[S+1].  Seeds follow 0x5EED0000 + config id.
"""
from dataclasses import dataclass

import numpy as np

BLOCK_TS = 1500400800000  # 2-hour aligned, epoch ms


@dataclass
class Batch:
    block_ts: np.ndarray
    ts: np.ndarray
    val: np.ndarray
    pt_off: np.ndarray

    @property
    def nseries(self):
        return len(self.block_ts)

    @property
    def npoints(self):
        return int(self.pt_off[-1])

    def series(self, s):
        a, b = int(self.pt_off[s]), int(self.pt_off[s + 1])
        return self.ts[a:b], self.val[a:b]

    def uniform_npoints(self):
        lens = np.diff(self.pt_off.astype(np.int64))
        return int(lens[0]) if len(lens) and np.all(lens == lens[0]) else None


def _bits(doubles):
    return np.ascontiguousarray(doubles, dtype=np.float64).view(np.uint64)


def _uniform_batch(ts2d, val2d_bits):
    S, P = ts2d.shape
    return Batch(np.full(S, BLOCK_TS, dtype=np.int64), np.ascontiguousarray(ts2d.reshape(-1), dtype=np.int64),
                 np.ascontiguousarray(val2d_bits.reshape(-1), dtype=np.uint64),
                 (np.arange(S + 1, dtype=np.uint64) * np.uint64(P)))


def _cadence(S, P, step_ms):
    return BLOCK_TS + step_ms * (np.arange(1, P + 1, dtype=np.int64))[None, :].repeat(S, 0)


def random_walk(S=1, P=7200, seed=0x5EED0001):
    """cfg1/cfg3: 1 s cadence, Gaussian random walk starting at 100."""
    rng = np.random.default_rng(seed)
    v = 100.0 + np.cumsum(rng.standard_normal((S, P)), axis=1)
    return _uniform_batch(_cadence(S, P, 1000), _bits(v))


def quantised_walk(S=1, P=7200, seed=0x5EED0003):
    """cfg3 variant: steps in {-1/4, 0, +1/4}."""
    rng = np.random.default_rng(seed)
    v = 100.0 + 0.25 * np.cumsum(rng.integers(-1, 2, size=(S, P)), axis=1)
    return _uniform_batch(_cadence(S, P, 1000), _bits(v))


def gauge_integer(S=1000, P=120, seed=0x5EED0002):
    """cfg2(a): 60 s cadence; integer gauge, start U[0,1000], +-1 with p=0.1 per step."""
    rng = np.random.default_rng(seed)
    start = rng.integers(0, 1001, size=(S, 1)).astype(np.float64)
    step = np.where(rng.random((S, P)) < 0.1, rng.integers(0, 2, size=(S, P)) * 2 - 1, 0)
    step[:, 0] = 0
    return _uniform_batch(_cadence(S, P, 60000), _bits(start + np.cumsum(step, axis=1)))


def gauge_drift(S=1000, P=120, seed=0x5EED0012):
    """cfg2(b): 60 s cadence; slow drift rounded to two decimals."""
    rng = np.random.default_rng(seed)
    start = rng.uniform(0, 1000, size=(S, 1))
    v = np.round(start + np.cumsum(rng.normal(0, 0.05, size=(S, P)), axis=1), 2)
    return _uniform_batch(_cadence(S, P, 60000), _bits(v))


def adversarial(S=64, P=7200, seed=0x5EED0004):
    """cfg4: uniformly random 64-bit values (NaN patterns included) and jittered,
    strictly increasing timestamps whose delta-of-deltas hit every bucket:
    0, 9-, 12-, 16-bit codes, 36-bit codes and the dod == -1 special case."""
    rng = np.random.default_rng(seed)
    val = rng.integers(0, 2 ** 64, size=(S, P), dtype=np.uint64)
    cat = rng.choice(6, size=(S, P), p=[0.2, 0.2, 0.2, 0.15, 0.15, 0.1])
    mag = np.select(
        [cat == 0, cat == 1, cat == 2, cat == 3, cat == 4, cat == 5],
        [np.zeros((S, P), dtype=np.int64), rng.integers(2, 65, size=(S, P)), rng.integers(65, 257, size=(S, P)),
         rng.integers(257, 2049, size=(S, P)), rng.integers(2049, 200000, size=(S, P)),
         np.ones((S, P), dtype=np.int64)])
    sign = rng.integers(0, 2, size=(S, P)) * 2 - 1
    sign[cat == 5] = -1  # dod == -1
    dod = mag * sign
    delta = np.empty((S, P), dtype=np.int64)
    cur = np.full(S, 500000, dtype=np.int64)
    for i in range(P):
        nxt = cur + dod[:, i]
        bad = (nxt < 1000) | (nxt > 2000000)
        nxt = np.where(bad, cur - dod[:, i], nxt)  # mirror the step: same bucket, opposite sign
        nxt = np.where((nxt < 1000) | (nxt > 2000000), cur, nxt)
        delta[:, i] = nxt
        cur = nxt
    delta[:, 0] = rng.integers(1, 2 ** 27 - 1, size=S)  # first delta: anywhere in the 27-bit domain
    ts = BLOCK_TS + np.cumsum(delta, axis=1)
    return _uniform_batch(ts, val)


def readme_workload(S=1, P=100000):
    """The reference benchmark's data: ts = block + 60 + 60*(i+1), value = (double) i
    (reference: src/main/java/.../benchmark/EncodingBenchmark.java:49,60-69)."""
    ts = (BLOCK_TS + 60 + 60 * np.arange(1, P + 1, dtype=np.int64))[None, :].repeat(S, 0)
    v = np.arange(P, dtype=np.float64)[None, :].repeat(S, 0)
    return _uniform_batch(ts, _bits(v))


def skewed(total_points=1 << 20, seed=0x5EED0005, pmin=10, pmax=100000):
    """cfg5: series lengths log-uniform in [pmin, pmax]; 30 % constant series, 30 %
    integer-valued counters, 40 % random walks; 1 s cadence."""
    rng = np.random.default_rng(seed)
    lens = []
    tot = 0
    while tot < total_points:
        n = int(np.exp(rng.uniform(np.log(pmin), np.log(pmax))))
        n = max(pmin, min(pmax, n))
        lens.append(n)
        tot += n
    lens = np.array(lens, dtype=np.int64)
    S = len(lens)
    pt_off = np.zeros(S + 1, dtype=np.uint64)
    np.cumsum(lens, out=pt_off[1:])
    ts = np.empty(tot, dtype=np.int64)
    val = np.empty(tot, dtype=np.float64)
    kind = rng.choice(3, size=S, p=[0.3, 0.3, 0.4])
    for s in range(S):
        a, n = int(pt_off[s]), int(lens[s])
        ts[a:a + n] = BLOCK_TS + 1000 * np.arange(1, n + 1, dtype=np.int64)
        if kind[s] == 0:
            val[a:a + n] = rng.uniform(0, 1000)
        elif kind[s] == 1:
            val[a:a + n] = rng.integers(0, 1000) + np.cumsum(rng.integers(0, 5, size=n))
        else:
            val[a:a + n] = 100.0 + np.cumsum(rng.standard_normal(n))
    return Batch(np.full(S, BLOCK_TS, dtype=np.int64), ts, _bits(val), pt_off)


def ragged_mix(seed=7, nseries=200, pmax=300):
    """Small ragged batch with empty and single-point series, for edge-case tests."""
    rng = np.random.default_rng(seed)
    lens = rng.integers(0, pmax, size=nseries)
    lens[:6] = [0, 1, 2, 0, 63, 64]
    pt_off = np.zeros(nseries + 1, dtype=np.uint64)
    np.cumsum(lens, out=pt_off[1:])
    tot = int(pt_off[-1])
    ts = np.empty(tot, dtype=np.int64)
    val = np.empty(tot, dtype=np.uint64)
    block = BLOCK_TS + rng.integers(0, 1000, size=nseries) * 7200000
    for s in range(nseries):
        a, n = int(pt_off[s]), int(lens[s])
        if n == 0:
            continue
        deltas = rng.integers(1, 3000, size=n) if s % 3 else np.full(n, 1000)
        ts[a:a + n] = block[s] + np.cumsum(deltas)
        mode = s % 4
        if mode == 0:
            val[a:a + n] = _bits(100.0 + np.cumsum(rng.standard_normal(n)))
        elif mode == 1:
            val[a:a + n] = rng.integers(0, 2 ** 64, size=n, dtype=np.uint64)
        elif mode == 2:
            val[a:a + n] = rng.integers(0, 2 ** 31, size=n).astype(np.uint64)  # small longs (addValue(long,long))
        else:
            val[a:a + n] = _bits(np.full(n, 42.5))
    return Batch(block.astype(np.int64), ts, val, pt_off)
