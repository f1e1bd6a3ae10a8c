"""ctypes binding of libgtsc.so (the C ABI in include/gtsc.h).

Thin by design: it loads the shared library, declares the prototypes, and
offers a Context whose methods take either torch CUDA tensors (device entry
points) or numpy arrays (host entry points).  There is no fallback of any
kind: if the library or a GPU is missing, calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GTSC_LIB") or os.path.join(os.path.dirname(_HERE), "csrc", "libgtsc.so")

OK, ERR_ARG, ERR_CAPACITY, ERR_HIP, ERR_STREAM = 0, -1, -2, -3, -4
FMT_V2, FMT_V1 = 0, 1
ST_OVERFLOW, ST_TRUNCATED, ST_MALFORMED = 1, 2, 4

# every symbol include/gtsc.h declares
SYMBOLS = (
    "gtsc_version", "gtsc_last_error", "gtsc_max_words", "gtsc_create", "gtsc_destroy", "gtsc_set_stream",
    "gtsc_sync", "gtsc_encode_device", "gtsc_decode_device", "gtsc_compact_device", "gtsc_encode_host",
    "gtsc_decode_host", "gtsc_host_alloc", "gtsc_host_free", "gtsc_encode_device_ex", "gtsc_words_to_bytes_device",
    "gtsc_bytes_to_words_device", "gtsc_set_predictor", "gtsc_encode_host_ex", "gtsc_compact_device_noalloc", "gtsc_set_decode_ring",
    "gtsc_set_tunable", "gtsc_get_tunable", "gtsc_last_dispatch", "gtsc_host_pipeline_idle", "gtsc_encode_host_multi",
    "gtsc_decode_host_multi",
)
# gtsc_set_tunable ids and gtsc_last_dispatch bits (include/gtsc.h)
(TUNE_LONG_POINTS, TUNE_VLONG_POINTS, TUNE_WAVE_DECODE_MAX, TUNE_WAVE_DECODE_MAX_DENSE, TUNE_WG_DECODE_MAX, TUNE_WG_DECODE_VLONG_MAX,
 TUNE_WG_DENSE_BITS, TUNE_DENSE_BITS, TUNE_SPARSE_ENCODE_MIN, TUNE_COOP_ENCODE_MAX, TUNE_SPARSE_HINT, TUNE_DECODE_RING, TUNE_RUN_SKIP,
 TUNE_DIRECT_KERNELS) = range(1, 15)
(DISPATCH_LANE_DIRECT, DISPATCH_LANE_TILED, DISPATCH_LANE_RING16, DISPATCH_LANE_BOTH_RINGS, DISPATCH_WAVE, DISPATCH_WG, DISPATCH_PLAN,
 DISPATCH_COOP, DISPATCH_SPLIT) = (1 << i for i in range(9))
NOALLOC_MAX_SERIES = 8_000_000
PRED_LAST, PRED_DFCM = 0, 1
ENC_NO_CLOSE = 1


class GtscError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("gtsc error %d: %s" % (code, msg))
        self.code = code


_lib = None


def lib():
    """Load libgtsc.so (built in-tree by `make -C gorilla-tsc_amd/csrc`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libgtsc.so not built: run `make -C %s`" % os.path.dirname(LIB_PATH))
    # One HIP runtime per process: the torch wheel carries its own libamdhip64
    # (requested as "libamdhip64.so", soname libamdhip64.so.7).  Loaded first, it
    # also satisfies libgtsc's "libamdhip64.so.7"; loaded second, the process
    # would hold two runtimes and the later one finds no device.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, u64, i32 = C.c_void_p, C.c_uint64, C.c_int
    L.gtsc_version.restype = i32
    L.gtsc_last_error.restype = C.c_char_p
    L.gtsc_max_words.restype = u64
    L.gtsc_max_words.argtypes = [u64]
    L.gtsc_create.restype = i32
    L.gtsc_create.argtypes = [i32, C.POINTER(vp)]
    L.gtsc_destroy.restype = i32
    L.gtsc_destroy.argtypes = [vp]
    L.gtsc_set_stream.restype = i32
    L.gtsc_set_stream.argtypes = [vp, vp]
    L.gtsc_set_predictor.restype = i32
    L.gtsc_set_predictor.argtypes = [vp, i32, i32]
    L.gtsc_set_decode_ring.restype = i32
    L.gtsc_set_decode_ring.argtypes = [vp, i32]
    L.gtsc_set_tunable.restype = i32
    L.gtsc_set_tunable.argtypes = [vp, i32, u64]
    L.gtsc_get_tunable.restype = i32
    L.gtsc_get_tunable.argtypes = [vp, i32, C.POINTER(u64)]
    L.gtsc_last_dispatch.restype = i32
    L.gtsc_last_dispatch.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(u64)]
    L.gtsc_host_pipeline_idle.restype = i32
    L.gtsc_host_pipeline_idle.argtypes = [vp]
    L.gtsc_sync.restype = i32
    L.gtsc_sync.argtypes = [vp]
    L.gtsc_encode_device.restype = i32
    L.gtsc_encode_device.argtypes = [vp, i32, u64, vp, vp, vp, vp, u64, vp, vp, u64, vp, vp]
    L.gtsc_encode_device_ex.restype = i32
    L.gtsc_encode_device_ex.argtypes = [vp, i32, C.c_uint32, u64, vp, vp, vp, vp, u64, vp, vp, u64, vp, vp, vp]
    L.gtsc_words_to_bytes_device.restype = i32
    L.gtsc_words_to_bytes_device.argtypes = [vp, u64, vp, vp, u64, vp, vp, vp, u64, vp]
    L.gtsc_bytes_to_words_device.restype = i32
    L.gtsc_bytes_to_words_device.argtypes = [vp, u64, vp, vp, u64, vp, vp, vp, u64, vp]
    L.gtsc_decode_device.restype = i32
    L.gtsc_decode_device.argtypes = [vp, i32, u64, vp, vp, u64, vp, vp, vp, vp, u64, vp, vp]
    L.gtsc_compact_device.restype = i32
    L.gtsc_compact_device.argtypes = [vp, u64, vp, vp, u64, vp, vp, vp]
    L.gtsc_encode_host.restype = i32
    L.gtsc_encode_host.argtypes = [vp, i32, u64, vp, vp, vp, vp, u64, vp, u64, vp]
    L.gtsc_compact_device_noalloc.restype = i32
    L.gtsc_compact_device_noalloc.argtypes = [vp, u64, vp, vp, u64, vp, vp, vp]
    L.gtsc_encode_host_ex.restype = i32
    L.gtsc_encode_host_ex.argtypes = [vp, i32, C.c_uint32, u64, vp, vp, vp, vp, u64, vp, u64, vp, vp]
    L.gtsc_decode_host.restype = i32
    L.gtsc_decode_host.argtypes = [vp, i32, u64, vp, vp, vp, vp, vp, u64, vp, vp]
    L.gtsc_encode_host_multi.restype = i32
    L.gtsc_encode_host_multi.argtypes = [i32, C.POINTER(vp), i32, C.c_uint32, u64, vp, vp, vp, vp, u64, vp, u64, vp, vp]
    L.gtsc_decode_host_multi.restype = i32
    L.gtsc_decode_host_multi.argtypes = [i32, C.POINTER(vp), i32, u64, vp, vp, vp, vp, vp, u64, vp, vp]
    L.gtsc_host_alloc.restype = vp
    L.gtsc_host_alloc.argtypes = [C.c_size_t]
    L.gtsc_host_free.argtypes = [vp]
    _lib = L
    return L


def max_words(npoints):
    return int(lib().gtsc_max_words(int(npoints)))


def _check(rc):
    if rc != OK:
        raise GtscError(rc, lib().gtsc_last_error().decode())


def _dp(t):
    """device pointer of a torch tensor (or None)"""
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous()
    return C.c_void_p(t.data_ptr())


def _hp(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def encode_host_multi(ctxs, block_ts, ts, val, pt_off=None, uniform_npoints=0, fmt=FMT_V2, options=0):
    """gtsc_encode_host_multi over the given Contexts: returns (words, off, nbits)."""
    block_ts = np.ascontiguousarray(block_ts, dtype=np.int64)
    ts = np.ascontiguousarray(ts, dtype=np.int64)
    val = np.ascontiguousarray(val, dtype=np.uint64)
    S = len(block_ts)
    if pt_off is not None:
        pt_off = np.ascontiguousarray(pt_off, dtype=np.uint64)
        lens = np.diff(pt_off.astype(np.int64))
    else:
        lens = np.full(S, int(uniform_npoints), dtype=np.int64)
    bits = 64 + np.where(lens > 0, 91 + 114 * (lens - 1), 0) + 37
    cap = int((bits // 64 + 2).sum())
    out = np.zeros(max(cap, 1), dtype=np.uint64)
    off = np.zeros(S + 1, dtype=np.uint64)
    nbits = np.zeros(S, dtype=np.uint64)
    handles = (C.c_void_p * len(ctxs))(*[c._h for c in ctxs])
    _check(lib().gtsc_encode_host_multi(len(ctxs), handles, fmt, int(options), S, _hp(block_ts), _hp(ts), _hp(val), _hp(pt_off),
                                        int(uniform_npoints), _hp(out), cap, _hp(off), _hp(nbits)))
    return out[: int(off[-1])], off, nbits


def decode_host_multi(ctxs, words, in_off, pt_off=None, uniform_capacity=0, fmt=FMT_V2, check=True):
    """gtsc_decode_host_multi over the given Contexts: returns (ts, val, npoints, status)."""
    words = np.ascontiguousarray(words, dtype=np.uint64)
    in_off = np.ascontiguousarray(in_off, dtype=np.uint64)
    S = len(in_off) - 1
    if pt_off is not None:
        pt_off = np.ascontiguousarray(pt_off, dtype=np.uint64)
        total = int(pt_off[-1])
    else:
        total = S * int(uniform_capacity)
    ts = np.zeros(total, dtype=np.int64)
    val = np.zeros(total, dtype=np.uint64)
    npts = np.zeros(S, dtype=np.uint64)
    st = np.zeros(S, dtype=np.uint32)
    handles = (C.c_void_p * len(ctxs))(*[c._h for c in ctxs])
    rc = lib().gtsc_decode_host_multi(len(ctxs), handles, fmt, S, _hp(words), _hp(in_off), _hp(ts), _hp(val), _hp(pt_off),
                                      int(uniform_capacity), _hp(npts), _hp(st))
    if check:
        _check(rc)
    return ts, val, npts, st


class Context:
    """gtsc_ctx owner.  Device methods enqueue on the context's stream; call sync()."""

    def __init__(self, device=-1):
        h = C.c_void_p()
        _check(lib().gtsc_create(int(device), C.byref(h)))
        self._h = h
        # Device calls return before their kernels have run, and torch's caching allocator reuses a
        # tensor's memory as soon as the tensor dies: every tensor handed to a device call is held
        # here until the next sync().
        self._held = {}

    def _hold(self, *tensors):
        for t in tensors:
            if t is not None:
                self._held[id(t)] = t
        if len(self._held) > 65536:  # a caller that never calls sync(): forget the oldest half
            for k in list(self._held)[: len(self._held) // 2]:
                del self._held[k]

    def close(self):
        if getattr(self, "_h", None):
            lib().gtsc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except TypeError:  # interpreter shutdown: the module's globals are gone already
            pass

    def set_stream(self, hip_stream):
        _check(lib().gtsc_set_stream(self._h, C.c_void_p(hip_stream) if hip_stream else None))

    def set_predictor(self, predictor, table_size=0):
        _check(lib().gtsc_set_predictor(self._h, int(predictor), int(table_size)))

    def set_decode_ring(self, ring_words):
        _check(lib().gtsc_set_decode_ring(self._h, int(ring_words)))

    def set_tunable(self, which, value):
        _check(lib().gtsc_set_tunable(self._h, int(which), int(value)))

    def get_tunable(self, which):
        v = C.c_uint64()
        _check(lib().gtsc_get_tunable(self._h, int(which), C.byref(v)))
        return int(v.value)

    def last_dispatch(self, with_plan=False):
        """Bits of the kernels the last device call enqueued (DISPATCH_*); with_plan: also what the device
        made of the last decoded batch, as a list of six counts (waits for the stream)."""
        k = C.c_uint32()
        plan = (C.c_uint64 * 6)() if with_plan else None
        _check(lib().gtsc_last_dispatch(self._h, C.byref(k), plan))
        return (int(k.value), [int(x) for x in plan]) if with_plan else int(k.value)

    def host_pipeline_idle(self):
        rc = lib().gtsc_host_pipeline_idle(self._h)
        if rc < 0:
            _check(rc)
        return bool(rc)

    def sync(self):
        try:
            _check(lib().gtsc_sync(self._h))
        finally:
            self._held.clear()

    # ---- device-resident -------------------------------------------------
    def encode_device(self, block_ts, ts, val, out_words, out_nwords, pt_off=None, uniform_npoints=0,
                      out_off=None, uniform_slot_words=0, status=None, fmt=FMT_V2, options=0, out_nbits=None):
        S = block_ts.numel()
        self._hold(block_ts, ts, val, out_words, out_nwords, pt_off, out_off, status, out_nbits)
        if options or out_nbits is not None:
            _check(lib().gtsc_encode_device_ex(self._h, fmt, int(options), S, _dp(block_ts), _dp(ts), _dp(val), _dp(pt_off),
                                               int(uniform_npoints), _dp(out_words), _dp(out_off), int(uniform_slot_words),
                                               _dp(out_nwords), _dp(out_nbits), _dp(status)))
            return
        _check(lib().gtsc_encode_device(self._h, fmt, S, _dp(block_ts), _dp(ts), _dp(val), _dp(pt_off),
                                        int(uniform_npoints), _dp(out_words), _dp(out_off), int(uniform_slot_words),
                                        _dp(out_nwords), _dp(status)))

    def words_to_bytes_device(self, words, nbits, out_bytes, out_nbytes, word_off=None, uniform_slot_words=0,
                              byte_off=None, uniform_slot_bytes=0):
        self._hold(words, nbits, out_bytes, out_nbytes, word_off, byte_off)
        _check(lib().gtsc_words_to_bytes_device(self._h, nbits.numel(), _dp(words), _dp(word_off), int(uniform_slot_words),
                                                _dp(nbits), _dp(out_bytes), _dp(byte_off), int(uniform_slot_bytes),
                                                _dp(out_nbytes)))

    def bytes_to_words_device(self, data, out_words, out_nwords, nbytes=None, byte_off=None, uniform_slot_bytes=0,
                              word_off=None, uniform_slot_words=0):
        self._hold(data, out_words, out_nwords, nbytes, byte_off, word_off)
        _check(lib().gtsc_bytes_to_words_device(self._h, out_nwords.numel(), _dp(data), _dp(byte_off), int(uniform_slot_bytes),
                                                _dp(nbytes), _dp(out_words), _dp(word_off), int(uniform_slot_words),
                                                _dp(out_nwords)))

    def decode_device(self, words, ts, val, out_npoints, nseries, in_off=None, uniform_slot_words=0, in_nwords=None,
                      pt_off=None, uniform_capacity=0, status=None, fmt=FMT_V2):
        self._hold(words, ts, val, out_npoints, in_off, in_nwords, pt_off, status)
        _check(lib().gtsc_decode_device(self._h, fmt, int(nseries), _dp(words), _dp(in_off), int(uniform_slot_words),
                                        _dp(in_nwords), _dp(ts), _dp(val), _dp(pt_off), int(uniform_capacity),
                                        _dp(out_npoints), _dp(status)))

    def compact_device(self, slot_words, nwords, compact_words, compact_off, slot_off=None, uniform_slot_words=0,
                       noalloc=False):
        self._hold(slot_words, nwords, compact_words, compact_off, slot_off)
        f = lib().gtsc_compact_device_noalloc if noalloc else lib().gtsc_compact_device
        _check(f(self._h, nwords.numel(), _dp(slot_words), _dp(slot_off), int(uniform_slot_words), _dp(nwords),
                 _dp(compact_words), _dp(compact_off)))

    # ---- host memory -----------------------------------------------------
    def encode_host(self, block_ts, ts, val, pt_off=None, uniform_npoints=0, fmt=FMT_V2, capacity_words=None,
                    options=0, want_nbits=False):
        """numpy in, numpy out: returns (words, off) with the images back to back
        (and the streams' bit lengths when want_nbits is set)."""
        block_ts = np.ascontiguousarray(block_ts, dtype=np.int64)
        ts = np.ascontiguousarray(ts, dtype=np.int64)
        val = np.ascontiguousarray(val, dtype=np.uint64)
        S = len(block_ts)
        if pt_off is not None:
            pt_off = np.ascontiguousarray(pt_off, dtype=np.uint64)
            lens = np.diff(pt_off.astype(np.int64))
        else:
            lens = np.full(S, int(uniform_npoints), dtype=np.int64)
        if capacity_words is None:
            bits = 64 + np.where(lens > 0, 91 + 114 * (lens - 1), 0) + 37
            capacity_words = int((bits // 64 + 2).sum())
        out = np.zeros(max(capacity_words, 1), dtype=np.uint64)
        off = np.zeros(S + 1, dtype=np.uint64)
        if options or want_nbits:
            nbits = np.zeros(S, dtype=np.uint64) if want_nbits else None
            _check(lib().gtsc_encode_host_ex(self._h, fmt, int(options), S, _hp(block_ts), _hp(ts), _hp(val), _hp(pt_off),
                                             int(uniform_npoints), _hp(out), int(capacity_words), _hp(off), _hp(nbits)))
            return (out[: int(off[-1])], off, nbits) if want_nbits else (out[: int(off[-1])], off)
        _check(lib().gtsc_encode_host(self._h, fmt, S, _hp(block_ts), _hp(ts), _hp(val), _hp(pt_off),
                                      int(uniform_npoints), _hp(out), int(capacity_words), _hp(off)))
        return out[: int(off[-1])], off

    def decode_host(self, words, in_off, pt_off=None, uniform_capacity=0, fmt=FMT_V2, check=True):
        """numpy in, numpy out: returns (ts, val, npoints, status)."""
        words = np.ascontiguousarray(words, dtype=np.uint64)
        in_off = np.ascontiguousarray(in_off, dtype=np.uint64)
        S = len(in_off) - 1
        if pt_off is not None:
            pt_off = np.ascontiguousarray(pt_off, dtype=np.uint64)
            total = int(pt_off[-1])
        else:
            total = S * int(uniform_capacity)
        ts = np.zeros(total, dtype=np.int64)
        val = np.zeros(total, dtype=np.uint64)
        npts = np.zeros(S, dtype=np.uint64)
        st = np.zeros(S, dtype=np.uint32)
        rc = lib().gtsc_decode_host(self._h, fmt, S, _hp(words), _hp(in_off), _hp(ts), _hp(val), _hp(pt_off),
                                    int(uniform_capacity), _hp(npts), _hp(st))
        if check:
            _check(rc)
        return ts, val, npts, st
