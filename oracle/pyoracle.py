"""ctypes loader for the C oracle (oracle/liboracle.so): TEST INFRASTRUCTURE ONLY.

Imported by tests/, the smoke check and bench.py's cpu_baseline leg; never by
the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")

V2, V1 = 0, 1
OK, END, ERR_TRUNCATED, ERR_MALFORMED = 0, 1, -1, -2


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("gorilla_oracle.c", "gorilla_oracle.h")]
    if force or not os.path.exists(_LIB) or any(
            os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(_LIB) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        L = C.CDLL(_LIB)
        vp, u64, i64, i32 = C.c_void_p, C.c_uint64, C.c_int64, C.c_int
        L.gor_enc_new.restype = vp
        L.gor_enc_new.argtypes = [i32, i64, i32]
        L.gor_enc_free.argtypes = [vp]
        L.gor_enc_add.argtypes = [vp, i64, u64]
        L.gor_enc_close.argtypes = [vp]
        L.gor_enc_bits.restype = u64
        L.gor_enc_bits.argtypes = [vp]
        L.gor_enc_words.restype = C.c_size_t
        L.gor_enc_words.argtypes = [vp, vp, C.c_size_t]
        L.gor_enc_bytes.restype = C.c_size_t
        L.gor_enc_bytes.argtypes = [vp, vp, C.c_size_t]
        L.gor_dec_new.restype = vp
        L.gor_dec_new.argtypes = [i32, vp, C.c_size_t, i32]
        L.gor_dec_free.argtypes = [vp]
        L.gor_dec_next.restype = i32
        L.gor_dec_next.argtypes = [vp, C.POINTER(i64), C.POINTER(u64)]
        L.gor_max_words.restype = u64
        L.gor_max_words.argtypes = [u64]
        L.gor_encode_batch.restype = i32
        L.gor_encode_batch.argtypes = [i32, u64, vp, vp, vp, vp, vp, vp, vp, vp, i32]
        L.gor_decode_batch.restype = i32
        L.gor_decode_batch.argtypes = [i32, u64, vp, vp, vp, vp, vp, vp, vp, vp, i32]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Encoder:
    """One block, streaming: mirrors GorillaCompressor / Compressor usage."""

    def __init__(self, block_ts, fmt=V2, dfcm_size=0):
        self._h = lib().gor_enc_new(fmt, int(block_ts), int(dfcm_size))

    def add(self, ts, value_bits):
        lib().gor_enc_add(self._h, int(ts), int(value_bits) & 0xFFFFFFFFFFFFFFFF)

    def close(self):
        lib().gor_enc_close(self._h)

    @property
    def nbits(self):
        return lib().gor_enc_bits(self._h)

    def words(self):
        n = lib().gor_enc_words(self._h, None, 0)
        a = np.zeros(n, dtype=np.uint64)
        lib().gor_enc_words(self._h, _p(a), n)
        return a

    def bytes(self):
        n = lib().gor_enc_bytes(self._h, None, 0)
        a = np.zeros(n, dtype=np.uint8)
        lib().gor_enc_bytes(self._h, _p(a), n)
        return a.tobytes()

    def __del__(self):
        if getattr(self, "_h", None):
            lib().gor_enc_free(self._h)
            self._h = None


def encode_block(block_ts, ts, vals, fmt=V2, close=True, dfcm_size=0):
    e = Encoder(block_ts, fmt, dfcm_size)
    for t, v in zip(ts, vals):
        e.add(int(t), int(v))
    if close:
        e.close()
    return e.words()


def decode_block(words, fmt=V2, dfcm_size=0):
    """Returns (ts int64[], val uint64[], rc) with rc END or a negative error."""
    w = np.ascontiguousarray(words, dtype=np.uint64)
    d = lib().gor_dec_new(fmt, _p(w), len(w), int(dfcm_size))
    ts, vs = [], []
    t, v = C.c_int64(), C.c_uint64()
    while True:
        rc = lib().gor_dec_next(d, C.byref(t), C.byref(v))
        if rc != OK:
            break
        ts.append(t.value)
        vs.append(v.value)
    lib().gor_dec_free(d)
    return np.array(ts, dtype=np.int64), np.array(vs, dtype=np.uint64), rc


def max_words(p):
    return int(lib().gor_max_words(int(p)))


def encode_batch(block_ts, ts, val, pt_off, out_off=None, fmt=V2, nthreads=1):
    """Returns (out_words, out_off, out_nwords, status)."""
    block_ts = np.ascontiguousarray(block_ts, dtype=np.int64)
    ts = np.ascontiguousarray(ts, dtype=np.int64)
    val = np.ascontiguousarray(val, dtype=np.uint64)
    pt_off = np.ascontiguousarray(pt_off, dtype=np.uint64)
    S = len(block_ts)
    if out_off is None:
        lens = np.diff(pt_off.astype(np.int64))
        bits = 64 + np.where(lens > 0, 91 + 114 * (lens - 1), 0) + 37
        out_off = np.zeros(S + 1, dtype=np.uint64)
        np.cumsum(bits // 64 + 2, out=out_off[1:])
    out_off = np.ascontiguousarray(out_off, dtype=np.uint64)
    out = np.zeros(int(out_off[-1]), dtype=np.uint64)
    nw = np.zeros(S, dtype=np.uint64)
    st = np.zeros(S, dtype=np.uint32)
    lib().gor_encode_batch(fmt, S, _p(block_ts), _p(ts), _p(val), _p(pt_off), _p(out), _p(out_off),
                           _p(nw), _p(st), nthreads)
    return out, out_off, nw, st


def decode_batch(words, in_off, in_nwords, pt_off, fmt=V2, nthreads=1):
    """Returns (ts, val, npoints, status)."""
    words = np.ascontiguousarray(words, dtype=np.uint64)
    in_off = np.ascontiguousarray(in_off, dtype=np.uint64)
    in_nwords = None if in_nwords is None else np.ascontiguousarray(in_nwords, dtype=np.uint64)
    pt_off = np.ascontiguousarray(pt_off, dtype=np.uint64)
    S = len(in_off) - 1
    ts = np.zeros(int(pt_off[-1]), dtype=np.int64)
    val = np.zeros(int(pt_off[-1]), dtype=np.uint64)
    npts = np.zeros(S, dtype=np.uint64)
    st = np.zeros(S, dtype=np.uint32)
    lib().gor_decode_batch(fmt, S, _p(words), _p(in_off), _p(in_nwords), _p(ts), _p(val), _p(pt_off),
                           _p(npts), _p(st), nthreads)
    return ts, val, npts, st
