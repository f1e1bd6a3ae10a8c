"""Bit-string restatement of the Gorilla formats: TEST INFRASTRUCTURE ONLY.

A second, independently shaped statement of what the reference
(burmanm/gorilla-tsc, src/main/java/fi/iki/yak/ts/compression/gorilla/, "G/")
writes, used to cross-check the C oracle.  It is written as a closed form per
point producing a Python string of '0'/'1' characters, so it shares no
structure (bit writer, word flipping, state layout) with the C oracle.

Parity status: not captured from a JVM (none is available); agreement between
this file, the C oracle and hand-derived vectors is the best available pin on
the compressed bytes.  Only small inputs: pure-Python loops.
"""

M64 = (1 << 64) - 1
M32 = (1 << 32) - 1


def _i32(x):
    x &= M32
    return x - (1 << 32) if x >> 31 else x


def _i64(x):
    x &= M64
    return x - (1 << 64) if x >> 63 else x


def _bits(value, n):
    """low n bits of value, MSB first (G/LongArrayOutput.java:115-131 masks to n bits)"""
    return format(value & ((1 << n) - 1), "0%db" % n) if n else ""


def _clz64(x):
    return 64 - x.bit_length()


def _ctz64(x):
    return (x & -x).bit_length() - 1


def encode_bits(block_ts, points, fmt="v2", close=True):
    """points: iterable of (ts:int64, value_bits:uint64). Returns the bit string."""
    out = [_bits(block_ts, 64)]  # G/GorillaCompressor.java:39-41
    stored_ts = 0
    stored_delta = 0
    prev = 0
    lead, trail = None, 0  # None = Integer.MAX_VALUE (G/ValueCompressor.java:12)
    for ts, v in points:
        v &= M64
        if stored_ts == 0:  # G/GorillaCompressor.java:50,65 ; G/Compressor.java:43,58
            d = _i64(ts - block_ts)
            stored_delta = _i32(d) if fmt == "v2" else d
            stored_ts = ts
            out.append(_bits(stored_delta, 27))
            out.append(_bits(v, 64))
            prev = v
            continue
        # ---- timestamp
        if fmt == "v2":  # G/GorillaCompressor.java:99-143
            nd = _i32(ts - stored_ts)
            dod = _i32(nd - stored_delta)
            if dod == 0:
                out.append("0")
            else:
                z = ((((dod << 1) & M32) ^ ((dod >> 31) & M32)) - 1) & M32
                if 1 <= z <= 127:
                    out.append("10" + _bits(z, 7))
                elif 128 <= z <= 511:
                    out.append("110" + _bits(z, 9))
                elif 512 <= z <= 4095:
                    out.append("1110" + _bits(z, 12))
                else:  # z == 0 (dod == -1) or z >= 4096
                    out.append("1111" + _bits(z, 32))
                stored_delta = nd
        else:  # G/Compressor.java:94-118
            nd = _i64(ts - stored_ts)
            dod = _i64(nd - stored_delta)
            if dod == 0:
                out.append("0")
            elif -63 <= dod <= 64:
                out.append("10" + _bits(dod, 7))
            elif -255 <= dod <= 256:
                out.append("110" + _bits(dod, 9))
            elif -2047 <= dod <= 2048:
                out.append("1110" + _bits(dod, 12))
            else:
                out.append("1111" + _bits(dod, 32))
            stored_delta = nd
        stored_ts = ts
        # ---- value
        x = prev ^ v
        prev = v
        if x == 0:
            out.append("0")
            continue
        lz, tz = _clz64(x), _ctz64(x)
        if fmt == "v1" and lz >= 32:  # G/Compressor.java:132-134
            lz = 31
        if lead is not None and lz >= lead and tz >= trail:
            out.append("10" + _bits(x >> trail, 64 - lead - trail))
        else:
            sig = 64 - lz - tz
            if fmt == "v2":  # G/ValueCompressor.java:78-91
                out.append("11" + _bits(lz, 6) + _bits(sig - 1, 6) + _bits(x >> tz, sig))
            else:  # G/Compressor.java:171-181 (sig 64 stored as 0 by the 6-bit mask)
                out.append("11" + _bits(lz, 5) + _bits(sig, 6) + _bits(x >> tz, sig))
            lead, trail = lz, tz
    if close:  # G/GorillaCompressor.java:84-89
        out.append("1111" + "1" * 32 + "0")
    return "".join(out)


def words_image(bits, closed=True):
    """LongArrayOutput.getLongArray() image for a bit string (G/LongArrayOutput.java:136-145)."""
    n = len(bits)
    nwords = n // 64 + (2 if closed else 1)
    padded = bits + "0" * (nwords * 64 - n)
    return [int(padded[i * 64:(i + 1) * 64], 2) for i in range(nwords)]


def bytes_image(bits, closed=True):
    """ByteBufferBitOutput image after flush()+flip() (G/ByteBufferBitOutput.java:93-96)."""
    n = len(bits)
    nbytes = n // 8 + (1 if closed else 0)
    padded = bits + "0" * max(0, nbytes * 8 - n)
    return bytes(int(padded[i * 8:(i + 1) * 8], 2) for i in range(nbytes))


def decode_words(words, fmt="v2"):
    """Inverse, following G/GorillaDecompressor.java:51-112 / G/Decompressor.java:44-130.
    Returns list of (ts, value_bits).  Raises IndexError on a truncated stream."""
    bits = "".join(format(w & M64, "064b") for w in words)
    limit = len(bits)
    pos = 0

    def get(n):
        nonlocal pos
        if pos + n >= limit:  # eager next-word load, G/LongArrayInput.java:48-52
            raise IndexError("truncated")
        v = int(bits[pos:pos + n], 2)
        pos += n
        return v

    def prefix(maxbits):
        val = 0
        for _ in range(maxbits):
            val <<= 1
            if get(1):
                val |= 1
            else:
                break
        return val

    block_ts = _i64(get(64))
    out = []
    stored_ts = 0
    stored_delta = 0
    val = 0
    lead, trail = None, 0
    while True:
        if stored_ts == 0:
            stored_delta = get(27)
            if stored_delta == (1 << 27) - 1:
                return out
            val = get(64)
            stored_ts = _i64(block_ts + stored_delta)
        else:
            c = prefix(4)
            nb = {0: 0, 2: 7, 6: 9, 14: 12, 15: 32}[c]
            dd = 0
            if nb:
                raw = get(nb)
                if nb == 32 and raw == M32:
                    return out
                if fmt == "v2":
                    z = (raw + 1) & M32
                    dd = _i32((z >> 1) ^ (-(z & 1) & M32))
                else:
                    if nb != 32 and raw > (1 << (nb - 1)):
                        raw -= 1 << nb
                    dd = _i32(raw)
            stored_delta = _i64(stored_delta + dd)
            stored_ts = _i64(stored_ts + stored_delta)
            if get(1):
                if get(1):
                    if fmt == "v2":
                        lead = get(6)
                        sig = get(6) + 1
                    else:
                        lead = get(5)
                        sig = get(6) or 64
                    trail = 64 - sig - lead
                n = 64 - lead - trail
                x = (get(n) << (trail & 63)) & M64
                val ^= x
        out.append((stored_ts, val))
