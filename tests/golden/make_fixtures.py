#!/usr/bin/env python3
"""Regenerates tests/golden/*.json.  Run in the build container only (it reads
the reference's test sources under /root/reference, which do not travel).

Two files are written:

reference_test_inputs.json
    The literal input data of the reference's own tests (values only):
    src/test/java/fi/iki/yak/ts/compression/gorilla/EncodeGorillaTest.java
      :52-61 (simpleEncodeAndDecodeTest), :68-324 (willItBlend),
      :338-349 (testEncodeSimilarFloats)
    and the identical tables of EncodeTest.java.  The reference asserts only
    decode(encode(x)) == x on them; so do we (bit-exactly).

known_answers.json
    Expected compressed images.  NOT captured from a JVM (none exists in this
    image): each image is produced by the C oracle, required to equal the
    independent bit-string restatement (oracle/spec_codec.py), and -- where a
    vector was derived by hand or by the survey's two scratch restatements --
    compared with that literal too.
"""
import hashlib
import json
import os
import re
import struct
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as O  # noqa: E402
import spec_codec as SP  # noqa: E402

REF_T = "/root/reference/src/test/java/fi/iki/yak/ts/compression/gorilla/"


def d2u(x):
    return struct.unpack("<Q", struct.pack("<d", float(x)))[0]


def parse_blend(path):
    src = open(path).read()
    seg = src[src.index("willItBlend"):]
    seg = seg[: seg.index("comparePairsToCompression")]
    block = int(re.search(r"blockTimestamp = (\d+)L", seg).group(1))
    pairs = [(int(a), int(b)) for a, b in re.findall(r"new Pair\((\d+)L,\s*(\d+)\)", seg)]
    return block, pairs


def main():
    b1, p1 = parse_blend(REF_T + "EncodeGorillaTest.java")
    b2, p2 = parse_blend(REF_T + "EncodeTest.java")
    assert (b1, p1) == (b2, p2) and len(p1) == 254
    inputs = {
        "willItBlend": {"block_ts": b1, "ts": [p[0] for p in p1], "val_bits": [p[1] for p in p1]},
        "simple": {
            # block is wall-clock "now" truncated to the hour in the reference; any hour works
            "block_ts": 1500400800000,
            "offsets": [10, 20, 28, 84, 400, 2300, 16384, 16500],
            "doubles": [1.0, -2.0, -2.5, 65537.0, 2147483650.0, -16384.0, 2.8, -38.0],
        },
        "similarFloats": {
            "block_ts": 1425254400000,  # 2015-03-02T00:00Z
            "offsets": [1, 2, 3, 4, 5],
            "doubles": [6.00065e+06, 6.000656e+06, 6.000657e+06, 6.000659e+06, 6.000661e+06],
        },
    }
    json.dump(inputs, open(os.path.join(HERE, "reference_test_inputs.json"), "w"), indent=0)

    cases = {}

    def add(name, block, ts, vals, fmt, close=True, literal=None, literal_sha=None):
        w = [int(x) for x in O.encode_block(block, ts, vals, fmt=O.V2 if fmt == "v2" else O.V1, close=close)]
        bits = SP.encode_bits(block, list(zip(ts, vals)), fmt, close=close)
        assert w == SP.words_image(bits, closed=close), name
        if literal is not None:
            assert w == literal, name
        be = b"".join(struct.pack(">Q", x) for x in w)
        sha = hashlib.sha256(be).hexdigest()
        if literal_sha is not None:
            assert sha == literal_sha, name
        cases[name] = {
            "format": fmt, "closed": close, "block_ts": block, "ts": list(ts), "val_bits": [int(v) for v in vals],
            "nbits": len(bits), "words_hex": ["%016x" % x for x in w], "sha256_be": sha,
            "bytes_hex": SP.bytes_image(bits, closed=close).hex(),
        }

    B = 1500400800000
    # hand-derived: header word; 27-bit sentinel-free close marker 0xF FFFFFFFF 0 right after it
    add("empty_v2", B, [], [], "v2", literal=[0x0000015D56DB5100, 0xFFFFFFFFF0000000, 0x0])
    add("empty_v1", B, [], [], "v1", literal=[0x0000015D56DB5100, 0xFFFFFFFFF0000000, 0x0])
    one = d2u(1.0)
    add("two_equal_v2", B, [B + 1, B + 2], [one, one], "v2",
        literal=[0x15D56DB5100, 0x27FE000000, 0x7FFFFFFFF, 0x8000000000000000, 0x0])
    add("two_equal_v2_snapshot", B, [B + 1, B + 2], [one, one], "v2", close=False,
        literal=[0x15D56DB5100, 0x27FE000000, 0x0])
    add("word_boundary_v2", B, [B + 1000], [d2u(5.0)], "v2")
    assert cases["word_boundary_v2"]["nbits"] == 192 and cases["word_boundary_v2"]["words_hex"][-3:] == [
        "0000001ffffffffe", "0" * 16, "0" * 16]
    sf = inputs["similarFloats"]
    ts = [sf["block_ts"] + o for o in sf["offsets"]]
    vals = [d2u(x) for x in sf["doubles"]]
    add("similarFloats_v2", sf["block_ts"], ts, vals, "v2",
        literal=[0x14BD7C84000, 0x282ADC8050, 0xDD0F5C205, 0xC005BE0FFFFFFFFF, 0xE000000000000000, 0x0])
    add("similarFloats_v1", sf["block_ts"], ts, vals, "v1",
        literal=[0x14BD7C84000, 0x282ADC8050, 0xFA26BF865, 0x25BFFFFFFFFC0000, 0x0])
    si = inputs["simple"]
    ts = [si["block_ts"] + o for o in si["offsets"]]
    vals = [d2u(x) for x in si["doubles"]]
    add("simple_v2", si["block_ts"], ts, vals, "v2",
        literal=[0x15D56DB5100, 0x147FE000000, 0xC02FFFE05, 0x9A06BF80DC07A000, 0xF103E3C510001000, 0x7B17F02981300000,
                 0x7C00017CBF03E8, 0xD6666666666667E, 0xDA3D4022B333, 0x33333333FFFFFFFF, 0xF000000000000000, 0x0])
    add("simple_v1", si["block_ts"], ts, vals, "v1",
        literal=[0x15D56DB5100, 0x147FE000000, 0xC067FFDFB, 0x68398607203D0007, 0x84133C7100010007, 0x98C3055026000000,
                 0xF800017CC60FE03, 0x5999999999999FFF, 0xFFE4B85008ACCCCC, 0xCCCCCCFFFFFFFFFC, 0x0])
    wb = inputs["willItBlend"]
    add("willItBlend_v2", wb["block_ts"], wb["ts"], wb["val_bits"], "v2",
        literal_sha="619a5caf727baca8ab84b0f93d438eae809ac06acd65692b8491a5059d8b0c5f")
    add("willItBlend_v1", wb["block_ts"], wb["ts"], wb["val_bits"], "v1",
        literal_sha="c28c67d7199d416e40d7f65a0805dd20dda690477b7c0bd78495d8c5e252a02e")
    # delta-of-delta bucket edges: third point at B+30000+d after two at a 10 s cadence
    for d in (-1, 1, -2, 2, 64, -64, 65, -65, 256, -256, 257, -257, 2048, -2048, 2049, -2049, 100000, -100000):
        for fmt in ("v2", "v1"):
            add("dod_%s_%d" % (fmt, d), B, [B + 10000, B + 20000, B + 30000 + d], [0, 0, 0], fmt)
    json.dump(cases, open(os.path.join(HERE, "known_answers.json"), "w"), indent=0)
    print("wrote", len(inputs), "input sets and", len(cases), "known answers")


if __name__ == "__main__":
    main()
