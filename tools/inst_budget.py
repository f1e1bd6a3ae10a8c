#!/usr/bin/env python3
"""Static instruction budget of one decoded point in the lane-per-series decompressor, from the ISA.

Compiles kernels.hip for gfx950 with line tables (no GPU needed), takes decode_tiled_kernel<FMT_V2, RING>, finds the
second of the unrolled pairs of the lock-step loop and walks the instructions a wave issues for ONE point along a
stated path (which way each conditional branch goes is part of the variant, and printed).  Every instruction is put
in a class by the source line its line-table entry names (window assembly, code parse, payload extract, commit,
votes, ring upkeep, tile writes, ...) and by the unit that issues it (VALU, SALU, LDS, VMEM, branch, wait).

usage: tools/inst_budget.py [--ring 16|8] [--keep DIR] > profiles/<name>.txt
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile
from collections import OrderedDict, defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gorilla-tsc_amd", "csrc")


def compile_listing(workdir):
    out = os.path.join(workdir, "kernels_dbg.s")
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-gline-tables-only", "--cuda-device-only", "-S",
           os.path.join(CSRC, "kernels.hip"), "-I", CSRC, "-o", out]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    return out


def source_line_of(path, needle):
    for i, ln in enumerate(open(path), 1):
        if needle in ln and not ln.lstrip().startswith(("//", "#define")):
            return i
    raise SystemExit("no line with %r in %s" % (needle, path))


def load_kernel(listing, symbol):
    """-> (instructions [(text, file, line)], labels {name: index}, file table {no: basename})"""
    ins, labels, files = [], {}, {}
    cur, inside = (0, 0), False
    for ln in open(listing):
        t = ln.strip()
        m = re.match(r'\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', t) or re.match(r'\.file\s+(\d+)\s+"([^"]+)"', t)
        if m:
            files[int(m.group(1))] = os.path.basename(m.group(2))
        if not inside:
            inside = t.startswith(symbol + ":")
            continue
        if t.startswith(".loc"):
            f = t.split()
            cur = (int(f[1]), int(f[2]))
            continue
        if not t or t.startswith(";"):
            continue
        m = re.match(r"^(\.LBB[0-9_]+):", t)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        if t.startswith(".Lfunc_end"):
            break
        if t.startswith(".") or re.match(r"^[A-Za-z_.$0-9]+:", t):
            continue
        text = t.split(";")[0].strip()
        ins.append((text, cur[0], cur[1]))
    return ins, labels, files


def unit_of(op):
    if op.startswith("s_cbranch") or op == "s_branch":
        return "branch"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "VMEM"
    if op.startswith("s_"):
        return "SALU"
    return "VALU"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ring", type=int, default=16)
    ap.add_argument("--keep", default=None, help="directory for the listing (default: a temporary one)")
    args = ap.parse_args()
    tmp = args.keep or tempfile.mkdtemp(prefix="inst_budget_")
    os.makedirs(tmp, exist_ok=True)
    listing = compile_listing(tmp)
    khip = os.path.join(CSRC, "kernels.hip")
    L0, L1 = source_line_of(khip, "GTSC_POINT(0)"), source_line_of(khip, "GTSC_POINT(1)")
    symbol = "_ZN4gtsc19decode_tiled_kernelILi0ELj%dEEEvNS_10DecodeArgsE" % args.ring
    ins, labels, files = load_kernel(listing, symbol)
    # (the same file may have several numbers in the table: instructions carry the file's name from here on)
    ins = [(t, files.get(f, "?"), l) for t, f, l in ins]
    codec, main_src = "codec_device.hpp", "kernels.hip"
    dev_fn = ["amd_device_functions.h"]

    # line ranges of the decoder's source by what the lines do (found by their text, so that edits elsewhere do not move them)
    csrc_lines = open(os.path.join(CSRC, "codec_device.hpp")).read().split("\n")

    def at(needle, after=0):
        for i in range(after, len(csrc_lines)):
            if needle in csrc_lines[i]:
                return i + 1
        raise SystemExit("codec_device.hpp: no line with %r" % needle)
    L_WINDOW = at("return (a << off) | ((b >> 1) >> (63 - off));")
    L_TS = (at("const uint32_t top = (uint32_t)(hi >> 32);"), at("p.dd = ones ? dd : 0;") + 1)
    L_HDR = (at("const uint32_t xh = (uint32_t)((hi << clen) >> 32);"), at("const uint32_t n = c0 ? p.nbits : 0u;"))
    L_PAY = (at("uint64_t pay = hi << p2;"), at("p.bits = "))
    L_POS = (at("p.o1 = off + clen;"), at("p.o3 = p.o2 + n;"))
    L_ADV = (at("bool runs_off(uint32_t o) const"), at("void advance(uint32_t o)"))
    L_COMMIT = (at("void commit(const Parsed& p)"), at("stored_ts += (uint64_t)stored_delta;", at("void commit(const Parsed& p)")))
    L_VOTES = (at("uint64_t lanes(bool p)"), at("bool all_lanes(bool p)"))

    def within(l, r):
        return r[0] <= l <= r[1]

    def klass(text, f, l):
        op = text.split()[0]
        u = unit_of(op)
        if f == codec:
            if l == L_WINDOW:
                return "window assembly (64 bits at the lane's bit position out of two ring words)"
            if within(l, L_TS):
                return "timestamp code parse (bucket, payload, zig-zag)"
            if within(l, L_HDR):
                return "value header parse (control bits, lead/sig fields, selects against the kept window)"
            if within(l, L_PAY):
                return "value payload extract (shift, mask; the second window half when a lane needs it)"
            if within(l, L_POS) or within(l, L_ADV):
                return "bit position upkeep (offsets, word index)"
            if within(l, L_COMMIT):
                return "commit arithmetic (xor into the value, delta and timestamp adds)"
            if within(l, L_VOTES):
                return "votes (the comparison that makes each lane mask)"
            return "other decoder code"
        if f in dev_fn:
            return "timestamp code parse (bucket, payload, zig-zag)"
        if f == main_src and l in (L0, L1):
            if u == "LDS":
                return "ring window read (LDS)" if op.startswith("ds_read") else "tile writes (LDS)"
            if u in ("SALU", "branch"):
                return "vote combining and the path branches (scalar)"
            if op.startswith("v_mov"):
                return "state hand-over (the tried point becomes the decoder's: 64-bit moves)"
            if u == "wait":
                return "waits (s_waitcnt)"
            return "addresses (ring slot, tile slot) and the point count"
        if f == main_src and (L0 < l < L0 + 40 or l in range(120, 140)):
            return "ring upkeep (top-up checks, landing and issuing of global loads)"
        if u == "wait":
            return "waits (s_waitcnt)"
        return "compiler glue (exec mask restores, copies without a source line)"

    # the start of a point: `if (slow_m == 0)` -- s_cmp_lg_u64 + s_mov_b64 + s_cbranch_scc1 at the macro's line
    def starts(line):
        r = []
        for i in range(len(ins) - 2):
            if ins[i][0].startswith("s_cmp_lg_u64") and ins[i][2] == line and ins[i][1] == main_src \
                    and ins[i + 2][0].startswith("s_cbranch_scc1") and ins[i + 2][2] == line:
                r.append(i)
        return r
    s0, s1 = starts(L0), starts(L1)
    if len(s0) < 3 or len(s1) < 3:
        raise SystemExit("could not find the unrolled points (%d, %d)" % (len(s0), len(s1)))
    begin = s0[1]
    end = set(s1)   # (the compiler keeps several copies of the next point's head: whichever the path reaches)

    def walk(rules, stop):
        """follows the path from `begin`.  rules: {(branch kind, source line): [taken?, ...]} consumed per occurrence;
        a branch over an empty exec mask (execz) without a rule falls through (some lane is always live)."""
        counts = OrderedDict()
        path = []
        i, used = begin, defaultdict(int)
        n = 0
        while i not in stop:
            text, f, l = ins[i]
            op = text.split()[0]
            k = klass(text, f, l)
            counts.setdefault(k, defaultdict(int))[unit_of(op)] += 1
            n += 1
            if n > 2000:
                raise SystemExit("path does not reach its end: " + " ".join(path))
            if op == "s_branch":
                i = labels[text.split()[1]]
                continue
            if op.startswith("s_cbranch"):
                kind = op.replace("s_cbranch_", "")
                key = (kind, l if f == main_src else f + ":%d" % l)
                if key in rules and used[key] < len(rules[key]):
                    taken = rules[key][used[key]]
                    used[key] += 1
                elif kind == "execz":
                    taken = False
                else:
                    raise SystemExit("no rule for %s (%s:%d) after %s" % (text, f, l, path))
                path.append("%s@%s:%d %s" % (kind, f, l, "taken" if taken else "falls through"))
                i = labels[text.split()[1]] if taken else i + 1
                continue
            i += 1
        return counts, path

    units = ["VALU", "SALU", "LDS", "VMEM", "branch", "wait", "nop"]

    def report(title, note, decisions, stop):
        counts, path = walk(decisions, stop)
        print("== " + title)
        print("   " + note)
        print("   path: " + ", ".join(path))
        print("   %-92s %s" % ("class", " ".join("%6s" % u for u in units)))
        tot = defaultdict(int)
        for k, c in counts.items():
            print("   %-92s %s" % (k, " ".join("%6d" % c.get(u, 0) for u in units)))
            for u in units:
                tot[u] += c.get(u, 0)
        print("   %-92s %s" % ("TOTAL", " ".join("%6d" % tot[u] for u in units)))
        print()
        return tot

    print("Instruction budget of one point, decode_tiled_kernel<FMT_V2, %d> (static, from the gfx950 ISA with line tables;" % args.ring)
    print("point = GTSC_POINT(0) of the second unrolled pair, kernels.hip:%d, up to the GTSC_POINT(1) that follows it, kernels.hip:%d)." % (L0, L1))
    print("Every lane of a wave is one series, so a wave-instruction here is 1/64 instruction per point and series;")
    print("SQ_INSTS_* per point and wave in the PMC summaries are dynamic averages of these paths and of the chunk code around them.")
    print()
    L_PAY_VOTE = at("if (any_lane(p2 + n > 64))")
    return report, end, labels, L_PAY_VOTE


if __name__ == "__main__":
    report, end, labels, L_PAY_VOTE = main()
    L0 = int([ln for ln in open(os.path.join(CSRC, "kernels.hip")).read().split("\n") if "GTSC_POINT(0)" in ln and "#define" not in ln][0] and
             next(i for i, ln in enumerate(open(os.path.join(CSRC, "kernels.hip")), 1) if "GTSC_POINT(0)" in ln and "#define" not in ln and not ln.lstrip().startswith("//")))

    def rules(ts_zero, fits, pend, low):
        return {
            ("scc1", L0): [False, ts_zero],          # slow_m != 0 ? ; every live lane's timestamp code is the bit '0' ?
            ("vccz", "codec_device.hpp:%d" % L_PAY_VOTE): [fits],  # no lane's payload reaches past the first 64 window bits ?
            ("scc0", L0): [True],                    # rare_m == 0: the point commits in every lane
            ("scc1", L0 + 1): [False],               # dense == 0 ? (a dense stream looks at its ring after every point)
            ("execz", L0 + 2): [not pend],           # no lane has a top-up in flight
            ("vccz", L0 + 2): [True],                # the landing words are not the image's last ones
            ("execz", L0 + 4): [not low],            # no lane is due a top-up
        }
    variants = [
        ("A. regular cadence, payload inside the first window half, nothing to land or request",
         "gauges, drift: the common point of a compressible stream", rules(True, True, False, False)),
        ("B. regular cadence, some lane's payload crosses into the second window half, nothing to land or request",
         "random walk, adversarial values: 50 and more payload bits per point", rules(True, False, False, False)),
        ("C. as B, and a top-up lands and the next one is requested (every lane in step)",
         "a 32-byte top-up lasts 4 points at 64 bits per point, 2 at 128: B and C alternate accordingly", rules(True, False, True, True)),
        ("D. some lane carries a timestamp code, payload crosses, nothing to land or request",
         "jittered timestamps over dense values", rules(False, False, False, False)),
    ]
    for title, note, r in variants:
        try:
            report(title, note, r, end)
        except SystemExit as e:
            print("== " + title + "\n   not walked: " + str(e) + "\n")
