#!/usr/bin/env python3
"""Condenses the rocprofv3 CSVs written by tools/profile.sh into one table:
per gtsc kernel, average duration and average counter values per launch."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
rows = []
for path in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        if "gtsc" in r["Name"]:
            rows.append((r["Name"], int(r["Calls"]), float(r["AverageNs"]), float(r["MinNs"]), float(r["MaxNs"])))
print("== kernel-trace stats (ns)")
for name, calls, avg, mn, mx in rows:
    print("%-60s calls=%d avg=%.0f min=%.0f max=%.0f" % (name[:60], calls, avg, mn, mx))

def bench_line(path):
    if not os.path.exists(path):
        return None
    for ln in open(path):
        if ln.startswith("{") and '"metric"' in ln:
            return json.loads(ln)
    return None

# the bench lines of the same lease: the process under the trace and the untraced runs either side of it
print("== bench.py's own event times in this lease (ms: mean, and min/median/max over its timed steps)")
for label, fn in (("untraced, before", "untraced_before.log"), ("under the kernel trace", "trace.log"), ("untraced, after", "untraced_after.log")):
    b = bench_line(os.path.join(root, fn))
    if b is None:
        continue
    r = b["roofline"]
    es, ds = r.get("encode_ms_spread", {}), r.get("decode_ms_spread", {})
    print("%-24s compress %.4f (%s/%s/%s)  decompress %.4f (%s/%s/%s)  value %.1f %s  frac %.3f" % (
        label, r["encode_ms"], es.get("min"), es.get("median"), es.get("max"),
        r["decode_ms"], ds.get("min"), ds.get("median"), ds.get("max"), b["value"], b["unit"], r["frac"]))
counters = defaultdict(lambda: defaultdict(list))
for path in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        k = r.get("Kernel_Name", "")
        if "gtsc" in k:
            counters[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("== PMC averages per launch")
for k, cs in counters.items():
    print(k[:70])
    for c in sorted(cs):
        v = cs[c]
        print("   %-28s %.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))
    if "FETCH_SIZE" in cs:
        # FETCH_SIZE is in KiB and counts wide coalesced reads at half their bytes on gfx950
        f = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"])
        print("   fetched bytes (FETCH_SIZE KiB x1024 x2): %.4g" % (f * 1024 * 2))
    if "WRITE_SIZE" in cs:
        w = sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"])
        print("   written bytes (WRITE_SIZE KiB x1024):    %.4g" % (w * 1024))

# HBM traffic per launch for bench.py's roofline.traffic (default workload only)
traffic = {"workload": "gauge", "series": 1000000, "points": 120, "source": os.path.basename(root.rstrip("/"))}
for k, cs in counters.items():
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        f = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"]) * 1024 * 2
        w = sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"]) * 1024
        name = "encode" if "encode" in k else "decode" if "decode" in k else None
        if name and ("<0>" in k or "<0, " in k):  # the v2 kernels (the decoder also carries its ring size)
            traffic[name + "_bytes"] = round(f + w)
            traffic[name + "_fetch_bytes"] = round(f)
            traffic[name + "_write_bytes"] = round(w)
json.dump(traffic, open(os.path.join(root, "traffic.json"), "w"), indent=1)
