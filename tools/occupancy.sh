#!/bin/bash
# Runs on the GPU box: mean waves per CU and which resource held launches back, per kernel.
OUT=gpurun_out/occ; mkdir -p $OUT; export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline"
timeout -s KILL 240 rocprofv3 --pmc MeanOccupancyPerActiveCU --output-format csv -d $OUT/occ -- python3 bench.py $ARGS > $OUT/occ.log 2>&1 || { echo occ failed; tail -3 $OUT/occ.log; }
timeout -s KILL 240 rocprofv3 --pmc SPI_RA_LDS_CU_FULL_CSN SPI_RA_VGPR_SIMD_FULL_CSN --output-format csv -d $OUT/spi -- python3 bench.py $ARGS > $OUT/spi.log 2>&1 || { echo spi failed; tail -3 $OUT/spi.log; }
python3 - <<'PY'
import csv, glob, collections
for d in ("occ", "spi"):
    acc = collections.defaultdict(list)
    for p in glob.glob("gpurun_out/occ/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(p)):
            if "gtsc" in r["Kernel_Name"]:
                acc[(r["Kernel_Name"][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(k, sum(v) / len(v), "LDS", end=" ")
        print()
PY
