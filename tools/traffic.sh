#!/bin/bash
# Runs on the GPU box: HBM fetch and write bytes of the codec kernels for a workload.
# usage: tools/traffic.sh <tag> [bench args...]
TAG=$1; shift
OUT=gpurun_out/traffic_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline $*"
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $ARGS > $OUT/fetch.log 2>&1 || echo fetch failed
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- python3 bench.py $ARGS > $OUT/write.log 2>&1 || echo write failed
python3 tools/summarize_profile.py $OUT | grep -E "gtsc|FETCH|WRITE|TCC|bytes"
