#!/bin/bash
# Runs on the GPU box: instruction-cache counters of the codec kernels for a workload.  usage: tools/icache.sh <tag> [bench args...]
TAG=$1; shift
OUT=gpurun_out/icache_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-extras $*"
timeout -s KILL 240 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $OUT/pmc_ic -- python3 bench.py $ARGS > $OUT/ic.log 2>&1 || { echo failed; tail -5 $OUT/ic.log; exit 1; }
python3 tools/summarize_profile.py $OUT | grep -E "gtsc|SQ"
