#!/bin/bash
# Runs on the GPU box: instruction mix of the codec kernels for a workload.  usage: tools/insts.sh <tag> [bench args...]
TAG=$1; shift
OUT=gpurun_out/insts_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline $*"
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc_sq1 -- python3 bench.py $ARGS > $OUT/sq1.log 2>&1 || echo failed
python3 tools/summarize_profile.py $OUT | grep -E "gtsc|SQ_"
