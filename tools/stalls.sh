#!/bin/bash
# Runs on the GPU box: where the waves' cycles go (waits by cause, instruction fetch, issue
# activity) for a workload.  usage: tools/stalls.sh <tag> [bench args...]
TAG=$1; shift
OUT=gpurun_out/stalls_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline $*"
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc_sq1 -- python3 bench.py $ARGS > $OUT/sq1.log 2>&1 || { echo failed; tail -3 $OUT/sq1.log; }
timeout -s KILL 240 rocprofv3 --pmc SQ_WAIT_IFETCH SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_INSTS_VALU --output-format csv -d $OUT/pmc_sq2 -- python3 bench.py $ARGS > $OUT/sq2.log 2>&1 || { echo failed; tail -3 $OUT/sq2.log; }
python3 tools/summarize_profile.py $OUT | grep -E "gtsc|SQ_"
