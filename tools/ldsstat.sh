#!/bin/bash
# Runs on the GPU box: LDS and wait counters of the codec kernels.  usage: tools/ldsstat.sh <tag> [bench args...]
# With GTSC_LIB pointing at a build with -DGTSC_ABLATE=5|6|7 (decoder without its tile writes / ring reads / both;
# results are wrong, hence --no-verify) the differences give the conflicts per class of LDS instruction.
TAG=$1; shift
OUT=gpurun_out/lds_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-verify $*"
timeout -s KILL 240 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_sq1 -- python3 bench.py $ARGS > $OUT/sq1.log 2>&1 || { echo failed; tail -3 $OUT/sq1.log; }
python3 tools/summarize_profile.py $OUT | grep -E "gtsc|SQ_"
