#!/bin/bash
# Runs on the GPU box: an untraced bench.py run, the same command under the kernel trace, a second untraced run,
# then separate PMC passes; the summary puts the three bench lines next to the trace's averages.
# usage: tools/profile.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/
set -u
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-extras $*"
run() { # name, rocprof flags...
  local name=$1; shift
  timeout -s KILL 240 rocprofv3 "$@" --output-format csv -d $OUT/$name -- python3 bench.py $ARGS > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $OUT/$name.log; return 1; }
}
# the kernel trace runs the benchmark's own default step counts, so that its averages are the bench line's
SHORT_ARGS=$ARGS; ARGS="--no-cpu-baseline --no-extras $*"
timeout -k 10 400 python3 bench.py $ARGS > $OUT/untraced_before.log 2>&1 || { echo "untraced run failed"; tail -5 $OUT/untraced_before.log; exit 1; }
run trace --kernel-trace --stats || exit 1
timeout -k 10 400 python3 bench.py $ARGS > $OUT/untraced_after.log 2>&1 || { echo "untraced run failed"; tail -5 $OUT/untraced_after.log; exit 1; }
ARGS=$SHORT_ARGS
run pmc_sq1 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU || exit 1
run pmc_sq2 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM || exit 1
run pmc_fetch --pmc FETCH_SIZE || exit 1
run pmc_write --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum || exit 1
run pmc_grbm --pmc GRBM_GUI_ACTIVE || exit 1
python3 tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
