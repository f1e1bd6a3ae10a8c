#!/bin/bash
# Copies what tools/profile.sh left in gpurun_out/prof_<tag>/ into the tracked profiles/ directory as <name>_*.
# usage: tools/keep_profile.sh <tag> <name> [traffic]     ("traffic": also refresh profiles/traffic.json, default workload only)
set -eu
SRC=gpurun_out/prof_$1; NAME=$2
cp $SRC/summary.txt profiles/${NAME}_summary.txt
cp "$(find $SRC/trace -name '*kernel_stats.csv' | head -1)" profiles/${NAME}_kernel_stats.csv
grep -h '^{' $SRC/trace.log | tail -1 > profiles/${NAME}_bench_in_traced_process.json
for w in before after; do
  [ -f $SRC/untraced_$w.log ] && grep -h '^{' $SRC/untraced_$w.log | tail -1 > profiles/${NAME}_bench_untraced_$w.json
done
[ "${3:-}" = traffic ] && cp $SRC/traffic.json profiles/traffic.json
ls -la profiles/${NAME}_*
