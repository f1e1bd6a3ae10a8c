#!/bin/bash
# Runs on the GPU box: the evidence a round keeps under profiles/ -- the default bench line, the
# kernel-trace + PMC summaries of the four device-resident workloads, the per-GPU share of the
# 4M x 7,200 configuration, and the memory-pattern microbenchmarks.  usage: tools/round_evidence.sh <tag>
set -u
TAG=${1:-rX}
OUT=gpurun_out/evidence_$TAG
mkdir -p $OUT
timeout -k 10 400 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || { echo "bench failed"; tail -5 $OUT/bench_default.err; exit 1; }
bash tools/profile.sh ${TAG}_default > $OUT/prof_default.log 2>&1 || { echo "profile default failed"; exit 1; }
bash tools/profile.sh ${TAG}_drift --workload gauge_drift > $OUT/prof_drift.log 2>&1 || { echo "profile drift failed"; exit 1; }
bash tools/profile.sh ${TAG}_rw --workload random_walk --series 262144 --points 7200 > $OUT/prof_rw.log 2>&1 || { echo "profile rw failed"; exit 1; }
bash tools/profile.sh ${TAG}_adv --workload adversarial --series 65536 --points 7200 > $OUT/prof_adv.log 2>&1 || { echo "profile adv failed"; exit 1; }
bash tools/profile.sh ${TAG}_gauge7200 --workload gauge --series 262144 --points 7200 > $OUT/prof_gauge7200.log 2>&1 || { echo "profile gauge7200 failed"; exit 1; }
for W in gauge random_walk; do
  timeout -k 10 300 python bench.py --workload $W --series 524288 --points 7200 --steps 3 --warmup 2 --no-cpu-baseline --no-extras > $OUT/cfg3_share_524288x7200_$W.json 2>/dev/null || { echo "cfg3 share $W failed"; exit 1; }
done
timeout -k 10 200 python -u tools/block_class_time.py 100000 2>/dev/null | grep bits > $OUT/lone_block_classes.txt || { echo "block classes failed"; exit 1; }
timeout -k 10 200 python -u tools/block_class_time.py 7200 2>/dev/null | grep bits > $OUT/lone_block_classes_7200.txt || { echo "block classes 7200 failed"; exit 1; }
timeout -k 10 300 python tools/host_bench.py 2>/dev/null | grep "^{" > $OUT/host_bench.json || { echo "host bench failed"; exit 1; }
{ timeout -k 10 120 tools/ubench/decstore 120 && timeout -k 10 120 tools/ubench/decstore 128 && timeout -k 10 120 tools/ubench/encload 120 && timeout -k 10 120 tools/ubench/encload 128; } > $OUT/memory_patterns.txt 2>&1 || echo "ubench failed"
echo evidence ok
