#!/bin/bash
# The host entry points over and over (pageable and pinned memory, few and many chunks); stops at the first failure.
mkdir -p gpurun_out
: > gpurun_out/host_soak.txt
for i in 1 2 3 4 5 6; do
  GTSC_CHUNK_POINTS=1000000 timeout -k 5 120 python -u tests/host_chunks_worker.py skewed_many >> gpurun_out/host_soak.txt 2>&1 || { echo "skewed_many failed in round $i"; exit 1; }
  GTSC_CHUNK_POINTS=300000 timeout -k 5 120 python -u tests/host_chunks_worker.py skewed >> gpurun_out/host_soak.txt 2>&1 || { echo "skewed failed in round $i"; exit 1; }
  GTSC_CHUNK_POINTS=3000 timeout -k 5 120 python -u tests/host_chunks_worker.py >> gpurun_out/host_soak.txt 2>&1 || { echo "ragged failed in round $i"; exit 1; }
  timeout -k 5 120 python -u tools/host_bench.py skewed 26 2>/dev/null | grep "^{" | cut -c100-260 >> gpurun_out/host_soak.txt || { echo "host_bench failed in round $i"; exit 1; }
  echo "round $i ok"
done
