#!/bin/bash
# Runs on the GPU box: tools/ldsstat.sh for the full library and for the decoder builds without its per-point tile writes
# (build/ablate5.so), without its per-point ring window reads (ablate6.so) and without both (ablate7.so), on the
# default workload and on a dense one.  usage: tools/ldsstat_all.sh > gpurun_out/ldsstat.txt
for wl in "gauge 1000000 120" "random_walk 65536 7200"; do
  set -- $wl
  echo "#### $1 $2 x $3"
  echo "== full kernels"
  bash tools/ldsstat.sh full_$1 --workload $1 --series $2 --points $3 | grep -A9 decode_tiled
  for e in 5 6 7; do
    echo "== GTSC_ABLATE=$e"
    GTSC_LIB=$PWD/build/ablate$e.so bash tools/ldsstat.sh a${e}_$1 --workload $1 --series $2 --points $3 | grep -A9 decode_tiled
  done
done
