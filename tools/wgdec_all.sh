#!/bin/bash
# The three decompressors side by side on batches of equal blocks (tools/wgdec_cmp.sh), written line by line.
mkdir -p gpurun_out
out=gpurun_out/three_decompressors.txt
: > $out
WL="counter gauge random_walk readme" bash tools/wgdec_cmp.sh 7200 64 256 1024 4096 16384 >> $out 2>&1 || exit 1
WL="counter readme gauge" bash tools/wgdec_cmp.sh 100000 64 256 512 1024 2048 >> $out 2>&1 || exit 1
cat $out
