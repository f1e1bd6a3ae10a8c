#!/bin/bash
# A/B of library variants on the GPU box: tools/ab.sh "<bench args>" lib1.so lib2.so ...
ARGS=$1; shift
for lib in "$@"; do
  echo "== $lib"
  GTSC_LIB=$PWD/$lib timeout -k 10 200 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-verify $ARGS 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('enc_ms',r['encode_ms'],'dec_ms',r['decode_ms'],'enc GB/s',d['compress_GBps_per_gpu'],'dec GB/s',d['decompress_GBps_per_gpu'])" || echo failed
done
