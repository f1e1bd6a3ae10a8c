set -e
for wl in gauge random_walk; do
  for thr in 1000000000 1024; do
    echo "== $wl 16668x7200 GTSC_LONG_POINTS=$thr"
    GTSC_LONG_POINTS=$thr timeout -k 10 120 python bench.py --workload $wl --series 16668 --points 7200 --steps 10 --warmup 3 --no-cpu-baseline | python -c "import json,sys; r=json.loads(sys.stdin.readline()); print(r['compress_GBps_per_gpu'], r['decompress_GBps_per_gpu'], r['roofline']['encode_ms'], r['roofline']['decode_ms'], r['config']['compressed_bits_per_point'])"
  done
done
for wl in gauge random_walk adversarial; do
  for thr in 1000000000 1024; do
    echo "== $wl 65536x7200 GTSC_LONG_POINTS=$thr"
    GTSC_LONG_POINTS=$thr timeout -k 10 120 python bench.py --workload $wl --series 65536 --points 7200 --steps 5 --warmup 2 --no-cpu-baseline | python -c "import json,sys; r=json.loads(sys.stdin.readline()); print(r['compress_GBps_per_gpu'], r['decompress_GBps_per_gpu'], r['roofline']['encode_ms'], r['roofline']['decode_ms'], r['config']['compressed_bits_per_point'])"
  done
done
GTSC_LONG_POINTS=1024 timeout -k 10 120 python tools/long_block_time.py
