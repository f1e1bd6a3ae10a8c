fmt='import json,sys; r=json.loads(sys.stdin.readline()); print(r["compress_GBps_per_gpu"], r["decompress_GBps_per_gpu"], r["roofline"]["encode_ms"], r["roofline"]["decode_ms"])'
for cfg in "gauge 16668 7200" "gauge_drift 16668 7200" "random_walk 16668 7200" "gauge 8192 7200" "random_walk 8192 7200"; do
for mx in 4096 1000000; do
  set -- $cfg
  echo "== LONG_DECODE_MAX=$mx $1 $2x$3"
  GTSC_LONG_DECODE_MAX=$mx timeout -k 10 200 python bench.py --workload $1 --series $2 --points $3 --steps 10 --warmup 4 --no-cpu-baseline --no-extras 2>/dev/null | python -c "$fmt" || exit 1
done
done
