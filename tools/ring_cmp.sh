# the decoder's two ring forms (GTSC_RING=8|16) on dense workloads of short and long series
fmt='import json,sys; r=json.loads(sys.stdin.readline()); print(r["compress_GBps_per_gpu"], r["decompress_GBps_per_gpu"], r["roofline"]["encode_ms"], r["roofline"]["decode_ms"])'
for cfg in "gauge_drift 1000000 120" "random_walk 1000000 120" "adversarial 1000000 120" "gauge_drift 65536 7200" "random_walk 262144 7200" "adversarial 65536 7200" "random_walk 131072 1024"; do
for ring in 8 16; do
  set -- $cfg
  echo "== RING=$ring $1 $2x$3"
  GTSC_RING=$ring timeout -k 10 200 python bench.py --workload $1 --series $2 --points $3 --steps 6 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "$fmt" || exit 1
done
done
