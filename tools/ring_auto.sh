fmt='import json,sys; r=json.loads(sys.stdin.readline()); print(r["compress_GBps_per_gpu"], r["decompress_GBps_per_gpu"], r["roofline"]["encode_ms"], r["roofline"]["decode_ms"])'
for cfg in "gauge_drift 1000000 120" "gauge_drift 65536 7200" "random_walk 262144 7200" "random_walk 131072 1024" "random_walk 300000 1024" "random_walk 524288 7200"; do
  set -- $cfg
  echo "== $1 $2x$3"
  timeout -k 10 300 python bench.py --workload $1 --series $2 --points $3 --steps 5 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "$fmt" || exit 1
done
