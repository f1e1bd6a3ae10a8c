# decode timings over the sparse and dense workloads (default ring choice): bash tools/dec_all.sh
fmt='import json,sys; r=json.loads(sys.stdin.readline()); print(r["compress_GBps_per_gpu"], r["decompress_GBps_per_gpu"], r["roofline"]["encode_ms"], r["roofline"]["decode_ms"])'
for cfg in "gauge 1000000 120" "gauge_drift 1000000 120" "random_walk 1000000 120" "gauge 65536 7200" "random_walk 262144 7200" "adversarial 65536 7200"; do
  set -- $cfg
  echo "== $1 $2x$3"
  timeout -k 10 200 python bench.py --workload $1 --series $2 --points $3 --steps 10 --warmup 4 --no-cpu-baseline --no-extras 2>/dev/null | python -c "$fmt" || exit 1
done
