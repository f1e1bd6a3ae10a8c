# dense decode timings of the lane-per-series decoder with NP parser waves per writer wave (GTSC_DEC_NP; 0 = no writer wave)
fmt='import json,sys; r=json.loads(sys.stdin.readline()); print(r["compress_GBps_per_gpu"], r["decompress_GBps_per_gpu"], r["roofline"]["encode_ms"], r["roofline"]["decode_ms"])'
for np in ${NPS:-0 2 3 7}; do
for cfg in "gauge_drift 1000000 120" "random_walk 131072 7200" "adversarial 65536 7200"; do
  set -- $cfg
  echo "== NP=$np $1 $2x$3"
  GTSC_DEC_NP=$np timeout -k 10 200 python bench.py --workload $1 --series $2 --points $3 --steps 10 --warmup 4 --no-cpu-baseline --no-extras 2>/dev/null | python -c "$fmt" || exit 1
done
done
