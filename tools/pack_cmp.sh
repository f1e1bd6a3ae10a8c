# narrow decoder waves for small batches (GTSC_DECODE_LANES_LOG2: 6 = always 64 series per wave): bash tools/pack_cmp.sh
fmt='import json,sys; r=json.loads(sys.stdin.readline()); print(r["compress_GBps_per_gpu"], r["decompress_GBps_per_gpu"], r["roofline"]["encode_ms"], r["roofline"]["decode_ms"])'
for cfg in "gauge 16668 7200" "gauge 32768 7200" "random_walk 16668 7200" "gauge 8192 120" "gauge 40000 120"; do
for l in 6 0; do
  set -- $cfg
  echo "== LANES_LOG2=$l $1 $2x$3"
  GTSC_DECODE_LANES_LOG2=$l timeout -k 10 200 python bench.py --workload $1 --series $2 --points $3 --steps 10 --warmup 4 --no-cpu-baseline --no-extras 2>/dev/null | python -c "$fmt" || exit 1
done
done
