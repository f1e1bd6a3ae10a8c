# decode: lane-per-series vs wave-per-block on batches of few long blocks; host-to-host rates
set -e
fmt='import json,sys; r=json.loads(sys.stdin.readline()); print(r["compress_GBps_per_gpu"], r["decompress_GBps_per_gpu"], r["roofline"]["encode_ms"], r["roofline"]["decode_ms"], r["config"]["compressed_bits_per_point"])'
for S in 1024 4096 16668; do
for wl in gauge random_walk; do
  for mx in 0 1000000; do
    echo "== $wl ${S}x7200 GTSC_LONG_DECODE_MAX=$mx"
    GTSC_LONG_DECODE_MAX=$mx timeout -k 10 120 python bench.py --workload $wl --series $S --points 7200 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "$fmt"
  done
done
done
echo "== host_bench skewed 2^26"; timeout -k 10 200 python tools/host_bench.py skewed 26 2>/dev/null
echo "== host_bench gauge 2^26"; timeout -k 10 200 python tools/host_bench.py gauge 26 2>/dev/null
