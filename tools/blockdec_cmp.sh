# decode ms of S x 7,200 batches through the lane-per-series kernel (GTSC_LONG_DECODE_MAX=0) and the
# wave-per-block kernel (=1000000): bash tools/blockdec_cmp.sh
fmt='import json,sys; r=json.loads(sys.stdin.readline()); print(r["decompress_GBps_per_gpu"], r["roofline"]["decode_ms"])'
for S in 1024 4096 12288 16668; do
for wl in gauge random_walk gauge_drift adversarial; do
for mx in 0 1000000; do
  echo "== $wl ${S}x7200 LONG_DECODE_MAX=$mx"
  GTSC_LONG_DECODE_MAX=$mx timeout -k 10 200 python bench.py --workload $wl --series $S --points 7200 --steps 8 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "$fmt" || exit 1
done
done
done
