fmt='import json,sys; r=json.loads(sys.stdin.readline()); print(r["compress_GBps_per_gpu"], r["decompress_GBps_per_gpu"], r["roofline"]["encode_ms"], r["roofline"]["decode_ms"], r["config"]["compressed_bits_per_point"])'
for S in 256 1024 4096; do
for wl in adversarial gauge_drift; do
  for mx in 0 1000000; do
    echo "== $wl ${S}x7200 GTSC_LONG_DECODE_MAX=$mx"
    GTSC_LONG_DECODE_MAX=$mx timeout -k 10 120 python bench.py --workload $wl --series $S --points 7200 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "$fmt"
  done
done
done
