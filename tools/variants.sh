# dense decode timings (16-word ring forced) for every library build under build/: bash tools/variants.sh
fmt='import json,sys; r=json.loads(sys.stdin.readline()); print(r["decompress_GBps_per_gpu"], r["roofline"]["decode_ms"])'
for lib in ${LIBS:-build/*.so}; do
for cfg in "gauge_drift 1000000 120" "random_walk 131072 7200" "adversarial 65536 7200" "random_walk 1000000 120"; do
  set -- $cfg
  echo "== $lib $1 $2x$3"
  GTSC_RING=16 GTSC_LIB=$PWD/$lib timeout -k 10 200 python bench.py --workload $1 --series $2 --points $3 --steps 8 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "$fmt" || exit 1
done
done
