# decode ms of S x P batches of long blocks through the three decompressors: lane-per-series ("tiled"), wave-per-block
# ("wave"), workgroup-per-block ("wg"): [WL="workloads"] bash tools/wgdec_cmp.sh [P] [S...]
P=${1:-7200}; shift
SS=${*:-16 128 512 2048 8192}
fmt='import json,sys; r=json.loads(sys.stdin.readline()); print("%8.3f ms" % r["roofline"]["decode_ms"])'
for S in $SS; do
for wl in ${WL:-gauge counter random_walk gauge_drift adversarial}; do
  line="$wl ${S}x$P:"
  for k in tiled wave wg; do
    case $k in
      tiled) env="GTSC_WG_DECODE_MAX=0 GTSC_WG_DECODE_VLONG_MAX=0 GTSC_LONG_DECODE_MAX=0 GTSC_LONG_DECODE_MAX_DENSE=0";;
      wave)  env="GTSC_WG_DECODE_MAX=0 GTSC_WG_DECODE_VLONG_MAX=0 GTSC_LONG_DECODE_MAX=1000000 GTSC_LONG_DECODE_MAX_DENSE=1000000";;
      wg)    env="GTSC_WG_DECODE_MAX=1000000 GTSC_WG_DENSE_BITS=1000";;
    esac
    r=$(env $env timeout -k 10 200 python bench.py --workload $wl --series $S --points $P --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "$fmt") || exit 1
    line="$line  $k $r"
  done
  echo "$line"
done
done
