#!/bin/bash
# Checks of the workgroup-per-block decompressor: parity subset with every block forced through it, then timings
# (plain build) and cycle stamps (the -DGTSC_WG_STAMP build, if present).
mkdir -p gpurun_out
export GTSC_LONG_POINTS=0 GTSC_WG_DECODE_MAX=1000000000
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py -x -q -m gpu -p no:cacheprovider --timeout 120 --timeout-method thread \
  -k "known_answer or reference_test_inputs or ragged_batches or benchmark_shapes or garbage or damaged or quirks or zero_timestamps or high_half or overflow or uniform_layouts" > gpurun_out/wg_parity.log 2>&1
rc=$?
echo "pytest rc=$rc" >> gpurun_out/wg_parity.log
tail -5 gpurun_out/wg_parity.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 170 python -u tools/stress_parity.py 8 9000 > gpurun_out/wg_stress.log 2>&1
rc=$?
tail -3 gpurun_out/wg_stress.log
if [ $rc -ne 0 ]; then exit $rc; fi
unset GTSC_LONG_POINTS GTSC_WG_DECODE_MAX
timeout -k 10 200 python -u tools/block_class_time.py 100000 > gpurun_out/wg_classes.log 2>&1
rc2=$?
cat gpurun_out/wg_classes.log
if [ $rc2 -ne 0 ]; then exit $rc2; fi
if [ -f gorilla-tsc_amd/csrc/libgtsc_stamp.so ]; then
  GTSC_LIB=$PWD/gorilla-tsc_amd/csrc/libgtsc_stamp.so timeout -k 10 200 python -u tools/block_class_time.py 100000 > gpurun_out/wg_stamp.log 2>&1
fi
