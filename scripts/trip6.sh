#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python scripts/compare_libs.py libfbx_before.so libfbx.so > gpurun_out/t6_compare.log 2>&1
for wl in mle_state mle_state3; do
  for lib in libfbx_before.so libfbx.so libfbx_before.so libfbx.so; do
    FBX_LIBRARY=$PWD/forest-benchmarking_amd/$lib python bench.py --workload $wl --steps 5 --warmup 1 --cpu-sample 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl $lib', d['value'], d['ms_per_step'])" >> gpurun_out/t6_mle_ab.log 2>&1
  done
done
python -m pytest tests -m gpu -q > gpurun_out/t6_tests.log 2>&1
tail -3 gpurun_out/t6_compare.log | cut -c1-200; cat gpurun_out/t6_mle_ab.log; tail -4 gpurun_out/t6_tests.log
