#!/bin/bash
# round 6, trip 2: the pruned sources against the pre-pruning build (bit for bit + same-box timing), then the whole GPU suite
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python scripts/compare_libs.py libfbx_before.so libfbx.so > gpurun_out/t2_compare.log 2>&1
python scripts/ab_time.py libfbx_before.so libfbx.so 1024 fixed > gpurun_out/t2_ab.log 2>&1
python scripts/ab_time.py libfbx_before.so libfbx.so 1024 converge >> gpurun_out/t2_ab.log 2>&1
python scripts/ab_time.py libfbx_before.so libfbx.so 8192 fixed >> gpurun_out/t2_ab.log 2>&1
AB_NQ=3 AB_BASIS=sic python scripts/ab_time.py libfbx_before.so libfbx.so 256 fixed >> gpurun_out/t2_ab.log 2>&1
python -m pytest tests -m gpu -x -q > gpurun_out/t2_tests.log 2>&1
cat gpurun_out/t2_compare.log gpurun_out/t2_ab.log; tail -5 gpurun_out/t2_tests.log
