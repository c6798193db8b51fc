#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python scripts/ab_many3.py libfbx_before.so libfbx.so libfbx_lad8.so libfbx_lad2.so --pauli > gpurun_out/t3_ab_pauli.log 2>&1
python scripts/ab_many3.py libfbx_before.so libfbx.so libfbx_lad8.so libfbx_lad2.so > gpurun_out/t3_ab_sic.log 2>&1
python scripts/compare_libs.py libfbx_before.so libfbx.so > gpurun_out/t3_compare.log 2>&1
python -m pytest tests/test_timed_mode_goldens.py tests/test_pgdb3_gpu.py tests/test_repeated_datasets.py -m gpu -x -q > gpurun_out/t3_tests.log 2>&1
cat gpurun_out/t3_ab_pauli.log gpurun_out/t3_ab_sic.log; tail -3 gpurun_out/t3_compare.log; tail -3 gpurun_out/t3_tests.log
