#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
bash scripts/profile_bench.sh r06 > gpurun_out/t7_prof.log 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/t7_bench_default.log 2> gpurun_out/t7_bench_default.err
python bench.py > gpurun_out/t7_bench_noflags.log 2> gpurun_out/t7_bench_noflags.err
python scripts/state_time.py > gpurun_out/t7_state_time.log 2>&1
python scripts/shots_time.py > gpurun_out/t7_shots_time.log 2>&1
python scripts/bootstrap_time.py > gpurun_out/t7_bootstrap_time.log 2>&1
tail -1 gpurun_out/t7_bench_default.log | cut -c1-300
