#!/bin/bash
# round 6: the whole GPU suite, then the rocprofv3 evidence of every bench workload with the round's library, then the default bench run
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/t5_tests.log 2>&1
bash scripts/profile_bench.sh r06 > gpurun_out/t5_prof.log 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/t5_bench_default.log 2> gpurun_out/t5_bench_default.err
tail -3 gpurun_out/t5_tests.log; tail -1 gpurun_out/t5_bench_default.log | cut -c1-400
