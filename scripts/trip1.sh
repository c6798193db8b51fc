#!/bin/bash
# round 6, first GPU trip: new tests, the state-MLE workload and its profile, 3-qubit phase split, the default bench run
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_cost_grad_gpu.py tests/test_multirank_gpu.py -m gpu -x -q > gpurun_out/t1_tests.log 2>&1
python bench.py --workload mle_state --steps 5 --warmup 1 > gpurun_out/t1_mle2.log 2>&1
python bench.py --workload mle_state3 --steps 5 --warmup 1 > gpurun_out/t1_mle3.log 2>&1
bash scripts/profile_bench.sh r06a mle_state mle_state3 > gpurun_out/t1_prof.log 2>&1
python scripts/phase_profile3.py sic > gpurun_out/t1_phase3_sic.log 2>&1
python scripts/phase_profile3.py pauli > gpurun_out/t1_phase3_pauli.log 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/t1_bench_default.log 2> gpurun_out/t1_bench_default.err
tail -c 300 gpurun_out/t1_tests.log
