#!/bin/bash
# rocprofv3 evidence of a round for every bench workload (run ON the GPU box):
#   bash scripts/profile_bench.sh r06 [workloads...]
# Per workload: one --kernel-trace --stats run and, in SEPARATE runs (never combined with a trace domain), the --pmc groups
#   FETCH_SIZE | WRITE_SIZE | fp64 operation counts | two SQ occupancy / stall groups.
# scripts/summarize_profile.py condenses them into gpurun_out/profile_<tag>/ (copy to profiles/<tag>/) and writes the two files
# bench.py reads: pmc_traffic.json (HBM bytes per launch) and pmc_flops.json (fp64 operations per launch, counted by the hardware).
set -u
TAG=${1:-r06}; shift || true
WLS=${*:-"pgdb lean8192 lean65536 sweep sweep3 pgdb3 pgdb3pauli pgdb1 mle_state mle_state3 shots"}
cd "$(dirname "$0")/.."
REPO=$PWD
export TMPDIR=/tmp
OUT=$REPO/gpurun_out/prof_$TAG
DST=$REPO/gpurun_out/profile_$TAG
mkdir -p "$OUT" "$DST"
bench_args() {
    case $1 in
        pgdb)       echo "--workload pgdb" ;;
        lean8192)   echo "--workload pgdb --batch 8192" ;;
        lean65536)  echo "--workload pgdb --batch 65536" ;;
        sweep)      echo "--workload sweep" ;;
        sweep3)     echo "--workload sweep3" ;;
        pgdb3)      echo "--workload pgdb3" ;;
        pgdb3pauli) echo "--workload pgdb3 --in-basis pauli" ;;
        pgdb1)      echo "--workload pgdb1" ;;
        mle_state)  echo "--workload mle_state" ;;
        mle_state3) echo "--workload mle_state3" ;;
        shots)      echo "--workload shots" ;;
    esac
}
cd /tmp
for wl in $WLS; do
    ARGS="$(bench_args $wl) --cpu-sample 0"
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$wl" -o trace -- python $REPO/bench.py $ARGS --steps 5 --warmup 1 > "$OUT/bench_trace_$wl.log" 2>&1
    n=0
    for grp in "FETCH_SIZE" "WRITE_SIZE" \
               "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU SQ_WAVES SQ_INSTS_VALU_MFMA_F64" \
               "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
               "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU"; do
        n=$((n + 1))
        rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc${n}_$wl" -o pmc -- python $REPO/bench.py $ARGS --steps 2 --warmup 1 > "$OUT/bench_pmc${n}_$wl.log" 2>&1
    done
done
cd "$REPO"
python scripts/summarize_profile.py "$OUT" "$DST" "$TAG" $WLS
