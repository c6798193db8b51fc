#!/bin/bash
# rocprofv3 evidence for one round, all four bench workloads (run ON the GPU box):
#   bash scripts/profile_round.sh r02
# Pass 1: --kernel-trace --stats of `python bench.py --workload <pgdb|sweep|pgdb3|pgdb1> --cpu-sample 0 --steps 5 --warmup 1`.
# Passes 2..: --pmc only (never combined with a trace domain), one counter group per pass:
#   FETCH_SIZE | WRITE_SIZE | SQ group A | SQ group B.
# Summaries land in gpurun_out/profile_<tag>/ (copy what should be judged into profiles/<tag>/).
set -u
TAG=${1:-r02}
cd "$(dirname "$0")/.."
REPO=$PWD
export TMPDIR=/tmp
OUT=$REPO/gpurun_out/prof_$TAG
DST=$REPO/gpurun_out/profile_$TAG
rm -rf "$OUT"; mkdir -p "$OUT" "$DST"
BENCH="python $REPO/bench.py --cpu-sample 0"
cd /tmp
rocprofv3 -L > "$OUT/counters_available.txt" 2>&1
# one trace run per workload, so that a kernel's average duration is over the bench launches only
for wl in pgdb sweep pgdb3 pgdb1; do
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$wl" -o trace -- $BENCH --workload $wl --steps 5 --warmup 1 > "$OUT/bench_trace_$wl.log" 2>&1
done
pass() {   # name, counters...: one run per workload so that a kernel's mean is over identical launches
    local name=$1; shift
    for wl in pgdb sweep pgdb3 pgdb1; do
        rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_${name}_$wl" -o pmc -- $BENCH --workload $wl --steps 2 --warmup 1 > "$OUT/bench_pmc_${name}_$wl.log" 2>&1
    done
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sqa SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS
pass sqb SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU
pass sqc SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_MFMA SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_FLAT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC
cd "$REPO"
python scripts/summarize_round.py "$OUT" "$DST" "$TAG"
