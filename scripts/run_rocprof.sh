#!/bin/bash
# Collect the rocprofv3 kernel-trace summary (and, in separate passes, the HBM PMC counters) for
# the default bench.py run.  Run on the GPU box:  bash scripts/run_rocprof.sh <tag>
set -u
TAG=${1:-r01}
cd "$(dirname "$0")/.."
REPO=$PWD
export TMPDIR=/tmp
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python "$REPO/bench.py" --steps 5 --warmup 1 --cpu-sample 0 > "$OUT/bench_trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o pmc -- python "$REPO/bench.py" --steps 2 --warmup 1 --cpu-sample 0 > "$OUT/bench_pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o pmc -- python "$REPO/bench.py" --steps 2 --warmup 1 --cpu-sample 0 > "$OUT/bench_pmc_write.log" 2>&1
cd "$REPO"
find "$OUT" -name "*.csv" | head -20
python scripts/summarize_rocprof.py "$OUT" "$TAG"
