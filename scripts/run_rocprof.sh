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
# secondary workload: the HBM-bound conversion sweep
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_sweep" -o trace -- python "$REPO/bench.py" --workload sweep --steps 5 --warmup 1 --cpu-sample 0 > "$OUT/bench_sweep_trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_sweep_fetch" -o pmc -- python "$REPO/bench.py" --workload sweep --steps 2 --warmup 1 --cpu-sample 0 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_sweep_write" -o pmc -- python "$REPO/bench.py" --workload sweep --steps 2 --warmup 1 --cpu-sample 0 > /dev/null 2>&1
cd "$REPO"
python - "$OUT" "$TAG" <<'PY'
import csv, glob, json, os, sys
out, tag = sys.argv[1], sys.argv[2]
dst = os.path.join("gpurun_out", f"profile_{tag}")
res = {}
for name in ("fetch", "write"):
    f = glob.glob(os.path.join(out, f"pmc_sweep_{name}", "**", "*counter_collection.csv"), recursive=True)
    if not f: continue
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if "sweep2q" in r.get("Kernel_Name", "")]
    if vals: res[name.upper() + "_SIZE_KiB_per_launch"] = sum(vals) / len(vals)
if len(res) == 2:
    res["sweep_kernel_hbm_bytes_per_launch"] = res["FETCH_SIZE_KiB_per_launch"] * 1024 * 2 + res["WRITE_SIZE_KiB_per_launch"] * 1024
json.dump(res, open(os.path.join(dst, "pmc_traffic_sweep.json"), "w"), indent=1)
st = glob.glob(os.path.join(out, "trace_sweep", "**", "*kernel_stats.csv"), recursive=True)
if st: open(os.path.join(dst, "sweep_kernel_stats.csv"), "w").write(open(st[0]).read())
print(json.dumps(res, indent=1)); print(open(os.path.join(dst, "sweep_kernel_stats.csv")).read() if st else "")
PY
grep -h '"metric"' "$OUT/bench_sweep_trace.log" "$OUT/bench_trace.log"
# third workload: 3-qubit PGDB (BASELINE configs[3])
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_pgdb3" -o trace -- python "$REPO/bench.py" --workload pgdb3 --in-basis sic --steps 3 --warmup 1 > "$OUT/bench_pgdb3_trace.log" 2>&1
cd "$REPO"
ST3=$(find "$OUT/trace_pgdb3" -name "*kernel_stats.csv" | head -1)
[ -n "$ST3" ] && cp "$ST3" "gpurun_out/profile_$TAG/pgdb3_kernel_stats.csv" && cat "$ST3"
grep -h '"metric"' "$OUT/bench_pgdb3_trace.log" | cut -c1-300
