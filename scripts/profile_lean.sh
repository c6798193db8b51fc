#!/bin/bash
# rocprofv3 evidence for BASELINE configs[4]'s per-GPU share: 8192 two-qubit reconstructions on the two-waves-per-SIMD kernel
# (run ON the GPU box):   bash scripts/profile_lean.sh r03     ->  gpurun_out/profile_<tag>/lean8192_*
set -u
TAG=${1:-r03}
cd "$(dirname "$0")/.."
REPO=$PWD
export TMPDIR=/tmp
OUT=$REPO/gpurun_out/prof_lean_$TAG
DST=$REPO/gpurun_out/profile_$TAG
rm -rf "$OUT"; mkdir -p "$OUT" "$DST"
BENCH="python $REPO/bench.py --cpu-sample 0 --workload pgdb --batch 8192"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- $BENCH --steps 5 --warmup 1 > "$OUT/bench_trace.log" 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU" \
           "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_FLAT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
    name=$(echo $grp | cut -d' ' -f1)
    rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc_$name" -o pmc -- $BENCH --steps 2 --warmup 1 > "$OUT/bench_pmc_$name.log" 2>&1
done
cd "$REPO"
python - "$OUT" "$DST" <<'PY'
import csv, glob, json, os, sys
out, dst = sys.argv[1], sys.argv[2]
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    open(os.path.join(dst, "lean8192_rocprofv3_kernel_stats.csv"), "w").write(open(f).read())
js = [l for l in open(os.path.join(out, "bench_trace.log")) if l.startswith("{")]
if js:
    open(os.path.join(dst, "bench_lean8192_line_under_rocprof.json"), "w").write(js[-1])
pmc = {}
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "pgdb_lean_kernel" in r.get("Kernel_Name", ""):
            pmc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
e = {c: {"mean": sum(v) / len(v), "launches": len(v)} for c, v in pmc.items()}
if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
    e["hbm_bytes_per_launch"] = e["FETCH_SIZE"]["mean"] * 1024 * 2 + e["WRITE_SIZE"]["mean"] * 1024
g = lambda c: e[c]["mean"] if c in e else None
if g("SQ_WAVE_CYCLES"):
    for k, c in (("valu_active_over_wave_cycles", "SQ_ACTIVE_INST_VALU"), ("lds_active_over_wave_cycles", "SQ_ACTIVE_INST_LDS"),
                 ("wait_any_over_wave_cycles", "SQ_WAIT_ANY"), ("wait_inst_over_wave_cycles", "SQ_WAIT_INST_ANY"),
                 ("vmem_active_over_wave_cycles", "SQ_ACTIVE_INST_VMEM")):
        if g(c) is not None:
            e[k] = g(c) / g("SQ_WAVE_CYCLES")
json.dump({"note": "pgdb_lean_kernel<2,9>, 8192 two-qubit reconstructions, fixed 100 iterations: means per launch, separate --pmc passes",
           "counters": e}, open(os.path.join(dst, "lean8192_pmc_counters.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in e.items() if not isinstance(v, dict)}, indent=1))
PY
