"""Condense the rocprofv3 output of scripts/profile_round.sh: kernel-trace stats, per-kernel dispatch
summary, and the PMC passes as per-kernel means per launch (JSON).  HBM bytes follow the guide's gfx950
correction: FETCH_SIZE (KiB) x 1024 x 2 + WRITE_SIZE (KiB) x 1024."""
import csv, glob, json, os, sys
out, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
os.makedirs(dst, exist_ok=True)
KERNELS = {"pgdb_kernel": "pgdb", "pgdb3_kernel": "pgdb3", "sweep2q_pair_kernel": "sweep", "random_kraus_kernel": "random_kraus"}


def short(name):
    for k, v in KERNELS.items():
        if k in name:
            return v
    return None


lines = []
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    lines.append(f"# {os.path.relpath(f, out)}")
    lines += [l.rstrip() for l in open(f)]
    open(os.path.join(dst, "rocprofv3_kernel_stats.csv"), "w").write(open(f).read())
trace = glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True)
if trace:
    byk = {}
    for r in csv.DictReader(open(trace[0])):
        byk.setdefault(r["Kernel_Name"], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r))
    lines.append("# per-kernel dispatch summary from kernel_trace.csv")
    for k, v in byk.items():
        ds = [x[0] for x in v]
        r = v[0][1]
        lines.append(f"{k[:110]}: calls={len(ds)} avg_ms={sum(ds)/len(ds)/1e6:.3f} min_ms={min(ds)/1e6:.3f} "
                     f"max_ms={max(ds)/1e6:.3f} grid={r.get('Grid_Size','?')} wg={r.get('Workgroup_Size','?')} "
                     f"lds={r.get('LDS_Block_Size','?')} vgpr={r.get('VGPR_Count','?')} accum_vgpr={r.get('Accum_VGPR_Count','?')} "
                     f"sgpr={r.get('SGPR_Count','?')} scratch={r.get('Scratch_Size', r.get('Private_Segment_Size','?'))}")
open(os.path.join(dst, "kernel_stats.txt"), "w").write("\n".join(lines) + "\n")

pmc = {}
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r.get("Kernel_Name", ""))
            if k is None:
                continue
            pmc.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
summary = {"tag": tag, "note": "means per kernel launch; separate --pmc passes (never combined with a trace domain); "
                               "bench.py --workload <pgdb|sweep|pgdb3> --steps 2 --warmup 1 --cpu-sample 0, i.e. every launch of a kernel "
                               "in a pass is the bench launch (pgdb: B = 1024, fixed 100 iterations)",
           "kernels": {}}
for k, cs in pmc.items():
    e = {c: {"mean": sum(v) / len(v), "launches": len(v)} for c, v in cs.items()}
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        e["hbm_bytes_per_launch"] = e["FETCH_SIZE"]["mean"] * 1024 * 2 + e["WRITE_SIZE"]["mean"] * 1024
    g = lambda c: e[c]["mean"] if c in e else None
    if g("SQ_BUSY_CYCLES") and g("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
        e["mfma_busy_over_sq_busy"] = g("SQ_VALU_MFMA_BUSY_CYCLES") / g("SQ_BUSY_CYCLES")
    if g("SQ_WAVE_CYCLES") and g("SQ_ACTIVE_INST_VALU") is not None:
        e["valu_active_over_wave_cycles"] = g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES")
    if g("SQ_LDS_IDX_ACTIVE") and g("SQ_LDS_BANK_CONFLICT") is not None:
        e["lds_bank_conflict_over_lds_active"] = g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")
    summary["kernels"][k] = e
json.dump(summary, open(os.path.join(dst, "pmc_counters.json"), "w"), indent=1)
for name in ("bench_trace.log",):
    p = os.path.join(out, name)
    if os.path.exists(p):
        js = [l for l in open(p) if l.startswith("{")]
        if js:
            open(os.path.join(dst, "bench_line_under_rocprof.json"), "w").write(js[-1])
print(open(os.path.join(dst, "kernel_stats.txt")).read()[:4000])
print(json.dumps(summary, indent=1)[:6000])
