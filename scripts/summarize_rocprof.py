"""Condense rocprofv3 output (kernel stats + PMC passes) into small text/JSON summaries."""
import csv, glob, json, os, sys
out, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "gpurun_out", f"profile_{tag}")
os.makedirs(dst, exist_ok=True)
lines = []
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    lines.append(f"# {os.path.relpath(f, out)}")
    lines += [l.rstrip() for l in open(f)]
trace = glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True)
if trace:
    rows = list(csv.DictReader(open(trace[0])))
    byk = {}
    for r in rows:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        byk.setdefault(r["Kernel_Name"], []).append((d, r))
    lines.append("# per-kernel dispatch summary from kernel_trace.csv")
    for k, v in byk.items():
        ds = [x[0] for x in v]
        r = v[0][1]
        lines.append(f"{k[:100]}: calls={len(ds)} avg_ms={sum(ds)/len(ds)/1e6:.3f} min_ms={min(ds)/1e6:.3f} "
                     f"max_ms={max(ds)/1e6:.3f} grid={r.get('Grid_Size','?')} wg={r.get('Workgroup_Size','?')} "
                     f"lds={r.get('LDS_Block_Size','?')} vgpr={r.get('VGPR_Count','?')} accum_vgpr={r.get('Accum_VGPR_Count','?')} sgpr={r.get('SGPR_Count','?')}")
open(os.path.join(dst, "kernel_stats.txt"), "w").write("\n".join(lines) + "\n")
pmc = {}
for name in ("fetch", "write"):
    files = glob.glob(os.path.join(out, f"pmc_{name}", "**", "*counter_collection.csv"), recursive=True)
    if not files:
        continue
    rows = list(csv.DictReader(open(files[0])))
    acc = {}
    for r in rows:
        if "pgdb_kernel" not in r.get("Kernel_Name", ""):
            continue
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for c, v in acc.items():
        pmc[c] = {"per_launch_mean": sum(v) / len(v), "launches": len(v)}
summary = {"tag": tag, "counters": pmc}
# MI355X_MICROARCH.md, HBM section: FETCH_SIZE / WRITE_SIZE are in KiB on rocprofv3's derived
# counters; FETCH_SIZE reads half the bytes of a wide coalesced stream on gfx950 -> doubled.
if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
    fetch = pmc["FETCH_SIZE"]["per_launch_mean"] * 1024 * 2
    write = pmc["WRITE_SIZE"]["per_launch_mean"] * 1024
    summary["pgdb_kernel_hbm_bytes_per_launch"] = fetch + write
    summary["note"] = "FETCH_SIZE(KiB) x 1024 x 2 (gfx950 half-count correction) + WRITE_SIZE(KiB) x 1024"
json.dump(summary, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
print(open(os.path.join(dst, "kernel_stats.txt")).read()[:3000])
print(json.dumps(summary, indent=1))
