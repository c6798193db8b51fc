"""Condense the rocprofv3 output of scripts/profile_bench.sh.

    python scripts/summarize_profile.py <raw dir> <summary dir> <tag> <workloads...>

Per workload: the kernel-trace stats CSV, the bench line printed under rocprofv3, and the --pmc passes as means per launch of the
workload's dominant kernel.  HBM bytes follow the guide's gfx950 correction (FETCH_SIZE counts 64-byte requests as 32: KiB x 1024
x 2; WRITE_SIZE KiB x 1024).  fp64 operations per launch = (ADD + MUL + TRANS + 2 FMA) wave-instructions x 64 lanes + MFMA_MOPS x
512 (the counter's unit).  Writes <summary dir>/{pmc_counters.json, pmc_traffic.json, pmc_flops.json, kernel_stats.txt}."""
import csv
import glob
import json
import os
import sys

out, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
wls = sys.argv[4:]
os.makedirs(dst, exist_ok=True)
# workload -> (substring of the dominant kernel's name, key in pmc_traffic.json, items per launch)
WL = {"pgdb": ("pgdb_kernel<2, 9>", "pgdb_kernel_hbm_bytes_per_launch", 1024, "pgdb_kernel<2,9>"),
      "lean8192": ("pgdb_lean_pieces_kernel<2, 9,", "pgdb_lean8192_hbm_bytes_per_launch", 8192, "pgdb_lean_pieces_kernel<2,9>"),
      "lean65536": ("pgdb_lean_pieces_kernel<2, 9,", "pgdb_lean65536_hbm_bytes_per_launch", 65536, "pgdb_lean_pieces_kernel<2,9>"),
      "sweep": ("sweep2q_pair_kernel", "sweep_kernel_hbm_bytes_per_launch", 1000000, "sweep2q_pair_kernel"),
      "sweep3": ("sweep3_regs_kernel", "sweep3_kernel_hbm_bytes_per_launch", 65536, "sweep3_regs_kernel"),
      "pgdb3": ("pgdb3_kernel<4>", "pgdb3_kernel_hbm_bytes_per_launch", 256, "pgdb3_kernel<4>"),
      "pgdb3pauli": ("pgdb3_kernel<14>", "pgdb3pauli_kernel_hbm_bytes_per_launch", 256, "pgdb3_kernel<14>"),
      "pgdb1": ("pgdb1_step_kernel", "pgdb1_kernel_hbm_bytes_per_launch", 1 << 20, "pgdb1_step_kernel"),
      "mle_state": ("mle_state_packed_kernel<2>", "mle_state2_kernel_hbm_bytes_per_launch", 1 << 20, "mle_state_packed_kernel<2>"),
      "mle_state3": ("mle_state_plain3_kernel", "mle_state3_kernel_hbm_bytes_per_launch", 1 << 18, "mle_state_plain3_kernel"),
      "shots": ("shots_pipe", "shots_kernel_hbm_bytes_per_launch", 1024 * 540, "shots_pipe_kernel<2,2>")}
# workloads whose bench step is MANY launches of the kernel (one per outer iteration): counters are summed over a step's launches.
# The --pmc passes run bench.py with --steps 2 --warmup 1 = 3 calls.
PER_CALL = {"pgdb1": 3}

FP64_PEAK, CUS, CLOCK_HZ = 78.6e12, 256, 2.4e9     # MI355X_MICROARCH.md: fp64 vector = fp64 MFMA dense peak, 256 CUs, 2.4 GHz
kernel_s = {}                                       # workload -> mean duration of its dominant kernel (kernel trace), seconds
lines, summary, traffic, flops = [], {"tag": tag, "kernels": {}}, {"tag": tag}, {"tag": tag}
summary["note"] = ("means per launch of each workload's dominant kernel (pgdb1: sums over the launches of one call); separate --pmc passes (never combined with a trace domain); "
                   "bench.py <workload> --steps 2 --warmup 1 --cpu-sample 0")
traffic["note"] = ("FETCH_SIZE(KiB) x 1024 x 2 (gfx950 half-count correction, MI355X_MICROARCH.md HBM section) + WRITE_SIZE(KiB) x 1024; "
                   "separate --pmc passes; per launch of the bench workload")
flops["note"] = ("fp64 operations per launch counted by the hardware: (SQ_INSTS_VALU_ADD_F64 + MUL_F64 + TRANS_F64 + 2 x FMA_F64) x 64 lanes "
                 "+ SQ_INSTS_VALU_MFMA_MOPS_F64 x 512; EXEC-masked lanes count as active (upper bound on useful work); key = kernel@items")
for wl in wls:
    sub, tkey, items, kname = WL[wl]
    for f in glob.glob(os.path.join(out, f"trace_{wl}", "**", "*kernel_stats.csv"), recursive=True):
        lines.append(f"# rocprofv3 --kernel-trace --stats -- python bench.py ({wl}) --cpu-sample 0 --steps 5 --warmup 1")
        lines += [l.rstrip() for l in open(f)]
        open(os.path.join(dst, f"{wl}_rocprofv3_kernel_stats.csv"), "w").write(open(f).read())
    for f in glob.glob(os.path.join(out, f"trace_{wl}", "**", "*kernel_trace.csv"), recursive=True):
        byk = {}
        for r in csv.DictReader(open(f)):
            byk.setdefault(r["Kernel_Name"], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r))
        lines.append(f"# per-kernel dispatch summary ({wl})")
        for k, v in byk.items():
            ds = [x[0] for x in v]
            r = v[0][1]
            if sub in k:
                # (pgdb1: one call = all its launches; the trace run is --steps 5 --warmup 1 = 6 calls)
                kernel_s[wl] = sum(ds) / (6 if wl in PER_CALL else len(ds)) / 1e9
            lines.append(f"{k[:110]}: calls={len(ds)} avg_ms={sum(ds)/len(ds)/1e6:.3f} min_ms={min(ds)/1e6:.3f} max_ms={max(ds)/1e6:.3f} "
                         f"lds={r.get('LDS_Block_Size','?')} vgpr={r.get('VGPR_Count','?')} accum_vgpr={r.get('Accum_VGPR_Count','?')} "
                         f"sgpr={r.get('SGPR_Count','?')} scratch={r.get('Scratch_Size', r.get('Private_Segment_Size','?'))}")
    p = os.path.join(out, f"bench_trace_{wl}.log")
    if os.path.exists(p):
        js = [l for l in open(p) if l.startswith("{")]
        if js:
            open(os.path.join(dst, f"bench_{wl}_line_under_rocprof.json"), "w").write(js[-1])
            if wl in PER_CALL:
                # a call of many launches that may OVERLAP (two pipelines on two streams since round 6): the sum of the launches'
                # durations is not the call's duration -- the HIP-event time of the call, from the bench line of the traced run
                try:
                    kernel_s[wl] = json.loads(js[-1])["roofline"]["kernel_ms"] / 1e3
                except Exception:
                    pass
    pmc = {}
    for d in sorted(glob.glob(os.path.join(out, f"pmc*_{wl}"))):
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if sub in r.get("Kernel_Name", ""):
                    pmc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    if wl not in PER_CALL:
        # launches of the same kernel that are not the workload's (sweep3: the one-item conversion that builds the reference matrix)
        pmc = {c: [x for x in v if x >= 0.01 * max(v)] if max(v) > 0 else v for c, v in pmc.items()}
    e = {c: {"mean": sum(v) / (PER_CALL[wl] if wl in PER_CALL else len(v)), "launches": len(v)} for c, v in pmc.items()}
    g = lambda c: e[c]["mean"] if c in e else None
    if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
        e["hbm_bytes_per_launch"] = g("FETCH_SIZE") * 1024 * 2 + g("WRITE_SIZE") * 1024
        traffic[tkey] = e["hbm_bytes_per_launch"]
        traffic[wl + "_FETCH_SIZE_KiB"], traffic[wl + "_WRITE_SIZE_KiB"] = g("FETCH_SIZE"), g("WRITE_SIZE")
    if g("SQ_INSTS_VALU_FMA_F64") is not None:
        valu = (g("SQ_INSTS_VALU_ADD_F64") + g("SQ_INSTS_VALU_MUL_F64") + g("SQ_INSTS_VALU_TRANS_F64") + 2 * g("SQ_INSTS_VALU_FMA_F64")) * 64
        mfma = (g("SQ_INSTS_VALU_MFMA_MOPS_F64") or 0.0) * 512
        e["fp64_flop_per_launch"] = valu + mfma
        flops[f"{kname}@{items}"] = {"flop_per_launch": valu + mfma, "valu_flop": valu, "mfma_flop": mfma, "items_per_launch": items,
                                     "add": g("SQ_INSTS_VALU_ADD_F64"), "mul": g("SQ_INSTS_VALU_MUL_F64"), "fma": g("SQ_INSTS_VALU_FMA_F64"),
                                     "trans": g("SQ_INSTS_VALU_TRANS_F64"), "mfma_mops": g("SQ_INSTS_VALU_MFMA_MOPS_F64")}
    if g("SQ_WAVE_CYCLES"):
        for k, c in (("valu_active_over_wave_cycles", "SQ_ACTIVE_INST_VALU"), ("lds_active_over_wave_cycles", "SQ_ACTIVE_INST_LDS"),
                     ("wait_any_over_wave_cycles", "SQ_WAIT_ANY"), ("wait_inst_over_wave_cycles", "SQ_WAIT_INST_ANY"),
                     ("vmem_active_over_wave_cycles", "SQ_ACTIVE_INST_VMEM")):
            if g(c) is not None:
                e[k] = g(c) / g("SQ_WAVE_CYCLES")
    # MFMA utilisation AS a utilisation (<= 1 by construction): matrix-core flops over the dense fp64 MFMA peak for the kernel's
    # duration, and matrix-core busy cycles over the cycles the chip's 1024 matrix-core pipes (4 per CU) had in that time
    if kernel_s.get(wl) and "fp64_flop_per_launch" in e:
        e["kernel_ms"] = 1e3 * kernel_s[wl]
        e["mfma_frac_of_fp64_peak"] = mfma / FP64_PEAK / kernel_s[wl]
        e["fp64_frac_of_peak"] = (valu + mfma) / FP64_PEAK / kernel_s[wl]
    if kernel_s.get(wl) and g("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
        e["mfma_busy_over_pipe_cycles"] = g("SQ_VALU_MFMA_BUSY_CYCLES") / (4 * CUS * kernel_s[wl] * CLOCK_HZ)
    if g("SQ_LDS_IDX_ACTIVE") and g("SQ_LDS_BANK_CONFLICT") is not None:
        e["lds_bank_conflict_over_lds_active"] = g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")
    summary["kernels"][wl] = e


def merge(name, new):
    """keep what an earlier (partial) run of the round wrote for other workloads"""
    p = os.path.join(dst, name)
    old = json.load(open(p)) if os.path.exists(p) else {}
    if "kernels" in new:
        old.setdefault("kernels", {}).update(new.pop("kernels"))
    old.update(new)
    json.dump(old, open(p, "w"), indent=1)


open(os.path.join(dst, "kernel_stats.txt"), "a").write("\n".join(lines) + "\n")
merge("pmc_counters.json", summary)
merge("pmc_traffic.json", traffic)
merge("pmc_flops.json", flops)
print("\n".join(lines)[:3000])
print(json.dumps({k: {c: v for c, v in e.items() if not isinstance(v, dict)} for k, e in json.load(open(os.path.join(dst, "pmc_counters.json")))["kernels"].items()}, indent=1))
print(json.dumps(json.load(open(os.path.join(dst, "pmc_flops.json"))), indent=1)[:3000])
