"""Per-item deviation of the TIMED mode (fixed 100 iterations) from the reference-generated fixtures, next to the reference's own
spread under a re-ordering of the settings (tests/golden/process_2q_*_fixed100_spread.npz, make_goldens.py --fixed-spread), for
the default line search, eig_rel_tol = 0, and the literal line search (FBX_MODE_LS_REFERENCE).  GPU box.
usage: python scripts/fixed_mode_items.py [out.json]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "forest-benchmarking_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from fbx import tomography, _lib
from fbx.design import process_design
from fbx_oracle import superops as so, measures as om
_lib.set_device(0)
GOLD = os.path.join(ROOT, "tests", "golden")
out = {}
for basis in ("pauli", "sic"):
    g = np.load(os.path.join(GOLD, f"process_2q_{basis}_fixed100.npz"))
    sp = np.load(os.path.join(GOLD, f"process_2q_{basis}_fixed100_spread.npz"))
    design = process_design(2, basis)
    e, c = g["expectations"], g["counts"]
    nb = e.shape[0]
    fid = lambda ch, b: om.process_fidelity(so.kraus2pauli_liouville([g["unitaries"][b]]), so.choi2pauli_liouville(ch))
    rows = {"reference_spread": sp["fixed_spread"].max(axis=1).tolist(), "conv_iter": g["conv_iter"].tolist()}
    for label, kw in (("default", {}), ("tol0", dict(eig_rel_tol=0.0)), ("tol0_ls_reference", dict(eig_rel_tol=0.0, line_search="reference")),
                      ("ls_reference", dict(line_search="reference"))):
        t0 = time.perf_counter()
        got, st = tomography.pgdb_process_estimate_batch(design, e, c, mode="fixed", max_iters=100, return_stats=True, trace_iters=100, **kw)
        dt = time.perf_counter() - t0
        dev = np.abs(got - g["pgdb_fixed"]).reshape(nb, -1).max(axis=1)
        fdev = np.array([abs(fid(got[b], b) - fid(g["pgdb_fixed"][b], b)) for b in range(nb)])
        dyk_eq = all(np.array_equal(st["trace"][b][:100, 0], g["dykstra"][b][:100]) for b in range(nb))
        bt_eq = sum(int(np.array_equal(st["trace"][b][:100, 1], g["backtracks"][b][:100])) for b in range(nb))
        rows[label] = {"dev": dev.tolist(), "fid_dev": fdev.tolist(), "le_1e-9": float((dev <= 1e-9).mean()), "le_1e-8": float((dev <= 1e-8).mean()),
                       "max": float(dev.max()), "fid_max": float(fdev.max()), "dykstra_traces_equal": dyk_eq,
                       "items_with_all_100_halving_counts_equal": bt_eq, "backtracks": st["backtracks"].tolist(),
                       "cost_evals": st["cost_evals"].tolist(), "call_s": dt}
        print(basis, label, f"<=1e-9 {rows[label]['le_1e-9']:.3f} <=1e-8 {rows[label]['le_1e-8']:.3f} max {dev.max():.2e} fid {fdev.max():.2e} "
              f"dykstra equal {dyk_eq}, items with every halving count equal {bt_eq}/{nb}", flush=True)
    out[basis] = rows
    spread = np.maximum(np.array(rows["reference_spread"]), 1e-10)
    for label in ("default", "tol0", "tol0_ls_reference"):
        r = np.array(rows[label]["dev"]) / spread
        print(basis, label, "dev / max(reference spread, 1e-10): max", f"{r.max():.1f}", "items above 2x:", int((r > 2).sum()), "above 10x:", int((r > 10).sum()))
# cost of the literal line search at the headline batch
from fbx import synthetic
design, _, e, c = synthetic.process_batch(2, "pauli", 1024)
d_e, d_c = _lib.DeviceBuffer.from_array(e), _lib.DeviceBuffer.from_array(c)
d_choi = _lib.DeviceBuffer(1024 * 256 * 16)
import ctypes
ms = ctypes.c_double()
for label, flag in (("exact", 0), ("reference", _lib.MODE_LS_REFERENCE)):
    ts = []
    for rep in range(4):
        _lib.check(_lib.lib().fbx_timer_begin())
        _lib.check(_lib.lib().fbx_pgdb_process_dev(design.handle, 1024, d_e.ptr, d_c.ptr, 1, _lib.MODE_FIXED | flag, 100, d_choi.ptr, None, None, None, None, None))
        _lib.check(_lib.lib().fbx_timer_end(ctypes.byref(ms))); ts.append(ms.value)
    out[f"b1024_ms_{label}"] = min(ts[1:])
    print("B=1024 fixed-100", label, f"{min(ts[1:]):.2f} ms")
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"))
