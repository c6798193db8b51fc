"""Parity survey on the first 256 bench items (2 qubits, Pauli in-basis) against oracle results computed ahead
(scripts/cache/oracle_pauli.npz: produced in the build container by running the oracle, scripts/make_parity_cache.py).
Prints the deviation distribution of the Choi matrices and the count mismatches, both modes, for FBX_LIBRARY."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "forest-benchmarking_amd"))
import numpy as np
from fbx import synthetic, tomography, _lib
_lib.set_device(0)
CACHE = sys.argv[1] if len(sys.argv) > 1 else "oracle_pauli.npz"       # any file written by make_parity_cache.py
z = np.load(os.path.join(ROOT, "scripts", "cache", CACHE))
N = z["fixed"].shape[0]
BASIS = str(z["basis"]) if "basis" in z.files else "pauli"
FIRST = int(z["first"]) if "first" in z.files else 0
TP = bool(z["tp"]) if "tp" in z.files else True
NQ = 3 if BASIS.endswith("3") else 2
design, us, e, c = synthetic.process_batch(NQ, BASIS.rstrip("3"), N, first_item=FIRST)
for mode, key, kw in (("fixed-100", "fixed", dict(mode="fixed", max_iters=100)), ("converge", "conv", {})):
    got, st = tomography.pgdb_process_estimate_batch(design, e, c, trace_preserving=TP, return_stats=True, **kw)
    d = np.abs(got - z[key]).reshape(N, -1).max(axis=1)
    print(f"{os.path.basename(os.environ.get('FBX_LIBRARY', 'libfbx.so')):14s} {CACHE[7:-4]:14s} {mode:9s} max {d.max():.1e}  >1e-9: {(d > 1e-9).sum()}  >1e-10: {(d > 1e-10).sum()}  >1e-8: {(d > 1e-8).sum()}  "
          f"dykstra mismatches {(st['dykstra'] != z[key + '_dyk']).sum()}  halving-count mismatches {(st['backtracks'] != z[key + '_bt']).sum()}"
          + (f"  iteration mismatches {(st['iterations'] != z['conv_it']).sum()}" if key == "conv" else "")
          + (f"  items above 1e-9: {[(FIRST + int(i), float('%.1e' % d[i])) for i in np.flatnonzero(d > 1e-9)]}" if key == "conv" and (d > 1e-9).any() else ""))
    if key == "fixed":
        top = np.argsort(-d)[:3]
        print("      largest fixed-100 deviations:", [(FIRST + int(i), float('%.1e' % d[i]), int(st['backtracks'][i]), int(z['fixed_bt'][i])) for i in top], "(item, dev, halvings gpu, oracle)")
