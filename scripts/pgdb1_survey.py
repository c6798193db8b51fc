"""Parity survey of the lane-per-item single-qubit kernel against the oracle: N experiments per design, converge mode --
deviation histogram, iteration / Dykstra / halving counts.  usage: python scripts/pgdb1_survey.py [N]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "forest-benchmarking_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from fbx import synthetic, tomography, _lib
from fbx_oracle import design as od, estimators as oe
_lib.set_device(0); _lib.set_option("pgdb_packed_1q", 2.0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
for basis, tp in (("pauli", True), ("sic", True), ("pauli", False)):
    design, us, e, c = synthetic.process_batch(1, basis, N)
    got, st = tomography.pgdb_process_estimate_batch(design, e, c, trace_preserving=tp, return_stats=True)
    d = od.Design(1, "process", design.in_labels, design.paulis, design.coefs)
    A = oe.design_matrix_A(d)
    dev = np.zeros(N); bad_it = bad_dy = bad_bt = 0
    for b in range(N):
        want, w = oe.pgdb_process_estimate(d, e[b], c[b], trace_preserving=tp, A=A, return_stats=True)
        dev[b] = np.abs(got[b] - want).max()
        bad_it += st["iterations"][b] != w["iterations"]; bad_dy += st["dykstra"][b] != w["dykstra"]; bad_bt += st["backtracks"][b] != w["backtracks"]
    print(f"{basis} tp={tp} N={N}: max dev {dev.max():.2e}, > 1e-9: {(dev > 1e-9).sum()}, > 1e-8: {(dev > 1e-8).sum()}, non-finite {(~np.isfinite(dev)).sum()}; "
          f"iteration mismatches {bad_it}, Dykstra {bad_dy}, halvings {bad_bt}; worst items {np.argsort(dev)[-3:]}", flush=True)
