"""Per-item deviation of the GPU estimate from the reference goldens (tests/golden/process_*.npz)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "forest-benchmarking_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from fbx import tomography, _lib
from fbx.design import process_design
from fbx_oracle import design as od, estimators as oe
_lib.set_device(0)
for n, basis in ((2, "sic"), (2, "pauli")):
    g = np.load(os.path.join(ROOT, "tests", "golden", f"process_{n}q_{basis}.npz"))
    design = process_design(n, basis)
    got, st = tomography.pgdb_process_estimate_batch(design, g["expectations"], g["counts"], return_stats=True)
    d = np.abs(got - g["pgdb"]).reshape(got.shape[0], -1).max(axis=1)
    print(basis, "gpu vs golden:", " ".join(f"{x:.1e}" for x in d))
    o = od.Design(n, "process", design.in_labels, design.paulis, design.coefs)
    A = oe.design_matrix_A(o)
    bad = [b for b in range(got.shape[0]) if d[b] > 1e-10]
    for b in bad:
        want, ws = oe.pgdb_process_estimate(o, g["expectations"][b], g["counts"][b], A=A, return_stats=True)
        print("  item", b, "oracle vs golden %.1e" % np.abs(want - g["pgdb"][b]).max(), "gpu vs oracle %.1e" % np.abs(want - got[b]).max(),
              "iters", st["iterations"][b], ws["iterations"], "dyk", st["dykstra"][b], ws["dykstra"], "bt", st["backtracks"][b], ws["backtracks"])
