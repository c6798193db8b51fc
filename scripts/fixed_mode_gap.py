"""Fixed-100 parity gap study: GPU (library given by FBX_LIBRARY) vs the oracle on the first N bench items.
The oracle results are cached in gpurun_out/oracle_fixed100.npz (computed once, ~5 s per item).
usage: python scripts/fixed_mode_gap.py [N]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "forest-benchmarking_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from fbx import synthetic, tomography, _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
_lib.set_device(0)
design, us, e, c = synthetic.process_batch(2, "pauli", N)
cache = os.path.join(ROOT, "gpurun_out", f"oracle_fixed100_{N}.npz")
if os.path.exists(cache):
    z = np.load(cache); want, wbt = z["choi"], z["bt"]
else:
    from fbx_oracle import design as od, estimators as oe
    d = od.Design(2, "process", design.in_labels, design.paulis, design.coefs)
    A = oe.design_matrix_A(d)
    res = [oe.pgdb_process_estimate(d, e[b], c[b], mode="fixed", max_iters=100, A=A, return_stats=True) for b in range(N)]
    want = np.array([r[0] for r in res]); wbt = np.array([r[1]["backtracks"] for r in res])
    os.makedirs(os.path.dirname(cache), exist_ok=True)
    np.savez(cache, choi=want, bt=wbt)
got, st = tomography.pgdb_process_estimate_batch(design, e, c, mode="fixed", max_iters=100, return_stats=True)
d = np.abs(got - want).reshape(N, -1).max(axis=1)
print(os.environ.get("FBX_LIBRARY", "libfbx.so"), "max|dChoi| per item:", " ".join(f"{x:.1e}" for x in d))
print("   backtracks gpu:", st["backtracks"].tolist()); print("   backtracks ora:", wbt.tolist())
