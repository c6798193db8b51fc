"""Bit-for-bit comparison of two builds of the library on the same inputs (2-qubit PGDB, both kernels; 3-qubit PGDB).
usage: python scripts/compare_libs.py libA.so libB.so"""
import os, subprocess, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, hashlib
sys.path.insert(0, os.path.join(sys.argv[1], "forest-benchmarking_amd"))
import numpy as np
from fbx import synthetic, tomography, _lib
_lib.set_device(0)
out = []
for n, basis, B, kw in ((2, "pauli", 256, dict(mode="fixed", max_iters=100)), (2, "pauli", 2304, {}), (2, "sic", 64, {}), (1, "pauli", 32, {}),
                        (3, "sic", 4, dict(mode="fixed", max_iters=12))):
    design, _, e, c = synthetic.process_batch(n, basis, min(B, 256))
    if B > 256:
        e = np.tile(e, (B // 256, 1)); c = np.tile(c, (B // 256, 1))
    choi, st = tomography.pgdb_process_estimate_batch(design, e, c, return_stats=True, **kw)
    h = hashlib.sha256(np.ascontiguousarray(choi).tobytes())
    for k in ("iterations", "dykstra", "backtracks"):
        h.update(np.ascontiguousarray(st[k]).tobytes())
    out.append(h.hexdigest()[:16])
# everything else that shares the touched headers: state MLE, Choi projections, conversions (incl. the eigh routes), the sweeps
import warnings
warnings.simplefilter("ignore")
def H(*arrs):
    h = hashlib.sha256()
    for a in arrs:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:12]
from fbx.operator_tools import project_superoperators as ps, superoperator_transformations as st_
from fbx.operator_tools import project_state_matrix as psm
for n, B in ((1, 64), (2, 64), (3, 16)):
    design, _, e, c = synthetic.state_batch(n, B, mixed=0.05)
    out.append(H(tomography.iterative_mle_state_estimate_batch(design, e, c, maxiter=60)))
    out.append(H(tomography.iterative_mle_state_estimate_batch(design, e, c, maxiter=30, beta=0.5)))
rs = np.random.RandomState(3)
for n, B in ((1, 16), (2, 16), (3, 4)):
    D = 4 ** n
    x = rs.randn(B, D, D) + 1j * rs.randn(B, D, D)
    x = x + x.conj().transpose(0, 2, 1) + np.eye(D) * 2
    for kind in range(4):
        out.append(H(*ps.proj_choi_batch(kind, x, return_iters=True)))
    out.append(H(st_.convert_batch("choi", "chi", x)))
    out.append(H(st_.convert_batch("choi", "pauli_liouville", x)))
    k = synthetic.kraus_batch(n, 4, B, seed=5)
    out.append(H(st_.convert_batch("kraus", "chi", k)))
    out.append(H(psm.project_state_matrix_to_physical_batch(x[:, :2 ** n, :2 ** n])))
lib = _lib.lib()
for n, B in ((2, 1001), (3, 33)):
    K, D = 4, 4 ** n
    k = np.ascontiguousarray(synthetic.kraus_batch(n, K, B, seed=9))
    ref = np.ascontiguousarray(st_.convert_batch("kraus", "pauli_liouville", k[:1])[0])
    choi, ptm, chi = (np.empty((B, D, D), dtype=np.complex128) for _ in range(3))
    fid = np.empty(B)
    _lib.check(lib.fbx_kraus_sweep(n, B, K, _lib.dptr(k.view(np.float64)), _lib.dptr(ref.view(np.float64)), _lib.dptr(choi.view(np.float64)),
                                   _lib.dptr(ptm.view(np.float64)), _lib.dptr(chi.view(np.float64)), _lib.dptr(fid)))
    out.append(H(choi, ptm, chi, fid))
print(" ".join(out))
'''
res = []
for lib in sys.argv[1:3]:
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT], env=dict(os.environ, FBX_LIBRARY=os.path.join(ROOT, "forest-benchmarking_amd", lib)),
                       capture_output=True, text=True)
    print(f"{lib:24s} {r.stdout.strip()} {r.stderr.strip()[-300:]}")
    res.append(r.stdout.strip())
print("IDENTICAL" if res[0] == res[1] and res[0] else "DIFFERENT")
