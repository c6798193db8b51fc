import sys, os
sys.path.insert(0, "forest-benchmarking_amd")
import numpy as np
from fbx import synthetic, tomography, _lib
_lib.set_device(0)
for basis in ("pauli", "sic"):
    design, _, e, c = synthetic.process_batch(2, basis, 128)
    reps = 2048 // 128
    eb, cb = np.tile(e, (reps, 1)), np.tile(c, (reps, 1))
    for kw in (dict(mode="converge"), dict(mode="fixed", max_iters=60), dict(mode="converge", trace_preserving=False), dict(mode="fixed", max_iters=100)):
        small, ss = tomography.pgdb_process_estimate_batch(design, e, c, return_stats=True, **kw)
        big, sb = tomography.pgdb_process_estimate_batch(design, eb, cb, return_stats=True, **kw)
        d = np.abs(big[:128] - small).reshape(128, -1).max(axis=1)
        print(basis, kw, "max", d.max(), "median", np.median(d), ">1e-10:", int((d > 1e-10).sum()), "bt diff max", np.abs(sb["backtracks"][:128].astype(int) - ss["backtracks"]).max(),
              "counts equal", all(np.array_equal(sb[k][:128], ss[k]) for k in ("iterations", "dykstra")), "cost diff", np.abs(sb["cost"][:128] - ss["cost"]).max())
