import sys, os, ctypes
sys.path.insert(0, "forest-benchmarking_amd")
import numpy as np
from fbx import synthetic, tomography, _lib
_lib.set_device(0)
B = 16384
design, us, e, c = synthetic.process_batch(2, "pauli", B)
def run(env, **kw):
    for k in ("FBX_LEAN_PIECES", "FBX_LEAN_PIECE_ITERS"): os.environ.pop(k, None)
    os.environ.update(env)
    return tomography.pgdb_process_estimate_batch(design, e, c, return_stats=True, **kw)
for kw in (dict(mode="fixed", max_iters=100), dict(mode="converge")):
    ref, rs = run({"FBX_LEAN_PIECES": "1"}, **kw)
    for env in ({"FBX_LEAN_PIECES": "64", "FBX_LEAN_PIECE_ITERS": "1"}, {"FBX_LEAN_PIECES": "50", "FBX_LEAN_PIECE_ITERS": "2"}, {"FBX_LEAN_PIECES": "8"}):
        for rep in range(3):
            got, gs = run(env, **kw)
            ok = np.array_equal(ref, got) and all(np.array_equal(np.asarray(rs[k]), np.asarray(gs[k])) for k in rs)
            print(kw, env, "rep", rep, "identical", ok, flush=True)
            assert ok
print("stress ok")
