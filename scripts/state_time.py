"""Throughput of the batched state estimators (diagnostic)."""
import sys, os, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'forest-benchmarking_amd'))
import numpy as np
from fbx import synthetic, tomography, _lib
_lib.set_device(0)
warnings.simplefilter("ignore")
for n, B in ((1, 16384), (2, 16384), (3, 4096), (4, 1024), (5, 512)):
    design, rhos, e, c = synthetic.state_batch(n, min(B, 1024 if n < 4 else 64), mixed=0.05)
    reps = B // e.shape[0]
    e = np.tile(e, (reps, 1)); c = np.tile(c, (reps, 1))
    for name, fn in (("linear_inv", lambda: tomography.linear_inv_state_estimate_batch(design, e)),
                     ("mle maxiter=1000", lambda: tomography.iterative_mle_state_estimate_batch(design, e, c, maxiter=1000, return_stats=True))):
        fn()
        t = time.time(); out = fn(); dt = time.time() - t
        extra = ''
        if isinstance(out, tuple):
            extra = ' mean iterations %.0f' % out[1]['iterations'].mean()
        print('n=%d B=%d %-18s %.1f ms  %.3e states/s%s' % (n, B, name, dt * 1e3, B / dt, extra))
