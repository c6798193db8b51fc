"""Process-fidelity bootstrap on resident data: B two-qubit process tomographies x R Beta
resamples -> PGDB -> PTM -> fidelity; prints the time of the resampler alone and of the whole pipeline."""
import sys, os, ctypes, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'forest-benchmarking_amd'))
import numpy as np
from fbx import synthetic, _lib, tomography
from fbx.operator_tools import convert_batch
_lib.set_device(0)
lib = _lib.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
R = int(sys.argv[2]) if len(sys.argv) > 2 else 40
design, us, e, c = synthetic.process_batch(2, "pauli", B)
ideal = convert_batch("kraus", "pauli_liouville", us[:, None])
m = design.m
d_e, d_c = _lib.DeviceBuffer.from_array(e), _lib.DeviceBuffer.from_array(c)
d_er, d_cr = _lib.DeviceBuffer(R * B * m * 8), _lib.DeviceBuffer(R * B * m * 8)
ms = ctypes.c_double()
best = 1e9
for _ in range(4):
    _lib.check(lib.fbx_timer_begin())
    _lib.check(lib.fbx_beta_resample_dev(B * m, R, d_e.ptr, d_c.ptr, 1.0, 7, d_er.ptr, d_cr.ptr))
    _lib.check(lib.fbx_timer_end(ctypes.byref(ms))); best = min(best, ms.value)
print('resample: %d draws in %.3f ms = %.2e draws/s' % (R * B * m, best, R * B * m / best * 1e3))
t0 = time.perf_counter(); x = 2 * np.random.beta(np.tile((e + 1) / 2 * c + 1, (2, 1, 1)), np.tile(c - (e + 1) / 2 * c + 1, (2, 1, 1))) - 1
dt = time.perf_counter() - t0
print('numpy beta on the host: %.2e draws/s (one core)' % (x.size / dt))
for it in range(2):
    t0 = time.perf_counter()
    mean, var = tomography.process_fidelity_variance_batch(design, e, c, ideal, n_resamples=R, seed=it)
    dt = time.perf_counter() - t0
    print('bootstrap: %d x %d reconstructions in %.3f s = %.0f recon/s; fidelity %.4f +- %.4f (item 0)' %
          (B, R, dt, B * R / dt, mean[0], np.sqrt(var[0])))
