"""Per-phase cycle breakdown of the 3-qubit PGDB kernel (needs libfbx_prof.so: build.py --profile)."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["FBX_LIBRARY"] = os.path.join(ROOT, "forest-benchmarking_amd", os.environ.get("FBX_PROF_LIB", "libfbx_prof.so"))
sys.path.insert(0, os.path.join(ROOT, 'forest-benchmarking_amd'))
import numpy as np
from fbx import synthetic, tomography, _lib
basis = sys.argv[1] if len(sys.argv) > 1 else 'sic'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
mode = sys.argv[3] if len(sys.argv) > 3 else 'fixed'
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 100
design, us, e, c = synthetic.process_batch(3, basis, B)
_lib.set_device(0)
lib = _lib.lib()
buf = _lib.DeviceBuffer(B * 8 * 8)
lib.fbx_debug_set_phase_buffer.argtypes = [ctypes.c_void_p]
lib.fbx_debug_set_phase_buffer(buf.ptr)
t = time.time()
choi, st = tomography.pgdb_process_estimate_batch(design, e, c, mode=mode, max_iters=iters if mode == 'fixed' else 0, return_stats=True)
dt = time.time() - t
ph = buf.to_array(np.int64, (B, 8))
names = ['jacobi', 'reconstruct', 'dykstra-other', 'transforms', 'gradient', 'linesearch', 'warm-rotate', 'predict']
tot = ph.sum(1)
print('3q', basis, 'B', B, mode, 'time %.1f ms' % (1e3 * dt), 'recon/s %.0f' % (B / dt))
print('cycles/item mean %.3e max %.3e' % (tot.mean(), tot.max()))
for i, n in enumerate(names):
    print('  %-14s mean %.3e (%.1f%%)' % (n, ph[:, i].mean(), 100 * ph[:, i].sum() / tot.sum()))
sweeps = st['jacobi_sweeps']
print('dykstra iters mean %.1f; sweeps per eigh %.2f; jacobi cycles per eigh %.0f, per round %.0f; warm rotate per eigh %.0f' % (
    st['dykstra'].mean(), sweeps.sum() / st['dykstra'].sum(), ph[:, 0].sum() / st['dykstra'].sum(),
    ph[:, 0].sum() / (sweeps.sum() * 63.0), ph[:, 6].sum() / st['dykstra'].sum()))
print('outer iterations mean %.1f; per outer iteration: transforms %.0f predict %.0f gradient %.0f linesearch %.0f' % (
    st['iterations'].mean(), *(ph[:, k].sum() / st['iterations'].sum() for k in (3, 7, 4, 5))))
