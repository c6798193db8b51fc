"""Device-resident bandwidth of the small reduction kernels: process fidelity, DFE estimate, calibration rescale."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "forest-benchmarking_amd"))
import numpy as np
from fbx import _lib
_lib.set_device(0)
lib = _lib.lib()
rs = np.random.RandomState(0)


def timed(call, reps=6):
    ms = ctypes.c_double(); best = 1e9
    for rep in range(reps):
        _lib.check(lib.fbx_timer_begin()); _lib.check(call()); _lib.check(lib.fbx_timer_end(ctypes.byref(ms)))
        if rep: best = min(best, ms.value)
    return best


for n, B in ((1, 2_000_000), (2, 200_000), (3, 20_000)):
    D = 4 ** n
    a = rs.randn(B, D, D * 2); ref = rs.randn(1, D, D * 2)
    da, dr = _lib.DeviceBuffer.from_array(a), _lib.DeviceBuffer.from_array(np.broadcast_to(ref, a.shape).copy())
    dfe_, dfp = _lib.DeviceBuffer(B * 8), _lib.DeviceBuffer(B * 8)
    t = timed(lambda: lib.fbx_process_fidelity_dev(n, B, dr.ptr, da.ptr, dfe_.ptr, dfp.ptr))
    got = dfp.to_array(np.float64, (B,))[:3]
    print(f"process_fidelity n={n} B={B}: {t:.3f} ms  {2 * a.nbytes / t / 1e6:.0f} GB/s  {B / t / 1e3:.1f} M items/s")
    for b in (da, dr, dfe_, dfp): b.free()
lib.fbx_dfe_estimate_dev = getattr(lib, "fbx_dfe_estimate_dev", None)
for m, B in ((15, 1_000_000), (255, 100_000)):
    e = rs.rand(B, m); se = rs.rand(B, m) * 0.01
    mean, err = np.empty(B), np.empty(B)
    import time
    t0 = time.time(); _lib.check(lib.fbx_dfe_estimate(2, 1, B, m, _lib.dptr(e), _lib.dptr(se), _lib.dptr(mean), _lib.dptr(err))); dt = time.time() - t0
    print(f"dfe_estimate (host form incl. transfers) m={m} B={B}: {1e3 * dt:.1f} ms  {(e.nbytes + se.nbytes) / dt / 1e9:.1f} GB/s")
for B, m in ((4096, 540),):
    n = B * m
    e, se = rs.rand(B, m), rs.rand(B, m) * 0.01
    idx = rs.randint(0, 16, size=m).astype(np.int32)
    cm, cv = rs.rand(B, 16) * 0.2 + 0.8, rs.rand(B, 16) * 1e-4
    bufs = [_lib.DeviceBuffer.from_array(x) for x in (e, se, idx, cm, cv)]
    o1, o2 = _lib.DeviceBuffer(n * 8), _lib.DeviceBuffer(n * 8)
    t = timed(lambda: lib.fbx_calibrate_expectations_dev(B, m, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, 16, bufs[3].ptr, bufs[4].ptr, o1.ptr, o2.ptr))
    print(f"calibrate_expectations B={B} m={m}: {t:.3f} ms  {32 * n / t / 1e6:.0f} GB/s")
