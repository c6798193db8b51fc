"""Resident throughput of the state-side kernels: measures, physical projection, channel application, linear inversion."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "forest-benchmarking_amd"))
import numpy as np
from fbx import _lib, synthetic
from fbx.design import process_design
_lib.set_device(0)
lib = _lib.lib()
rs = np.random.RandomState(0)


def timed(call, reps=5):
    ms = ctypes.c_double(); best = 1e9
    for rep in range(reps):
        _lib.check(lib.fbx_timer_begin()); _lib.check(call()); _lib.check(lib.fbx_timer_end(ctypes.byref(ms)))
        if rep: best = min(best, ms.value)
    return best


for n, B in ((1, 1_000_000), (2, 400_000), (3, 100_000)):
    d = 2 ** n
    g = rs.randn(2, 4096, d, d) + 1j * rs.randn(2, 4096, d, d)
    rho = g[0] @ g[0].conj().transpose(0, 2, 1); rho /= np.trace(rho, axis1=1, axis2=2)[:, None, None]
    sig = g[1] @ g[1].conj().transpose(0, 2, 1); sig /= np.trace(sig, axis1=1, axis2=2)[:, None, None]
    rho = np.ascontiguousarray(np.tile(rho, (B // 4096 + 1, 1, 1))[:B]); sig = np.ascontiguousarray(np.tile(sig, (B // 4096 + 1, 1, 1))[:B])
    dr, dsg = _lib.DeviceBuffer.from_array(rho), _lib.DeviceBuffer.from_array(sig)
    o = [_lib.DeviceBuffer(B * 8) for _ in range(4)]
    dout = _lib.DeviceBuffer(rho.nbytes)
    t = timed(lambda: lib.fbx_state_measures_dev(n, B, dr.ptr, dsg.ptr, o[0].ptr, o[1].ptr, o[2].ptr, o[3].ptr))
    print(f"state_measures (4 measures) n={n} B={B}: {t:.3f} ms  {B / t / 1e3:.1f} M pairs/s  {2 * rho.nbytes / t / 1e6:.0f} GB/s")
    t = timed(lambda: lib.fbx_state_measures_dev(n, B, dr.ptr, dsg.ptr, o[0].ptr, None, o[2].ptr, o[3].ptr))
    print(f"state_measures (no fidelity)  n={n} B={B}: {t:.3f} ms  {B / t / 1e3:.1f} M pairs/s  {2 * rho.nbytes / t / 1e6:.0f} GB/s")
    bad = rho - 0.3 * np.eye(d) / d
    db = _lib.DeviceBuffer.from_array(np.ascontiguousarray(bad))
    t = timed(lambda: lib.fbx_proj_state_physical_dev(n, B, db.ptr, dout.ptr))
    print(f"proj_state_physical           n={n} B={B}: {t:.3f} ms  {B / t / 1e3:.1f} M states/s")
    for b in [dr, dsg, dout, db] + o: b.free()
for n, B in ((1, 400_000), (2, 100_000)):
    d = 2 ** n; D = d * d
    ks = synthetic.kraus_batch(n, 2, 1024, seed=3)
    design = process_design(n, "pauli")
    e = np.ascontiguousarray(np.tile(rs.rand(1024, design.m) * 2 - 1, (B // 1024 + 1, 1))[:B])
    de = _lib.DeviceBuffer.from_array(e); dc = _lib.DeviceBuffer(B * D * D * 16)
    t = timed(lambda: lib.fbx_linv_process_dev(design.handle, B, de.ptr, dc.ptr))
    print(f"linv_process n={n} m={design.m} B={B}: {t:.3f} ms  {B / t / 1e3:.1f} M items/s  {(e.nbytes + B * D * D * 16) / t / 1e6:.0f} GB/s")
    rho = np.ascontiguousarray(np.tile(np.eye(d, dtype=complex) / d, (B, 1, 1)))
    drho = _lib.DeviceBuffer.from_array(rho); dout = _lib.DeviceBuffer(rho.nbytes)
    t = timed(lambda: lib.fbx_apply_choi_dev(n, B, dc.ptr, drho.ptr, dout.ptr))
    print(f"apply_choi   n={n} B={B}: {t:.3f} ms  {B / t / 1e3:.1f} M items/s  {(B * D * D * 16) / t / 1e6:.0f} GB/s")
    for b in (de, dc, drho, dout): b.free()
