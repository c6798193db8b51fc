"""Binned relaunch of the single-qubit lane-per-item kernel (fbx_pgdb1.hip) against the persistent kernel: bit-identical results,
times.  usage: python scripts/pgdb1_binned_time.py [log2 B ...]"""
import sys, os, ctypes
sys.path.insert(0, "forest-benchmarking_amd")
import numpy as np
from fbx import synthetic, _lib
_lib.set_device(0)
_lib.set_option("pgdb_packed_1q", 2.0)
ms = ctypes.c_double()
sizes = [int(a) for a in sys.argv[1:]] or [16, 18, 20]
def run(design, B, d_e, d_c, bufs, mode, iters, env):
    for k, v in env.items(): os.environ[k] = str(v)
    best = 1e9
    for rep in range(3):
        _lib.check(_lib.lib().fbx_timer_begin())
        _lib.check(_lib.lib().fbx_pgdb_process_dev(design.handle, B, d_e.ptr, d_c.ptr, 1, mode, iters, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, bufs[3].ptr, bufs[4].ptr, None))
        _lib.check(_lib.lib().fbx_timer_end(ctypes.byref(ms)))
        if rep: best = min(best, ms.value)
    out = (bufs[0].to_array(np.float64, (B, 32)), bufs[1].to_array(np.int32, (B,)), bufs[2].to_array(np.int32, (B,)), bufs[3].to_array(np.int32, (B,)), bufs[4].to_array(np.float64, (B,)))
    return best, out
for basis in ("pauli", "sic"):
    design, us, e0, c0 = synthetic.process_batch(1, basis, 16384)
    for lb in sizes:
        B = 1 << lb
        reps = (B + 16383) // 16384
        e = np.tile(e0, (reps, 1))[:B]; c = np.tile(c0, (reps, 1))[:B]
        d_e, d_c = _lib.DeviceBuffer.from_array(e), _lib.DeviceBuffer.from_array(c)
        bufs = [_lib.DeviceBuffer(B * 32 * 8), _lib.DeviceBuffer(B * 4), _lib.DeviceBuffer(B * 4), _lib.DeviceBuffer(B * 4), _lib.DeviceBuffer(B * 8)]
        for mode, iters, name in ((_lib.MODE_CONVERGE, 0, "converge"), (_lib.MODE_FIXED, 30, "fixed-30")):
            t0, ref = run(design, B, d_e, d_c, bufs, mode, iters, {"FBX_P1_BINNED": 0})
            print(f"1q {basis} B=2^{lb} {name}: persistent {t0:.2f} ms = {B / t0 * 1e3:.3g} /s  mean/max iters {ref[1].mean():.1f}/{ref[1].max()}", flush=True)
            for tail in (8192,):
                t1, got = run(design, B, d_e, d_c, bufs, mode, iters, {"FBX_P1_BINNED": 2, "FBX_P1_TAIL": tail})
                same = all(np.array_equal(a, b) for a, b in zip(ref, got))
                print(f"      binned tail={tail}: {t1:.2f} ms = {B / t1 * 1e3:.3g} /s  identical={same}", flush=True)
        for b in [d_e, d_c] + bufs: b.free()
