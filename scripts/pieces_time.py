"""The two-waves 2-qubit kernel in pieces (fbx_pgdb_lean.hip) against whole reconstructions: bit-identical results, times.
usage: python scripts/pieces_time.py [B ...]"""
import sys, os, ctypes
sys.path.insert(0, "forest-benchmarking_amd")
import numpy as np
from fbx import synthetic, _lib
_lib.set_device(0)
ms = ctypes.c_double()
sizes = [int(a) for a in sys.argv[1:]] or [8192]
design, us, e0, c0 = synthetic.process_batch(2, "pauli", max(sizes))
def run(B, d_e, d_c, bufs, mode, iters, env, reps=3):
    for k, v in env.items(): os.environ[k] = str(v)
    best = 1e9
    for rep in range(reps):
        _lib.check(_lib.lib().fbx_timer_begin())
        _lib.check(_lib.lib().fbx_pgdb_process_dev(design.handle, B, d_e.ptr, d_c.ptr, 1, mode, iters, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, bufs[3].ptr, bufs[4].ptr, bufs[5].ptr))
        _lib.check(_lib.lib().fbx_timer_end(ctypes.byref(ms)))
        if rep: best = min(best, ms.value)
    out = (bufs[0].to_array(np.float64, (B, 512)), bufs[1].to_array(np.int32, (B,)), bufs[2].to_array(np.int32, (B,)), bufs[3].to_array(np.int32, (B,)),
           bufs[4].to_array(np.float64, (B,)), bufs[5].to_array(np.int32, (B, 4)))
    return best, out
for B in sizes:
    d_e, d_c = _lib.DeviceBuffer.from_array(e0[:B]), _lib.DeviceBuffer.from_array(c0[:B])
    bufs = [_lib.DeviceBuffer(B * 512 * 8), _lib.DeviceBuffer(B * 4), _lib.DeviceBuffer(B * 4), _lib.DeviceBuffer(B * 4), _lib.DeviceBuffer(B * 8), _lib.DeviceBuffer(B * 16)]
    for mode, iters, name in ((_lib.MODE_FIXED, 100, "fixed-100"), (_lib.MODE_CONVERGE, 0, "converge")):
        t0, ref = run(B, d_e, d_c, bufs, mode, iters, {"FBX_LEAN_PIECES": 1})
        print(f"2q pauli B={B} {name}: whole {t0:.2f} ms = {B / t0:.1f} k/s", flush=True)
        for P in (4, 8, 16):
            t1, got = run(B, d_e, d_c, bufs, mode, iters, {"FBX_LEAN_PIECES": P})
            same = all(np.array_equal(a, b) for a, b in zip(ref, got))
            print(f"      {P:2d} pieces: {t1:.2f} ms = {B / t1:.1f} k/s  identical={same}", flush=True)
    for b in [d_e, d_c] + bufs: b.free()
