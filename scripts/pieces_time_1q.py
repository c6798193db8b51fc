"""The one-wave kernel in pieces (pgdb_pieces_kernel, fbx_pgdb.hip) for single-qubit batches below the lane-per-item kernel's
range: times against whole reconstructions, bit-identical outputs.  usage: python scripts/pieces_time_1q.py [B ...]"""
import sys, os, ctypes
sys.path.insert(0, "forest-benchmarking_amd")
import numpy as np
from fbx import synthetic, _lib
_lib.set_device(0)
_lib.set_option("pgdb_packed_1q", 0.0)
ms = ctypes.c_double()
sizes = [int(a) for a in sys.argv[1:]] or [2048, 4096, 8192]
for basis in ("pauli", "sic"):
    design, us, e0, c0 = synthetic.process_batch(1, basis, max(sizes))
    for B in sizes:
        d_e, d_c = _lib.DeviceBuffer.from_array(e0[:B]), _lib.DeviceBuffer.from_array(c0[:B])
        bufs = [_lib.DeviceBuffer(B * 32 * 8), _lib.DeviceBuffer(B * 4), _lib.DeviceBuffer(B * 4), _lib.DeviceBuffer(B * 4), _lib.DeviceBuffer(B * 8)]
        def run(P, mode, iters):
            os.environ["FBX_LEAN_PIECES"] = str(P)
            best = 1e9
            for rep in range(4):
                _lib.check(_lib.lib().fbx_timer_begin())
                _lib.check(_lib.lib().fbx_pgdb_process_dev(design.handle, B, d_e.ptr, d_c.ptr, 1, mode, iters, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, bufs[3].ptr, bufs[4].ptr, None))
                _lib.check(_lib.lib().fbx_timer_end(ctypes.byref(ms)))
                if rep: best = min(best, ms.value)
            return best, (bufs[0].to_array(np.float64, (B, 32)), bufs[1].to_array(np.int32, (B,)), bufs[2].to_array(np.int32, (B,)), bufs[3].to_array(np.int32, (B,)), bufs[4].to_array(np.float64, (B,)))
        for mode, iters, name in ((_lib.MODE_CONVERGE, 0, "converge"), (_lib.MODE_FIXED, 100, "fixed-100")):
            t0, ref = run(1, mode, iters)
            line = f"1q {basis} B={B} {name}: whole {t0:.2f} ms = {B / t0:.0f} k/s;"
            for P in (4, 8, 16):
                t1, got = run(P, mode, iters)
                line += f"  {P} pieces {t1:.2f} ms ({'=' if all(np.array_equal(a, b) for a, b in zip(ref, got)) else 'DIFFERENT'})"
            print(line, flush=True)
        for b in [d_e, d_c] + bufs: b.free()
