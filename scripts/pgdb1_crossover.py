import sys, ctypes
sys.path.insert(0, "forest-benchmarking_amd")
import numpy as np
from fbx import synthetic, _lib
_lib.set_device(0)
ms = ctypes.c_double()
for basis in ("pauli", "sic"):
    design, us, e_all, c_all = synthetic.process_batch(1, basis, 16384)
    for B in (2048, 4096, 8192, 16384):
        e = e_all[:B].copy(); c = c_all[:B].copy()
        d_e, d_c = _lib.DeviceBuffer.from_array(e), _lib.DeviceBuffer.from_array(c)
        d_choi = _lib.DeviceBuffer(B * 16 * 16)
        out = []
        for packed in (2.0, 0.0):
            with _lib.option("pgdb_packed_1q", packed):
                best = 1e9
                for rep in range(4):
                    _lib.check(_lib.lib().fbx_timer_begin())
                    _lib.check(_lib.lib().fbx_pgdb_process_dev(design.handle, B, d_e.ptr, d_c.ptr, 1, _lib.MODE_CONVERGE, 0, d_choi.ptr, None, None, None, None, None))
                    _lib.check(_lib.lib().fbx_timer_end(ctypes.byref(ms)))
                    if rep: best = min(best, ms.value)
                out.append(best)
        print(f"{basis} B={B}: lane-per-item {out[0]:.2f} ms, wave-per-item {out[1]:.2f} ms")
