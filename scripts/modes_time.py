"""Throughput of the 2-qubit PGDB path by mode and batch size on distinct items (inputs resident)."""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'forest-benchmarking_amd'))
import numpy as np
from fbx import synthetic, _lib
_lib.set_device(0)
lib = _lib.lib()
ms = ctypes.c_double()
for B in (1024, 2048, 4096, 8192, 16384):
    design, _, e, c = synthetic.process_batch(2, 'pauli', min(B, 4096))
    if B > 4096:
        e = np.tile(e, (B // 4096, 1)); c = np.tile(c, (B // 4096, 1))
    d_e, d_c = _lib.DeviceBuffer.from_array(e), _lib.DeviceBuffer.from_array(c)
    d_choi = _lib.DeviceBuffer(B * 256 * 16); d_it = _lib.DeviceBuffer(B * 4)
    for mode, mi, name in ((_lib.MODE_CONVERGE, 0, 'converge'), (_lib.MODE_FIXED, 100, 'fixed-100')):
        best = 1e9
        for rep in range(3):
            _lib.check(lib.fbx_timer_begin())
            _lib.check(lib.fbx_pgdb_process_dev(design.handle, B, d_e.ptr, d_c.ptr, 1, mode, mi, d_choi.ptr, d_it.ptr, None, None, None, None))
            _lib.check(lib.fbx_timer_end(ctypes.byref(ms)))
            best = min(best, ms.value)
        it = d_it.to_array(np.int32, (B,))
        print('B %5d %-9s %.1f ms  %.0f recon/s  (mean outer iterations %.1f)' % (B, name, best, B / best * 1e3, it.mean()))
