import sys, time, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'forest-benchmarking_amd'))
import numpy as np
from fbx import synthetic, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
n, K, D = 2, 4, 16
_lib.set_device(0)
lib = _lib.lib()
base = synthetic.kraus_batch(n, K, 4096, seed=1)
ks = np.ascontiguousarray(np.tile(base, (B // 4096 + 1, 1, 1, 1))[:B])
ref = np.ascontiguousarray(np.eye(D, dtype=np.complex128))
d_k = _lib.DeviceBuffer.from_array(ks); d_r = _lib.DeviceBuffer.from_array(ref)
d_c = _lib.DeviceBuffer(B * D * D * 16); d_p = _lib.DeviceBuffer(B * D * D * 16); d_x = _lib.DeviceBuffer(B * D * D * 16)
d_f = _lib.DeviceBuffer(B * 8)
ms = ctypes.c_double()
for rep in range(3):
    _lib.check(lib.fbx_timer_begin())
    _lib.check(lib.fbx_kraus_sweep_dev(n, B, K, d_k.ptr, d_r.ptr, d_c.ptr, d_p.ptr, d_x.ptr, d_f.ptr))
    _lib.check(lib.fbx_timer_end(ctypes.byref(ms)))
    bytes_ = B * (K * D * 16 + 3 * D * D * 16 + 8)
    print('B', B, 'ms %.3f' % ms.value, 'items/s %.3e' % (B / ms.value * 1e3), 'GB/s %.1f' % (bytes_ / ms.value / 1e6),
          'frac of 8 TB/s %.3f' % (bytes_ / ms.value / 1e6 / 8000))
