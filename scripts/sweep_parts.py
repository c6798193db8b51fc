"""Which outputs cost what in the fused sweep kernel (diagnostic)."""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'forest-benchmarking_amd'))
import numpy as np
from fbx import synthetic, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
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
null = ctypes.c_void_p(0)
for name, (c, p, x) in {'all': (d_c.ptr, d_p.ptr, d_x.ptr), 'choi+ptm': (d_c.ptr, d_p.ptr, null), 'ptm only': (null, d_p.ptr, null),
                        'fidelity only': (null, null, null)}.items():
    best = 1e9
    for rep in range(4):
        _lib.check(lib.fbx_timer_begin())
        _lib.check(lib.fbx_kraus_sweep_dev(n, B, K, d_k.ptr, d_r.ptr, c, p, x, d_f.ptr))
        _lib.check(lib.fbx_timer_end(ctypes.byref(ms)))
        best = min(best, ms.value)
    nout = sum(1 for q in (c, p, x) if q is not null and getattr(q, 'value', 1))
    bytes_ = B * (K * D * 16 + nout * D * D * 16 + 8)
    print('%-14s best ms %.3f  GB/s %.1f' % (name, best, bytes_ / best / 1e6))
