"""Throughput of the pairwise conversions and projections on device-resident batches (diagnostic)."""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'forest-benchmarking_amd'))
import numpy as np
from fbx import synthetic, _lib
_lib.set_device(0)
lib = _lib.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
D = 4 ** n
B = {1: 1000000, 2: 200000, 3: 32768}[n]
ks = synthetic.kraus_batch(n, 4, 4096, seed=1)
ks = np.ascontiguousarray(np.tile(ks, (B // 4096 + 1, 1, 1, 1))[:B])
d_k = _lib.DeviceBuffer.from_array(ks)
bufs = {r: _lib.DeviceBuffer(B * D * D * 16) for r in ("choi", "superop", "ptm", "chi", "tmp")}
ms = ctypes.c_double()
def timed(name, fn):
    fn(); best = 1e9
    for _ in range(3):
        _lib.check(lib.fbx_timer_begin()); fn(); _lib.check(lib.fbx_timer_end(ctypes.byref(ms))); best = min(best, ms.value)
    print('%-28s %.2f ms  %.2e items/s' % (name, best, B / best * 1e3))
R = {"kraus": _lib.REP_KRAUS, "choi": _lib.REP_CHOI, "superop": _lib.REP_SUPEROP, "ptm": _lib.REP_PAULI_LIOUVILLE, "chi": _lib.REP_CHI}
timed("kraus -> choi", lambda: _lib.check(lib.fbx_convert_dev(R["kraus"], R["choi"], n, B, d_k.ptr, 4, bufs["choi"].ptr)))
timed("kraus -> ptm", lambda: _lib.check(lib.fbx_convert_dev(R["kraus"], R["ptm"], n, B, d_k.ptr, 4, bufs["ptm"].ptr)))
timed("choi -> ptm", lambda: _lib.check(lib.fbx_convert_dev(R["choi"], R["ptm"], n, B, bufs["choi"].ptr, 0, bufs["tmp"].ptr)))
timed("choi -> chi (eigh route)", lambda: _lib.check(lib.fbx_convert_dev(R["choi"], R["chi"], n, B, bufs["choi"].ptr, 0, bufs["tmp"].ptr)))
timed("superop -> ptm", lambda: _lib.check(lib.fbx_convert_dev(R["superop"], R["ptm"], n, B, bufs["choi"].ptr, 0, bufs["tmp"].ptr)))
timed("ptm -> superop", lambda: _lib.check(lib.fbx_convert_dev(R["ptm"], R["superop"], n, B, bufs["ptm"].ptr, 0, bufs["tmp"].ptr)))
timed("ptm -> choi", lambda: _lib.check(lib.fbx_convert_dev(R["ptm"], R["choi"], n, B, bufs["ptm"].ptr, 0, bufs["tmp"].ptr)))
timed("proj physical (TP)", lambda: _lib.check(lib.fbx_proj_choi_dev(_lib.PROJ_PHYSICAL_TP, n, B, bufs["ptm"].ptr, bufs["tmp"].ptr, None)))
timed("proj CP", lambda: _lib.check(lib.fbx_proj_choi_dev(_lib.PROJ_CP, n, B, bufs["ptm"].ptr, bufs["tmp"].ptr, None)))
