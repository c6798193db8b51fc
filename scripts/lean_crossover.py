import sys, os, ctypes
lib = sys.argv[1]
os.environ["FBX_LIBRARY"] = os.path.join("/root/repo/forest-benchmarking_amd", lib)
sys.path.insert(0, "forest-benchmarking_amd")
import numpy as np
from fbx import synthetic, _lib
_lib.set_device(0)
ms = ctypes.c_double()
design, us, e0, c0 = synthetic.process_batch(2, "pauli", 2048)
for B in (768, 1024, 1152, 1280, 1536, 2048):
    d_e, d_c = _lib.DeviceBuffer.from_array(e0[:B]), _lib.DeviceBuffer.from_array(c0[:B])
    d_choi = _lib.DeviceBuffer(B * 512 * 8)
    for mode, iters, name in ((_lib.MODE_FIXED, 100, "fixed-100"), (_lib.MODE_CONVERGE, 0, "converge")):
        best = 1e9
        for rep in range(4):
            _lib.check(_lib.lib().fbx_timer_begin())
            _lib.check(_lib.lib().fbx_pgdb_process_dev(design.handle, B, d_e.ptr, d_c.ptr, 1, mode, iters, d_choi.ptr, None, None, None, None, None))
            _lib.check(_lib.lib().fbx_timer_end(ctypes.byref(ms)))
            if rep: best = min(best, ms.value)
        print(f"{lib} B={B} {name}: {best:.2f} ms = {B / best:.1f} k/s", flush=True)
    for b in (d_e, d_c, d_choi): b.free()
