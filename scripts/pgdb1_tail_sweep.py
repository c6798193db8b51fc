import sys, os, ctypes
sys.path.insert(0, "forest-benchmarking_amd")
import numpy as np
from fbx import synthetic, _lib
_lib.set_device(0)
_lib.set_option("pgdb_packed_1q", 2.0)
ms = ctypes.c_double()
B = 1 << 20
for basis in ("pauli", "sic"):
    design, us, e0, c0 = synthetic.process_batch(1, basis, 16384)
    d_e, d_c = _lib.DeviceBuffer.from_array(np.tile(e0, (64, 1))), _lib.DeviceBuffer.from_array(np.tile(c0, (64, 1)))
    bufs = [_lib.DeviceBuffer(B * 32 * 8), _lib.DeviceBuffer(B * 4)]
    for tail in (1024, 2048, 4096, 8192, 16384, 32768, 65536):
        for chk in (4, 8, 16):
            os.environ["FBX_P1_TAIL"] = str(tail); os.environ["FBX_P1_CHECK"] = str(chk)
            ts = []
            for rep in range(4):
                _lib.check(_lib.lib().fbx_timer_begin())
                _lib.check(_lib.lib().fbx_pgdb_process_dev(design.handle, B, d_e.ptr, d_c.ptr, 1, _lib.MODE_CONVERGE, 0, bufs[0].ptr, bufs[1].ptr, None, None, None, None))
                _lib.check(_lib.lib().fbx_timer_end(ctypes.byref(ms)))
                if rep: ts.append(ms.value)
            print(f"{basis} tail={tail} check={chk}: {min(ts):.2f} ms", flush=True)
