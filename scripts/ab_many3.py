"""Same-box timing of several builds of the library on the 3-qubit bench workload (256 DISTINCT items, SIC, 100 fixed iterations):
python scripts/ab_many3.py libA.so libB.so ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, ctypes
sys.path.insert(0, os.path.join(sys.argv[1], "forest-benchmarking_amd"))
import numpy as np
from fbx import synthetic, _lib
_lib.set_device(0)
B = 256
design, _, e, c = synthetic.process_batch(3, sys.argv[2], B)
d_e, d_c = _lib.DeviceBuffer.from_array(e), _lib.DeviceBuffer.from_array(c)
d_choi = _lib.DeviceBuffer(B * 4096 * 16)
ms = ctypes.c_double(); ts = []
for rep in range(3):
    _lib.check(_lib.lib().fbx_timer_begin())
    _lib.check(_lib.lib().fbx_pgdb_process_dev(design.handle, B, d_e.ptr, d_c.ptr, 1, _lib.MODE_FIXED, 100, d_choi.ptr, None, None, None, None, None))
    _lib.check(_lib.lib().fbx_timer_end(ctypes.byref(ms))); ts.append(ms.value)
import hashlib
print(min(ts[1:]), hashlib.md5(d_choi.to_array(np.float64, (B * 4096 * 2,)).tobytes()).hexdigest()[:8])
'''
libs = [a for a in sys.argv[1:] if a.endswith(".so")]
basis = "pauli" if "--pauli" in sys.argv else "sic"
for rnd in range(2):
    for lib in libs:
        out = subprocess.run([sys.executable, "-c", CHILD, ROOT, basis], env=dict(os.environ, FBX_LIBRARY=os.path.join(ROOT, "forest-benchmarking_amd", lib)), capture_output=True, text=True)
        print(f"{lib:28s} {basis} ms, md5(choi): {out.stdout.strip()}", out.stderr[-300:] if out.returncode else "", flush=True)
