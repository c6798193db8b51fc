"""A/B of two builds of the library on ONE box, single-qubit PGDB on the lane-per-item kernel (2^20 items, Pauli in-basis, to
convergence): python scripts/ab_time1.py libA.so libB.so"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, ctypes
sys.path.insert(0, os.path.join(sys.argv[1], "forest-benchmarking_amd"))
import numpy as np
from fbx import synthetic, _lib
_lib.set_device(0); _lib.set_option("pgdb_packed_1q", 2.0)
B = 1 << 20
design, _, e, c = synthetic.process_batch(1, "pauli", 16384)
e = np.tile(e, (64, 1)); c = np.tile(c, (64, 1))
d_e, d_c = _lib.DeviceBuffer.from_array(e), _lib.DeviceBuffer.from_array(c)
d_choi = _lib.DeviceBuffer(B * 16 * 16)
ms = ctypes.c_double(); ts = []
for rep in range(4):
    _lib.check(_lib.lib().fbx_timer_begin())
    _lib.check(_lib.lib().fbx_pgdb_process_dev(design.handle, B, d_e.ptr, d_c.ptr, 1, _lib.MODE_CONVERGE, 0, d_choi.ptr, None, None, None, None, None))
    _lib.check(_lib.lib().fbx_timer_end(ctypes.byref(ms))); ts.append(ms.value)
print(min(ts[1:]))
'''
a, b = sys.argv[1], sys.argv[2]
for rnd in range(2):
    for lib in (a, b):
        out = subprocess.run([sys.executable, "-c", CHILD, ROOT], env=dict(os.environ, FBX_LIBRARY=os.path.join(ROOT, "forest-benchmarking_amd", lib)), capture_output=True, text=True)
        print(lib, out.stdout.strip(), out.stderr[-200:] if out.returncode else "")
