"""A/B of two builds of the library on ONE box: alternate launches of each (separate processes), same inputs.
usage: python scripts/ab_time.py libA.so libB.so [B] [mode]      (AB_NQ=3 AB_BASIS=sic in the environment: another design)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, ctypes
sys.path.insert(0, os.path.join(sys.argv[1], "forest-benchmarking_amd"))
import numpy as np
from fbx import synthetic, _lib
_lib.set_device(0)
B = int(sys.argv[2]); mode = _lib.MODE_FIXED if sys.argv[3] == "fixed" else _lib.MODE_CONVERGE
NQ = int(os.environ.get("AB_NQ", "2")); D2 = 16 ** NQ
design, _, e, c = synthetic.process_batch(NQ, os.environ.get("AB_BASIS", "pauli"), min(B, 2048))  # (distinct items up to 2048, tiled beyond)
if B > 2048:
    e = np.tile(e, (B // 2048, 1)); c = np.tile(c, (B // 2048, 1))
d_e, d_c = _lib.DeviceBuffer.from_array(e), _lib.DeviceBuffer.from_array(c)
d_choi = _lib.DeviceBuffer(B * D2 * 16)
ms = ctypes.c_double(); ts = []
for rep in range(7 if NQ < 3 else 4):
    _lib.check(_lib.lib().fbx_timer_begin())
    _lib.check(_lib.lib().fbx_pgdb_process_dev(design.handle, B, d_e.ptr, d_c.ptr, 1, mode, 100 if mode else 0, d_choi.ptr, None, None, None, None, None))
    _lib.check(_lib.lib().fbx_timer_end(ctypes.byref(ms))); ts.append(ms.value)
print(min(ts[1:]), sorted(ts[1:])[len(ts[1:]) // 2])
'''
a, b = sys.argv[1], sys.argv[2]
B = sys.argv[3] if len(sys.argv) > 3 else "1024"
mode = sys.argv[4] if len(sys.argv) > 4 else "fixed"
res = {a: [], b: []}
for rnd in range(3):
    for lib in (a, b):
        out = subprocess.run([sys.executable, "-c", CHILD, ROOT, B, mode], env=dict(os.environ, FBX_LIBRARY=os.path.join(ROOT, "forest-benchmarking_amd", lib)),
                             capture_output=True, text=True).stdout.split()
        res[lib].append(float(out[0]))
for lib in (a, b):
    print(f"{lib:24s} B={B} {mode}: best of each round (ms): " + " ".join(f"{x:.2f}" for x in res[lib]))
