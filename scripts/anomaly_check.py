"""Round-1 anomaly hunt (docs/history/DESIGN_rounds1-4.md 5.7): the diagnostics build (libfbx_prof.so: phase timers) once produced, on 3 of
11 boxes and only in the FIRST launch of a process, reconstructions with far too many Dykstra iterations.  This runs
the first launch of a fresh process with each library and compares every counter and the Choi matrices.
usage: python scripts/anomaly_check.py   (prints one line per library; exit code 1 on a mismatch)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os
sys.path.insert(0, os.path.join(sys.argv[1], "forest-benchmarking_amd"))
import numpy as np
from fbx import synthetic, tomography, _lib
_lib.set_device(0)
design, _, e, c = synthetic.process_batch(2, "pauli", 256)
if "prof" in os.environ.get("FBX_LIBRARY", ""):
    import ctypes
    buf = _lib.DeviceBuffer(256 * 8 * 8)
    _lib.lib().fbx_debug_set_phase_buffer.argtypes = [ctypes.c_void_p]
    _lib.lib().fbx_debug_set_phase_buffer(buf.ptr)
first, s1 = tomography.pgdb_process_estimate_batch(design, e, c, mode="fixed", max_iters=100, return_stats=True)
second, s2 = tomography.pgdb_process_estimate_batch(design, e, c, mode="fixed", max_iters=100, return_stats=True)
np.savez(sys.argv[2], first=first, second=second, d1=s1["dykstra"], d2=s2["dykstra"], w1=s1["jacobi_sweeps"], w2=s2["jacobi_sweeps"])
'''
import numpy as np
out = {}
for lib in ("libfbx.so", "libfbx_prof.so"):
    path = os.path.join(ROOT, "forest-benchmarking_amd", lib)
    if not os.path.exists(path):
        print(lib, "not built"); continue
    fn = f"/tmp/anomaly_{lib}.npz"
    subprocess.run([sys.executable, "-c", CHILD, ROOT, fn], env=dict(os.environ, FBX_LIBRARY=path), check=True)
    out[lib] = np.load(fn)
    z = out[lib]
    print(f"{lib:16s} first launch: mean Dykstra {z['d1'].mean():.2f} max {z['d1'].max()}  sweeps {z['w1'].mean():.1f}; second launch: "
          f"{z['d2'].mean():.2f} / {z['w2'].mean():.1f}; first == second: {np.array_equal(z['first'], z['second'])}")
bad = False
if len(out) == 2:
    a, b = out["libfbx.so"], out["libfbx_prof.so"]
    same = np.array_equal(a["d1"], b["d1"]) and np.abs(a["first"] - b["first"]).max() < 1e-9
    print("diagnostics build reproduces the product library's first launch:", same, " max |dChoi| %.1e" % np.abs(a["first"] - b["first"]).max())
    bad = not same
sys.exit(1 if bad else 0)
