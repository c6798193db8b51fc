"""Kernel time of the 1024-item blocks the 8 ranks of the weak-scaling bench own (diagnostic)."""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'forest-benchmarking_amd'))
import numpy as np
from fbx import synthetic, tomography, _lib
_lib.set_device(0)
lib = _lib.lib()
ms = ctypes.c_double()
for r in range(8):
    design, _, e, c = synthetic.process_batch(2, 'pauli', 1024, first_item=r * 1024)
    best = 1e9
    for rep in range(2):
        _lib.check(lib.fbx_timer_begin())
        choi, st = tomography.pgdb_process_estimate_batch(design, e, c, mode='fixed', max_iters=100, return_stats=True)
        _lib.check(lib.fbx_timer_end(ctypes.byref(ms)))
        best = min(best, ms.value)
    print('block', r, 'ms %.1f' % best, 'dyk mean %.0f max %d' % (st['dykstra'].mean(), st['dykstra'].max()))
