"""Line-search branch counts (needs libfbx_dbg.so built with -DFBX_DEBUG_LS): full cost evaluations per item."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["FBX_LIBRARY"] = os.path.join(ROOT, "forest-benchmarking_amd", "libfbx_dbg.so")
os.environ["FBX_DEBUG_SWEEPS"] = "1"
sys.path.insert(0, os.path.join(ROOT, 'forest-benchmarking_amd'))
import numpy as np
from fbx import synthetic, tomography, _lib
_lib.set_device(0)
design, us, e, c = synthetic.process_batch(2, 'pauli', 1024)
choi, st = tomography.pgdb_process_estimate_batch(design, e, c, mode='fixed', max_iters=100, return_stats=True)
v = st['backtracks'].astype(np.int64)
full = v & 1023; not_listed = (v >> 10) & 1023; nan = v >> 20
os.environ.pop("FBX_DEBUG_SWEEPS")
for k in (636, 320, 818, 0, 1):
    print('item', k, 'full evals', full[k], 'of which list overflow', not_listed[k], 'rmax nan', nan[k])
print('mean full evals', full.mean(), 'mean overflow', not_listed.mean(), 'nan', nan.mean())
