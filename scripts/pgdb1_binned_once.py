"""One binned single-qubit call (for rocprofv3 --kernel-trace: the per-launch durations of pgdb1_step_kernel).
usage: python scripts/pgdb1_binned_once.py [log2 B] [basis]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "forest-benchmarking_amd"))
import numpy as np
from fbx import synthetic, _lib
_lib.set_device(0)
_lib.set_option("pgdb_packed_1q", 2.0)
lb = int(sys.argv[1]) if len(sys.argv) > 1 else 20
basis = sys.argv[2] if len(sys.argv) > 2 else "pauli"
os.environ["FBX_P1_BINNED"] = "2"
B = 1 << lb
design, us, e0, c0 = synthetic.process_batch(1, basis, 16384)
reps = (B + 16383) // 16384
d_e, d_c = _lib.DeviceBuffer.from_array(np.tile(e0, (reps, 1))[:B]), _lib.DeviceBuffer.from_array(np.tile(c0, (reps, 1))[:B])
d_choi, d_it = _lib.DeviceBuffer(B * 32 * 8), _lib.DeviceBuffer(B * 4)
for rep in range(2):
    _lib.check(_lib.lib().fbx_pgdb_process_dev(design.handle, B, d_e.ptr, d_c.ptr, 1, _lib.MODE_CONVERGE, 0, d_choi.ptr, d_it.ptr, None, None, None, None))
    _lib.synchronize() if hasattr(_lib, "synchronize") else None
it = d_it.to_array(np.int32, (B,))
print("active items per step:", [int((it > t).sum()) for t in range(0, 60, 2)])
