"""Where a wavefront of the single-qubit lane-per-item kernel spends its cycles (profile build: python
forest-benchmarking_amd/build.py --profile).
usage: python scripts/pgdb1_phase_profile.py [log2 B] [basis]; the library prints one P1PROF line per call on stderr."""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["FBX_LIBRARY"] = os.path.join(ROOT, "forest-benchmarking_amd", "libfbx_prof.so")
sys.path.insert(0, os.path.join(ROOT, "forest-benchmarking_amd"))
import numpy as np
from fbx import synthetic, _lib
_lib.set_device(0)
_lib.set_option("pgdb_packed_1q", 2.0)
lb = int(sys.argv[1]) if len(sys.argv) > 1 else 20
basis = sys.argv[2] if len(sys.argv) > 2 else "pauli"
B = 1 << lb
design, us, e0, c0 = synthetic.process_batch(1, basis, 16384)
reps = (B + 16383) // 16384
d_e, d_c = _lib.DeviceBuffer.from_array(np.tile(e0, (reps, 1))[:B]), _lib.DeviceBuffer.from_array(np.tile(c0, (reps, 1))[:B])
d_choi, d_it = _lib.DeviceBuffer(B * 32 * 8), _lib.DeviceBuffer(B * 4)
ms = ctypes.c_double()
for binned in (0, 2):
    for mode, iters, name in ((_lib.MODE_CONVERGE, 0, "converge"), (_lib.MODE_FIXED, 30, "fixed-30")):
        os.environ["FBX_P1_BINNED"] = str(binned)
        for rep in range(1):
            _lib.check(_lib.lib().fbx_timer_begin())
            _lib.check(_lib.lib().fbx_pgdb_process_dev(design.handle, B, d_e.ptr, d_c.ptr, 1, mode, iters, d_choi.ptr, d_it.ptr, None, None, None, None))
            _lib.check(_lib.lib().fbx_timer_end(ctypes.byref(ms)))
        sys.stderr.flush()
        print(f"## {basis} B=2^{lb} {name} binned={binned}: {ms.value:.2f} ms (profile build; the P1PROF lines above are this call)", file=sys.stderr, flush=True)
