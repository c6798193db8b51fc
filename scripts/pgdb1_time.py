"""Throughput of single-qubit PGDB: the lane-per-item kernel (csrc/fbx_pgdb1.hip, default) against the wavefront-per-item
kernel it replaces, resident inputs, distinct experiments; and the single-qubit process-fidelity bootstrap."""
import sys, os, ctypes, time
sys.path.insert(0, "forest-benchmarking_amd")
import numpy as np
from fbx import synthetic, tomography, _lib
_lib.set_device(0)
ms = ctypes.c_double()
NMAX = int(os.environ.get("FBX_P1_NMAX", 262144))
for basis in ("pauli", "sic"):
    design, us, e_all, c_all = synthetic.process_batch(1, basis, 16384)
    for B in (1024, 16384, 65536, NMAX):
        reps = (B + 16383) // 16384
        e = np.tile(e_all, (reps, 1))[:B]; c = np.tile(c_all, (reps, 1))[:B]
        d_e, d_c = _lib.DeviceBuffer.from_array(e), _lib.DeviceBuffer.from_array(c)
        d_choi = _lib.DeviceBuffer(B * 16 * 16)
        d_it = _lib.DeviceBuffer(B * 4)
        for packed in (2.0, 0.0):
            with _lib.option("pgdb_packed_1q", packed):
                for mode, name in ((_lib.MODE_CONVERGE, "converge"), (_lib.MODE_FIXED, "fixed-100")):
                    if mode == _lib.MODE_FIXED and B > 65536:
                        continue
                    best = 1e9
                    for rep in range(3):
                        _lib.check(_lib.lib().fbx_timer_begin())
                        _lib.check(_lib.lib().fbx_pgdb_process_dev(design.handle, B, d_e.ptr, d_c.ptr, 1, mode, 100 if mode == _lib.MODE_FIXED else 0, d_choi.ptr, d_it.ptr, None, None, None, None))
                        _lib.check(_lib.lib().fbx_timer_end(ctypes.byref(ms)))
                        if rep: best = min(best, ms.value)
                    it = d_it.to_array(np.int32, (B,)) if hasattr(d_it, "to_array") else None
                    extra = f"  mean/max outer iterations {it.mean():.1f}/{it.max()}" if it is not None else ""
                    print(f"1q {basis} m={design.m} B={B} {'packed' if packed else 'wave  '} {name}: {best:.2f} ms  {B / best * 1e3:.3g} recon/s{extra}", flush=True)
# the single-qubit bootstrap (process_fidelity_variance_batch): 1024 experiments x 40 resamples
design, us, e, c = synthetic.process_batch(1, "pauli", 1024)
from fbx.operator_tools import superoperator_transformations as st
target = np.array([st.kraus2pauli_liouville(u) for u in us])
for packed in (2.0, 1.0, 0.0):
    with _lib.option("pgdb_packed_1q", packed):
        for rep in range(3):
            t0 = time.perf_counter()
            mean, var = tomography.process_fidelity_variance_batch(design, e, c, target, n_resamples=40, seed=1)
            dt = time.perf_counter() - t0
        print(f"bootstrap 1q pauli 1024 x 40 resamples { {2.0: 'packed', 1.0: 'default', 0.0: 'wave  '}[packed] }: {dt * 1e3:.1f} ms = {1024 * 40 / dt:.3g} recon/s  mean fidelity {mean.mean():.6f}", flush=True)
