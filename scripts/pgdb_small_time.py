"""Device-resident PGDB throughput of the small designs: 1 qubit (Pauli / SIC in-basis) and 2 qubits with the SIC in-basis.
(1-qubit fixed-100 is dominated by the rare items whose Dykstra projection needs hundreds of iterations -- in the
reference too: bench item 122 takes 26 520 Dykstra iterations in 100 outer iterations in the oracle and in the kernel.)"""
import sys, os, ctypes, time
sys.path.insert(0, "forest-benchmarking_amd")
import numpy as np
from fbx import synthetic, tomography, _lib
_lib.set_device(0)
for n, basis, B in ((1, "pauli", 16384), (1, "sic", 16384), (2, "sic", 8192)):
    design, _, e, c = synthetic.process_batch(n, basis, 1024)
    e = np.tile(e, (B // 1024, 1)); c = np.tile(c, (B // 1024, 1))
    d_e, d_c = _lib.DeviceBuffer.from_array(e), _lib.DeviceBuffer.from_array(c)
    D = 4 ** n
    d_choi = _lib.DeviceBuffer(B * D * D * 16)
    ms = ctypes.c_double()
    for mode, name in ((_lib.MODE_FIXED, "fixed-100"), (_lib.MODE_CONVERGE, "converge")):
        best = 1e9
        for rep in range(4):
            _lib.check(_lib.lib().fbx_timer_begin())
            _lib.check(_lib.lib().fbx_pgdb_process_dev(design.handle, B, d_e.ptr, d_c.ptr, 1, mode, 100 if mode == _lib.MODE_FIXED else 0, d_choi.ptr, None, None, None, None, None))
            _lib.check(_lib.lib().fbx_timer_end(ctypes.byref(ms)))
            if rep: best = min(best, ms.value)
        print(f"n={n} {basis} m={design.m} B={B} {name}: {best:.2f} ms  {B / best * 1e3:.0f} recon/s")
