"""Transfer-inclusive rate of the host-pointer PGDB call on page-locked buffers against the HBM-resident launch of the same
items (2-qubit Pauli design, fixed 100 iterations; 2048 distinct experiments tiled)."""
import sys, os, time, ctypes
sys.path.insert(0, "forest-benchmarking_amd")
import numpy as np
from fbx import synthetic, tomography, _lib
_lib.set_device(0)
design, _, e0, c0 = synthetic.process_batch(2, "pauli", 2048)
ms = ctypes.c_double()
for B in (int(x) for x in (sys.argv[1:] or ["8192", "65536"])):
    reps = -(-B // 2048)
    e, c = np.tile(e0, (reps, 1))[:B], np.tile(c0, (reps, 1))[:B]
    d_e, d_c = _lib.DeviceBuffer.from_array(e), _lib.DeviceBuffer.from_array(c)
    d_choi = _lib.DeviceBuffer(B * 256 * 16)
    res = []
    for rep in range(3):
        _lib.check(_lib.lib().fbx_timer_begin())
        _lib.check(_lib.lib().fbx_pgdb_process_dev(design.handle, B, d_e.ptr, d_c.ptr, 1, _lib.MODE_FIXED, 100, d_choi.ptr, None, None, None, None, None))
        _lib.check(_lib.lib().fbx_timer_end(ctypes.byref(ms))); res.append(ms.value)
    ref = d_choi.to_array(np.complex128, (B, 16, 16))
    pe, pc = _lib.pinned_copy(e), _lib.pinned_copy(c)
    out = _lib.pinned_empty((B, 16, 16), np.complex128)
    ts = []
    for rep in range(5 if B > 16384 else 9):
        t0 = time.perf_counter()
        tomography.pgdb_process_estimate_batch(design, pe, pc, mode="fixed", max_iters=100, out=out)
        ts.append(time.perf_counter() - t0)
    same = np.array_equal(out, ref)
    r, h = min(res[1:]), 1e3 * float(np.median(ts[1:]))
    print(f"B={B}: resident {r:.2f} ms, host call (pinned, as dispatched) {h:.2f} ms = {100 * r / h:.1f} % of resident; identical results: {same}", flush=True)
    for chunk, name in ((1 << 20, "one upload, one launch, one download"), (2048, "pipelined, first / last stage 2048"), (4096, "pipelined, 4096"), (8192, "pipelined, 8192")):
        with _lib.option("pgdb_host_chunk", float(chunk)):
            ts = []
            for rep in range(4):
                t0 = time.perf_counter()
                tomography.pgdb_process_estimate_batch(design, pe, pc, mode="fixed", max_iters=100, out=out)
                ts.append(time.perf_counter() - t0)
        h2 = 1e3 * float(np.median(ts[1:]))
        print(f"      {name}: {h2:.2f} ms = {100 * r / h2:.1f} %  identical {np.array_equal(out, ref)}", flush=True)
    del pe, pc, out
    for b in (d_e, d_c, d_choi): b.free()
    _lib.release_workspace()
