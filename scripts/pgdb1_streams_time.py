"""The binned relaunch of the single-qubit kernel as interleaved pipelines (round 6): chunk size x number of streams, against
one pipeline on one stream (round 5's schedule) -- times and bit-identical outputs.  usage: python scripts/pgdb1_streams_time.py [log2 B]"""
import sys, os, ctypes
sys.path.insert(0, "forest-benchmarking_amd")
import numpy as np
from fbx import synthetic, _lib
_lib.set_device(0)
_lib.set_option("pgdb_packed_1q", 2.0)
ms = ctypes.c_double()
lb = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = 1 << lb
def run(design, d_e, d_c, bufs, mode, iters, env):
    for k, v in env.items(): os.environ[k] = str(v)
    ts = []
    for rep in range(4):
        _lib.check(_lib.lib().fbx_timer_begin())
        _lib.check(_lib.lib().fbx_pgdb_process_dev(design.handle, B, d_e.ptr, d_c.ptr, 1, mode, iters, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, bufs[3].ptr, bufs[4].ptr, None))
        _lib.check(_lib.lib().fbx_timer_end(ctypes.byref(ms)))
        if rep: ts.append(ms.value)
    out = (bufs[0].to_array(np.float64, (B, 32)), bufs[1].to_array(np.int32, (B,)), bufs[2].to_array(np.int32, (B,)), bufs[3].to_array(np.int32, (B,)), bufs[4].to_array(np.float64, (B,)))
    return min(ts), out
for basis in ("pauli", "sic"):
    design, us, e0, c0 = synthetic.process_batch(1, basis, 16384)
    reps = (B + 16383) // 16384
    d_e, d_c = _lib.DeviceBuffer.from_array(np.tile(e0, (reps, 1))[:B]), _lib.DeviceBuffer.from_array(np.tile(c0, (reps, 1))[:B])
    bufs = [_lib.DeviceBuffer(B * 32 * 8), _lib.DeviceBuffer(B * 4), _lib.DeviceBuffer(B * 4), _lib.DeviceBuffer(B * 4), _lib.DeviceBuffer(B * 8)]
    for mode, iters, name in ((_lib.MODE_CONVERGE, 0, "converge"), (_lib.MODE_FIXED, 30, "fixed-30")):
        t0, ref = run(design, d_e, d_c, bufs, mode, iters, {"FBX_P1_BINNED": 2, "FBX_P1_CHUNK": B, "FBX_P1_STREAMS": 1})
        print(f"1q {basis} B=2^{lb} {name}: one pipeline, one stream {t0:.2f} ms = {B / t0 * 1e3:.3g} /s", flush=True)
        for streams, div in ((1, 4), (2, 2), (2, 4), (2, 8), (2, 16)):
            t1, got = run(design, d_e, d_c, bufs, mode, iters, {"FBX_P1_BINNED": 2, "FBX_P1_CHUNK": B // div, "FBX_P1_STREAMS": streams})
            same = all(np.array_equal(a, b) for a, b in zip(ref, got))
            print(f"      {div:2d} chunks on {streams} stream(s): {t1:.2f} ms = {B / t1 * 1e3:.3g} /s  identical={same}", flush=True)
    for b in [d_e, d_c] + bufs: b.free()
