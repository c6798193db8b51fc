"""Would cutting each 2-qubit reconstruction of an 8192-item launch into P pieces of 100 / P outer iterations (state carried in
HBM, pieces handed out from one queue: all first pieces, then all second pieces, ...) shorten the launch?  Per-item wave cycles of
the two-waves kernel (profile build) for 25 / 50 / 75 / 100 fixed iterations give the four quarter durations of every item; list
scheduling with the precedence constraint on 2048 wave slots.  usage: python scripts/piece_study.py [B]"""
import ctypes, os, sys, heapq
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["FBX_LIBRARY"] = os.path.join(ROOT, "forest-benchmarking_amd", "libfbx_prof.so")
sys.path.insert(0, os.path.join(ROOT, "forest-benchmarking_amd"))
import numpy as np
from fbx import synthetic, tomography, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
design, us, e, c = synthetic.process_batch(2, "pauli", B)
_lib.set_device(0)
lib = _lib.lib()
buf = _lib.DeviceBuffer(B * 8 * 8)
lib.fbx_debug_set_phase_buffer.argtypes = [ctypes.c_void_p]
lib.fbx_debug_set_phase_buffer(buf.ptr)
def cycles(iters):
    tomography.pgdb_process_estimate_batch(design, e, c, mode="fixed", max_iters=iters)
    return buf.to_array(np.int64, (B, 8)).sum(1).astype(float)
cum = np.stack([cycles(k) for k in (25, 50, 75, 100)], axis=1)
q = np.diff(np.concatenate([np.zeros((B, 1)), cum], axis=1), axis=1)          # [B, 4] quarter durations
np.save(os.path.join(ROOT, "gpurun_out", f"quarters_{B}.npy"), q)
def schedule(pieces, slots):
    """pieces [B, P]: queue order = piece-major; a piece starts when a slot is free AND its predecessor is done"""
    Bn, P = pieces.shape
    free = [0.0] * slots
    heapq.heapify(free)
    done = np.zeros(Bn)
    for p in range(P):
        for i in range(Bn):
            t = heapq.heappop(free)
            start = max(t, done[i])
            done[i] = start + pieces[i, p]
            heapq.heappush(free, done[i])
    return done.max()
slots = 2048
ideal = q.sum() / slots
for P, pcs in ((1, q.sum(1, keepdims=True)), (2, np.stack([q[:, 0] + q[:, 1], q[:, 2] + q[:, 3]], axis=1)), (4, q)):
    print(f"B={B} P={P}: makespan / ideal {schedule(pcs, slots) / ideal:.4f}   (piece max/mean {pcs.max() / pcs.mean():.2f})")
