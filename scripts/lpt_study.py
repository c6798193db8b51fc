"""How well does the time a 2-qubit reconstruction spends in its first k outer iterations predict the rest?  Per-item wave cycles
of the two-waves kernel (profile build libfbx_prof.so) for k and for 100 fixed iterations; list scheduling on 2048 wave slots in
natural order against: iterations [0, k) in natural order, then [k, 100) longest-first by the measured early time.
usage: python scripts/lpt_study.py [B]"""
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
def makespan(order, dur, slots):
    h = [0.0] * slots
    heapq.heapify(h)
    for i in order:
        heapq.heappush(h, heapq.heappop(h) + dur[i])
    return max(h)
slots = 2048
tot = cycles(100)
ideal = tot.sum() / slots
print(f"B={B}: cycles per item max/mean {tot.max() / tot.mean():.3f}; natural order makespan / ideal {makespan(range(B), tot, slots) / ideal:.4f}; "
      f"longest-first with the true times {makespan(np.argsort(-tot), tot, slots) / ideal:.4f}")
for k in (5, 10, 15, 20, 30):
    early = cycles(k)
    late = tot - early
    m1 = makespan(range(B), early, slots)
    print(f"split at {k:2d}: early share {early.sum() / tot.sum():.3f}  corr(early, late) {np.corrcoef(early, late)[0, 1]:.3f};  "
          f"(launch 1 natural + launch 2 longest-first by early time) / ideal {(m1 + makespan(np.argsort(-early), late, slots)) / ideal:.4f};  "
          f"launch 2 with the true late times {(m1 + makespan(np.argsort(-late), late, slots)) / ideal:.4f}")
