"""Per-phase cycle breakdown of the PGDB kernel (needs libfbx_prof.so: build.py --profile)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["FBX_LIBRARY"] = os.path.join(ROOT, "forest-benchmarking_amd", os.environ.get("FBX_PROF_LIB", "libfbx_prof.so"))
sys.path.insert(0, os.path.join(ROOT, 'forest-benchmarking_amd'))
import numpy as np
from fbx import synthetic, tomography, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
mode = sys.argv[2] if len(sys.argv) > 2 else 'fixed'
design, us, e, c = synthetic.process_batch(int(os.environ.get('AB_NQ', '2')), os.environ.get('AB_BASIS', 'pauli'), B)     # AB_NQ=3 AB_BASIS=sic: the 3-qubit kernel
_lib.set_device(0)
lib = _lib.lib()
if os.environ.get('AB_PIECES'):          # AB_PIECES=1: whole reconstructions in the two-waves kernel (its phase timers do not follow pieces)
    _lib.set_option('pgdb_pieces', float(os.environ['AB_PIECES']))
buf = _lib.DeviceBuffer(B * 8 * 8)
lib.fbx_debug_set_phase_buffer.argtypes = [ctypes.c_void_p]
lib.fbx_debug_set_phase_buffer(buf.ptr)
import time
for rep in range(2):
    t = time.time()
    choi, st = tomography.pgdb_process_estimate_batch(design, e, c, mode=mode, max_iters=100 if mode == 'fixed' else 0, return_stats=True)
    dt = time.time() - t
ph = buf.to_array(np.int64, (B, 8))
names = ['jacobi', 'reconstruct', 'hermitize+wait', 'transforms', 'gradient', 'linesearch', 'store+TP-proj', 'stop-test']
tot = ph.sum(1)
print('B', B, mode, 'time %.1f ms' % (1e3 * dt), 'recon/s %.0f' % (B / dt))
print('cycles/item mean %.3e max %.3e' % (tot.mean(), tot.max()))
print('slowest items', np.argsort(-tot)[:6], (np.sort(tot)[::-1][:6] / 1e6).round(1), 'Mcycles; p50 %.1f p90 %.1f p99 %.1f' % tuple(np.percentile(tot, [50, 90, 99]) / 1e6))
for i, n in enumerate(names):
    print('  %-14s mean %.3e (%.1f%%)  max-item share %.3e' % (n, ph[:, i].mean(), 100 * ph[:, i].sum() / tot.sum(), ph[tot.argmax(), i]))
print('per eigh jacobi cycles %.0f ; dykstra iters mean %.1f max %d; backtracks mean %.1f max %d' % (
    ph[:, 0].sum() / st['dykstra'].sum(), st['dykstra'].mean(), st['dykstra'].max(), st['backtracks'].mean(), st['backtracks'].max()))
print('per cost eval cycles %.0f' % (ph[:, 5].sum() / (st['backtracks'].sum() + 100 * B)))
if os.environ.get("FBX_PHASE_SAVE"):
    # per-item wave cycles + a list-scheduling model of the launch: 2 x 1024 wave slots (two-waves kernel) or 1024 (one-wave kernel)
    import heapq
    np.save(os.environ["FBX_PHASE_SAVE"], ph)
    def makespan(order, dur, slots):
        h = [0.0] * slots
        heapq.heapify(h)
        for i in order:
            heapq.heappush(h, heapq.heappop(h) + dur[i])
        return max(h)
    slots = 2048 if B >= 2048 else 1024
    ideal = tot.sum() / slots
    print('cycles per item: max/mean %.3f p99/mean %.3f p90/mean %.3f' % (tot.max() / tot.mean(), np.percentile(tot, 99) / tot.mean(), np.percentile(tot, 90) / tot.mean()))
    print('list scheduling on %d slots: natural order makespan / ideal %.4f; longest first %.4f' % (slots, makespan(range(B), tot, slots) / ideal, makespan(np.argsort(-tot), tot, slots) / ideal))
