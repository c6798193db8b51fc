import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'forest-benchmarking_amd'))
import numpy as np
from fbx import synthetic, tomography, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
design, us, e, c = synthetic.process_batch(2, 'pauli', 64)
reps = (B + 63) // 64
e = np.tile(e, (reps, 1))[:B]; c = np.tile(c, (reps, 1))[:B]
_lib.set_device(0)
print(_lib.device_name())
for mode, mi in (('converge', 0), ('fixed', 100)):
    for rep in range(2):
        t = time.time()
        choi, st = tomography.pgdb_process_estimate_batch(design, e, c, mode=mode, max_iters=mi, return_stats=True)
        dt = time.time() - t
        print(mode, 'B', B, 'time %.3f s' % dt, 'recon/s %.1f' % (B / dt), 'iters mean', st['iterations'].mean(),
              'dyk mean', st['dykstra'].mean(), 'bt mean', st['backtracks'].mean())
