import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'forest-benchmarking_amd'))
import numpy as np
from fbx import synthetic, tomography, _lib
_lib.set_device(0)
basis = sys.argv[1] if len(sys.argv) > 1 else 'sic'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
design, us, e, c = synthetic.process_batch(3, basis, 32)
e = np.tile(e, (B // 32, 1)); c = np.tile(c, (B // 32, 1))
for mode, mi in (('converge', 0), ('fixed', 100)):
    t = time.time()
    got, st = tomography.pgdb_process_estimate_batch(design, e, c, mode=mode, max_iters=mi, return_stats=True)
    dt = time.time() - t
    print(basis, mode, 'B', B, 'time %.2f s' % dt, 'recon/s %.1f' % (B / dt), 'iters mean %.1f' % st['iterations'].mean(),
          'dyk mean %.1f' % st['dykstra'].mean(), 'bt mean %.1f' % st['backtracks'].mean())
