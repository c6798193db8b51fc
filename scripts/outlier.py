import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'forest-benchmarking_amd'))
import numpy as np
from fbx import synthetic, tomography, _lib
_lib.set_device(0)
design, us, e, c = synthetic.process_batch(2, 'pauli', 1024)
for mode, mi in (('fixed', 100), ('converge', 0)):
    choi, st = tomography.pgdb_process_estimate_batch(design, e, c, mode=mode, max_iters=mi, return_stats=True)
    k = int(st['dykstra'].argmax())
    order = np.argsort(-st['dykstra'])[:6]
    print(mode, 'top items', order, 'dyk', st['dykstra'][order], 'iters', st['iterations'][order], 'sweeps', st['jacobi_sweeps'][order],
          'sweeps/eigh', (st['jacobi_sweeps'][order] / st['dykstra'][order]).round(2))
    print('  median dyk', np.median(st['dykstra']), 'median sweeps/eigh', np.median(st['jacobi_sweeps'] / st['dykstra']).round(2))
    u = us[k]
    ev = np.linalg.eigvals(u)
    print('  outlier', k, 'unitary eigenphases', np.sort(np.angle(ev)).round(3), 'min |e|', np.abs(e[k]).min().round(4), 'max |e|', np.abs(e[k]).max().round(4),
          'n(|e|==1):', int((np.abs(e[k]) == 1).sum()))
