import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'forest-benchmarking_amd')); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np
from fbx import synthetic, tomography, _lib
from fbx_oracle import design as od, estimators as oe
_lib.set_device(0)
basis = sys.argv[1] if len(sys.argv) > 1 else 'sic'
design, us, e, c = synthetic.process_batch(3, basis, 2)
print('m', design.m)
t = time.time(); d = od.Design(3, 'process', design.in_labels, design.paulis, design.coefs); A = oe.design_matrix_A(d); print('A built', A.shape, time.time() - t)
for mi in (1, 2, 4):
    t = time.time()
    got, st = tomography.pgdb_process_estimate_batch(design, e, c, mode='fixed', max_iters=mi, return_stats=True)
    tg = time.time() - t
    t = time.time()
    w, ws = oe.pgdb_process_estimate(d, e[0], c[0], A=A, mode='fixed', max_iters=mi, return_stats=True)
    print('iters', mi, 'gpu %.2fs oracle %.1fs' % (tg, time.time() - t), 'maxdiff %.2e' % np.abs(got[0] - w).max(), 'dyk', st['dykstra'][0], ws['dykstra'], 'bt', st['backtracks'][0], ws['backtracks'], 'cost', st['cost'][0], ws['cost'])
