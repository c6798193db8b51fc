"""Compare HIP PGDB against the oracle on N items starting at item FIRST: Choi difference and iteration /
Dykstra counts.  usage: check_counts.py N [FIRST]; FBX_CHECK_TOL=<tol> lists the items above a tighter tolerance."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'forest-benchmarking_amd')); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np
from fbx import synthetic, tomography, _lib
from fbx_oracle import design as od, estimators as oe
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
_lib.set_device(0)
design, us, e, c = synthetic.process_batch(2, 'pauli', N, first_item=first)
d = od.Design(2, 'process', design.in_labels, design.paulis, design.coefs)
A = oe.design_matrix_A(d)
for mode, mi in (('converge', 0), ('fixed', 100)):
    got, st = tomography.pgdb_process_estimate_batch(design, e, c, mode=mode, max_iters=mi, return_stats=True)
    bad = 0; worst = 0.0
    for b in range(N):
        w, ws = oe.pgdb_process_estimate(d, e[b], c[b], A=A, mode=mode, max_iters=mi, return_stats=True)
        diff = np.abs(got[b] - w).max(); worst = max(worst, diff)
        if st['iterations'][b] != ws['iterations'] or st['dykstra'][b] != ws['dykstra'] or diff > float(os.environ.get('FBX_CHECK_TOL', '1e-9')):
            bad += 1
            print('  item', first + b, 'diff %.2e' % diff, 'iters', st['iterations'][b], ws['iterations'], 'dyk', st['dykstra'][b], ws['dykstra'])
    print(mode, 'items', N, 'mismatching', bad, 'worst Choi diff %.2e' % worst)
