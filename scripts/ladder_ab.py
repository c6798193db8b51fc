"""A/B of the one-pass halving ladder: run with FBX_LIBRARY pointing at a build with -DFBX_DBG_NOLADDER=1 and
without, each writing its results; `compare` prints the differences.  usage: ladder_ab.py run <out.npz> | compare a b"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'forest-benchmarking_amd'))
import numpy as np
if sys.argv[1] == 'run':
    from fbx import synthetic, tomography, _lib
    _lib.set_device(0)
    design, us, e, c = synthetic.process_batch(2, 'pauli', 1024)
    out = {}
    for mode, mi in (('fixed', 100), ('converge', 0)):
        choi, st = tomography.pgdb_process_estimate_batch(design, e, c, mode=mode, max_iters=mi, return_stats=True)
        out[mode + '_choi'] = choi
        for k in ('iterations', 'dykstra', 'backtracks', 'cost'):
            out[mode + '_' + k] = st[k]
    np.savez(sys.argv[2], **out)
else:
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    for mode in ('fixed', 'converge'):
        d = np.abs(a[mode + '_choi'] - b[mode + '_choi']).max(axis=(1, 2))
        print(mode, 'choi max diff %.2e, items differing %d' % (d.max(), (d > 0).sum()),
              '| iterations differ', int((a[mode + '_iterations'] != b[mode + '_iterations']).sum()),
              'dykstra differ', int((a[mode + '_dykstra'] != b[mode + '_dykstra']).sum()),
              'backtracks differ', int((a[mode + '_backtracks'] != b[mode + '_backtracks']).sum()))
