"""The basis guard: a build that damages every stored basis loaded for Dykstra iteration 1
(-DFBX_DBG_CORRUPT_BASIS -DFBX_DEBUG_REJECT, linked as forest-benchmarking_amd/libfbx_cor.so) must reject them
(Frobenius-norm test in front of the eigensolver) and give the results of the normal library."""
import sys, os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    os.environ["FBX_LIBRARY"] = os.path.join(ROOT, "forest-benchmarking_amd", sys.argv[1])
    sys.path.insert(0, os.path.join(ROOT, 'forest-benchmarking_amd'))
    import numpy as np
    from fbx import synthetic, tomography, _lib
    _lib.set_device(0)
    design, us, e, c = synthetic.process_batch(2, 'pauli', 256)
    choi, st = tomography.pgdb_process_estimate_batch(design, e, c, mode='converge', return_stats=True)
    np.savez(sys.argv[2], choi=choi, dyk=st['dykstra'], it=st['iterations'], bt=st['jacobi_sweeps'])
else:
    import numpy as np
    for lib, f in (("libfbx.so", "/tmp/p.npz"), ("libfbx_cor.so", "/tmp/q.npz")):
        subprocess.run([sys.executable, __file__, lib, f], check=True)
    p, q = np.load("/tmp/p.npz"), np.load("/tmp/q.npz")
    print('rejected bases (corrupting build):', int((q['bt'] // 1000000).sum()), 'in', int((q['bt'] >= 1000000).sum()), 'items')
    print('iterations equal', bool((p['it'] == q['it']).all()), 'dykstra equal', bool((p['dyk'] == q['dyk']).all()), 'choi max diff %.2e' % np.abs(p['choi'] - q['choi']).max())
