import sys, os
sys.path.insert(0, "forest-benchmarking_amd")
import numpy as np
from fbx import synthetic, tomography, _lib
_lib.set_device(0)
design, _, e, c = synthetic.process_batch(2, "pauli", 1024)
for K in (50, 100):
    _, st = tomography.pgdb_process_estimate_batch(design, e, c, mode="fixed", max_iters=K, return_stats=True)
    print(os.environ.get("FBX_LIBRARY","libfbx.so")[-18:], "K", K, "mean sweeps %.1f dyk %.1f bt %.1f terms %.1f cost_evals %.1f sums %.1f | item636 sweeps %d dyk %d bt %d" % (
        st["jacobi_sweeps"].mean(), st["dykstra"].mean(), st["backtracks"].mean(), st["eig_terms"].mean(), st["cost_evals"].mean(), st["power_sum_passes"].mean(),
        st["jacobi_sweeps"][636], st["dykstra"][636], st["backtracks"][636]))
