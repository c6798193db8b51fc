"""How much the lanes of a wavefront of the packed single-qubit kernel diverge: per-outer-iteration Dykstra counts and
halvings (fbx_pgdb_process_ex trace) of 16 384 experiments, grouped 64 to a wavefront as the kernel does."""
import sys
sys.path.insert(0, "forest-benchmarking_amd")
import numpy as np
from fbx import synthetic, tomography, _lib
_lib.set_device(0)
_lib.set_option('pgdb_packed_1q', 2.0)
for basis in ("pauli", "sic"):
    design, us, e, c = synthetic.process_batch(1, basis, 16384)
    T = 160
    got, st = tomography.pgdb_process_estimate_batch(design, e, c, return_stats=True, trace_iters=T)
    tr = st["trace"].astype(np.int64)            # [B, T, 2]
    it = st["iterations"]
    dyk = tr[:, :, 0].reshape(256, 64, T)
    bt = tr[:, :, 1].reshape(256, 64, T)
    alive = (np.arange(T)[None, :] < it[:, None]).reshape(256, 64, T)
    print(basis, "mean outer iterations", it.mean(), "max", it.max(), " mean over waves of max over lanes", it.reshape(256, 64).max(axis=1).mean())
    print("  Dykstra per outer iteration: mean", dyk[alive].mean(), " p50/p90/p99/max", np.percentile(dyk[alive], [50, 90, 99]), dyk.max())
    mx = dyk.max(axis=1)                           # [256, T] max over lanes per step (lanes aligned at step k: static assignment)
    any_alive = alive.any(axis=1)
    print("  per wave step: mean of max-over-lanes Dykstra", mx[any_alive].mean(), " sum over steps per wave (mean)", mx.sum(axis=1).mean(),
          " vs per-lane total mean", dyk.sum(axis=2).mean())
    print("  halvings per outer iteration: mean", bt[alive].mean(), " p90/p99/max", np.percentile(bt[alive], [90, 99]), bt.max(),
          "  max-over-lanes mean", bt.max(axis=1)[any_alive].mean())
    w = st["cost_evals"]; ps = st["power_sum_passes"]; sw = st["jacobi_sweeps"]
    print("  per item: full cost evaluations", w.mean(), " power-sum passes", ps.mean(), " sweeps per decomposition", sw.sum() / st["dykstra"].sum())
