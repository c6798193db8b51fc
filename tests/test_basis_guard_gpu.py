"""The stored-basis guard of the PGDB kernels: a test-only build of the library
(libfbx_cor.so: -DFBX_DBG_CORRUPT_BASIS -DFBX_DEBUG_REJECT) damages every eigenvector basis that is loaded
from the HBM store for Dykstra iteration 1.  The damaged bases must be rejected by the Frobenius-norm test
in front of the eigensolver, and the reconstruction must come out as with the product library."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "forest-benchmarking_amd")

_CHILD = r"""
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from fbx import synthetic, tomography, _lib
_lib.set_device(0)
design, us, e, c = synthetic.process_batch(2, 'pauli', 96)
choi, st = tomography.pgdb_process_estimate_batch(design, e, c, mode='converge', return_stats=True)
np.savez(sys.argv[2], choi=choi, dyk=st['dykstra'], it=st['iterations'], sw=st['backtracks'])
"""


def _run(lib, out, debug_sweeps):
    env = dict(os.environ, FBX_LIBRARY=os.path.join(PKG, lib))
    env.pop("FBX_DEBUG_SWEEPS", None)
    if debug_sweeps:
        env["FBX_DEBUG_SWEEPS"] = "1"          # the backtracks output carries the sweep counter instead
    subprocess.run([sys.executable, "-c", _CHILD, PKG, out], check=True, env=env, timeout=300)
    return np.load(out)


def test_damaged_bases_are_rejected_and_results_unchanged(gpu, tmp_path):
    if not os.path.exists(os.path.join(PKG, "libfbx_cor.so")):
        pytest.skip("libfbx_cor.so not built (python forest-benchmarking_amd/build.py --guard-test)")
    good = _run("libfbx.so", str(tmp_path / "good.npz"), False)
    bad = _run("libfbx_cor.so", str(tmp_path / "bad.npz"), True)
    rejected = bad["sw"] // 1000000             # FBX_DEBUG_REJECT adds 1e6 to the sweep counter per rejection
    assert rejected.sum() > 100 and (rejected > 0).mean() > 0.3
    assert np.array_equal(good["it"], bad["it"]) and np.array_equal(good["dyk"], bad["dyk"])
    assert np.abs(good["choi"] - bad["choi"]).max() < 1e-11
