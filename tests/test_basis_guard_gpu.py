"""The stored-basis guard of the PGDB kernels (docs/history/DESIGN_rounds1-4.md 5.7): a test-only build of the library
(libfbx_cor.so: -DFBX_DBG_CORRUPT_BASIS -DFBX_DEBUG_REJECT -DFBX_DIAGNOSTICS) damages every eigenvector basis that is loaded
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
np.savez(sys.argv[2], choi=choi, dyk=st['dykstra'], it=st['iterations'], sw=st['jacobi_sweeps'])
"""


def _run(lib, out):
    env = dict(os.environ, FBX_LIBRARY=os.path.join(PKG, lib))
    subprocess.run([sys.executable, "-c", _CHILD, PKG, out], check=True, env=env, timeout=300)
    return np.load(out)


def test_damaged_bases_are_rejected_and_results_unchanged(gpu, tmp_path):
    if not os.path.exists(os.path.join(PKG, "libfbx_cor.so")):
        pytest.skip("libfbx_cor.so not built (python forest-benchmarking_amd/build.py --guard-test)")
    good = _run("libfbx.so", str(tmp_path / "good.npz"))
    bad = _run("libfbx_cor.so", str(tmp_path / "bad.npz"))
    rejected = bad["sw"] // 1000000             # FBX_DEBUG_REJECT adds 1e6 to the Jacobi-sweep counter per rejection
    assert (good["sw"] < 1000000).all() and (good["sw"] > 0).all()
    assert rejected.sum() > 100 and (rejected > 0).mean() > 0.3
    assert np.array_equal(good["it"], bad["it"]) and np.array_equal(good["dyk"], bad["dyk"])
    assert np.abs(good["choi"] - bad["choi"]).max() < 1e-11


_LOG_CHILD = r"""
import ctypes, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from fbx import _lib
_lib.set_device(0)
lib = _lib.lib()
rs = np.random.RandomState(0)
x = np.concatenate([10.0 ** rs.uniform(-6, 0.31, 200000), [1e-6, 1.0, 0.5, 0.70710678118654752, 2.0, 1.5]])
out = np.empty_like(x)
lib.fbx_debug_log.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
_lib.check(lib.fbx_debug_log(x.ctypes.data, out.ctypes.data, x.size))
np.savez(sys.argv[2], x=x, out=out)
"""


def test_device_log_is_accurate_to_an_ulp(gpu, tmp_path):
    """The line search uses its own natural log (argument reduction + minimax polynomial).  The hook that
    exposes it on host arrays (fbx_debug_log) exists in diagnostics builds only, not in libfbx.so."""
    if not os.path.exists(os.path.join(PKG, "libfbx_cor.so")):
        pytest.skip("libfbx_cor.so not built")
    assert not hasattr(gpu.lib(), "fbx_debug_log")
    out = str(tmp_path / "log.npz")
    env = dict(os.environ, FBX_LIBRARY=os.path.join(PKG, "libfbx_cor.so"))
    subprocess.run([sys.executable, "-c", _LOG_CHILD, PKG, out], check=True, env=env, timeout=300)
    z = np.load(out)
    x, got = z["x"], z["out"]
    want = np.log(x)
    ulp = np.spacing(np.abs(want)) + 1e-300
    assert np.max(np.abs(got - want) / np.maximum(ulp, 2.3e-16 * np.abs(x - 1))) <= 1.5
    assert got[-5] == 0.0
