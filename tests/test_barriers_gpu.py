"""The LDS-only workgroup barriers of the 3-qubit kernels (csrc/fbx_pgdb3.hip: FBX_LDS_ONLY_BARRIERS replaces every __syncthreads()
of that translation unit -- the shared block reductions and the 64 x 64 eigensolver included -- by fence(workgroup, local) +
s_barrier, so that basis write-backs and prefetches stay in flight across them).  That is only correct while no thread reads global
memory another thread of the launch wrote and while the direct-to-LDS basis prefetch is waited for explicitly.  libfbx_fullbar.so
is the same source with full barriers (-DFBX_FULL_BARRIERS): both builds must agree BIT FOR BIT, per-iteration traces included --
a data race through global memory or a missed wait shows up as a difference (reference loop: tomography.py:542-594 with the
projection of operator_tools/project_superoperators.py:87-144)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "forest-benchmarking_amd")

_CHILD = r"""
import sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from fbx import synthetic, tomography, _lib
from fbx.operator_tools import project_superoperators as ps
_lib.set_device(0)
out = {}
for basis, nb in (("sic", 24), ("pauli", 6)):
    design, us, e, c = synthetic.process_batch(3, basis, nb)
    for mode, iters in (("fixed", 40), ("converge", 0)):
        choi, st = tomography.pgdb_process_estimate_batch(design, e, c, mode=mode, max_iters=iters, return_stats=True, trace_iters=160)
        out[f"{basis}_{mode}_choi"] = choi
        for k in ("iterations", "dykstra", "backtracks", "jacobi_sweeps", "trace", "cost"):
            out[f"{basis}_{mode}_{k}"] = np.asarray(st[k])
rs = np.random.RandomState(5)
x = rs.randn(8, 64, 64) + 1j * rs.randn(8, 64, 64)
out["proj_physical"] = np.array([ps.proj_choi_to_physical(m) for m in x])
np.savez(sys.argv[2], **out)
"""


def _run(lib, out):
    env = dict(os.environ, FBX_LIBRARY=os.path.join(PKG, lib))
    subprocess.run([sys.executable, "-c", _CHILD, PKG, out], check=True, env=env, timeout=900)
    return np.load(out)


def test_lds_only_barriers_reproduce_full_barriers_bit_for_bit(gpu, tmp_path):
    if not os.path.exists(os.path.join(PKG, "libfbx_fullbar.so")):
        pytest.skip("libfbx_fullbar.so not built (python forest-benchmarking_amd/build.py --guard-test)")
    a = _run("libfbx.so", str(tmp_path / "lds_only.npz"))
    b = _run("libfbx_fullbar.so", str(tmp_path / "full.npz"))
    assert sorted(a.files) == sorted(b.files) and len(a.files) > 20
    for k in a.files:
        assert np.array_equal(a[k], b[k]), k
    assert (a["sic_fixed_iterations"] == 40).all() and a["sic_converge_dykstra"].min() > 100


def test_published_rotation_solver_reproduces_the_local_one_bit_for_bit(gpu, tmp_path):
    """Round 5 rebuilt the 64 x 64 eigensolver of the 3-qubit kernels (csrc/fbx_eigh64.hpp: rotations evaluated once and published,
    owner-indexed LDS layout, eigenvector rings in DPP rows, the eigenvector role one round behind) without changing one operation
    on one number: libfbx_localrot.so is the same source with the round-4 solver (-DFBX_EIGH64_LOCAL_ROTATIONS) and must agree BIT
    FOR BIT -- estimates, counters, per-iteration traces, costs and the batched projection (replaces scipy.linalg.eigh at
    operator_tools/project_superoperators.py:30)."""
    if not os.path.exists(os.path.join(PKG, "libfbx_localrot.so")):
        pytest.skip("libfbx_localrot.so not built (python forest-benchmarking_amd/build.py --guard-test)")
    a = _run("libfbx.so", str(tmp_path / "published.npz"))
    b = _run("libfbx_localrot.so", str(tmp_path / "local.npz"))
    assert sorted(a.files) == sorted(b.files) and len(a.files) > 20
    for k in a.files:
        assert np.array_equal(a[k], b[k]), k
