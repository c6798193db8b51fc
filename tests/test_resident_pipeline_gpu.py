"""A whole pipeline on device pointers -- raw shots -> observable moments -> PGDB process estimates ->
physical projection -> Pauli transfer matrices -> process fidelity against the ideal gate -- with one
upload of the bit arrays and one download of the fidelities, checked against the host-pointer entry
points step by step (bitstrings go straight to Choi matrices on the device)."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_shots_to_fidelity_without_leaving_hbm(gpu):
    from fbx import _lib, synthetic, tomography
    from fbx import distance_measures as dm
    from fbx.design import process_design
    from fbx.operator_tools import convert_batch
    from fbx.operator_tools.project_superoperators import proj_choi_batch
    from fbx.observable_estimation import shots_to_obs_moments_batch
    lib = _lib.lib()
    n, B, shots = 1, 6, 400
    design = process_design(n, "sic")
    m, D = design.m, 4 ** n
    us = np.array([synthetic.haar_unitary(2, np.random.RandomState(300 + b)) for b in range(B)])
    exact = synthetic.exact_process_expectations(design, us, 0.0)
    rng = np.random.default_rng(5)
    # one-qubit settings: a single measured bit per shot, P(bit = 0) = (1 + e) / 2
    bits = (rng.random((B * m, shots, n)) >= ((1 + exact.reshape(-1)) / 2)[:, None, None]).astype(np.uint8)
    masks = np.ones((B * m, n), np.uint8)

    # ---- resident path
    d_bits, d_masks = _lib.DeviceBuffer.from_array(bits), _lib.DeviceBuffer.from_array(masks)
    d_mean, d_var = _lib.DeviceBuffer(B * m * 8), _lib.DeviceBuffer(B * m * 8)
    _lib.check(lib.fbx_shots_to_moments_dev(n, B * m, shots, d_bits.ptr, d_masks.ptr, None, 0, d_mean.ptr, d_var.ptr))
    d_counts = _lib.DeviceBuffer.from_array(np.full((B, m), float(shots)))
    d_choi, d_proj, d_ptm = (_lib.DeviceBuffer(B * D * D * 16) for _ in range(3))
    d_it = _lib.DeviceBuffer(B * 4)
    _lib.check(lib.fbx_pgdb_process_dev(design.handle, B, d_mean.ptr, d_counts.ptr, 1, _lib.MODE_CONVERGE, 0,
                                        d_choi.ptr, d_it.ptr, None, None, None))
    _lib.check(lib.fbx_proj_choi_dev(_lib.PROJ_PHYSICAL_TP, n, B, d_choi.ptr, d_proj.ptr, None))
    _lib.check(lib.fbx_convert_dev(_lib.REP_CHOI, _lib.REP_PAULI_LIOUVILLE, n, B, d_proj.ptr, 0, d_ptm.ptr))
    ideal = convert_batch("kraus", "pauli_liouville", us[:, None])           # [B, D, D]
    d_ideal = _lib.DeviceBuffer.from_array(np.ascontiguousarray(ideal))
    d_fp = _lib.DeviceBuffer(B * 8)
    _lib.check(lib.fbx_process_fidelity_dev(n, B, d_ideal.ptr, d_ptm.ptr, None, d_fp.ptr))
    _lib.synchronize()
    fid = d_fp.to_array(np.float64, (B,))

    # ---- the same through the host-pointer entry points
    mean, var = shots_to_obs_moments_batch(bits, masks)
    assert np.array_equal(d_mean.to_array(np.float64, (B * m,)), mean)
    choi = tomography.pgdb_process_estimate_batch(design, mean.reshape(B, m), np.full((B, m), float(shots)))
    assert np.array_equal(d_choi.to_array(np.complex128, (B, D, D)), choi)
    proj = proj_choi_batch(_lib.PROJ_PHYSICAL_TP, choi)
    ptm = convert_batch("choi", "pauli_liouville", proj)
    want = dm.process_fidelity_batch(ideal, ptm)
    assert np.array_equal(fid, want)
    assert (fid > 0.9).all() and (fid <= 1.0 + 1e-9).all()
