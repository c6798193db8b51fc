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
                                        d_choi.ptr, d_it.ptr, None, None, None, None))
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


def test_state_bootstrap_without_leaving_hbm(gpu):
    """expectations -> Beta resamples -> iterative MLE -> physical projection -> fidelity / purity, all on
    device pointers; every step equals the host-pointer entry point on the downloaded intermediate."""
    from fbx import _lib, synthetic, tomography
    from fbx import distance_measures as dm
    from fbx.operator_tools.project_state_matrix import project_state_matrix_to_physical_batch
    lib, DB = _lib.lib(), _lib.DeviceBuffer
    n, B, R = 2, 3, 10
    design, rhos, e, c = synthetic.state_batch(n, B, shots=300, mixed=0.05)
    m, d = design.m, 2 ** n
    d_e, d_c = DB.from_array(e), DB.from_array(c)
    d_er, d_cr = DB(R * B * m * 8), DB(R * B * m * 8)
    _lib.check(lib.fbx_beta_resample_dev(B * m, R, d_e.ptr, d_c.ptr, 1.0, 21, d_er.ptr, d_cr.ptr))
    d_rho, d_lin, d_phys, d_rop = (DB(R * B * d * d * 16) for _ in range(4))
    d_it, d_hit = DB(R * B * 4), DB(R * B * 4)
    _lib.check(lib.fbx_mle_state_dev(design.handle, R * B, d_er.ptr, d_cr.ptr, 0.1, 0.0, 0.0, 1e-7, 500,
                                     d_rho.ptr, d_it.ptr, d_hit.ptr))
    _lib.check(lib.fbx_linv_state_dev(design.handle, R * B, d_er.ptr, d_lin.ptr))
    _lib.check(lib.fbx_proj_state_physical_dev(n, R * B, d_lin.ptr, d_phys.ptr))
    _lib.check(lib.fbx_r_operator_dev(design.handle, R * B, d_rho.ptr, d_er.ptr, d_rop.ptr))
    d_ll = DB(R * B * 8)
    _lib.check(lib.fbx_state_log_likelihood_dev(design.handle, R * B, d_rho.ptr, d_er.ptr, d_cr.ptr, d_ll.ptr))
    d_tgt = DB.from_array(np.ascontiguousarray(np.tile(rhos, (R, 1, 1))))
    d_pur, d_fid, d_td, d_hs = (DB(R * B * 8) for _ in range(4))
    _lib.check(lib.fbx_state_measures_dev(n, R * B, d_tgt.ptr, d_phys.ptr, d_pur.ptr, d_fid.ptr, d_td.ptr, d_hs.ptr))
    d_w, d_v = DB(R * B * d * 8), DB(R * B * d * d * 16)
    _lib.check(lib.fbx_eigh_dev(d, R * B, d_phys.ptr, d_w.ptr, d_v.ptr))
    _lib.synchronize()

    e_rs = tomography.resample_expectations_with_beta_batch(e, c, R, seed=21).reshape(R * B, m)
    c_rs = np.tile(c, (R, 1))
    assert np.array_equal(d_er.to_array(np.float64, (R * B, m)), e_rs)
    assert np.array_equal(d_cr.to_array(np.float64, (R * B, m)), c_rs)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mle = tomography.iterative_mle_state_estimate_batch(design, e_rs, c_rs, epsilon=0.1, tol=1e-7, maxiter=500)
    assert np.array_equal(d_rho.to_array(np.complex128, (R * B, d, d)), mle)
    lin = tomography.linear_inv_state_estimate_batch(design, e_rs)
    assert np.array_equal(d_lin.to_array(np.complex128, (R * B, d, d)), lin)
    phys = project_state_matrix_to_physical_batch(lin)
    assert np.array_equal(d_phys.to_array(np.complex128, (R * B, d, d)), phys)
    assert np.array_equal(d_rop.to_array(np.complex128, (R * B, d, d)), tomography._R_batch(mle, design, e_rs))
    assert np.array_equal(d_ll.to_array(np.float64, (R * B,)),
                          tomography.state_log_likelihood_batch(mle, design, e_rs, c_rs))
    tgt = np.ascontiguousarray(np.tile(rhos, (R, 1, 1)))
    want = dm.state_measures_batch(tgt, phys, ("purity", "fidelity", "trace_distance", "hs_ip"))
    for key, buf in (("purity", d_pur), ("fidelity", d_fid), ("trace_distance", d_td), ("hs_ip", d_hs)):
        assert np.array_equal(buf.to_array(np.float64, (R * B,)), want[key]), key
    w, v = _lib.eigh_batch(phys)
    assert np.array_equal(d_w.to_array(np.float64, (R * B, d)), w)
    fid = d_fid.to_array(np.float64, (R, B))
    assert (fid.mean(axis=0) > 0.8).all() and (fid.var(axis=0) > 0).all()


def test_linv_process_and_apply_choi_on_device_pointers(gpu):
    from fbx import _lib, synthetic, tomography
    from fbx.operator_tools.apply_superoperator import apply_choi_matrix_2_state_batch
    lib, DB = _lib.lib(), _lib.DeviceBuffer
    for n in (1, 2):
        B, d = 4, 2 ** n
        D = d * d
        design, us, e, c = synthetic.process_batch(n, "sic", B)
        d_e, d_choi = DB.from_array(e), DB(B * D * D * 16)
        _lib.check(lib.fbx_linv_process_dev(design.handle, B, d_e.ptr, d_choi.ptr))
        rho = np.zeros((B, d, d), np.complex128); rho[:, 0, 0] = 1.0
        d_rho, d_out = DB.from_array(rho), DB(B * D * 16)
        _lib.check(lib.fbx_apply_choi_dev(n, B, d_choi.ptr, d_rho.ptr, d_out.ptr))
        _lib.synchronize()
        choi = tomography.linear_inv_process_estimate_batch(design, e)
        assert np.array_equal(d_choi.to_array(np.complex128, (B, D, D)), choi)
        assert np.array_equal(d_out.to_array(np.complex128, (B, d, d)), apply_choi_matrix_2_state_batch(choi, rho))


def test_process_tomography_walkthrough_example(gpu):
    """examples/process_tomography_walkthrough.py end to end: settings -> results -> linear inversion / PGDB -> projections ->
    fidelities, Watrous bounds -> bootstrap error bars -> plot inputs, all through the reference-named API."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "process_tomography_walkthrough.py")
    spec = importlib.util.spec_from_file_location("walkthrough", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = mod.main(verbose=False)
    assert 0.93 < out["fidelity_pgdb"] < 0.99                        # a CNOT with 3 % depolarising noise
    assert out["fidelity_closest_unitary"] > 0.995                  # its unitary part is the CNOT
    assert abs(out["fidelity_linear_inversion_projected"] - out["fidelity_pgdb"]) < 0.02
    mean, err = out["bootstrap_fidelity"]
    assert abs(mean - out["fidelity_pgdb"]) < 5 * err + 5e-3 and 1e-4 < err < 1e-2
    assert out["ptm"].shape == (16, 16) and out["labels"][:3] == ["II", "IX", "IY"]
    lo, hi = out["diamond_norm_bounds_to_ideal"]
    assert 0 < lo < hi


def test_state_tomography_from_shots_example(gpu):
    """examples/state_tomography_from_shots.py end to end: bitstrings -> moments -> linear inversion / MLE variants ->
    projection -> measures -> bootstrap variance."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "state_tomography_from_shots.py")
    spec = importlib.util.spec_from_file_location("state_example", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = mod.main(verbose=False)
    for name in ("... projected to physical", "MLE", "max-entropy MLE", "hedged MLE"):
        fid, tdist, pur = out[name]
        assert fid > 0.97 and tdist < 0.08 and 0.75 < pur < 1.0, (name, out[name])
    mean, err = out["bootstrap purity"]
    assert abs(mean - out["true purity"]) < 5 * err + 0.02 and 1e-4 < err < 0.05
