"""Edge cases of the estimators through the C ABI, each against the oracle on the same inputs:
ragged designs (shuffled, duplicated and missing settings, non-unit coefficients), degenerate data
(noise-free expectations of +-1 and 0 that put model probabilities on the 1e-6 clip, settings with
zero counts), the smallest designs, and batch sizes around the launch geometry."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _od(design):
    from fbx_oracle import design as od
    return od.Design(design.n_qubits, design.kind, design.in_labels, design.paulis, design.coefs)


def _oracle_pgdb(design, e, c, **kw):
    from fbx_oracle import estimators as oe
    d = _od(design)
    A = oe.design_matrix_A(d)
    res = [oe.pgdb_process_estimate(d, e[b], c[b], A=A, return_stats=True, **kw) for b in range(e.shape[0])]
    return np.array([r[0] for r in res]), [r[1] for r in res]


def _check(design, e, c, tol=1e-9, **kw):
    from fbx import tomography
    got, st = tomography.pgdb_process_estimate_batch(design, e, c, return_stats=True, **kw)
    want, ws = _oracle_pgdb(design, e, c, **kw)
    assert np.abs(got - want).max() < tol
    for b in range(e.shape[0]):
        assert st["iterations"][b] == ws[b]["iterations"] and st["dykstra"][b] == ws[b]["dykstra"]
    return got


@pytest.mark.parametrize("n", [1, 2])
def test_ragged_designs(gpu, n):
    from fbx import synthetic
    from fbx.design import Design
    full, us, e, c = synthetic.process_batch(n, "pauli", 2)
    rng = np.random.default_rng(4)
    # shuffled order
    perm = rng.permutation(full.m)
    d1 = Design(n, "process", full.in_labels[perm], full.paulis[perm])
    _check(d1, e[:, perm], c[:, perm], mode="fixed", max_iters=6)
    # one fifth of the settings missing, some of the rest measured twice with different data
    keep = np.sort(rng.choice(full.m, size=(4 * full.m) // 5, replace=False))
    dup = keep[: full.m // 10]
    idx = np.concatenate([keep, dup])
    e2 = np.concatenate([e[:, keep], np.clip(e[:, dup] + 0.05, -1, 1)], axis=1)
    c2 = np.concatenate([c[:, keep], 0.5 * c[:, dup]], axis=1)
    d2 = Design(n, "process", full.in_labels[idx], full.paulis[idx])
    _check(d2, e2, c2, mode="fixed", max_iters=6)
    # non-unit observable coefficients: expectation of (coef * P) is coef * <P>
    coefs = rng.choice([0.5, -1.0, 2.0, 1.0], size=full.m)
    d3 = Design(n, "process", full.in_labels, full.paulis, coefs)
    _check(d3, e * coefs / np.abs(coefs).max(), c, mode="fixed", max_iters=6)


def test_linear_inversion_with_missing_settings(gpu):
    from fbx import synthetic, tomography
    from fbx.design import Design
    from fbx_oracle import estimators as oe
    full, us, e, c = synthetic.process_batch(2, "sic", 2)
    keep = np.sort(np.random.default_rng(9).choice(full.m, size=full.m - 17, replace=False))
    d = Design(2, "process", full.in_labels[keep], full.paulis[keep])
    got = tomography.linear_inv_process_estimate_batch(d, e[:, keep])
    for b in range(2):
        want = oe.linear_inv_process_estimate(_od(d), e[b, keep])
        assert np.abs(got[b] - want).max() < 1e-9


@pytest.mark.parametrize("basis", ["pauli", "sic"])
def test_noise_free_clifford_data_sits_on_the_clip(gpu, basis):
    """Exact expectations of CNOT: +-1 and 0 only, so half of the model probabilities of the true
    channel are exactly zero and the 1e-6 clip of tomography.py:613 is active throughout."""
    from fbx import synthetic
    from fbx.design import process_design
    design = process_design(2, basis)
    cnot = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 0, 1], [0, 0, 1, 0]], dtype=complex)
    cz = np.diag([1, 1, 1, -1]).astype(complex)
    us = np.array([cnot, cz])
    e = synthetic.exact_process_expectations(design, us, 0.0)
    e = np.where(np.abs(e) < 1e-12, 0.0, np.clip(e, -1, 1))
    if basis == "pauli":
        assert (np.abs(e) == 1).any() and (e == 0).any()
    c = np.full(e.shape, 1000.0)
    got = _check(design, e, c, tol=1e-8)
    from fbx_oracle import superops as so, measures as om
    for b in range(2):
        f = om.process_fidelity(so.kraus2pauli_liouville(us[b]), so.choi2pauli_liouville(got[b]))
        assert f > 0.99


def test_settings_with_zero_counts(gpu):
    from fbx import synthetic
    design, us, e, c = synthetic.process_batch(2, "sic", 2)
    c = c.copy(); e = e.copy()
    dead = np.random.default_rng(2).choice(design.m, size=40, replace=False)
    c[:, dead] = 0.0
    e[:, dead] = 0.0
    _check(design, e, c, mode="fixed", max_iters=8)


def test_smallest_designs(gpu):
    """A single setting and a single input state: hopeless tomography, still the same numbers."""
    from fbx.design import Design
    d1 = Design(1, "process", np.array([[4]], np.uint8), np.array([[3]], np.uint8))          # Z+ -> Z
    _check(d1, np.array([[0.8], [-1.0]]), np.array([[100.0], [7.0]]), mode="fixed", max_iters=4)
    d2 = Design(2, "process", np.array([[0, 2]] * 3, np.uint8), np.array([[1, 0], [0, 2], [3, 3]], np.uint8))
    _check(d2, np.array([[0.1, -0.3, 0.9]]), np.array([[50.0, 60.0, 70.0]]), mode="fixed", max_iters=4)


@pytest.mark.parametrize("B", [1, 63, 64, 65, 257, 1025])
def test_batch_sizes_around_the_launch_geometry(gpu, B):
    """Items are independent: any batch size gives the same per-item answers as a batch of one."""
    from fbx import synthetic, tomography
    design, us, e, c = synthetic.process_batch(1, "sic", 5)
    reps = -(-B // 5)
    eb, cb = np.tile(e, (reps, 1))[:B], np.tile(c, (reps, 1))[:B]
    got, st = tomography.pgdb_process_estimate_batch(design, eb, cb, return_stats=True)
    ref, rst = tomography.pgdb_process_estimate_batch(design, e, c, return_stats=True)
    for b in range(B):
        assert np.array_equal(got[b], ref[b % 5]) and st["iterations"][b] == rst["iterations"][b % 5]


def test_state_estimators_on_ragged_designs(gpu):
    from fbx import synthetic, tomography
    from fbx.design import Design
    from fbx_oracle import estimators as oe
    import warnings
    full, rhos, e, c = synthetic.state_batch(2, 3, mixed=0.1)
    keep = np.array([0, 2, 3, 5, 5, 7, 8, 11, 12, 14, 1])              # missing, repeated, out of order
    d = Design(2, "state", None, full.paulis[keep])
    od_ = _od(d)
    lin = tomography.linear_inv_state_estimate_batch(d, e[:, keep])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mle = tomography.iterative_mle_state_estimate_batch(d, e[:, keep], c[:, keep], maxiter=200)
        for b in range(3):
            assert np.abs(lin[b] - oe.linear_inv_state_estimate(od_, e[b, keep])).max() < 1e-10
            want = oe.iterative_mle_state_estimate(od_, e[b, keep], c[b, keep], maxiter=200)
            assert np.abs(mle[b] - want).max() < 1e-9


def test_batches_beyond_one_launch_chunk(gpu):
    """More than 8192 items go through several launches that share the basis-store workspace (the wavefront-per-item
    kernel: single-qubit batches of this size would otherwise take the lane-per-item kernel, tests/test_pgdb1_gpu.py)."""
    from fbx import synthetic, tomography, _lib
    design, us, e, c = synthetic.process_batch(1, "sic", 7)
    B = 8192 + 300
    reps = -(-B // 7)
    eb, cb = np.tile(e, (reps, 1))[:B], np.tile(c, (reps, 1))[:B]
    with _lib.option("pgdb_packed_1q", 0.0):
        got, st = tomography.pgdb_process_estimate_batch(design, eb, cb, return_stats=True)
        ref, rst = tomography.pgdb_process_estimate_batch(design, e, c, return_stats=True)
    idx = np.arange(B) % 7
    assert np.array_equal(got, ref[idx])
    assert np.array_equal(st["iterations"], rst["iterations"][idx]) and np.array_equal(st["dykstra"], rst["dykstra"][idx])
    assert np.array_equal(st["cost"], rst["cost"][idx])


def test_design_with_more_than_576_settings(gpu):
    """2-qubit designs between 577 and 1024 settings use the 16-settings-per-lane instantiation."""
    from fbx import synthetic
    from fbx.design import Design
    full, us, e, c = synthetic.process_batch(2, "pauli", 2)
    rng = np.random.default_rng(12)
    extra = rng.choice(full.m, size=200, replace=False)
    idx = np.concatenate([np.arange(full.m), extra])
    d = Design(2, "process", full.in_labels[idx], full.paulis[idx])
    e2 = np.concatenate([e, np.clip(e[:, extra] + rng.normal(0, 0.03, size=(2, 200)), -1, 1)], axis=1)
    c2 = np.concatenate([c, c[:, extra]], axis=1)
    assert d.m == 740
    _check(d, e2, c2, mode="fixed", max_iters=8)
    _check(d, e2, c2)


@pytest.mark.timeout(120)
@pytest.mark.parametrize("mode", ["converge", "fixed"])
def test_non_finite_input_terminates_and_stays_local(gpu, mode):
    """A NaN / Inf expectation can never satisfy the reference's stopping rules (its loops would spin);
    here the poisoned item ends with a non-finite result and its neighbours are untouched."""
    from fbx import synthetic, tomography
    for n in (1, 2):
        design, us, e, c = synthetic.process_batch(n, "pauli", 5)
        kw = dict(mode=mode, max_iters=20 if mode == "fixed" else 0)
        clean = tomography.pgdb_process_estimate_batch(design, e, c, **kw)
        bad = e.copy()
        bad[1, 3] = np.nan
        bad[3, 0] = np.inf
        got = tomography.pgdb_process_estimate_batch(design, bad, c, **kw)
        assert not np.isfinite(got[1]).all() and not np.isfinite(got[3]).all()
        for b in (0, 2, 4):
            assert np.array_equal(got[b], clean[b])
    # state estimator: capped by maxiter, same locality
    design, rhos, e, c = synthetic.state_batch(2, 4, shots=200)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        clean = tomography.iterative_mle_state_estimate_batch(design, e, c, maxiter=200)
        bad = e.copy(); bad[2, 1] = np.nan
        got = tomography.iterative_mle_state_estimate_batch(design, bad, c, maxiter=200)
    assert not np.isfinite(got[2]).all()
    for b in (0, 1, 3):
        assert np.array_equal(got[b], clean[b])
