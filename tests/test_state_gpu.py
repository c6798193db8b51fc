"""HIP state-tomography estimators and state measures vs the reference goldens / the oracle."""
import os
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def gold(n):
    return np.load(os.path.join(GOLD, f"state_{n}q.npz"))


def _design(n):
    from fbx.design import state_design
    return state_design(n)


@pytest.mark.parametrize("n", [1, 2])
def test_linear_inversion(gpu, n):
    from fbx import tomography as T
    g = gold(n)
    got = T.linear_inv_state_estimate_batch(_design(n), g["expectations"])
    assert np.abs(got - g["linv"]).max() < 1e-13


@pytest.mark.parametrize("n", [1, 2])
def test_iterative_mle_variants(gpu, n):
    from fbx import tomography as T
    g = gold(n)
    d = _design(n)
    e, c = g["expectations"], g["counts"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got, st = T.iterative_mle_state_estimate_batch(d, e, c, maxiter=100, return_stats=True)
    assert np.abs(got - g["mle100"]).max() < 1e-11
    assert st["hit_max"].all()                      # 99 updates do not reach tol=1e-9
    got = T.iterative_mle_state_estimate_batch(d, e, c, epsilon=0.5, tol=1e-6, maxiter=2000)
    assert np.abs(got - g["mle_tol"]).max() < 1e-9
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = T.iterative_mle_state_estimate_batch(d, e, c, beta=0.5, epsilon=1e-4, maxiter=60)
        assert np.abs(got - g["hedged"]).max() < 1e-9
        got = T.iterative_mle_state_estimate_batch(d, e, c, entropy_penalty=0.005, maxiter=60)
        assert np.abs(got - g["maxent"]).max() < 1e-10


def test_mle_iteration_count_and_warning(gpu):
    """maxiter=N performs N-1 updates and warns (tomography.py:241-246)."""
    from fbx import tomography as T
    from fbx_oracle import design as od, estimators as oe
    g = gold(1)
    d = _design(1)
    with pytest.warns(UserWarning, match="Maximum number of iterations"):
        got, st = T.iterative_mle_state_estimate_batch(d, g["expectations"][:2], g["counts"][:2], maxiter=5,
                                                       return_stats=True)
    assert (st["iterations"] == 5).all()
    o = od.state_design(1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = oe.iterative_mle_state_estimate(o, g["expectations"][0], g["counts"][0], maxiter=5)
    assert np.abs(got[0] - want).max() < 1e-13
    # converged run: same iteration count as the oracle
    got, st = T.iterative_mle_state_estimate_batch(d, g["expectations"][:3], g["counts"][:3], epsilon=0.5,
                                                   tol=1e-6, maxiter=2000, return_stats=True)
    for b in range(3):
        _, wst = oe.iterative_mle_state_estimate(o, g["expectations"][b], g["counts"][b], epsilon=0.5,
                                                 tol=1e-6, maxiter=2000, return_stats=True)
        assert st["iterations"][b] == wst["iterations"]
    with pytest.raises(ValueError):
        T.iterative_mle_state_estimate_batch(d, g["expectations"], g["counts"], entropy_penalty=0.1, beta=0.1)


@pytest.mark.parametrize("n", [1, 2])
def test_r_operator_and_log_likelihood(gpu, n):
    from fbx import tomography as T
    g = gold(n)
    d = _design(n)
    r = T._R_batch(g["mle100"], d, g["expectations"])
    assert np.abs(r - g["r_op"]).max() < 1e-12
    ll = T.state_log_likelihood_batch(g["mle100"], d, g["expectations"], g["counts"])
    assert np.abs(ll - g["loglik"]).max() < 1e-9 * np.abs(g["loglik"]).max()


def test_r_operator_hand_calculation(gpu):
    """The worked 3:7 example of the reference's test-suite (tests/test_state_tomography.py:78-96)."""
    from fbx import tomography as T
    from fbx.observable_estimation import ExperimentResult, ExperimentSetting, PauliTerm, zeros_state
    p0 = np.array([[1, 0], [0, 0]]); p1 = np.array([[0, 0], [0, 1]])
    pp = np.array([[1, 1], [1, 1]]) / 2; pm = np.array([[1, -1], [-1, 1]]) / 2
    rho = np.eye(2) / 2
    exp = (3 - 7) / 10
    for op, a, b in (("Z", p0, p1), ("X", pp, pm)):
        res = [ExperimentResult(ExperimentSetting(zeros_state([0]), PauliTerm({0: op})), exp, 10)]
        want = ((3 / 0.5) * a + (7 / 0.5) * b) / 10
        np.testing.assert_allclose(T._R(rho, res, [0]), want, atol=1e-12)
    # fixed point (Eq. 5 of Rehacek et al.): R rho R = rho for exact data
    ident = ExperimentResult(ExperimentSetting(zeros_state([0]), PauliTerm({})), 1, 1)
    zres = [ident, ExperimentResult(ExperimentSetting(zeros_state([0]), PauliTerm({0: "Z"})), 1, 1)]
    r = T._R(p0.astype(complex), zres, [0])
    np.testing.assert_allclose(r @ p0 @ r, p0, atol=1e-12)


@pytest.mark.parametrize("n", [1, 2])
def test_state_projection_and_measures(gpu, n):
    from fbx import distance_measures as dm
    from fbx.operator_tools.project_state_matrix import project_state_matrix_to_physical_batch
    g = np.load(os.path.join(GOLD, f"superops_{n}q.npz"))
    got = project_state_matrix_to_physical_batch(g["state_unphys"])
    assert np.abs(got - g["state_proj"]).max() < 1e-12
    m = dm.state_measures_batch(g["rho"], g["sigma"])
    assert np.abs(m["purity"] - g["purity"]).max() < 1e-13
    assert np.abs(m["fidelity"] - g["fidelity"]).max() < 1e-11
    assert np.abs(m["trace_distance"] - g["trace_distance"]).max() < 1e-13
    assert np.abs(m["hs_ip"] - g["hs_ip"]).max() < 1e-13


def test_smolin_example(gpu):
    """Fig. 1 of Smolin-Gambetta-Smith as used by the reference (tests/test_project_state_matrix.py:12-14),
    embedded in an 8-dimensional state (d must be a power of two on the device path)."""
    from fbx.operator_tools.project_state_matrix import project_state_matrix_to_physical
    from fbx_oracle import superops as so
    eigs = np.diag(np.array([-11.0 / 20, 1.0 / 10, 7.0 / 20, 1.0 / 2, 3.0 / 5, 0.0, 0.0, 0.0])).astype(complex)
    got = project_state_matrix_to_physical(eigs)
    np.testing.assert_allclose(got, so.project_state_matrix_to_physical(eigs), atol=1e-13)
    np.testing.assert_allclose(np.diag(got)[:5].real, [0, 0, 1.0 / 5, 7.0 / 20, 9.0 / 20], atol=1e-13)


def test_three_qubit_state_path(gpu):
    from fbx import synthetic, tomography as T
    from fbx_oracle import design as od, estimators as oe
    d, rhos, e, c = synthetic.state_batch(3, 2, mixed=0.2)
    o = od.state_design(3)
    got = T.linear_inv_state_estimate_batch(d, e)
    for b in range(2):
        assert np.abs(got[b] - oe.linear_inv_state_estimate(o, e[b])).max() < 1e-12
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = T.iterative_mle_state_estimate_batch(d, e, c, maxiter=40)
        for b in range(2):
            assert np.abs(got[b] - oe.iterative_mle_state_estimate(o, e[b], c[b], maxiter=40)).max() < 1e-11


def test_reference_signatures_on_result_lists(gpu):
    from fbx import tomography as T
    from fbx.observable_estimation import ExperimentResult
    g = gold(2)
    settings = T.generate_state_tomography_settings([0, 1])
    res = [ExperimentResult(s, float(g["expectations"][0][k]), int(g["counts"][0][k])) for k, s in enumerate(settings)]
    np.testing.assert_allclose(T.linear_inv_state_estimate(res, [0, 1]), g["linv"][0], atol=1e-13)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        np.testing.assert_allclose(T.iterative_mle_state_estimate(res, [0, 1], maxiter=100), g["mle100"][0], atol=1e-11)
    assert abs(T.state_log_likelihood(g["mle100"][0], res, [0, 1]) - g["loglik"][0]) < 1e-9 * abs(g["loglik"][0])


def test_bootstrap_batched_equals_one_at_a_time(gpu):
    """estimate_variance (tomography.py:412-453): the batched device path and the reference-style
    loop consume the np.random stream identically, so they agree to rounding for the same seed."""
    import functools
    from fbx import distance_measures as dm, tomography as T
    from fbx.observable_estimation import ExperimentResult
    g = gold(2)
    qubits = [0, 1]
    settings = T.generate_state_tomography_settings(qubits)
    res = [ExperimentResult(s, float(g["expectations"][0][k]), int(g["counts"][0][k])) for k, s in enumerate(settings)]
    target = g["truth"][0]
    est = functools.partial(T.iterative_mle_state_estimate, epsilon=0.5, tol=1e-6, maxiter=300)
    np.random.seed(11)
    fast = T.estimate_variance(res, qubits, est, dm.fidelity, target_state=target, n_resamples=12,
                               project_to_physical=True)
    np.random.seed(11)
    slow = T.estimate_variance(res, qubits, lambda r, q: est(r, q), dm.fidelity, target_state=target,
                               n_resamples=12, project_to_physical=True)       # opaque callable -> loop
    assert abs(fast[0] - slow[0]) < 1e-9 and abs(fast[1] - slow[1]) < 1e-9
    np.random.seed(3)
    m, v = T.estimate_variance(res, qubits, T.linear_inv_state_estimate, dm.purity, n_resamples=40)
    assert 0.2 < m < 1.2 and v >= 0
    with pytest.raises(ValueError):
        T.estimate_variance(res, qubits, T.linear_inv_state_estimate, dm.fidelity)


def test_eigh_entry_point_and_validators(gpu):
    from fbx import _lib
    from fbx import operator_tools as ot
    rs = np.random.RandomState(2)
    for N in (2, 4, 8, 16):
        a = rs.randn(5, N, N) + 1j * rs.randn(5, N, N)
        w, v = _lib.eigh_batch(a)
        for b in range(5):
            ww = np.linalg.eigvalsh(a[b])                      # numpy reads the lower triangle too
            assert np.abs(w[b] - ww).max() < 1e-12
            low = np.tril(a[b], -1); h = low + low.conj().T + np.diag(np.diag(a[b]).real)
            assert np.abs(h @ v[b] - v[b] * w[b]).max() < 1e-11
            assert np.abs(v[b].conj().T @ v[b] - np.eye(N)).max() < 1e-12
    x = np.array([[0, 1], [1, 0]], dtype=complex)
    cx = ot.kraus2choi(x)
    assert ot.choi_is_cptp(cx) and ot.choi_is_unitary(cx) and ot.choi_is_unital(cx)
    assert ot.choi_is_trace_preserving(cx) and ot.choi_is_completely_positive(cx)
    assert not ot.choi_is_trace_preserving(ot.kraus2choi(x - 0.1 * np.eye(2)))
    assert not ot.choi_is_completely_positive(-cx)
    ks = [np.array([[1, 0], [0, np.sqrt(0.9)]]), np.array([[0, np.sqrt(0.1)], [0, 0]])]
    assert ot.kraus_operators_are_valid(ks) and not ot.kraus_operators_are_valid([ks[0]])
    assert not ot.choi_is_unitary(ot.kraus2choi(ks)) and not ot.choi_is_unital(ot.kraus2choi(ks))
    ops = ot.choi2kraus(ot.kraus2choi(ks))
    assert len(ops) == 2 and np.abs(ot.kraus2choi(ops) - ot.kraus2choi(ks)).max() < 1e-12
    with pytest.raises(ValueError):
        ot.is_positive_semidefinite_matrix(np.array([[1, 2], [3, 4.0]]))
    with pytest.raises(ValueError):
        ot.is_hermitian_matrix(np.zeros((2, 3)))


def test_proj_choi_to_unitary(gpu):
    """tests/test_project_superoperators.py:86-113 of the reference, as data."""
    from fbx import operator_tools as ot
    import known_answers as ka
    for u in (ka.CNOT, ka.Z, ka.H):
        c = ot.kraus2choi(u)
        np.testing.assert_allclose(ot.proj_choi_to_unitary(c), c, atol=1e-10)

    def bit_flip(p):
        return [np.sqrt(1 - p) * ka.I2, np.sqrt(p) * ka.X]
    np.testing.assert_allclose(ot.proj_choi_to_unitary(ot.kraus2choi(bit_flip(0.1))), ot.kraus2choi(ka.I2), atol=1e-10)
    np.testing.assert_allclose(ot.proj_choi_to_unitary(ot.kraus2choi(bit_flip(0.9))), ot.kraus2choi(ka.X), atol=1e-10)
