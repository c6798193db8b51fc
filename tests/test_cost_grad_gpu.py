"""_cost / _grad_cost / _extract_from_results (tomography.py:494-539, :597-633) as directly callable functions:
fbx_pgdb_cost_grad evaluates ONE cost and ONE gradient with the reconstruction kernels' own device functions (Pauli transform,
prediction table, per-state weights, gradient coefficients, inverse transform), held here against the oracle's dense
`-n^T log(clip(A vec E))` and `-unvec(A^H (n / clip(A vec E)))` to 1e-12 -- on fixture estimates (the reference's own converged and
100-iteration estimates), on the starting point, on a non-physical Hermitian matrix and on an estimate that sits ON the 1e-6 clip
(the identity channel: an input eigenstate of the measured Pauli has one outcome of probability exactly zero)."""
import os

import numpy as np
import pytest

from fbx_oracle import design as od, estimators as oe

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _unitary_choi(u):
    d = u.shape[0]
    v = u.reshape(-1, 1, order="F")                                  # column-stacking vec (superoperator_transformations.py:33-51)
    return v @ v.conj().T


def _estimates(g, n_items, D, rng):
    ests, tags = [], []
    for b in range(n_items):
        ests += [g["pgdb_conv"][b], g["pgdb_fixed"][b]]; tags += [f"conv{b}", f"fixed{b}"]
    ests.append(np.eye(D) / int(np.sqrt(D))); tags.append("start")
    h = rng.standard_normal((D, D)) + 1j * rng.standard_normal((D, D))
    ests.append(g["pgdb_conv"][0] + 0.05 * (h + h.conj().T)); tags.append("non-physical")     # negative probabilities -> clipped
    return ests, tags


@pytest.mark.parametrize("n,basis,n_items", [(2, "pauli", 4), (2, "sic", 3)])
def test_cost_and_gradient_against_the_oracle_two_qubits(gpu, n, basis, n_items):
    from fbx import tomography, synthetic
    g = np.load(os.path.join(GOLD, f"process_{n}q_{basis}_fixed100.npz"))
    design = synthetic.process_design(n, basis)
    d = od.process_design(n, basis)
    A = oe.design_matrix_A(d)
    D = 4 ** n
    rng = np.random.default_rng(5)
    ests, tags = _estimates(g, n_items, D, rng)
    ests.append(_unitary_choi(np.eye(2 ** n))); tags.append("on-the-clip")
    nvs = []
    for k, tag in enumerate(tags):
        b = int(tag[-1]) if tag[-1].isdigit() else 0
        nvs.append(oe.counts_vector(g["expectations"][b], g["counts"][b])[:, 0])
    nv = np.array(nvs)
    assert np.array_equal(tomography.normalised_counts(g["expectations"][:1], g["counts"][:1])[0], nv[0])     # the reference's n, bit for bit
    cost, grad = tomography.cost_and_gradient_batch(design, nv, np.array(ests))
    on_clip = 0
    for k, tag in enumerate(tags):
        want_c = oe.cost(A, nv[k][:, None], ests[k])
        want_g = oe.grad_cost(A, nv[k][:, None], ests[k])
        assert abs(want_c.imag).max() < 1e-12
        scale = max(1.0, float(np.abs(want_g).max()))
        assert abs(cost[k] - want_c.real.item()) <= 1e-12 * max(1.0, abs(want_c.real.item())), (tag, cost[k], want_c)
        assert np.abs(grad[k] - want_g).max() <= 1e-12 * scale, (tag, np.abs(grad[k] - want_g).max(), scale)
        p = (A @ ests[k].reshape(-1, 1, order="F")).real
        on_clip += int((p < 1e-6).sum()) if tag == "on-the-clip" else 0
    assert on_clip >= 4                                               # the clip really is exercised (eta = n / 1e-6 there)
    # the reference-signature functions on one experiment
    Am = tomography.DesignMatrix(design)
    c1 = tomography._cost(Am, nv[0][:, None], ests[0])
    assert c1.shape == (1, 1) and c1[0, 0] == cost[0]
    assert np.array_equal(tomography._grad_cost(Am, nv[0][:, None], ests[0]), grad[0])
    c_eps = tomography._cost(Am, nv[-1][:, None], ests[-1], eps=1e-3)            # eps is an argument, as in the reference
    assert abs(c_eps[0, 0] - oe.cost(A, nv[-1][:, None], ests[-1], eps=1e-3).real.item()) < 1e-12


def test_cost_and_gradient_one_qubit_and_extract_from_results(gpu):
    from fbx import tomography, synthetic
    from fbx.observable_estimation import ExperimentResult
    g = np.load(os.path.join(GOLD, "process_1q_pauli.npz"))
    design, us, e, c = synthetic.process_batch(1, "pauli", 3)
    d = od.process_design(1, "pauli")
    A = oe.design_matrix_A(d)
    settings = tomography.generate_process_tomography_settings([0], "pauli")
    results = [ExperimentResult(setting=s, expectation=float(x), std_err=0.0, total_counts=int(k)) for s, x, k in zip(settings, e[0], c[0])]
    Am, n = tomography._extract_from_results(results, [0])
    assert Am.shape == A.shape and n.shape == (2 * design.m, 1)
    assert np.array_equal(n, oe.counts_vector(e[0], c[0]))
    est = tomography.pgdb_process_estimate(results, [0])
    for E in (est, np.eye(4) / 2, _unitary_choi(us[0]), _unitary_choi(np.eye(2))):
        assert abs(tomography._cost(Am, n, E)[0, 0] - oe.cost(A, n, E).real.item()) < 1e-12
        want = oe.grad_cost(A, n, E)
        assert np.abs(tomography._grad_cost(Am, n, E) - want).max() <= 1e-12 * max(1.0, np.abs(want).max())


def test_cost_and_gradient_three_qubits(gpu):
    from fbx import tomography, synthetic
    g = np.load(os.path.join(GOLD, "process_3q_sic_fixed100.npz"))
    design = synthetic.process_design(3, "sic")
    d = od.process_design(3, "sic")
    A = oe.design_matrix_A(d, sparse=True)
    rng = np.random.default_rng(7)
    ests, tags = _estimates(g, 1, 64, rng)
    ests.append(_unitary_choi(np.eye(8))); tags.append("on-the-clip")
    nv = np.tile(oe.counts_vector(g["expectations"][0], g["counts"][0])[:, 0], (len(ests), 1))
    cost, grad = tomography.cost_and_gradient_batch(design, nv, np.array(ests))
    for k, tag in enumerate(tags):
        v = ests[k].reshape(-1, 1, order="F")
        p = np.clip(np.asarray(A @ v), a_min=1e-6, a_max=None)
        want_c = (-nv[k][None, :] @ np.log(p)).item()
        want_g = np.asarray(-(A.conj().T @ (nv[k][:, None] / p))).reshape(64, 64, order="F")
        assert abs(cost[k] - want_c.real) <= 1e-12 * max(1.0, abs(want_c.real)), (tag, cost[k], want_c)
        assert np.abs(grad[k] - want_g).max() <= 1e-12 * max(1.0, np.abs(want_g).max()), (tag, np.abs(grad[k] - want_g).max())
    cost2, none = tomography.cost_and_gradient_batch(design, nv, np.array(ests), gradient=False)
    assert none is None and np.array_equal(cost2, cost)


def test_cost_grad_argument_errors(gpu):
    from fbx import tomography, synthetic
    design = synthetic.process_design(1, "sic")
    with pytest.raises(ValueError):
        tomography.cost_and_gradient_batch(design, np.zeros((1, 5)), np.eye(4)[None] / 2)
    sdesign = synthetic.state_design(1)
    with pytest.raises(ValueError):
        tomography.cost_and_gradient_batch(sdesign, np.zeros((1, 2 * sdesign.m)), np.eye(4)[None] / 2)
    with pytest.raises(ValueError):
        tomography.cost_and_gradient_batch(design, np.zeros((1, 2 * design.m)), np.eye(4)[None] / 2, eps=-1.0)
