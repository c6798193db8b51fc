"""Merged / repeated datasets: the reference loops over WHATEVER result list it is given (tomography.py:494-539, :273-338), so the
same design measured several times -- with different shot counts per repetition -- is one experiment of reps x m settings.
tests/golden/repeated.npz (make_goldens.py --repeated, produced by the reference itself) holds such lists beyond the resident
sizes of the kernels: 2-qubit process designs of 1620 (Pauli x 3) and 1200 (SIC x 5) settings (register-resident kernels:
1024), a 1-qubit one of 270 (256), state designs of 4200 settings (64 KiB of per-setting LDS staging = 3970).  Until round 5
those calls returned FBX_ERR_UNSUPPORTED (3 qubits: beyond 14 336 settings, tests at the end); now they take the streamed forms (csrc/fbx_pgdb_body.hpp STREAM: outcome slots read
from HBM / L2; csrc/fbx_state.hip r_operator_elem: no per-setting staging) and must give the reference's answer for that list.

CPU: the oracle against the fixture.  GPU: the kernels against the fixture, and the streamed kernels against the resident ones
on a design both can run."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _g():
    return np.load(os.path.join(GOLD, "repeated.npz"))


def _process_design(g, tag, n, mod):
    p = g[f"{tag}_paulis"]
    return mod.Design(n, "process", g[f"{tag}_in_labels"], p, np.ones(len(p)))


def _state_design(g, tag, n, mod):
    p = g[f"{tag}_paulis"]
    return mod.Design(n, "state", np.full_like(p, 4), p, np.ones(len(p)))


# ------------------------------------------------------------------------------------------------ CPU
def test_oracle_reproduces_the_repeated_dataset_fixtures():
    from fbx_oracle import design as od, estimators as oe
    g = _g()
    for tag, n in (("p2pauli", 2), ("p1pauli", 1)):
        d = _process_design(g, tag, n, od)
        assert d.m == g[f"{tag}_e"].shape[1] and d.m > (1024 if n == 2 else 256)
        est = oe.pgdb_process_estimate(d, g[f"{tag}_e"][0], g[f"{tag}_c"][0])
        assert np.abs(est - g[f"{tag}_pgdb"][0]).max() < 1e-12
        assert np.abs(oe.linear_inv_process_estimate(d, g[f"{tag}_e"][0]) - g[f"{tag}_linv"][0]).max() < 1e-10
    d = _state_design(g, "s1", 1, od)
    assert d.m == 4200
    rho = oe.iterative_mle_state_estimate(d, g["s1_e"][0], g["s1_c"][0], maxiter=40)
    assert np.abs(rho - g["s1_mle40"][0]).max() < 1e-12
    assert abs(oe.state_log_likelihood(g["s1_mle40"][0], d, g["s1_e"][0], g["s1_c"][0]) - g["s1_loglik"][0]) < 1e-9 * abs(g["s1_loglik"][0])


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("tag,n", [("p2pauli", 2), ("p2sic", 2), ("p1pauli", 1)])
def test_process_tomography_of_a_repeated_dataset(gpu, tag, n):
    from fbx import design as fd, tomography
    from fbx_oracle import design as od, estimators as oe
    g = _g()
    d = _process_design(g, tag, n, fd)
    e, c = g[f"{tag}_e"], g[f"{tag}_c"]
    got, st = tomography.pgdb_process_estimate_batch(d, e, c, return_stats=True)
    assert np.abs(got - g[f"{tag}_pgdb"]).max() < 1e-9                                  # the reference's answer for that list
    # the oracle's counts for item 0 (every outer iteration / Dykstra iteration)
    _, os_ = oe.pgdb_process_estimate(_process_design(g, tag, n, od), e[0], c[0], return_stats=True)
    assert st["iterations"][0] == os_["iterations"] and st["dykstra"][0] == os_["dykstra"]
    tni = tomography.pgdb_process_estimate_batch(d, e[:1], c[:1], trace_preserving=False)
    assert np.abs(tni - g[f"{tag}_pgdb_tni"]).max() < 1e-9
    lin = tomography.linear_inv_process_estimate_batch(d, e)
    assert np.abs(lin - g[f"{tag}_linv"]).max() < 1e-10
    # fixed iteration count and the per-iteration trace work there too; a larger batch than one wave slot per item
    reps = 5
    big, sb = tomography.pgdb_process_estimate_batch(d, np.tile(e, (reps, 1)), np.tile(c, (reps, 1)), mode="fixed", max_iters=7,
                                                     return_stats=True, trace_iters=7)
    assert (sb["iterations"] == 7).all() and np.array_equal(big[:e.shape[0]], big[-e.shape[0]:])
    assert (sb["trace"][:, :, 0].sum(axis=1) == sb["dykstra"]).all()


@pytest.mark.gpu
def test_streamed_kernel_agrees_with_the_resident_one(gpu):
    """A design BOTH forms can run: the 540-setting Pauli design padded with 540 zero-count repetitions of itself is the same
    likelihood (a setting with total_counts = 0 contributes nothing, tomography.py:528-538) and takes the streamed kernel
    (1080 > 1024 settings); the estimates must agree with the register-resident kernels' to rounding, counts equal."""
    from fbx import design as fd, synthetic, tomography
    base, _, e, c = synthetic.process_batch(2, "pauli", 24)
    want, sw = tomography.pgdb_process_estimate_batch(base, e, c, return_stats=True)
    twice = fd.Design(2, "process", np.tile(base.in_labels, (2, 1)), np.tile(base.paulis, (2, 1)))
    e2 = np.concatenate([e, np.zeros_like(e)], axis=1)
    c2 = np.concatenate([c, np.zeros_like(c)], axis=1)
    got, sg = tomography.pgdb_process_estimate_batch(twice, e2, c2, return_stats=True)
    assert np.abs(got - want).max() < 1e-10
    for k in ("iterations", "dykstra"):
        assert np.array_equal(sg[k], sw[k]), k
    assert (sg["backtracks"] == sw["backtracks"]).mean() >= 0.9       # (another summation order: a last, stalled line search may differ)
    f_w, s_w = tomography.pgdb_process_estimate_batch(base, e, c, mode="fixed", max_iters=100, return_stats=True)
    f_g, s_g = tomography.pgdb_process_estimate_batch(twice, e2, c2, mode="fixed", max_iters=100, return_stats=True)
    assert np.array_equal(s_g["dykstra"], s_w["dykstra"]) and np.abs(f_g - f_w).max() < 1e-8


@pytest.mark.gpu
@pytest.mark.parametrize("tag,n", [("s2", 2), ("s1", 1)])
def test_state_tomography_of_a_repeated_dataset(gpu, tag, n):
    from fbx import design as fd, tomography
    g = _g()
    d = _state_design(g, tag, n, fd)
    e, c = g[f"{tag}_e"], g[f"{tag}_c"]
    assert d.m == 4200
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        rho = tomography.iterative_mle_state_estimate_batch(d, e, c, maxiter=40)
        hed = tomography.iterative_mle_state_estimate_batch(d, e, c, beta=0.5, epsilon=1e-4, maxiter=12)
    assert np.abs(rho - g[f"{tag}_mle40"]).max() < 1e-11
    assert np.abs(hed - g[f"{tag}_hedged12"]).max() < 1e-10
    assert np.abs(tomography.linear_inv_state_estimate_batch(d, e) - g[f"{tag}_linv"]).max() < 1e-11
    assert np.abs(tomography._R_batch(g[f"{tag}_mle40"], d, e) - g[f"{tag}_r_op"]).max() < 1e-10
    ll = tomography.state_log_likelihood_batch(g[f"{tag}_mle40"], d, e, c)
    assert np.abs(ll - g[f"{tag}_loglik"]).max() < 1e-9 * np.abs(g[f"{tag}_loglik"]).max()


@pytest.mark.gpu
def test_three_qubit_repeated_dataset(gpu):
    """16 128 settings (the 3-qubit SIC design measured four times): beyond the 14 336 of the resident 3-qubit instantiations; the
    kernel's 32-slot instantiation keeps its per-slot arrays in scratch.  tests/golden/repeated_3q.npz holds the reference's
    answer for that list (make_goldens.py --repeated3q: a 2.1 GB dense design matrix)."""
    from fbx import design as fd, tomography
    g = np.load(os.path.join(GOLD, "repeated_3q.npz"))
    p = g["paulis"]
    d = fd.Design(3, "process", g["in_labels"], p, np.ones(len(p)))
    assert d.m == 16128
    got, st = tomography.pgdb_process_estimate_batch(d, g["e"], g["c"], return_stats=True)
    assert np.abs(got - g["pgdb"]).max() < 1e-9
    assert st["iterations"][0] > 20


@pytest.mark.gpu
def test_three_qubit_large_instantiations_agree_with_the_resident_ones(gpu):
    """The SIC design padded with zero-count repetitions of itself is the same likelihood: 4 x 4032 = 16 128 settings take the 32-slot
    instantiation, 9 x 4032 = 36 288 the 64-slot one; both must reproduce the resident kernel's estimate with every count equal."""
    from fbx import design as fd, synthetic, tomography
    base, _, e, c = synthetic.process_batch(3, "sic", 3)
    want, sw = tomography.pgdb_process_estimate_batch(base, e, c, return_stats=True)
    for reps in (4, 9):
        big = fd.Design(3, "process", np.tile(base.in_labels, (reps, 1)), np.tile(base.paulis, (reps, 1)))
        e2 = np.concatenate([e] + [np.zeros_like(e)] * (reps - 1), axis=1)
        c2 = np.concatenate([c] + [np.zeros_like(c)] * (reps - 1), axis=1)
        got, sg = tomography.pgdb_process_estimate_batch(big, e2, c2, return_stats=True)
        assert np.abs(got - want).max() < 1e-10, reps
        for k in ("iterations", "dykstra"):
            assert np.array_equal(sg[k], sw[k]), (reps, k)
