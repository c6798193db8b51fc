"""3-qubit (64 x 64 Choi) PGDB process tomography -- BASELINE config 4 -- vs the reference golden
(one converged SIC-basis reconstruction produced by forest.benchmarking itself) and the oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_converged_estimate_matches_reference_golden(gpu):
    from fbx import tomography
    from fbx.design import process_design
    g = np.load(os.path.join(GOLD, "process_3q_sic.npz"))
    design = process_design(3, "sic")
    assert design.m == 4032 == g["expectations"].shape[1]
    got, st = tomography.pgdb_process_estimate_batch(design, g["expectations"], g["counts"], return_stats=True)
    assert g["pgdb"].shape[0] >= 4                      # golden subsets of several items
    assert np.abs(got - g["pgdb"]).max() < 1e-9
    from fbx_oracle import measures as om, superops as so
    for b in range(got.shape[0]):
        # physical sanity of the estimate itself: Hermitian, trace preserving
        assert np.abs(got[b] - got[b].conj().T).max() < 1e-12
        pt = np.einsum("iojo->ij", got[b].reshape(8, 8, 8, 8))
        assert np.abs(pt - np.eye(8)).max() < 1e-12
        ideal = so.kraus2pauli_liouville(g["unitaries"][b])
        f = om.process_fidelity(ideal, so.choi2pauli_liouville(got[b]))
        f_ref = om.process_fidelity(ideal, so.choi2pauli_liouville(g["pgdb"][b]))
        assert abs(f - f_ref) < 1e-8 and f > 0.8


def test_few_iterations_match_oracle_incl_counts(gpu):
    """Fixed 3 outer iterations against the dense-A oracle (8064 x 4096 design matrix), two items,
    trace-preserving and trace-non-increasing.  With the eigensolver held at 1e-13 throughout
    (fbx_set_option('pgdb3_eig_rel_tol', 0)) the trajectory is the oracle's to 1e-11; with the default (inexact
    projections while the iteration is far from its fixed point, 1e-7 x the outer step: include/fbx.h) the counts
    are still equal and the estimate after three O(0.1 .. 1) steps sits within 1e-7 of it -- the price of an
    early iterate, not of the result: run to convergence both agree with the oracle to 1e-10 (the golden test
    above, scripts/parity_survey.py)."""
    from fbx import synthetic, tomography, _lib
    from fbx_oracle import design as od, estimators as oe
    design, us, e, c = synthetic.process_batch(3, "sic", 2, first_item=5)
    d = od.Design(3, "process", design.in_labels, design.paulis, design.coefs)
    A = oe.design_matrix_A(d)
    assert _lib.get_option("pgdb3_eig_rel_tol") == 1e-7
    for tp in (True, False):
        want = [oe.pgdb_process_estimate(d, e[b], c[b], trace_preserving=tp, A=A, mode="fixed", max_iters=3,
                                         return_stats=True) for b in range(2)]
        for rel_tol, tol in ((0.0, 1e-11), (None, 1e-7)):
            with _lib.option("pgdb3_eig_rel_tol", 1e-7 if rel_tol is None else rel_tol):
                got, st = tomography.pgdb_process_estimate_batch(design, e, c, trace_preserving=tp, mode="fixed",
                                                                 max_iters=3, return_stats=True)
            for b in range(2):
                assert np.abs(got[b] - want[b][0]).max() < tol
                assert st["dykstra"][b] == want[b][1]["dykstra"] and st["backtracks"][b] == want[b][1]["backtracks"]
                assert abs(st["cost"][b] - want[b][1]["cost"]) < tol
    assert _lib.get_option("pgdb3_eig_rel_tol") == 1e-7


def test_batch_of_256_properties(gpu):
    """Config 4 as benchmarked (batch 256, 100 fixed iterations): every item Hermitian + trace preserving,
    duplicates bit-identical, work counters consistent."""
    from fbx import synthetic, tomography
    design, us, e, c = synthetic.process_batch(3, "sic", 32)
    e = np.tile(e, (8, 1)); c = np.tile(c, (8, 1))
    got, st = tomography.pgdb_process_estimate_batch(design, e, c, mode="fixed", max_iters=100, return_stats=True)
    assert (st["iterations"] == 100).all()
    assert (st["jacobi_sweeps"] >= st["dykstra"]).all() and (st["eig_terms"] <= 64 * st["dykstra"]).all()
    # one evaluation at the start, one per iteration, one per halving -- plus what the four-at-a-time pass of a long halving run
    # evaluates beyond the accepted step (at most three per outer iteration: only the run's last pass can be cut short)
    extra = st["cost_evals"] - (1 + 100 + st["backtracks"])
    assert (extra >= 0).all() and (extra <= 3 * 100).all() and extra.max() > 0
    assert np.abs(got - got.conj().transpose(0, 2, 1)).max() < 1e-12
    pt = np.einsum("biojo->bij", got.reshape(-1, 8, 8, 8, 8))
    assert np.abs(pt - np.eye(8)).max() < 1e-12
    assert np.array_equal(got[:32], got[224:])


def test_choi_projections_3q_match_oracle(gpu):
    """project_superoperators.py:19-144 on 64 x 64 Choi matrices (1024-thread kernel)."""
    from fbx.operator_tools import project_superoperators as ps
    from fbx import _lib
    from fbx_oracle import superops as so
    rng = np.random.default_rng(33)
    xs = []
    for k in range(3):
        a = rng.normal(size=(64, 64)) + 1j * rng.normal(size=(64, 64))
        xs.append((a + a.conj().T) / 16 + np.eye(64) / 8 * (k != 2))
    x = np.stack(xs)
    got = ps.proj_choi_batch(_lib.PROJ_CP, x)
    for b in range(3):
        assert np.abs(got[b] - so.proj_choi_to_completely_positive(x[b])).max() < 1e-11
    got = ps.proj_choi_batch(_lib.PROJ_TP, x)
    for b in range(3):
        assert np.abs(got[b] - so.proj_choi_to_trace_preserving(x[b])).max() < 1e-12
    got = ps.proj_choi_batch(_lib.PROJ_TNI, x)
    for b in range(3):
        assert np.abs(got[b] - so.proj_choi_to_trace_non_increasing(x[b])).max() < 1e-11
    for kind, tp in ((_lib.PROJ_PHYSICAL_TP, True), (_lib.PROJ_PHYSICAL_TNI, False)):
        got, iters = ps.proj_choi_batch(kind, x, return_iters=True)
        for b in range(3):
            want, n_it = so.proj_choi_to_physical(x[b], tp, return_iters=True)
            assert iters[b] == n_it
            assert np.abs(got[b] - want).max() < 1e-10


def test_linear_inversion_3q_matches_reference_golden_and_oracle(gpu):
    """tomography.py:459-491 for three qubits: the golden holds the reference's own pinv solution."""
    from fbx import synthetic, tomography
    from fbx.design import process_design
    from fbx_oracle import design as od, estimators as oe
    g = np.load(os.path.join(GOLD, "process_3q_sic.npz"))
    design = process_design(3, "sic")
    got = tomography.linear_inv_process_estimate_batch(design, g["expectations"])
    if "linv" in g.files:
        assert np.abs(got - g["linv"]).max() < 1e-9
    d = od.Design(3, "process", design.in_labels, design.paulis, design.coefs)
    want = oe.linear_inv_process_estimate(d, g["expectations"][0])
    assert np.abs(got[0] - want).max() < 1e-9


def test_pauli_in_basis_3q_matches_oracle(gpu):
    """Config 4 stretch: 3 qubits with the Pauli in-basis (13 608 settings, 216 input states, the
    14-settings-per-thread instantiation) against the oracle with a sparse design matrix."""
    from fbx import synthetic, tomography
    from fbx_oracle import design as od, estimators as oe
    design, us, e, c = synthetic.process_batch(3, "pauli", 1, first_item=2)
    assert design.m == 13608
    d = od.Design(3, "process", design.in_labels, design.paulis, design.coefs)
    A = oe.design_matrix_A(d, sparse=True)
    from fbx import _lib
    with _lib.option("pgdb3_eig_rel_tol", 0.0):       # the oracle's trajectory iteration by iteration
        got, st = tomography.pgdb_process_estimate_batch(design, e, c, mode="fixed", max_iters=2, return_stats=True)
    want, ws = oe.pgdb_process_estimate(d, e[0], c[0], A=A, mode="fixed", max_iters=2, return_stats=True)
    assert np.abs(got[0] - want).max() < 1e-11
    assert st["dykstra"][0] == ws["dykstra"] and st["backtracks"][0] == ws["backtracks"]
    assert abs(st["cost"][0] - ws["cost"]) < 1e-11
    lin = tomography.linear_inv_process_estimate_batch(design, e)
    pt = np.einsum("iojo->ij", lin[0].reshape(8, 8, 8, 8))
    assert np.abs(lin[0] - lin[0].conj().T).max() < 1e-12 and np.abs(pt - np.eye(8)).max() < 1e-9
