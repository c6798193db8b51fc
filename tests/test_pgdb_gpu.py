"""Parity of the HIP PGDB process-tomography path against the CPU oracle (through the C ABI)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CHOI_TOL = 1e-9        # max-abs on Choi entries (SURVEY.md 8d parity tolerance)
FID_TOL = 1e-8         # process fidelity


def _oracle_design(design):
    from fbx_oracle import design as od
    return od.Design(design.n_qubits, design.kind, design.in_labels, design.paulis, design.coefs)


def _oracle_pgdb(design, e, c, **kw):
    from fbx_oracle import estimators as oe
    od_ = _oracle_design(design)
    A = oe.design_matrix_A(od_)
    outs, stats = [], []
    for b in range(e.shape[0]):
        est, st = oe.pgdb_process_estimate(od_, e[b], c[b], A=A, return_stats=True, **kw)
        outs.append(est)
        stats.append(st)
    return np.array(outs), stats


def _process_fidelity_to_truth(choi, u):
    from fbx_oracle import superops as so, measures as om
    return om.process_fidelity(so.kraus2pauli_liouville(u), so.choi2pauli_liouville(choi))


@pytest.mark.parametrize("n,basis,batch", [(1, "pauli", 6), (1, "sic", 6), (2, "sic", 4), (2, "pauli", 4)])
def test_pgdb_converge_matches_oracle(gpu, n, basis, batch):
    from fbx import synthetic, tomography
    design, us, e, c = synthetic.process_batch(n, basis, batch)
    got, st = tomography.pgdb_process_estimate_batch(design, e, c, return_stats=True)
    want, wst = _oracle_pgdb(design, e, c)
    assert np.abs(got - want).max() < CHOI_TOL
    for b in range(batch):
        assert st["iterations"][b] == wst[b]["iterations"]
        assert st["dykstra"][b] == wst[b]["dykstra"]
        # Halvings inside a *stalled* final iteration compare costs that differ by rounding
        # noise only (inexact Dykstra projection -> ascent direction, alpha -> 0), so the count
        # is summation-order dependent there; everything else must agree exactly.
        assert abs(int(st["backtracks"][b]) - wst[b]["backtracks"]) <= 50
        assert abs(st["cost"][b] - wst[b]["cost"]) < 1e-10
        f_got = _process_fidelity_to_truth(got[b], us[b])
        f_want = _process_fidelity_to_truth(want[b], us[b])
        assert abs(f_got - f_want) < FID_TOL


@pytest.mark.parametrize("n,basis", [(1, "pauli"), (2, "pauli")])
def test_pgdb_fixed_100_matches_oracle(gpu, n, basis):
    from fbx import synthetic, tomography
    design, us, e, c = synthetic.process_batch(n, basis, 3)
    got, st = tomography.pgdb_process_estimate_batch(design, e, c, mode="fixed", max_iters=100,
                                                     return_stats=True)
    want, wst = _oracle_pgdb(design, e, c, mode="fixed", max_iters=100)
    assert (st["iterations"] == 100).all()
    assert np.abs(got - want).max() < CHOI_TOL
    for b in range(3):
        assert abs(_process_fidelity_to_truth(got[b], us[b])
                   - _process_fidelity_to_truth(want[b], us[b])) < FID_TOL


def test_pgdb_trace_non_increasing(gpu):
    from fbx import synthetic, tomography
    design, us, e, c = synthetic.process_batch(1, "pauli", 4)
    got = tomography.pgdb_process_estimate_batch(design, e, c, trace_preserving=False)
    want, _ = _oracle_pgdb(design, e, c, trace_preserving=False)
    assert np.abs(got - want).max() < CHOI_TOL
