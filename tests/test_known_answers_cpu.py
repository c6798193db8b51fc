"""Known answers of the reference's tests/docs evaluated on the CPU oracle (runs everywhere)."""
import numpy as np

import known_answers as ka
from fbx_oracle import measures as om, superops as so


def test_conversions():
    ka.check_conversions(so)


def test_projections():
    ka.check_projections(so)


def test_process_fidelity():
    ka.check_process_fidelity(so, om)


def test_sic_povm_properties():
    """tests/test_observable_estimation.py:873-912: sum_i |s_i><s_i| = 2 I and tr(Pi_a Pi_b) = 1/3."""
    from fbx_oracle import design as od
    vs = od.STATE_VECTORS[6:10]
    assert np.allclose(sum(np.outer(v, v.conj()) for v in vs), 2 * np.eye(2))
    for a in range(4):
        for b in range(a + 1, 4):
            assert np.isclose(abs(np.vdot(vs[a], vs[b])) ** 2, 1 / 3)


def test_trace_distance_is_half_the_induced_one_norm():
    rho = np.array([[0.7, 0.2], [0.2, 0.3]]); sig = np.eye(2) / 2
    assert np.isclose(om.trace_distance(rho, sig), 0.5 * np.abs(rho - sig).sum(0).max())
