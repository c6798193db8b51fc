"""The same known answers evaluated through the HIP library (reference function names)."""
import numpy as np
import pytest

import known_answers as ka

pytestmark = pytest.mark.gpu


def test_conversions(gpu):
    from fbx import operator_tools as ot
    ka.check_conversions(ot)


def test_projections(gpu):
    from fbx import operator_tools as ot
    ka.check_projections(ot)


def test_process_fidelity(gpu):
    from fbx import distance_measures as dm, operator_tools as ot
    ka.check_process_fidelity(ot, dm)


def test_non_square_kraus_to_superop(gpu):
    """tests/test_superoperator_transformations.py:167-173: M_0 = 1 (x) <0| is 2 x 4."""
    from fbx.operator_tools import kraus2superop
    m0 = np.kron(np.eye(2), np.array([[1, 0]]))
    assert np.allclose(kraus2superop(m0), np.kron(m0.conj(), m0))
    rng = np.random.default_rng(0)
    ks = [rng.normal(size=(4, 2)) + 1j * rng.normal(size=(4, 2)) for _ in range(3)]
    want = sum(np.kron(k.conj(), k) for k in ks)
    got = kraus2superop(ks)
    assert got.shape == (16, 4) and np.abs(got - want).max() < 1e-13
