"""The cases of the reference's own test-suite that the other GPU test files did not touch yet, one test per
reference test (values and thresholds are the reference's; the test bodies are this repository's):
  tests/test_channel_approximation.py:20-36      Pauli twirl of a Pauli channel / of amplitude damping (arXiv:1701.03708 eq. 7)
  tests/test_distance_measures.py:61-67          smith_fidelity
  tests/test_validate_operators.py:19-76         is_square / symmetric / identity / idempotent / normal / hermitian /
                                                 unitary / positive (semi)definite, incl. the ValueErrors and a 3 x 3 matrix
  tests/test_validate_superoperator.py:9-61      Kraus validity, hermiticity / trace preservation / CP (D = 2 and D = 3),
                                                 unital, unitary
  tests/test_superoperator_transformations.py:117-136,215-224   vec / unvec, Kraus completeness, superop2kraus
"""
import numpy as np
import pytest

from known_answers import (H, HAD_CHOI, IZ_KRAUS, IZ_SUPER, X, Y, Z, amplitude_damping_chi, amplitude_damping_choi,
                           amplitude_damping_kraus, amplitude_damping_super)

pytestmark = pytest.mark.gpu

NORMAL_NOT_UNITARY = np.array([[1, 1, 0], [0, 1, 1], [1, 0, 1]])          # en.wikipedia.org/wiki/Normal_matrix
SIGMA_MINUS = (X + 1j * Y) / 2
PROJ_ZERO = np.array([[1, 0], [0, 0]])


def pauli_channel_chi(px, py, pz):
    return np.diag([1 - px - py - pz, px, py, pz]).astype(complex)


# ----------------------------------------------------------------------------------------------- twirling
def test_pauli_twirl_of_pauli_channel(gpu):
    from fbx.operator_tools import pauli_twirl_chi_matrix
    rs = np.random.RandomState(1)
    for _ in range(4):
        chi = pauli_channel_chi(*(rs.rand(3) / 3))
        assert np.allclose(chi, pauli_twirl_chi_matrix(chi))             # already diagonal: unchanged


def test_pauli_twirl_of_amp_damp(gpu):
    from fbx.operator_tools import pauli_twirl_chi_matrix
    for p in (0.0, 0.13, 0.5, 0.97, 1.0):
        s = np.sqrt(1 - p)
        want = np.diag([(2 + 2 * s - p) / 4, p / 4, p / 4, (2 - 2 * s - p) / 4])
        assert np.allclose(want, pauli_twirl_chi_matrix(amplitude_damping_chi(p)))


# ----------------------------------------------------------------------------------------------- measures
def test_smith_fidelity(gpu):
    from fbx import distance_measures as dm
    pure = np.array([[1.0, 0.0], [0.0, 0.0]])
    assert np.isclose(dm.smith_fidelity(np.eye(2) / 2, pure, 0.001), 0.9996534864594093, rtol=0.01)
    # power 1 is the square of the ordinary fidelity's root: tr sqrt(sqrt(rho) sigma sqrt(rho))
    assert np.isclose(dm.smith_fidelity(np.eye(2) / 2, pure, 1) ** 2, dm.fidelity(np.eye(2) / 2, pure))


# ----------------------------------------------------------------------------------------------- validators
def test_is_square_matrix(gpu):
    from fbx.operator_tools import is_square_matrix
    assert is_square_matrix(np.eye(3))
    with pytest.raises(ValueError):
        is_square_matrix(np.zeros((2, 2, 2)))
    assert not is_square_matrix(np.array([[1, 0]]))


def test_is_symmetric_matrix(gpu):
    from fbx.operator_tools import is_symmetric_matrix
    assert is_symmetric_matrix(X) and not is_symmetric_matrix(SIGMA_MINUS)
    for bad in (np.zeros((2, 2, 2)), np.array([[1, 0]])):
        with pytest.raises(ValueError):
            is_symmetric_matrix(bad)


def test_is_identity_matrix(gpu):
    from fbx.operator_tools import is_identity_matrix
    assert not is_identity_matrix(Z) and is_identity_matrix(np.eye(3))
    for bad in (np.zeros((2, 2, 2)), np.array([[1, 0]])):
        with pytest.raises(ValueError):
            is_identity_matrix(bad)


def test_is_idempotent_matrix(gpu):
    from fbx.operator_tools import is_idempotent_matrix
    assert not is_idempotent_matrix(SIGMA_MINUS)
    assert is_idempotent_matrix(PROJ_ZERO) and is_idempotent_matrix(np.diag([0, 1, 0]))


def test_is_normal_hermitian_unitary(gpu):
    from fbx.operator_tools import is_hermitian_matrix, is_normal_matrix, is_unitary_matrix, haar_rand_unitary
    assert is_normal_matrix(NORMAL_NOT_UNITARY) and not is_normal_matrix(SIGMA_MINUS)
    assert not is_hermitian_matrix(NORMAL_NOT_UNITARY) and is_hermitian_matrix(X) and is_hermitian_matrix(Y)
    assert not is_unitary_matrix(NORMAL_NOT_UNITARY) and is_unitary_matrix(Y)
    assert is_unitary_matrix(haar_rand_unitary(4))


def test_positive_definite_and_semidefinite_thresholds(gpu):
    """atol = 1e-8 on the eigenvalues (validate_operator.py:118-150); the eigenvalues come from fbx_eigh."""
    from fbx.operator_tools import is_positive_definite_matrix, is_positive_semidefinite_matrix
    assert not is_positive_definite_matrix(np.array([[-1e-08, 0], [0, 0.1]]))
    assert is_positive_definite_matrix(np.array([[0.5e-08, 0], [0, 0.1]]))
    assert not is_positive_semidefinite_matrix(np.array([[-1e-07, 0], [0, 0.1]]))
    assert is_positive_semidefinite_matrix(np.array([[-1e-08, 0], [0, 0.1]]))
    assert is_positive_semidefinite_matrix(np.array([[0.5e-08, 0], [0, 0.1]]))


def test_kraus_operators_are_valid(gpu):
    from fbx.operator_tools import kraus_operators_are_valid
    assert kraus_operators_are_valid(amplitude_damping_kraus(0.37))
    assert kraus_operators_are_valid(H)                                   # a single operator, not a list
    assert not kraus_operators_are_valid(amplitude_damping_kraus(0.1)[0])


def test_random_bcsz_maps_are_hermiticity_and_trace_preserving_and_cp(gpu):
    from fbx.operator_tools import (choi_is_completely_positive, choi_is_hermitian_preserving, choi_is_trace_preserving,
                                    rand_map_with_BCSZ_dist)
    choi = rand_map_with_BCSZ_dist(2, 2)
    assert choi_is_hermitian_preserving(choi) and choi_is_trace_preserving(choi) and choi_is_completely_positive(choi)
    assert choi_is_completely_positive(rand_map_with_BCSZ_dist(3, 2))    # a qutrit map: 9 x 9 Choi matrix


def test_choi_is_unital_and_unitary(gpu):
    from fbx.operator_tools import chi2choi, choi_is_unital, choi_is_unitary
    p = np.array([0.2, 0.5, 0.3]) * 0.9
    choi = chi2choi(pauli_channel_chi(*p))
    assert choi_is_unital(choi) and not choi_is_unitary(choi)
    assert choi_is_unital(HAD_CHOI) and choi_is_unitary(HAD_CHOI)
    assert not choi_is_unital(amplitude_damping_choi(0.1)) and not choi_is_unitary(amplitude_damping_choi(0.1))


# ----------------------------------------------------------------------------------------------- vec / Kraus
def test_vec_and_unvec(gpu):
    from fbx.operator_tools import unvec, vec
    a = np.array([[1, 2], [3, 4]])
    b = np.array([[1, 2, 5], [3, 4, 6]])
    c = np.arange(1, 10).reshape(3, 3)
    assert np.array_equal(vec(a), [[1], [3], [2], [4]])                  # column stacking
    assert np.array_equal(vec(b), [[1], [3], [2], [4], [5], [6]])
    assert np.array_equal(unvec(vec(a)), a) and np.array_equal(unvec(vec(c)), c)


def test_kraus_ops_sum_to_identity():
    k0, k1 = amplitude_damping_kraus(0.61)
    np.testing.assert_array_almost_equal_nulp(k0.conj().T @ k0 + k1.conj().T @ k1, np.eye(2))


def test_superop_to_kraus(gpu):
    from fbx.operator_tools import superop2kraus
    assert np.allclose(superop2kraus(IZ_SUPER), IZ_KRAUS)
    for p in (0.2, 0.75):
        ops = superop2kraus(amplitude_damping_super(p))
        # eigenvalue order puts the damping operator first; signs are the eigenvectors' (as in the reference)
        assert np.allclose([np.abs(ops[1]), np.abs(ops[0])], amplitude_damping_kraus(p))
