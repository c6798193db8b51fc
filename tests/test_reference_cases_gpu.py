"""The cases of the reference's own test-suite that the other GPU test files did not touch yet, one test per
reference test (values and thresholds are the reference's; the test bodies are this repository's):
  tests/test_channel_approximation.py:20-36      Pauli twirl of a Pauli channel / of amplitude damping (arXiv:1701.03708 eq. 7)
  tests/test_distance_measures.py:61-67          smith_fidelity
  tests/test_validate_operators.py:19-76         is_square / symmetric / identity / idempotent / normal / hermitian /
                                                 unitary / positive (semi)definite, incl. the ValueErrors and a 3 x 3 matrix
  tests/test_validate_superoperator.py:9-61      Kraus validity, hermiticity / trace preservation / CP (D = 2 and D = 3),
                                                 unital, unitary
  tests/test_superoperator_transformations.py:117-136,215-224,281-287   vec / unvec, Kraus completeness, superop2kraus,
                                                 the Choi <-> superoperator reshuffle as an involution
  tests/test_distance_measures.py:23-46,138-176  purity (dim_renorm, qutrit), QCB of mixed states, Hilbert-Schmidt
                                                 inner product (Cauchy-Schwarz, linearity)
  tests/test_random_operators.py:183-441         first / second moments of the Haar unitaries, purity moments of the
                                                 Ginibre and Bures ensembles, BCSZ maps -- on the DEVICE streams
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


# ----------------------------------------------------------------------------------------------- more measures
def test_purity_standard_and_renormalised(gpu):
    """tests/test_distance_measures.py:23-46, incl. the qutrit."""
    from fbx import distance_measures as dm
    r0, r1, r2, r3 = np.diag([1.0, 0]), np.diag([0.9, 0.1]), np.diag([0.5, 0.5]), np.eye(3) / 3
    assert dm.purity(r0) == 1.0 and np.allclose(dm.purity(r1), 0.82) and dm.purity(r2) == 0.5
    assert np.isclose(dm.purity(r3), 1 / 3, rtol=0, atol=1e-15)
    assert dm.purity(r0, dim_renorm=True) == 1.0
    assert np.allclose(dm.purity(r1, dim_renorm=True), 2 * (0.82 - 0.5))
    assert dm.purity(r2, dim_renorm=True) == 0.0
    assert abs(dm.purity(r3, dim_renorm=True)) < 1e-15


def test_hilbert_schmidt_ip_cauchy_schwarz_and_linearity(gpu):
    """tests/test_distance_measures.py:155-176."""
    from fbx import distance_measures as dm
    from fbx.operator_tools import haar_rand_unitary
    ur = haar_rand_unitary(2, np.random.RandomState(7))
    u = ur + ur.conj().T
    a, b = np.eye(2), np.eye(2) / 3
    ip = dm.hilbert_schmidt_ip
    assert ip(u, u) * ip(a, a) >= abs(ip(a, u)) ** 2 and ip(u, u) * ip(b, b) >= abs(ip(b, u)) ** 2
    assert np.allclose(0.17 * ip(u, a) + 0.6713 * ip(u, b), ip(u, 0.17 * a + 0.6713 * b))


def test_qcb_for_mixed_states(gpu):
    """tests/test_distance_measures.py:138-149: a 0.9 / 0.1 state against its 90-degree rotation."""
    from fbx import distance_measures as dm
    g = np.array([[0.0, -1.0], [1.0, 0.0]])
    rho = np.diag([0.9, 0.1])
    qcb, s = dm.quantum_chernoff_bound(rho, g @ rho @ g.T)
    assert np.allclose(qcb, 0.6)


def test_choi_superop_reshuffle_is_an_involution(gpu):
    """tests/test_superoperator_transformations.py:281-287."""
    from fbx.operator_tools import choi2superop, kraus2choi, kraus2superop, superop2choi
    assert np.allclose(choi2superop(choi2superop(np.eye(4))), np.eye(4))
    assert np.allclose(superop2choi(superop2choi(np.eye(4))), np.eye(4))
    hc, hs = kraus2choi(H), kraus2superop(H)
    assert np.allclose(choi2superop(choi2superop(hc)), hc) and np.allclose(superop2choi(superop2choi(hs)), hs)


# ----------------------------------------------------------------------------------------------- moments of the streams
def test_random_unitaries_first_moment(gpu):
    """tests/test_random_operators.py:183-214 on the DEVICE stream: E[U (x) U^dagger] = SWAP / D
    (arXiv:0809.3813 p. 2), 200 000 unitaries per dimension instead of 50 000."""
    from fbx.operator_tools import permute_tensor_factors
    from fbx.operator_tools.random_operators import haar_rand_unitary_batch
    for dim in (2, 4):
        u = haar_rand_unitary_batch(dim, 200_000, seed=21)
        avg = np.einsum("bij,bkl->ikjl", u, u.conj().transpose(0, 2, 1)).reshape(dim * dim, dim * dim) / u.shape[0]
        swap = permute_tensor_factors(dim, [1, 0])
        assert np.linalg.norm(avg - swap / dim) <= 0.02


def test_random_unitaries_second_moment(gpu):
    """tests/test_random_operators.py:218-285 on the device stream: E[U (x) U (x) U^dagger (x) U^dagger] for D = 2,
    eq. 5.17 of arXiv:0711.1017."""
    from fbx.operator_tools import permute_tensor_factors
    from fbx.operator_tools.random_operators import haar_rand_unitary_batch
    u = haar_rand_unitary_batch(2, 200_000, seed=22)
    ud = u.conj().transpose(0, 2, 1)
    var = np.einsum("bai,bcj,bek,bgl->acegijkl", u, u, ud, ud, optimize=True).reshape(16, 16) / u.shape[0]
    p = {name: permute_tensor_factors(2, perm) for name, perm in
         (("3412", [2, 3, 0, 1]), ("4321", [3, 2, 1, 0]), ("4312", [3, 2, 0, 1]), ("3421", [2, 3, 1, 0]))}
    want = (p["3412"] + p["4321"]) / 3 - (p["4312"] + p["3421"]) / 6
    assert np.allclose(np.around(want, 2), np.around(var.real, 2), atol=0.01)


def test_ginibre_and_bures_second_moments(gpu):
    """tests/test_random_operators.py:352-420 on the device streams: <tr rho^2> = (D + K) / (D K + 1)
    (Zyczkowski & Sommers 2001, eq. 3.20) and (5 D^2 + 1) / (2 D (D^2 + 2)) (Sommers & Zyczkowski 2004, eq. 3.1)."""
    from fbx.operator_tools.random_operators import bures_measure_state_matrix_batch, ginibre_state_matrix_batch
    for dim in (2, 4):
        rho = ginibre_state_matrix_batch(dim, 2, 20_000, seed=23)
        assert abs(np.einsum("bij,bji->b", rho, rho).real.mean() - (dim + 2) / (dim * 2 + 1)) < 1e-2
        rho = bures_measure_state_matrix_batch(dim, 20_000, seed=24)
        assert abs(np.einsum("bij,bji->b", rho, rho).real.mean() - (5 * dim ** 2 + 1) / (2 * dim * (dim ** 2 + 2))) < 1e-2


def test_bcsz_maps_from_the_device_are_cptp(gpu):
    """tests/test_random_operators.py:425-441 for a batch from the device stream."""
    from fbx.operator_tools import choi_is_completely_positive, choi_is_trace_preserving
    from fbx.operator_tools.random_operators import rand_map_with_BCSZ_dist_batch
    for choi in rand_map_with_BCSZ_dist_batch(2, 2, 10, seed=25):
        assert choi_is_completely_positive(choi) and choi_is_trace_preserving(choi)
