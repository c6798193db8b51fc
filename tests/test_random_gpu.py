"""Device random operators (fbx_random_operators / fbx_random_kraus) against the
oracle's restatement of the same counter-based stream and the reference's arithmetic
(operator_tools/random_operators.py:21-157), plus the distributional properties the reference's own
tests check (tests/test_random_operators.py: unitarity, trace one, positivity, CPTP)."""
import numpy as np
import pytest

from fbx_oracle import acquisition as oa, superops as so

pytestmark = pytest.mark.gpu


def test_ginibre_matches_the_oracle_stream_and_is_standard_normal(gpu):
    from fbx.operator_tools import random_operators as ro
    g = ro.ginibre_matrix_complex_batch(4, 3, 50000, seed=11)
    for item in (0, 1, 49999):
        assert np.abs(g[item] - oa.ginibre_matrix(11, item, 4, 3)).max() < 1e-13
    x = np.concatenate([g.real.ravel(), g.imag.ravel()])
    assert abs(x.mean()) < 5e-3 and abs(x.var() - 1) < 1e-2 and abs((x ** 4).mean() - 3) < 5e-2
    assert abs(np.mean(g.real * g.imag)) < 5e-3
    # an item does not depend on the batch it is generated in
    part = ro.ginibre_matrix_complex_batch(4, 3, 10, seed=11, first_item=49990)
    assert np.array_equal(part, g[49990:])
    assert not np.array_equal(ro.ginibre_matrix_complex_batch(4, 3, 4, seed=12), g[:4])


@pytest.mark.parametrize("dim", [2, 4, 8])
def test_haar_unitaries_states_and_mixed_states(gpu, dim):
    from fbx.operator_tools import random_operators as ro
    B = 2000
    u = ro.haar_rand_unitary_batch(dim, B, seed=3)
    assert np.abs(u.conj().transpose(0, 2, 1) @ u - np.eye(dim)).max() < 1e-13
    for item in (0, 7, B - 1):
        assert np.abs(u[item] - oa.haar_unitary(3, item, dim)).max() < 1e-12
    # Haar moments: E |U_ij|^2 = 1 / d, E U_ij = 0
    assert np.abs((np.abs(u) ** 2).mean(axis=0) - 1 / dim).max() < 6 / np.sqrt(B)
    assert np.abs(u.mean(axis=0)).max() < 6 / np.sqrt(B)
    psi = ro.haar_rand_state_batch(dim, B, seed=3)
    assert psi.shape == (B, dim, 1) and np.array_equal(psi[:, :, 0], u[:, :, 0])
    for rank in (1, dim):
        rho = ro.ginibre_state_matrix_batch(dim, rank, 500, seed=5)
        assert np.abs(np.trace(rho, axis1=1, axis2=2) - 1).max() < 1e-14
        w = np.linalg.eigvalsh(rho)
        assert w.min() > -1e-14 and (np.sum(w > 1e-12, axis=1) == rank).all()
        assert np.abs(rho[3] - oa.ginibre_state(5, 3, dim, rank)).max() < 1e-13
    rho = ro.bures_measure_state_matrix_batch(dim, 500, seed=6)
    assert np.abs(np.trace(rho, axis1=1, axis2=2) - 1).max() < 1e-14 and np.linalg.eigvalsh(rho).min() > -1e-14
    assert np.abs(rho[9] - oa.bures_state(6, 9, dim)).max() < 1e-12
    with pytest.raises(ValueError):
        ro.ginibre_state_matrix_batch(dim, dim + 1, 2)


@pytest.mark.parametrize("dim,K", [(2, 1), (2, 4), (4, 4), (4, 16), (8, 3)])
def test_random_kraus_sets_are_cptp_and_match_the_oracle(gpu, dim, K):
    from fbx.operator_tools import random_operators as ro
    B = 300
    ks = ro.random_kraus_batch(dim, K, B, seed=17, first_item=1000)
    tp = np.einsum("bkji,bkjl->bil", ks.conj(), ks)
    assert np.abs(tp - np.eye(dim)).max() < 1e-12
    for b in (0, 5, B - 1):
        assert np.abs(ks[b] - oa.random_kraus(17, 1000 + b, dim, K)).max() < 1e-11
    choi = ro.rand_map_with_BCSZ_dist_batch(dim, K, B, seed=17, first_item=1000)
    for b in (0, B - 1):
        assert np.abs(choi[b] - so.kraus2choi(list(ks[b]))).max() < 1e-12
    # Choi of a CPTP map: PSD of rank <= K, partial trace over the output = identity
    w = np.linalg.eigvalsh(choi)
    assert w.min() > -1e-12 and (np.sum(w > 1e-10, axis=1) <= K).all()
    pt = np.einsum("biojo->bij", choi.reshape(B, dim, dim, dim, dim))
    assert np.abs(pt - np.eye(dim)).max() < 1e-12


def test_bad_arguments(gpu):
    from fbx.operator_tools import random_operators as ro
    with pytest.raises(ValueError):
        ro.haar_rand_unitary_batch(3, 2)
    with pytest.raises(ValueError):
        ro.random_kraus_batch(4, 0, 2)
    assert ro.haar_rand_unitary_batch(4, 0).shape == (0, 4, 4)
