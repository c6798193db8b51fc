"""shots -> moments kernel vs the oracle and the reference's exact test values."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_reference_known_values(gpu):
    """tests/test_observable_estimation.py:521-550 of the reference."""
    from fbx.observable_estimation import PauliTerm, ratio_variance, shots_to_obs_moments
    bs = np.array([[0, 1]] * 1000)
    mean, var = shots_to_obs_moments(bs, [0, 1], PauliTerm({0: "Z", 1: "X"}))
    assert mean == -1.0 and var == 0.0
    assert shots_to_obs_moments(bs, [0, 1], PauliTerm({}, 0.5)) == (0.5, 0)
    assert ratio_variance(1.0, 0.1, 2.0, 0.05) == 0.028125
    assert ratio_variance(0.0, 0.1, 2.0, 0.05) == 0.025
    with pytest.raises(ValueError):
        shots_to_obs_moments(bs, [0, 1], PauliTerm({0: "Z"}, 1j))


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 8])
@pytest.mark.parametrize("shots", [1, 7, 1000, 4099])
def test_against_oracle(gpu, n, shots):
    from fbx.observable_estimation import shots_to_obs_moments_batch
    from fbx_oracle import acquisition as oa
    rs = np.random.RandomState(n * 100 + shots)
    S = 9
    bits = (rs.uniform(size=(S, shots, n)) < rs.uniform(0.05, 0.95, size=(S, 1, n))).astype(np.uint8)
    masks = (rs.uniform(size=(S, n)) < 0.6).astype(np.uint8)
    masks[0] = 0                                     # identity term
    masks[1] = 1
    coefs = rs.uniform(-2, 2, size=S)
    for prior in (False, True):
        mean, var = shots_to_obs_moments_batch(bits, masks, coefs, prior)
        for s in range(S):
            wm, wv = oa.shots_to_obs_moments(bits[s], masks[s], coefs[s], prior)
            assert abs(mean[s] - wm) < 1e-14 and abs(var[s] - wv) < 1e-15 + 1e-13 * abs(wv)


def test_large_stream_exact_counts(gpu):
    """10^7 shots of 2 bytes: the integer counts must be exact (no float accumulation)."""
    from fbx.observable_estimation import shots_to_obs_moments_batch
    shots = 10_000_001
    bits = np.zeros((1, shots, 2), dtype=np.uint8)
    bits[0, ::3, 0] = 1
    bits[0, ::5, 1] = 1
    mean, var = shots_to_obs_moments_batch(bits, [[1, 1]])
    par = (bits[0, :, 0] ^ bits[0, :, 1]).astype(np.int64)
    m = (shots - 2 * par.sum()) / shots
    assert mean[0] == m and abs(var[0] - (1 - m * m) / shots) < 1e-20


@pytest.mark.parametrize("n,shots", [(3, 1000), (3, 1003), (5, 1000), (6, 24), (7, 1048), (2, 1000), (8, 40), (9, 1000), (3, 20000)])
def test_every_counting_path_is_exact(gpu, n, shots):
    """The wave-per-setting and workgroup-per-setting forms, the 16-byte path (n | 16), the packed path of
    n = 3, 5, 6, 7 (runs of 16 shots + byte-wise tail) and the byte-wise fallback (n = 9; records that are not
    8-byte aligned, shots = 1003): means equal numpy's to the last bit for every setting."""
    from fbx import _lib
    rs = np.random.RandomState(n * 1000 + shots)
    S = 37
    bits = rs.randint(0, 2, size=(S, shots, n)).astype(np.uint8)
    mask = rs.randint(0, 2, size=(S, n)).astype(np.uint8)
    mask[0] = 0                                                     # an identity term
    mean, var = np.empty(S), np.empty(S)
    _lib.check(_lib.lib().fbx_shots_to_moments(n, S, shots, bits.ctypes.data_as(_lib.C.POINTER(_lib.C.c_uint8)),
                                               mask.ctypes.data_as(_lib.C.POINTER(_lib.C.c_uint8)), None, 0,
                                               _lib.dptr(mean), _lib.dptr(var)))
    prod = np.where(mask[:, None, :] != 0, 1 - 2 * bits.astype(np.int64), 1).prod(axis=2)
    want = prod.sum(axis=1) / shots
    assert np.array_equal(mean, want)
    assert np.allclose(var[1:], (1 - want[1:] ** 2) / shots, rtol=1e-15, atol=0) and var[0] == 0.0
