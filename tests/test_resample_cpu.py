"""Counter-based bootstrap generator, CPU side: the Philox4x32-10 restatement against the published
Random123 known-answer vectors, and the Beta sampler built on it against the distribution the
reference draws from (np.random.beta, tomography.py:402)."""
import numpy as np
from scipy import stats

from fbx_oracle import acquisition as A


def test_philox_known_answers():
    # Random123 kat_vectors: philox4x32_10 <counter> <key> -> <output>
    kat = [
        ([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
        ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
        ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
         [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
    ]
    for ctr, key, want in kat:
        got = A.philox4x32_10(np.array([ctr], dtype=np.uint32), np.array([key], dtype=np.uint32))[0]
        assert [int(v) for v in got] == want


def test_counter_based_properties():
    e = np.array([[0.1, -0.4, 0.9], [0.0, 1.0, -1.0]])
    c = np.array([[100.0, 50.0, 10.0], [1000.0, 20.0, 20.0]])
    r8 = A.beta_resample(e, c, 8, seed=3)
    r3 = A.beta_resample(e, c, 3, seed=3)
    assert r8.shape == (8, 2, 3)
    assert np.array_equal(r8[:3], r3)                       # element (r, i) does not depend on R
    assert not np.array_equal(r8, A.beta_resample(e, c, 8, seed=4))
    assert (np.abs(r8) <= 1).all()
    # invalid Beta parameters (|e| > 1) give NaN, nothing else does
    bad = A.beta_resample(np.array([1.5, 0.0]), np.array([10.0, 10.0]), 4, seed=0)
    assert np.isnan(bad[:, 0]).all() and np.isfinite(bad[:, 1]).all()


def test_distribution_matches_numpy_beta():
    e = np.array([0.0, 0.5, -0.9, 1.0, 0.2])
    c = np.array([1000.0, 100.0, 50.0, 10.0, 0.0])          # counts = 0: Beta(1, 1)
    R = 20000
    r = (A.beta_resample(e, c, R, seed=11) + 1) / 2
    for i in range(e.size):
        a = (e[i] + 1) / 2 * c[i] + 1
        b = c[i] - (e[i] + 1) / 2 * c[i] + 1
        assert stats.kstest(r[:, i], stats.beta(a, b).cdf).pvalue > 1e-3
        assert abs(r[:, i].mean() - a / (a + b)) < 5 * np.sqrt(a * b / ((a + b) ** 2 * (a + b + 1)) / R)
    # a prior below one exercises the a < 1 boost
    r = (A.beta_resample(np.array([0.0]), np.array([0.0]), R, prior_counts=0.5, seed=2) + 1) / 2
    assert stats.kstest(r[:, 0], stats.beta(0.5, 0.5).cdf).pvalue > 1e-3
