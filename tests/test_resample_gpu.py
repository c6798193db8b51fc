"""Device bootstrap resampling (fbx_beta_resample) against the CPU statement of the same counter-based
generator, and the resident process-fidelity bootstrap built on it."""
import numpy as np
import pytest
from scipy import stats

pytestmark = pytest.mark.gpu


def test_generator_matches_oracle(gpu):
    from fbx import tomography
    from fbx_oracle import acquisition as A
    rng = np.random.default_rng(0)
    e = np.clip(rng.normal(0, 0.6, (7, 33)), -1, 1)
    e[0, :3] = [1.0, -1.0, 0.0]
    c = rng.integers(0, 2000, (7, 33)).astype(float)
    for seed, prior in ((0, 1.0), (2 ** 40 + 17, 1.0), (5, 0.25)):
        got = tomography.resample_expectations_with_beta_batch(e, c, 9, prior_counts=prior, seed=seed)
        want = A.beta_resample(e, c, 9, prior_counts=prior, seed=seed)
        assert got.shape == want.shape == (9, 7, 33)
        # integer stream bit-exact; the transcendental functions differ in the last bits, which can
        # (very rarely) flip an acceptance test -> allow a handful of elements to differ
        close = np.abs(got - want) <= 1e-12
        assert close.mean() > 0.999, close.mean()
    # prefix property and seed sensitivity on the device
    a = tomography.resample_expectations_with_beta_batch(e, c, 4, seed=1)
    b = tomography.resample_expectations_with_beta_batch(e, c, 2, seed=1)
    assert np.array_equal(a[:2], b)
    assert not np.array_equal(a, tomography.resample_expectations_with_beta_batch(e, c, 4, seed=2))


def test_distribution_and_bad_parameters(gpu):
    from fbx import tomography, _lib
    e = np.array([0.3, -0.95, 2.0])
    c = np.array([400.0, 30.0, 10.0])
    R = 50000
    r = tomography.resample_expectations_with_beta_batch(e, c, R, seed=123)
    assert np.isnan(r[:, 2]).all()
    for i in range(2):
        a = (e[i] + 1) / 2 * c[i] + 1
        b = c[i] - (e[i] + 1) / 2 * c[i] + 1
        x = (r[:, i] + 1) / 2
        assert stats.kstest(x, stats.beta(a, b).cdf).pvalue > 1e-3
    with pytest.raises(ValueError):
        tomography.resample_expectations_with_beta_batch(e, c, 3, prior_counts=0.0, seed=1)


def test_estimate_variance_with_device_generator(gpu):
    """The state bootstrap with the device stream agrees statistically with the reference stream."""
    from fbx import tomography, synthetic
    from fbx import distance_measures as dm
    from fbx.observable_estimation import ExperimentResult
    _, _, e, c = synthetic.state_batch(1, 1, shots=2000, first_item=4, mixed=0.1)
    settings = tomography.generate_state_tomography_settings([0])
    results = [ExperimentResult(s, float(e[0][k]), int(c[0][k])) for k, s in enumerate(settings)]
    np.random.seed(0)
    m_ref, v_ref = tomography.estimate_variance(results, [0], tomography.linear_inv_state_estimate, dm.purity,
                                                n_resamples=400)
    m_dev, v_dev = tomography.estimate_variance(results, [0], tomography.linear_inv_state_estimate, dm.purity,
                                                n_resamples=400, seed=9)
    assert abs(m_ref - m_dev) < 5 * np.sqrt(v_ref / 400) + 1e-12
    assert 0.6 < v_dev / v_ref < 1.6


def test_process_fidelity_bootstrap_resident(gpu):
    from fbx import tomography, synthetic
    from fbx.design import process_design
    from fbx.operator_tools import convert_batch
    from fbx import distance_measures as dm
    n, B, R = 1, 5, 24
    design, us, e, c = synthetic.process_batch(n, "pauli", B, shots=500)
    ideal = convert_batch("kraus", "pauli_liouville", us[:, None])
    mean, var, samples = tomography.process_fidelity_variance_batch(design, e, c, ideal, n_resamples=R, seed=3,
                                                                    return_samples=True)
    assert samples.shape == (R, B) and mean.shape == var.shape == (B,)
    # the same numbers step by step through the host-pointer entry points
    e_rs = tomography.resample_expectations_with_beta_batch(e, c, R, seed=3)
    choi = tomography.pgdb_process_estimate_batch(design, e_rs.reshape(R * B, -1), np.tile(c, (R, 1)))
    ptm = convert_batch("choi", "pauli_liouville", choi)
    want = dm.process_fidelity_batch(np.tile(ideal, (R, 1, 1)), ptm).reshape(R, B)
    assert np.array_equal(samples, want)
    assert np.allclose(mean, want.mean(axis=0)) and np.allclose(var, want.var(axis=0))
    # error bars have the right scale: the point estimate lies within a few bootstrap sigmas of the mean
    point = dm.process_fidelity_batch(ideal, convert_batch("choi", "pauli_liouville",
                                                           tomography.pgdb_process_estimate_batch(design, e, c)))
    assert (np.abs(point - mean) < 6 * np.sqrt(var) + 5e-3).all()
    assert (var > 0).all() and (mean > 0.9).all()
