"""Shots -> observable moments -- CPU restatement of observable_estimation.py:804-853, :1052-1090.

TEST INFRASTRUCTURE (see package docstring).  Pinned by the exact values of the reference's tests
(tests/test_observable_estimation.py:521-550) and, in the build container, against the reference."""
import numpy as np
from scipy.stats import beta


def shots_to_obs_moments(bitarray, obs_mask, coeff=1.0, use_beta_dist_unbiased_prior=False):
    """obs_mask[q] != 0 where the observable acts on column q of the [n_shots, n_qubits] bit array."""
    idxs = [i for i, m in enumerate(obs_mask) if m]
    if len(idxs) == 0:
        return coeff, 0
    obs_strings = np.asarray(bitarray)[:, idxs]
    obs_vals = np.prod(1 - 2 * obs_strings.astype(np.int64), axis=1)
    if use_beta_dist_unbiased_prior:
        n_minus, n_plus = np.bincount(obs_vals == 1, minlength=2)
        m, v = beta.mean(n_plus + 1, n_minus + 1), beta.var(n_plus + 1, n_minus + 1)
        return (2 * m - 1) * coeff, 4 * v * coeff ** 2
    obs_vals = coeff * obs_vals
    return np.mean(obs_vals).item(), np.var(obs_vals).item() / len(bitarray)


def ratio_variance(a, var_a, b, var_b):
    """observable_estimation.py:1052-1090."""
    return var_a / b ** 2 + (a ** 2 * var_b) / b ** 4
