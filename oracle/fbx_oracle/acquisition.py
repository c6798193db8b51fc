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


def estimate_dfe(expectations, std_errs, n_qubits, kind):
    """direct_fidelity_estimation.py:286-307 on plain arrays: (mean fidelity, standard error).

    state:   F = 1/d + (1 - 1/d) mean(e)
    process: average gate fidelity from the Choi-state fidelity p = 1/d^2 + (1 - 1/d^2) mean(e),
             F = (d^2 p + d) / (d^2 + d)
    The error bar is the root of the summed squared standard errors over m, times dF/dmean(e)."""
    d = 2.0 ** n_qubits
    e = np.asarray(expectations, dtype=float)
    s = np.asarray(std_errs, dtype=float)
    rms = np.sqrt(np.sum(s * s)) / e.size
    which = kind.lower()
    if which == "state":
        slope = 1.0 - 1.0 / d
        return 1.0 / d + slope * e.mean(), slope * rms
    if which == "process":
        slope = 1.0 - 1.0 / d ** 2
        p = 1.0 / d ** 2 + slope * e.mean()
        return (d * d * p + d) / (d * d + d), d / (d + 1.0) * slope * rms
    raise ValueError("Kind can only be 'state' or 'process'.")
