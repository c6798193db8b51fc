"""State / process distance measures -- CPU restatement of distance_measures.py.

TEST INFRASTRUCTURE (see package docstring).  diamond_norm_distance (cvxpy SDP,
distance_measures.py:378-437) is not restated: cvxpy is absent from this image and an SDP
solver is not part of the accelerated path.
"""
import numpy as np
from scipy.linalg import fractional_matrix_power
from scipy.optimize import minimize_scalar

from .superops import sqrtm_psd


def _item(x, tol):
    return np.ndarray.item(np.real_if_close(x, tol))


def purity(rho, dim_renorm=False, tol=1000):
    """distance_measures.py:14-36."""
    p = np.trace(rho @ rho)
    if dim_renorm:
        dim = rho.shape[0]
        p = (dim / (dim - 1.0)) * (p - 1.0 / dim)
    return _item(p, tol)


def impurity(rho, dim_renorm=False, tol=1000):
    """distance_measures.py:39-61."""
    imp = 1 - np.trace(rho @ rho)
    if dim_renorm:
        dim = rho.shape[0]
        imp = (dim / (dim - 1.0)) * imp
    return _item(imp, tol)


def fidelity(rho, sigma, tol=1000):
    """distance_measures.py:64-84."""
    sqrt_rho = sqrtm_psd(rho)
    fid = (np.trace(sqrtm_psd(sqrt_rho @ sigma @ sqrt_rho))) ** 2
    return _item(fid, tol)


def infidelity(rho, sigma, tol=1000):
    """distance_measures.py:87-97."""
    return 1 - fidelity(rho, sigma, tol)


def trace_distance(rho, sigma):
    """distance_measures.py:100-114 -- note: induced 1-norm (max abs column sum)."""
    return 0.5 * np.linalg.norm(rho - sigma, 1)


def bures_distance(rho, sigma):
    """distance_measures.py:117-131."""
    return np.sqrt(2 * (1 - np.sqrt(fidelity(rho, sigma))))


def bures_angle(rho, sigma):
    """distance_measures.py:134-150."""
    return np.arccos(np.sqrt(fidelity(rho, sigma)))


def quantum_chernoff_bound(rho, sigma, tol=1000):
    """distance_measures.py:153-195."""
    def f(s):
        s = np.real_if_close(s)
        return np.trace(np.matmul(fractional_matrix_power(rho, s),
                                  fractional_matrix_power(sigma, 1 - s)))
    f_min = minimize_scalar(f, bounds=(0, 1), method='bounded')
    return np.real_if_close(f_min.fun, tol), np.real_if_close(f_min.x, tol)


def hilbert_schmidt_ip(a, b, tol=1000):
    """distance_measures.py:198-216."""
    return _item(np.trace(np.matmul(np.transpose(np.conj(a)), b)), tol)


def smith_fidelity(rho, sigma, power):
    """distance_measures.py:219-240."""
    if power < 0:
        raise ValueError("Power must be positive")
    if power >= 2:
        raise ValueError("Power must be less than 2")
    return np.sqrt(fidelity(rho, sigma)) ** power


def total_variation_distance(p, q):
    """distance_measures.py:243-265."""
    rowsp, colsp = p.shape
    rowsq, colsq = q.shape
    if not (colsp == colsq == 1 and rowsp > 1 and rowsq > 1):
        raise ValueError("Arrays must be the same length")
    return 0.5 * np.sum(np.abs(p - q))


def entanglement_fidelity(pl0, pl1, tol=1000):
    """distance_measures.py:271-312."""
    assert pl0.shape == pl1.shape
    assert pl0.shape[0] == pl1.shape[1]
    dim = int(np.sqrt(pl0.shape[0]))
    fe = np.trace(np.matmul(np.transpose(np.conj(pl0)), pl1)) / (dim ** 2)
    return _item(fe, tol)


def process_fidelity(pl0, pl1):
    """distance_measures.py:315-359."""
    assert pl0.shape == pl1.shape
    assert pl0.shape[0] == pl1.shape[1]
    dim = int(np.sqrt(pl0.shape[0]))
    return (dim * entanglement_fidelity(pl0, pl1) + 1) / (dim + 1)


def process_infidelity(pl0, pl1):
    """distance_measures.py:362-375."""
    return 1 - process_fidelity(pl0, pl1)


def _is_square(n):
    return n == np.round(np.sqrt(n)) ** 2


def watrous_bounds(choi):
    """distance_measures.py:440-462."""
    if len(choi.shape) != 2:
        raise ValueError("Watrous bounds only defined for matrices")
    if not (_is_square(choi.shape[0]) and _is_square(choi.shape[1])):
        raise ValueError("Choi matrix must have dimensions that are perfect squares")
    _, s, _ = np.linalg.svd(choi)
    nuclear_norm = np.sum(s)
    return nuclear_norm, choi.shape[0] * nuclear_norm
