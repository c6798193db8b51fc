"""State / process distance measures on MI355X, with the reference's names and signatures.

Mirror of forest/benchmarking/distance_measures.py.  Scalars are returned as python floats
like the reference (``np.real_if_close(...).item()``).  ``*_batch`` variants take stacked
matrices ``[B, d, d]``.  ``diamond_norm_distance`` (a cvxpy SDP in the reference,
distance_measures.py:378-437) and ``quantum_chernoff_bound`` (a scalar minimisation over
fractional matrix powers, :153-195) are outside the accelerated path and not provided.
"""
import numpy as np

from . import _lib


def _nq(dim):
    n = int(round(np.log2(dim)))
    if 2 ** n != dim:
        raise ValueError("matrix dimension must be a power of two")
    return n


def state_measures_batch(rho, sigma=None, which=("purity", "fidelity", "trace_distance", "hs_ip")):
    """Any subset of {purity(rho), fidelity, trace_distance, hilbert_schmidt_ip}(rho, sigma)."""
    rho = _lib.c128(rho)
    rho = rho.reshape((-1,) + rho.shape[-2:])
    B, d = rho.shape[0], rho.shape[-1]
    sig = rho if sigma is None else _lib.c128(sigma).reshape(rho.shape)
    outs = {k: np.empty(B) for k in which}
    _lib.check(_lib.lib().fbx_state_measures(
        _nq(d), B, _lib.dptr(rho.view(np.float64)), _lib.dptr(sig.view(np.float64)),
        _lib.dptr(outs.get("purity")), _lib.dptr(outs.get("fidelity")),
        _lib.dptr(outs.get("trace_distance")), _lib.dptr(outs.get("hs_ip"))))
    return outs


def purity(rho: np.ndarray, dim_renorm=False, tol: float = 1000) -> float:
    """distance_measures.py:14-36."""
    p = state_measures_batch(rho, None, ("purity",))["purity"][0]
    if dim_renorm:
        dim = rho.shape[0]
        p = (dim / (dim - 1.0)) * (p - 1.0 / dim)
    return float(p)


def impurity(rho: np.ndarray, dim_renorm=False, tol: float = 1000) -> float:
    """distance_measures.py:39-61."""
    imp = 1 - state_measures_batch(rho, None, ("purity",))["purity"][0]
    if dim_renorm:
        dim = rho.shape[0]
        imp = (dim / (dim - 1.0)) * imp
    return float(imp)


def fidelity(rho: np.ndarray, sigma: np.ndarray, tol: float = 1000) -> float:
    """distance_measures.py:64-84."""
    return float(state_measures_batch(rho, sigma, ("fidelity",))["fidelity"][0])


def infidelity(rho: np.ndarray, sigma: np.ndarray, tol: float = 1000) -> float:
    """distance_measures.py:87-97."""
    return 1 - fidelity(rho, sigma, tol)


def trace_distance(rho: np.ndarray, sigma: np.ndarray) -> float:
    """distance_measures.py:100-114 -- half the induced 1-norm, like the reference."""
    return float(state_measures_batch(rho, sigma, ("trace_distance",))["trace_distance"][0])


def bures_distance(rho: np.ndarray, sigma: np.ndarray) -> float:
    """distance_measures.py:117-131."""
    return float(np.sqrt(2 * (1 - np.sqrt(fidelity(rho, sigma)))))


def bures_angle(rho: np.ndarray, sigma: np.ndarray) -> float:
    """distance_measures.py:134-150."""
    return float(np.arccos(np.sqrt(fidelity(rho, sigma))))


def hilbert_schmidt_ip(A: np.ndarray, B: np.ndarray, tol: float = 1000) -> float:
    """distance_measures.py:198-216 (real part; the reference returns a real for Hermitian input)."""
    return float(state_measures_batch(A, B, ("hs_ip",))["hs_ip"][0])


def smith_fidelity(rho: np.ndarray, sigma: np.ndarray, power) -> float:
    """distance_measures.py:219-240."""
    if power < 0:
        raise ValueError("Power must be positive")
    if power >= 2:
        raise ValueError("Power must be less than 2")
    return float(np.sqrt(fidelity(rho, sigma)) ** power)


def total_variation_distance(P: np.ndarray, Q: np.ndarray) -> float:
    """distance_measures.py:243-265 (host reduction over two probability vectors)."""
    rowsp, colsp = P.shape
    rowsq, colsq = Q.shape
    if not (colsp == colsq == 1 and rowsp > 1 and rowsq > 1):
        raise ValueError("Arrays must be the same length")
    return 0.5 * np.sum(np.abs(P - Q))


def process_fidelity_batch(pauli_lio0, pauli_lio1, entanglement=False):
    a = _lib.c128(pauli_lio0)
    b = _lib.c128(pauli_lio1)
    a = a.reshape((-1,) + a.shape[-2:])
    b = b.reshape((-1,) + b.shape[-2:])
    if a.shape[0] == 1 and b.shape[0] > 1:
        a = np.ascontiguousarray(np.broadcast_to(a, b.shape))
    assert a.shape == b.shape
    assert a.shape[-2] == a.shape[-1]
    B, D = a.shape[0], a.shape[-1]
    n = _nq(int(round(np.sqrt(D))))
    fe = np.empty(B)
    fp = np.empty(B)
    _lib.check(_lib.lib().fbx_process_fidelity(n, B, _lib.dptr(a.view(np.float64)),
                                               _lib.dptr(b.view(np.float64)), _lib.dptr(fe),
                                               _lib.dptr(fp)))
    return fe if entanglement else fp


def entanglement_fidelity(pauli_lio0: np.ndarray, pauli_lio1: np.ndarray, tol: float = 1000) -> float:
    """distance_measures.py:271-312."""
    assert pauli_lio0.shape == pauli_lio1.shape
    assert pauli_lio0.shape[0] == pauli_lio1.shape[1]
    return float(process_fidelity_batch(pauli_lio0, pauli_lio1, entanglement=True)[0])


def process_fidelity(pauli_lio0: np.ndarray, pauli_lio1: np.ndarray) -> float:
    """distance_measures.py:315-359."""
    assert pauli_lio0.shape == pauli_lio1.shape
    assert pauli_lio0.shape[0] == pauli_lio1.shape[1]
    return float(process_fidelity_batch(pauli_lio0, pauli_lio1)[0])


def process_infidelity(pauli_lio0: np.ndarray, pauli_lio1: np.ndarray) -> float:
    """distance_measures.py:362-375."""
    return 1 - process_fidelity(pauli_lio0, pauli_lio1)
