"""State / process distance measures on MI355X, with the reference's names and signatures.

Mirror of forest/benchmarking/distance_measures.py.  Scalars are returned as python floats
like the reference (``np.real_if_close(...).item()``).  ``*_batch`` variants take stacked
matrices ``[B, d, d]``.  ``quantum_chernoff_bound`` and ``watrous_bounds`` take their spectra from
the device eigensolver; ``diamond_norm_distance`` is the reference's cvxpy semidefinite program
(distance_measures.py:378-437) and, like there, needs cvxpy to be installed.
"""
from typing import Tuple

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
    if d & (d - 1) or d < 2:
        # not a power of two (e.g. a qutrit): embedded with zero padding, which leaves all four
        # measures unchanged (tr, spectra of products and column sums only gain zeros)
        p = 2
        while p < d:
            p *= 2

        def pad(x):
            y = np.zeros((B, p, p), dtype=np.complex128)
            y[:, :d, :d] = x
            return y
        same = sig is rho
        rho = pad(rho)
        sig = rho if same else pad(sig)
        d = p
    if d > 8:
        return _state_measures_large(rho, sig, which)
    outs = {k: np.empty(B) for k in which}
    _lib.check(_lib.lib().fbx_state_measures(
        _nq(d), B, _lib.dptr(rho.view(np.float64)), _lib.dptr(sig.view(np.float64)),
        _lib.dptr(outs.get("purity")), _lib.dptr(outs.get("fidelity")),
        _lib.dptr(outs.get("trace_distance")), _lib.dptr(outs.get("hs_ip"))))
    return outs


def _state_measures_large(rho, sig, which):
    """Dimensions above 8 (4 and 5 qubits, up to 1024): the reference's formulas step by step on the generic device
    primitives -- ``fbx_matmul`` for the products, ``fbx_eigh`` (HBM-resident above 64) for the matrix square roots
    (distance_measures.py:14-36, 64-84, 100-114, 198-216).  The kernels of ``fbx_state_measures`` keep whole states
    in LDS and stop at three qubits."""
    from .operator_tools.calculational import sqrtm_psd_batch
    if rho.shape[-1] > 1024:
        raise _lib.FbxError(_lib.FBX_ERR_UNSUPPORTED, "state measures: dimensions above 1024 are outside this build")
    outs = {}
    if "purity" in which:
        outs["purity"] = np.trace(_lib.matmul_batch(rho, rho), axis1=1, axis2=2).real
    if "hs_ip" in which:
        outs["hs_ip"] = np.trace(_lib.matmul_batch(rho, sig, conj_t_a=True), axis1=1, axis2=2).real
    if "trace_distance" in which:                                  # half the induced 1-norm, as the reference has it
        outs["trace_distance"] = 0.5 * np.abs(rho - sig).sum(axis=1).max(axis=1)
    if "fidelity" in which:
        root = sqrtm_psd_batch(rho)
        inner = _lib.matmul_batch(_lib.matmul_batch(root, sig), root)
        w = _lib.eigh_batch(inner, eigenvectors=False)              # tr sqrtm_psd(.) = sum of sqrt of the clipped eigenvalues
        outs["fidelity"] = np.sqrt(np.maximum(w, 0)).sum(axis=1) ** 2
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


def quantum_chernoff_bound(rho: np.ndarray, sigma: np.ndarray, tol: float = 1000) -> Tuple[float, float]:
    """distance_measures.py:153-195: min over 0 <= s <= 1 of tr(rho^s sigma^(1-s)) and its argmin.

    Both spectra come from one ``fbx_eigh`` call; with rho = sum a_i |v_i><v_i| and
    sigma = sum b_j |w_j><w_j| the objective is sum_ij a_i^s b_j^(1-s) |<v_i|w_j>|^2, minimised by
    the same bounded scalar search the reference uses (scipy ``minimize_scalar``)."""
    from scipy.optimize import minimize_scalar
    w, v = _lib.eigh_batch(np.stack([_lib.c128(rho), _lib.c128(sigma)]))
    a, b = np.maximum(w[0], 0.0), np.maximum(w[1], 0.0)
    overlap = np.abs(v[0].conj().T @ v[1]) ** 2
    pa, pb = a > 0, b > 0

    def f(s):
        s = float(np.real(s))
        fa = np.where(pa, a, 1.0) ** s * pa
        fb = np.where(pb, b, 1.0) ** (1.0 - s) * pb
        return float(fa @ overlap @ fb)

    res = minimize_scalar(f, bounds=(0, 1), method="bounded")
    return np.real_if_close(res.fun, tol), np.real_if_close(res.x, tol)


def hilbert_schmidt_ip(A: np.ndarray, B: np.ndarray, tol: float = 1000):
    """distance_measures.py:198-216: tr(A^H B) through ``np.real_if_close`` -- a float when the
    imaginary part is negligible (always for Hermitian operands), else the complex value.  The kernel
    reduces Re tr(X^H Y); Im tr(A^H B) = Re tr((iA)^H B) is a second reduction of the same kernel."""
    A, B = np.asarray(A), np.asarray(B)
    re = float(state_measures_batch(A, B, ("hs_ip",))["hs_ip"][0])
    if not (np.iscomplexobj(A) or np.iscomplexobj(B)):
        return re
    im = float(state_measures_batch(1j * A, B, ("hs_ip",))["hs_ip"][0])
    return np.real_if_close(complex(re, im), tol).item()


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


def _is_square(n):
    return n == np.round(np.sqrt(n)) ** 2


def _pow2_at_least(n):
    p = 2
    while p < n:
        p *= 2
    return p


def _singular_values(a: np.ndarray) -> np.ndarray:
    """Singular values through the device Hermitian eigensolver (N <= 64): |eig(a)| for Hermitian
    input, otherwise the non-negative half of the spectrum of the dilation [[0, a], [a^H, 0]]
    (zero-padded to a power of two, which only adds zero singular values)."""
    a = _lib.c128(a)
    r, c = a.shape
    if r == c and np.array_equal(a, a.conj().T) and _pow2_at_least(r) <= 64:
        p = _pow2_at_least(r)
        h = np.zeros((p, p), dtype=np.complex128)
        h[:r, :r] = a
        return np.abs(_lib.eigh_batch(h[None], eigenvectors=False)[0])
    if _pow2_at_least(r + c) <= 64:
        p = _pow2_at_least(r + c)
        h = np.zeros((p, p), dtype=np.complex128)
        h[:r, r:r + c] = a
        h[r:r + c, :r] = a.conj().T
        lam = _lib.eigh_batch(h[None], eigenvectors=False)[0]
        return np.sort(lam)[::-1][:min(r, c)].clip(min=0.0)
    p = _pow2_at_least(c)
    if p > 64:
        raise ValueError("matrices beyond 64 columns are not supported by the device eigensolver")
    g = np.zeros((p, p), dtype=np.complex128)
    g[:c, :c] = a.conj().T @ a
    return np.sqrt(np.maximum(_lib.eigh_batch(g[None], eigenvectors=False)[0], 0.0))


def watrous_bounds(choi: np.ndarray) -> Tuple[float, float]:
    """distance_measures.py:440-460: (nuclear norm, dim * nuclear norm) of a Choi-matrix difference."""
    choi = np.asarray(choi)
    if len(choi.shape) != 2:
        raise ValueError("Watrous bounds only defined for matrices")
    if not (_is_square(choi.shape[0]) and _is_square(choi.shape[1])):
        raise ValueError("Choi matrix must have dimensions that are perfect squares")
    nuclear_norm = float(np.sum(_singular_values(choi)))
    return nuclear_norm, choi.shape[0] * nuclear_norm


def _watrous_sdp_value(delta: np.ndarray, dim: int, restarts: int = 3, seed: int = 0) -> float:
    """Optimum of the SDP of ``diamond_norm_distance`` below without a general-purpose solver.

    For fixed rho the inner problem  max tr(J W), 0 <= W <= R := 1 (x) rho  is solved in closed form: with W = S X S,
    S = 1 (x) T, T Hermitian with T^2 = rho, 0 <= X <= 1, the optimum is the sum of the positive eigenvalues of S J S (X = the
    projector on its positive eigenspace).  What is left is  max over Hermitian T of  g(T) / tr(T^2),
    g(T) = tr[(S J S)_+]  -- homogeneous of degree two, d^2 real unknowns (4 / 16 / 64 for 1 / 2 / 3 qubits), the partial
    maximum of a linear function over a convex set, hence concave in rho.  Gradient (envelope theorem, dS = 1 (x) dT):
    dg = tr(A (1 (x) dT)),  A = X S J + J S X,  i.e.  grad g = Tr_1(A).  Maximised with L-BFGS from a few starts."""
    from scipy.optimize import minimize
    big = dim * dim
    J = (delta + delta.conj().T) / 2

    def unpack(x):
        t = np.zeros((dim, dim), dtype=np.complex128)
        iu = np.triu_indices(dim, 1)
        nd = len(iu[0])
        t[np.diag_indices(dim)] = x[:dim]
        t[iu] = x[dim:dim + nd] + 1j * x[dim + nd:]
        return t + np.triu(t, 1).conj().T

    def pack(g):                                     # gradient wrt the real parameters of a Hermitian matrix
        iu = np.triu_indices(dim, 1)
        return np.concatenate([np.real(np.diag(g)), 2 * np.real(g[iu]), 2 * np.imag(g[iu])])

    def neg_quotient(x):
        t = unpack(x)
        n2 = float(np.real(np.trace(t @ t)))
        s = np.kron(np.eye(dim), t)
        m = s @ J @ s
        lam, v = np.linalg.eigh((m + m.conj().T) / 2)
        pos = lam > 0
        g = float(lam[pos].sum())
        xp = (v[:, pos] * 1.0) @ v[:, pos].conj().T
        a = xp @ s @ J
        a = a + a.conj().T
        grad = np.einsum("iaib->ab", a.reshape(dim, dim, dim, dim))      # Tr over the first factor
        q = g / n2
        return -q, -pack((grad - 2 * q * t) / n2)

    rs = np.random.RandomState(seed)
    best = 0.0
    starts = [np.eye(dim)] + [rs.randn(dim, dim) + 1j * rs.randn(dim, dim) for _ in range(max(restarts - 1, 0))]
    for t0 in starts:
        t0 = (t0 + t0.conj().T) / 2
        t0 = t0 / np.sqrt(np.real(np.trace(t0 @ t0)))
        iu = np.triu_indices(dim, 1)
        x0 = np.concatenate([np.real(np.diag(t0)), np.real(t0[iu]), np.imag(t0[iu])])
        res = minimize(neg_quotient, x0, jac=True, method="L-BFGS-B", options={"maxiter": 500, "ftol": 1e-14, "gtol": 1e-10})
        best = max(best, -float(res.fun))
    return best


def diamond_norm_distance(choi0: np.ndarray, choi1: np.ndarray) -> float:
    """distance_measures.py:378-437: Watrous' simplified SDP for the diamond norm of the difference
    of two CPTP maps.  A convex program, not a kernel: formulated with cvxpy exactly as the
    reference does when cvxpy is importable.

    maximise  Re tr(J^H W)   s.t.  W >= 0,  W <= 1 (x) rho,  rho >= 0,  tr rho = 1
    with J the Hermitian part of choi0 - choi1; the distance is twice the optimum.

    Without cvxpy (this image has none) the same optimum comes from ``_watrous_sdp_value`` -- the SDP reduced to a smooth
    concave maximisation over the d x d density matrix, host numpy / scipy -- pinned to the reference's known answers
    (tests/test_distance_measures.py:186-218, rtol 1e-2) in tests/test_extras_cpu.py."""
    assert choi0.shape == choi1.shape
    assert choi0.shape[0] == choi1.shape[1]
    big = choi0.shape[0]
    dim = int(np.sqrt(big))
    delta = np.asarray(choi0) - np.asarray(choi1)
    delta = (delta + delta.conj().T) / 2
    try:
        import cvxpy as cvx
    except ImportError:
        return 2.0 * _watrous_sdp_value(delta, dim)
    rho = cvx.Variable((dim, dim), hermitian=True)
    w = cvx.Variable((big, big), hermitian=True)
    constraints = [rho >> 0, cvx.trace(rho) == 1, w >> 0, cvx.kron(np.eye(dim), rho) - w >> 0]
    problem = cvx.Problem(cvx.Maximize(cvx.real(cvx.trace(delta.conj().T @ w))), constraints)
    problem.solve()
    return problem.value * 2
