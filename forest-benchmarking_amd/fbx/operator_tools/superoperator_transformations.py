"""Superoperator representation changes on MI355X.

Mirror of operator_tools/superoperator_transformations.py:33-438.  ``vec`` / ``unvec`` are
host-side reshapes exactly as in the reference; every conversion runs in libfbx
(``fbx_convert``).  ``*_batch`` helpers accept stacked inputs ``[B, ...]``.  Conversions *to*
Kraus operators are eigenvector-valued (defined only up to phase / degeneracy,
superoperator_transformations.py:325-336): ``fbx_choi2kraus`` runs the eigendecomposition and assembles
the operators on the device with a fixed phase convention (``choi2kraus_batch``; ``choi2kraus`` is its
B = 1 case and keeps a host assembly around ``fbx_eigh`` for dimensions that are not qubit systems).
"""
from typing import Optional, Tuple

import numpy as np

from .. import _lib

_REPS = {"kraus": _lib.REP_KRAUS, "choi": _lib.REP_CHOI, "superop": _lib.REP_SUPEROP,
         "pauli_liouville": _lib.REP_PAULI_LIOUVILLE, "chi": _lib.REP_CHI}

__all__ = ["vec", "unvec", "convert_batch", "kraus2chi", "kraus2superop", "kraus2pauli_liouville",
           "kraus2choi", "chi2pauli_liouville", "chi2superop", "chi2choi", "superop2chi",
           "superop2pauli_liouville", "superop2choi", "pauli_liouville2chi",
           "pauli_liouville2superop", "pauli_liouville2choi", "choi2chi", "choi2superop",
           "choi2pauli_liouville", "pauli2computational_basis_matrix",
           "computational2pauli_basis_matrix", "choi2kraus", "superop2kraus", "pauli_liouville2kraus",
           "chi2kraus", "choi2kraus_batch", "superop2kraus_batch", "pauli_liouville2kraus_batch", "chi2kraus_batch"]


def vec(matrix: np.ndarray) -> np.ndarray:
    """Column-stacking vectorisation (superoperator_transformations.py:33-51)."""
    return np.asarray(matrix).T.reshape((-1, 1))


def unvec(vector: np.ndarray, shape: Optional[Tuple[int, int]] = None) -> np.ndarray:
    """Inverse of vec (superoperator_transformations.py:54-79)."""
    vector = np.asarray(vector)
    if shape is None:
        dim = int(np.sqrt(vector.size))
        shape = dim, dim
    return vector.reshape(*shape).T


_BASIS_FREE = {("kraus", "superop"), ("kraus", "choi"), ("superop", "choi"), ("choi", "superop")}


def _nq_from_D(D):
    """Number of qubits of a D x D superoperator, or None when D is not 4^n with n <= 5."""
    d = int(round(np.sqrt(D)))
    n = int(round(np.log2(d))) if d > 0 else 0
    return n if 4 ** n == D and 1 <= n <= 5 else None


def convert_batch(src: str, dst: str, x) -> np.ndarray:
    """Stacked conversion: x is [B, K, d, d] for src == 'kraus', else [B, D, D]; returns [B, D, D].
    Qubit systems (d = 2^n, n <= 5) take the Pauli-aware kernels (``fbx_convert``); any other dimension
    is served for the basis-free pairs (kraus -> superop / choi, superop <-> choi; ``fbx_convert_general``)."""
    x = _lib.c128(x)
    if src == "kraus":
        if x.ndim != 4 or x.shape[-1] != x.shape[-2]:
            raise ValueError("kraus input must be [B, K, d, d] with square operators")
        B, K, d = x.shape[0], x.shape[1], x.shape[-1]
        n = _nq_from_D(d * d)
    else:
        if x.ndim != 3 or x.shape[-1] != x.shape[-2]:
            raise ValueError("input must be [B, D, D]")
        B, K = x.shape[0], 0
        d = int(round(np.sqrt(x.shape[-1])))
        if d * d != x.shape[-1]:
            raise ValueError("superoperator dimension must be a perfect square")
        n = _nq_from_D(x.shape[-1])
    D = d * d
    out = np.empty((B, D, D), dtype=np.complex128)
    if n is None:
        if (src, dst) not in _BASIS_FREE:
            raise ValueError(f"{src} -> {dst} needs the Pauli basis of a qubit system (dimension 2^n, n <= 5); "
                             f"got dimension {d}")
        _lib.check(_lib.lib().fbx_convert_general(_REPS[src], _REPS[dst], d, B, _lib.dptr(x.view(np.float64)),
                                                  K, _lib.dptr(out.view(np.float64))))
        return out
    _lib.check(_lib.lib().fbx_convert(_REPS[src], _REPS[dst], n, B, _lib.dptr(x.view(np.float64)),
                                      K, _lib.dptr(out.view(np.float64))))
    return out


def _kraus_stack(kraus_ops):
    """single-ndarray-as-one-operator convenience (superoperator_transformations.py:90-92)."""
    if isinstance(kraus_ops, np.ndarray):
        if len(kraus_ops[0].shape) < 2:
            kraus_ops = [kraus_ops]
    return np.stack([np.asarray(k, dtype=np.complex128) for k in kraus_ops])[None]


def _one(src, dst, x):
    return convert_batch(src, dst, np.asarray(x)[None])[0]


def kraus2chi(kraus_ops):
    """superoperator_transformations.py:82-97."""
    return convert_batch("kraus", "chi", _kraus_stack(kraus_ops))[0]


def kraus2superop(kraus_ops):
    """superoperator_transformations.py:100-145.  Non-square Kraus operators (M x N, measurement
    theory / error correction; the reference returns an M^2 x N^2 matrix) are zero-padded to the
    enclosing 2^n x 2^n square, converted on the device, and the rows / columns that belong to the
    padding are dropped -- sum conj(K) (x) K has no contribution from zero entries."""
    ks = _kraus_stack(kraus_ops)
    rows, cols = ks.shape[-2:]
    if rows == cols:
        return convert_batch("kraus", "superop", ks)[0]
    n = 2
    while n < max(rows, cols):
        n *= 2
    padded = np.zeros(ks.shape[:-2] + (n, n), dtype=np.complex128)
    padded[..., :rows, :cols] = ks
    full = convert_batch("kraus", "superop", padded)[0].reshape(n, n, n, n)
    return np.ascontiguousarray(full[:rows, :rows, :cols, :cols]).reshape(rows * rows, cols * cols)


def kraus2pauli_liouville(kraus_ops):
    """superoperator_transformations.py:148-156."""
    return convert_batch("kraus", "pauli_liouville", _kraus_stack(kraus_ops))[0]


def kraus2choi(kraus_ops):
    """superoperator_transformations.py:159-182."""
    return convert_batch("kraus", "choi", _kraus_stack(kraus_ops))[0]


def chi2pauli_liouville(chi_matrix):
    """superoperator_transformations.py:185-192."""
    return _one("chi", "pauli_liouville", chi_matrix)


def chi2superop(chi_matrix):
    """superoperator_transformations.py:207-214."""
    return _one("chi", "superop", chi_matrix)


def chi2choi(chi_matrix):
    """superoperator_transformations.py:217-226."""
    return _one("chi", "choi", chi_matrix)


def superop2chi(superop):
    """superoperator_transformations.py:241-250."""
    return _one("superop", "chi", superop)


def superop2pauli_liouville(superop):
    """superoperator_transformations.py:253-264."""
    return _one("superop", "pauli_liouville", superop)


def superop2choi(superop):
    """superoperator_transformations.py:267-277."""
    return _one("superop", "choi", superop)


def pauli_liouville2chi(pl_matrix):
    """superoperator_transformations.py:291-298."""
    return _one("pauli_liouville", "chi", pl_matrix)


def pauli_liouville2superop(pl_matrix):
    """superoperator_transformations.py:301-312."""
    return _one("pauli_liouville", "superop", pl_matrix)


def pauli_liouville2choi(pl_matrix):
    """superoperator_transformations.py:315-322."""
    return _one("pauli_liouville", "choi", pl_matrix)


def choi2kraus_batch(choi, tol: float = 1e-9):
    """superoperator_transformations.py:325-336 for a stack of n-qubit Choi matrices [B, D, D] (n <= 5), on the device
    (``fbx_choi2kraus``: eigendecomposition + assembly).  Returns ``(kraus [B, D, d, d], counts [B])``: the first
    ``counts[b]`` operators of item b are sqrt(lambda_i) unvec(v_i) for the eigenpairs with |lambda_i| > tol in ascending
    eigenvalue order -- the reference's list --, the other slots are zero.  The phase of each eigenvector is fixed so that its
    first non-negligible component is real and positive."""
    choi = _lib.c128(choi)
    if choi.ndim != 3 or choi.shape[-1] != choi.shape[-2]:
        raise ValueError("choi input must be [B, D, D]")
    B, D = choi.shape[0], choi.shape[-1]
    n = _nq_from_D(D)
    if n is None:
        raise ValueError("choi2kraus_batch serves qubit systems (D = 4^n, n <= 5); use choi2kraus for other dimensions")
    d = 2 ** n
    kraus = np.empty((B, D, d, d), dtype=np.complex128)
    counts = np.zeros(B, dtype=np.int32)
    _lib.check(_lib.lib().fbx_choi2kraus(n, B, _lib.dptr(choi.view(np.float64)), float(tol),
                                         _lib.dptr(kraus.view(np.float64)), _lib.iptr(counts)))
    return kraus, counts


def superop2kraus_batch(superop, tol: float = 1e-9):
    """superoperator_transformations.py:229-238 for a stack [B, D, D]; see choi2kraus_batch."""
    return choi2kraus_batch(convert_batch("superop", "choi", superop), tol)


def pauli_liouville2kraus_batch(pl_matrix, tol: float = 1e-9):
    """superoperator_transformations.py:280-288 for a stack [B, D, D]; see choi2kraus_batch."""
    return choi2kraus_batch(convert_batch("pauli_liouville", "choi", pl_matrix), tol)


def chi2kraus_batch(chi_matrix, tol: float = 1e-9):
    """superoperator_transformations.py:195-204 for a stack [B, D, D]; see choi2kraus_batch."""
    return choi2kraus_batch(convert_batch("chi", "choi", chi_matrix), tol)


def choi2kraus(choi, tol: float = 1e-9):
    """superoperator_transformations.py:325-336: one Kraus operator sqrt(lambda_i) unvec(v_i) per
    eigenpair of the Choi matrix with |lambda_i| > tol.  The operators are defined up to the phase of each
    eigenvector; the phase is fixed so that the first non-zero component of every eigenvector is real and
    positive, which is what LAPACK hands the reference for the operators its tests compare entry by entry
    (tests/test_superoperator_transformations.py:215-216, IZKraus; probed on Haar unitaries).  Qubit systems run
    ``fbx_choi2kraus`` (the B = 1 case of choi2kraus_batch); any other dimension (a qutrit's 9 x 9 Choi matrix)
    takes its eigendecomposition from ``fbx_eigh`` and assembles the list here with the same convention."""
    choi = np.asarray(choi, dtype=np.complex128)
    if choi.ndim == 2 and _nq_from_D(choi.shape[0]) is not None and choi.shape[0] == choi.shape[1]:
        kraus, counts = choi2kraus_batch(choi[None], tol)
        return [np.array(kraus[0, i]) for i in range(int(counts[0]))]
    w, v = _lib.eigh_batch(choi[None])
    ops = []
    for ev, evec in zip(w[0], v[0].T):
        if not abs(ev) > tol:
            continue
        big = np.flatnonzero(np.abs(evec) > 1e-12 * np.linalg.norm(evec))
        if big.size:
            evec = evec * (abs(evec[big[0]]) / evec[big[0]])
        ops.append(np.lib.scimath.sqrt(ev) * unvec(np.array([evec]).T))
    return ops


def superop2kraus(superop):
    """superoperator_transformations.py:229-238."""
    return choi2kraus(superop2choi(superop))


def pauli_liouville2kraus(pl_matrix):
    """superoperator_transformations.py:280-288."""
    return choi2kraus(pauli_liouville2choi(pl_matrix))


def chi2kraus(chi_matrix):
    """superoperator_transformations.py:195-204."""
    return pauli_liouville2kraus(chi2pauli_liouville(chi_matrix))


def choi2chi(choi):
    """superoperator_transformations.py:339-348 (through the eigendecomposition: |C| for
    non-CP input, eigenvalues with |lambda| <= 1e-9 dropped -- reproduced on the device)."""
    return _one("choi", "chi", choi)


def choi2superop(choi):
    """superoperator_transformations.py:351-361."""
    return _one("choi", "superop", choi)


def choi2pauli_liouville(choi):
    """superoperator_transformations.py:364-371."""
    return _one("choi", "pauli_liouville", choi)


def pauli2computational_basis_matrix(dim) -> np.ndarray:
    """superoperator_transformations.py:374-408: columns are vec(P_k).  Read off the device basis
    change: chi2choi(e_k e_0^T) = vec(P_k) vec(I)^H, whose column 0 is vec(P_k)."""
    D = dim ** 2
    eye = np.zeros((D, D, D), dtype=np.complex128)
    for k in range(D):
        eye[k, k, 0] = 1.0          # chi = e_k e_0^T  ->  choi = vec(P_k) vec(P_0)^H
    choi = convert_batch("chi", "choi", eye)
    return np.ascontiguousarray(choi[:, :, 0].T)


def computational2pauli_basis_matrix(dim) -> np.ndarray:
    """superoperator_transformations.py:411-438."""
    return pauli2computational_basis_matrix(dim).conj().T / dim
