"""Choi-matrix projections on MI355X (mirror of operator_tools/project_superoperators.py)."""
import numpy as np

from .. import _lib

__all__ = ["proj_choi_to_completely_positive", "proj_choi_to_trace_non_increasing",
           "proj_choi_to_trace_preserving", "proj_choi_to_physical", "proj_choi_batch",
           "proj_choi_to_unitary"]


def proj_choi_batch(kind: int, choi, return_iters=False):
    x = _lib.c128(choi)
    x = x.reshape((-1,) + x.shape[-2:])
    B, D = x.shape[0], x.shape[-1]
    d = int(round(np.sqrt(D)))
    n = int(round(np.log2(d)))
    if 4 ** n != D or x.shape[-2] != D:
        raise ValueError("Choi matrices must be 4^n x 4^n")
    out = np.empty_like(x)
    iters = np.zeros(B, dtype=np.int32)
    _lib.check(_lib.lib().fbx_proj_choi(kind, n, B, _lib.dptr(x.view(np.float64)),
                                        _lib.dptr(out.view(np.float64)), _lib.iptr(iters)))
    return (out, iters) if return_iters else out


def proj_choi_to_completely_positive(choi: np.ndarray, check_finite: bool = True) -> np.ndarray:
    """project_superoperators.py:19-34."""
    return proj_choi_batch(_lib.PROJ_CP, np.asarray(choi)[None])[0]


def proj_choi_to_trace_non_increasing(choi: np.ndarray) -> np.ndarray:
    """project_superoperators.py:37-59."""
    return proj_choi_batch(_lib.PROJ_TNI, np.asarray(choi)[None])[0]


def proj_choi_to_trace_preserving(choi: np.ndarray) -> np.ndarray:
    """project_superoperators.py:62-84."""
    return proj_choi_batch(_lib.PROJ_TP, np.asarray(choi)[None])[0]


def proj_choi_to_physical(choi: np.ndarray, make_trace_preserving: bool = True) -> np.ndarray:
    """project_superoperators.py:87-144 (Dykstra, Birgin-Raydan stopping rule)."""
    kind = _lib.PROJ_PHYSICAL_TP if make_trace_preserving else _lib.PROJ_PHYSICAL_TNI
    return proj_choi_batch(kind, np.asarray(choi)[None])[0]


def proj_choi_to_unitary(choi: np.ndarray, check_finite: bool = True) -> np.ndarray:
    """project_superoperators.py:147-175: the unitary channel closest to a process.

    Both decompositions run on the device (``fbx_eigh``): the top eigenvector of the Hermitised
    Choi matrix gives the dominant Kraus operator K, and its polar factor U V^H (the reference takes
    it from an SVD) is K (K^H K)^{-1/2} through the d x d eigendecomposition of K^H K.  The d x d
    products in between are host glue, as is the final global-phase convention."""
    from .superoperator_transformations import kraus2choi, unvec
    choi = np.asarray(choi, dtype=np.complex128)
    dim = int(np.sqrt(choi.shape[0]))
    herm = (choi + choi.conj().T) / 2
    vals, vs = _lib.eigh_batch(herm[None])
    kraus = unvec(vs[0][:, np.argmax(vals[0])].reshape((dim * dim, 1)))
    mu, w = _lib.eigh_batch((kraus.conj().T @ kraus)[None])
    inv_sqrt = (w[0] / np.sqrt(mu[0])) @ w[0].conj().T
    unitary = kraus @ inv_sqrt
    phase = np.angle(unitary[0, 0])
    return kraus2choi(np.exp(-1j * phase) * unitary)
