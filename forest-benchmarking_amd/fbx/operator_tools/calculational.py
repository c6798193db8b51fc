"""Small matrix helpers with the reference's names (forest/benchmarking/operator_tools/calculational.py).

``sqrtm_psd`` goes through the device eigensolver (``fbx_eigh``); ``partial_trace`` of an operator on two
subsystems runs on the device (``fbx_partial_trace``); for longer subsystem lists it is an index shuffle plus a
trace and stays on the host (the partial trace the estimators need lives inside the kernels,
csrc/fbx_choi.hpp ``partial_trace_out``).
"""
import numpy as np

from .. import _lib

__all__ = ["partial_trace", "partial_trace_bipartite_batch", "outer_product", "inner_product", "sqrtm_psd"]


def partial_trace_bipartite_batch(rho, dim_a: int, dim_b: int, keep: int) -> np.ndarray:
    """Stack [B, dim_a dim_b, dim_a dim_b] of operators on A (x) B -> Tr_B (keep = 0) or Tr_A (keep = 1), on the device
    (``fbx_partial_trace``)."""
    x = _lib.c128(rho)
    if x.ndim != 3 or x.shape[1] != dim_a * dim_b or x.shape[2] != dim_a * dim_b:
        raise ValueError("rho must be [B, dim_a * dim_b, dim_a * dim_b]")
    n = dim_a if keep == 0 else dim_b
    out = np.empty((x.shape[0], n, n), dtype=np.complex128)
    _lib.check(_lib.lib().fbx_partial_trace(int(dim_a), int(dim_b), int(keep), x.shape[0], _lib.dptr(x.view(np.float64)),
                                            _lib.dptr(out.view(np.float64))))
    return out


def partial_trace(rho, keep, dims, optimize=False):
    """calculational.py:5-35: trace out every subsystem whose index is not in ``keep``.

    ``dims`` lists the subsystem dimensions in tensor order; the kept subsystems stay in their
    original relative order."""
    dims = [int(x) for x in np.asarray(dims).ravel()]
    kept = set(int(k) for k in np.asarray(keep).ravel())
    n = len(dims)
    if n == 2 and len(kept) == 1 and kept <= {0, 1} and dims[0] * dims[1] <= 4096 and np.asarray(rho).shape == (dims[0] * dims[1],) * 2:
        return partial_trace_bipartite_batch(np.asarray(rho)[None], dims[0], dims[1], next(iter(kept)))[0]
    t = np.asarray(rho).reshape(dims + dims)
    keep_axes = [i for i in range(n) if i in kept]
    drop_axes = [i for i in range(n) if i not in kept]
    # rows: kept then dropped; columns likewise; then sum the diagonal of the dropped part
    t = np.transpose(t, keep_axes + drop_axes + [n + i for i in keep_axes] + [n + i for i in drop_axes])
    nk = int(np.prod([dims[i] for i in keep_axes])) if keep_axes else 1
    nd = int(np.prod([dims[i] for i in drop_axes])) if drop_axes else 1
    t = t.reshape(nk, nd, nk, nd)
    return np.einsum("ajbj->ab", t, optimize=optimize)


def _check_kets(a, b):
    rows1, cols1 = a.shape
    rows2, cols2 = b.shape
    if not (cols1 == cols2 == 1 and rows1 > 1 and rows2 > 1):
        raise ValueError("The vectors do not have the correct dimensions.")


def outer_product(bra1: np.ndarray, bra2: np.ndarray) -> np.ndarray:
    """calculational.py:38-52: |bra1><bra2| for two (dim, 1) column vectors."""
    _check_kets(bra1, bra2)
    return bra1.reshape(-1, 1) * bra2.conj().reshape(1, -1)


def inner_product(bra1: np.ndarray, bra2: np.ndarray) -> complex:
    """calculational.py:55-72: <bra1|bra2> as a (1, 1) array, like the reference."""
    _check_kets(bra1, bra2)
    return bra1.conj().T @ bra2


def sqrtm_psd_batch(matrices) -> np.ndarray:
    """V sqrt(max(lambda, 0)) V^H for stacked Hermitian matrices [B, N, N], N <= 1024: eigendecomposition and the
    product both on the device (``fbx_eigh``, ``fbx_matmul``)."""
    w, v = _lib.eigh_batch(matrices)
    return _lib.matmul_batch(v, v, conj_t_b=True, scale=np.sqrt(np.maximum(w, 0)))


def sqrtm_psd(matrix: np.ndarray, check_finite: bool = True) -> np.ndarray:
    """calculational.py:77-91."""
    matrix = np.asarray(matrix)
    if check_finite and not np.isfinite(matrix).all():
        raise ValueError("array must not contain infs or NaNs")
    return sqrtm_psd_batch(matrix[None])[0]
