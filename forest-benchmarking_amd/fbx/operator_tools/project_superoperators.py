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
    if d * d != D or x.shape[-2] != D:
        raise ValueError("Choi matrices must be d^2 x d^2")
    n = int(round(np.log2(d)))
    if 2 ** n != d or n > 3:
        return _proj_general(kind, x, d, return_iters)
    out = np.empty_like(x)
    iters = np.zeros(B, dtype=np.int32)
    _lib.check(_lib.lib().fbx_proj_choi(kind, n, B, _lib.dptr(x.view(np.float64)),
                                        _lib.dptr(out.view(np.float64)), _lib.iptr(iters)))
    return (out, iters) if return_iters else out


# ---- any other dimension (a qutrit's 9 x 9, the 256 x 256 of four qubits; up to 1024): the reference's algorithms
# step by step, every eigendecomposition, matrix product and partial trace on the device (fbx_eigh -- HBM-resident
# above 64 --, fbx_matmul, fbx_partial_trace); the Dykstra bookkeeping (differences, the scalar stopping rule) and
# the Kronecker placement of a d x d correction are index work on the host.  The fused kernels stop at three qubits.
def _cp_general(x):
    herm = (x + x.conj().transpose(0, 2, 1)) / 2
    w, v = _lib.eigh_batch(herm)
    return _lib.matmul_batch(v, v, conj_t_b=True, scale=np.maximum(w, 0))


def _tp_general(x, d, non_increasing=False):
    from .calculational import partial_trace_bipartite_batch
    pt = partial_trace_bipartite_batch(x, d, d, 0)
    if non_increasing:                                          # project_superoperators.py:37-59
        w, v = _lib.eigh_batch((pt + pt.conj().transpose(0, 2, 1)) / 2)
        target = _lib.matmul_batch(v, v, conj_t_b=True, scale=np.minimum(w, 1))
    else:                                                       # :62-84
        target = np.broadcast_to(np.eye(d), pt.shape)
    return x - np.einsum("bij,kl->bikjl", (pt - target) / d, np.eye(d)).reshape(x.shape)


def _proj_general(kind, x, d, return_iters):
    if x.shape[-1] > 1024:
        raise _lib.FbxError(_lib.FBX_ERR_UNSUPPORTED, "Choi projections: dimensions above 1024 are outside this build")
    iters = np.zeros(x.shape[0], dtype=np.int32)
    if kind == _lib.PROJ_CP:
        out = _cp_general(x)
    elif kind in (_lib.PROJ_TP, _lib.PROJ_TNI):
        out = _tp_general(x, d, kind == _lib.PROJ_TNI)
    else:
        # Dykstra, :87-144, the WHOLE batch in lockstep: every iteration is one batched eigendecomposition + product (CP) and one
        # batched partial trace (TP / TNI) over the items that are still running -- 3 device calls per iteration instead of 3 per
        # iteration and item.  An item's numbers do not depend on its neighbours (one workgroup per matrix), and its scalar
        # stopping rule is evaluated with the very expressions of the one-at-a-time form: same result, bit for bit.
        out = np.empty_like(x)
        tni = kind == _lib.PROJ_PHYSICAL_TNI
        B = x.shape[0]
        old_cp = np.zeros_like(x); old_tp = np.zeros_like(x); last_cp = np.zeros_like(x)
        last_state = x.copy()
        active = np.arange(B)
        while active.size:
            iters[active] += 1
            pre_cp = last_state[active] - old_cp[active]
            cp = _cp_general(pre_cp)
            new_cp = cp - pre_cp
            pre_tp = cp - old_tp[active]
            new_state = _tp_general(pre_tp, d, tni)
            new_tp = new_state - pre_tp
            running = np.zeros(active.size, dtype=bool)
            for k, b in enumerate(active):
                crit = (np.linalg.norm(new_cp[k] - old_cp[b]) ** 2 + np.linalg.norm(new_tp[k] - old_tp[b]) ** 2
                        + 2 * abs(np.vdot(old_tp[b], new_state[k] - last_state[b])) + 2 * abs(np.vdot(old_cp[b], cp[k] - last_cp[b])))
                running[k] = crit >= 1e-4                        # (a NaN ends the item, as `not crit >= 1e-4` does)
            done = active[~running]
            out[done] = new_state[~running]
            keep = active[running]
            old_cp[keep], old_tp[keep], last_cp[keep], last_state[keep] = new_cp[running], new_tp[running], cp[running], new_state[running]
            active = keep
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
