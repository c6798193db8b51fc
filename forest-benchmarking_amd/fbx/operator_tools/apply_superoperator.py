"""Channel application on MI355X (mirror of operator_tools/apply_superoperator.py)."""
import numpy as np

from .. import _lib

__all__ = ["apply_choi_matrix_2_state", "apply_choi_matrix_2_state_batch",
           "apply_kraus_ops_2_state"]


def apply_choi_matrix_2_state_batch(choi, state) -> np.ndarray:
    c = _lib.c128(choi)
    c = c.reshape((-1,) + c.shape[-2:])
    s = _lib.c128(state)
    s = s.reshape((-1,) + s.shape[-2:])
    d = s.shape[-1]
    n = int(round(np.log2(d)))
    if c.shape[-1] != d * d or 2 ** n != d or c.shape[0] != s.shape[0]:
        raise ValueError("Dimensions of state and Choi matrix are incompatible")
    out = np.empty_like(s)
    _lib.check(_lib.lib().fbx_apply_choi(n, c.shape[0], _lib.dptr(c.view(np.float64)),
                                         _lib.dptr(s.view(np.float64)),
                                         _lib.dptr(out.view(np.float64))))
    return out


def apply_choi_matrix_2_state(choi: np.ndarray, state: np.ndarray) -> np.ndarray:
    """apply_superoperator.py:60-90: Tr_in[(rho^T (x) I) Choi]."""
    return apply_choi_matrix_2_state_batch(np.asarray(choi)[None], np.asarray(state)[None])[0]


def apply_kraus_ops_2_state(kraus_ops, state: np.ndarray) -> np.ndarray:
    """apply_superoperator.py:33-57 for square Kraus operators: the Kraus set is converted to
    its Choi matrix on the device and applied there.  Like the reference (real-typed
    accumulator, :53) the result must be real: a complex result raises."""
    from .superoperator_transformations import kraus2choi, _kraus_stack
    ks = _kraus_stack(kraus_ops)[0]
    dim, _ = state.shape
    rows, cols = ks[0].shape
    if dim != cols:
        raise ValueError("Dimensions of state and Kraus operator are incompatible")
    if rows != cols:
        raise ValueError("only square Kraus operators are supported on the device path")
    out = apply_choi_matrix_2_state(kraus2choi(list(ks)), np.asarray(state, dtype=np.complex128))
    if np.abs(out.imag).max() > 0:
        raise TypeError("Cannot cast complex result to the reference's real accumulator")
    return np.ascontiguousarray(out.real)
