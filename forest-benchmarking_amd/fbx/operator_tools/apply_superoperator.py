"""Channel application on MI355X (mirror of operator_tools/apply_superoperator.py).

The device kernel (``fbx_apply_choi``) works on 2^n-dimensional spaces, n <= 3.  Other dimensions up to 8
(a qutrit; the non-square Kraus operators of apply_superoperator.py:33-57) are embedded: operators are
zero-padded to the next power of two, which changes neither Tr_in[(rho^T (x) 1) Choi] nor
sum_k K rho K^H on the original block.  Dimensions 9..32 (4 and 5 qubits) take the reference's formula on the
generic primitives (``fbx_matmul``, ``fbx_partial_trace``)."""
import numpy as np

from .. import _lib

__all__ = ["apply_choi_matrix_2_state", "apply_choi_matrix_2_state_batch",
           "apply_kraus_ops_2_state"]


def _pow2_at_least(d):
    p = 2
    while p < d:
        p *= 2
    return p


def _embed_choi(c, d, p):
    """Choi matrices on C^d (x) C^d, index i_in * d + i_out, re-indexed into C^p (x) C^p."""
    B = c.shape[0]
    out = np.zeros((B, p, p, p, p), dtype=np.complex128)
    out[:, :d, :d, :d, :d] = c.reshape(B, d, d, d, d)
    return out.reshape(B, p * p, p * p)


def apply_choi_matrix_2_state_batch(choi, state) -> np.ndarray:
    c = _lib.c128(choi)
    c = c.reshape((-1,) + c.shape[-2:])
    s = _lib.c128(state)
    s = s.reshape((-1,) + s.shape[-2:])
    d = s.shape[-1]
    if c.shape[-1] != d * d or c.shape[-2] != d * d or s.shape[-2] != d or c.shape[0] != s.shape[0]:
        raise ValueError("Dimensions of state and Choi matrix are incompatible")
    p = _pow2_at_least(d)
    if p > 8:
        # 4 and 5 qubits (and anything else up to dimension 32): the reference's formula literally,
        # Tr_in[Choi (rho^T (x) 1)] (apply_superoperator.py:87-90), on the generic device primitives
        if d > 32:
            raise _lib.FbxError(_lib.FBX_ERR_UNSUPPORTED, "apply_choi_matrix_2_state: dimensions above 32 (5 qubits) "
                                                          "are outside this build")
        from .calculational import partial_trace_bipartite_batch
        lift = np.einsum("bji,kl->bikjl", s, np.eye(d)).reshape(s.shape[0], d * d, d * d)     # rho^T (x) 1
        return partial_trace_bipartite_batch(_lib.matmul_batch(c, lift), d, d, 1)
    if p != d:
        c = _embed_choi(c, d, p)
        sp = np.zeros((s.shape[0], p, p), dtype=np.complex128)
        sp[:, :d, :d] = s
        s = sp
    out = np.empty_like(s)
    _lib.check(_lib.lib().fbx_apply_choi(p.bit_length() - 1, c.shape[0], _lib.dptr(c.view(np.float64)),
                                         _lib.dptr(s.view(np.float64)),
                                         _lib.dptr(out.view(np.float64))))
    return np.ascontiguousarray(out[:, :d, :d])


def apply_choi_matrix_2_state(choi: np.ndarray, state: np.ndarray) -> np.ndarray:
    """apply_superoperator.py:60-90: Tr_in[(rho^T (x) I) Choi]."""
    return apply_choi_matrix_2_state_batch(np.asarray(choi)[None], np.asarray(state)[None])[0]


def apply_kraus_ops_2_state(kraus_ops, state: np.ndarray) -> np.ndarray:
    """apply_superoperator.py:33-57: sum_k K rho K^H, Kraus operators rows x cols (not necessarily
    square), state cols x cols, result rows x rows.  The set goes to its Choi matrix on the device
    (``fbx_convert``) and is applied there (``fbx_apply_choi``).

    Reference quirk kept: the reference accumulates into ``np.zeros((rows, rows))``,
    a REAL array, so any complex-typed operand makes its ``+=`` raise numpy's casting TypeError; real
    operands give a real result."""
    from .superoperator_transformations import convert_batch, _kraus_stack
    state = np.asarray(state)
    ks = _kraus_stack(kraus_ops)[0]
    if isinstance(kraus_ops, np.ndarray):
        raw = [kraus_ops] if kraus_ops.ndim == 2 else list(kraus_ops)
    else:
        raw = [np.asarray(k) for k in kraus_ops]
    dim, _ = state.shape
    rows, cols = ks[0].shape
    if dim != cols:
        raise ValueError("Dimensions of state and Kraus operator are incompatible")
    if np.iscomplexobj(state) or any(np.iscomplexobj(k) for k in raw):
        raise TypeError("Cannot cast ufunc 'add' output from dtype('complex128') to dtype('float64') with "
                        "casting rule 'same_kind' (the reference accumulates Kraus products into a real array, "
                        "apply_superoperator.py:53-55)")
    p = _pow2_at_least(max(rows, cols))
    if p > 8:
        raise _lib.FbxError(_lib.FBX_ERR_UNSUPPORTED, "apply_kraus_ops_2_state: dimensions above 8 (3 qubits) "
                                                      "are outside this build")
    kp = np.zeros((1, ks.shape[0], p, p), dtype=np.complex128)
    kp[0, :, :rows, :cols] = ks
    sp = np.zeros((1, p, p), dtype=np.complex128)
    sp[0, :cols, :cols] = state
    out = apply_choi_matrix_2_state_batch(convert_batch("kraus", "choi", kp), sp)[0]
    return np.ascontiguousarray(out[:rows, :rows].real)
