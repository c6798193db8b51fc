"""Physicality checks of channels (mirror of operator_tools/validate_superoperator.py:40-157).

Partial traces, channel application and eigenvalues come from the device library; the final
comparison is the reference's ``np.allclose`` predicate."""
from typing import Sequence

import numpy as np

from .apply_superoperator import apply_choi_matrix_2_state
from .calculational import partial_trace_bipartite_batch
from .project_superoperators import proj_choi_to_trace_preserving
from .superoperator_transformations import _kraus_stack, choi2kraus
from .validate_operator import is_hermitian_matrix, is_identity_matrix, is_positive_semidefinite_matrix

__all__ = ["kraus_operators_are_valid", "choi_is_hermitian_preserving", "choi_is_trace_preserving",
           "choi_is_completely_positive", "choi_is_cptp", "choi_is_unital", "choi_is_unitary"]


def kraus_operators_are_valid(kraus_ops: Sequence[np.ndarray], rtol: float = 1e-05, atol: float = 1e-08) -> bool:
    """validate_superoperator.py:40-62."""
    ks = _kraus_stack(kraus_ops)[0]
    povm = [np.transpose(op).conjugate().dot(op) for op in ks]
    all_psd = all(is_positive_semidefinite_matrix(e) for e in povm)
    return all_psd and is_identity_matrix(sum(povm), rtol, atol)


def choi_is_hermitian_preserving(choi, rtol: float = 1e-05, atol: float = 1e-08) -> bool:
    """validate_superoperator.py:65-77."""
    return is_hermitian_matrix(choi, rtol, atol)


def choi_is_trace_preserving(choi, rtol: float = 1e-05, atol: float = 1e-08) -> bool:
    """validate_superoperator.py:80-97: Tr_out(choi) == I.  choi - proj_TP(choi) = kron((pt - I)/d, I)
    on the device, so pt is read off its corner blocks."""
    choi = np.asarray(choi, dtype=np.complex128)
    dim = int(np.sqrt(choi.shape[0]))
    if dim not in (2, 4, 8):                                    # beyond the 1-3 qubit projection kernels: Tr_out directly
        return is_identity_matrix(partial_trace_bipartite_batch(choi[None], dim, dim, 0)[0], rtol, atol)
    diff = choi - proj_choi_to_trace_preserving(choi)          # kron((pt - I)/dim, I_dim)
    pt = diff[::dim, ::dim] * dim + np.eye(dim)
    return is_identity_matrix(pt, rtol, atol)


def choi_is_completely_positive(choi, rtol: float = 1e-05, atol: float = 1e-08) -> bool:
    """validate_superoperator.py:100-112."""
    return is_positive_semidefinite_matrix(choi, rtol, atol)


def choi_is_cptp(choi, rtol: float = 1e-05, atol: float = 1e-08) -> bool:
    """validate_superoperator.py:115-127."""
    tp = choi_is_trace_preserving(choi, rtol, atol)
    cp = choi_is_completely_positive(choi, rtol, atol)
    return cp and tp


def choi_is_unital(choi, rtol: float = 1e-05, atol: float = 1e-08) -> bool:
    """validate_superoperator.py:130-145."""
    dim = int(np.sqrt(np.asarray(choi).shape[0]))
    if dim not in (2, 4, 8):                                    # the channel applied to the identity = Tr_in of the Choi matrix
        out = partial_trace_bipartite_batch(np.asarray(choi, dtype=np.complex128)[None], dim, dim, 1)[0]
        return is_identity_matrix(out, rtol, atol)
    out = apply_choi_matrix_2_state(np.asarray(choi, dtype=np.complex128), np.identity(dim, dtype=np.complex128))
    return is_identity_matrix(out, rtol, atol)


def choi_is_unitary(choi, limit: float = 1e-09) -> bool:
    """validate_superoperator.py:148-157."""
    return len(choi2kraus(choi, tol=limit)) == 1
