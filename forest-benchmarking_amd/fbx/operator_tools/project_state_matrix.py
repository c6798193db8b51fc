"""project_state_matrix_to_physical on MI355X (operator_tools/project_state_matrix.py:6-52)."""
import numpy as np

from .. import _lib


def project_state_matrix_to_physical_batch(rho) -> np.ndarray:
    x = _lib.c128(rho)
    x = x.reshape((-1,) + x.shape[-2:])
    d = x.shape[-1]
    n = int(round(np.log2(d)))
    if 2 ** n != d or x.shape[-2] != d:
        raise ValueError("state matrices must be 2^n x 2^n")
    out = np.empty_like(x)
    _lib.check(_lib.lib().fbx_proj_state_physical(n, x.shape[0], _lib.dptr(x.view(np.float64)),
                                                  _lib.dptr(out.view(np.float64))))
    return out


def project_state_matrix_to_physical(rho: np.ndarray) -> np.ndarray:
    """Closest (2-norm) trace-one PSD matrix, Smolin-Gambetta-Smith."""
    return project_state_matrix_to_physical_batch(np.asarray(rho)[None])[0]
