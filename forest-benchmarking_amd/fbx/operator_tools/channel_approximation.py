"""Pauli twirl of a chi matrix (operator_tools/channel_approximation.py:31-49): the twirled channel
keeps the diagonal of chi.  ``fbx_pauli_twirl_chi`` for a batch, the reference's signature for one."""
import numpy as np

from .. import _lib

__all__ = ["pauli_twirl_chi_matrix", "pauli_twirl_chi_matrix_batch"]


def pauli_twirl_chi_matrix_batch(chi) -> np.ndarray:
    c = _lib.c128(chi)
    if c.ndim != 3 or c.shape[-1] != c.shape[-2]:
        raise ValueError("chi matrices must be stacked as [B, D, D]")
    out = np.empty_like(c)
    _lib.check(_lib.lib().fbx_pauli_twirl_chi(c.shape[0], c.shape[-1], _lib.dptr(c.view(np.float64)),
                                              _lib.dptr(out.view(np.float64))))
    return out


def pauli_twirl_chi_matrix(chi_matrix: np.ndarray) -> np.ndarray:
    """channel_approximation.py:31-49.  Real input comes back real, like ``np.diag(chi.diagonal())``."""
    chi_matrix = np.asarray(chi_matrix)
    out = pauli_twirl_chi_matrix_batch(chi_matrix[None])[0]
    return out if np.iscomplexobj(chi_matrix) else np.ascontiguousarray(out.real)
