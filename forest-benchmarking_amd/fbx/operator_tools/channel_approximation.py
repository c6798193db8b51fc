"""Pauli twirl of a chi matrix (operator_tools/channel_approximation.py:31-49): keep the
diagonal.  A host-side slice, as in the reference."""
import numpy as np

__all__ = ["pauli_twirl_chi_matrix"]


def pauli_twirl_chi_matrix(chi_matrix: np.ndarray) -> np.ndarray:
    return np.diag(chi_matrix.diagonal())
