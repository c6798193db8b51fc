"""Matrix predicates (mirror of operator_tools/validate_operator.py:6-150).

The comparisons are ``np.allclose`` predicates exactly as in the reference; the
positive-(semi)definiteness checks take their eigenvalues from the device eigensolver
(``fbx_eigh``: any N up to 1024 -- qutrits and 3- to 5-qubit Choi matrices included; N <= 64 in LDS, above that
with the matrix in HBM; larger matrices raise ``FbxError`` -- not the ``ValueError`` the reference reserves for
non-Hermitian input)."""
import numpy as np

from .. import _lib

__all__ = ["is_square_matrix", "is_symmetric_matrix", "is_identity_matrix", "is_idempotent_matrix",
           "is_normal_matrix", "is_hermitian_matrix", "is_unitary_matrix",
           "is_positive_definite_matrix", "is_positive_semidefinite_matrix"]


def is_square_matrix(matrix: np.ndarray) -> bool:
    if len(matrix.shape) != 2:
        raise ValueError("The object is not a matrix.")
    rows, cols = matrix.shape
    return rows == cols


def _square(matrix):
    if not is_square_matrix(matrix):
        raise ValueError("The matrix is not square.")


def is_symmetric_matrix(matrix, rtol: float = 1e-05, atol: float = 1e-08) -> bool:
    _square(matrix)
    return np.allclose(matrix, matrix.T, rtol=rtol, atol=atol)


def is_identity_matrix(matrix, rtol: float = 1e-05, atol: float = 1e-08) -> bool:
    _square(matrix)
    return np.allclose(matrix, np.eye(len(matrix)), rtol=rtol, atol=atol)


def is_idempotent_matrix(matrix, rtol: float = 1e-05, atol: float = 1e-08) -> bool:
    _square(matrix)
    return np.allclose(matrix, matrix @ matrix, rtol=rtol, atol=atol)


def is_normal_matrix(matrix, rtol: float = 1e-05, atol: float = 1e-08) -> bool:
    _square(matrix)
    return np.allclose(matrix.T.conj() @ matrix, matrix @ matrix.T.conj(), rtol=rtol, atol=atol)


def is_hermitian_matrix(matrix, rtol: float = 1e-05, atol: float = 1e-08) -> bool:
    _square(matrix)
    return np.allclose(matrix, matrix.T.conj(), rtol=rtol, atol=atol)


def is_unitary_matrix(matrix, rtol: float = 1e-05, atol: float = 1e-08) -> bool:
    _square(matrix)
    eye = np.eye(len(matrix))
    return (np.allclose(matrix.T.conj() @ matrix, eye, rtol=rtol, atol=atol)
            and np.allclose(matrix @ matrix.T.conj(), eye, rtol=rtol, atol=atol))


def _eigvalsh(matrix):
    return _lib.eigh_batch(np.asarray(matrix)[None], eigenvectors=False)[0]


def is_positive_definite_matrix(matrix, rtol: float = 1e-05, atol: float = 1e-08) -> bool:
    if not is_hermitian_matrix(matrix, rtol, atol):
        raise ValueError("The matrix is not Hermitian.")
    return all(x > -abs(atol) for x in _eigvalsh(matrix))


def is_positive_semidefinite_matrix(matrix, rtol: float = 1e-05, atol: float = 1e-08) -> bool:
    if not is_hermitian_matrix(matrix, rtol, atol):
        raise ValueError("The matrix is not Hermitian.")
    return all(x >= -abs(atol) for x in _eigvalsh(matrix))
