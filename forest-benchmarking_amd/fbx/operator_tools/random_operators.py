"""Random states, unitaries and channels with the reference's names and RNG draw order
(forest/benchmarking/operator_tools/random_operators.py), so that code seeded through
``np.random.seed`` / ``RandomState`` sees the same matrices.  These are input generators for the
estimators, plain host numpy -- nothing here is on the accelerated path.
"""
from typing import List, Optional, Union

import numpy as np
from numpy.random import RandomState

__all__ = ["ginibre_matrix_complex", "haar_rand_unitary", "haar_rand_state", "ginibre_state_matrix",
           "bures_measure_state_matrix", "rand_map_with_BCSZ_dist", "permute_tensor_factors"]


def ginibre_matrix_complex(dim: int, k: int, rs: Optional[RandomState] = None) -> np.ndarray:
    """random_operators.py:21-46: dim x k with N(0,1) + i N(0,1) entries (real block drawn first)."""
    gen = np.random if rs is None else rs
    re = gen.randn(dim, k)
    im = gen.randn(dim, k)
    return re + 1j * im


def haar_rand_unitary(dim: int, rs=None) -> np.ndarray:
    """random_operators.py:49-72 (Mezzadri): QR of a Ginibre matrix with the phases of diag(R)
    moved into Q."""
    z = ginibre_matrix_complex(dim, dim, rs)
    q, r = np.linalg.qr(z)
    dr = np.diagonal(r)
    return q * (dr / np.abs(dr))[None, :]


def haar_rand_state(dim: int) -> np.ndarray:
    """random_operators.py:75-87: first column of a Haar unitary, as a (dim, 1) ket."""
    return haar_rand_unitary(dim)[:, :1].copy()


def ginibre_state_matrix(dim: int, rank: int) -> np.ndarray:
    """random_operators.py:90-109."""
    if rank > dim:
        raise ValueError("The rank of the state matrix cannot exceed the dimension.")
    a = ginibre_matrix_complex(dim, rank)
    m = a @ a.conj().T
    return m / np.trace(m)


def bures_measure_state_matrix(dim: int) -> np.ndarray:
    """random_operators.py:112-134: (1 + U) A A^H (1 + U)^H, normalised; A is drawn before U."""
    a = ginibre_matrix_complex(dim, dim)
    u = haar_rand_unitary(dim)
    w = np.eye(dim) + u
    p = w @ (a @ a.conj().T) @ w.conj().T
    return p / np.trace(p)


def _inv_sqrt_pd(m):
    w, v = np.linalg.eigh(m)
    return (v / np.sqrt(w)) @ v.conj().T


def rand_map_with_BCSZ_dist(dim: int, kraus_rank: int) -> np.ndarray:
    """random_operators.py:137-163 (Bruzda et al.): Choi matrix of a random CPTP map,
    (rho_in^{-1/2} (x) 1) X X^H (rho_in^{-1/2} (x) 1) in the column-stacking convention."""
    x = ginibre_matrix_complex(dim ** 2, kraus_rank)
    rho = x @ x.conj().T
    red = np.einsum("iaja->ij", rho.reshape(dim, dim, dim, dim))
    q = np.kron(_inv_sqrt_pd(red), np.eye(dim))
    return (q @ rho @ q).astype(np.complex128)


def permute_tensor_factors(dims: Union[int, List[int]], perm: List[int]) -> np.ndarray:
    """random_operators.py:166-216: permutation matrix moving tensor factor perm[i] to slot i."""
    n = len(perm)
    if isinstance(dims, int):
        dim_list = [dims] * n
    else:
        assert len(dims) == len(perm), "Please specify the dimension of each factor to be permuted."
        dim_list = [int(x) for x in dims]
    total = int(np.prod(dim_list))
    t = np.eye(total, total).reshape(dim_list + dim_list)
    t = np.transpose(t, [int(p) for p in perm] + [n + i for i in range(n)])
    return t.reshape(total, total)
