"""Random states, unitaries and channels with the reference's names and RNG draw order
(forest/benchmarking/operator_tools/random_operators.py), so that code seeded through
``np.random.seed`` / ``RandomState`` sees the same matrices.  These are input generators for the
estimators.

Two families:
  * the reference-signature functions below are host numpy with the reference's draw order (a seeded
    run reproduces the reference's matrices bit for bit) -- one matrix per call, like the reference;
  * the ``*_batch`` functions generate B items ON THE DEVICE (``fbx_random_operators`` /
    ``fbx_random_kraus``: counter-based Philox4x32-10 stream keyed by ``seed`` and the item id), which
    is what the batched estimators and the conversion sweep consume.  Item ``first_item + b`` is the
    same matrix whatever the batch size, launch shape or split over GPUs; parity with the
    reference is distributional.
"""
from typing import List, Optional, Union

import numpy as np
from numpy.random import RandomState

__all__ = ["ginibre_matrix_complex", "haar_rand_unitary", "haar_rand_state", "ginibre_state_matrix",
           "bures_measure_state_matrix", "rand_map_with_BCSZ_dist", "permute_tensor_factors",
           "ginibre_matrix_complex_batch", "haar_rand_unitary_batch", "haar_rand_state_batch",
           "ginibre_state_matrix_batch", "bures_measure_state_matrix_batch", "random_kraus_batch",
           "rand_map_with_BCSZ_dist_batch"]


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


# --------------------------------------------------------------------------------------------------
# device generators (batched)
# --------------------------------------------------------------------------------------------------
def _qubits_of(dim):
    n = int(dim).bit_length() - 1
    if dim not in (2, 4, 8):
        raise ValueError("the device generators take dim in {2, 4, 8} (1..3 qubits)")
    return n


def _stream_seed(seed) -> int:
    """The Philox key of one batched call.  The device streams are keyed by (seed, item id, element) only, NOT by the
    kind of operator: two calls with the same seed and item ids draw from the same normals (a Haar unitary is then the Q
    factor of the Ginibre matrix of the same call signature).  ``seed=None`` therefore takes a fresh 64-bit key from
    numpy's global stream -- independent calls, like the reference's functions, and reproducible under ``np.random.seed``;
    pass explicit, DIFFERENT seeds to name streams yourself."""
    if seed is None:
        return int(np.random.randint(0, 2 ** 63 - 1, dtype=np.int64)) * 2 + int(np.random.randint(0, 2))
    return int(seed) & (2 ** 64 - 1)


def _device_random(kind, dim, cols_or_rank, batch, seed, first_item, shape):
    from .. import _lib
    out = np.empty((int(batch),) + shape, dtype=np.complex128)
    _lib.check(_lib.lib().fbx_random_operators(kind, int(dim), int(cols_or_rank), int(batch),
                                               _stream_seed(seed), int(first_item),
                                               _lib.dptr(out.view(np.float64))))
    return out


def ginibre_matrix_complex_batch(dim: int, k: int, batch: int, seed: Optional[int] = None, first_item: int = 0) -> np.ndarray:
    """``[batch, dim, k]`` complex Ginibre matrices (random_operators.py:21-46)."""
    from .. import _lib
    return _device_random(_lib.RAND_GINIBRE, dim, k, batch, seed, first_item, (int(dim), int(k)))


def haar_rand_unitary_batch(dim: int, batch: int, seed: Optional[int] = None, first_item: int = 0) -> np.ndarray:
    """``[batch, dim, dim]`` Haar unitaries (random_operators.py:49-72)."""
    from .. import _lib
    _qubits_of(dim)
    return _device_random(_lib.RAND_UNITARY, dim, 0, batch, seed, first_item, (dim, dim))


def haar_rand_state_batch(dim: int, batch: int, seed: Optional[int] = None, first_item: int = 0) -> np.ndarray:
    """``[batch, dim, 1]`` Haar-random kets (random_operators.py:75-89)."""
    from .. import _lib
    _qubits_of(dim)
    return _device_random(_lib.RAND_STATE_VECTOR, dim, 0, batch, seed, first_item, (dim,))[:, :, None]


def ginibre_state_matrix_batch(dim: int, rank: int, batch: int, seed: Optional[int] = None, first_item: int = 0) -> np.ndarray:
    """``[batch, dim, dim]`` rank-``rank`` states of the Ginibre ensemble (random_operators.py:92-112)."""
    from .. import _lib
    _qubits_of(dim)
    if rank > dim:
        raise ValueError("The rank of the state matrix cannot exceed the dimension.")
    return _device_random(_lib.RAND_GINIBRE_STATE, dim, rank, batch, seed, first_item, (dim, dim))


def bures_measure_state_matrix_batch(dim: int, batch: int, seed: Optional[int] = None, first_item: int = 0) -> np.ndarray:
    """``[batch, dim, dim]`` states of the Bures measure (random_operators.py:115-132)."""
    from .. import _lib
    _qubits_of(dim)
    return _device_random(_lib.RAND_BURES_STATE, dim, 0, batch, seed, first_item, (dim, dim))


def random_kraus_batch(dim: int, kraus_rank: int, batch: int, seed: Optional[int] = None, first_item: int = 0) -> np.ndarray:
    """``[batch, kraus_rank, dim, dim]`` CPTP Kraus sets K_j = G_j S^{-1/2} (BCSZ ensemble in Kraus form)."""
    from .. import _lib
    n = _qubits_of(dim)
    out = np.empty((int(batch), int(kraus_rank), dim, dim), dtype=np.complex128)
    _lib.check(_lib.lib().fbx_random_kraus(n, int(batch), int(kraus_rank), _stream_seed(seed),
                                           int(first_item), _lib.dptr(out.view(np.float64))))
    return out


def rand_map_with_BCSZ_dist_batch(dim: int, kraus_rank: int, batch: int, seed: Optional[int] = None, first_item: int = 0) -> np.ndarray:
    """``[batch, dim^2, dim^2]`` Choi matrices of random CPTP maps (random_operators.py:135-157): the Kraus
    sets are generated and converted without leaving HBM."""
    from .. import _lib
    n = _qubits_of(dim)
    B, K, D = int(batch), int(kraus_rank), dim * dim
    lib = _lib.lib()
    d_k, d_c = _lib.DeviceBuffer(max(16, B * K * D * 16)), _lib.DeviceBuffer(max(16, B * D * D * 16))
    _lib.check(lib.fbx_random_kraus_dev(n, B, K, _stream_seed(seed), int(first_item), d_k.ptr))
    _lib.check(lib.fbx_convert_dev(_lib.REP_KRAUS, _lib.REP_CHOI, n, B, d_k.ptr, K, d_c.ptr))
    _lib.synchronize()
    out = d_c.to_array(np.complex128, (B, D, D))
    d_k.free(); d_c.free()
    return out
