"""Kraus-operator bookkeeping (operator_tools/compose_superoperators.py:7-44) on the device.

``tensor_channel_kraus`` / ``compose_channel_kraus`` keep the reference's signatures (two lists of
operators in, a list out, in the reference's order: k1 outer, k2 inner); the ``*_batch`` forms take
stacked sets ``[B, K, rows, cols]`` -- all pairwise products of a batch in one launch
(``fbx_kraus_pairs``)."""
from typing import Sequence

import numpy as np

from .. import _lib

__all__ = ["tensor_channel_kraus", "compose_channel_kraus", "tensor_channel_kraus_batch",
           "compose_channel_kraus_batch"]


def _pairs(tensor, k2, k1):
    a, b = _lib.c128(k2), _lib.c128(k1)
    if a.ndim != 4 or b.ndim != 4 or a.shape[0] != b.shape[0]:
        raise ValueError("Kraus sets must be stacked as [B, K, rows, cols] with equal B")
    B, K2, r2, c2 = a.shape
    _, K1, r1, c1 = b.shape
    if not tensor and c2 != r1:
        raise ValueError("shapes of the Kraus operators do not compose")
    ro, co = (r2 * r1, c2 * c1) if tensor else (r2, c1)
    out = np.empty((B, K1 * K2, ro, co), dtype=np.complex128)
    _lib.check(_lib.lib().fbx_kraus_pairs(int(tensor), B, K2, r2, c2, K1, r1, c1, _lib.dptr(a.view(np.float64)),
                                          _lib.dptr(b.view(np.float64)), _lib.dptr(out.view(np.float64))))
    return out


def tensor_channel_kraus_batch(k2, k1) -> np.ndarray:
    """[B, K2, ., .] x [B, K1, ., .] -> [B, K1 K2, ., .]: operator j K2 + l is kron(k2[l], k1[j])."""
    return _pairs(True, k2, k1)


def compose_channel_kraus_batch(k2, k1) -> np.ndarray:
    """[B, K2, ., .] x [B, K1, ., .] -> [B, K1 K2, ., .]: operator j K2 + l is k2[l] @ k1[j] (k1 acts first)."""
    return _pairs(False, k2, k1)


def _stack(ops):
    return np.stack([np.asarray(k, dtype=np.complex128) for k in ops])[None]


def tensor_channel_kraus(k2: Sequence[np.ndarray], k1: Sequence[np.ndarray]) -> Sequence[np.ndarray]:
    """compose_superoperators.py:7-24: Kraus operators of the channel k2 (x) k1."""
    return list(tensor_channel_kraus_batch(_stack(k2), _stack(k1))[0])


def compose_channel_kraus(k2: Sequence[np.ndarray], k1: Sequence[np.ndarray]) -> Sequence[np.ndarray]:
    """compose_superoperators.py:27-44: Kraus operators of "k1 then k2"."""
    return list(compose_channel_kraus_batch(_stack(k2), _stack(k1))[0])
