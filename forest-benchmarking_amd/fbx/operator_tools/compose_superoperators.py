"""Kraus-operator bookkeeping (operator_tools/compose_superoperators.py:7-44): list products
of small matrices, host-side exactly as in the reference (no batch axis, no hot loop)."""
from typing import Sequence

import numpy as np

__all__ = ["tensor_channel_kraus", "compose_channel_kraus"]


def tensor_channel_kraus(k2: Sequence[np.ndarray], k1: Sequence[np.ndarray]) -> Sequence[np.ndarray]:
    return [np.kron(k2l, k1j) for k1j in k1 for k2l in k2]


def compose_channel_kraus(k2: Sequence[np.ndarray], k1: Sequence[np.ndarray]) -> Sequence[np.ndarray]:
    return [np.dot(k2l, k1j) for k1j in k1 for k2l in k2]
