"""Inputs of the reference's state / process plots, prepared on the device.

The reference draws with matplotlib (``forest/benchmarking/plotting/state_process.py``, ``hinton.py``); what it
draws are small arrays derived from tomography estimates.  This module produces exactly those arrays for whole
batches of estimates -- the Pauli transfer matrix of a process (``plot_pauli_transfer_matrix``,
state_process.py:90), the Pauli-Liouville vector of a state (``plot_pauli_rep_of_state`` /
``plot_pauli_bar_rep_of_state``, :10-87) with their axis labels, and the geometry of the Hinton diagrams
(hinton.py:12-36, 52-118) -- and leaves the drawing to whoever owns a matplotlib axis.  The basis changes run in
``libfbx`` (``fbx_convert``, ``fbx_pauli_vector``); the Hinton geometry is arithmetic on the few numbers that
end up on the screen.
"""
import itertools
from typing import List, Optional, Tuple

import numpy as np

from . import _lib
from .operator_tools.superoperator_transformations import convert_batch


def pauli_labels(n_qubits: int) -> List[str]:
    """Labels of ``n_qubit_pauli_basis(n)`` (utils.py:398-409): 'II', 'IX', ... first letter = first tensor factor."""
    if n_qubits < 1:
        raise ValueError(f"n = {n_qubits} should be at least 1.")
    return ["".join(letters) for letters in itertools.product("IXYZ", repeat=n_qubits)]


def _n_qubits_of_state(dim: int) -> int:
    n = int(round(np.log2(dim)))
    if dim < 2 or 2 ** n != dim:
        raise ValueError("state dimension must be a power of two")
    return n


def pauli_transfer_matrix_plot_inputs(choi) -> Tuple[np.ndarray, List[str]]:
    """Choi matrix (or a stack [B, D, D]) -> (real Pauli transfer matrices of the same leading shape, labels): the
    ``ptransfermatrix`` and ``labels`` arguments of ``plot_pauli_transfer_matrix`` (state_process.py:90-135)."""
    x = _lib.c128(choi)
    single = x.ndim == 2
    stack = x[None] if single else x
    ptm = convert_batch("choi", "pauli_liouville", stack)
    n = int(round(np.log2(stack.shape[-1]) / 2))
    out = np.real_if_close(ptm)                     # what the reference does first (:104)
    return (out[0] if single else out), pauli_labels(n)


def state_pauli_rep_plot_inputs(rho, column: bool = True) -> Tuple[np.ndarray, List[str]]:
    """Density matrix (or a stack [B, d, d]) -> (real Pauli-Liouville vector(s), labels): ``np.real(c2p @ vec(rho))``
    as in the example of ``plot_pauli_rep_of_state`` (state_process.py:14-28).  The plot wants an (N, 1) or (1, N)
    array (:36-37): ``column`` selects which; stacks come back as [B, N, 1] / [B, 1, N]."""
    x = _lib.c128(rho)
    single = x.ndim == 2
    stack = np.ascontiguousarray(x[None] if single else x)
    if stack.ndim != 3 or stack.shape[-1] != stack.shape[-2]:
        raise ValueError("rho must be [d, d] or [B, d, d]")
    n = _n_qubits_of_state(stack.shape[-1])
    if n > 5:
        raise _lib.FbxError(_lib.FBX_ERR_UNSUPPORTED, "Pauli vectors above 5 qubits are outside this build")
    B, DD = stack.shape[0], 4 ** n
    out = np.empty((B, DD))
    _lib.check(_lib.lib().fbx_pauli_vector(n, B, _lib.dptr(stack.view(np.float64)), _lib.dptr(out)))
    out = out[:, :, None] if column else out[:, None, :]
    return (out[0] if single else out), pauli_labels(n)


def hinton_plot_inputs(matrix, max_weight: Optional[float] = 1.0) -> dict:
    """Geometry of ``hinton(matrix, max_weight)`` (hinton.py:12-36): side length and colour angle of every square.
    ``max_weight`` None / 0 -> the next power of two above the largest magnitude, as the reference picks it."""
    w = np.asarray(matrix)
    if not max_weight:
        max_weight = 2 ** np.ceil(np.log(np.abs(w).max()) / np.log(2))
    return {"size": np.sqrt(np.abs(w) / max_weight),
            "angle": np.arctan2(np.real(w), np.imag(w)),       # the reference's argument order (:28)
            "max_weight": float(max_weight),
            "xlim": (-max_weight / 2, w.shape[0] - max_weight / 2),
            "ylim": (-max_weight / 2, w.shape[1] - max_weight / 2)}


def hinton_real_plot_inputs(matrix, max_weight: Optional[float] = None) -> dict:
    """Geometry of ``hinton_real`` (hinton.py:52-118): per entry the sign class (+1 where the real part is positive,
    else -1 -- zero counts as negative there, :113-118), the blob area ``min(1, |w| / max_weight)`` and the
    default ``max_weight`` = 1.25 x the largest diagonal magnitude (1 if that is not positive, :98-101)."""
    w = np.asarray(matrix)
    if max_weight is None:
        max_weight = 1.25 * np.max(np.abs(np.diag(w)))
        if max_weight <= 0.0:
            max_weight = 1.0
    mag = np.abs(w)
    return {"sign": np.where(np.real(w) > 0.0, 1, -1),
            "area": np.minimum(1.0, mag / max_weight),
            "max_weight": float(max_weight),
            "bounds": [-max_weight, -0.0001, 0.0001, max_weight],
            "ticks": [-max_weight / 2, 0, max_weight / 2]}
