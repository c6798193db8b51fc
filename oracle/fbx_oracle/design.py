"""Experiment designs: labels, matrices, canonical setting orders, SoA flattening.

TEST INFRASTRUCTURE (see package docstring).

Label codes (shared with include/fbx.h):
  one-qubit input states: 0:X+ 1:X- 2:Y+ 3:Y- 4:Z+ 5:Z- 6:SIC0 7:SIC1 8:SIC2 9:SIC3
  one-qubit Paulis:       0:I  1:X  2:Y  3:Z
Column q of a label array refers to ``qubits[q]``; ``qubits[0]`` is the LEFT-most tensor
factor (tomography.py:154-158: the estimators reverse ``qubits`` before calling pyquil's
right-to-left lifting helpers, so the net effect is a plain left-to-right kron).
"""
import itertools
from dataclasses import dataclass

import numpy as np

STATE_CODES = {("X", 0): 0, ("X", 1): 1, ("Y", 0): 2, ("Y", 1): 3, ("Z", 0): 4, ("Z", 1): 5,
               ("SIC", 0): 6, ("SIC", 1): 7, ("SIC", 2): 8, ("SIC", 3): 9}
PAULI_CODES = {"I": 0, "X": 1, "Y": 2, "Z": 3}

# pyquil.simulation.matrices.STATES (pyquil==4.5.0), restated.
_s2, _s3 = np.sqrt(2), np.sqrt(3)
STATE_VECTORS = np.array([
    [1 / _s2, 1 / _s2], [1 / _s2, -1 / _s2],
    [1 / _s2, 1j / _s2], [1 / _s2, -1j / _s2],
    [1, 0], [0, 1],
    [1, 0],
    [1 / _s3, _s2 / _s3],
    [1 / _s3, np.exp(-2j * np.pi / 3) * _s2 / _s3],
    [1 / _s3, np.exp(2j * np.pi / 3) * _s2 / _s3],
], dtype=complex)

PAULI_MATRICES = np.array([
    [[1, 0], [0, 1]], [[0, 1], [1, 0]], [[0, -1j], [1j, 0]], [[1, 0], [0, -1]],
], dtype=complex)


def state_matrix(codes) -> np.ndarray:
    """d x d density matrix of a product input state; codes[0] is the left-most factor.
    Restates pyquil ``lifted_state_operator`` as called at tomography.py:483,513."""
    mat = np.array([[1.0 + 0j]])
    for c in codes:
        v = STATE_VECTORS[c][:, None]
        mat = np.kron(mat, v @ v.conj().T)
    return mat


def pauli_matrix(codes, coefficient=1.0) -> np.ndarray:
    """d x d matrix of a Pauli term; codes[0] is the left-most factor.
    Restates pyquil ``lifted_pauli`` as called at tomography.py:160,327,364,484,515."""
    mat = np.array([[1.0 + 0j]])
    for c in codes:
        mat = np.kron(mat, PAULI_MATRICES[c])
    return mat * coefficient


@dataclass
class Design:
    """A tomography design shared by a batch: m settings on n qubits."""
    n_qubits: int
    kind: str                 # 'state' or 'process'
    in_labels: np.ndarray     # [m, n] uint8 (state codes; ignored for kind == 'state')
    paulis: np.ndarray        # [m, n] uint8
    coefs: np.ndarray         # [m] float64 observable coefficients

    @property
    def m(self):
        return self.paulis.shape[0]

    @property
    def dim(self):
        return 2 ** self.n_qubits


def traceless_pauli_codes(n):
    """utils.py:146-156: itertools.product('IXYZ', repeat=n) minus the all-identity term."""
    return np.array(list(itertools.product(range(4), repeat=n))[1:], dtype=np.uint8)


def state_design(n) -> Design:
    """tomography.py:31-43 (_state_tomo_settings): zeros state in, every traceless Pauli out."""
    p = traceless_pauli_codes(n)
    return Design(n, "state", np.full_like(p, 4), p, np.ones(len(p)))


def process_design(n, in_basis="pauli") -> Design:
    """tomography.py:63-97,116-121: outer loop product input states, inner loop Paulis."""
    if in_basis.upper() == "SIC":
        states = [6, 7, 8, 9]
    elif in_basis.upper() == "PAULI":
        states = [0, 1, 2, 3, 4, 5]
    else:
        raise ValueError(f"Unknown basis {in_basis}")
    p = traceless_pauli_codes(n)
    ins, outs = [], []
    for s in itertools.product(states, repeat=n):
        for o in p:
            ins.append(s)
            outs.append(o)
    return Design(n, "process", np.array(ins, dtype=np.uint8), np.array(outs, dtype=np.uint8),
                  np.ones(len(outs)))


def flatten_results(results, qubits, kind):
    """Duck-typed List[ExperimentResult] -> (Design, expectations[m], counts[m]).

    Accepts any objects with ``.setting.in_state`` (iterable of objects with
    ``label, index, qubit``), ``.setting.observable`` (``obs[q]`` -> 'I'|'X'|'Y'|'Z' and
    ``.coefficient``), ``.expectation``, ``.total_counts`` (observable_estimation.py:36-213,
    694-733)."""
    n = len(qubits)
    m = len(results)
    ins = np.full((m, n), 4, dtype=np.uint8)
    outs = np.zeros((m, n), dtype=np.uint8)
    coefs = np.ones(m)
    e = np.zeros(m)
    c = np.zeros(m)
    for k, r in enumerate(results):
        obs = r.setting.observable
        for q_pos, q in enumerate(qubits):
            outs[k, q_pos] = PAULI_CODES[obs[q]]
        coefs[k] = np.real(obs.coefficient)
        if kind == "process":
            by_qubit = {s.qubit: s for s in r.setting.in_state}
            for q_pos, q in enumerate(qubits):
                s = by_qubit[q]
                ins[k, q_pos] = STATE_CODES[(s.label, s.index)]
        e[k] = r.expectation
        c[k] = r.total_counts
    return Design(n, kind, ins, outs, coefs), e, c
