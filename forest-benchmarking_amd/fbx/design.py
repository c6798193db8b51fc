"""Designs: the data-independent half of a tomography experiment, shared by a batch.

Mirrors the setting generators of the reference (tomography.py:31-123) and flattens
``List[ExperimentResult]`` (observable_estimation.py:694-733) into the SoA layout of the C
ABI.  Label codes are those of include/fbx.h.
"""
import ctypes as C
import itertools

import numpy as np

from . import _lib

STATE_CODES = {("X", 0): 0, ("X", 1): 1, ("Y", 0): 2, ("Y", 1): 3, ("Z", 0): 4, ("Z", 1): 5,
               ("SIC", 0): 6, ("SIC", 1): 7, ("SIC", 2): 8, ("SIC", 3): 9}
STATE_LABELS = {v: k for k, v in STATE_CODES.items()}
PAULI_CODES = {"I": 0, "X": 1, "Y": 2, "Z": 3}
PAULI_LABELS = "IXYZ"


class Design:
    """m settings on n qubits + the device handle created from them."""

    def __init__(self, n_qubits, kind, in_labels, paulis, coefs=None):
        self.n_qubits = int(n_qubits)
        self.kind = kind
        self.paulis = np.ascontiguousarray(paulis, dtype=np.uint8).reshape(-1, self.n_qubits)
        self.m = self.paulis.shape[0]
        if in_labels is None:
            in_labels = np.full_like(self.paulis, 4)
        self.in_labels = np.ascontiguousarray(in_labels, dtype=np.uint8).reshape(self.m, self.n_qubits)
        self.coefs = (np.ones(self.m) if coefs is None
                      else np.ascontiguousarray(coefs, dtype=np.float64).reshape(self.m))
        self._handle = None

    @property
    def dim(self):
        return 2 ** self.n_qubits

    @property
    def n_states(self):
        """Distinct product input states of the design (S of the device tables)."""
        return len({r.tobytes() for r in self.in_labels})

    def key(self):
        return (self.n_qubits, self.kind, self.in_labels.tobytes(), self.paulis.tobytes(),
                self.coefs.tobytes())

    @property
    def handle(self):
        if self._handle is None:
            h = C.c_void_p()
            kind = _lib.KIND_PROCESS if self.kind == "process" else _lib.KIND_STATE
            _lib.check(_lib.lib().fbx_design_create(
                self.n_qubits, kind, self.m,
                self.in_labels.ctypes.data_as(C.POINTER(C.c_uint8)),
                self.paulis.ctypes.data_as(C.POINTER(C.c_uint8)),
                self.coefs.ctypes.data_as(C.POINTER(C.c_double)), C.byref(h)))
            self._handle = h
        return self._handle

    def __del__(self):
        try:
            if self._handle is not None:
                _lib.lib().fbx_design_destroy(self._handle)
                self._handle = None
        except Exception:
            pass


def traceless_pauli_codes(n):
    """utils.py:146-156 order: itertools.product('IXYZ', repeat=n) without the identity."""
    return np.array(list(itertools.product(range(4), repeat=n))[1:], dtype=np.uint8)


def state_design(n_qubits) -> Design:
    """Settings of generate_state_tomography_experiment (tomography.py:31-60)."""
    return Design(n_qubits, "state", None, traceless_pauli_codes(n_qubits))


def process_design(n_qubits, in_basis="pauli") -> Design:
    """Settings of generate_process_tomography_experiment (tomography.py:63-123)."""
    if in_basis.upper() == "SIC":
        states = [6, 7, 8, 9]
    elif in_basis.upper() == "PAULI":
        states = [0, 1, 2, 3, 4, 5]
    else:
        raise ValueError(f"Unknown basis {in_basis}")
    p = traceless_pauli_codes(n_qubits)
    ins = np.repeat(np.array(list(itertools.product(states, repeat=n_qubits)), dtype=np.uint8),
                    len(p), axis=0)
    outs = np.tile(p, (len(states) ** n_qubits, 1))
    return Design(n_qubits, "process", ins, outs)


_design_cache = {}


def flatten_results(results, qubits, kind):
    """List[ExperimentResult] (duck-typed) -> (Design, expectations[m], total_counts[m]).

    ``qubits[0]`` is the left-most tensor factor (tomography.py:149-158).  Designs are cached
    by content so repeated calls with the same settings reuse the device copy."""
    qubits = list(qubits)
    n, m = len(qubits), len(results)
    ins = np.full((m, n), 4, dtype=np.uint8)
    outs = np.zeros((m, n), dtype=np.uint8)
    coefs = np.ones(m)
    e = np.zeros(m)
    c = np.zeros(m)
    for k, r in enumerate(results):
        obs = r.setting.observable
        if not set(obs.get_qubits()) <= set(qubits):
            raise ValueError(f"observable {obs} acts on qubits outside {qubits}")
        for pos, q in enumerate(qubits):
            outs[k, pos] = PAULI_CODES[obs[q]]
        coef = complex(getattr(obs, "coefficient", 1.0))
        if abs(coef.imag) > 0:
            raise ValueError("observable coefficients must be real")
        coefs[k] = coef.real
        if kind == "process":
            by_qubit = {s.qubit: s for s in r.setting.in_state}
            for pos, q in enumerate(qubits):
                s = by_qubit[q]
                ins[k, pos] = STATE_CODES[(s.label, s.index)]
        e[k] = np.real(r.expectation)
        c[k] = r.total_counts
    d = Design(n, kind, ins, outs, coefs)
    cached = _design_cache.get(d.key())
    if cached is None:
        if len(_design_cache) > 64:
            _design_cache.clear()
        _design_cache[d.key()] = d
        cached = d
    return cached, e, c


# --------------------------------------------------------------------------------------------------
# compact structure-of-arrays archive for the batched estimators: one design shared
# by B experiments, label codes as in include/fbx.h
# --------------------------------------------------------------------------------------------------
def save_batch(fn, design: "Design", expectations, total_counts=None):
    """Write (design, expectations[B, m], total_counts[B, m]) to a compressed ``.npz``."""
    e = np.asarray(expectations, dtype=np.float64).reshape(-1, design.m)
    arrays = dict(format=np.array("fbx-soa-1"), n_qubits=np.array(design.n_qubits),
                  kind=np.array(design.kind), in_labels=design.in_labels, paulis=design.paulis,
                  coefs=design.coefs, expectations=e)
    if total_counts is not None:
        arrays["total_counts"] = np.asarray(total_counts, dtype=np.float64).reshape(e.shape)
    np.savez_compressed(fn, **arrays)
    return fn


def load_batch(fn):
    """Inverse of :func:`save_batch`: ``(Design, expectations, total_counts or None)``; the design is
    taken from the content-keyed cache when an identical one already lives on the device."""
    with np.load(fn) as z:
        if str(z["format"]) != "fbx-soa-1":
            raise ValueError(f"{fn}: not an fbx structure-of-arrays archive")
        d = Design(int(z["n_qubits"]), str(z["kind"]), z["in_labels"], z["paulis"], z["coefs"])
        e = z["expectations"]
        c = z["total_counts"] if "total_counts" in z.files else None
    d = _design_cache.setdefault(d.key(), d)
    return d, e, c
