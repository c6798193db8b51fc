"""Import harness for the *reference* (rigetti/forest-benchmarking at /root/reference).

TEST INFRASTRUCTURE, build-authored, used ONLY in the build container to (i) validate
the numpy oracle under ``oracle/`` against the reference itself and (ii) generate the
golden fixtures committed under ``tests/golden/``.  It never runs on the GPU box
(``/root/reference`` does not exist there) and nothing here is reference source.

The reference imports two third-party packages that are absent from this image and
cannot be installed (no network):

* ``pyquil`` (pinned ``pyquil==4.5.0``, requirements-ci.txt:91) -- the estimators use
  ``pyquil.simulation.tools.lifted_pauli`` / ``lifted_state_operator`` and the state
  table of ``pyquil.simulation.matrices`` (call sites: tomography.py:12,160,327,364,
  483-484,513,515).  Their published algorithm is restated below: the d x d matrix of a
  Pauli term / product state is the Kronecker product over ``qubits`` *in list order
  with the first listed qubit as the right-most tensor factor*, times the term's
  coefficient.  Everything else imported from pyquil (Program, gates, api) is only
  needed for the module import to succeed and is an inert placeholder.
* ``git`` (gitpython; utils.py:9) -- inert placeholder.

Always run with PYTHONDONTWRITEBYTECODE=1 so nothing is written under /root/reference.
"""
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
REFERENCE_ROOT = os.environ.get("FBX_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "forest", "benchmarking"))


class _Inert:
    """Placeholder for pyquil objects the hot path never touches."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Inert()

    def __getattr__(self, name):
        return _Inert()


class PauliTerm:
    """Stand-in for pyquil.paulis.PauliTerm: a coefficient times a product of 1q Paulis."""

    def __init__(self, op, index, coefficient=1.0):
        self._ops = {}
        if op != "I" and index is not None:
            self._ops[index] = op
        self.coefficient = complex(coefficient)

    @classmethod
    def from_list(cls, terms_list, coefficient=1.0):
        t = cls("I", 0, coefficient)
        for op, q in terms_list:
            if op != "I":
                t._ops[q] = op
        return t

    @classmethod
    def from_compact_str(cls, s):
        import re
        coef_str, rest = s.split("*") if "*" in s else ("1", s)
        t = cls("I", 0, complex(coef_str))
        if rest.strip() != "I":
            for op, q in re.findall(r"([XYZ])(\d+)", rest):
                t._ops[int(q)] = op
        return t

    def __getitem__(self, q):
        return self._ops.get(q, "I")

    def __iter__(self):
        yield from self._ops.items()

    def __len__(self):
        return len(self._ops)

    def get_qubits(self):
        return list(self._ops.keys())

    def operations_as_set(self):
        return frozenset(self._ops.items())

    def id(self, sort_ops=True):
        return "".join(f"{op}{q}" for q, op in sorted(self._ops.items()))

    def compact_str(self):
        body = self.id() or "I"
        return f"{self.coefficient}*{body}"

    def __mul__(self, other):
        if isinstance(other, PauliTerm):
            t = PauliTerm("I", 0, self.coefficient * other.coefficient)
            t._ops = dict(self._ops)
            for q, op in other._ops.items():
                if q in t._ops:
                    raise NotImplementedError("stub: products on the same qubit unused")
                t._ops[q] = op
            return t
        t = PauliTerm("I", 0, self.coefficient * other)
        t._ops = dict(self._ops)
        return t

    __rmul__ = __mul__

    def __eq__(self, other):
        return (isinstance(other, PauliTerm) and self._ops == other._ops
                and np.isclose(self.coefficient, other.coefficient))

    def __hash__(self):
        return hash((frozenset(self._ops.items()), round(self.coefficient.real, 12)))

    def __repr__(self):
        return self.compact_str()


def sI(q=None):
    return PauliTerm("I", q)


def sX(q):
    return PauliTerm("X", q)


def sY(q):
    return PauliTerm("Y", q)


def sZ(q):
    return PauliTerm("Z", q)


def is_identity(term):
    return len(term) == 0


_I2 = np.eye(2)
_X = np.array([[0.0, 1.0], [1.0, 0.0]])
_Y = np.array([[0.0, -1.0j], [1.0j, 0.0]])
_Z = np.array([[1.0, 0.0], [0.0, -1.0]])
_H = np.array([[1.0, 1.0], [1.0, -1.0]]) / np.sqrt(2)
_CNOT = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 0, 1], [0, 0, 1, 0]], dtype=float)
_GATES = {"I": _I2, "X": _X, "Y": _Y, "Z": _Z}

_SIC0 = np.array([1, 0])
_SIC1 = np.array([1, np.sqrt(2)]) / np.sqrt(3)
_SIC2 = np.array([1, np.exp(-np.pi * 2j / 3) * np.sqrt(2)]) / np.sqrt(3)
_SIC3 = np.array([1, np.exp(np.pi * 2j / 3) * np.sqrt(2)]) / np.sqrt(3)
_STATES = {
    "X": [np.array([1, 1]) / np.sqrt(2), np.array([1, -1]) / np.sqrt(2)],
    "Y": [np.array([1, 1j]) / np.sqrt(2), np.array([1, -1j]) / np.sqrt(2)],
    "Z": [np.array([1, 0]), np.array([0, 1])],
    "SIC": [_SIC0, _SIC1, _SIC2, _SIC3],
}


def lifted_pauli(pauli_term, qubits):
    """pyquil.simulation.tools.lifted_pauli restated for a single PauliTerm."""
    mat = np.array([1.0 + 0.0j])
    for q in qubits:
        mat = np.kron(_GATES[pauli_term[q]], mat)
    return mat * pauli_term.coefficient


def lifted_state_operator(state, qubits):
    """pyquil.simulation.tools.lifted_state_operator restated."""
    mat = 1.0
    for q in qubits:
        oneq = state[q]
        assert oneq.qubit == q
        v = _STATES[oneq.label][oneq.index][:, np.newaxis]
        mat = np.kron(v @ v.conj().T, mat)
    return mat


def _install_stubs():
    if "pyquil" in sys.modules and getattr(sys.modules["pyquil"], "_fbx_stub", False):
        return

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    inert = _Inert
    mod("git", Repo=inert)
    mod("pyquil", Program=inert, get_qc=inert(), _fbx_stub=True)
    mod("pyquil.api", QuantumComputer=inert, QVM=inert, BenchmarkConnection=inert,
        WavefunctionSimulator=inert, QPUCompiler=inert, get_benchmarker=inert())
    gates = mod("pyquil.gates")
    for g in ["I", "RX", "RY", "RZ", "H", "MEASURE", "RESET", "CZ", "XY", "X", "Y", "Z",
              "CNOT", "S", "T", "PHASE", "CPHASE", "SWAP", "ISWAP", "CCNOT", "NOT", "AND",
              "OR", "MOVE", "EXCHANGE", "IOR", "XOR", "NEG", "ADD", "SUB", "MUL", "DIV"]:
        setattr(gates, g, inert())
    gates.QUANTUM_GATES = {}
    mod("pyquil.quil", Program=inert, address_qubits=inert(), merge_programs=inert(),
        Pragma=inert)
    mod("pyquil.quilbase", Delay=inert, Gate=inert, Pragma=inert, Measurement=inert,
        DefGate=inert, Declare=inert)
    mod("pyquil.quilatom", Qubit=inert, QubitPlaceholder=inert, MemoryReference=inert)
    mod("pyquil.quantum_processor", NxQuantumProcessor=inert)
    mod("pyquil.noise", NoiseModel=inert)
    mod("pyquil.paulis", PauliTerm=PauliTerm, sI=sI, sX=sX, sY=sY, sZ=sZ,
        is_identity=is_identity, PauliSum=inert)
    mod("pyquil.simulation", NumpyWavefunctionSimulator=inert)
    mod("pyquil.simulation.matrices", I=_I2, X=_X, Y=_Y, Z=_Z, H=_H, CNOT=_CNOT,
        STATES=_STATES, QUANTUM_GATES=_GATES)
    mod("pyquil.simulation.tools", lifted_pauli=lifted_pauli,
        lifted_state_operator=lifted_state_operator, program_unitary=inert())
    mod("pyquil.external")
    mod("pyquil.experiment", _symmetrization=inert)


def load_reference():
    """Return a namespace with the reference modules on the hot path."""
    if not reference_available():
        raise RuntimeError(f"reference not found at {REFERENCE_ROOT}")
    _install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import importlib
    ns = types.SimpleNamespace()
    ns.tomography = importlib.import_module("forest.benchmarking.tomography")
    ns.operator_tools = importlib.import_module("forest.benchmarking.operator_tools")
    ns.calculational = importlib.import_module("forest.benchmarking.operator_tools.calculational")
    ns.random_operators = importlib.import_module("forest.benchmarking.operator_tools.random_operators")
    ns.project_state_matrix = importlib.import_module(
        "forest.benchmarking.operator_tools.project_state_matrix")
    ns.distance_measures = importlib.import_module("forest.benchmarking.distance_measures")
    ns.observable_estimation = importlib.import_module("forest.benchmarking.observable_estimation")
    ns.utils = importlib.import_module("forest.benchmarking.utils")
    ns.PauliTerm = PauliTerm
    return ns
