"""Input records of the hot path, without pyquil.

Plain-data mirrors of the reference classes the estimators consume
(observable_estimation.py:36-213 ``_OneQState`` / ``TensorProductState`` /
``ExperimentSetting``, :694-733 ``ExperimentResult``; ``pyquil.paulis.PauliTerm`` reduced
to what the estimators read: ``term[qubit]`` and ``.coefficient``).  The estimators are
duck-typed, so the reference's own objects work as well.
"""
import re
from dataclasses import dataclass
from typing import Tuple, Union


@dataclass(frozen=True)
class _OneQState:
    label: str
    index: int
    qubit: int

    def __str__(self):
        if self.label in ['X', 'Y', 'Z']:
            return f"{self.label}{'+' if self.index == 0 else '-'}_{self.qubit}"
        return f'{self.label}{self.index}_{self.qubit}'

    @classmethod
    def from_str(cls, s):
        ma = re.match(r'\s*(\w+)([\d+-])_(\d+)\s*', s)
        if ma is None:
            raise ValueError(f"Couldn't parse '{s}'")
        index = {'+': 0, '-': 1}.get(ma.group(2))
        if index is None:
            index = int(ma.group(2))
        return _OneQState(label=ma.group(1), index=index, qubit=int(ma.group(3)))


@dataclass(frozen=True)
class TensorProductState:
    states: Tuple[_OneQState]

    def __init__(self, states=None):
        object.__setattr__(self, 'states', tuple(states) if states is not None else tuple())

    def __mul__(self, other):
        return TensorProductState(self.states + other.states)

    def __str__(self):
        return ' * '.join(str(s) for s in self.states)

    def __getitem__(self, qubit):
        for s in self.states:
            if s.qubit == qubit:
                return s
        raise IndexError()

    def __iter__(self):
        yield from self.states

    def __len__(self):
        return len(self.states)

    @classmethod
    def from_str(cls, s):
        if s == '':
            return TensorProductState()
        return TensorProductState(tuple(_OneQState.from_str(x) for x in s.split('*')))


def SIC0(q): return TensorProductState((_OneQState('SIC', 0, q),))
def SIC1(q): return TensorProductState((_OneQState('SIC', 1, q),))
def SIC2(q): return TensorProductState((_OneQState('SIC', 2, q),))
def SIC3(q): return TensorProductState((_OneQState('SIC', 3, q),))
def plusX(q): return TensorProductState((_OneQState('X', 0, q),))
def minusX(q): return TensorProductState((_OneQState('X', 1, q),))
def plusY(q): return TensorProductState((_OneQState('Y', 0, q),))
def minusY(q): return TensorProductState((_OneQState('Y', 1, q),))
def plusZ(q): return TensorProductState((_OneQState('Z', 0, q),))
def minusZ(q): return TensorProductState((_OneQState('Z', 1, q),))


def zeros_state(qubits):
    return TensorProductState(_OneQState('Z', 0, q) for q in qubits)


class PauliTerm:
    """A coefficient times a tensor product of one-qubit Paulis (read-only subset of pyquil's)."""

    def __init__(self, ops=None, coefficient=1.0):
        self._ops = {q: op for q, op in (ops or {}).items() if op != 'I'}
        self.coefficient = complex(coefficient)

    @classmethod
    def from_list(cls, terms_list, coefficient=1.0):
        return cls({q: op for op, q in terms_list}, coefficient)

    def __getitem__(self, qubit):
        return self._ops.get(qubit, 'I')

    def __iter__(self):
        yield from self._ops.items()

    def __len__(self):
        return len(self._ops)

    def get_qubits(self):
        return list(self._ops)

    def compact_str(self):
        body = ''.join(f'{op}{q}' for q, op in sorted(self._ops.items())) or 'I'
        return f'{self.coefficient}*{body}'

    @classmethod
    def from_compact_str(cls, s):
        coef, rest = s.split('*') if '*' in s else ('1', s)
        ops = {int(q): op for op, q in re.findall(r'([XYZ])(\d+)', rest)}
        return cls(ops, complex(coef))

    def __eq__(self, other):
        return (isinstance(other, PauliTerm) and self._ops == other._ops
                and self.coefficient == other.coefficient)

    def __hash__(self):
        return hash((frozenset(self._ops.items()), self.coefficient))

    def __repr__(self):
        return self.compact_str()


@dataclass(frozen=True, init=False)
class ExperimentSetting:
    in_state: TensorProductState
    observable: PauliTerm

    def __init__(self, in_state, observable):
        object.__setattr__(self, 'in_state', in_state)
        object.__setattr__(self, 'observable', observable)

    def __str__(self):
        return f'{self.in_state}→{self.observable.compact_str()}'

    @classmethod
    def from_str(cls, s):
        instr, outstr = s.split('→')
        return ExperimentSetting(in_state=TensorProductState.from_str(instr),
                                 observable=PauliTerm.from_compact_str(outstr))


@dataclass(frozen=True)
class ExperimentResult:
    setting: ExperimentSetting
    expectation: Union[float, complex]
    total_counts: int
    std_err: Union[float, complex] = None
    raw_expectation: Union[float, complex] = None
    raw_std_err: float = None
    calibration_expectation: Union[float, complex] = None
    calibration_std_err: Union[float, complex] = None
    calibration_counts: int = None
