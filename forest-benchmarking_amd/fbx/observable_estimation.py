"""Input records of the hot path, without pyquil.

Plain-data mirrors of the reference classes the estimators consume
(observable_estimation.py:36-213 ``_OneQState`` / ``TensorProductState`` /
``ExperimentSetting``, :694-733 ``ExperimentResult``; ``pyquil.paulis.PauliTerm`` reduced
to what the estimators read: ``term[qubit]`` and ``.coefficient``).  The estimators are
duck-typed, so the reference's own objects work as well.
"""
import ctypes as _C
import json
import re
from dataclasses import dataclass
from typing import Dict, Iterable, List, Sequence, Tuple, Union

import numpy as np


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


    def serializable(self):
        """observable_estimation.py:721-733 (same keys, setting as its string form)."""
        return {
            'type': 'ExperimentResult',
            'setting': str(self.setting),
            'expectation': self.expectation,
            'std_err': self.std_err,
            'total_counts': self.total_counts,
            'raw_expectation': self.raw_expectation,
            'raw_std_err': self.raw_std_err,
            'calibration_expectation': self.calibration_expectation,
            'calibration_std_err': self.calibration_std_err,
            'calibration_counts': self.calibration_counts,
        }


# ==================================================================================================
# interchange (observable_estimation.py:356-389) and result bookkeeping (:1145-1173)
# ==================================================================================================
class OperatorEncoder(json.JSONEncoder):
    def default(self, o):
        if isinstance(o, ExperimentSetting):
            return str(o)
        if isinstance(o, ExperimentResult):
            return o.serializable()
        if isinstance(o, (np.integer,)):
            return int(o)
        if isinstance(o, (np.floating,)):
            return float(o)
        return super().default(o)


def to_json(fn, obj):
    """observable_estimation.py:367-373: same file layout as the reference (indent 2, settings as
    ``'X+_0 * Z-_1→(1+0j)*X0Z1'`` strings), for ExperimentSetting / ExperimentResult objects and
    lists of them."""
    with open(fn, 'w') as f:
        json.dump(obj, f, cls=OperatorEncoder, indent=2, ensure_ascii=False)
    return fn


def _result_object_hook(obj):
    if obj.get('type') == 'ExperimentResult':
        fields = {k: v for k, v in obj.items() if k != 'type'}
        fields['setting'] = ExperimentSetting.from_str(fields['setting'])
        return ExperimentResult(**fields)
    return obj


def read_json(fn):
    """observable_estimation.py:384-389.  Unlike the reference (whose hook only rebuilds
    ObservablesExperiment and leaves results as dicts) ExperimentResult records come back as
    ExperimentResult objects, ready for the estimators; files written by the reference parse too."""
    with open(fn) as f:
        return json.load(f, object_hook=_result_object_hook)


def get_results_by_qubit_groups(results: Iterable, qubit_groups: Sequence[Sequence[int]]) -> Dict[Tuple[int, ...], list]:
    """observable_estimation.py:1145-1173: results whose observable acts inside a group, keyed by
    the sorted group; order kept, a result may land in several overlapping groups.  One entry is
    the natural unit of the batched estimators."""
    groups = [tuple(sorted(g)) for g in qubit_groups]
    out = {g: [] for g in groups}
    sets = [set(g) for g in groups]
    for res in results:
        acts_on = set(res.setting.observable.get_qubits())
        for g, gs in zip(groups, sets):
            if acts_on <= gs:
                out[g].append(res)
    return out


# ==================================================================================================
# shots -> moments (observable_estimation.py:804-853, :1052-1090): the step just before the estimators
# ==================================================================================================
def shots_to_obs_moments_batch(bitarrays, obs_masks, coefs=None, use_beta_dist_unbiased_prior=False):
    """Mean and variance-of-the-mean of the +-1 products for S settings at once.

    bitarrays [S, n_shots, n_qubits] of 0/1 (any integer dtype; converted to uint8),
    obs_masks [S, n_qubits] non-zero where the setting's observable acts, coefs [S] (default 1)."""
    from . import _lib
    bits = np.ascontiguousarray(bitarrays, dtype=np.uint8)
    if bits.ndim != 3:
        raise ValueError("bitarrays must be [S, n_shots, n_qubits]")
    S, shots, n = bits.shape
    masks = np.ascontiguousarray(obs_masks, dtype=np.uint8).reshape(S, n)
    cf = None if coefs is None else np.ascontiguousarray(coefs, dtype=np.float64).reshape(S)
    mean = np.empty(S)
    var = np.empty(S)
    u8 = _C.POINTER(_C.c_uint8)
    _lib.check(_lib.lib().fbx_shots_to_moments(n, S, shots, bits.ctypes.data_as(u8), masks.ctypes.data_as(u8),
                                               _lib.dptr(cf), int(bool(use_beta_dist_unbiased_prior)),
                                               _lib.dptr(mean), _lib.dptr(var)))
    return mean, var


def shots_to_obs_moments(bitarray: np.ndarray, qubits: List[int], observable,
                         use_beta_dist_unbiased_prior: bool = False) -> Tuple[float, float]:
    """observable_estimation.py:804-853 with the reference's signature."""
    coeff = complex(observable.coefficient)
    if not np.isclose(coeff.imag, 0):
        raise ValueError("The coefficient of an observable should not be complex.")
    obs_qubits = [q for q, _ in observable]
    mask = np.array([1 if q in obs_qubits else 0 for q in qubits], dtype=np.uint8)
    if not mask.any():                      # identity term
        return coeff.real, 0
    bitarray = np.asarray(bitarray)
    assert bitarray.shape[1] == len(qubits), 'qubits should label each column of the bitarray'
    mean, var = shots_to_obs_moments_batch(bitarray[None], mask[None], [coeff.real],
                                           use_beta_dist_unbiased_prior)
    return float(mean[0]), float(var[0])


def ratio_variance(a, var_a, b, var_b):
    """observable_estimation.py:1052-1090: Var[A/B] ~ var_a / b^2 + a^2 var_b / b^4 (element-wise)."""
    return var_a / b ** 2 + (a ** 2 * var_b) / b ** 4
