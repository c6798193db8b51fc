"""Input records of the hot path, without pyquil.

Plain-data stand-ins for the reference classes the estimators consume (observable_estimation.py:36-213
``_OneQState`` / ``TensorProductState`` / ``ExperimentSetting``, :694-733 ``ExperimentResult``;
``pyquil.paulis.PauliTerm`` reduced to what the estimators read: ``term[qubit]``, iteration and
``.coefficient``).  What is kept from the reference is the INTERFACE -- attribute names, the text
form of a setting (``'X+_0 * SIC2_1→(1+0j)*X0Z1'``) and the key set of a serialised result, i.e. the
interchange schema of SURVEY.md 8f-3 -- the objects themselves are thin records around the label
codes of include/fbx.h.  The estimators are duck-typed, so the reference's own objects work as well.
"""
import ctypes as _C
import json
import re
from typing import Dict, Iterable, List, Sequence, Tuple, Union

import numpy as np

# one-qubit preparation labels <-> codes of include/fbx.h (0:X+ 1:X- 2:Y+ 3:Y- 4:Z+ 5:Z- 6..9:SIC0..3)
_PAULI_AXES = ("X", "Y", "Z")
_SIGNS = "+-"


class _OneQState:
    """A named one-qubit preparation: ``label`` in {X, Y, Z, SIC}, ``index`` (0 / 1 = the +1 / -1
    eigenstate of a Pauli, 0..3 for SIC), on ``qubit``."""
    __slots__ = ("label", "index", "qubit")

    def __init__(self, label, index, qubit):
        object.__setattr__(self, "label", str(label))
        object.__setattr__(self, "index", int(index))
        object.__setattr__(self, "qubit", int(qubit))

    def __setattr__(self, *_):
        raise AttributeError("one-qubit states are immutable")

    def _key(self):
        return self.label, self.index, self.qubit

    def __eq__(self, other):
        return isinstance(other, _OneQState) and self._key() == other._key()

    def __hash__(self):
        return hash(self._key())

    @property
    def code(self) -> int:
        """The state code of include/fbx.h."""
        if self.label in _PAULI_AXES:
            return 2 * _PAULI_AXES.index(self.label) + self.index
        return 6 + self.index

    def __str__(self):
        mark = _SIGNS[self.index] if self.label in _PAULI_AXES else str(self.index)
        return f"{self.label}{mark}_{self.qubit}"

    def __repr__(self):
        return f"_OneQState({self})"

    @classmethod
    def from_str(cls, s):
        """Inverse of ``str``: ``'X+_14'``, ``'SIC2_0'``."""
        body, sep, qubit = s.strip().rpartition("_")
        if not sep or len(body) < 2 or not qubit.isdigit() or not body[:-1].isidentifier():
            raise ValueError(f"Couldn't parse '{s}'")
        mark = body[-1]
        if mark in _SIGNS:
            index = _SIGNS.index(mark)
        elif mark.isdigit():
            index = int(mark)
        else:
            raise ValueError(f"Couldn't parse '{s}'")
        return cls(body[:-1], index, int(qubit))


class TensorProductState:
    """A product of one-qubit preparations; order of the factors is kept for printing, equality
    ignores it."""
    __slots__ = ("states",)

    def __init__(self, states=None):
        object.__setattr__(self, "states", tuple(states) if states is not None else ())

    def __setattr__(self, *_):
        raise AttributeError("product states are immutable")

    def __mul__(self, other):
        return TensorProductState(self.states + tuple(other.states))

    def __iter__(self):
        return iter(self.states)

    def __len__(self):
        return len(self.states)

    def __getitem__(self, qubit):
        hit = [s for s in self.states if s.qubit == qubit]
        if not hit:
            raise IndexError()
        return hit[0]

    def states_as_set(self):
        return frozenset(self.states)

    def __eq__(self, other):
        return isinstance(other, TensorProductState) and self.states_as_set() == other.states_as_set()

    def __hash__(self):
        return hash(self.states_as_set())

    def __str__(self):
        return " * ".join(map(str, self.states))

    def __repr__(self):
        return f"TensorProductState[{self}]"

    @classmethod
    def from_str(cls, s):
        return cls(_OneQState.from_str(tok) for tok in s.split("*")) if s else cls()


def _one_qubit_constructor(label, index, name):
    def make(q):
        return TensorProductState((_OneQState(label, index, q),))
    make.__name__ = make.__qualname__ = name
    make.__doc__ = f"The {name} preparation on qubit q (observable_estimation.py:131-168)."
    return make


for _i in range(4):
    globals()[f"SIC{_i}"] = _one_qubit_constructor("SIC", _i, f"SIC{_i}")
for _ax in _PAULI_AXES:
    globals()[f"plus{_ax}"] = _one_qubit_constructor(_ax, 0, f"plus{_ax}")
    globals()[f"minus{_ax}"] = _one_qubit_constructor(_ax, 1, f"minus{_ax}")
del _i, _ax


def zeros_state(qubits: Iterable[int]) -> TensorProductState:
    """|0...0> on the given qubits (observable_estimation.py:171-172)."""
    return TensorProductState(_OneQState("Z", 0, q) for q in qubits)


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

    def operations_as_set(self):
        return frozenset(self._ops.items())

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


class ExperimentSetting:
    """One (prepared state, measured observable) pair of an experiment; prints / parses as
    ``'<in_state>→<observable>'`` (observable_estimation.py:175-213)."""
    __slots__ = ("in_state", "observable")

    def __init__(self, in_state: TensorProductState, observable: PauliTerm):
        object.__setattr__(self, "in_state", in_state)
        object.__setattr__(self, "observable", observable)

    def __setattr__(self, *_):
        raise AttributeError("settings are immutable")

    def __eq__(self, other):
        return (isinstance(other, ExperimentSetting) and self.in_state == other.in_state
                and self.observable == other.observable)

    def __hash__(self):
        return hash((self.in_state, self.observable))

    def __str__(self):
        return f"{self.in_state}→{self.observable.compact_str()}"

    def __repr__(self):
        return f"ExperimentSetting[{self}]"

    def serializable(self):
        return str(self)

    @classmethod
    def from_str(cls, s: str):
        prepared, _, measured = s.partition("→")
        return cls(TensorProductState.from_str(prepared), PauliTerm.from_compact_str(measured))


_RESULT_FIELDS = ("setting", "expectation", "total_counts", "std_err", "raw_expectation", "raw_std_err",
                  "calibration_expectation", "calibration_std_err", "calibration_counts")


class ExperimentResult:
    """The outcome of one setting: what the estimators read is ``setting``, ``expectation`` and
    ``total_counts``; the raw / calibration fields are filled by the readout-calibration rescale
    (observable_estimation.py:694-733 for the field set)."""
    __slots__ = _RESULT_FIELDS

    def __init__(self, setting, expectation, total_counts, std_err=None, raw_expectation=None, raw_std_err=None,
                 calibration_expectation=None, calibration_std_err=None, calibration_counts=None):
        for name, value in zip(_RESULT_FIELDS, (setting, expectation, total_counts, std_err, raw_expectation,
                                                raw_std_err, calibration_expectation, calibration_std_err,
                                                calibration_counts)):
            object.__setattr__(self, name, value)

    def __setattr__(self, *_):
        raise AttributeError("results are immutable")

    def _key(self):
        return tuple(getattr(self, f) for f in _RESULT_FIELDS)

    def __eq__(self, other):
        return isinstance(other, ExperimentResult) and self._key() == other._key()

    def __hash__(self):
        return hash(self._key())

    def __str__(self):
        return f"{self.setting}: {self.expectation} +- {self.std_err}"

    def __repr__(self):
        return f"ExperimentResult[{self}]"

    def serializable(self):
        """The interchange record (same keys as observable_estimation.py:721-733, setting as its text form)."""
        rec = {"type": "ExperimentResult", "setting": str(self.setting)}
        rec.update((f, getattr(self, f)) for f in ("expectation", "std_err", "total_counts", "raw_expectation",
                                                   "raw_std_err", "calibration_expectation",
                                                   "calibration_std_err", "calibration_counts"))
        return rec


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



def calibrate_expectations_batch(expectations, std_errs, cal_means, cal_vars, cal_index=None):
    """The readout-calibration rescale for B experiments x m settings on the device
    (``fbx_calibrate_expectations``): ``expectations / cal_means[c]`` and the standard error of that ratio
    (:func:`ratio_variance`), ``c = cal_index[k]`` being the calibration of setting k's observable (default:
    one calibration per setting).  Returns ``(corrected[B, m], std_err[B, m])``."""
    from . import _lib
    e = np.ascontiguousarray(expectations, dtype=np.float64)
    se = np.ascontiguousarray(std_errs, dtype=np.float64)
    if e.ndim == 1:
        e, se = e[None], se[None]
    if e.shape != se.shape or e.ndim != 2:
        raise ValueError("expectations and std_errs must both be [B, m]")
    cm = np.ascontiguousarray(cal_means, dtype=np.float64).ravel()
    cv = np.ascontiguousarray(cal_vars, dtype=np.float64).ravel()
    if cm.shape != cv.shape:
        raise ValueError("cal_means and cal_vars must have one entry per calibration")
    ci = None if cal_index is None else np.ascontiguousarray(cal_index, dtype=np.int32).ravel()
    if ci is not None and ci.shape[0] != e.shape[1]:
        raise ValueError("cal_index must have one entry per setting")
    mean, err = np.empty_like(e), np.empty_like(e)
    _lib.check(_lib.lib().fbx_calibrate_expectations(e.shape[0], e.shape[1], _lib.dptr(e), _lib.dptr(se), _lib.iptr(ci),
                                                     cm.shape[0], _lib.dptr(cm), _lib.dptr(cv), _lib.dptr(mean),
                                                     _lib.dptr(err)))
    return mean, err


def calibrate_observable_estimates_from_moments(expt_results, calibrations) -> List["ExperimentResult"]:
    """The analysis half of ``calibrate_observable_estimates`` (observable_estimation.py:963-1049): the
    acquisition half (running the calibration programs on a QPU) is outside the path; given its outcome --
    ``calibrations[observable.operations_as_set()] = (obs_mean, obs_var, counts)``, the
    ``shots_to_obs_moments`` of each observable's calibration run (:1021-1022) -- every result's
    expectation is divided by its observable's calibration expectation, its standard error becomes the
    ratio's (:1028-1037), and the raw / calibration fields are filled like the reference's."""
    expt_results = list(expt_results)
    keys, index = [], []
    for r in expt_results:
        k = operations_as_set(r.setting.observable)
        if k not in calibrations:
            raise KeyError(f"no calibration for observable {r.setting.observable}")
        if k not in keys:
            keys.append(k)
        index.append(keys.index(k))
    means = [calibrations[k][0] for k in keys]
    variances = [calibrations[k][1] for k in keys]
    def _real(x, what):
        # The reference keeps complex expectations complex here (observables with complex coefficients).  The device
        # rescale is real arithmetic: an imaginary part that is not rounding noise is refused, not dropped silently.
        if abs(np.imag(x)) > 1e-12 * max(1.0, abs(np.real(x))):
            raise ValueError(f"calibrate_observable_estimates_from_moments: {what} {x!r} has a non-negligible imaginary "
                             f"part; the calibration rescale works on real expectations")
        return float(np.real(x))
    e = [_real(r.expectation, "expectation") for r in expt_results]
    se = [_real(r.std_err, "std_err") for r in expt_results]
    means = [_real(x, "calibration expectation") for x in means]
    mean, err = calibrate_expectations_batch(e, se, means, variances, index)
    out = []
    for r, k, m_, s_ in zip(expt_results, [keys[i] for i in index], mean[0], err[0]):
        obs_mean, obs_var, counts = calibrations[k]
        out.append(ExperimentResult(setting=r.setting, expectation=float(m_), std_err=float(s_),
                                    total_counts=r.total_counts, raw_expectation=r.expectation,
                                    raw_std_err=r.std_err, calibration_expectation=obs_mean,
                                    calibration_std_err=float(np.sqrt(obs_var)), calibration_counts=counts))
    return out


def operations_as_set(observable):
    """pyquil's ``PauliTerm.operations_as_set()``: the (qubit, operator) pairs without the coefficient."""
    if hasattr(observable, "operations_as_set"):
        return observable.operations_as_set()
    return frozenset(observable)
