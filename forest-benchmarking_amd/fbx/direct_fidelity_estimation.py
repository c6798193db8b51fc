"""Analysis half of direct fidelity estimation (forest/benchmarking/direct_fidelity_estimation.py:
224-307).  The experiment generators and the acquisition talk to a quantum computer and are out of
scope; ``estimate_dfe`` consumes the same ``ExperimentResult`` records as the tomography estimators.
"""
import functools
from typing import List, Tuple

import numpy as np

from . import _lib


def estimate_dfe_batch(expectations, std_errs, n_qubits: int, kind: str):
    """B experiments of m settings each: ``(fidelity[B], standard_error[B])``."""
    k = kind.lower()
    if k not in ("state", "process"):
        raise ValueError('Kind can only be \'state\' or \'process\'.')
    e = np.ascontiguousarray(expectations, dtype=np.float64)
    e = e.reshape(-1, e.shape[-1])
    se = np.ascontiguousarray(std_errs, dtype=np.float64).reshape(e.shape)
    B, m = e.shape
    mean, err = np.empty(B), np.empty(B)
    _lib.check(_lib.lib().fbx_dfe_estimate(int(n_qubits), _lib.KIND_PROCESS if k == "process" else _lib.KIND_STATE,
                                           B, m, _lib.dptr(e), _lib.dptr(se), _lib.dptr(mean), _lib.dptr(err)))
    return mean, err


def estimate_dfe(results: List, kind: str) -> Tuple[float, float]:
    """direct_fidelity_estimation.py:224-307: mean fidelity and its standard error; the qubit count
    is read off the union of the observables' supports, like the reference."""
    qubits = functools.reduce(lambda x, y: set(x) | set(y),
                              [res.setting.observable.get_qubits() for res in results])
    e = np.array([np.real(res.expectation) for res in results], dtype=np.float64)
    se = np.array([res.std_err for res in results], dtype=np.float64)
    mean, err = estimate_dfe_batch(e[None], se[None], len(qubits), kind)
    return float(mean[0]), float(err[0])
