"""Tomography estimators with the reference's names and signatures, running on MI355X.

Mirror of forest/benchmarking/tomography.py:130-633.  Every estimator takes the same
``(results: List[ExperimentResult], qubits: List[int], **kwargs)`` and returns the same dense
complex128 ``np.ndarray``; ``*_batch`` variants take SoA arrays ``expectations[B, m]``,
``total_counts[B, m]`` plus a :class:`fbx.design.Design` and are where the throughput lives.
"""
import warnings
from typing import Callable, List, Sequence, Tuple

import numpy as np

from . import _lib
from . import distance_measures as dm
from .design import Design, flatten_results, process_design, state_design  # noqa: F401
from .observable_estimation import (ExperimentResult, ExperimentSetting, PauliTerm,
                                    TensorProductState, zeros_state, SIC0, SIC1, SIC2, SIC3,  # noqa: F401
                                    plusX, minusX, plusY, minusY, plusZ, minusZ)  # noqa: F401
from .observable_estimation import _OneQState
from .design import PAULI_LABELS, STATE_LABELS

import functools


# ==================================================================================================
# Experiment settings (tomography.py:31-123) -- read off the design tables (fbx.design), which hold
# the canonical order: input states outer, traceless Pauli observables inner.  No pyquil Program.
# ==================================================================================================
def _settings_of(design: Design, qubits: Sequence[int]) -> List[ExperimentSetting]:
    qubits = list(qubits)
    out = []
    for prep_codes, pauli_codes in zip(design.in_labels, design.paulis):
        prepared = TensorProductState(_OneQState(*STATE_LABELS[int(c)], q) for c, q in zip(prep_codes, qubits))
        measured = PauliTerm({q: PAULI_LABELS[int(c)] for c, q in zip(pauli_codes, qubits)})
        out.append(ExperimentSetting(in_state=prepared, observable=measured))
    return out


def generate_state_tomography_settings(qubits: List[int]) -> List[ExperimentSetting]:
    """The settings of generate_state_tomography_experiment (tomography.py:46-60): |0..0> in, every
    traceless Pauli out."""
    return _settings_of(state_design(len(qubits)), qubits)


def generate_process_tomography_settings(qubits: List[int], in_basis='pauli') -> List[ExperimentSetting]:
    """The settings of generate_process_tomography_experiment (tomography.py:100-123); an unknown
    ``in_basis`` raises the reference's ``ValueError``."""
    return _settings_of(process_design(len(qubits), in_basis), qubits)


def _batch_arrays(design, expectations, total_counts=None):
    e = np.ascontiguousarray(expectations, dtype=np.float64)
    if e.ndim == 1:
        e = e[None, :]
    if e.ndim != 2 or e.shape[1] != design.m:
        raise ValueError(f"expectations must have shape [B, {design.m}]")
    if total_counts is None:
        return e, None
    c = np.ascontiguousarray(total_counts, dtype=np.float64)
    if c.ndim == 1:
        c = np.ascontiguousarray(np.broadcast_to(c[None, :], e.shape))
    if c.shape != e.shape:
        raise ValueError("total_counts must have the shape of expectations")
    return e, c


# ==================================================================================================
# STATE tomography
# ==================================================================================================
def linear_inv_state_estimate_batch(design: Design, expectations) -> np.ndarray:
    e, _ = _batch_arrays(design, expectations)
    d = design.dim
    out = np.empty((e.shape[0], d, d), dtype=np.complex128)
    _lib.check(_lib.lib().fbx_linv_state(design.handle, e.shape[0], _lib.dptr(e),
                                         _lib.dptr(out.view(np.float64))))
    return out


def linear_inv_state_estimate(results: List[ExperimentResult], qubits: List[int]) -> np.ndarray:
    """tomography.py:130-165."""
    design, e, _ = flatten_results(results, qubits, "state")
    return linear_inv_state_estimate_batch(design, e)[0]


def iterative_mle_state_estimate_batch(design: Design, expectations, total_counts, epsilon=.1,
                                       entropy_penalty=0.0, beta=0.0, tol=1e-9, maxiter=10_000,
                                       return_stats=False):
    if (entropy_penalty != 0.0) and (beta != 0.0):
        raise ValueError("One can't sensibly do entropy penalty and hedging. Do one or the other"
                         " but not both.")
    e, c = _batch_arrays(design, expectations, total_counts)
    B, d = e.shape[0], design.dim
    rho = np.empty((B, d, d), dtype=np.complex128)
    iters = np.zeros(B, dtype=np.int32)
    hit = np.zeros(B, dtype=np.int32)
    _lib.check(_lib.lib().fbx_mle_state(design.handle, B, _lib.dptr(e), _lib.dptr(c),
                                        float(epsilon), float(entropy_penalty), float(beta),
                                        float(tol), int(maxiter), _lib.dptr(rho.view(np.float64)),
                                        _lib.iptr(iters), _lib.iptr(hit)))
    if hit.any():
        warnings.warn('Maximum number of iterations reached before convergence.')
    if return_stats:
        return rho, {"iterations": iters, "hit_max": hit.astype(bool)}
    return rho


def iterative_mle_state_estimate(results: List[ExperimentResult], qubits: List[int], epsilon=.1,
                                 entropy_penalty=0.0, beta=0.0, tol=1e-9, maxiter=10_000) \
        -> np.ndarray:
    """tomography.py:168-270 (diluted iterative MLE; max-entropy or hedged variants)."""
    if (entropy_penalty != 0.0) and (beta != 0.0):
        raise ValueError("One can't sensibly do entropy penalty and hedging. Do one or the other"
                         " but not both.")
    design, e, c = flatten_results(results, qubits, "state")
    return iterative_mle_state_estimate_batch(design, e, c, epsilon, entropy_penalty, beta, tol,
                                              maxiter)[0]


def _R_batch(states, design: Design, expectations) -> np.ndarray:
    e, _ = _batch_arrays(design, expectations)
    d = design.dim
    rho = _lib.c128(states).reshape(-1, d, d)
    out = np.empty_like(rho)
    _lib.check(_lib.lib().fbx_r_operator(design.handle, rho.shape[0], _lib.dptr(rho.view(np.float64)),
                                         _lib.dptr(e), _lib.dptr(out.view(np.float64))))
    return out


def _R(state, results, qubits):
    """tomography.py:273-338.  NOTE: like the reference's private helper, ``qubits`` here is
    the *reversed* list the public estimators pass down (tomography.py:233,248)."""
    design, e, _ = flatten_results(results, list(qubits)[::-1], "state")
    return _R_batch(np.asarray(state)[None], design, e)[0]


def state_log_likelihood_batch(states, design: Design, expectations, total_counts) -> np.ndarray:
    e, c = _batch_arrays(design, expectations, total_counts)
    d = design.dim
    rho = _lib.c128(states).reshape(-1, d, d)
    out = np.empty(rho.shape[0])
    _lib.check(_lib.lib().fbx_state_log_likelihood(design.handle, rho.shape[0],
                                                   _lib.dptr(rho.view(np.float64)), _lib.dptr(e),
                                                   _lib.dptr(c), _lib.dptr(out)))
    return out


def state_log_likelihood(state: np.ndarray, results, qubits: Sequence[int]) -> float:
    """tomography.py:341-375 (log10 likelihood)."""
    design, e, c = flatten_results(list(results), qubits, "state")
    return float(state_log_likelihood_batch(np.asarray(state)[None], design, e, c)[0])


def _resample_expectations_with_beta(results, prior_counts=1):
    """tomography.py:378-409: one Beta-posterior redraw of every expectation, from numpy's global
    stream in result order (one vectorised ``np.random.beta`` call draws in exactly that order)."""
    e = np.array([np.real(r.expectation) for r in results], dtype=float)
    c = np.array([r.total_counts for r in results], dtype=float)
    redrawn = resample_expectations_with_beta_batch(e, c, 1, prior_counts)[0]
    return [ExperimentResult(setting=r.setting, expectation=float(x), std_err=r.std_err, total_counts=r.total_counts)
            for r, x in zip(results, redrawn)]


def resample_expectations_with_beta_batch(expectations, total_counts, n_resamples, prior_counts=1, seed=None):
    """Beta-resampled expectations, ``[n_resamples, *expectations.shape]``.

    ``seed=None``: drawn on the host from the global np.random stream in the order the reference
    draws them (resample by resample, result by result; tomography.py:391-402) -- expectations must
    then be one experiment ``[m]``.  ``seed=int``: drawn on the device by the counter-based generator
    of ``fbx_beta_resample`` (reproducible, any batch shape; same distribution, different stream)."""
    e = np.asarray(expectations, dtype=float)
    c = np.asarray(total_counts, dtype=float)
    if seed is not None:
        e = np.ascontiguousarray(e)
        c = np.ascontiguousarray(np.broadcast_to(c, e.shape))
        out = np.empty((int(n_resamples),) + e.shape)
        _lib.check(_lib.lib().fbx_beta_resample(e.size, int(n_resamples), _lib.dptr(e), _lib.dptr(c),
                                               float(prior_counts), int(seed) & (2 ** 64 - 1), _lib.dptr(out)))
        return out
    num_plus = ((e + 1) / 2) * c
    num_minus = c - num_plus
    a = np.broadcast_to(num_plus + prior_counts, (n_resamples, e.size))
    b = np.broadcast_to(num_minus + prior_counts, (n_resamples, e.size))
    return 2 * np.random.beta(a, b) - 1


_BATCHED_ESTIMATORS = {}     # filled below: reference-signature estimator -> batched implementation


def _batched_form(tomo_estimator):
    """(batched function, kwargs) when `tomo_estimator` is one of this module's state estimators
    (possibly wrapped in functools.partial), else None."""
    kwargs = {}
    f = tomo_estimator
    while isinstance(f, functools.partial):
        if f.args:
            return None
        kwargs = {**f.keywords, **kwargs}
        f = f.func
    impl = _BATCHED_ESTIMATORS.get(f)
    return (impl, kwargs) if impl is not None else None


def estimate_variance(results: List[ExperimentResult], qubits: List[int], tomo_estimator: Callable,
                      functional: Callable, target_state=None, n_resamples: int = 40,
                      project_to_physical: bool = False, seed=None) -> Tuple[float, float]:
    """tomography.py:412-453 (bootstrap error bar of a functional of the state).

    When `tomo_estimator` is one of this module's state estimators and `functional` one of
    ``dm.purity / fidelity / infidelity / trace_distance / hilbert_schmidt_ip`` the whole bootstrap
    is three batched device calls (estimate, project, measure) over the `n_resamples` resampled
    experiments -- the resamples are the batch axis; any other callables take the reference's
    one-at-a-time loop.  ``seed`` (extension): None = the reference's np.random stream, an int = the
    device generator (batched path only)."""
    from .operator_tools.project_state_matrix import (project_state_matrix_to_physical,
                                                       project_state_matrix_to_physical_batch)
    if functional != dm.purity:
        if target_state is None:
            raise ValueError("You're not using the `purity` functional. "
                             "Please specify a target state.")
    batched = _batched_form(tomo_estimator)
    measures = {dm.purity: "purity", dm.fidelity: "fidelity", dm.infidelity: "fidelity",
                dm.trace_distance: "trace_distance", dm.hilbert_schmidt_ip: "hs_ip"}
    if batched is not None and functional in measures:
        impl, kwargs = batched
        design, e, c = flatten_results(results, qubits, "state")
        e_rs = resample_expectations_with_beta_batch(e, c, n_resamples, seed=seed)
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            rhos = impl(design, e_rs, np.broadcast_to(c, e_rs.shape), **kwargs)
        for w in {str(w.message) for w in caught}:
            warnings.warn(w)
        if project_to_physical:
            rhos = project_state_matrix_to_physical_batch(rhos)
        key = measures[functional]
        if functional == dm.purity:
            vals = dm.state_measures_batch(rhos, None, (key,))[key]
        else:
            tgt = np.broadcast_to(np.asarray(target_state, dtype=np.complex128), rhos.shape)
            vals = dm.state_measures_batch(np.ascontiguousarray(tgt), rhos, (key,))[key]
            if functional == dm.infidelity:
                vals = 1 - vals
        return np.mean(vals), np.var(vals)
    # opaque callables: one reconstruction per resample, through whatever the caller passed in
    def one_sample():
        rho = tomo_estimator(_resample_expectations_with_beta(results), qubits)
        if project_to_physical:
            rho = project_state_matrix_to_physical(rho)
        value = dm.purity(rho, dim_renorm=False) if functional == dm.purity else functional(target_state, rho)
        return np.real(value)

    values = np.array([one_sample() for _ in range(n_resamples)])
    return np.mean(values), np.var(values)


_BATCHED_ESTIMATORS[linear_inv_state_estimate] = lambda design, e, c: linear_inv_state_estimate_batch(design, e)
_BATCHED_ESTIMATORS[iterative_mle_state_estimate] = iterative_mle_state_estimate_batch


# ==================================================================================================
# PROCESS tomography
# ==================================================================================================
def linear_inv_process_estimate_batch(design: Design, expectations) -> np.ndarray:
    e, _ = _batch_arrays(design, expectations)
    D = design.dim ** 2
    out = np.empty((e.shape[0], D, D), dtype=np.complex128)
    _lib.check(_lib.lib().fbx_linv_process(design.handle, e.shape[0], _lib.dptr(e),
                                           _lib.dptr(out.view(np.float64))))
    return out


def linear_inv_process_estimate(results: List[ExperimentResult], qubits: List[int]) -> np.ndarray:
    """tomography.py:459-491."""
    design, e, _ = flatten_results(results, qubits, "process")
    return linear_inv_process_estimate_batch(design, e)[0]


def pgdb_process_estimate_batch(design: Design, expectations, total_counts, trace_preserving=True,
                                mode="converge", max_iters=0, return_stats=False, eig_rel_tol=None,
                                trace_iters=0, out=None, devices=None, line_search="exact"):
    """Batched pgdb_process_estimate.  ``mode='converge'`` is the reference loop (optionally
    capped by ``max_iters``); ``mode='fixed'`` runs exactly ``max_iters`` outer iterations.
    ``eig_rel_tol``: the eigensolver tolerance factor for THIS call (None = the process default,
    0 = the reference's trajectory iteration by iteration; include/fbx.h fbx_pgdb_process_ex).
    ``trace_iters`` > 0 adds ``stats['trace']`` [B, trace_iters, 2]: Dykstra iterations and halvings of
    every outer iteration.  ``out``: a C-contiguous complex128 [B, D, D] array for the result; when it and both inputs
    are page-locked (``fbx._lib.pinned_empty`` / ``pinned_copy``) the library overlaps transfers and kernels.
    ``devices``: ``'all'`` or a list of GPU ordinals -- the batch is split into contiguous blocks over those GPUs inside
    this one call (``fbx._lib.set_devices``; results are bit-identical to the single-GPU call); None keeps the process's
    current device list.
    ``line_search``: ``'exact'`` (default) tests the exact cost difference of a small step; ``'reference'`` evaluates the
    full cost at every halving and uses the reference's rounded comparison (``FBX_MODE_LS_REFERENCE``, include/fbx.h) --
    identical up to the reference's own stopping point, another rounding-driven walk past it."""
    if devices is not None:
        _lib.set_devices(devices)
    if mode not in ("converge", "fixed"):
        raise ValueError("mode must be 'converge' or 'fixed'")
    if line_search not in ("exact", "reference"):
        raise ValueError("line_search must be 'exact' or 'reference'")
    e, c = _batch_arrays(design, expectations, total_counts)
    B, D = e.shape[0], design.dim ** 2
    if out is None:
        choi = np.empty((B, D, D), dtype=np.complex128)
    else:
        if out.shape != (B, D, D) or out.dtype != np.complex128 or not out.flags.c_contiguous:
            raise ValueError(f"out must be a C-contiguous complex128 array of shape {(B, D, D)}")
        choi = out
    iters = np.zeros(B, dtype=np.int32)
    dyk = np.zeros(B, dtype=np.int32)
    bt = np.zeros(B, dtype=np.int32)
    cost = np.zeros(B)
    work = np.zeros((B, 4), dtype=np.int32)
    trace = np.zeros((B, int(trace_iters), 2), dtype=np.int32) if trace_iters > 0 else None
    _lib.check(_lib.lib().fbx_pgdb_process_ex(
        design.handle, B, _lib.dptr(e), _lib.dptr(c), int(bool(trace_preserving)),
        (_lib.MODE_FIXED if mode == "fixed" else _lib.MODE_CONVERGE) | (_lib.MODE_LS_REFERENCE if line_search == "reference" else 0),
        int(max_iters), -1.0 if eig_rel_tol is None else float(eig_rel_tol),
        _lib.dptr(choi.view(np.float64)), _lib.iptr(iters), _lib.iptr(dyk), _lib.iptr(bt),
        _lib.dptr(cost), _lib.iptr(work), _lib.iptr(trace), int(trace_iters) if trace is not None else 0))
    if return_stats:
        st = {"iterations": iters, "dykstra": dyk, "backtracks": bt, "cost": cost,
              "jacobi_sweeps": work[:, 0], "eig_terms": work[:, 1], "cost_evals": work[:, 2],
              "power_sum_passes": work[:, 3]}
        if trace is not None:
            st["trace"] = trace
        return choi, st
    return choi


def pgdb_process_estimate(results: List[ExperimentResult], qubits: List[int],
                          trace_preserving=True) -> np.ndarray:
    """tomography.py:542-594 (projected gradient descent with backtracking)."""
    design, e, c = flatten_results(results, qubits, "process")
    return pgdb_process_estimate_batch(design, e, c, trace_preserving)[0]


# --------------------------------------------------------------------------------------------------
# _extract_from_results / _cost / _grad_cost (tomography.py:494-539, :597-633) as callable functions.
# The reference's A in C^{2m x D^2} is never materialised here: the design tables stand for it.
# --------------------------------------------------------------------------------------------------
class DesignMatrix:
    """What ``_extract_from_results`` returns in the place of the reference's dense ``A``: the design tables the kernels apply
    it through (``p = A vec(E)`` is ``T = R C`` plus ``2 m`` look-ups).  ``shape`` is that of the dense matrix."""

    def __init__(self, design: Design):
        self.design = design
        self.shape = (2 * design.m, design.dim ** 4)

    def __repr__(self):
        return f"DesignMatrix({self.design.n_qubits} qubits, {self.design.m} settings; dense shape {self.shape})"


def normalised_counts(expectations, total_counts) -> np.ndarray:
    """The reference's ``n`` (tomography.py:528-538) for a batch: [B, 2 m], row 2 k / 2 k + 1 = the +1 / -1 counts of result k
    over the grand total of the experiment -- the same three floating-point operations per entry as the reference's."""
    e = np.atleast_2d(np.asarray(expectations, dtype=np.float64))
    c = np.broadcast_to(np.atleast_2d(np.asarray(total_counts, dtype=np.float64)), e.shape)
    plus = (1 + e) / 2
    n = np.empty(e.shape[:1] + (2 * e.shape[1],))
    n[:, 0::2] = c * plus
    n[:, 1::2] = c * (1 - plus)
    return n / c.sum(axis=1, keepdims=True)


def _extract_from_results(results: List[ExperimentResult], qubits: List[int]) -> Tuple[DesignMatrix, np.ndarray]:
    """tomography.py:494-539: ``(A, n)`` with ``n`` the [2 m, 1] column of normalised counts exactly as the reference builds it and
    ``A`` a :class:`DesignMatrix` (the tables that apply the reference's dense matrix)."""
    design, e, c = flatten_results(results, qubits, "process")
    return DesignMatrix(design), normalised_counts(e, c)[0][:, None]


def cost_and_gradient_batch(design: Design, n, estimates, eps=1e-6, gradient=True):
    """``(_cost, _grad_cost)`` for a batch: ``n`` [B, 2 m] (:func:`normalised_counts`), ``estimates`` [B, D, D] Hermitian Choi
    matrices; one launch of ``fbx_pgdb_cost_grad`` (the reconstruction kernels' own table products).  Returns ``cost[B]`` and
    ``grad[B, D, D]`` (None when ``gradient`` is False)."""
    design = design.design if isinstance(design, DesignMatrix) else design
    D = design.dim ** 2
    est = _lib.c128(estimates).reshape(-1, D, D)
    nv = np.ascontiguousarray(np.asarray(n, dtype=np.float64).reshape(est.shape[0], -1))
    if nv.shape[1] != 2 * design.m:
        raise ValueError(f"n must hold 2 m = {2 * design.m} normalised counts per estimate")
    cost = np.empty(est.shape[0])
    grad = np.empty_like(est) if gradient else None
    _lib.check(_lib.lib().fbx_pgdb_cost_grad(design.handle, est.shape[0], _lib.dptr(nv), _lib.dptr(est.view(np.float64)), float(eps),
                                             _lib.dptr(cost), _lib.dptr(grad.view(np.float64)) if gradient else None))
    return cost, grad


def _cost(A, n, estimate, eps=1e-6):
    """tomography.py:597-614 -- returns the reference's [1, 1] array (``-n.T @ log(p)``)."""
    return cost_and_gradient_batch(A, np.asarray(n).reshape(1, -1), np.asarray(estimate)[None], eps, gradient=False)[0].reshape(1, 1)


def _grad_cost(A, n, estimate, eps=1e-6):
    """tomography.py:617-633."""
    return cost_and_gradient_batch(A, np.asarray(n).reshape(1, -1), np.asarray(estimate)[None], eps)[1][0]


def process_fidelity_variance_batch(design: Design, expectations, total_counts, target_ptm,
                                    n_resamples: int = 40, seed: int = 0, prior_counts=1,
                                    trace_preserving=True, mode="converge", max_iters=0,
                                    return_samples=False):
    """Bootstrap error bars of the process fidelity of B process tomographies (the
    process analogue of ``estimate_variance``, tomography.py:412-453): every experiment is resampled
    ``n_resamples`` times (Beta posterior of each expectation, tomography.py:378-409), all
    ``n_resamples * B`` resampled experiments are reconstructed by PGDB in ONE launch, converted to
    Pauli transfer matrices and compared with ``target_ptm`` ([D, D] or [B, D, D]).  Everything between
    the upload of (expectations, counts) and the download of the fidelities stays in HBM.
    Returns (mean[B], var[B]) (and the [n_resamples, B] fidelities with ``return_samples``)."""
    if mode not in ("converge", "fixed"):
        raise ValueError("mode must be 'converge' or 'fixed'")
    e, c = _batch_arrays(design, expectations, total_counts)
    B, m, n, D = e.shape[0], design.m, design.n_qubits, design.dim ** 2
    R = int(n_resamples)
    tgt = np.asarray(target_ptm, dtype=np.complex128)
    if tgt.shape == (D, D):
        tgt = np.broadcast_to(tgt, (B, D, D))
    if tgt.shape != (B, D, D):
        raise ValueError("target_ptm must be [D, D] or [B, D, D]")
    if R < 1 or B == 0:
        raise ValueError("need n_resamples >= 1 and a non-empty batch")
    lib, DB = _lib.lib(), _lib.DeviceBuffer
    d_e, d_c = DB.from_array(e), DB.from_array(c)
    d_er, d_cr = DB(R * B * m * 8), DB(R * B * m * 8)
    _lib.check(lib.fbx_beta_resample_dev(B * m, R, d_e.ptr, d_c.ptr, float(prior_counts),
                                         int(seed) & (2 ** 64 - 1), d_er.ptr, d_cr.ptr))
    d_choi, d_ptm = DB(R * B * D * D * 16), DB(R * B * D * D * 16)
    _lib.check(lib.fbx_pgdb_process_dev(design.handle, R * B, d_er.ptr, d_cr.ptr, int(bool(trace_preserving)),
                                        _lib.MODE_FIXED if mode == "fixed" else _lib.MODE_CONVERGE, int(max_iters),
                                        d_choi.ptr, None, None, None, None, None))
    _lib.check(lib.fbx_convert_dev(_lib.REP_CHOI, _lib.REP_PAULI_LIOUVILLE, n, R * B, d_choi.ptr, 0, d_ptm.ptr))
    d_tgt = DB.from_array(np.ascontiguousarray(np.broadcast_to(tgt, (R, B, D, D))))
    d_f = DB(R * B * 8)
    _lib.check(lib.fbx_process_fidelity_dev(n, R * B, d_tgt.ptr, d_ptm.ptr, None, d_f.ptr))
    _lib.synchronize()
    fid = d_f.to_array(np.float64, (R, B))
    for buf in (d_e, d_c, d_er, d_cr, d_choi, d_ptm, d_tgt, d_f):
        buf.free()
    if return_samples:
        return fid.mean(axis=0), fid.var(axis=0), fid
    return fid.mean(axis=0), fid.var(axis=0)


def estimate_by_qubit_groups(results, qubit_groups, kind="process", estimator="pgdb", **kwargs):
    """Tomography of several qubit groups measured in one (merged) experiment: split the results
    with ``get_results_by_qubit_groups`` (observable_estimation.py:1145-1173, the process notebook's
    per-pair loop), keep for every group the settings that belong to its tomography (observable and,
    for processes, the prepared state both inside the group), and run every set of groups that share
    a design as ONE batched call.  Returns ``{sorted group tuple: estimate}``.

    ``estimator``: 'pgdb' | 'linear_inv' for processes, 'mle' | 'linear_inv' for states; keyword
    arguments go to the batched estimator."""
    from .observable_estimation import get_results_by_qubit_groups
    if kind not in ("process", "state"):
        raise ValueError("kind must be 'process' or 'state'")
    by_group = get_results_by_qubit_groups(results, qubit_groups)
    batches = {}                                   # design key -> (design, [group], [e], [c])
    for group, res in by_group.items():
        res = [r for r in res if len(r.setting.observable.get_qubits()) > 0]
        if not res:
            raise ValueError(f"no results for qubit group {group}")
        design, e, c = flatten_results(res, list(group), kind)
        entry = batches.setdefault(design.key(), (design, [], [], []))
        entry[1].append(group); entry[2].append(e); entry[3].append(c)
    out = {}
    for design, groups, es, cs in batches.values():
        e, c = np.stack(es), np.stack(cs)
        if kind == "process":
            if estimator == "pgdb":
                est = pgdb_process_estimate_batch(design, e, c, **kwargs)
            elif estimator == "linear_inv":
                est = linear_inv_process_estimate_batch(design, e)
            else:
                raise ValueError("process estimators: 'pgdb', 'linear_inv'")
        else:
            if estimator == "mle":
                est = iterative_mle_state_estimate_batch(design, e, c, **kwargs)
            elif estimator == "linear_inv":
                est = linear_inv_state_estimate_batch(design, e)
            else:
                raise ValueError("state estimators: 'mle', 'linear_inv'")
        for g, x in zip(groups, est):
            out[g] = x
    return out
