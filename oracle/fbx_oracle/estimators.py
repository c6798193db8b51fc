"""Tomography estimators -- CPU restatement of forest/benchmarking/tomography.py:130-633.

TEST INFRASTRUCTURE (see package docstring).  Inputs are the SoA form of a list of
ExperimentResults: a :class:`fbx_oracle.design.Design` plus ``expectations[m]`` and
``total_counts[m]`` (use :func:`fbx_oracle.design.flatten_results` for result objects).

Reference quirks kept on purpose: complex ``p`` with lexicographic
``np.clip`` / comparisons in PGDB, ``maxiter-1`` updates in the iterative MLE, ``_R`` with a
complex predicted expectation, ``log10`` likelihood, un-projected PGDB output.

Extension next to the faithful mode: ``pgdb_process_estimate(mode='fixed', max_iters=K)``
runs exactly K outer iterations (the benchmark mode, "100 iters"); ``mode='converge'`` is
the reference's loop (optionally capped by ``max_iters``).
"""
import warnings

import numpy as np
from scipy.linalg import logm, pinv

from . import measures as dm
from .design import Design, pauli_matrix, state_matrix
from .superops import (vec, unvec, proj_choi_to_physical, project_state_matrix_to_physical)


# ---------------------------------------------------------------------------- states
def linear_inv_state_estimate(design: Design, expectations) -> np.ndarray:
    """tomography.py:130-165."""
    meas = np.vstack([vec(pauli_matrix(design.paulis[k], design.coefs[k])).T.conj()
                      for k in range(design.m)])
    rho = pinv(meas) @ np.asarray(expectations)
    dim = design.dim
    return unvec(rho) + np.eye(dim) / dim


def R_operator(state, design: Design, expectations, ops=None) -> np.ndarray:
    """tomography.py:273-338 (``_R``)."""
    tiny = np.finfo(float).tiny
    update = np.zeros_like(state, dtype=complex)
    eye = np.eye(update.shape[0])
    for k in range(design.m):
        op = pauli_matrix(design.paulis[k], design.coefs[k]) if ops is None else ops[k]
        meas_exp = expectations[k]
        pred_exp = np.trace(op @ state)
        for sign in [1, -1]:
            f_j_over_n = (1 + sign * meas_exp) / 2
            pr_j = (1 + sign * pred_exp) / 2
            pi_j = (eye + sign * op) / 2
            update += f_j_over_n / (pr_j + tiny) * pi_j
    return update / design.m


def iterative_mle_state_estimate(design: Design, expectations, total_counts, epsilon=.1,
                                 entropy_penalty=0.0, beta=0.0, tol=1e-9, maxiter=10_000,
                                 return_stats=False):
    """tomography.py:168-270.  Performs at most ``maxiter - 1`` updates (``:241-246``)."""
    if (entropy_penalty != 0.0) and (beta != 0.0):
        raise ValueError("One can't sensibly do entropy penalty and hedging. Do one or the other"
                         " but not both.")
    dim = design.dim
    eye = np.eye(dim, dim)
    num_meas = sum(total_counts)
    ops = [pauli_matrix(design.paulis[k], design.coefs[k]) for k in range(design.m)]
    rho = eye / dim
    iteration = 1
    hit_max = False
    while True:
        rho_temp = rho
        if iteration >= maxiter:
            warnings.warn('Maximum number of iterations reached before convergence.')
            hit_max = True
            break
        Tk = R_operator(rho, design, expectations, ops) - eye
        if entropy_penalty > 0.0:
            constraint = logm(rho) - eye * np.trace(rho @ logm(rho))
            Tk -= entropy_penalty * constraint
        if beta > 0.0:
            Tk *= num_meas / 2
            Tk += beta * (pinv(rho) - dim * eye) / 2
        update_map = (eye + epsilon * Tk)
        rho = update_map @ rho @ update_map
        rho /= np.trace(rho)
        if np.linalg.norm(rho - rho_temp, 'fro') < tol:
            break
        iteration += 1
    if return_stats:
        return rho, {"iterations": iteration, "hit_max": hit_max}
    return rho


def state_log_likelihood(state, design: Design, expectations, total_counts) -> float:
    """tomography.py:341-375 (log10, terms with pr <= 0 skipped)."""
    ll = 0
    for k in range(design.m):
        n = total_counts[k]
        op = pauli_matrix(design.paulis[k], design.coefs[k])
        pred_exp = np.real(np.trace(op @ state))
        for sign in [1, -1]:
            f_j = n * (1 + sign * expectations[k]) / 2
            pr_j = (1 + sign * pred_exp) / 2
            if pr_j <= 0:
                continue
            ll += f_j * np.log10(pr_j)
    return ll


def resample_expectations_with_beta(expectations, total_counts, prior_counts=1, rs=None):
    """tomography.py:378-409.  ``rs=None`` draws from the global np.random stream like the
    reference (one scalar draw per result, in order)."""
    rs = np.random if rs is None else rs
    out = np.empty(len(expectations))
    for k in range(len(expectations)):
        num_plus = ((expectations[k] + 1) / 2) * total_counts[k]
        num_minus = total_counts[k] - num_plus
        out[k] = 2 * rs.beta(num_plus + prior_counts, num_minus + prior_counts) - 1
    return out


def estimate_variance(design, expectations, total_counts, tomo_estimator, functional,
                      target_state=None, n_resamples=40, project_to_physical=False, rs=None):
    """tomography.py:412-453.  ``tomo_estimator(design, e, counts) -> rho``."""
    if functional != dm.purity:
        if target_state is None:
            raise ValueError("You're not using the `purity` functional. "
                             "Please specify a target state.")
    samples = []
    for _ in range(n_resamples):
        e = resample_expectations_with_beta(expectations, total_counts, rs=rs)
        rho = tomo_estimator(design, e, total_counts)
        if project_to_physical:
            rho = project_state_matrix_to_physical(rho)
        if functional == dm.purity:
            samples.append(np.real(dm.purity(rho, dim_renorm=False)))
        else:
            samples.append(np.real(functional(target_state, rho)))
    return np.mean(samples), np.var(samples)


# ---------------------------------------------------------------------------- processes
def linear_inv_process_estimate(design: Design, expectations) -> np.ndarray:
    """tomography.py:459-491."""
    meas = np.vstack([
        vec(np.kron(state_matrix(design.in_labels[k]).conj(),
                    pauli_matrix(design.paulis[k], design.coefs[k]))).conj().T
        for k in range(design.m)])
    rho = pinv(meas) @ np.asarray(expectations)
    dim = design.dim
    return unvec(rho) + np.eye(dim ** 2) / dim


def design_matrix_A(design: Design, sparse: bool = False):
    """The data-independent half of tomography.py:494-539: A in C^{2m x D^2}.

    ``sparse=True`` returns the same matrix as scipy CSR (built in row chunks) -- the dense form of
    the 3-qubit Pauli design is 27 216 x 4096 complex (1.8 GB), 24 % of it non-zero."""
    dim = design.dim
    eye = np.eye(dim)

    def rows_of(k):
        rho_in = state_matrix(design.in_labels[k])
        op = pauli_matrix(design.paulis[k], design.coefs[k])
        return (vec(np.kron(rho_in, ((eye + op) / 2).T)).T[0], vec(np.kron(rho_in, ((eye - op) / 2).T)).T[0])

    if not sparse:
        rows = []
        for k in range(design.m):
            rows.extend(rows_of(k))
        return np.asarray(rows) / dim ** 2
    import scipy.sparse as sp
    chunks, chunk = [], []
    for k in range(design.m):
        chunk.extend(rows_of(k))
        if len(chunk) >= 1024 or k == design.m - 1:
            chunks.append(sp.csr_matrix(np.asarray(chunk) / dim ** 2))
            chunk = []
    return sp.vstack(chunks, format="csr")


def counts_vector(expectations, total_counts) -> np.ndarray:
    """The data-dependent half of tomography.py:528-538: n in R^{2m x 1}."""
    e = np.asarray(expectations, dtype=float)
    c = np.asarray(total_counts, dtype=float)
    plus = (1 + e) / 2
    n = np.empty(2 * len(e))
    n[0::2] = c * plus
    n[1::2] = c * (1 - plus)
    grand_total = 0
    for x in total_counts:       # sequential sum like the reference's += loop
        grand_total += x
    return n[:, None] / grand_total


def cost(A, n, estimate, eps=1e-6):
    """tomography.py:597-614 -- complex (1,1) array."""
    p = A @ vec(estimate)
    p = np.clip(p, a_min=eps, a_max=None)
    return - n.T @ np.log(p)


def grad_cost(A, n, estimate, eps=1e-6):
    """tomography.py:617-633."""
    p = A @ vec(estimate)
    p = np.clip(p, a_min=eps, a_max=None)
    eta = n / p
    return unvec(-A.conj().T @ eta)


def pgdb_process_estimate(design: Design, expectations, total_counts, trace_preserving=True,
                          mode="converge", max_iters=0, A=None, return_stats=False):
    """tomography.py:542-594 (projected gradient descent with backtracking).

    mode='converge': the reference loop (``max_iters`` > 0 adds a cap, an extension).
    mode='fixed':    exactly ``max_iters`` outer iterations, no convergence test.
    Stats: outer iterations, total Dykstra (= eigh) iterations, total backtracking halvings,
    final cost (real part), and ``trace``: [(Dykstra iterations, halvings)] of every outer iteration."""
    if A is None:
        A = design_matrix_A(design)
    n = counts_vector(expectations, total_counts)
    dim = design.dim
    est = np.eye(dim ** 2, dim ** 2, dtype=complex) / dim
    old_cost = cost(A, n, est)
    mu = 3 / (2 * dim ** 2)
    gamma = .3
    iters = dykstra = backtracks = 0
    trace = []
    new_cost = old_cost
    while True:
        if mode == "fixed" and iters >= max_iters:
            break
        gradient = grad_cost(A, n, est)
        proj, d_it = proj_choi_to_physical(est - gradient / mu, trace_preserving, return_iters=True)
        dykstra += d_it
        update = proj - est
        alpha = 1
        new_cost = cost(A, n, est + alpha * update)
        change = gamma * alpha * np.dot(vec(update).conj().T, vec(gradient))
        bt0 = backtracks
        while new_cost > old_cost + change:
            alpha = .5 * alpha
            change = .5 * change
            new_cost = cost(A, n, est + alpha * update)
            backtracks += 1
            if alpha < 1e-15:
                break
        trace.append((d_it, backtracks - bt0))
        est += alpha * update
        iters += 1
        if mode == "converge":
            if old_cost - new_cost < 1e-10:
                break
            if max_iters and iters >= max_iters:
                break
        old_cost = new_cost
    if return_stats:
        return est, {"iterations": iters, "dykstra": dykstra, "backtracks": backtracks,
                     "cost": float(np.real(new_cost).ravel()[0]), "trace": trace}
    return est
