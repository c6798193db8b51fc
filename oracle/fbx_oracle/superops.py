"""Superoperator algebra, projections, validators -- CPU restatement.

TEST INFRASTRUCTURE (see package docstring).  Citations are to
forest/benchmarking/operator_tools/<file>:<line> of the reference.
Conventions: column-stacking vec, un-normalised Choi on H_in (x) H_out, Pauli order
itertools.product('IXYZ', repeat=n).
"""
import itertools
from typing import Sequence

import numpy as np
from scipy import linalg as sla

from .design import PAULI_MATRICES


# ------------------------------------------------------------------ vec / unvec
def vec(matrix):
    """superoperator_transformations.py:33-51 -- column stacking."""
    return np.asarray(matrix).T.reshape((-1, 1))


def unvec(vector, shape=None):
    """superoperator_transformations.py:54-79."""
    vector = np.asarray(vector)
    if shape is None:
        dim = int(np.sqrt(vector.size))
        shape = dim, dim
    return vector.reshape(*shape).T


# ------------------------------------------------------------------ Pauli basis
def n_qubit_pauli_matrices(n):
    """utils.py:328-409 (PAULI_BASIS ** n): list of 4**n d x d matrices in IXYZ product order."""
    mats = []
    for codes in itertools.product(range(4), repeat=n):
        m = np.array([[1.0 + 0j]])
        for c in codes:
            m = np.kron(m, PAULI_MATRICES[c])
        mats.append(m)
    return mats


def pauli2computational_basis_matrix(dim):
    """superoperator_transformations.py:374-408: columns are vec(P_k)."""
    n = int(np.log2(dim))
    return np.hstack([vec(p) for p in n_qubit_pauli_matrices(n)]).astype(complex)


def computational2pauli_basis_matrix(dim):
    """superoperator_transformations.py:411-438."""
    return pauli2computational_basis_matrix(dim).conj().T / dim


def _as_kraus_list(kraus_ops):
    """single-ndarray-as-one-Kraus-op convenience, superoperator_transformations.py:90-92."""
    if isinstance(kraus_ops, np.ndarray):
        if len(kraus_ops[0].shape) < 2:
            kraus_ops = [kraus_ops]
    return kraus_ops


# ------------------------------------------------------------------ from Kraus
def kraus2chi(kraus_ops):
    """superoperator_transformations.py:82-97."""
    kraus_ops = _as_kraus_list(kraus_ops)
    dim = np.asarray(kraus_ops[0]).shape[0]
    c2p = computational2pauli_basis_matrix(dim)
    c_vecs = [c2p @ vec(k) for k in kraus_ops]
    return sum([c @ c.conj().T for c in c_vecs])


def kraus2superop(kraus_ops):
    """superoperator_transformations.py:100-145 (non-square Kraus allowed)."""
    kraus_ops = _as_kraus_list(kraus_ops)
    rows, cols = np.asarray(kraus_ops[0]).shape
    superop = np.zeros((rows ** 2, cols ** 2), dtype=complex)
    for op in kraus_ops:
        superop += np.kron(np.asarray(op).conj(), op)
    return superop


def kraus2pauli_liouville(kraus_ops):
    """superoperator_transformations.py:148-156."""
    return superop2pauli_liouville(kraus2superop(kraus_ops))


def kraus2choi(kraus_ops):
    """superoperator_transformations.py:159-182."""
    kraus_ops = _as_kraus_list(kraus_ops)
    return sum([vec(op) @ vec(op).conj().T for op in kraus_ops])


# ------------------------------------------------------------------ from chi
def chi2pauli_liouville(chi):
    """superoperator_transformations.py:185-192."""
    return choi2pauli_liouville(chi2choi(chi))


def chi2kraus(chi):
    """superoperator_transformations.py:195-204."""
    return pauli_liouville2kraus(chi2pauli_liouville(chi))


def chi2superop(chi):
    """superoperator_transformations.py:207-214."""
    return pauli_liouville2superop(chi2pauli_liouville(chi))


def chi2choi(chi):
    """superoperator_transformations.py:217-226."""
    dim = int(np.sqrt(np.asarray(chi).shape[0]))
    p2c = pauli2computational_basis_matrix(dim)
    return p2c @ chi @ p2c.conj().T


# ------------------------------------------------------------------ from superop
def superop2kraus(superop):
    """superoperator_transformations.py:229-238."""
    return choi2kraus(superop2choi(superop))


def superop2chi(superop):
    """superoperator_transformations.py:241-250."""
    return kraus2chi(superop2kraus(superop))


def superop2pauli_liouville(superop):
    """superoperator_transformations.py:253-264."""
    dim = int(np.sqrt(np.asarray(superop).shape[0]))
    c2p = computational2pauli_basis_matrix(dim)
    return c2p @ superop @ c2p.conj().T * dim


def superop2choi(superop):
    """superoperator_transformations.py:267-277 (involutive reshuffle)."""
    dim = int(np.sqrt(np.asarray(superop).shape[0]))
    return np.reshape(superop, [dim] * 4).swapaxes(0, 3).reshape([dim ** 2, dim ** 2])


# ------------------------------------------------------------------ from Pauli-Liouville
def pauli_liouville2kraus(pl):
    """superoperator_transformations.py:280-288."""
    return choi2kraus(pauli_liouville2choi(pl))


def pauli_liouville2chi(pl):
    """superoperator_transformations.py:291-298."""
    return kraus2chi(pauli_liouville2kraus(pl))


def pauli_liouville2superop(pl):
    """superoperator_transformations.py:301-312."""
    dim = int(np.sqrt(np.asarray(pl).shape[0]))
    p2c = pauli2computational_basis_matrix(dim)
    return p2c @ pl @ p2c.conj().T / dim


def pauli_liouville2choi(pl):
    """superoperator_transformations.py:315-322."""
    return superop2choi(pauli_liouville2superop(pl))


# ------------------------------------------------------------------ from Choi
def choi2kraus(choi, tol=1e-9):
    """superoperator_transformations.py:325-336: eigh; keep |lambda| > tol; scimath sqrt."""
    eigvals, v = np.linalg.eigh(choi)
    return [np.lib.scimath.sqrt(ev) * unvec(np.array([evec]).T)
            for ev, evec in zip(eigvals, v.T) if abs(ev) > tol]


def choi2chi(choi):
    """superoperator_transformations.py:339-348."""
    return kraus2chi(choi2kraus(choi))


def choi2superop(choi):
    """superoperator_transformations.py:351-361."""
    dim = int(np.sqrt(np.asarray(choi).shape[0]))
    return np.reshape(choi, [dim] * 4).swapaxes(0, 3).reshape([dim ** 2, dim ** 2])


def choi2pauli_liouville(choi):
    """superoperator_transformations.py:364-371."""
    return superop2pauli_liouville(choi2superop(choi))


# ------------------------------------------------------------------ calculational.py
def partial_trace(rho, keep, dims, optimize=False):
    """calculational.py:5-35."""
    keep = np.asarray(keep)
    dims = np.asarray(dims)
    ndim = dims.size
    nkeep = np.prod(dims[keep])
    idx1 = [i for i in range(ndim)]
    idx2 = [ndim + i if i in keep else i for i in range(ndim)]
    rho_a = rho.reshape(np.tile(dims, 2))
    rho_a = np.einsum(rho_a, idx1 + idx2, optimize=optimize)
    return rho_a.reshape(nkeep, nkeep)


def outer_product(bra1, bra2):
    """calculational.py:38-52."""
    rows1, cols1 = bra1.shape
    rows2, cols2 = bra2.shape
    if not (cols1 == cols2 == 1 and rows1 > 1 and rows2 > 1):
        raise ValueError("The vectors do not have the correct dimensions.")
    return np.outer(bra1, bra2.conj())


def inner_product(bra1, bra2):
    """calculational.py:55-72."""
    rows1, cols1 = bra1.shape
    rows2, cols2 = bra2.shape
    if not (cols1 == cols2 == 1 and rows1 > 1 and rows2 > 1):
        raise ValueError("The vectors do not have the correct dimensions.")
    return np.transpose(bra1.conj()) @ bra2


def sqrtm_psd(matrix, check_finite=True):
    """calculational.py:77-91."""
    w, v = sla.eigh(matrix, check_finite=check_finite)
    w = np.sqrt(np.maximum(w, 0))
    return (v * w).dot(v.conj().T)


# ------------------------------------------------------------------ project_superoperators.py
def proj_choi_to_completely_positive(choi, check_finite=True):
    """project_superoperators.py:19-34."""
    herm = (choi + choi.conj().T) / 2
    evals, v = sla.eigh(herm, check_finite=check_finite)
    evals[evals < 0] = 0
    return v @ np.diag(evals) @ v.conj().T


def proj_choi_to_trace_non_increasing(choi):
    """project_superoperators.py:37-59."""
    dim = int(np.sqrt(choi.shape[0]))
    pt = partial_trace(choi, dims=[dim, dim], keep=[0])
    herm = (pt + pt.conj().T) / 2
    d, v = sla.eigh(herm)
    d[d > 1] = 1
    projection = v @ np.diag(d) @ v.conj().T
    return choi - np.kron((pt - projection) / dim, np.eye(dim))


def proj_choi_to_trace_preserving(choi):
    """project_superoperators.py:62-84."""
    dim = int(np.sqrt(choi.shape[0]))
    pt = partial_trace(choi, dims=[dim, dim], keep=[0])
    return choi - np.kron((pt - np.eye(dim)) / dim, np.eye(dim))


def proj_choi_to_physical(choi, make_trace_preserving=True, return_iters=False):
    """project_superoperators.py:87-144 -- Dykstra with the Birgin-Raydan stopping rule
    (< 1e-4), no iteration cap, returns the last TP/TNI iterate."""
    old_cp_change = np.zeros_like(choi)
    old_tp_change = np.zeros_like(choi)
    last_cp_projection = np.zeros_like(choi)
    last_state = choi
    iters = 0
    while True:
        iters += 1
        pre_cp = last_state - old_cp_change
        cp_projection = proj_choi_to_completely_positive(pre_cp)
        new_cp_change = cp_projection - pre_cp

        pre_tp = cp_projection - old_tp_change
        if make_trace_preserving:
            new_state = proj_choi_to_trace_preserving(pre_tp)
        else:
            new_state = proj_choi_to_trace_non_increasing(pre_tp)
        new_tp_change = new_state - pre_tp

        cp_cc = new_cp_change - old_cp_change
        tp_cc = new_tp_change - old_tp_change
        state_change = new_state - last_state
        if np.linalg.norm(cp_cc) ** 2 + np.linalg.norm(tp_cc) ** 2 \
                + 2 * abs(np.dot(vec(old_tp_change).conj().T, vec(state_change))) \
                + 2 * abs(np.dot(vec(old_cp_change).conj().T,
                                 vec(cp_projection - last_cp_projection))) < 1e-4:
            break
        old_cp_change = new_cp_change
        old_tp_change = new_tp_change
        last_cp_projection = cp_projection
        last_state = new_state
    if return_iters:
        return new_state, iters
    return new_state


def proj_choi_to_unitary(choi, check_finite=True):
    """project_superoperators.py:147-175."""
    dim = int(np.sqrt(choi.shape[0]))
    herm = (choi + choi.conj().T) / 2
    vals, vs = sla.eigh(herm, check_finite=check_finite)
    kraus = unvec(vs[:, np.argmax(vals)].reshape((dim * dim, 1)))
    u, _, v = sla.svd(kraus)
    unitary = u @ v
    phase = np.angle(unitary[0, 0])
    return kraus2choi(np.exp(-1j * phase) * unitary)


# ------------------------------------------------------------------ project_state_matrix.py
def project_state_matrix_to_physical(rho):
    """project_state_matrix.py:6-52 (Smolin-Gambetta-Smith)."""
    rho_impure = rho / np.trace(rho)
    dimension = rho_impure.shape[0]
    eigvals, eigvecs = sla.eigh(rho_impure)
    if np.min(eigvals) >= 0:
        return rho_impure
    eigvals = list(eigvals)
    eigvals.reverse()
    eigvals_new = [0.0] * len(eigvals)
    i = dimension
    accumulator = 0.0
    while eigvals[i - 1] + accumulator / float(i) < 0:
        accumulator += eigvals[i - 1]
        i -= 1
    for j in range(i):
        eigvals_new[j] = eigvals[j] + accumulator / float(i)
    eigvals_new.reverse()
    return eigvecs @ np.diag(eigvals_new) @ np.conj(eigvecs.T)


# ------------------------------------------------------------------ apply_superoperator.py
def apply_kraus_ops_2_state(kraus_ops, state):
    """apply_superoperator.py:33-57 (real-typed accumulator at :53 reproduced)."""
    kraus_ops = _as_kraus_list(kraus_ops)
    dim, _ = state.shape
    rows, cols = kraus_ops[0].shape
    if dim != cols:
        raise ValueError("Dimensions of state and Kraus operator are incompatible")
    new_state = np.zeros((rows, rows))
    for m in kraus_ops:
        new_state += m @ state @ np.transpose(m.conj())
    return new_state


def apply_choi_matrix_2_state(choi, state):
    """apply_superoperator.py:60-90."""
    dim = int(np.sqrt(np.asarray(choi).shape[0]))
    tot = np.kron(state.transpose(), np.identity(dim)) @ choi
    return partial_trace(tot, [1], [dim, dim])


# ------------------------------------------------------------------ compose / twirl
def tensor_channel_kraus(k2: Sequence[np.ndarray], k1: Sequence[np.ndarray]):
    """compose_superoperators.py:7-24."""
    return [np.kron(k2l, k1j) for k1j in k1 for k2l in k2]


def compose_channel_kraus(k2: Sequence[np.ndarray], k1: Sequence[np.ndarray]):
    """compose_superoperators.py:27-44."""
    return [np.dot(k2l, k1j) for k1j in k1 for k2l in k2]


def pauli_twirl_chi_matrix(chi):
    """channel_approximation.py:31-49."""
    return np.diag(chi.diagonal())


# ------------------------------------------------------------------ validate_operator.py
def is_square_matrix(matrix):
    """validate_operator.py:6-18."""
    if len(matrix.shape) != 2:
        raise ValueError("The object is not a matrix.")
    rows, cols = matrix.shape
    return rows == cols


def _need_square(matrix):
    if not is_square_matrix(matrix):
        raise ValueError("The matrix is not square.")


def is_symmetric_matrix(matrix, rtol=1e-05, atol=1e-08):
    """validate_operator.py:21-33."""
    _need_square(matrix)
    return np.allclose(matrix, matrix.T, rtol=rtol, atol=atol)


def is_identity_matrix(matrix, rtol=1e-05, atol=1e-08):
    """validate_operator.py:36-49."""
    _need_square(matrix)
    return np.allclose(matrix, np.eye(len(matrix)), rtol=rtol, atol=atol)


def is_idempotent_matrix(matrix, rtol=1e-05, atol=1e-08):
    """validate_operator.py:52-64."""
    _need_square(matrix)
    return np.allclose(matrix, matrix @ matrix, rtol=rtol, atol=atol)


def is_normal_matrix(matrix, rtol=1e-05, atol=1e-08):
    """validate_operator.py:67-81."""
    _need_square(matrix)
    return np.allclose(matrix.T.conj() @ matrix, matrix @ matrix.T.conj(), rtol=rtol, atol=atol)


def is_hermitian_matrix(matrix, rtol=1e-05, atol=1e-08):
    """validate_operator.py:84-96."""
    _need_square(matrix)
    return np.allclose(matrix, matrix.T.conj(), rtol=rtol, atol=atol)


def is_unitary_matrix(matrix, rtol=1e-05, atol=1e-08):
    """validate_operator.py:99-115."""
    _need_square(matrix)
    ab = matrix.T.conj() @ matrix
    ba = matrix @ matrix.T.conj()
    eye = np.eye(len(matrix))
    return np.allclose(ab, eye, rtol=rtol, atol=atol) and np.allclose(ba, eye, rtol=rtol, atol=atol)


def is_positive_definite_matrix(matrix, rtol=1e-05, atol=1e-08):
    """validate_operator.py:118-133."""
    if not is_hermitian_matrix(matrix, rtol, atol):
        raise ValueError("The matrix is not Hermitian.")
    evals, _ = np.linalg.eigh(matrix)
    return all(x > -abs(atol) for x in evals)


def is_positive_semidefinite_matrix(matrix, rtol=1e-05, atol=1e-08):
    """validate_operator.py:136-150."""
    if not is_hermitian_matrix(matrix, rtol, atol):
        raise ValueError("The matrix is not Hermitian.")
    evals, _ = np.linalg.eigh(matrix)
    return all(x >= -abs(atol) for x in evals)


# ------------------------------------------------------------------ validate_superoperator.py
def kraus_operators_are_valid(kraus_ops, rtol=1e-05, atol=1e-08):
    """validate_superoperator.py:40-62."""
    kraus_ops = _as_kraus_list(kraus_ops)
    povm = [np.transpose(op).conjugate().dot(op) for op in kraus_ops]
    all_psd = all(is_positive_semidefinite_matrix(e) for e in povm)
    return all_psd and is_identity_matrix(sum(povm), rtol, atol)


def choi_is_hermitian_preserving(choi, rtol=1e-05, atol=1e-08):
    """validate_superoperator.py:65-77."""
    return is_hermitian_matrix(choi, rtol, atol)


def choi_is_trace_preserving(choi, rtol=1e-05, atol=1e-08):
    """validate_superoperator.py:80-97."""
    dim = int(np.sqrt(choi.shape[0]))
    return is_identity_matrix(partial_trace(choi, [0], [dim, dim]), rtol, atol)


def choi_is_completely_positive(choi, rtol=1e-05, atol=1e-08):
    """validate_superoperator.py:100-112."""
    return is_positive_semidefinite_matrix(choi, rtol, atol)


def choi_is_cptp(choi, rtol=1e-05, atol=1e-08):
    """validate_superoperator.py:115-127."""
    tp = choi_is_trace_preserving(choi, rtol, atol)
    cp = choi_is_completely_positive(choi, rtol, atol)
    return cp and tp


def choi_is_unital(choi, rtol=1e-05, atol=1e-08):
    """validate_superoperator.py:130-145."""
    dim = int(np.sqrt(choi.shape[0]))
    return is_identity_matrix(apply_choi_matrix_2_state(choi, np.identity(dim)), rtol, atol)


def choi_is_unitary(choi, limit=1e-09):
    """validate_superoperator.py:148-157."""
    return len(choi2kraus(choi, tol=limit)) == 1


# ------------------------------------------------------------------ random_operators.py
def ginibre_matrix_complex(dim, k, rs=None):
    """random_operators.py:21-46."""
    if rs is None:
        rs = np.random
    return rs.randn(dim, k) + 1j * rs.randn(dim, k)


def haar_rand_unitary(dim, rs=None):
    """random_operators.py:49-72 (QR + phase fix)."""
    if rs is None:
        rs = np.random
    z = ginibre_matrix_complex(dim=dim, k=dim, rs=rs)
    q, r = np.linalg.qr(z)
    diag = np.diagonal(r)
    return np.matmul(q, np.diag(diag) / np.absolute(diag))
