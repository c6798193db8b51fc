"""Synthetic tomography data (host-side input generation; no estimator arithmetic).

Recipe of SURVEY.md 8d, mirroring how the reference's tests build data
(tests/test_process_tomography.py:56,87: ``haar_rand_unitary(d, rs=RandomState(52))``):
item b has truth U_b = Haar unitary from ``RandomState(1000 + b)`` (QR with phase fix,
operator_tools/random_operators.py:49-72), exact expectations e_k = tr[P_k U rho_in U^+]
(optionally depolarised), and k_+ ~ Binomial(shots, (1 + e_k)/2) from
``RandomState(2000 + b)``; ``expectation = 2 k_+/shots - 1``, ``total_counts = shots``.
"""
import itertools

import numpy as np

from .design import Design, process_design, state_design

_s2, _s3 = np.sqrt(2), np.sqrt(3)
# pyquil.simulation.matrices.STATES (pyquil==4.5.0) in the code order of include/fbx.h
STATE_VECTORS = np.array([
    [1 / _s2, 1 / _s2], [1 / _s2, -1 / _s2], [1 / _s2, 1j / _s2], [1 / _s2, -1j / _s2],
    [1, 0], [0, 1], [1, 0], [1 / _s3, _s2 / _s3],
    [1 / _s3, np.exp(-2j * np.pi / 3) * _s2 / _s3], [1 / _s3, np.exp(2j * np.pi / 3) * _s2 / _s3],
], dtype=complex)
PAULIS_1Q = np.array([[[1, 0], [0, 1]], [[0, 1], [1, 0]], [[0, -1j], [1j, 0]], [[1, 0], [0, -1]]],
                     dtype=complex)


def haar_unitary(dim, rs):
    """Haar-random unitary: Ginibre -> QR -> fix the phases of R's diagonal."""
    z = rs.randn(dim, dim) + 1j * rs.randn(dim, dim)
    q, r = np.linalg.qr(z)
    diag = np.diagonal(r)
    return q @ (np.diag(diag) / np.absolute(diag))


def product_state_matrix(codes):
    mat = np.array([[1.0 + 0j]])
    for c in codes:
        v = STATE_VECTORS[c][:, None]
        mat = np.kron(mat, v @ v.conj().T)
    return mat


def pauli_matrix(codes):
    mat = np.array([[1.0 + 0j]])
    for c in codes:
        mat = np.kron(mat, PAULIS_1Q[c])
    return mat


def exact_process_expectations(design: Design, unitaries, depolarizing=0.0):
    """e[b, k] = coef_k * tr[P_k E_b(rho_in,k)] for E_b = (1-lam) U.U^+ + lam tr(.) I/d."""
    d = design.dim
    u = np.asarray(unitaries).reshape(-1, d, d)
    keys = [tuple(r) for r in design.in_labels]
    uniq = list(dict.fromkeys(keys))
    sidx = np.array([uniq.index(k) for k in keys])
    rhos = np.array([product_state_matrix(k) for k in uniq])                  # [S, d, d]
    pkeys = [tuple(r) for r in design.paulis]
    puniq = list(dict.fromkeys(pkeys))
    pidx = np.array([puniq.index(k) for k in pkeys])
    ps = np.array([pauli_matrix(k) for k in puniq])                          # [P, d, d]
    out = np.einsum('bij,sjk,blk->bsil', u, rhos, u.conj())                   # U rho U^+
    if depolarizing:
        out = (1 - depolarizing) * out + depolarizing * np.eye(d)[None, None] / d
    t = np.real(np.einsum('pxy,bsyx->bsp', ps, out))                          # tr(P out)
    return t[:, sidx, pidx] * design.coefs[None, :]


def exact_state_expectations(design: Design, states):
    """e[b, k] = coef_k * tr[P_k rho_b]."""
    d = design.dim
    rho = np.asarray(states).reshape(-1, d, d)
    ps = np.array([pauli_matrix(k) for k in design.paulis])
    return np.real(np.einsum('kxy,byx->bk', ps, rho)) * design.coefs[None, :]


def sample_expectations(exact, shots, first_item=0, seed_base=2000):
    """Binomial sampling per item with RandomState(seed_base + item)."""
    exact = np.asarray(exact)
    e = np.empty_like(exact)
    for b in range(exact.shape[0]):
        rs = np.random.RandomState(seed_base + first_item + b)
        kp = rs.binomial(shots, np.clip((1 + exact[b]) / 2, 0, 1))
        e[b] = 2 * kp / shots - 1
    return e, np.full(exact.shape, float(shots))


def process_batch(n_qubits, in_basis="pauli", batch=1, shots=1000, first_item=0,
                  depolarizing=0.0):
    """(design, unitaries[B,d,d], expectations[B,m], counts[B,m]) for items
    first_item .. first_item + batch - 1 of the SURVEY 8d recipe."""
    design = process_design(n_qubits, in_basis)
    d = design.dim
    us = np.array([haar_unitary(d, np.random.RandomState(1000 + first_item + b))
                   for b in range(batch)])
    exact = exact_process_expectations(design, us, depolarizing)
    e, c = sample_expectations(exact, shots, first_item)
    return design, us, e, c


def state_batch(n_qubits, batch=1, shots=1000, first_item=0, mixed=0.0):
    """(design, states[B,d,d], expectations[B,m], counts[B,m]); truth = Haar pure state
    (first column of haar_unitary(d, RandomState(1000 + item))), optionally mixed with I/d."""
    design = state_design(n_qubits)
    d = design.dim
    rhos = []
    for b in range(batch):
        psi = haar_unitary(d, np.random.RandomState(1000 + first_item + b))[:, :1]
        rho = psi @ psi.conj().T
        rhos.append((1 - mixed) * rho + mixed * np.eye(d) / d)
    rhos = np.array(rhos)
    exact = exact_state_expectations(design, rhos)
    e, c = sample_expectations(exact, shots, first_item)
    return design, rhos, e, c


def kraus_batch(n_qubits, n_kraus, batch, seed=0):
    """Random CPTP Kraus sets [B, K, d, d]: G_j Ginibre, K_j = G_j S^{-1/2}, S = sum G_j^+ G_j."""
    d = 2 ** n_qubits
    rs = np.random.RandomState(seed)
    g = rs.randn(batch, n_kraus, d, d) + 1j * rs.randn(batch, n_kraus, d, d)
    s = np.einsum('bkji,bkjl->bil', g.conj(), g)
    w, v = np.linalg.eigh(s)
    s_inv_half = np.einsum('bij,bj,bkj->bik', v, 1 / np.sqrt(w), v.conj())
    return np.einsum('bkij,bjl->bkil', g, s_inv_half)
