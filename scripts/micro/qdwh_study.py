"""CPU study for the round-4 review's question (iii): would a QDWH polar iteration on the matrix cores beat the warm-started
Jacobi for the CP projection X+ = (X + |X|)/2, |X| = U^H X with U the unitary polar factor of the Hermitian X?

Replays the Dykstra inputs of the oracle's PGDB run on a few bench items (the matrices proj_choi_to_completely_positive sees,
operator_tools/project_superoperators.py:19-34), runs the dynamically weighted Halley iteration of Nakatsukasa-Bai-Gygi (QR-based
form, weights from the running lower bound l of sigma_min / sigma_max) until ||U_k - U_{k-1}||_F < 1e-13, and reports the
iteration counts and the deviation of (X + U X)/2 ... from the eigh-based projection.

    python scripts/micro/qdwh_study.py [items]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "forest-benchmarking_amd"))
from fbx import synthetic                                   # noqa: E402
from fbx_oracle import design as od, estimators as oe, superops as so   # noqa: E402


def qdwh_polar(x, tol=1e-13, max_it=30):
    """Unitary polar factor of a square matrix; returns (U, iterations, QR-type iterations)."""
    n = x.shape[0]
    alpha = np.linalg.norm(x, 2)
    u = x / alpha
    smin = np.linalg.svd(u, compute_uv=False)[-1]
    l = max(smin, 1e-17)                                     # (an estimate is enough in practice; exact here)
    it = qr_its = 0
    while it < max_it:
        l = min(l, 1.0)
        l2 = l * l
        dd = (4 * (1 - l2) / (l2 * l2)) ** (1 / 3)
        sq = np.sqrt(1 + dd)
        a = sq + 0.5 * np.sqrt(8 - 4 * dd + 8 * (2 - l2) / (l2 * sq))
        b = (a - 1) ** 2 / 4
        c = a + b - 1
        if c > 100:                                          # ill-conditioned: QR-based step
            q, _ = np.linalg.qr(np.vstack([np.sqrt(c) * u, np.eye(n)]))
            un = (b / c) * u + (a - b / c) / np.sqrt(c) * q[:n] @ q[n:].conj().T
            qr_its += 1
        else:                                                # Cholesky-based step
            z = np.eye(n) + c * u.conj().T @ u
            w = np.linalg.cholesky(z)
            un = (b / c) * u + (a - b / c) * np.linalg.solve(w.conj().T, np.linalg.solve(w, u.conj().T)).conj().T
        it += 1
        done = np.linalg.norm(un - u) < tol
        u = un
        l = l * (a + b * l2) / (1 + c * l2)
        if done:
            break
    return u, it, qr_its


def main():
    items = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    design, _, e, c = synthetic.process_batch(2, "pauli", items)
    d = od.Design(design.n_qubits, design.kind, design.in_labels, design.paulis, design.coefs)
    A = oe.design_matrix_A(d)
    seen = []
    orig = so.proj_choi_to_completely_positive

    def spy(choi):
        seen.append(np.array(choi))
        return orig(choi)

    so.proj_choi_to_completely_positive = spy
    try:
        for b in range(items):
            oe.pgdb_process_estimate(d, e[b], c[b], A=A, mode="fixed", max_iters=40)
    finally:
        so.proj_choi_to_completely_positive = orig
    its, qrs, dev, cond = [], [], [], []
    for x in seen[:: max(1, len(seen) // 400)]:
        h = (x + x.conj().T) / 2
        lam, v = np.linalg.eigh(h)
        want = (v * np.maximum(lam, 0)) @ v.conj().T
        u, it, qr = qdwh_polar(h)
        got = (h + u.conj().T @ h) / 2                       # |X| = U^H X for Hermitian X
        got = (got + got.conj().T) / 2
        its.append(it); qrs.append(qr); dev.append(np.abs(got - want).max())
        cond.append(np.abs(lam).max() / max(np.abs(lam).min(), 1e-300))
    its, qrs = np.array(its), np.array(qrs)
    print(f"{len(seen)} CP projections recorded over {items} reconstructions x 40 outer iterations; {len(its)} sampled")
    print(f"QDWH iterations: mean {its.mean():.2f}, max {its.max()}; QR-type among them: mean {qrs.mean():.2f}; "
          f"condition numbers: median {np.median(cond):.1e}, max {max(cond):.1e}")
    print(f"projection deviation from the eigh form: max {max(dev):.1e}")


if __name__ == "__main__":
    main()
