"""CPU study (numpy, test infrastructure only): how many Jacobi sweeps would the warm-started CP projections of
PGDB need if the stored basis were first corrected to first order on the matrix cores?

Runs the oracle's PGDB on a few bench items, records every matrix that goes into proj_choi_to_completely_positive
together with its position in the Dykstra run, and replays the kernel's warm-start policy (basis of the same slot
of the previous outer iteration once the outer step is below 1e-3, else the previous slot of the same run):
    plain     : sweeps of a cyclic (round-robin) Jacobi from V'HV until off^2 <= 1e-26 ||H||^2
    refined   : E_ij = M_ij / (l_j - l_i) where |M_ij| <= kappa |l_j - l_i| (else 0), W = exp(E) to third order,
                V <- V W, then Jacobi as above
usage: python scripts/micro/refine_study.py [items] [kappa]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "forest-benchmarking_amd"))
import numpy as np
from fbx_oracle import estimators, superops
from fbx_oracle.design import process_design
from fbx import synthetic

N = 16
TOL2 = 1e-26


def pairs_of_round(r, n=N):
    idx = [0] + [1 + (k + r) % (n - 1) for k in range(n - 1)]
    return [(idx[k], idx[n - 1 - k]) for k in range(n // 2)]


def jacobi_sweeps(M, V, max_sweeps=40):
    """Parallel-order cyclic Jacobi on Hermitian M (already in the basis V); returns sweeps, M, V."""
    M = M.copy(); V = V.copy()
    n2 = np.linalg.norm(M) ** 2
    for sweep in range(max_sweeps + 1):
        off2 = n2 - np.sum(np.abs(np.diag(M)) ** 2)
        off2 = np.sum(np.abs(M - np.diag(np.diag(M))) ** 2)
        if not off2 > TOL2 * n2:
            return sweep, M, V
        for r in range(N - 1):
            J = np.eye(N, dtype=complex)
            for p, q in pairs_of_round(r):
                a, d, b = M[p, p].real, M[q, q].real, M[p, q]
                ab = abs(b)
                if ab == 0.0:
                    continue
                tau = (d - a) / (2 * ab)
                t = np.sign(tau) / (abs(tau) + np.hypot(1.0, tau)) if tau != 0 else 1.0
                c = 1 / np.hypot(1.0, t); s = t * c
                ph = b / ab
                J[p, p] = c; J[q, q] = c; J[p, q] = s * ph; J[q, p] = -s * np.conj(ph)
            M = J.conj().T @ M @ J
            V = V @ J
    return max_sweeps, M, V


def refine(H, V, kappa):
    M = V.conj().T @ H @ V
    lam = np.diag(M).real
    gap = lam[None, :] - lam[:, None]
    ok = np.abs(M) <= kappa * np.abs(gap)
    np.fill_diagonal(ok, False)
    E = np.where(ok, M / np.where(gap == 0, 1, gap), 0)
    E = (E - E.conj().T) / 2          # anti-Hermitian by construction where both (i,j), (j,i) are ok
    W = np.eye(N) + E @ (np.eye(N) + E @ (np.eye(N) + E / 3) / 2)
    return V @ W, ok.sum() / (N * N - N)


VARIANTS = ("plain", "same-slot", "best-of-two")


def off_of(H, V):
    M = V.conj().T @ H @ V
    return np.sum(np.abs(M - np.diag(np.diag(M))) ** 2)


def main():
    items = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    kappa = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
    design, us, e, c = synthetic.process_batch(2, "pauli", max(items, 4))
    odesign = process_design(2, "pauli")
    A = estimators.design_matrix_A(odesign)
    tot = dict(plain=0, refined=0, refine_calls=0, decomp=0, frozen=0, unit=0.0)
    hists, byit = {}, {}
    for b in range(items):
        log = []                     # (outer iteration, slot, H)
        state = dict(it=-1, slot=0)
        real_cp = superops.proj_choi_to_completely_positive
        real_phys = superops.proj_choi_to_physical

        def cp(choi, check_finite=True):
            log.append((state["it"], state["slot"], (choi + choi.conj().T) / 2))
            state["slot"] += 1
            return real_cp(choi, check_finite)

        def phys(*a, **k):
            state["it"] += 1; state["slot"] = 0
            return real_phys(*a, **k)

        superops.proj_choi_to_completely_positive = cp
        estimators.proj_choi_to_physical = phys
        try:
            est, st = estimators.pgdb_process_estimate(odesign, e[b], c[b], mode="fixed", max_iters=100, A=A, return_stats=True)
        finally:
            superops.proj_choi_to_completely_positive = real_cp
            estimators.proj_choi_to_physical = real_phys
        # replay
        for variant in VARIANTS:
            store = {}
            prev_first = None
            for it, slot, H in log:
                if slot == 0:
                    step = np.inf if prev_first is None else np.linalg.norm(H - prev_first)
                    prev_first = H
                    use_prev = step < 1e-3
                cands = []
                if slot in store:
                    cands.append(store[slot])
                if slot > 0 and (slot - 1) in store:
                    cands.append(store[slot - 1])
                if variant == "same-slot":
                    V = cands[0] if cands else None
                elif variant == "best-of-two":
                    V = min(cands, key=lambda X: off_of(H, X)) if cands else None
                elif use_prev and slot in store:
                    V = store[slot]
                elif slot > 0 and (slot - 1) in store:
                    V = store[slot - 1]
                elif slot == 0 and 0 in store:
                    V = store[0]
                else:
                    V = None
                if V is None:
                    sw, M, V2 = jacobi_sweeps(H, np.eye(N, dtype=complex))
                else:
                    M0 = V.conj().T @ H @ V
                    n2 = np.linalg.norm(M0) ** 2
                    off2 = np.sum(np.abs(M0 - np.diag(np.diag(M0))) ** 2)
                    if variant == "refined" and off2 > TOL2 * n2 and off2 < 1e-4 * n2:
                        V, frac = refine(H, V, kappa)
                        tot["refine_calls"] += 1
                        M0 = V.conj().T @ H @ V
                    sw, M, V2 = jacobi_sweeps(M0, V)
                store[slot] = V2
                tot[variant] = tot.get(variant, 0) + sw
                hists.setdefault(variant, np.zeros(12, int))[min(sw, 11)] += 1
                byit.setdefault(variant, np.zeros(100))[it] += sw
                if variant == "plain":
                    tot["decomp"] += 1
                if variant == "refined":
                    tot["unit"] = max(tot["unit"], np.linalg.norm(V2.conj().T @ V2 - np.eye(N)))
        print(f"item {b}: decompositions {len(log)} dykstra {st['dykstra']}", flush=True)
    print(f"items {items} kappa {kappa}: decompositions {tot['decomp']}  refinements {tot['refine_calls']}  max ||V'V - 1|| {tot['unit']:.1e}")
    for v in VARIANTS:
        print(f"{v:12s} sweeps {tot[v]:6d}  histogram {hists[v]}")
        print("   sweeps per outer iteration (x10 iterations):", byit[v].reshape(10, 10).sum(1).astype(int))


if __name__ == "__main__":
    main()
