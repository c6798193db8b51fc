"""Two-qubit state tomography from raw bitstrings: shots -> observable moments -> linear inversion / iterative MLE (plain,
maximum-entropy, hedged) -> physical projection -> distance measures -> bootstrap variance of a functional.

    python examples/state_tomography_from_shots.py               (needs libfbx.so and an MI355X)

The quantum computer is replaced by sampling: for every one of the 15 settings the prepared state is measured in the
product eigenbasis of the setting's Pauli observable, `shots` times.  From `bitarrays` on, every number comes from libfbx."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "forest-benchmarking_amd"))
from fbx import distance_measures as dm, tomography                                            # noqa: E402
from fbx.design import state_design                                                            # noqa: E402
from fbx.observable_estimation import ExperimentResult, shots_to_obs_moments_batch             # noqa: E402
from fbx.operator_tools.project_state_matrix import project_state_matrix_to_physical           # noqa: E402

H = np.array([[1, 1], [1, -1]]) / np.sqrt(2)
TO_Z = {0: np.eye(2), 3: np.eye(2), 1: H, 2: H @ np.diag([1, -1j])}     # basis change that maps the Pauli's eigenbasis to Z


def measure(rho, pauli_codes, shots, rs):
    """`shots` bitstrings [shots, n] of rho measured in the eigenbasis of the Pauli word (codes 0:I 1:X 2:Y 3:Z)."""
    u = np.array([[1.0]])
    for code in pauli_codes:
        u = np.kron(u, TO_Z[int(code)])
    p = np.real(np.diag(u @ rho @ u.conj().T)).clip(0)
    outcomes = rs.choice(len(p), size=shots, p=p / p.sum())
    n = len(pauli_codes)
    return ((outcomes[:, None] >> np.arange(n - 1, -1, -1)) & 1).astype(np.uint8)


def main(shots=4000, verbose=True):
    say = print if verbose else (lambda *a, **k: None)
    qubits = [0, 1]
    rs = np.random.RandomState(5)
    bell = np.zeros((4, 4), dtype=complex); bell[np.ix_([0, 3], [0, 3])] = 0.5
    rho_true = 0.92 * bell + 0.08 * np.eye(4) / 4

    settings = tomography.generate_state_tomography_settings(qubits)             # the 15 non-identity Pauli words
    design = state_design(2)
    bitarrays = np.array([measure(rho_true, codes, shots, rs) for codes in design.paulis])           # [15, shots, 2]

    # shots -> moments: mean and variance of the mean of the +-1 products, all settings in one call
    mean, var = shots_to_obs_moments_batch(bitarrays, design.paulis != 0)
    results = [ExperimentResult(setting=s, expectation=float(m), std_err=float(np.sqrt(v)), total_counts=shots)
               for s, m, v in zip(settings, mean, var)]
    say("first results:", *[str(r) for r in results[:2]], sep="\n  ")

    # estimators, reference signatures
    rho_lin = tomography.linear_inv_state_estimate(results, qubits)
    # (tolerances and the hedging parameters are the ones the reference's own tests use, tests/test_state_tomography.py:150-229)
    rho_mle = tomography.iterative_mle_state_estimate(results, qubits, tol=1e-4)
    rho_maxent = tomography.iterative_mle_state_estimate(results, qubits, epsilon=.1, entropy_penalty=.005, tol=1e-4)
    rho_hedged = tomography.iterative_mle_state_estimate(results, qubits, epsilon=.0001, beta=.5, tol=1e-3)
    rho_phys = project_state_matrix_to_physical(rho_lin)
    out = {}
    for name, est in (("linear inversion", rho_lin), ("... projected to physical", rho_phys), ("MLE", rho_mle),
                      ("max-entropy MLE", rho_maxent), ("hedged MLE", rho_hedged)):
        out[name] = (dm.fidelity(rho_true, est) if name != "linear inversion" else float("nan"), dm.trace_distance(rho_true, est),
                     dm.purity(est))
        say(f"{name:28s} fidelity {out[name][0]:.4f}  trace distance {out[name][1]:.4f}  purity {out[name][2]:.4f}")
    say("log-likelihood (log10) of the MLE:", tomography.state_log_likelihood(rho_mle, results, qubits))

    # bootstrap variance of a functional of the estimate (tomography.py:378-453): all resamples are one device batch
    from functools import partial
    mean_p, var_p = tomography.estimate_variance(results, qubits, partial(tomography.iterative_mle_state_estimate, tol=1e-4),
                                                 dm.purity, n_resamples=40, project_to_physical=True)
    out["bootstrap purity"] = (float(mean_p), float(np.sqrt(var_p)))
    say(f"purity of the MLE estimate: {mean_p:.4f} +- {np.sqrt(var_p):.4f} (true state: {dm.purity(rho_true):.4f})")
    out["true purity"] = dm.purity(rho_true)
    return out


if __name__ == "__main__":
    main()
