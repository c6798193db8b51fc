"""Two-qubit process tomography, start to finish, with the reference's own function names -- the flow of the reference's
process-tomography notebook (settings -> results -> estimates -> projections -> metrics -> error bars -> plot inputs),
minus the quantum computer: the "measured" expectations are sampled from the exact ones of a noisy CNOT.

    python examples/process_tomography_walkthrough.py            (needs libfbx.so and an MI355X)

Everything numerical below runs in libfbx's HIP kernels; swap `fbx` for `forest.benchmarking` in the imports and the
same script runs on the reference (except the *_batch lines, which are this library's many-experiments-at-once forms)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "forest-benchmarking_amd"))
from fbx import distance_measures as dm, plotting, synthetic, tomography                      # noqa: E402
from fbx.design import process_design                                                           # noqa: E402
from fbx.observable_estimation import ExperimentResult                                          # noqa: E402
from fbx.operator_tools import (choi2pauli_liouville, choi_is_cptp, kraus2choi, kraus2pauli_liouville,     # noqa: E402
                                proj_choi_to_physical, proj_choi_to_unitary)


def main(shots=2000, n_boot=40, verbose=True):
    say = print if verbose else (lambda *a, **k: None)
    qubits = [0, 1]
    cnot = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 0, 1], [0, 0, 1, 0]], dtype=complex)

    # 1. the experiment: 540 settings (36 input states x 15 Pauli observables), as generate_process_tomography_settings
    settings = tomography.generate_process_tomography_settings(qubits, in_basis="pauli")
    say(f"{len(settings)} settings, e.g. {settings[0]}  ...  {settings[-1]}")

    # 2. "data": exact expectations of a slightly depolarised CNOT, then binomial sampling with `shots` shots per setting
    design = process_design(2, "pauli")                    # the same 540 settings as SoA tables (for the *_batch calls)
    exact = synthetic.exact_process_expectations(design, cnot[None], depolarizing=0.03)
    expect, counts = synthetic.sample_expectations(exact, shots, first_item=7)
    results = [ExperimentResult(setting=s, expectation=float(e), std_err=float(np.sqrt((1 - e * e) / shots)), total_counts=shots)
               for s, e in zip(settings, expect[0])]

    # 3. estimates, reference signatures: List[ExperimentResult], qubits -> Choi matrix
    choi_lin = tomography.linear_inv_process_estimate(results, qubits)
    choi_mle = tomography.pgdb_process_estimate(results, qubits)
    say("linear inversion is CPTP:", choi_is_cptp(choi_lin), "| PGDB estimate is CPTP (1e-3):", choi_is_cptp(choi_mle, atol=1e-3))

    # 4. operator tools: projections and representations
    choi_phys = proj_choi_to_physical(choi_lin)
    choi_unitary = proj_choi_to_unitary(choi_mle)
    ptm_ideal = kraus2pauli_liouville(cnot)
    out = {"fidelity_linear_inversion_projected": dm.process_fidelity(ptm_ideal, choi2pauli_liouville(choi_phys)),
           "fidelity_pgdb": dm.process_fidelity(ptm_ideal, choi2pauli_liouville(choi_mle)),
           "fidelity_closest_unitary": dm.process_fidelity(ptm_ideal, choi2pauli_liouville(choi_unitary)),
           "diamond_norm_bounds_to_ideal": dm.watrous_bounds(choi_mle - kraus2choi(cnot))}
    for k, v in out.items():
        say(f"{k}: {v}")

    # 5. error bars: every one of `n_boot` Beta-resampled experiments reconstructed in ONE launch, resident in HBM
    mean, var = tomography.process_fidelity_variance_batch(design, expect, counts, ptm_ideal, n_resamples=n_boot, seed=1)
    out["bootstrap_fidelity"] = (float(mean[0]), float(np.sqrt(var[0])))
    say(f"process fidelity {mean[0]:.4f} +- {np.sqrt(var[0]):.4f}  ({n_boot} resamples)")

    # 6. what the reference's plot_pauli_transfer_matrix would draw
    ptm, labels = plotting.pauli_transfer_matrix_plot_inputs(choi_mle)
    out["ptm"], out["labels"] = ptm, labels
    say("PTM", ptm.shape, "labels", labels[:5], "...; largest deviation from the ideal PTM:", float(np.abs(ptm - ptm_ideal.real).max()))
    return out


if __name__ == "__main__":
    main()
