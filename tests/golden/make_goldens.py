"""Generate the golden fixtures in this directory by RUNNING THE REFERENCE.

Build-container only (needs /root/reference; see _ref_harness.py):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_goldens.py

Inputs come from the build's own synthetic-data generator (fbx.synthetic, SURVEY.md 8d
recipe); every expected output below is produced by the reference's own functions
(forest.benchmarking.tomography / operator_tools / distance_measures) called on the
reference's own ExperimentResult objects.  Fixtures are data only (inputs + outputs).
"""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "forest-benchmarking_amd"))
sys.dont_write_bytecode = True

import _ref_harness  # noqa: E402
from fbx import synthetic  # noqa: E402

ref = _ref_harness.load_reference()
T, OT, DM, OE = ref.tomography, ref.operator_tools, ref.distance_measures, ref.observable_estimation
PSM = ref.project_state_matrix


def ref_results(settings, e, c):
    return [OE.ExperimentResult(setting=s, expectation=float(e[k]), std_err=0.0,
                                total_counts=int(c[k])) for k, s in enumerate(settings)]


def process_settings(qubits, basis):
    f = T._pauli_process_tomo_settings if basis == "pauli" else T._sic_process_tomo_settings
    return list(f(qubits))


def make_process(n, basis, batch, tni_items):
    qubits = list(range(n))
    design, us, e, c = synthetic.process_batch(n, basis, batch)
    settings = process_settings(qubits, basis)
    assert len(settings) == design.m
    pgdb, linv, pgdb_tni = [], [], []
    for b in range(batch):
        res = ref_results(settings, e[b], c[b])
        pgdb.append(T.pgdb_process_estimate(res, qubits))
        linv.append(T.linear_inv_process_estimate(res, qubits))
        if b < tni_items:
            pgdb_tni.append(T.pgdb_process_estimate(res, qubits, trace_preserving=False))
    np.savez_compressed(os.path.join(HERE, f"process_{n}q_{basis}.npz"),
                        n_qubits=n, in_labels=design.in_labels, paulis=design.paulis,
                        unitaries=us, expectations=e, counts=c, pgdb=np.array(pgdb),
                        linv=np.array(linv), pgdb_tni=np.array(pgdb_tni))
    print("process", n, basis, "done")


def make_process_3q(batch=4):
    """BASELINE config 4 shape: 3 qubits, SIC in-basis (4032 settings), `batch` items -- the reference's
    pgdb_process_estimate run to convergence (about a minute per item here)."""
    qubits = [0, 1, 2]
    design, us, e, c = synthetic.process_batch(3, "sic", batch)
    settings = process_settings(qubits, "sic")
    assert len(settings) == design.m
    est, linv = [], []
    for b in range(batch):
        res = ref_results(settings, e[b], c[b])
        est.append(T.pgdb_process_estimate(res, qubits))
        linv.append(T.linear_inv_process_estimate(res, qubits))
        print("process 3 sic item", b, "done", flush=True)
    np.savez_compressed(os.path.join(HERE, "process_3q_sic.npz"), n_qubits=3, unitaries=us,
                        expectations=e, counts=c, pgdb=np.array(est), linv=np.array(linv))
    print("process 3 sic done")


def make_state(n, batch, tol_maxiter=2000):
    """(4 and 5 qubits: ``tol_maxiter`` 300 -- the reference takes ~0.1 s per iteration on 1023 settings)"""
    qubits = list(range(n))
    design, rhos, e, c = synthetic.state_batch(n, batch, mixed=0.1)
    settings = list(T._state_tomo_settings(qubits))
    assert len(settings) == design.m
    out = {k: [] for k in ("linv", "mle100", "mle_tol", "hedged", "maxent", "loglik", "r_op")}
    for b in range(batch):
        res = ref_results(settings, e[b], c[b])
        out["linv"].append(T.linear_inv_state_estimate(res, qubits))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m100 = T.iterative_mle_state_estimate(res, qubits, maxiter=100)
            out["mle100"].append(m100)
            out["mle_tol"].append(T.iterative_mle_state_estimate(res, qubits, epsilon=0.5, tol=1e-6,
                                                                 maxiter=tol_maxiter))
            out["hedged"].append(T.iterative_mle_state_estimate(res, qubits, beta=0.5, epsilon=1e-4,
                                                                maxiter=60))
            out["maxent"].append(T.iterative_mle_state_estimate(res, qubits, entropy_penalty=0.005,
                                                                maxiter=60))
        out["loglik"].append(T.state_log_likelihood(m100, res, qubits))
        out["r_op"].append(T._R(m100, res, qubits[::-1]))
    np.savez_compressed(os.path.join(HERE, f"state_{n}q.npz"), n_qubits=n, paulis=design.paulis,
                        truth=rhos, expectations=e, counts=c,
                        **{k: np.array(v) for k, v in out.items()})
    print("state", n, "done")


def rand_herm(rs, D, scale=1.0):
    g = rs.randn(D, D) + 1j * rs.randn(D, D)
    return scale * (g + g.conj().T) / 2


def make_superops(n, batch):
    d, D = 2 ** n, 4 ** n
    rs = np.random.RandomState(100 + n)
    out = {}
    for K in (1, 2, 4):
        ks = synthetic.kraus_batch(n, K, batch, seed=10 * n + K)
        out[f"kraus{K}"] = ks
        out[f"kraus{K}_choi"] = np.array([OT.kraus2choi(list(k)) for k in ks])
        out[f"kraus{K}_superop"] = np.array([OT.kraus2superop(list(k)) for k in ks])
        out[f"kraus{K}_ptm"] = np.array([OT.kraus2pauli_liouville(list(k)) for k in ks])
        out[f"kraus{K}_chi"] = np.array([OT.kraus2chi(list(k)) for k in ks])
    choi = out["kraus4_choi"]
    out["choi2chi"] = np.array([OT.choi2chi(x) for x in choi])
    out["choi2superop"] = np.array([OT.choi2superop(x) for x in choi])
    out["choi2ptm"] = np.array([OT.choi2pauli_liouville(x) for x in choi])
    out["chi2choi"] = np.array([OT.chi2choi(x) for x in out["kraus4_chi"]])
    out["chi2ptm"] = np.array([OT.chi2pauli_liouville(x) for x in out["kraus4_chi"]])
    out["chi2superop"] = np.array([OT.chi2superop(x) for x in out["kraus4_chi"]])
    out["superop2choi"] = np.array([OT.superop2choi(x) for x in out["kraus4_superop"]])
    out["superop2ptm"] = np.array([OT.superop2pauli_liouville(x) for x in out["kraus4_superop"]])
    out["superop2chi"] = np.array([OT.superop2chi(x) for x in out["kraus4_superop"]])
    out["ptm2choi"] = np.array([OT.pauli_liouville2choi(x) for x in out["kraus4_ptm"]])
    out["ptm2superop"] = np.array([OT.pauli_liouville2superop(x) for x in out["kraus4_ptm"]])
    out["ptm2chi"] = np.array([OT.pauli_liouville2chi(x) for x in out["kraus4_ptm"]])
    # non-CP Hermitian input: choi2chi goes through |C|
    herm = np.array([rand_herm(rs, D) for _ in range(batch)])
    out["herm"] = herm
    out["herm_choi2chi"] = np.array([OT.choi2chi(x) for x in herm])
    # process fidelity against a fixed reference unitary channel
    u_ref = synthetic.haar_unitary(d, np.random.RandomState(7))
    ptm_ref = OT.kraus2pauli_liouville(u_ref)
    out["ptm_ref"] = ptm_ref
    out["proc_fid"] = np.array([DM.process_fidelity(ptm_ref, x) for x in out["kraus4_ptm"]])
    out["ent_fid"] = np.array([DM.entanglement_fidelity(ptm_ref, x) for x in out["kraus4_ptm"]])
    # projections: general complex input and Hermitian perturbations of CPTP maps
    gen = np.array([rs.randn(D, D) + 1j * rs.randn(D, D) for _ in range(batch)]) * 0.3
    near = choi + np.array([rand_herm(rs, D, 0.2) for _ in range(batch)])
    for name, x in (("gen", gen), ("near", near)):
        out[f"proj_{name}_in"] = x
        out[f"proj_{name}_cp"] = np.array([OT.proj_choi_to_completely_positive(v) for v in x])
        out[f"proj_{name}_tp"] = np.array([OT.proj_choi_to_trace_preserving(v) for v in x])
        out[f"proj_{name}_tni"] = np.array([OT.proj_choi_to_trace_non_increasing(v) for v in x])
    out["proj_near_phys_tp"] = np.array([OT.proj_choi_to_physical(v) for v in near])
    out["proj_near_phys_tni"] = np.array([OT.proj_choi_to_physical(v, False) for v in near])
    # states: unphysical Hermitian trace-ish-one matrices and physical pairs
    unphys = np.array([np.eye(d) / d + rand_herm(rs, d, 0.25) for _ in range(batch)])
    out["state_unphys"] = unphys
    out["state_proj"] = np.array([PSM.project_state_matrix_to_physical(v) for v in unphys])
    g1 = rs.randn(batch, d, d) + 1j * rs.randn(batch, d, d)
    g2 = rs.randn(batch, d, d) + 1j * rs.randn(batch, d, d)
    rho = np.array([g @ g.conj().T / np.trace(g @ g.conj().T) for g in g1])
    sig = np.array([g @ g.conj().T / np.trace(g @ g.conj().T) for g in g2])
    out["rho"], out["sigma"] = rho, sig
    out["purity"] = np.array([DM.purity(r) for r in rho])
    out["fidelity"] = np.array([DM.fidelity(r, s) for r, s in zip(rho, sig)])
    out["trace_distance"] = np.array([DM.trace_distance(r, s) for r, s in zip(rho, sig)])
    out["hs_ip"] = np.array([DM.hilbert_schmidt_ip(r, s) for r, s in zip(rho, sig)])
    out["apply_choi"] = np.array([OT.apply_choi_matrix_2_state(cx, r) for cx, r in zip(choi, rho)])
    np.savez_compressed(os.path.join(HERE, f"superops_{n}q.npz"), n_qubits=n, **out)
    print("superops", n, "done")


def make_extras():
    """Rows a19 / a27 / a28-a29 extras: calculational helpers, seeded random operators (the draw
    order is part of the contract), quantum Chernoff bound, Watrous bounds."""
    RO = ref.random_operators
    CALC = ref.calculational
    out = {}
    np.random.seed(1234)
    out["ro_ginibre_3_2"] = RO.ginibre_matrix_complex(3, 2)
    out["ro_haar_u4"] = RO.haar_rand_unitary(4)
    out["ro_haar_state4"] = RO.haar_rand_state(4)
    out["ro_ginibre_state_4_2"] = RO.ginibre_state_matrix(4, 2)
    out["ro_bures4"] = RO.bures_measure_state_matrix(4)
    out["ro_bcsz_2_2"] = RO.rand_map_with_BCSZ_dist(2, 2)
    out["ro_bcsz_4_3"] = RO.rand_map_with_BCSZ_dist(4, 3)
    rs = np.random.RandomState(7)
    out["ro_rs_ginibre_2_3"] = RO.ginibre_matrix_complex(2, 3, rs)
    out["ro_rs_haar_u3"] = RO.haar_rand_unitary(3, rs)
    for k, (dims, perm) in enumerate([(2, [1, 0]), (2, [1, 2, 0]), ([2, 3, 5], [2, 0, 1]), (3, [0, 2, 1]),
                                      ([2, 3, 2, 2], [2, 3, 1, 0])]):
        out[f"perm{k}_dims"] = np.atleast_1d(dims)
        out[f"perm{k}_perm"] = np.array(perm)
        out[f"perm{k}_out"] = RO.permute_tensor_factors(dims, perm)
    rs = np.random.RandomState(99)
    m24 = rs.randn(24, 24) + 1j * rs.randn(24, 24)
    out["pt_in"] = m24
    for k, keep in enumerate([[0], [1], [2], [0, 2], [1, 2], [0, 1, 2]]):
        out[f"pt{k}_keep"] = np.array(keep)
        out[f"pt{k}_out"] = CALC.partial_trace(m24, keep, [2, 3, 4])
    for N in (4, 16):
        g = rs.randn(N, N) + 1j * rs.randn(N, N)
        psd = g @ g.conj().T
        out[f"psd{N}"] = psd
        out[f"sqrtm{N}"] = CALC.sqrtm_psd(psd)
    a, b = rs.randn(5, 1) + 1j * rs.randn(5, 1), rs.randn(5, 1) + 1j * rs.randn(5, 1)
    out["ket_a"], out["ket_b"] = a, b
    out["outer"], out["inner"] = CALC.outer_product(a, b), CALC.inner_product(a, b)
    for d in (2, 4):
        pairs, qcbs = [], []
        for _ in range(3):
            # real symmetric states: with this scipy the reference's objective turns complex for
            # complex input and fractional_matrix_power rejects the complex exponent that follows
            g1, g2 = rs.randn(d, d), rs.randn(d, d)
            r1, r2 = g1 @ g1.T, g2 @ g2.T
            r1, r2 = r1 / np.trace(r1).real, r2 / np.trace(r2).real
            pairs.append((r1, r2))
            qcbs.append([float(x) for x in DM.quantum_chernoff_bound(r1, r2)])
        out[f"qcb{d}_rho"] = np.array([p[0] for p in pairs])
        out[f"qcb{d}_sigma"] = np.array([p[1] for p in pairs])
        out[f"qcb{d}"] = np.array(qcbs)
    herm16 = rand_herm(rs, 16)
    gen4 = rs.randn(4, 4) + 1j * rs.randn(4, 4)
    gen16 = rs.randn(16, 16) + 1j * rs.randn(16, 16)
    for name, x in (("herm16", herm16), ("gen4", gen4), ("gen16", gen16)):
        out[f"wat_{name}"] = x
        out[f"wat_{name}_out"] = np.array(DM.watrous_bounds(x))
    # direct fidelity estimate (direct_fidelity_estimation.py:224-307) on the reference's own records
    import importlib
    DFE = importlib.import_module("forest.benchmarking.direct_fidelity_estimation")
    for n in (1, 2, 3):
        qs, m = list(range(n)), 5 + 2 * n
        e, se = rs.uniform(-1, 1, size=m), rs.uniform(0, 0.1, size=m)
        res = [OE.ExperimentResult(
            setting=OE.ExperimentSetting(OE.zeros_state(qs), ref.PauliTerm.from_list([("XYZ"[(k + q) % 3], q) for q in qs])),
            expectation=float(e[k]), std_err=float(se[k]), total_counts=100) for k in range(m)]
        out[f"dfe{n}_e"], out[f"dfe{n}_se"] = e, se
        out[f"dfe{n}_state"] = np.array(DFE.estimate_dfe(res, "state"))
        out[f"dfe{n}_process"] = np.array(DFE.estimate_dfe(res, "process"))
    np.savez_compressed(os.path.join(HERE, "extras.npz"), **out)
    print("extras done")


def make_round2():
    """Rows a24 / a25 / a26 / a28 / f2 added in round 2: Kraus composition and tensoring, Pauli twirl,
    Kraus application (square and non-square, apply_superoperator.py:33-57), the calibration rescale
    (observable_estimation.py:1028-1037) and operators whose dimension is not a power of two."""
    CS = __import__("forest.benchmarking.operator_tools.compose_superoperators", fromlist=["x"])
    CA = __import__("forest.benchmarking.operator_tools.channel_approximation", fromlist=["x"])
    AS = __import__("forest.benchmarking.operator_tools.apply_superoperator", fromlist=["x"])
    VO = __import__("forest.benchmarking.operator_tools.validate_operator", fromlist=["x"])
    CALC = ref.calculational
    rs = np.random.RandomState(2024)
    out = {}

    def cset(K, r, c):
        return rs.randn(K, r, c) + 1j * rs.randn(K, r, c)
    cases = [(2, 2, 2, 3, 2, 2), (1, 4, 4, 4, 4, 4), (3, 2, 4, 2, 4, 2), (2, 3, 3, 2, 3, 3)]   # K2, r2, c2, K1, r1, c1
    for k, (K2, r2, c2, K1, r1, c1) in enumerate(cases):
        a, b = cset(K2, r2, c2), cset(K1, r1, c1)
        out[f"pairs{k}_k2"], out[f"pairs{k}_k1"] = a, b
        out[f"pairs{k}_tensor"] = np.array(CS.tensor_channel_kraus(list(a), list(b)))
        if c2 == r1:
            out[f"pairs{k}_compose"] = np.array(CS.compose_channel_kraus(list(a), list(b)))
    for D in (4, 16):
        chi = rs.randn(3, D, D) + 1j * rs.randn(3, D, D)
        out[f"twirl{D}_in"] = chi
        out[f"twirl{D}_out"] = np.array([CA.pauli_twirl_chi_matrix(x) for x in chi])
    # real Kraus sets on real states (the only operands the reference's real accumulator accepts)
    for k, (K, rows, cols) in enumerate([(2, 2, 2), (3, 4, 4), (2, 2, 4), (2, 3, 3), (1, 8, 8), (2, 4, 8)]):
        ks = rs.randn(K, rows, cols)
        g = rs.randn(cols, cols)
        st = g @ g.T / np.trace(g @ g.T)
        out[f"applyk{k}_kraus"], out[f"applyk{k}_state"] = ks, st
        out[f"applyk{k}_out"] = AS.apply_kraus_ops_2_state(list(ks), st)
    # calibration rescale: the two lines of observable_estimation.py:1033-1034 on random inputs
    e, se = rs.uniform(-1, 1, size=(5, 12)), rs.uniform(0.001, 0.1, size=(5, 12))
    cm, cv = rs.uniform(0.6, 0.99, size=4), rs.uniform(1e-5, 1e-3, size=4)
    idx = rs.randint(0, 4, size=12)
    out["cal_e"], out["cal_se"], out["cal_mean"], out["cal_var"], out["cal_index"] = e, se, cm, cv, idx
    out["cal_out_mean"] = e / cm[idx][None]
    out["cal_out_err"] = np.sqrt(OE.ratio_variance(e, se ** 2, cm[idx][None], cv[idx][None]))
    # dimensions that are not powers of two
    for d in (3, 5, 6):
        g1, g2 = rs.randn(d, d) + 1j * rs.randn(d, d), rs.randn(d, d) + 1j * rs.randn(d, d)
        r1, r2 = g1 @ g1.conj().T, g2 @ g2.conj().T
        r1, r2 = r1 / np.trace(r1).real, r2 / np.trace(r2).real
        out[f"gd{d}_rho"], out[f"gd{d}_sigma"] = r1, r2
        out[f"gd{d}_measures"] = np.array([DM.purity(r1), DM.fidelity(r1, r2), DM.trace_distance(r1, r2),
                                           DM.hilbert_schmidt_ip(r1, r2)])
        out[f"gd{d}_sqrtm"] = CALC.sqrtm_psd(r1)
        out[f"gd{d}_eigvals"] = np.linalg.eigvalsh(r1 - r2)
        out[f"gd{d}_psd"] = np.array([VO.is_positive_semidefinite_matrix(r1), VO.is_positive_semidefinite_matrix(r1 - r2),
                                      VO.is_positive_definite_matrix(r1)])
    u = synthetic.haar_unitary(4, np.random.RandomState(3))
    v = synthetic.haar_unitary(4, np.random.RandomState(4))
    out["hs_u"], out["hs_v"] = u, v
    out["hs_uv"] = np.array([DM.hilbert_schmidt_ip(u, v)])             # complex: the operands are not Hermitian
    qutrit_choi = OT.kraus2choi([np.diag([1.0, np.exp(0.3j), np.exp(-0.7j)])])
    out["qutrit_choi"] = qutrit_choi
    out["qutrit_apply"] = AS.apply_choi_matrix_2_state(qutrit_choi, out["gd3_rho"])
    np.savez_compressed(os.path.join(HERE, "round2.npz"), **out)
    print("round2 done")


# ------------------------------------------------------------------------------------------------
# Timed mode (round 3): the benchmark runs EXACTLY 100 outer iterations (FBX_MODE_FIXED), a loop the
# reference itself never runs past its stopping rule.  The fixtures below are produced by a driver loop
# that is tomography.py:563-592 statement by statement -- every number comes from the reference's own
# _extract_from_results / _cost / _grad_cost / proj_choi_to_physical -- with the `break` of :589 replaced
# by a record of the iteration at which it would have fired.  One run therefore yields the reference's
# converged estimate (a snapshot at that iteration, asserted bit-identical to a direct
# pgdb_process_estimate call for the first items of every set), the fixed-N estimate, and per-iteration
# Dykstra / halving counts and costs.
def ref_pgdb_trace(results, qubits, n_iters, trace_preserving=True, max_iters=400):
    """Runs until BOTH `n_iters` iterations are done (the fixed-N snapshot) and the stopping rule of :589 has fired
    once (the converged snapshot) -- three-qubit experiments can need more than 100 iterations to converge."""
    import importlib
    PS = importlib.import_module("forest.benchmarking.operator_tools.project_superoperators")
    calls = [0]
    orig_cp = PS.proj_choi_to_completely_positive

    def counting_cp(choi):                      # one CP projection per Dykstra iteration
        calls[0] += 1
        return orig_cp(choi)

    PS.proj_choi_to_completely_positive = counting_cp
    try:
        A, n = T._extract_from_results(results, qubits[::-1])
        dim = 2 ** len(qubits)
        est = np.eye(dim ** 2, dim ** 2, dtype=complex) / dim
        old_cost = T._cost(A, n, est)
        mu = 3 / (2 * dim ** 2)
        gamma = .3
        dyk, bts, costs = [], [], []
        conv_iter, conv_est, fixed_est = 0, None, None
        it = 0
        while it < max_iters and (fixed_est is None or conv_est is None):
            gradient = T._grad_cost(A, n, est)
            calls[0] = 0
            update = T.proj_choi_to_physical(est - gradient / mu, trace_preserving) - est
            dyk.append(calls[0])
            alpha = 1
            new_cost = T._cost(A, n, est + alpha * update)
            change = gamma * alpha * np.dot(OT.vec(update).conj().T, OT.vec(gradient))
            bt = 0
            while new_cost > old_cost + change:
                alpha = .5 * alpha
                change = .5 * change
                new_cost = T._cost(A, n, est + alpha * update)
                bt += 1
                if alpha < 1e-15:
                    break
            est += alpha * update
            it += 1
            bts.append(bt)
            costs.append(float(np.real(new_cost).ravel()[0]))
            if conv_est is None and old_cost - new_cost < 1e-10:      # tomography.py:589 would stop here
                conv_iter, conv_est = it, est.copy()
            if it == n_iters:
                fixed_est = est.copy()
            old_cost = new_cost
    finally:
        PS.proj_choi_to_completely_positive = orig_cp
    assert fixed_est is not None and conv_est is not None, "not converged within max_iters"
    L = 256                                      # common trace length of the fixtures; -1 = iteration not run
    pad = lambda v, dt: np.concatenate([np.asarray(v, dtype=dt), np.full(L - len(v), -1, dtype=dt)])
    return dict(fixed=fixed_est, conv=conv_est, conv_iter=conv_iter, dykstra=pad(dyk, np.int32),
                backtracks=pad(bts, np.int32), costs=pad(costs, np.float64))


def _fixed_worker(job):
    n, basis, b, n_iters, check_direct = job
    qubits = list(range(n))
    design, us, e, c = synthetic.process_batch(n, basis, 1, first_item=b)
    settings = process_settings(qubits, basis)
    res = ref_results(settings, e[0], c[0])
    # per-item cache (scratch, outside the repository) so that an interrupted run resumes
    cache = os.path.join(os.environ.get("TMPDIR", "/tmp"), "fbx_fixed_cache", f"{n}q_{basis}_{n_iters}_{b}.npz")
    if os.path.exists(cache):
        z = np.load(cache)
        tr = {k: z[k] for k in z.files}
        tr["conv_iter"] = int(tr["conv_iter"])
        return b, us[0], e[0], c[0], tr
    tr = ref_pgdb_trace(res, qubits, n_iters)
    if check_direct:
        direct = T.pgdb_process_estimate(res, qubits)
        assert tr["conv_iter"] > 0 and np.array_equal(direct, tr["conv"]), "driver loop departs from the reference"
    os.makedirs(os.path.dirname(cache), exist_ok=True)
    np.savez(cache + ".tmp.npz", **tr)
    os.replace(cache + ".tmp.npz", cache)
    print(f"fixed {n}q {basis} item {b}: conv_iter {tr['conv_iter']}, dykstra {tr['dykstra'].sum()}, "
          f"halvings {tr['backtracks'].sum()}", flush=True)
    return b, us[0], e[0], c[0], tr


def make_process_fixed(n, basis, batch, n_iters=100, n_direct=2, workers=4, tag=None):
    """`batch` bench items (synthetic.process_batch items 0..batch-1) through ref_pgdb_trace."""
    import multiprocessing as mp
    jobs = [(n, basis, b, n_iters, b < n_direct) for b in range(batch)]
    with mp.Pool(workers) as pool:
        out = sorted(pool.map(_fixed_worker, jobs, chunksize=1), key=lambda r: r[0])
    design = synthetic.process_batch(n, basis, 1)[0]
    np.savez_compressed(os.path.join(HERE, tag or f"process_{n}q_{basis}_fixed{n_iters}.npz"),
                        n_qubits=n, n_iters=n_iters, in_labels=design.in_labels, paulis=design.paulis,
                        unitaries=np.array([r[1] for r in out]), expectations=np.array([r[2] for r in out]),
                        counts=np.array([r[3] for r in out]),
                        pgdb_fixed=np.array([r[4]["fixed"] for r in out]),
                        pgdb_conv=np.array([r[4]["conv"] for r in out]),
                        conv_iter=np.array([r[4]["conv_iter"] for r in out], dtype=np.int32),
                        dykstra=np.array([r[4]["dykstra"] for r in out]),
                        backtracks=np.array([r[4]["backtracks"] for r in out]),
                        costs=np.array([r[4]["costs"] for r in out]))
    print("fixed", n, basis, batch, "done")


# Round 5: how far the reference's OWN fixed-100 estimate moves when the very same experiment is handed to it with its
# settings in another order (another summation order inside A @ vec(E) and n.T @ log p -- nothing else changes).  Past its
# stopping point the loop of tomography.py:578-585 is decided by the rounding of those sums, so this spread is the
# reproducibility of the reference in the timed mode, item by item; the GPU tests bound the kernel's deviation by it.
def _spread_worker(job):
    n, basis, b, n_iters, seed = job
    qubits = list(range(n))
    _, _, e, c = synthetic.process_batch(n, basis, 1, first_item=b)
    settings = process_settings(qubits, basis)
    perm = np.random.RandomState(1000 * seed + b).permutation(len(settings))
    res = ref_results([settings[k] for k in perm], e[0][perm], c[0][perm])
    tr = ref_pgdb_trace(res, qubits, n_iters)
    return b, seed, tr["fixed"], tr["conv"], int(tr["conv_iter"]), int(tr["backtracks"][:n_iters].sum()), int(tr["dykstra"][:n_iters].sum())


def make_fixed_spread(n, basis, seeds=(1, 2, 3), workers=8):
    import multiprocessing as mp
    g = np.load(os.path.join(HERE, f"process_{n}q_{basis}_fixed100.npz"))
    batch, n_iters = g["expectations"].shape[0], int(g["n_iters"])
    jobs = [(n, basis, b, n_iters, s) for b in range(batch) for s in seeds]
    with mp.Pool(workers) as pool:
        out = pool.map(_spread_worker, jobs, chunksize=1)
    K = len(seeds)
    fixed = np.zeros((batch, K)); conv = np.zeros((batch, K)); bts = np.zeros((batch, K), dtype=np.int64)
    for b, s, fx, cv, ci, bt, dy in out:
        k = seeds.index(s)
        assert ci == int(g["conv_iter"][b]) and dy == int(g["dykstra"][b][:n_iters].sum()), "a re-ordering changed an iteration count"
        fixed[b, k] = np.abs(fx - g["pgdb_fixed"][b]).max()
        conv[b, k] = np.abs(cv - g["pgdb_conv"][b]).max()
        bts[b, k] = bt
    np.savez_compressed(os.path.join(HERE, f"process_{n}q_{basis}_fixed100_spread.npz"), seeds=np.array(seeds),
                        fixed_spread=fixed, conv_spread=conv, backtracks=bts,
                        backtracks_unpermuted=np.array([int(g["backtracks"][b][:n_iters].sum()) for b in range(batch)]))
    print(f"spread {n}q {basis}: fixed-100 max {fixed.max():.2e} median of per-item max {np.median(fixed.max(axis=1)):.2e}; "
          f"converge max {conv.max():.2e}")


# Round 5: MERGED / REPEATED datasets.  The reference loops over whatever result list it is handed (tomography.py:494-539,
# :273-338): the same design measured several times -- here with different shot counts per repetition -- is one experiment of
# reps x m settings.  These lists exceed the register- / LDS-resident sizes of the kernels (2 qubits: 1024 settings, 1 qubit:
# 256, state designs: 64 KiB of per-setting staging) and exercise their streamed forms.
def _repeated(exact, reps, shots, first_item):
    es, cs = [], []
    for r in range(reps):
        e, c = synthetic.sample_expectations(exact, shots[r % len(shots)], first_item, seed_base=2000 + 1000 * r)
        es.append(e); cs.append(c)
    return np.concatenate(es, axis=1), np.concatenate(cs, axis=1)


def make_repeated():
    out = {}
    shots = (1000, 500, 2000)
    for tag, n, basis, reps, batch in (("p2pauli", 2, "pauli", 3, 3), ("p2sic", 2, "sic", 5, 2), ("p1pauli", 1, "pauli", 15, 3)):
        qubits = list(range(n))
        design = synthetic.process_design(n, basis)
        us = np.array([synthetic.haar_unitary(design.dim, np.random.RandomState(1000 + b)) for b in range(batch)])
        e, c = _repeated(synthetic.exact_process_expectations(design, us), reps, shots, 0)
        settings = process_settings(qubits, basis) * reps
        pg, lv, tni = [], [], []
        for b in range(batch):
            res = ref_results(settings, e[b], c[b])
            pg.append(T.pgdb_process_estimate(res, qubits))
            lv.append(T.linear_inv_process_estimate(res, qubits))
            if b == 0:
                tni.append(T.pgdb_process_estimate(res, qubits, trace_preserving=False))
        out.update({f"{tag}_in_labels": np.tile(design.in_labels, (reps, 1)), f"{tag}_paulis": np.tile(design.paulis, (reps, 1)),
                    f"{tag}_e": e, f"{tag}_c": c, f"{tag}_u": us, f"{tag}_pgdb": np.array(pg), f"{tag}_linv": np.array(lv),
                    f"{tag}_pgdb_tni": np.array(tni)})
        print("repeated", tag, e.shape, flush=True)
    for tag, n, reps, batch in (("s2", 2, 280, 2), ("s1", 1, 1400, 2)):
        qubits = list(range(n))
        design, rhos, _, _ = synthetic.state_batch(n, batch, mixed=0.1)
        e, c = _repeated(synthetic.exact_state_expectations(design, rhos), reps, shots, 0)
        settings = list(T._state_tomo_settings(qubits)) * reps
        mle, lv, ll, rop, hed = [], [], [], [], []
        for b in range(batch):
            res = ref_results(settings, e[b], c[b])
            lv.append(T.linear_inv_state_estimate(res, qubits))
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                m = T.iterative_mle_state_estimate(res, qubits, maxiter=40)
                hed.append(T.iterative_mle_state_estimate(res, qubits, beta=0.5, epsilon=1e-4, maxiter=12))
            mle.append(m)
            ll.append(T.state_log_likelihood(m, res, qubits))
            rop.append(T._R(m, res, qubits[::-1]))
        out.update({f"{tag}_paulis": np.tile(design.paulis, (reps, 1)), f"{tag}_e": e, f"{tag}_c": c, f"{tag}_truth": rhos,
                    f"{tag}_mle40": np.array(mle), f"{tag}_hedged12": np.array(hed), f"{tag}_linv": np.array(lv),
                    f"{tag}_loglik": np.array(ll), f"{tag}_r_op": np.array(rop)})
        print("repeated", tag, e.shape, flush=True)
    np.savez_compressed(os.path.join(HERE, "repeated.npz"), **out)
    print("repeated done")


def make_repeated_3q():
    """One 3-qubit experiment with the SIC design measured four times (16 128 settings; 500 / 1000 / 2000 / 1000 shots): beyond the
    14 336 settings of the resident 3-qubit instantiations.  The reference's dense A is 32 256 x 4096 complex = 2.1 GB."""
    n, basis, reps = 3, "sic", 4
    qubits = list(range(n))
    design = synthetic.process_design(n, basis)
    us = np.array([synthetic.haar_unitary(design.dim, np.random.RandomState(1000))])
    e, c = _repeated(synthetic.exact_process_expectations(design, us), reps, (500, 1000, 2000, 1000), 0)
    res = ref_results(process_settings(qubits, basis) * reps, e[0], c[0])
    est = T.pgdb_process_estimate(res, qubits)
    np.savez_compressed(os.path.join(HERE, "repeated_3q.npz"), in_labels=np.tile(design.in_labels, (reps, 1)),
                        paulis=np.tile(design.paulis, (reps, 1)), e=e, c=c, u=us, pgdb=np.array([est]))
    print("repeated 3q done", e.shape)


def make_sweep_3q(batch=6):
    """The 3-qubit leg of BASELINE configs[2]'s pipeline for `batch` random CPTP Kraus sets (K = 4, 8 x 8 operators): what the
    reference's kraus2choi / kraus2pauli_liouville / kraus2chi / choi2chi / process_fidelity return (round 4: the fused
    sweep3_kernel is checked against these; superops_3q.npz holds a single item)."""
    ks = synthetic.kraus_batch(3, 4, batch, seed=34)
    cnot = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 0, 1], [0, 0, 1, 0]], dtype=complex)
    had = np.array([[1, 1], [1, -1]], dtype=complex) / np.sqrt(2)
    ref = OT.kraus2pauli_liouville([np.kron(cnot, had)])
    out = {"kraus4": ks, "ptm_ref": ref,
           "choi": np.array([OT.kraus2choi(list(k)) for k in ks]),
           "ptm": np.array([OT.kraus2pauli_liouville(list(k)) for k in ks]),
           "chi": np.array([OT.kraus2chi(list(k)) for k in ks])}
    # (the reference's eigh route choi2chi gives the same matrices for these CP inputs, and its Pauli-Liouville matrices are real:
    # checked here, stored once)
    assert max(np.abs(OT.choi2chi(c) - x).max() for c, x in zip(out["choi"], out["chi"])) < 1e-12
    assert np.abs(out["ptm"].imag).max() < 1e-14 and np.abs(ref.imag).max() < 1e-14
    out["proc_fid"] = np.array([DM.process_fidelity(ref, p) for p in out["ptm"]])
    out["ptm"], out["ptm_ref"] = out["ptm"].real.copy(), ref.real.copy()
    np.savez_compressed(os.path.join(HERE, "sweep_3q.npz"), **out)
    print("sweep 3q done", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    np.random.seed(0)
    if "--sweep3q" in sys.argv:
        make_sweep_3q()
        sys.exit(0)
    if "--repeated3q" in sys.argv:        # round 5: a 3-qubit list of 16 128 settings (several minutes, ~6 GB)
        make_repeated_3q()
        sys.exit(0)
    if "--repeated" in sys.argv:          # round 5: merged / repeated datasets beyond the kernels' resident sizes (a few minutes)
        make_repeated()
        sys.exit(0)
    if "--fixed-spread" in sys.argv:      # round 5: the reference against its own re-ordered self (about a minute on 8 cores)
        make_fixed_spread(2, "pauli")
        make_fixed_spread(2, "sic")
        sys.exit(0)
    if "--fixed2q" in sys.argv:           # timed-mode fixtures, 2 qubits (a few minutes on 6 cores)
        make_process_fixed(2, "pauli", 64, workers=6)
        make_process_fixed(2, "sic", 16, workers=6)
        sys.exit(0)
    if "--fixed3q" in sys.argv:           # 3 qubits: 16 items, ~1.5 GB and a few minutes each
        if "--basis" in sys.argv and sys.argv[sys.argv.index("--basis") + 1] == "pauli":
            # round 4: the Pauli in-basis instantiation (13 608 settings: the reference's dense A is 27 216 x 4096
            # complex = 1.8 GB, ~5 GB per worker while it is assembled); 4 bench items, converge + fixed-100 snapshots
            make_process_fixed(3, "pauli", 4, workers=2, n_direct=1)
        else:
            make_process_fixed(3, "sic", 16, workers=4)
        sys.exit(0)
    if "--round2" in sys.argv:
        make_round2()
        sys.exit(0)
    if "--extras" in sys.argv:
        make_extras()
        sys.exit(0)
    if "--3q" in sys.argv:
        make_process_3q()
        make_superops(3, 1)
        sys.exit(0)
    if "--state45" in sys.argv:
        make_state(4, 3, tol_maxiter=300)
        make_state(5, 2, tol_maxiter=300)
        sys.exit(0)
    if "--2q" in sys.argv:
        make_process(2, "sic", 16, 2)
        make_process(2, "pauli", 16, 1)
        sys.exit(0)
    make_process(1, "pauli", 6, 3)
    make_process(1, "sic", 6, 3)
    make_process(2, "sic", 16, 2)
    make_process(2, "pauli", 16, 1)
    make_state(1, 6)
    make_state(2, 4)
    make_state(4, 3, tol_maxiter=300)
    make_state(5, 2, tol_maxiter=300)
    make_superops(1, 6)
    make_superops(2, 6)
    make_extras()
