"""The CPU oracle against the reference itself, imported from /root/reference through the stub
harness (build container only; skipped on the GPU box)."""
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.reference


@pytest.fixture(scope="module")
def ref():
    import _ref_harness
    return _ref_harness.load_reference()


def _results(ref, settings, e, c):
    ER = ref.observable_estimation.ExperimentResult
    return [ER(setting=s, expectation=float(e[k]), std_err=0.0, total_counts=int(c[k]))
            for k, s in enumerate(settings)]


def test_flatten_results_matches_canonical_designs(ref):
    from fbx_oracle import design as od
    T = ref.tomography
    for n, qubits in ((1, [3]), (2, [0, 1]), (2, [5, 2])):
        for basis, gen in (("pauli", T._pauli_process_tomo_settings), ("sic", T._sic_process_tomo_settings)):
            settings = list(gen(qubits))
            res = _results(ref, settings, np.zeros(len(settings)), np.ones(len(settings)))
            d, _, _ = od.flatten_results(res, qubits, "process")
            c = od.process_design(n, basis)
            assert (d.in_labels == c.in_labels).all() and (d.paulis == c.paulis).all()


def test_design_matrix_and_counts_vector(ref):
    from fbx_oracle import design as od, estimators as oe
    T = ref.tomography
    rs = np.random.RandomState(0)
    for n, basis in ((1, "pauli"), (2, "sic")):
        qubits = list(range(n))
        gen = T._pauli_process_tomo_settings if basis == "pauli" else T._sic_process_tomo_settings
        settings = list(gen(qubits))
        e = rs.uniform(-1, 1, len(settings)); c = rs.randint(100, 1000, len(settings))
        A, nvec = T._extract_from_results(_results(ref, settings, e, c), qubits[::-1])
        d = od.process_design(n, basis)
        assert np.array_equal(A, oe.design_matrix_A(d))
        assert np.array_equal(nvec, oe.counts_vector(e, c))


def test_pgdb_bitwise_equal_incl_general_designs(ref):
    from fbx_oracle import design as od, estimators as oe
    import sys
    from fbx import synthetic
    T = ref.tomography
    qubits = [0]
    design, _, e, c = synthetic.process_batch(1, "sic", 2, first_item=40)
    settings = list(T._sic_process_tomo_settings(qubits))
    for b in range(2):
        res = _results(ref, settings, e[b], c[b])
        want = T.pgdb_process_estimate(res, qubits)
        d, ee, cc = od.flatten_results(res, qubits, "process")
        assert np.array_equal(want, oe.pgdb_process_estimate(d, ee, cc))
        perm = np.random.RandomState(b).permutation(len(res))
        res2 = [res[i] for i in perm] + [res[1]]
        want = T.pgdb_process_estimate(res2, qubits, trace_preserving=False)
        d, ee, cc = od.flatten_results(res2, qubits, "process")
        assert np.array_equal(want, oe.pgdb_process_estimate(d, ee, cc, trace_preserving=False))


def test_state_estimators_bitwise(ref):
    from fbx_oracle import design as od, estimators as oe
    from fbx import synthetic
    T = ref.tomography
    qubits = [0, 1]
    design, _, e, c = synthetic.state_batch(2, 2, first_item=7, mixed=0.05)
    settings = list(T._state_tomo_settings(qubits))
    for b in range(2):
        res = _results(ref, settings, e[b], c[b])
        d, ee, cc = od.flatten_results(res, qubits, "state")
        assert np.allclose(T.linear_inv_state_estimate(res, qubits), oe.linear_inv_state_estimate(d, ee), atol=1e-15)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            assert np.array_equal(T.iterative_mle_state_estimate(res, qubits, maxiter=30),
                                  oe.iterative_mle_state_estimate(d, ee, cc, maxiter=30))
        rho = oe.linear_inv_state_estimate(d, ee)
        assert np.array_equal(T._R(rho, res, qubits[::-1]), oe.R_operator(rho, d, ee))
        assert T.state_log_likelihood(rho, res, qubits) == oe.state_log_likelihood(rho, d, ee, cc)


def test_operator_tools_random(ref):
    from fbx_oracle import superops as so, measures as om
    OT, DM = ref.operator_tools, ref.distance_measures
    rs = np.random.RandomState(5)
    for d in (2, 4):
        D = d * d
        x = rs.randn(D, D) + 1j * rs.randn(D, D)
        h = x + x.conj().T
        for name in ("choi2superop", "superop2choi", "choi2pauli_liouville", "pauli_liouville2choi",
                     "superop2pauli_liouville", "pauli_liouville2superop", "chi2choi", "chi2pauli_liouville",
                     "chi2superop", "proj_choi_to_trace_preserving", "proj_choi_to_trace_non_increasing",
                     "proj_choi_to_completely_positive"):
            assert np.allclose(getattr(OT, name)(x), getattr(so, name)(x), atol=1e-13), name
        for name in ("choi2chi", "superop2chi", "pauli_liouville2chi", "proj_choi_to_physical", "proj_choi_to_unitary"):
            arg = h if name != "superop2chi" else OT.choi2superop(h)
            if name == "pauli_liouville2chi":
                arg = OT.choi2pauli_liouville(h)
            assert np.allclose(getattr(OT, name)(arg), getattr(so, name)(arg), atol=1e-12), name
        assert np.array_equal(OT.computational2pauli_basis_matrix(d), so.computational2pauli_basis_matrix(d))
        g = rs.randn(d, d) + 1j * rs.randn(d, d)
        rho = g @ g.conj().T; rho /= np.trace(rho)
        g = rs.randn(d, d) + 1j * rs.randn(d, d)
        sig = g @ g.conj().T; sig /= np.trace(sig)
        for name in ("fidelity", "trace_distance", "bures_distance", "bures_angle", "hilbert_schmidt_ip"):
            assert np.isclose(getattr(DM, name)(rho, sig), getattr(om, name)(rho, sig), atol=1e-13), name
        assert np.isclose(DM.purity(rho), om.purity(rho))
        assert np.allclose(DM.watrous_bounds(h), om.watrous_bounds(h))


def test_shots_to_obs_moments(ref):
    from fbx_oracle import acquisition as oa
    OE = ref.observable_estimation
    rs = np.random.RandomState(9)
    qubits = [3, 0, 5]
    bits = (rs.uniform(size=(500, 3)) < 0.3).astype(int)
    for ops, coef in (({3: "X", 5: "Z"}, 1.0), ({0: "Y"}, -0.7), ({}, 2.0), ({3: "Z", 0: "Z", 5: "Z"}, 1.0)):
        term = ref.PauliTerm("I", 0, coef)
        term._ops = dict(ops)
        mask = [1 if q in ops else 0 for q in qubits]
        for prior in (False, True):
            want = OE.shots_to_obs_moments(bits, qubits, term, prior)
            got = oa.shots_to_obs_moments(bits, mask, coef, prior)
            assert np.allclose(want, got, rtol=0, atol=1e-15)
    assert OE.ratio_variance(1.0, 0.1, 2.0, 0.05) == oa.ratio_variance(1.0, 0.1, 2.0, 0.05) == 0.028125


def test_setting_generators_and_text_forms_match_the_reference(ref):
    """The shim's settings come from its design tables and its record classes are its own code; what must
    agree with the reference is the ORDER of the settings and every text form (the interchange schema)."""
    from fbx import tomography as T, observable_estimation as oe
    RT, ROE = ref.tomography, ref.observable_estimation
    for qubits in ([0], [3, 1], [2, 0, 1]):
        mine = T.generate_state_tomography_settings(qubits)
        theirs = list(RT._state_tomo_settings(qubits))
        assert [str(s) for s in mine] == [str(s) for s in theirs]
        if len(qubits) == 3:
            continue
        for basis, gen in (("pauli", RT._pauli_process_tomo_settings), ("sic", RT._sic_process_tomo_settings)):
            mine = T.generate_process_tomography_settings(qubits, basis)
            theirs = list(gen(qubits))
            assert [str(s) for s in mine] == [str(s) for s in theirs]
            # text -> object -> text through both implementations
            for s in theirs[:: max(1, len(theirs) // 7)]:
                assert str(oe.ExperimentSetting.from_str(str(s))) == str(s)
                assert str(ROE.ExperimentSetting.from_str(str(oe.ExperimentSetting.from_str(str(s))))) == str(s)
    with pytest.raises(ValueError):
        T.generate_process_tomography_settings([0], "bogus")
    # record serialisation: same keys and values as the reference's
    rs = ROE.ExperimentSetting(ROE.plusX(0) * ROE.SIC2(1), ref.PauliTerm.from_list([("X", 0), ("Z", 1)]))
    ms = oe.ExperimentSetting.from_str(str(rs))
    rr = ROE.ExperimentResult(setting=rs, expectation=0.25, std_err=0.01, total_counts=500, raw_expectation=0.2)
    mr = oe.ExperimentResult(setting=ms, expectation=0.25, std_err=0.01, total_counts=500, raw_expectation=0.2)
    import json
    assert json.loads(json.dumps(mr, cls=oe.OperatorEncoder)) == json.loads(json.dumps(rr, cls=ROE.OperatorEncoder))
    for text in ("X+_0", "Y-_12", "SIC3_7", " Z+_1 "):
        assert str(oe._OneQState.from_str(text)) == str(ROE._OneQState.from_str(text))
    for bad in ("X+", "nonsense", "X?_1"):
        with pytest.raises(ValueError):
            oe._OneQState.from_str(bad)
        with pytest.raises(ValueError):
            ROE._OneQState.from_str(bad)
    assert oe.plusX(0) * oe.minusZ(1) == oe.minusZ(1) * oe.plusX(0)
    assert hash(oe.zeros_state([0, 1])) == hash(oe.plusZ(1) * oe.plusZ(0))


def test_beta_resampling_keeps_the_reference_draw_order(ref):
    from fbx import tomography as T, observable_estimation as oe
    RT, ROE = ref.tomography, ref.observable_estimation
    settings_r = list(RT._state_tomo_settings([0, 1]))
    settings_m = T.generate_state_tomography_settings([0, 1])
    rs = np.random.RandomState(3)
    e = rs.uniform(-0.9, 0.9, size=len(settings_r))
    res_r = [ROE.ExperimentResult(setting=s, expectation=float(x), std_err=0.0, total_counts=300) for s, x in zip(settings_r, e)]
    res_m = [oe.ExperimentResult(setting=s, expectation=float(x), std_err=0.0, total_counts=300) for s, x in zip(settings_m, e)]
    np.random.seed(77)
    want = [r.expectation for r in RT._resample_expectations_with_beta(res_r)]
    np.random.seed(77)
    got = [r.expectation for r in T._resample_expectations_with_beta(res_m)]
    assert got == want
