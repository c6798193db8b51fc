"""Host-side mirrors (random_operators, calculational index helpers) and the oracle's spectral
measures against outputs of the reference itself (tests/golden/extras.npz, make_goldens.py --extras)."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def g():
    return np.load(os.path.join(GOLD, "extras.npz"))


def test_random_operators_follow_the_reference_draw_order(g):
    from fbx.operator_tools import random_operators as ro
    np.random.seed(1234)
    assert np.abs(ro.ginibre_matrix_complex(3, 2) - g["ro_ginibre_3_2"]).max() == 0
    assert np.abs(ro.haar_rand_unitary(4) - g["ro_haar_u4"]).max() < 1e-14
    assert np.abs(ro.haar_rand_state(4) - g["ro_haar_state4"]).max() < 1e-14
    assert np.abs(ro.ginibre_state_matrix(4, 2) - g["ro_ginibre_state_4_2"]).max() < 1e-14
    assert np.abs(ro.bures_measure_state_matrix(4) - g["ro_bures4"]).max() < 1e-13
    assert np.abs(ro.rand_map_with_BCSZ_dist(2, 2) - g["ro_bcsz_2_2"]).max() < 1e-12
    assert np.abs(ro.rand_map_with_BCSZ_dist(4, 3) - g["ro_bcsz_4_3"]).max() < 1e-11
    rs = np.random.RandomState(7)
    assert np.abs(ro.ginibre_matrix_complex(2, 3, rs) - g["ro_rs_ginibre_2_3"]).max() == 0
    assert np.abs(ro.haar_rand_unitary(3, rs) - g["ro_rs_haar_u3"]).max() < 1e-14
    with pytest.raises(ValueError):
        ro.ginibre_state_matrix(2, 3)


def test_random_operator_properties():
    from fbx.operator_tools import random_operators as ro
    np.random.seed(5)
    u = ro.haar_rand_unitary(8)
    assert np.abs(u @ u.conj().T - np.eye(8)).max() < 1e-13
    psi = ro.haar_rand_state(4)
    assert psi.shape == (4, 1) and abs(np.vdot(psi, psi) - 1) < 1e-13
    rho = ro.bures_measure_state_matrix(4)
    assert abs(np.trace(rho) - 1) < 1e-13 and np.linalg.eigvalsh(rho).min() > -1e-14
    choi = ro.rand_map_with_BCSZ_dist(4, 5)
    pt = np.einsum("iaja->ij", choi.reshape(4, 4, 4, 4))      # trace preserving: Tr_out = identity
    assert np.abs(pt - np.eye(4)).max() < 1e-12
    assert np.linalg.eigvalsh(choi).min() > -1e-12
    assert np.linalg.matrix_rank(choi, tol=1e-9) == 5


def test_permute_tensor_factors(g):
    from fbx.operator_tools import permute_tensor_factors
    for k in range(5):
        dims = g[f"perm{k}_dims"]
        dims = int(dims[0]) if dims.size == 1 else [int(x) for x in dims]
        got = permute_tensor_factors(dims, [int(x) for x in g[f"perm{k}_perm"]])
        assert np.array_equal(got, g[f"perm{k}_out"])
    # SWAP on two qubits exchanges the factors of a product operator
    swap = permute_tensor_factors(2, [1, 0])
    a, b = np.diag([1.0, 2.0]), np.array([[0.0, 1.0], [1.0, 0.0]])
    assert np.allclose(swap @ np.kron(a, b) @ swap.T, np.kron(b, a))


def test_partial_trace_outer_inner(g):
    from fbx.operator_tools import calculational as calc
    for k in range(6):
        got = calc.partial_trace(g["pt_in"], [int(x) for x in g[f"pt{k}_keep"]], [2, 3, 4])
        assert got.shape == g[f"pt{k}_out"].shape
        assert np.abs(got - g[f"pt{k}_out"]).max() < 1e-13
    assert np.abs(calc.outer_product(g["ket_a"], g["ket_b"]) - g["outer"]).max() < 1e-15
    assert np.abs(calc.inner_product(g["ket_a"], g["ket_b"]) - g["inner"]).max() < 1e-14
    with pytest.raises(ValueError):
        calc.outer_product(np.ones((1, 3)), np.ones((3, 1)))


def test_oracle_spectral_measures(g):
    from fbx_oracle import measures as om
    for d in (2, 4):
        for b in range(3):
            q, s = om.quantum_chernoff_bound(g[f"qcb{d}_rho"][b], g[f"qcb{d}_sigma"][b])
            assert abs(q - g[f"qcb{d}"][b, 0]) < 1e-9
    for name in ("herm16", "gen4", "gen16"):
        got = om.watrous_bounds(g[f"wat_{name}"])
        assert np.allclose(got, g[f"wat_{name}_out"], rtol=1e-12)


def test_diamond_norm_known_answers_of_the_reference():
    """distance_measures.py:378-437 with the SDP solved by fbx.distance_measures._watrous_sdp_value when cvxpy is absent
    (it is, in this image): the reference's own known answers (tests/test_distance_measures.py:186-218, rtol 1e-2 there), the
    two values its notebook records (docs/examples/distance_measures.ipynb cells 34-35), the closed form for a
    pair of unitaries on two qubits."""
    from scipy.linalg import fractional_matrix_power as matpow
    from fbx import distance_measures as dm
    X = np.array([[0, 1], [1, 0]], dtype=complex); Y = np.array([[0, -1j], [1j, 0]]); Z = np.diag([1.0 + 0j, -1.0])
    I = np.eye(2, dtype=complex); H = np.array([[1, 1], [1, -1]], dtype=complex) / np.sqrt(2)

    def kraus2choi(k):
        v = k.reshape(-1, 1, order="F")
        return v @ v.conj().T

    def kraus2superop(k):
        return np.kron(k.conj(), k)

    def superop2choi(sop, d=2):
        return sop.reshape([d] * 4).swapaxes(0, 3).reshape(d * d, d * d)

    assert np.isclose(dm.diamond_norm_distance(kraus2choi(I), kraus2choi(X)), 2.0, rtol=1e-6)
    for turns, target in [[1e-3, 3.141591e-3], [3.1e-3, 9.738899e-3], [1e-2, 3.141463e-2], [3.1e-2, 9.735089e-2],
                          [1e-1, 3.128689e-1], [3.1e-1, 9.358596e-1]]:
        assert np.isclose(dm.diamond_norm_distance(kraus2choi(X), kraus2choi(matpow(X, 1 + turns))), target, rtol=1e-5)
    for p, target in [[1e-3, 2e-3], [3.1e-3, 6.2e-3], [1e-2, 2e-2], [3.1e-2, 6.2e-2], [1e-1, 2e-1], [3.1e-1, 6.2e-1]]:
        c0 = superop2choi(kraus2superop(I) * (1 - p) + kraus2superop(H) * p)
        assert np.isclose(dm.diamond_norm_distance(c0, superop2choi(kraus2superop(I))), target, rtol=1e-6)
    assert np.isclose(dm.diamond_norm_distance(kraus2choi(I), kraus2choi(matpow(Y, 0.5))), np.sqrt(2), rtol=1e-6)
    # notebook: identity against exp(-0.2 i X) and against X
    from scipy.linalg import expm
    u = expm(-0.2j * X)
    assert np.isclose(dm.diamond_norm_distance(kraus2choi(I), kraus2choi(u)), 0.3973386615692544, rtol=1e-6)
    # two qubits: identity against exp(-i theta ZZ) -- eigenvalues exp(-+ i theta), distance 2 sin(theta)
    theta = 0.3
    u2 = expm(-1j * theta * np.kron(Z, Z))
    assert np.isclose(dm.diamond_norm_distance(kraus2choi(np.eye(4, dtype=complex)), kraus2choi(u2)), 2 * np.sin(theta), rtol=1e-5)
    # equal channels: zero
    assert abs(dm.diamond_norm_distance(kraus2choi(H), kraus2choi(H))) < 1e-9


def test_oracle_dfe_matches_reference(g):
    from fbx_oracle import acquisition as oa
    for n in (1, 2, 3):
        for kind in ("state", "process"):
            got = oa.estimate_dfe(g[f"dfe{n}_e"], g[f"dfe{n}_se"], n, kind)
            assert np.allclose(got, g[f"dfe{n}_{kind}"], rtol=1e-14, atol=0)
    with pytest.raises(ValueError):
        oa.estimate_dfe([0.1], [0.1], 1, "gate")


def test_results_by_qubit_groups_and_json_round_trip(tmp_path):
    """observable_estimation.py:1145-1173 with the reference's own example
    (tests/test_observable_estimation.py:1934-1965), and the JSON layout of :356-389, :721-733."""
    from fbx import observable_estimation as oe
    def res(ops, e):
        st = oe.zeros_state(sorted(q for _, q in ops))
        return oe.ExperimentResult(setting=oe.ExperimentSetting(st, oe.PauliTerm.from_list(ops)),
                                   expectation=e, std_err=0.01, total_counts=40)
    er1, er2, er3, er4 = res([("Z", 0)], 0.9), res([("Z", 1)], 0.8), res([("Z", 0), ("Z", 1)], 0.7), res([("X", 0), ("Z", 2)], 0.6)
    by = oe.get_results_by_qubit_groups([er1, er2, er3, er4], [(0,), (1,), (2, 0)])
    assert by == {(0,): [er1], (1,): [er2], (0, 2): [er1, er4]}
    fn = oe.to_json(str(tmp_path / "results.json"), [er1, er3, er4])
    import json
    raw = json.load(open(fn))
    assert raw[1]["type"] == "ExperimentResult" and raw[1]["setting"] == "Z+_0 * Z+_1→(1+0j)*Z0Z1"
    assert set(raw[0]) == {"type", "setting", "expectation", "std_err", "total_counts", "raw_expectation",
                           "raw_std_err", "calibration_expectation", "calibration_std_err", "calibration_counts"}
    back = oe.read_json(fn)
    assert back == [er1, er3, er4]
    assert str(oe.ExperimentSetting.from_str("X+_0 * SIC2_1→(1+0j)*X0Y1")) == "X+_0 * SIC2_1→(1+0j)*X0Y1"


def test_soa_archive_round_trip(tmp_path):
    from fbx import design as fd
    d = fd.process_design(1, "sic")
    rng = np.random.default_rng(0)
    e, c = rng.uniform(-1, 1, size=(3, d.m)), np.full((3, d.m), 500.0)
    fn = fd.save_batch(str(tmp_path / "batch.npz"), d, e, c)
    d2, e2, c2 = fd.load_batch(fn)
    assert d2.key() == d.key() and np.array_equal(e2, e) and np.array_equal(c2, c)
    fd.save_batch(fn, d, e)
    assert fd.load_batch(fn)[2] is None
