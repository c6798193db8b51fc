"""Host-side logic of the shim: designs, result flattening, synthetic data, sharding (CPU)."""
import numpy as np
import pytest


def test_canonical_designs_sizes_and_order():
    from fbx.design import process_design, state_design, traceless_pauli_codes
    assert [state_design(n).m for n in (1, 2, 3)] == [3, 15, 63]
    assert [process_design(n, "pauli").m for n in (1, 2)] == [18, 540]
    assert [process_design(n, "sic").m for n in (1, 2)] == [12, 240]
    p = traceless_pauli_codes(2)
    assert p[0].tolist() == [0, 1] and p[3].tolist() == [1, 0] and p[-1].tolist() == [3, 3]
    d = process_design(2, "pauli")
    assert d.in_labels[0].tolist() == [0, 0] and d.in_labels[15].tolist() == [0, 1]   # X+X+ then X+X-
    with pytest.raises(ValueError):
        process_design(1, "nope")


def test_flatten_results_matches_oracle_flattening():
    from fbx import tomography as T
    from fbx.design import flatten_results
    from fbx.observable_estimation import ExperimentResult
    from fbx_oracle import design as od
    qubits = [4, 1]
    settings = T.generate_process_tomography_settings(qubits, "sic")
    rs = np.random.RandomState(0)
    res = [ExperimentResult(s, rs.uniform(-1, 1), int(rs.randint(10, 99))) for s in settings]
    d, e, c = flatten_results(res, qubits, "process")
    o, oe_, oc = od.flatten_results(res, qubits, "process")
    assert (d.in_labels == o.in_labels).all() and (d.paulis == o.paulis).all()
    assert np.array_equal(e, oe_) and np.array_equal(c, oc)
    assert (d.in_labels == od.process_design(2, "sic").in_labels).all()
    d2, _, _ = flatten_results(res, qubits, "process")
    assert d2 is d                                   # cached by content


def test_setting_string_round_trip():
    from fbx.observable_estimation import ExperimentSetting, TensorProductState
    s = ExperimentSetting.from_str("X+_0 * Z-_1→(1+0j)*X0Z1")
    assert str(s.in_state) == "X+_0 * Z-_1" and s.observable[0] == "X" and s.observable[1] == "Z"
    assert s.observable[7] == "I"
    assert str(TensorProductState.from_str("SIC2_3")) == "SIC2_3"


def test_synthetic_expectations_are_exact_for_known_channels():
    from fbx import synthetic
    from fbx.design import process_design
    d = process_design(1, "pauli")
    e = synthetic.exact_process_expectations(d, np.eye(2)[None])
    # identity channel: <P> on the +-1 eigenstate of P is +-1, other Paulis 0
    want = []
    for s in range(6):
        for p in (1, 2, 3):
            axis = s // 2 + 1
            want.append((1 - 2 * (s % 2)) if axis == p else 0)
    assert np.allclose(e[0], want)
    ks = synthetic.kraus_batch(2, 4, 3)
    assert np.allclose(np.einsum("bkji,bkjl->bil", ks.conj(), ks), np.eye(4))


def test_shard_bounds_cover_everything_once():
    from fbx.parallel import shard_bounds
    for n in (0, 1, 7, 1024, 65536):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


def test_hinton_geometry_follows_the_reference_formulas():
    """plotting/hinton.py:12-36 (size, colour angle, automatic max_weight) and :52-118 (sign class, capped area,
    max_weight from the diagonal); values worked by hand."""
    from fbx import plotting
    m = np.array([[0.5, complex(0.0, -0.25)], [complex(0.0, 0.25), 3.0]])
    g = plotting.hinton_plot_inputs(m, max_weight=None)
    assert g["max_weight"] == 4.0                                   # next power of two above 3
    assert np.allclose(g["size"], np.sqrt(np.abs(m) / 4.0))
    assert np.allclose(g["angle"], [[np.pi / 2, np.pi], [0.0, np.pi / 2]])     # arctan2(re, im)
    assert g["xlim"] == (-2.0, 0.0) and plotting.hinton_plot_inputs(m)["max_weight"] == 1.0
    r = plotting.hinton_real_plot_inputs(np.array([[0.4, -0.1], [0.0, 0.2]]))
    assert r["max_weight"] == 0.5 and r["ticks"] == [-0.25, 0, 0.25]
    assert np.array_equal(r["sign"], [[1, -1], [-1, 1]])            # zero is drawn with the negative colour
    assert np.allclose(r["area"], [[0.8, 0.2], [0.0, 0.4]])
    assert plotting.hinton_real_plot_inputs(np.zeros((2, 2)))["max_weight"] == 1.0
    assert plotting.hinton_real_plot_inputs(np.eye(2), max_weight=0.5)["area"].max() == 1.0
    assert plotting.pauli_labels(2)[:6] == ["II", "IX", "IY", "IZ", "XI", "XX"]
    with pytest.raises(ValueError):
        plotting.pauli_labels(0)


def test_jacobi64_layout_is_conflict_free():
    """The LDS layout of the 64 x 64 Jacobi matrices (csrc/fbx_eigh.hpp sys_pos<64>, restated here): every b128 access
    of a round -- the permuted writes of the matrix and eigenvector blocks, the read-back of the own block, the pivot
    reads -- touches 8 distinct bank groups per group of 8 lanes; the row-major layout it replaces does not."""
    NB, PS = 32, 1024

    def seat(s):
        k = s >> 1
        if s & 1 == 0:
            return 0 if k == 0 else (2 * (NB - 1) + 1 if k == NB - 1 else 2 * (k + 1))
        return 2 if k == 0 else 2 * (k - 1) + 1

    def pos(I, K, e, new):
        if not new:
            return I * NB + K
        r = K & 7
        rho = (r ^ 1) if (K >= 8 and r < 2) else r
        return I * NB + ((K & ~7) | ((rho + 2 * (e & 1)) & 7))

    def extra_passes(groups):
        return sum(max(len({a for a in g if a % 8 == b}) for b in range(8)) - 1 for g in groups)

    for new in (False, True):
        assert len({(pl, pos(I, K, pl, new)) for pl in range(4) for I in range(NB) for K in range(NB)}) == 4 * NB * NB
        total = 0
        for e in range(4):
            a, b = e >> 1, e & 1
            mw, vw, own = [], [], []
            for I in range(NB):
                for J0 in range(0, NB, 8):
                    gm, gv = [], []
                    for J in range(J0, J0 + 8):
                        sa, sb = seat(2 * I + a), seat(2 * J + b)
                        pm, pv = (sa & 1) * 2 + (sb & 1), a * 2 + (sb & 1)
                        gm.append(pm * PS + pos(sa >> 1, sb >> 1, pm, new))
                        gv.append(pv * PS + pos(I, sb >> 1, pv, new))
                    mw.append(gm); vw.append(gv)
                    own.append([e * PS + pos(I, J, e, new) for J in range(J0, J0 + 8)])
            total += extra_passes(mw) + extra_passes(vw) + extra_passes(own)
        for pl in (0, 1, 3):
            total += extra_passes([[pl * PS + pos(J, J, pl, new) for J in range(J0, J0 + 8)] for J0 in range(0, NB, 8)])
        assert (total == 0) if new else (total == 384), total       # row-major: +37.5 % write passes
