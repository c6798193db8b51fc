"""Device eigensolver for 32 x 32 / 64 x 64, and the helpers that sit on it: sqrtm_psd, quantum
Chernoff bound, Watrous bounds, choi2kraus for three qubits -- against reference outputs
(tests/golden/extras.npz) and numpy."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def g():
    return np.load(os.path.join(GOLD, "extras.npz"))


@pytest.mark.parametrize("N", [2, 4, 8, 16, 32, 64])
def test_eigh_all_sizes(gpu, N):
    from fbx import _lib
    rng = np.random.default_rng(N)
    B = 5
    a = rng.normal(size=(B, N, N)) + 1j * rng.normal(size=(B, N, N))
    herm = a + a.conj().transpose(0, 2, 1)
    low = np.tril(herm) + 1j * np.triu(rng.normal(size=(B, N, N)), 1)      # upper triangle is garbage
    w, v = _lib.eigh_batch(low)
    for b in range(B):
        assert np.abs(w[b] - np.linalg.eigvalsh(herm[b])).max() < 1e-12 * N
        assert np.abs(v[b].conj().T @ v[b] - np.eye(N)).max() < 1e-12
        assert np.abs(herm[b] @ v[b] - v[b] * w[b]).max() < 1e-11 * N
    assert (np.diff(w, axis=1) >= 0).all()
    w_only = _lib.eigh_batch(low, eigenvectors=False)
    assert np.array_equal(w_only, w)
    degenerate = np.tile(np.eye(N)[None] * 3.0, (2, 1, 1))
    wd, vd = _lib.eigh_batch(degenerate)
    assert np.abs(wd - 3.0).max() == 0 and np.abs(vd - np.eye(N)).max() == 0


def test_eigh_64_structured_inputs(gpu):
    """The 64 x 64 workgroup solver (csrc/fbx_eigh64.hpp: rotations published by the thread that holds next round's pivot, records
    and work matrix in its private LDS layout) on inputs whose pivots are special: all-zero, rank one, block diagonal with
    repeated eigenvalues, off-diagonal entries whose squares underflow, 24 decades of dynamic range."""
    from fbx import _lib
    N = 64
    rng = np.random.default_rng(64)
    mats = [np.zeros((N, N), dtype=complex)]
    u = rng.normal(size=N) + 1j * rng.normal(size=N)
    mats.append(np.outer(u, u.conj()))                                            # rank one
    blk = rng.normal(size=(8, 8)) + 1j * rng.normal(size=(8, 8)); blk = blk + blk.conj().T
    mats.append(np.kron(np.eye(8), blk))                                          # every eigenvalue eight times
    d = np.diag(np.arange(N, dtype=float)).astype(complex)
    tiny = 1e-170 * (rng.normal(size=(N, N)) + 1j * rng.normal(size=(N, N)))
    mats.append(d + tiny + tiny.conj().T)                                         # |b|^2 underflows: identity rotations
    q, _ = np.linalg.qr(rng.normal(size=(N, N)) + 1j * rng.normal(size=(N, N)))
    spec = np.logspace(-12, 12, N) * np.where(np.arange(N) % 3 == 0, -1.0, 1.0)
    mats.append((q * spec) @ q.conj().T)
    a = np.stack([0.5 * (m + m.conj().T) for m in mats])
    w, v = _lib.eigh_batch(a)
    assert np.isfinite(w).all() and np.isfinite(v).all()
    assert (np.diff(w, axis=1) >= 0).all()
    for b in range(len(mats)):
        scale = max(1.0, np.abs(a[b]).max())
        assert np.abs(w[b] - np.linalg.eigvalsh(a[b])).max() < 1e-12 * N * scale, b
        assert np.abs(v[b].conj().T @ v[b] - np.eye(N)).max() < 1e-12, b
        assert np.abs(a[b] @ v[b] - v[b] * w[b]).max() < 1e-11 * N * scale, b
    assert np.abs(w[0]).max() == 0 and np.abs(v[0] - np.eye(N)).max() == 0        # nothing to rotate: the identity basis
    assert np.abs(w[3] - np.arange(N)).max() < 1e-150 * N + 1e-300                  # rotations of angle ~1e-170 leave the diagonal alone


def test_sqrtm_psd(gpu, g):
    from fbx.operator_tools import calculational as calc
    for N in (4, 16):
        got = calc.sqrtm_psd(g[f"psd{N}"])
        assert np.abs(got - g[f"sqrtm{N}"]).max() < 1e-11
        assert np.abs(got @ got - g[f"psd{N}"]).max() < 1e-10
    with pytest.raises(ValueError):
        calc.sqrtm_psd(np.array([[np.nan, 0], [0, 1.0]]))


def test_quantum_chernoff_bound(gpu, g):
    from fbx import distance_measures as dm
    for d in (2, 4):
        for b in range(3):
            q, s = dm.quantum_chernoff_bound(g[f"qcb{d}_rho"][b], g[f"qcb{d}_sigma"][b])
            assert abs(q - g[f"qcb{d}"][b, 0]) < 1e-9
            assert abs(s - g[f"qcb{d}"][b, 1]) < 1e-4          # flat minimum: argmin to optimiser tolerance
    # reference known answers (tests/test_distance_measures.py:118-141)
    rho = np.array([[1.0, 0.0], [0.0, 0.0]])
    psi = np.array([[np.cos(np.pi / 4)], [np.sin(np.pi / 4)]])
    assert np.allclose(dm.quantum_chernoff_bound(rho, psi @ psi.T)[0], 0.5)
    G = np.array([[0.0, -1.0], [1.0, 0.0]])
    r = np.diag([0.9, 0.1])
    assert np.allclose(dm.quantum_chernoff_bound(r, G @ r @ G.T)[0], 0.6)
    # complex states work too (the objective is real by construction)
    rng = np.random.default_rng(3)
    a = rng.normal(size=(4, 4)) + 1j * rng.normal(size=(4, 4)); a = a @ a.conj().T; a /= np.trace(a).real
    q, s = dm.quantum_chernoff_bound(a, a)
    assert abs(q - 1.0) < 1e-12


def test_watrous_bounds(gpu, g):
    from fbx import distance_measures as dm
    for name in ("herm16", "gen4", "gen16"):
        got = dm.watrous_bounds(g[f"wat_{name}"])
        assert np.allclose(got, g[f"wat_{name}_out"], rtol=1e-11)
    rng = np.random.default_rng(8)
    x = rng.normal(size=(64, 64)) + 1j * rng.normal(size=(64, 64))       # A^H A route
    assert np.isclose(dm.watrous_bounds(x)[0], np.linalg.svd(x, compute_uv=False).sum(), rtol=1e-9)
    h = x + x.conj().T
    assert np.isclose(dm.watrous_bounds(h)[0], np.abs(np.linalg.eigvalsh(h)).sum(), rtol=1e-11)
    with pytest.raises(ValueError):
        dm.watrous_bounds(np.ones((3, 3)))
    with pytest.raises(ValueError):
        dm.watrous_bounds(np.ones((4, 4, 4)))


def test_choi2kraus_three_qubits_round_trip(gpu):
    """superoperator_transformations.py:325-336 through the 64 x 64 device eigensolver; Kraus lists
    are only defined up to phases, so parity is on the rebuilt Choi matrix, as in the reference's
    own test (tests/test_superoperator_transformations.py:215-224)."""
    from fbx.operator_tools import choi2kraus, kraus2choi
    g3 = np.load(os.path.join(GOLD, "superops_3q.npz"))
    choi = g3["kraus4_choi"][0]
    ks = choi2kraus(choi)
    assert len(ks) == 4 and ks[0].shape == (8, 8)
    assert np.abs(kraus2choi(ks) - choi).max() < 1e-11


def test_dfe_estimate(gpu, g):
    from fbx import direct_fidelity_estimation as dfe
    from fbx import observable_estimation as oe
    for n in (1, 2, 3):
        e, se = g[f"dfe{n}_e"], g[f"dfe{n}_se"]
        for kind in ("state", "process"):
            mean, err = dfe.estimate_dfe_batch(np.tile(e, (4, 1)), np.tile(se, (4, 1)), n, kind)
            assert np.allclose(mean, g[f"dfe{n}_{kind}"][0], rtol=1e-14, atol=0)
            assert np.allclose(err, g[f"dfe{n}_{kind}"][1], rtol=1e-14, atol=0)
        qs = list(range(n))
        res = [oe.ExperimentResult(setting=oe.ExperimentSetting(oe.zeros_state(qs), oe.PauliTerm.from_list(
            [("XYZ"[(k + q) % 3], q) for q in qs])), expectation=float(e[k]), std_err=float(se[k]), total_counts=100)
            for k in range(len(e))]
        got = dfe.estimate_dfe(res, "process")
        assert np.allclose(got, g[f"dfe{n}_process"], rtol=1e-14, atol=0)
    big_e = np.random.default_rng(1).uniform(-1, 1, size=(1000, 4095))
    mean, err = dfe.estimate_dfe_batch(big_e, np.full_like(big_e, 0.03), 6, "state")
    assert np.allclose(mean, 1 / 64 + (63 / 64) * big_e.mean(axis=1), rtol=1e-13)
    assert np.allclose(err, (63 / 64) * 0.03 / np.sqrt(4095), rtol=1e-13)
    with pytest.raises(ValueError):
        dfe.estimate_dfe_batch(big_e[:1], big_e[:1], 2, "gate")


def test_estimate_by_qubit_groups_batches_pairs(gpu):
    """Two disjoint qubit pairs measured in one merged experiment: one batched PGDB call, results
    identical to the per-pair calls (the process notebook's loop over get_results_by_qubit_groups)."""
    from fbx import synthetic, tomography
    from fbx import observable_estimation as oe
    design, us, e, c = synthetic.process_batch(2, "sic", 2)
    results = []
    for b, pair in enumerate([(0, 1), (4, 5)]):
        for k, s in enumerate(tomography.generate_process_tomography_settings(list(pair), "sic")):
            results.append(oe.ExperimentResult(setting=s, expectation=float(e[b, k]), std_err=0.0,
                                               total_counts=int(c[b, k])))
    got = tomography.estimate_by_qubit_groups(results, [(0, 1), (5, 4)], kind="process")
    assert set(got) == {(0, 1), (4, 5)}
    want01 = tomography.pgdb_process_estimate([r for r in results[:design.m]], [0, 1])
    assert np.array_equal(got[(0, 1)], want01)
    want45 = tomography.pgdb_process_estimate([r for r in results[design.m:]], [4, 5])
    assert np.array_equal(got[(4, 5)], want45)
    lin = tomography.estimate_by_qubit_groups(results, [(0, 1)], kind="process", estimator="linear_inv")
    assert np.abs(lin[(0, 1)] - tomography.linear_inv_process_estimate(results[:design.m], [0, 1])).max() == 0


@pytest.mark.parametrize("N", [66, 100, 129, 256])
def test_eigh_beyond_lds_sizes(gpu, N):
    """64 < N <= 1024: the HBM-resident Jacobi (csrc/fbx_state.hip eigh_big_kernel) with numpy.linalg.eigh
    semantics -- lower triangle, ascending eigenvalues, A V = V diag(w), V unitary; odd sizes are padded."""
    from fbx import _lib
    rs = np.random.RandomState(N)
    g = rs.randn(2, N, N) + 1j * rs.randn(2, N, N)
    a = g + g.conj().transpose(0, 2, 1)
    a[1] = np.tril(a[1]) + 1j * 0.5 * np.triu(np.ones((N, N)), 1)      # garbage above the diagonal must be ignored
    w, v = _lib.eigh_batch(a)
    for b in range(2):
        low = np.tril(a[b]) + np.tril(a[b], -1).conj().T
        low[np.diag_indices(N)] = low[np.diag_indices(N)].real
        assert np.abs(w[b] - np.linalg.eigvalsh(low)).max() < 1e-11 * N
        assert np.all(np.diff(w[b]) >= 0)
        assert np.abs(low @ v[b] - v[b] * w[b]).max() < 1e-11 * N
        assert np.abs(v[b].conj().T @ v[b] - np.eye(N)).max() < 1e-12 * N
    if N >= 128:                        # the cooperative form was used above; one workgroup per matrix must agree
        with _lib.option("eigh_cooperative", 0):
            w1 = _lib.eigh_batch(a, eigenvectors=False)
        assert np.abs(w1 - w).max() < 1e-11 * N


def test_four_qubit_choi_validators_and_kraus(gpu):
    """What the large eigensolver unlocks: CP / CPTP checks and choi2kraus of a 256 x 256 Choi matrix, chi from a Choi."""
    from fbx.operator_tools import (choi2kraus, choi_is_completely_positive, choi_is_cptp, kraus2choi)
    rs = np.random.RandomState(44)
    u, _ = np.linalg.qr(rs.randn(16, 16) + 1j * rs.randn(16, 16))
    choi = kraus2choi(u)
    assert choi_is_completely_positive(choi) and choi_is_cptp(choi)
    ops = choi2kraus(choi)
    assert len(ops) == 1
    phase = np.vdot(u, ops[0]) / 16
    assert abs(abs(phase) - 1) < 1e-10 and np.abs(ops[0] - phase * u).max() < 1e-9
    assert not choi_is_completely_positive(choi - 0.1 * np.eye(256))
    from fbx.operator_tools import choi_is_trace_preserving, choi_is_unital, rand_map_with_BCSZ_dist
    from fbx.operator_tools.calculational import partial_trace
    assert choi_is_unital(choi) and not choi_is_trace_preserving(choi + 0.01 * np.eye(256))
    qutrit = rand_map_with_BCSZ_dist(3, 2)                           # 9 x 9: neither a qubit system nor a power of two
    assert choi_is_trace_preserving(qutrit) and choi_is_cptp(qutrit) and not choi_is_unital(qutrit)
    rho = rs.randn(12, 12) + 1j * rs.randn(12, 12)                   # operator on C^3 (x) C^4: both device partial traces
    t = rho.reshape(3, 4, 3, 4)
    assert np.abs(partial_trace(rho, [0], [3, 4]) - np.einsum("ajbj->ab", t)).max() < 1e-14
    assert np.abs(partial_trace(rho, [1], [3, 4]) - np.einsum("iaib->ab", t)).max() < 1e-14


def test_state_measures_for_four_and_five_qubits(gpu):
    """16 x 16 and 32 x 32 states: purity, fidelity, trace distance, Hilbert-Schmidt product and the Bures quantities
    against the oracle's restatement of the reference formulas; sqrtm_psd and the device matmul on the way."""
    from fbx import distance_measures as dm, _lib
    from fbx.operator_tools.calculational import sqrtm_psd
    from fbx_oracle import measures as om
    rs = np.random.RandomState(16)
    for d in (16, 32):
        g = rs.randn(2, d, d + 3) + 1j * rs.randn(2, d, d + 3)
        rho, sigma = (x @ x.conj().T for x in g)
        rho /= np.trace(rho).real; sigma /= np.trace(sigma).real
        assert abs(dm.purity(rho) - om.purity(rho)) < 1e-13
        assert abs(dm.purity(rho, dim_renorm=True) - om.purity(rho, dim_renorm=True)) < 1e-12
        assert abs(dm.fidelity(rho, sigma) - om.fidelity(rho, sigma)) < 1e-11
        assert abs(dm.fidelity(rho, rho) - 1.0) < 1e-11
        assert abs(dm.trace_distance(rho, sigma) - om.trace_distance(rho, sigma)) < 1e-14
        assert abs(dm.hilbert_schmidt_ip(rho, sigma) - om.hilbert_schmidt_ip(rho, sigma)) < 1e-13
        assert abs(dm.bures_distance(rho, sigma) - om.bures_distance(rho, sigma)) < 1e-10
        root = sqrtm_psd(rho)
        assert np.abs(root @ root - rho).max() < 1e-12
    a = rs.randn(3, 70, 70) + 1j * rs.randn(3, 70, 70)
    b = rs.randn(3, 70, 70) + 1j * rs.randn(3, 70, 70)
    sc = rs.rand(3, 70)
    assert np.abs(_lib.matmul_batch(a, b) - a @ b).max() < 1e-11
    want = np.einsum("bki,bk,bjk->bij", a.conj(), sc, b.conj())                  # A^H diag(s) B^H
    assert np.abs(_lib.matmul_batch(a, b, conj_t_a=True, conj_t_b=True, scale=sc) - want).max() < 1e-11


def test_channel_application_and_state_projection_beyond_three_qubits(gpu):
    """apply_choi_matrix_2_state / apply_kraus_ops_2_state on 16-dimensional states and
    project_state_matrix_to_physical for a qutrit, 4 and 5 qubits, against the oracle."""
    from fbx.operator_tools import apply_choi_matrix_2_state, kraus2choi
    from fbx.operator_tools.project_state_matrix import project_state_matrix_to_physical
    from fbx_oracle import superops as so
    rs = np.random.RandomState(23)
    ks = rs.randn(2, 16, 16) + 1j * rs.randn(2, 16, 16)
    g = rs.randn(16, 16) + 1j * rs.randn(16, 16)
    rho = g @ g.conj().T; rho /= np.trace(rho)
    choi = kraus2choi(list(ks))
    want = sum(k @ rho @ k.conj().T for k in ks)
    assert np.abs(apply_choi_matrix_2_state(choi, rho) - want).max() < 1e-11
    assert np.abs(so.apply_choi_matrix_2_state(choi, rho) - want).max() < 1e-11
    for d in (3, 16, 32):
        h = rs.randn(d, d) + 1j * rs.randn(d, d)
        h = h + h.conj().T
        h = h / np.trace(h).real * 1.0
        got = project_state_matrix_to_physical(h)
        assert np.abs(got - so.project_state_matrix_to_physical(h)).max() < 1e-11
        w = np.linalg.eigvalsh(got)
        assert w.min() > -1e-12 and abs(np.trace(got) - 1) < 1e-12
        good = h @ h.conj().T
        assert np.abs(project_state_matrix_to_physical(good) - good / np.trace(good)).max() < 1e-14     # physical: only rescaled


def test_choi_projections_beyond_three_qubits_and_for_a_qutrit(gpu):
    """CP / TP / TNI / physical (Dykstra) projections of 9 x 9 and 256 x 256 Choi matrices against the oracle, with
    the oracle's Dykstra iteration counts.  (A 256 x 256 eigendecomposition takes 0.1 s on the one CU it runs on, so
    the 4-qubit Dykstra case is one run from a mildly perturbed channel.)"""
    from fbx.operator_tools import project_superoperators as ps
    from fbx import _lib
    from fbx_oracle import superops as so
    rs = np.random.RandomState(31)
    for d, noise in ((3, 0.05), (16, 1e-3)):
        D = d * d
        k = rs.randn(2, d, d) + 1j * rs.randn(2, d, d)
        w, v = np.linalg.eigh(sum(q.conj().T @ q for q in k))
        k = k @ (v @ np.diag(w ** -0.5) @ v.conj().T)                # a CPTP pair of Kraus operators
        x = so.kraus2choi(list(k)) + noise * (rs.randn(D, D) + 1j * rs.randn(D, D))
        assert np.abs(ps.proj_choi_to_completely_positive(x) - so.proj_choi_to_completely_positive(x)).max() < 1e-11
        assert np.abs(ps.proj_choi_to_trace_preserving(x) - so.proj_choi_to_trace_preserving(x)).max() < 1e-13
        assert np.abs(ps.proj_choi_to_trace_non_increasing(x) - so.proj_choi_to_trace_non_increasing(x)).max() < 1e-11
        for tp, kind in ((True, _lib.PROJ_PHYSICAL_TP), (False, _lib.PROJ_PHYSICAL_TNI)):
            if d > 3 and not tp:
                continue
            want, it = so.proj_choi_to_physical(x, tp, return_iters=True)
            got, its = ps.proj_choi_batch(kind, x[None], return_iters=True)
            assert its[0] == it and np.abs(got[0] - want).max() < 1e-10
            assert d == 3 or it < 40


def test_generic_dykstra_runs_the_batch_in_lockstep(gpu):
    """Choi projections outside the fused kernels (qutrits, 4-5 qubits) advance the whole batch per Dykstra iteration -- three device
    calls per iteration, not per iteration and item.  Items leave when their own stopping rule fires: every item must equal its
    one-at-a-time result bit for bit, with its own iteration count (the oracle's)."""
    from fbx.operator_tools import project_superoperators as ps
    from fbx import _lib
    from fbx_oracle import superops as so
    rs = np.random.RandomState(77)
    d, D = 3, 9
    xs = []
    for noise in (0.02, 0.3, 0.08, 1.0, 0.0, 0.15):
        k = rs.randn(2, d, d) + 1j * rs.randn(2, d, d)
        w, v = np.linalg.eigh(sum(q.conj().T @ q for q in k))
        k = k @ (v @ np.diag(w ** -0.5) @ v.conj().T)
        xs.append(so.kraus2choi(list(k)) + noise * (rs.randn(D, D) + 1j * rs.randn(D, D)))
    xs = np.array(xs)
    for kind, tp in ((_lib.PROJ_PHYSICAL_TP, True), (_lib.PROJ_PHYSICAL_TNI, False)):
        got, its = ps.proj_choi_batch(kind, xs, return_iters=True)
        assert len(set(its.tolist())) >= 3                                         # the items really leave at different iterations
        for b in range(len(xs)):
            one, it1 = ps.proj_choi_batch(kind, xs[b][None], return_iters=True)
            assert np.array_equal(one[0], got[b]) and it1[0] == its[b]
            want, it = so.proj_choi_to_physical(xs[b], tp, return_iters=True)
            assert its[b] == it and np.abs(got[b] - want).max() < 1e-10
