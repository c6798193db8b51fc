"""HIP superoperator tools vs the golden vectors produced by the reference (tests/golden)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
TOL = 1e-12


def gold(n):
    return np.load(os.path.join(GOLD, f"superops_{n}q.npz"))


@pytest.mark.parametrize("n", [1, 2, 3])
def test_kraus_conversions(gpu, n):
    from fbx.operator_tools import convert_batch
    g = gold(n)
    for K in (1, 2, 4):
        ks = g[f"kraus{K}"]
        for dst, key in (("choi", "choi"), ("superop", "superop"), ("pauli_liouville", "ptm"), ("chi", "chi")):
            got = convert_batch("kraus", dst, ks)
            assert np.abs(got - g[f"kraus{K}_{key}"]).max() < TOL, (K, dst)


@pytest.mark.parametrize("n", [1, 2, 3])
def test_pairwise_conversions(gpu, n):
    from fbx.operator_tools import convert_batch
    g = gold(n)
    cases = [("choi", "chi", "kraus4_choi", "choi2chi"), ("choi", "superop", "kraus4_choi", "choi2superop"),
             ("choi", "pauli_liouville", "kraus4_choi", "choi2ptm"),
             ("chi", "choi", "kraus4_chi", "chi2choi"), ("chi", "pauli_liouville", "kraus4_chi", "chi2ptm"),
             ("chi", "superop", "kraus4_chi", "chi2superop"),
             ("superop", "choi", "kraus4_superop", "superop2choi"),
             ("superop", "pauli_liouville", "kraus4_superop", "superop2ptm"),
             ("superop", "chi", "kraus4_superop", "superop2chi"),
             ("pauli_liouville", "choi", "kraus4_ptm", "ptm2choi"),
             ("pauli_liouville", "superop", "kraus4_ptm", "ptm2superop"),
             ("pauli_liouville", "chi", "kraus4_ptm", "ptm2chi")]
    for src, dst, kin, kout in cases:
        got = convert_batch(src, dst, g[kin])
        assert np.abs(got - g[kout]).max() < 1e-11, (src, dst)


def test_pairwise_conversions_3q_register_passes_against_the_first_form(gpu):
    """The 3-qubit routes between Choi / superoperator / Pauli-Liouville run two butterfly stages per pass in registers
    (convert3_regs_kernel, csrc/fbx_superop.hip); the one-stage-per-pass kernels stay behind FBX_CONVERT3_V1=1.  Arbitrary
    complex matrices (the conversions are linear maps: no structure of the input may be assumed), a batch larger than the
    persistent grid: both forms agree to rounding, the reshuffle routes bit for bit where no arithmetic is involved, and
    every route followed by its inverse returns the input."""
    from fbx.operator_tools import convert_batch
    rs = np.random.RandomState(5)
    B = 2048 + 77
    x = rs.randn(B, 64, 64) + 1j * rs.randn(B, 64, 64)

    def both(src, dst, a):
        new = convert_batch(src, dst, a)
        os.environ["FBX_CONVERT3_V1"] = "1"
        try:
            old = convert_batch(src, dst, a)
        finally:
            os.environ.pop("FBX_CONVERT3_V1")
        return new, old
    for src, dst in (("choi", "pauli_liouville"), ("superop", "pauli_liouville"), ("pauli_liouville", "choi"), ("pauli_liouville", "superop")):
        new, old = both(src, dst, x)
        scale = np.abs(old).max()
        assert np.abs(new - old).max() < 1e-14 * scale, (src, dst)
        back = convert_batch(dst, src, new)
        assert np.abs(back - x).max() < 1e-13 * np.abs(x).max(), (src, dst)


@pytest.mark.parametrize("n", [1, 2, 3])
def test_choi2chi_non_cp_goes_through_abs(gpu, n):
    """Reference quirk: choi2chi of a non-CP matrix is the chi form of |C|."""
    from fbx.operator_tools import convert_batch
    g = gold(n)
    got = convert_batch("choi", "chi", g["herm"])
    assert np.abs(got - g["herm_choi2chi"]).max() < 1e-11


@pytest.mark.parametrize("n", [1, 2, 3])
def test_projections(gpu, n):
    from fbx import _lib
    from fbx.operator_tools.project_superoperators import proj_choi_batch
    g = gold(n)
    for name in ("gen", "near"):
        x = g[f"proj_{name}_in"]
        assert np.abs(proj_choi_batch(_lib.PROJ_CP, x) - g[f"proj_{name}_cp"]).max() < 1e-11
        assert np.abs(proj_choi_batch(_lib.PROJ_TP, x) - g[f"proj_{name}_tp"]).max() < 1e-12
        assert np.abs(proj_choi_batch(_lib.PROJ_TNI, x) - g[f"proj_{name}_tni"]).max() < 1e-11
    x = g["proj_near_in"]
    assert np.abs(proj_choi_batch(_lib.PROJ_PHYSICAL_TP, x) - g["proj_near_phys_tp"]).max() < 1e-10
    assert np.abs(proj_choi_batch(_lib.PROJ_PHYSICAL_TNI, x) - g["proj_near_phys_tni"]).max() < 1e-10


@pytest.mark.parametrize("n", [1, 2, 3])
def test_dykstra_iteration_counts_match_oracle(gpu, n):
    from fbx import _lib
    from fbx.operator_tools.project_superoperators import proj_choi_batch
    from fbx_oracle import superops as so
    g = gold(n)
    x = g["proj_near_in"]
    _, iters = proj_choi_batch(_lib.PROJ_PHYSICAL_TP, x, return_iters=True)
    want = [so.proj_choi_to_physical(v, True, return_iters=True)[1] for v in x]
    assert list(iters) == want


@pytest.mark.parametrize("n", [1, 2, 3])
def test_process_fidelity_and_apply(gpu, n):
    from fbx import distance_measures as dm
    from fbx.operator_tools import apply_choi_matrix_2_state_batch
    g = gold(n)
    fp = dm.process_fidelity_batch(g["ptm_ref"][None], g["kraus4_ptm"])
    fe = dm.process_fidelity_batch(g["ptm_ref"][None], g["kraus4_ptm"], entanglement=True)
    assert np.abs(fp - g["proc_fid"]).max() < 1e-13
    assert np.abs(fe - g["ent_fid"]).max() < 1e-13
    out = apply_choi_matrix_2_state_batch(g["kraus4_choi"], g["rho"])
    assert np.abs(out - g["apply_choi"]).max() < 1e-13


@pytest.mark.parametrize("n", [1, 2, 3])
def test_kraus_sweep_fused(gpu, n):
    """BASELINE config 3 pipeline: kraus -> choi -> PTM -> chi + process_fidelity, one kernel (1 and 2
    qubits; for 3 qubits the same entry point composes the pairwise 64 x 64 conversions)."""
    import ctypes
    from fbx import _lib
    g = gold(n)
    ks = np.ascontiguousarray(g["kraus4"])
    B, D = ks.shape[0], 4 ** n
    ref = np.ascontiguousarray(g["ptm_ref"], dtype=np.complex128)
    choi = np.empty((B, D, D), complex); ptm = np.empty_like(choi); chi = np.empty_like(choi)
    fid = np.empty(B)
    _lib.check(_lib.lib().fbx_kraus_sweep(n, B, 4, _lib.dptr(ks.view(np.float64)), _lib.dptr(ref.view(np.float64)),
                                          _lib.dptr(choi.view(np.float64)), _lib.dptr(ptm.view(np.float64)),
                                          _lib.dptr(chi.view(np.float64)), _lib.dptr(fid)))
    assert np.abs(choi - g["kraus4_choi"]).max() < TOL
    assert np.abs(ptm - g["kraus4_ptm"]).max() < TOL
    assert np.abs(chi - g["choi2chi"]).max() < 1e-11        # reference's eigh route, CP input
    assert np.abs(fid - g["proc_fid"]).max() < 1e-13
    # fidelity only (no matrix leaves the device) gives the same numbers
    fid2 = np.empty(B)
    _lib.check(_lib.lib().fbx_kraus_sweep(n, B, 4, _lib.dptr(ks.view(np.float64)), _lib.dptr(ref.view(np.float64)),
                                          None, None, None, _lib.dptr(fid2)))
    assert np.abs(fid2 - fid).max() < 1e-15


def test_kraus_sweep_3q_fused_kernel_against_reference_fixtures(gpu):
    """The fused 3-qubit sweep (csrc/fbx_superop.hip sweep3_regs_kernel: kraus -> superoperator -> Pauli-Liouville + fidelity,
    kraus -> Choi -> chi through one 64 x 64 LDS matrix) on six random CPTP Kraus sets against what the reference's
    kraus2choi / kraus2pauli_liouville / kraus2chi / process_fidelity returned (tests/golden/make_goldens.py --sweep3q);
    every subset of outputs gives the same numbers, and a batch larger than the persistent grid (2048 workgroups) repeats
    them item for item."""
    from fbx import _lib
    g = np.load(os.path.join(GOLD, "sweep_3q.npz"))
    ks = np.ascontiguousarray(g["kraus4"])
    B, D = ks.shape[0], 64
    assert B >= 6
    ref = np.ascontiguousarray(g["ptm_ref"].astype(np.complex128))

    def run(kraus, want):
        nb = kraus.shape[0]
        outs = {k: np.empty((nb, D, D), complex) for k in ("choi", "ptm", "chi") if k in want}
        fid = np.empty(nb) if "fid" in want else None
        _lib.check(_lib.lib().fbx_kraus_sweep(3, nb, 4, _lib.dptr(kraus.view(np.float64)), _lib.dptr(ref.view(np.float64)),
                                              *[_lib.dptr(outs[k].view(np.float64)) if k in outs else None for k in ("choi", "ptm", "chi")],
                                              _lib.dptr(fid) if fid is not None else None))
        return outs, fid

    outs, fid = run(ks, ("choi", "ptm", "chi", "fid"))
    assert np.abs(outs["choi"] - g["choi"]).max() < 1e-13
    assert np.abs(outs["ptm"] - g["ptm"]).max() < 1e-13
    assert np.abs(outs["chi"] - g["chi"]).max() < 1e-13
    assert np.abs(fid - g["proc_fid"]).max() < 1e-13
    for want in (("fid",), ("chi",), ("ptm", "fid"), ("choi",)):
        o2, f2 = run(ks, want)
        assert all(np.array_equal(o2[k], outs[k]) for k in o2) and (f2 is None or np.array_equal(f2, fid))
    big = np.ascontiguousarray(np.tile(ks, (400, 1, 1, 1)))                # 2400 items > 2048 workgroups
    ob, fb = run(big, ("ptm", "chi", "fid"))
    assert np.array_equal(ob["chi"].reshape(400, B, D, D), np.broadcast_to(outs["chi"], (400, B, D, D)))
    assert np.array_equal(fb.reshape(400, B), np.broadcast_to(fid, (400, B)))


def test_reference_signature_wrappers(gpu):
    """Same names / call shapes as forest.benchmarking.operator_tools on single matrices."""
    from fbx import operator_tools as ot
    g = gold(1)
    ks = list(g["kraus2"][0])
    assert np.abs(ot.kraus2choi(ks) - g["kraus2_choi"][0]).max() < TOL
    assert np.abs(ot.kraus2choi(ks[0]) - ot.kraus2choi([ks[0]])).max() == 0      # single-ndarray form
    choi = g["kraus4_choi"][0]
    assert np.abs(ot.choi2pauli_liouville(choi) - g["choi2ptm"][0]).max() < 1e-12
    assert np.abs(ot.proj_choi_to_physical(g["proj_near_in"][0]) - g["proj_near_phys_tp"][0]).max() < 1e-10
    p2c = ot.pauli2computational_basis_matrix(2)
    want = np.array([[1, 0, 0, 1], [0, 1, 1j, 0], [0, 1, -1j, 0], [1, 0, 0, -1]])
    assert np.abs(p2c - want).max() < 1e-15
    with pytest.raises(ValueError):
        ot.convert_batch("choi", "chi", np.zeros((1, 3, 3)))


@pytest.mark.parametrize("B,K", [(1, 4), (5, 1), (67, 3), (2051, 9), (3, 31)])
def test_fused_sweep_3q_kraus_counts_and_both_forms(gpu, B, K):
    """The 3-qubit sweep kernel (two butterfly stages per pass in registers, 256 threads) for 1 to 31 Kraus operators and
    batches on either side of the persistent grid: equal to the pairwise 64 x 64 conversions and to the round's first form
    of the kernel (one stage per pass through LDS, FBX_SWEEP3_V1=1) to rounding."""
    from fbx import _lib, synthetic
    from fbx import distance_measures as dm
    from fbx.operator_tools import convert_batch
    n, D = 3, 64
    nd = min(B, 8)
    ks = np.ascontiguousarray(synthetic.kraus_batch(n, K, nd, seed=K)[np.arange(B) % nd])
    ref = np.ascontiguousarray(convert_batch("kraus", "pauli_liouville", synthetic.kraus_batch(n, 1, 1, seed=99))[0])

    def run():
        choi = np.empty((B, D, D), complex); ptm = np.empty_like(choi); chi = np.empty_like(choi); fid = np.empty(B)
        _lib.check(_lib.lib().fbx_kraus_sweep(n, B, K, _lib.dptr(ks.view(np.float64)), _lib.dptr(ref.view(np.float64)),
                                              _lib.dptr(choi.view(np.float64)), _lib.dptr(ptm.view(np.float64)),
                                              _lib.dptr(chi.view(np.float64)), _lib.dptr(fid)))
        return choi, ptm, chi, fid
    choi, ptm, chi, fid = run()
    old = os.environ.get("FBX_SWEEP3_V1")
    os.environ["FBX_SWEEP3_V1"] = "1"
    try:
        choi1, ptm1, chi1, fid1 = run()
    finally:
        if old is None:
            os.environ.pop("FBX_SWEEP3_V1")
        else:
            os.environ["FBX_SWEEP3_V1"] = old
    assert np.array_equal(choi, choi1)                      # the same sums in the same order
    assert np.abs(ptm - ptm1).max() < 1e-14 and np.abs(chi - chi1).max() < 1e-14 and np.abs(fid - fid1).max() < 1e-14
    head = slice(0, nd)
    os.environ["FBX_CONVERT3_V1"] = "1"        # the general pairwise kernels (by default kraus -> X is this very sweep kernel)
    try:
        assert np.abs(choi[head] - convert_batch("kraus", "choi", ks[head])).max() < 1e-13
        assert np.abs(ptm[head] - convert_batch("kraus", "pauli_liouville", ks[head])).max() < 1e-13
        assert np.abs(chi[head] - convert_batch("kraus", "chi", ks[head])).max() < 1e-13
    finally:
        os.environ.pop("FBX_CONVERT3_V1")
    # ... and the default pairwise route from Kraus operators (the sweep kernel with one output) gives the sweep's numbers
    assert np.array_equal(convert_batch("kraus", "pauli_liouville", ks[head]), ptm[head])
    assert np.array_equal(convert_batch("kraus", "chi", ks[head]), chi[head])
    assert np.array_equal(convert_batch("kraus", "choi", ks[head]), choi[head])
    assert np.abs(fid[head] - dm.process_fidelity_batch(ref[None], ptm[head])).max() < 1e-13
    rep = np.arange(B) % nd
    assert np.array_equal(ptm, ptm[rep]) and np.array_equal(chi, chi[rep]) and np.array_equal(fid, fid[rep])


@pytest.mark.parametrize("B,K", [(1, 4), (7, 1), (129, 3), (4097, 16)])
def test_fused_sweep_odd_batches_and_kraus_counts(gpu, B, K):
    """The paired sweep kernel takes two Kraus sets per wavefront: odd batches, a single item and up to
    16 Kraus operators must give what the pairwise conversions give."""
    from fbx import _lib, synthetic
    from fbx import distance_measures as dm
    from fbx.operator_tools import convert_batch
    n, D = 2, 16
    ks = np.ascontiguousarray(synthetic.kraus_batch(n, K, min(B, 64), seed=K)[np.arange(B) % min(B, 64)])
    ref = convert_batch("kraus", "pauli_liouville", synthetic.kraus_batch(n, 1, 1, seed=99))
    choi = np.empty((B, D, D), complex); ptm = np.empty_like(choi); chi = np.empty_like(choi); fid = np.empty(B)
    _lib.check(_lib.lib().fbx_kraus_sweep(n, B, K, _lib.dptr(ks.view(np.float64)), _lib.dptr(np.ascontiguousarray(ref[0]).view(np.float64)),
                                          _lib.dptr(choi.view(np.float64)), _lib.dptr(ptm.view(np.float64)),
                                          _lib.dptr(chi.view(np.float64)), _lib.dptr(fid)))
    assert np.abs(choi - convert_batch("kraus", "choi", ks)).max() < 1e-13
    assert np.abs(ptm - convert_batch("kraus", "pauli_liouville", ks)).max() < 1e-13
    assert np.abs(chi - convert_batch("kraus", "chi", ks)).max() < 1e-13
    assert np.abs(fid - dm.process_fidelity_batch(ref, ptm)).max() < 1e-13


def test_full_size_sweep_properties(gpu):
    """BASELINE config 3 size (1e6 two-qubit Kraus sets, K = 4), device-resident: size-independent checks --
    tiled inputs give bit-identical outputs, a sample of items matches the pairwise conversions, every
    Choi matrix has trace d (trace preservation of the CPTP inputs), chi has trace 1, PTM[0][0] = 1, and
    the process fidelities lie in [1/(d+1), 1]."""
    import ctypes
    from fbx import _lib, synthetic
    from fbx.operator_tools import convert_batch
    n, K, D, B, T = 2, 4, 16, 1_000_000, 4096
    base = synthetic.kraus_batch(n, K, T, seed=3)
    ks = np.ascontiguousarray(np.tile(base, (B // T + 1, 1, 1, 1))[:B])
    ref = convert_batch("kraus", "pauli_liouville", synthetic.kraus_batch(n, 1, 1, seed=4))
    lib = _lib.lib()
    d_k = _lib.DeviceBuffer.from_array(ks); d_r = _lib.DeviceBuffer.from_array(np.ascontiguousarray(ref[0]))
    d_c, d_p, d_x = (_lib.DeviceBuffer(B * D * D * 16) for _ in range(3))
    d_f = _lib.DeviceBuffer(B * 8)
    _lib.check(lib.fbx_kraus_sweep_dev(n, B, K, d_k.ptr, d_r.ptr, d_c.ptr, d_p.ptr, d_x.ptr, d_f.ptr))
    _lib.synchronize()
    fid = d_f.to_array(np.float64, (B,))
    assert fid.min() >= 1 / 5 - 1e-12 and fid.max() <= 1 + 1e-12
    assert np.array_equal(fid[:T], fid[T:2 * T]) and np.array_equal(fid[:B % T], fid[B - B % T:])
    for buf, name in ((d_c, "choi"), (d_p, "pauli_liouville"), (d_x, "chi")):
        out = buf.to_array(np.complex128, (B, D, D))
        assert np.array_equal(out[:T], out[-(B % T) - T:-(B % T)]) , name            # last full tile == first tile
        sample = np.r_[0, 1, 4095, 4096, 123457, B - 2, B - 1]
        want = convert_batch("kraus", name, ks[sample])
        assert np.abs(out[sample] - want).max() < 1e-13, name
        # ... and the ORACLE (the reference's functions restated, pinned to the reference in tests/test_oracle_vs_reference.py)
        # on the same sample: the fused kernel itself against the reference's arithmetic, not against another kernel
        from fbx_oracle import superops as so
        f = {"choi": so.kraus2choi, "pauli_liouville": so.kraus2pauli_liouville, "chi": so.kraus2chi}[name]
        for k, b in enumerate(sample):
            assert np.abs(out[b] - f(list(ks[b]))).max() < 1e-13, (name, b)
            if name == "pauli_liouville":
                from fbx_oracle import measures as om
                assert abs(fid[b] - om.process_fidelity(ref[0], out[b])) < 1e-13
        tr = np.trace(out[::997], axis1=1, axis2=2)
        if name == "choi":
            assert np.abs(tr - 4).max() < 1e-12
        elif name == "chi":
            assert np.abs(tr - 1).max() < 1e-12
        else:
            assert np.abs(out[::997, 0, 0] - 1).max() < 1e-12
        del out


# ------------------------------------------------------------------ beyond three qubits / beyond qubits
def test_four_qubit_conversions_match_the_oracle(gpu):
    """256 x 256 superoperators (work matrices in HBM, csrc/fbx_superop.hip convert_big_kernel): every pairwise
    conversion that does not need the 256 x 256 eigendecomposition of choi2kraus, against the oracle's dense
    basis-change matrices."""
    from fbx.operator_tools import superoperator_transformations as st
    from fbx_oracle import superops as so
    rs = np.random.RandomState(4)
    g = rs.randn(2, 2, 16, 16) + 1j * rs.randn(2, 2, 16, 16)
    s = sum(k.conj().T @ k for k in g[0])
    w, v = np.linalg.eigh(s)
    fix = v @ np.diag(w ** -0.5) @ v.conj().T
    kraus = np.array([[k @ fix for k in g[0]], [k @ fix for k in g[0][::-1]]])          # two CPTP sets
    reps = {"choi": np.array([so.kraus2choi(list(k)) for k in kraus]), "superop": np.array([so.kraus2superop(list(k)) for k in kraus]),
            "pauli_liouville": np.array([so.kraus2pauli_liouville(list(k)) for k in kraus]), "chi": np.array([so.kraus2chi(list(k)) for k in kraus])}
    for dst in ("choi", "superop", "pauli_liouville", "chi"):
        got = st.convert_batch("kraus", dst, kraus)
        assert np.abs(got - reps[dst]).max() < 1e-11, ("kraus", dst)
    for src in ("choi", "superop", "pauli_liouville", "chi"):
        for dst in ("choi", "superop", "pauli_liouville"):
            if src != dst:
                got = st.convert_batch(src, dst, reps[src])
                assert np.abs(got - reps[dst]).max() < 1e-11, (src, dst)
    assert np.abs(reps["pauli_liouville"].imag).max() < 1e-12                            # a PTM is real
    # into chi from a Choi / superoperator / Pauli-Liouville matrix: the reference goes through choi2kraus (a 256 x 256
    # eigendecomposition); batched since round 5 (fbx_eigh_dev + V |lambda| V^H + the linear basis change, all resident)
    for src in ("choi", "superop", "pauli_liouville"):
        got = st.convert_batch(src, "chi", reps[src])
        assert np.abs(got - reps["chi"]).max() < 1e-9, (src, "chi")
    assert np.abs(st.choi2chi(reps["choi"][0]) - reps["chi"][0]).max() < 1e-9
    assert np.abs(st.pauli_liouville2chi(reps["pauli_liouville"][1]) - reps["chi"][1]).max() < 1e-9
    # a Hermitian, NON-CP "Choi" matrix: the reference's quirk (chi of |C|, eigenvalues within 1e-9 dropped) -- against the oracle
    h = rs.randn(256, 256) + 1j * rs.randn(256, 256)
    h = (h + h.conj().T) / 32
    assert np.abs(st.convert_batch("choi", "chi", h[None])[0] - so.choi2chi(h)).max() < 1e-9
    # more Kraus operators than the fused kernel stages in LDS (40 for 4 qubits): the basis-free kernel takes over
    many = (rs.randn(1, 50, 16, 16) + 1j * rs.randn(1, 50, 16, 16)) / 30
    for dst, f in (("choi", so.kraus2choi), ("pauli_liouville", so.kraus2pauli_liouville), ("chi", so.kraus2chi)):
        assert np.abs(st.convert_batch("kraus", dst, many)[0] - f(list(many[0]))).max() < 1e-10, ("50 operators", dst)


def test_five_qubit_conversions(gpu):
    """1024 x 1024 superoperators: the PTM of a product of single-qubit unitaries is the Kronecker product of the
    single-qubit PTMs; reshuffle and Pauli transforms invert each other."""
    from fbx.operator_tools import superoperator_transformations as st
    from fbx.operator_tools.random_operators import haar_rand_unitary_batch
    us = haar_rand_unitary_batch(2, 5, seed=9)
    big = us[0]
    ptm = st.kraus2pauli_liouville(us[0])
    for u in us[1:]:
        big = np.kron(big, u)
        ptm = np.kron(ptm, st.kraus2pauli_liouville(u))
    got = st.convert_batch("kraus", "pauli_liouville", big[None, None])[0]
    assert got.shape == (1024, 1024) and np.abs(got - ptm).max() < 1e-12
    sup = st.convert_batch("pauli_liouville", "superop", got[None])
    assert np.abs(sup[0] - np.kron(big.conj(), big)).max() < 1e-12
    choi = st.convert_batch("superop", "choi", sup)
    v = big.T.reshape(-1, 1)
    assert np.abs(choi[0] - v @ v.conj().T).max() < 1e-12
    assert np.abs(st.convert_batch("choi", "pauli_liouville", choi)[0] - ptm).max() < 1e-12
    # process fidelity on 1024 x 1024 Pauli transfer matrices: to itself 1, to the identity the product formula
    from fbx import distance_measures as dm
    assert abs(dm.process_fidelity(ptm, got) - 1.0) < 1e-12
    fe = np.prod([abs(np.trace(u)) ** 2 / 4 for u in us])                 # entanglement fidelity of a product unitary
    assert abs(dm.process_fidelity(np.eye(1024), got) - (32 * fe + 1) / 33) < 1e-12


def test_basis_free_conversions_in_any_dimension(gpu):
    """Qutrits and a 5-dimensional system: kraus2superop, kraus2choi and the reshuffle (fbx_convert_general) against
    the oracle; conversions that need a Pauli basis say so."""
    from fbx.operator_tools import superoperator_transformations as st
    from fbx_oracle import superops as so
    rs = np.random.RandomState(6)
    for d in (3, 5):
        ks = rs.randn(3, 2, d, d) + 1j * rs.randn(3, 2, d, d)
        sup = st.convert_batch("kraus", "superop", ks)
        choi = st.convert_batch("kraus", "choi", ks)
        for b in range(3):
            assert np.abs(sup[b] - so.kraus2superop(list(ks[b]))).max() < 1e-13
            assert np.abs(choi[b] - so.kraus2choi(list(ks[b]))).max() < 1e-13
            assert np.abs(st.superop2choi(sup[b]) - choi[b]).max() < 1e-13
            assert np.abs(st.choi2superop(choi[b]) - sup[b]).max() < 1e-13
        assert np.abs(st.kraus2superop(list(ks[0])) - sup[0]).max() == 0.0
        with pytest.raises(ValueError):
            st.kraus2pauli_liouville(list(ks[0]))
        with pytest.raises(ValueError):
            st.choi2chi(choi[0])


def test_projection_kernel_variants_agree_bit_for_bit(gpu):
    """Batches of 2048 and more take the two-wavefronts-per-SIMD copy of the projection kernel (proj_choi_w2_kernel):
    same arithmetic, so the same bits as the one-wavefront kernel that small batches use -- for every projection kind,
    Dykstra iteration counts included."""
    from fbx.operator_tools import project_superoperators as ps
    from fbx import _lib
    rs = np.random.RandomState(77)
    x = rs.randn(256, 16, 16) + 1j * rs.randn(256, 16, 16)
    x = 0.1 * x + np.eye(16)[None] / 4
    big = np.ascontiguousarray(np.tile(x, (9, 1, 1)))            # 2304 items
    for kind in (_lib.PROJ_CP, _lib.PROJ_TP, _lib.PROJ_TNI, _lib.PROJ_PHYSICAL_TP, _lib.PROJ_PHYSICAL_TNI):
        small, its = ps.proj_choi_batch(kind, x, return_iters=True)
        large, itl = ps.proj_choi_batch(kind, big, return_iters=True)
        assert np.array_equal(large[:256], small) and np.array_equal(large[-256:], small)
        assert np.array_equal(itl[:256], its)
