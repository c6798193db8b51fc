"""Known-answer values of the reference's own test-suite and docs, transcribed as data.

Sources (values only; the surrounding test logic is this repository's):
  tests/test_superoperator_transformations.py:12-114  amplitude damping / Hadamard / I(x)Z in all
      five representations (hand-derived in docs/superoperator_representations.md)
  tests/test_project_superoperators.py:7-113          CP / TP / TNI / CPTP projections
  tests/test_distance_measures.py:269-276, docs/examples/distance_measures.ipynb cell 29
  docs/examples/superoperator_tools.ipynb cell 63     proj_choi_to_physical(-kraus2choi(I))
"""
import numpy as np

X = np.array([[0, 1], [1, 0]], dtype=complex)
Y = np.array([[0, -1j], [1j, 0]])
Z = np.array([[1, 0], [0, -1]], dtype=complex)
I2 = np.eye(2, dtype=complex)
H = np.array([[1, 1], [1, -1]], dtype=complex) / np.sqrt(2)
CNOT = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 0, 1], [0, 0, 1, 0]], dtype=complex)


def amplitude_damping_kraus(p):
    return [np.array([[1, 0], [0, np.sqrt(1 - p)]], dtype=complex), np.array([[0, np.sqrt(p)], [0, 0]], dtype=complex)]


def amplitude_damping_chi(p):
    a, b = (1 + np.sqrt(1 - p)) ** 2, (-1 + np.sqrt(1 - p)) ** 2
    return 0.25 * np.array([[a, 0, 0, p], [0, p, -1j * p, 0], [0, 1j * p, p, 0], [p, 0, 0, b]])


def amplitude_damping_pauli(p):
    s = np.sqrt(1 - p)
    return np.array([[1, 0, 0, 0], [0, s, 0, 0], [0, 0, s, 0], [p, 0, 0, 1 - p]], dtype=complex)


def amplitude_damping_super(p):
    s = np.sqrt(1 - p)
    return np.array([[1, 0, 0, p], [0, s, 0, 0], [0, 0, s, 0], [0, 0, 0, 1 - p]], dtype=complex)


def amplitude_damping_choi(p):
    s = np.sqrt(1 - p)
    return np.array([[1, 0, 0, s], [0, 0, 0, 0], [0, 0, p, 0], [s, 0, 0, 1 - p]], dtype=complex)


HAD_CHI = 0.5 * np.array([[0, 0, 0, 0], [0, 1, 0, 1], [0, 0, 0, 0], [0, 1, 0, 1]], dtype=complex)
HAD_PAULI = np.array([[1, 0, 0, 0], [0, 0, 0, 1], [0, 0, -1, 0], [0, 1, 0, 0]], dtype=complex)
HAD_SUPER = 0.5 * np.array([[1, 1, 1, 1], [1, -1, 1, -1], [1, 1, -1, -1], [1, -1, -1, 1]], dtype=complex)
HAD_CHOI = 0.5 * np.array([[1, 1, 1, -1], [1, 1, 1, -1], [1, 1, 1, -1], [-1, -1, -1, 1]], dtype=complex)
IZ_KRAUS = np.kron(I2, Z)
IZ_SUPER = np.diag([1, -1, 1, -1, -1, 1, -1, 1, 1, -1, 1, -1, -1, 1, -1, 1]).astype(complex)

# process_fidelity(PTM(I), PTM(exp(-0.2 i X))) recorded in docs/examples/distance_measures.ipynb
PROCESS_FIDELITY_RX02 = 0.9736869980009618
# Dykstra tolerance is visible in the answer: superoperator_tools.ipynb cell 63
PROJ_PHYSICAL_MINUS_IDENTITY_DIAG = (0.33398438, 0.66601562, -0.33203125)


def check_conversions(ns, atol=1e-12):
    """ns: namespace with the reference's operator_tools function names."""
    for p in (0.1, 0.37, 0.9):
        ks = amplitude_damping_kraus(p)
        assert np.allclose(ns.kraus2chi(ks), amplitude_damping_chi(p), atol=atol)
        assert np.allclose(ns.kraus2pauli_liouville(ks), amplitude_damping_pauli(p), atol=atol)
        assert np.allclose(ns.kraus2superop(ks), amplitude_damping_super(p), atol=atol)
        assert np.allclose(ns.kraus2choi(ks), amplitude_damping_choi(p), atol=atol)
        assert np.allclose(ns.chi2pauli_liouville(amplitude_damping_chi(p)), amplitude_damping_pauli(p), atol=atol)
        assert np.allclose(ns.superop2choi(amplitude_damping_super(p)), amplitude_damping_choi(p), atol=atol)
        assert np.allclose(ns.choi2superop(amplitude_damping_choi(p)), amplitude_damping_super(p), atol=atol)
        assert np.allclose(ns.superop2pauli_liouville(amplitude_damping_super(p)), amplitude_damping_pauli(p), atol=atol)
        assert np.allclose(ns.pauli_liouville2superop(amplitude_damping_pauli(p)), amplitude_damping_super(p), atol=atol)
        assert np.allclose(ns.choi2chi(amplitude_damping_choi(p)), amplitude_damping_chi(p), atol=1e-10)
    assert np.allclose(ns.kraus2chi(H), HAD_CHI, atol=atol)
    assert np.allclose(ns.kraus2pauli_liouville(H), HAD_PAULI, atol=atol)
    assert np.allclose(ns.kraus2superop(H), HAD_SUPER, atol=atol)
    assert np.allclose(ns.kraus2choi(H), HAD_CHOI, atol=atol)
    assert np.allclose(ns.chi2pauli_liouville(HAD_CHI), HAD_PAULI, atol=atol)
    assert np.allclose(ns.kraus2superop(IZ_KRAUS), IZ_SUPER, atol=atol)
    assert np.allclose(ns.superop2chi(IZ_SUPER), ns.kraus2chi(IZ_KRAUS), atol=1e-10)
    assert np.allclose(ns.pauli_liouville2choi(ns.kraus2pauli_liouville(H)), ns.kraus2choi(H), atol=atol)
    for a in (I2, X, Y, Z):
        for b in (I2, X, Y, Z):
            k = np.kron(a, b)
            assert np.allclose(ns.pauli_liouville2choi(ns.kraus2pauli_liouville(k)), ns.kraus2choi(k), atol=atol)
            assert np.allclose(ns.superop2choi(ns.kraus2superop(k)), ns.kraus2choi(k), atol=atol)
    hc = ns.kraus2choi(H)
    assert np.allclose(ns.choi2superop(ns.choi2superop(hc)), hc, atol=atol)
    xz = np.zeros((16, 1)); xz[7] = 1.0
    p2c = ns.pauli2computational_basis_matrix(4)
    assert np.allclose((p2c @ xz).reshape(4, 4).T, np.kron(X, Z), atol=atol)
    assert np.allclose(ns.computational2pauli_basis_matrix(4) @ np.kron(X, Z).T.reshape(-1, 1), xz, atol=atol)


def check_projections(ns, atol=1e-10):
    eye = np.eye(4, dtype=complex)
    assert np.allclose(ns.proj_choi_to_completely_positive(eye), eye, atol=atol)
    big = np.diag([1.5, 10, 3, 0.5]).astype(complex)
    assert np.allclose(ns.proj_choi_to_completely_positive(big), big, atol=atol)
    mz = -np.kron(Z, I2)                       # eigenvalues -1,-1,+1,+1 : negative part removed
    assert np.allclose(ns.proj_choi_to_completely_positive(mz), np.diag([0, 0, 1, 1]), atol=atol)
    xx = np.kron(I2, X)
    assert np.allclose(ns.proj_choi_to_completely_positive(xx), np.kron(I2, np.array([[.5, .5], [.5, .5]])), atol=atol)
    yy = np.kron(I2, Y)
    assert np.allclose(ns.proj_choi_to_completely_positive(yy), np.kron(I2, np.array([[.5, -.5j], [.5j, .5]])), atol=atol)
    for k in (I2, X):
        choi = ns.kraus2choi(k)
        assert np.allclose(ns.proj_choi_to_trace_preserving(choi), choi, atol=atol)
        assert np.allclose(ns.proj_choi_to_trace_non_increasing(choi), choi, atol=atol)
        assert np.allclose(ns.proj_choi_to_physical(choi), choi, atol=atol)
    choi = ns.kraus2choi(X - I2 * .01)
    tp = ns.proj_choi_to_trace_preserving(choi)
    pt = np.einsum("iojo->ij", tp.reshape(2, 2, 2, 2))
    assert np.allclose(pt, I2, atol=atol)
    choi = np.array([[0., 0., 0., 0.], [0., 1.01, 1.01, 0.], [0., 1., 1., 0.], [0., 0., 0., 0.]], dtype=complex)
    tni = ns.proj_choi_to_trace_non_increasing(choi)
    assert np.allclose(np.einsum("iojo->ij", tni.reshape(2, 2, 2, 2)), I2, atol=1e-8)
    choi = np.array([[1.001, 0., 0., .99], [0., 0., 0., 0.], [0., 0., 0., 0.], [1.004, 0., 0., 1.01]], dtype=complex)
    assert np.allclose(ns.proj_choi_to_physical(choi), choi, atol=1e-2)
    choi = np.array([[1.1, 0.2, -0.4, .9], [.5, 0., 0., 0.], [0., 0., 0., 0.], [1.4, 0., 0., .8]], dtype=complex)
    phys = ns.proj_choi_to_physical(choi)
    assert np.allclose(np.einsum("iojo->ij", phys.reshape(2, 2, 2, 2)), I2, atol=1e-8)
    assert np.linalg.eigvalsh((phys + phys.conj().T) / 2).min() > -1e-1
    # the recorded notebook output: Dykstra's 1e-4 tolerance shows in the digits
    out = ns.proj_choi_to_physical(-ns.kraus2choi(I2))
    a, b, c = PROJ_PHYSICAL_MINUS_IDENTITY_DIAG
    assert abs(out[0, 0].real - a) < 1e-7 and abs(out[1, 1].real - b) < 1e-7 and abs(out[0, 3].real - c) < 1e-7


def check_process_fidelity(ns, dm):
    u = np.array([[np.cos(0.2), -1j * np.sin(0.2)], [-1j * np.sin(0.2), np.cos(0.2)]])
    f = dm.process_fidelity(ns.kraus2pauli_liouville(I2), ns.kraus2pauli_liouville(u))
    assert abs(f - PROCESS_FIDELITY_RX02) < 1e-14
    # identical unitaries have process fidelity 1, orthogonal Paulis 1/(d+1)
    assert abs(dm.process_fidelity(ns.kraus2pauli_liouville(H), ns.kraus2pauli_liouville(H)) - 1.0) < 1e-14
    assert abs(dm.process_fidelity(ns.kraus2pauli_liouville(X), ns.kraus2pauli_liouville(Z)) - 1.0 / 3) < 1e-14
    assert abs(dm.process_fidelity(ns.kraus2pauli_liouville(CNOT), ns.kraus2pauli_liouville(CNOT)) - 1.0) < 1e-14
