"""Batched Kraus-valued outputs through the C ABI: fbx_choi2kraus[_dev] (superoperator_transformations.py:325-336) and the
routes the reference sends through it (superop2kraus :229-238, pauli_liouville2kraus :280-288, chi2kraus :195-204).

Kraus operators are eigenvector-valued -- defined up to the phase of each eigenvector and, inside a degenerate eigenspace, up
to a unitary mixing -- so the assertions are the ones the reference's own tests make
(tests/test_superoperator_transformations.py:215-224, 263-271): kraus2choi(choi2kraus(C)) = C, |K| against the known
operators, plus the reference's operator COUNT and ORDER (ascending eigenvalue) against the oracle, and entry-by-entry equality
with the host assembly the one-item function used before (same phase convention) where the spectrum is simple."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _random_choi(n, B, K, seed):
    from fbx.operator_tools import random_operators as ro, convert_batch
    if n <= 3:
        ks = np.ascontiguousarray(ro.random_kraus_batch(2 ** n, K, B, seed=seed))      # [B, K, d, d] CPTP sets, generated on the device
    else:
        from fbx import synthetic
        ks = np.ascontiguousarray(synthetic.kraus_batch(n, K, B, seed=seed))
    return ks, convert_batch("kraus", "choi", ks)


@pytest.mark.parametrize("n,B,K", [(1, 257, 2), (2, 300, 4), (3, 33, 5), (4, 3, 3)])
def test_choi2kraus_batch_round_trip_counts_and_order(gpu, n, B, K):
    from fbx.operator_tools import choi2kraus_batch, convert_batch
    from fbx_oracle import superops as so
    _, choi = _random_choi(n, B, K, 11 + n)
    kraus, counts = choi2kraus_batch(choi)
    d, D = 2 ** n, 4 ** n
    assert kraus.shape == (B, D, d, d) and counts.shape == (B,)
    assert (counts == K).all()                                         # rank K: the other D - K eigenvalues are below 1e-9
    for b in range(B):
        assert not kraus[b, counts[b]:].any()                          # unused slots are zero
    # kraus2choi(choi2kraus(C)) = C -- on the device, the kept slots
    back = convert_batch("kraus", "choi", np.ascontiguousarray(kraus[:, :max(1, int(counts.max()))]))
    assert np.abs(back - choi).max() < 1e-12 * D
    # the reference's list: ascending eigenvalues, operator norms = the eigenvalues
    for b in range(0, B, max(1, B // 7)):
        ref = so.choi2kraus(choi[b])
        assert len(ref) == counts[b]
        w = np.linalg.eigvalsh(choi[b])
        w = w[np.abs(w) > 1e-9]
        norms = np.array([np.vdot(kraus[b, i], kraus[b, i]).real for i in range(counts[b])])
        assert np.allclose(norms, w, rtol=1e-10, atol=1e-12)
        assert (np.diff(norms) > -1e-12).all()
        # simple spectrum (random channel): each operator equals the reference's up to a phase
        for i, r in enumerate(ref):
            ov = np.vdot(r, kraus[b, i])
            assert abs(abs(ov) - norms[i]) < 1e-9 * max(1.0, norms[i])


def test_phase_convention_and_known_answers(gpu):
    """tests/test_superoperator_transformations.py:215-224, 263-271 on the batched entry point."""
    from fbx.operator_tools import choi2kraus_batch, superop2kraus_batch, pauli_liouville2kraus_batch, chi2kraus_batch, convert_batch
    # identity channel: one operator, |K| = I; Z-rotation-like IZ: entry by entry
    id_choi = np.array([[1, 0, 0, 1], [0, 0, 0, 0], [0, 0, 0, 0], [1, 0, 0, 1]], dtype=complex)
    kraus, counts = choi2kraus_batch(id_choi[None])
    assert counts[0] == 1 and np.allclose(np.abs(kraus[0, 0]), np.eye(2)) and not kraus[0, 1:].any()
    iz_super = np.diag([1, -1, -1, 1]).astype(complex)                 # Z . Z: superoperator of the single Kraus operator Z
    k, c = superop2kraus_batch(np.stack([iz_super, np.eye(4, dtype=complex)]))
    assert list(c) == [1, 1]
    assert np.allclose(k[0, 0], np.diag([1, -1])) and np.allclose(k[1, 0], np.eye(2))       # first non-zero component real positive
    # amplitude damping: |K| in the reference's order (the smaller eigenvalue first), any of the three non-Choi sources
    p = 0.3
    ad = np.array([[[1, 0], [0, np.sqrt(1 - p)]], [[0, np.sqrt(p)], [0, 0]]], dtype=complex)
    sup = convert_batch("kraus", "superop", ad[None])
    pl = convert_batch("kraus", "pauli_liouville", ad[None])
    chi = convert_batch("kraus", "chi", ad[None])
    for fn, x in ((superop2kraus_batch, sup), (pauli_liouville2kraus_batch, pl), (chi2kraus_batch, chi)):
        k, c = fn(x)
        assert c[0] == 2
        assert np.allclose([np.abs(k[0, 1]), np.abs(k[0, 0])], ad, atol=1e-12)


def test_negative_eigenvalues_and_tolerance(gpu):
    """A Hermitian, non-CP 'Choi' matrix: numpy's scimath square root makes the operator of a negative eigenvalue
    i sqrt(|lambda|) unvec(v); eigenvalues within tol are dropped and the count says so."""
    from fbx.operator_tools import choi2kraus_batch
    from fbx_oracle import superops as so
    rs = np.random.RandomState(5)
    g = rs.randn(6, 16, 16) + 1j * rs.randn(6, 16, 16)
    h = g + g.conj().transpose(0, 2, 1)
    kraus, counts = choi2kraus_batch(h)
    assert (counts == 16).all()
    for b in range(6):
        ref = so.choi2kraus(h[b])
        w = np.linalg.eigvalsh(h[b])
        for i, r in enumerate(ref):
            # sum_i K_i (x) conj(K_i)-type reconstructions do not apply to a non-CP input; compare operator by operator up to phase
            ov = np.vdot(r, kraus[b, i])
            assert abs(abs(ov) - abs(w[i])) < 1e-9 * abs(w).max()
            # the phase-fixed eigenvector is real positive in its first component: the operator's (0, 0) entry carries sqrt(lambda)'s phase
            lead = kraus[b, i].T.reshape(-1)[np.flatnonzero(np.abs(kraus[b, i].T.reshape(-1)) > 1e-12 * np.sqrt(abs(w[i])))[0]]
            assert abs(lead.imag if w[i] > 0 else lead.real) < 1e-12 * np.sqrt(abs(w).max()) and (lead.real > 0 if w[i] > 0 else lead.imag > 0)
    # a tolerance above some eigenvalues
    tol = float(np.sort(np.abs(np.linalg.eigvalsh(h[0])))[5]) * 1.0000001
    k2, c2 = choi2kraus_batch(h[:1], tol=tol)
    assert c2[0] == 10 and not k2[0, 10:].any()


def test_device_pointer_form_and_one_item_function(gpu):
    from fbx import _lib
    from fbx.operator_tools import choi2kraus, choi2kraus_batch
    _, choi = _random_choi(2, 64, 3, 3)
    kraus, counts = choi2kraus_batch(choi)
    d_c = _lib.DeviceBuffer.from_array(choi)
    d_k = _lib.DeviceBuffer(kraus.nbytes)
    d_n = _lib.DeviceBuffer(4 * 64)
    _lib.check(_lib.lib().fbx_choi2kraus_dev(2, 64, d_c.ptr, ctypes.c_double(1e-9), d_k.ptr, d_n.ptr))
    assert np.array_equal(d_k.to_array(np.complex128, kraus.shape), kraus)
    assert np.array_equal(d_n.to_array(np.int32, (64,)), counts)
    ops = choi2kraus(choi[7])
    assert len(ops) == counts[7] and all(np.array_equal(o, kraus[7, i]) for i, o in enumerate(ops))
    # B = 0 and bad arguments
    assert _lib.lib().fbx_choi2kraus_dev(2, 0, None, ctypes.c_double(1e-9), None, None) == 0
    assert _lib.lib().fbx_choi2kraus_dev(6, 1, d_c.ptr, ctypes.c_double(1e-9), d_k.ptr, None) != 0
    for b in (d_c, d_k, d_n):
        b.free()
