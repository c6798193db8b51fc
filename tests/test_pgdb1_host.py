"""The per-lane routines of the packed single-qubit PGDB kernel (csrc/fbx_pgdb1_core.hpp), compiled for the host by
tests/host_harness (test infrastructure only -- the package never loads it), against numpy / the oracle / the
reference-generated goldens.  Checks the algebra without a GPU; tests/test_pgdb1_gpu.py checks the kernel itself."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from fbx_oracle import design as od, estimators as oe, superops as so  # noqa: E402

HARNESS = os.path.join(ROOT, "tests", "host_harness")
GOLD = os.path.join(ROOT, "tests", "golden")
dp = ctypes.POINTER(ctypes.c_double)
ip = ctypes.POINTER(ctypes.c_int32)


@pytest.fixture(scope="module")
def lib():
    so_path = os.path.join(HARNESS, "libpgdb1_host.so")
    src = os.path.join(HARNESS, "pgdb1_host.cpp")
    core = os.path.join(ROOT, "forest-benchmarking_amd", "csrc", "fbx_pgdb1_core.hpp")
    if not os.path.exists(so_path) or os.path.getmtime(so_path) < max(os.path.getmtime(src), os.path.getmtime(core)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-Wno-unknown-pragmas", "-ffp-contract=on",
                               src, "-o", so_path])
    return ctypes.CDLL(so_path)


def _p(a, t=dp):
    return a.ctypes.data_as(t)


def _rand_herm(rs, n=4):
    a = rs.randn(n, n) + 1j * rs.randn(n, n)
    return (a + a.conj().T) / 2


def test_jacobi_4x4(lib):
    rs = np.random.RandomState(3)
    for trial in range(50):
        a = _rand_herm(rs)
        if trial % 5 == 0:                      # rank-deficient, like a Choi matrix
            v = rs.randn(4, 2) + 1j * rs.randn(4, 2)
            a = v @ v.conj().T
        a = np.ascontiguousarray(a)
        lam = np.zeros(4)
        v = np.zeros((4, 4), dtype=complex)
        sw = lib.pgdb1_host_eigh(_p(a.view(float)), _p(lam), _p(v.view(float)))
        assert 0 < sw < 12
        assert np.abs(v.conj().T @ v - np.eye(4)).max() < 1e-14
        assert np.abs(v @ np.diag(lam) @ v.conj().T - a).max() < 1e-13 * max(1.0, np.abs(a).max())
        assert np.abs(np.sort(lam) - np.linalg.eigvalsh(a)).max() < 1e-13 * max(1.0, np.abs(a).max())


def test_pauli_butterflies(lib):
    rs = np.random.RandomState(4)
    for _ in range(10):
        e = np.ascontiguousarray(_rand_herm(rs))
        r = np.zeros(16)
        lib.pgdb1_host_choi_to_pauli(_p(e.view(float)), _p(r))
        want = np.real(so.choi2pauli_liouville(e))
        assert np.abs(r.reshape(4, 4) - want).max() < 1e-14
        back = np.zeros((4, 4), dtype=complex)
        lib.pgdb1_host_pauli_to_choi(_p(r), _p(back.view(float)))
        assert np.abs(back - e).max() < 1e-14


@pytest.mark.parametrize("tp", [1, 0])
def test_dykstra(lib, tp):
    rs = np.random.RandomState(5)
    for _ in range(20):
        x = np.ascontiguousarray(np.eye(4) / 2 + 0.4 * _rand_herm(rs))
        out = np.zeros((4, 4), dtype=complex)
        it = lib.pgdb1_host_proj_physical(_p(x.view(float)), tp, _p(out.view(float)))
        want, wit = so.proj_choi_to_physical(x, make_trace_preserving=bool(tp), return_iters=True)
        assert it == wit
        assert np.abs(out - want).max() < 1e-12


def _grouped(design):
    """The grouping fbx_design_create makes (csrc/fbx_runtime.hip): settings by distinct input state, Bloch rows."""
    keys = [tuple(r) for r in design.in_labels]
    uniq = list(dict.fromkeys(keys))
    sidx = np.array([uniq.index(k) for k in keys])
    order = np.argsort(sidx, kind="stable").astype(np.int32)
    sptr = np.zeros(len(uniq) + 1, dtype=np.int32)
    for s in sidx:
        sptr[s + 1] += 1
    sptr = np.cumsum(sptr).astype(np.int32)
    pidx = np.array([int(p[0]) for p in design.paulis])
    sp = ((sidx[order].astype(np.uint32) << 16) | pidx[order].astype(np.uint32)).astype(np.uint32)
    coef = np.asarray(design.coefs, dtype=float)[order].copy()
    ct = np.array([[np.real(np.trace(od.pauli_matrix([j]) @ od.state_matrix(list(k)))) for j in range(4)] for k in uniq])
    return order, sptr, sp, coef, np.ascontiguousarray(ct)


def _run(lib, design, e, c, tp=1, mode=0, max_iters=0):
    order, sptr, sp, coef, ct = _grouped(design)
    B, m = e.shape
    choi = np.zeros((B, 4, 4), dtype=complex)
    it, dy, bt = (np.zeros(B, dtype=np.int32) for _ in range(3))
    cost = np.zeros(B)
    sw = np.zeros(B, dtype=np.int32)
    e = np.ascontiguousarray(e, dtype=float)
    c = np.ascontiguousarray(c, dtype=float)
    lib.pgdb1_host_run(m, len(sptr) - 1, int(np.all(coef == 1.0)), _p(sp, ctypes.POINTER(ctypes.c_uint32)), _p(coef),
                       _p(sptr, ip), _p(ct), _p(order, ip), ctypes.c_long(B), _p(e), _p(c), tp, mode, max_iters,
                       _p(choi.view(float)), _p(it, ip), _p(dy, ip), _p(bt, ip), _p(cost), _p(sw, ip))
    _run.sweeps = sw
    return choi, it, dy, bt, cost


@pytest.mark.parametrize("basis", ["pauli", "sic"])
def test_reconstructions_match_reference_goldens_and_oracle_counts(lib, basis):
    z = np.load(os.path.join(GOLD, f"process_1q_{basis}.npz"))
    design = od.process_design(1, basis)
    assert np.array_equal(design.in_labels, z["in_labels"]) and np.array_equal(design.paulis, z["paulis"])
    e, c = z["expectations"], z["counts"]
    choi, it, dy, bt, cost = _run(lib, design, e, c)
    assert np.abs(choi - z["pgdb"]).max() < 1e-9
    print("sweeps per decomposition", _run.sweeps / dy)
    assert np.all(_run.sweeps < 3.5 * dy)           # warm-started decompositions: well under the 4-6 sweeps of a cold one
    A = oe.design_matrix_A(design)
    for b in range(e.shape[0]):
        want, st = oe.pgdb_process_estimate(design, e[b], c[b], A=A, return_stats=True)
        assert np.abs(choi[b] - want).max() < 1e-9
        assert (it[b], dy[b]) == (st["iterations"], st["dykstra"])
        assert abs(cost[b] - st["cost"]) < 1e-10
    # trace-non-increasing variant (first three items of the golden set)
    n_tni = z["pgdb_tni"].shape[0]
    choi, it, dy, bt, cost = _run(lib, design, e[:n_tni], c[:n_tni], tp=0)
    assert np.abs(choi - z["pgdb_tni"]).max() < 1e-9


def test_fixed_mode_trajectory(lib):
    z = np.load(os.path.join(GOLD, "process_1q_pauli.npz"))
    design = od.process_design(1, "pauli")
    e, c = z["expectations"][:3], z["counts"][:3]
    choi, it, dy, bt, cost = _run(lib, design, e, c, mode=1, max_iters=5)
    A = oe.design_matrix_A(design)
    for b in range(3):
        want, st = oe.pgdb_process_estimate(design, e[b], c[b], A=A, mode="fixed", max_iters=5, return_stats=True)
        assert np.abs(choi[b] - want).max() < 1e-11
        assert (it[b], dy[b], bt[b]) == (5, st["dykstra"], st["backtracks"])
