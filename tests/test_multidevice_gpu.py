"""One call, several GPUs (fbx_set_devices): the host-pointer batch entry points split a batch into contiguous
blocks, one per entry of the device list, on worker threads inside the library -- the unit being split is the reference's
independent experiment (one entry of get_results_by_qubit_groups, observable_estimation.py:1145-1173).  Required: BIT-IDENTICAL
to the single-device call, stats and traces included.  On a one-GPU box the list [0, 0] exercises the threading, the per-worker
contexts and the block arithmetic (two workers share the device); with two or more GPUs the same checks run on 'all'."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture
def device_list(gpu):
    from fbx import _lib
    yield _lib
    _lib.set_devices([0])                       # back to one device for the rest of the session


def _lists(_lib):
    out = [[0, 0], [0, 0, 0]]
    if _lib.device_count() >= 2:
        out.append("all")
    return out


def test_pgdb_split_over_the_device_list_is_bit_identical(device_list):
    from fbx import synthetic, tomography
    _lib = device_list
    cases = [(2, "pauli", 301, dict(mode="converge")), (2, "sic", 64, dict(mode="fixed", max_iters=12)),
             (1, "pauli", 50, dict(mode="converge", trace_preserving=False)), (3, "sic", 5, dict(mode="fixed", max_iters=3))]
    for n, basis, B, kw in cases:
        design, _, e, c = synthetic.process_batch(n, basis, B)
        _lib.set_devices([0])
        want, wst = tomography.pgdb_process_estimate_batch(design, e, c, return_stats=True, trace_iters=40, **kw)
        for ids in _lists(_lib):
            got, st = tomography.pgdb_process_estimate_batch(design, e, c, return_stats=True, trace_iters=40, devices=ids, **kw)
            assert np.array_equal(got, want), (n, basis, ids)
            for k in wst:
                assert np.array_equal(st[k], wst[k]), (n, basis, ids, k)


def test_small_batches_and_page_locked_buffers(device_list):
    """Fewer than two items per list entry: not split.  Page-locked buffers: every worker runs its block through the
    pipelined host-pointer path on its own streams."""
    from fbx import synthetic, tomography
    _lib = device_list
    design, _, e, c = synthetic.process_batch(2, "sic", 3)
    _lib.set_devices([0])
    want = tomography.pgdb_process_estimate_batch(design, e, c)
    assert np.array_equal(tomography.pgdb_process_estimate_batch(design, e, c, devices=[0, 0]), want)
    B = 6000
    design, _, e, c = synthetic.process_batch(2, "sic", 64)
    e, c = np.tile(e, (B // 64 + 1, 1))[:B], np.tile(c, (B // 64 + 1, 1))[:B]
    pe, pc, out = _lib.pinned_copy(e), _lib.pinned_copy(c), _lib.pinned_empty((B, 16, 16), np.complex128)
    _lib.set_devices([0])
    want = tomography.pgdb_process_estimate_batch(design, pe, pc, mode="fixed", max_iters=8).copy()
    got = tomography.pgdb_process_estimate_batch(design, pe, pc, mode="fixed", max_iters=8, out=out, devices=[0, 0])
    assert got is out and np.array_equal(got, want)
    assert np.array_equal(got[:64], got[64:128])


def test_kraus_sweep_split_over_the_device_list(device_list):
    from fbx import synthetic
    _lib = device_list
    n, K, B, D = 2, 4, 1001, 16
    ks = np.ascontiguousarray(synthetic.kraus_batch(n, K, B, seed=3))
    ref = np.eye(D, dtype=np.complex128)

    def run():
        outs = [np.empty((B, D, D), dtype=np.complex128) for _ in range(3)]
        fid = np.empty(B)
        _lib.check(_lib.lib().fbx_kraus_sweep(n, B, K, _lib.dptr(ks.view(np.float64)), _lib.dptr(ref.view(np.float64)),
                                              *[_lib.dptr(o.view(np.float64)) for o in outs], _lib.dptr(fid)))
        return outs + [fid]

    _lib.set_devices([0])
    want = run()
    for ids in _lists(_lib):
        _lib.set_devices(ids)
        got = run()
        assert all(np.array_equal(g, w) for g, w in zip(got, want)), ids


def test_bad_device_lists_are_refused(device_list):
    _lib = device_list
    with pytest.raises(ValueError):
        _lib.set_devices([0, _lib.device_count()])
    with pytest.raises(ValueError):
        _lib.set_devices([])
    assert _lib.set_devices([0]) == (0,)


def test_alternating_device_lists_reuse_their_workers(device_list):
    """Round-5 advisor finding: a list change used to drop its surplus worker threads -- parked for ever with their streams,
    staging pools and multi-GB workspaces.  Now workers that leave the list release their device memory and are kept in a pool
    keyed by device; later lists reuse them.  Forty changes of the list must not grow the process's thread count, and results
    stay bit-identical throughout."""
    import os
    from fbx import synthetic, tomography
    _lib = device_list

    def threads():
        with open(f"/proc/{os.getpid()}/status") as f:
            return int([ln for ln in f if ln.startswith("Threads:")][0].split()[1])

    design, _, e, c = synthetic.process_batch(2, "sic", 40)
    _lib.set_devices([0])
    want = tomography.pgdb_process_estimate_batch(design, e, c, mode="fixed", max_iters=5)
    for ids in ([0, 0, 0], [0], [0, 0]):                     # warm: the pool now holds three workers of device 0
        assert np.array_equal(tomography.pgdb_process_estimate_batch(design, e, c, mode="fixed", max_iters=5, devices=ids), want)
    before = threads()
    for k in range(40):
        ids = ([0], [0, 0], [0, 0, 0], [0, 0])[k % 4]
        got = tomography.pgdb_process_estimate_batch(design, e, c, mode="fixed", max_iters=5, devices=ids)
        assert np.array_equal(got, want), (k, ids)
    assert threads() <= before, (before, threads())
    _lib.release_workspace()                                  # fans out to the workers of the current list
    assert np.array_equal(tomography.pgdb_process_estimate_batch(design, e, c, mode="fixed", max_iters=5, devices=[0, 0]), want)
