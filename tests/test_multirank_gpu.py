"""The N > 1 path.  On a box with ONE GPU: libfbx's RCCL communicator with a 1-rank world (every collective of
include/fbx.h's fbx_comm_* section), the bounded wait for a missing peer, and bench.py's own rank spawner with two
ranks sharing the device (--oversubscribe: RCCL refuses two ranks on one GPU, so that run exercises the launcher
contract, the rendezvous, the sharded workload and the recorded host fallback) -- and that WITHOUT --oversubscribe
the same run fails instead of printing a number.  On a box with TWO OR MORE GPUs (skipped otherwise): a real 2-rank
RcclComm -- all-gather, all-reduce, broadcast, run_sharded against the single-GPU answer -- and bench.py --gpus 2
over RCCL."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_communicator_single_rank(gpu):
    from fbx import parallel
    ident = parallel.RcclComm.new_unique_id()
    assert len(ident) == gpu.COMM_ID_BYTES and any(ident)
    comm = parallel.RcclComm(0, 1, ident)
    try:
        assert comm.rccl_version > 20000 and (comm.rank, comm.world) == (0, 1)
        v = comm.allreduce([1.5, -2.0, 7.0], "sum")
        assert np.array_equal(v, [1.5, -2.0, 7.0])
        assert np.array_equal(comm.allreduce([3.0], "max"), [3.0])
        a = (np.arange(24.0).reshape(2, 3, 4) + 1j).astype(np.complex128)
        g = comm.allgather(a)
        assert g.shape == (1, 2, 3, 4) and np.array_equal(g[0], a)
        odd = np.arange(7, dtype=np.uint8)                     # a size that is not a multiple of 8 bytes
        assert np.array_equal(comm.allgather(odd)[0], odd)
        buf = gpu.DeviceBuffer.from_array(np.arange(5.0))
        comm.broadcast_dev(buf.ptr, 40, 0)
        comm.barrier()
        assert np.array_equal(buf.to_array(np.float64, (5,)), np.arange(5.0))
        with pytest.raises(ValueError):                        # a second communicator needs fbx_comm_destroy first
            parallel.RcclComm(0, 1, ident)
        # the sharded helpers over the real communicator
        full, (lo, hi) = parallel.run_sharded(lambda x: x * 2, [np.arange(6.0)], comm)
        assert (lo, hi) == (0, 6) and np.array_equal(full, np.arange(6.0) * 2)
        s, m = parallel.reduce_summary([1.0, 2.0], [5.0], comm)
        assert np.array_equal(s, [1.0, 2.0]) and np.array_equal(m, [5.0])
    finally:
        comm.close()
    rank, world = gpu.C.c_int(-1), gpu.C.c_int(-1)
    gpu.check(gpu.lib().fbx_comm_info(gpu.C.byref(rank), gpu.C.byref(world), None))
    assert world.value == 0
    assert gpu.lib().fbx_comm_barrier() == gpu.FBX_ERR_BAD_ARG          # no communicator any more


def _bench(*args):
    """(compact headline = the LAST stdout line, the full record bench.py writes beside it)"""
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        detail = os.path.join(tmp, "detail.json")
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args, "--detail-out", detail], capture_output=True,
                             text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        lines = out.stdout.strip().splitlines()
        assert len(lines[-1]) < 4096, len(lines[-1])                       # what the driver's tail must hold whole
        assert all(isinstance(json.loads(ln), dict) for ln in lines)
        return json.loads(lines[-1]), json.load(open(detail))


def test_bench_spawns_its_own_ranks(gpu):
    """`python bench.py --gpus 2` with no launcher: two ranks, the configs[4]-shaped strong split (scaled down),
    the weak 1024-per-GPU leg (scaled down), one JSON line from rank 0."""
    line, full = _bench("--gpus", "2", "--oversubscribe", "--steps", "2", "--warmup", "1", "--total-batch", "600",
                        "--batch", "128", "--iters", "20")
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["steps"] == 2
    assert line["config"]["total_batch"] == 600 and line["config"]["batch_per_gpu"] == 300
    assert line["config"]["collectives"]["ranks"] == 2
    assert line["config"]["collectives"]["backend"] in ("rccl", "host-files")
    assert line["config"]["mean_outer_iters"] == 20.0 and line["value"] > 0
    assert line["config"]["per_gpu_1024"]["value"] > 0 and full["per_gpu_1024"]["scaling"] == "weak"
    assert 0 < line["roofline"]["frac"] < 1 and full["roofline"]["executed_flop"] > 0 and line["vs_baseline"] is None
    assert full["value"] == pytest.approx(line["value"], rel=1e-5)


def test_bench_single_rank_headline_only(gpu):
    line, full = _bench("--workload", "pgdb", "--steps", "2", "--warmup", "1", "--batch", "256", "--iters", "30", "--cpu-sample", "0")
    assert line["n_gpus"] == 1 and line["scaling"] == "weak" and line["config"]["batch_per_gpu"] == 256
    r = full["roofline"]
    # frac is the EXECUTED fraction of the fp64 peak (never above 1); the dense-A accounting figure sits beside it
    assert r["bound"] == "mfma" and 0 < r["frac"] < 1.0 and r["frac"] < r["dense_accounting_frac"] and r["kernel_ms"] <= full["ms_per_step"] * 1.05
    assert abs(r["achieved"] / r["peak"] - r["frac"]) < 1e-12 and r["executed_flop"] > 0
    assert 0 <= r["mfma_frac"] < r["frac"]                                # matrix-core share of the peak: a utilisation, <= 1
    assert "secondary" not in full and "cpu_baseline" not in full
    c = line["roofline"]
    assert set(c) >= {"bound", "achieved", "peak", "unit", "frac", "mfma_frac", "traffic", "kernel", "kernel_ms"}
    assert c["frac"] == pytest.approx(r["frac"], rel=1e-4) and c["kernel"].startswith("pgdb_kernel<2,")


def test_bench_state_mle_workload(gpu):
    """The state-estimator half of north_star has a bench line: rate, executed-flop roofline, the oracle beside it."""
    line, full = _bench("--workload", "mle_state", "--steps", "2", "--warmup", "1", "--iters", "100", "--cpu-sample", "6")
    assert line["metric"].startswith("state-tomography iterative-MLE") and line["unit"] == "reconstructions/s" and line["value"] > 1e5
    assert line["config"]["mean_outer_iters"] == 100.0                  # 1000-shot data never reaches tol 1e-9 in 99 updates
    assert full["config"]["max_abs_diff_vs_oracle"] < 1e-10 and full["config"]["max_trace_error"] < 1e-12
    assert line["roofline"]["kernel"] == "mle_state_packed_kernel<2>" and 0 < line["roofline"]["frac"] < 1
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["value"] > 0


def test_bench_shots_workload(gpu):
    """The caller-side reduction as a bench line: HBM roofline on algorithmic bytes, exact against the oracle."""
    line, full = _bench("--workload", "shots", "--steps", "3", "--warmup", "1", "--cpu-sample", "4")
    assert line["unit"] == "settings/s" and line["dtype"] == "u8" and line["roofline"]["bound"] == "hbm"
    assert 0.05 < line["roofline"]["frac"] < 1.0 and line["roofline"]["unit"] == "GB/s"
    assert full["config"]["matches_oracle_on_sample"] is True and line["cpu_baseline"]["value"] > 0


def test_bench_refuses_to_share_a_gpu_without_oversubscribe(gpu):
    """One GPU per rank or no number: two ranks on a one-GPU box must exit non-zero (no host-files fallback)."""
    import fbx
    if fbx.device_count() >= 2:
        pytest.skip("needs a box with fewer GPUs than ranks")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--total-batch", "64", "--batch", "64", "--iters", "5", "--spawn-timeout", "120"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert not any(l.startswith("{") for l in out.stdout.splitlines())


_TIMEOUT_WORKER = r"""
import os, sys, time
sys.path.insert(0, sys.argv[1])
from fbx import _lib, parallel
_lib.set_device(0)
ident = parallel.RcclComm.new_unique_id()
t0 = time.monotonic()
try:
    parallel.RcclComm(0, 2, ident, timeout=4.0)          # rank 1 never shows up
    print("UNEXPECTED: initialised"); sys.stdout.flush(); os._exit(1)
except _lib.FbxError as exc:
    waited = time.monotonic() - t0
    assert exc.code == _lib.FBX_ERR_RCCL and "did not return" in str(exc), str(exc)
# afterwards nothing blocks: info and destroy return at once, another init is refused with FBX_ERR_RCCL
t1 = time.monotonic()
w = _lib.C.c_int(-1)
_lib.check(_lib.lib().fbx_comm_info(None, _lib.C.byref(w), None))
assert w.value == 0
assert _lib.lib().fbx_comm_destroy() == 0
assert _lib.lib().fbx_comm_barrier() == _lib.FBX_ERR_RCCL
ident2 = (_lib.C.c_uint8 * 128).from_buffer_copy(ident)
assert _lib.lib().fbx_comm_init_timeout(ident2, 0, 1, 5.0) == _lib.FBX_ERR_RCCL
assert time.monotonic() - t1 < 2.0
# the rest of the library still works in this process
from fbx import synthetic, tomography
d, _, e, c = synthetic.process_batch(1, "sic", 2)
assert tomography.pgdb_process_estimate_batch(d, e, c).shape == (2, 4, 4)
print("OK waited %.1f" % waited); sys.stdout.flush()
os._exit(0)                                              # the helper thread still sits inside ncclCommInitRank
"""


def test_comm_init_times_out_on_a_missing_peer(gpu):
    out = subprocess.run([sys.executable, "-c", _TIMEOUT_WORKER, os.path.join(ROOT, "forest-benchmarking_amd")],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "OK waited" in out.stdout, (out.stdout[-1000:], out.stderr[-2000:])
    assert 3.5 < float(out.stdout.split("OK waited")[1].split()[0]) < 30.0


_TWO_RANK_WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from fbx import _lib, parallel, synthetic, tomography
comm, rdzv = parallel.init_from_env()                    # no fallback: RCCL or an exception
assert comm.backend == "rccl" and comm.world == 2 and comm.rank == int(os.environ["RANK"])
q = comm.query()
assert q["world"] == 2 and q["rank"] == comm.rank and q["device"] == int(os.environ["LOCAL_RANK"])
ordinal, pci = _lib.device_id()
rows = comm.allgather(np.frombuffer(pci.encode().ljust(32, b"\0"), dtype=np.uint8))
assert rows.shape == (2, 32) and bytes(rows[0]) != bytes(rows[1])               # two physical devices
# all-reduce: sum / max / min, and a vector longer than one staging chunk
v = comm.allreduce([1.0 + comm.rank, 10.0 * comm.rank], "sum"); assert np.array_equal(v, [3.0, 10.0])
assert comm.allreduce([float(comm.rank)], "max")[0] == 1.0 and comm.allreduce([float(comm.rank)], "min")[0] == 0.0
big = comm.allreduce(np.arange(10000.0) * (comm.rank + 1), "sum"); assert np.array_equal(big, np.arange(10000.0) * 3)
# all-gather of complex slabs, an odd byte count
a = (np.arange(12.0).reshape(3, 4) * (comm.rank + 1) + 1j * comm.rank).astype(np.complex128)
g = comm.allgather(a); assert g.shape == (2, 3, 4) and np.array_equal(g[comm.rank], a) and np.array_equal(g[1 - comm.rank].imag, np.full((3, 4), 1.0 - comm.rank))
odd = np.arange(7, dtype=np.uint8) + comm.rank; g = comm.allgather(odd); assert np.array_equal(g[0], np.arange(7)) and np.array_equal(g[1], np.arange(7) + 1)
# broadcast of design-sized constants from rank 1
buf = _lib.DeviceBuffer.from_array(np.arange(64.0) if comm.rank == 1 else np.zeros(64))
comm.broadcast_dev(buf.ptr, 512, 1); comm.barrier()
assert np.array_equal(buf.to_array(np.float64, (64,)), np.arange(64.0))
# the sharded estimator: 2-qubit PGDB on 37 items (ragged split 19 + 18), gathered on every rank
design, _, e, c = synthetic.process_batch(2, "sic", 37)
full, (lo, hi) = parallel.run_sharded(lambda eb, cb: tomography.pgdb_process_estimate_batch(design, eb, cb), [e, c], comm)
assert (lo, hi) == parallel.shard_bounds(37, comm.rank, 2) and full.shape == (37, 16, 16)
np.save(os.path.join(sys.argv[2], "full_%d.npy" % comm.rank), full)
s, m = parallel.reduce_summary([hi - lo, float(np.trace(full[lo:hi], axis1=1, axis2=2).real.sum())], [hi - lo], comm)
assert s[0] == 37 and m[0] == 19 and abs(s[1] - np.trace(full, axis1=1, axis2=2).real.sum()) < 1e-9
comm.barrier(); comm.close(); rdzv.close()
print("RANK_OK", comm.rank)
"""


def _needs_two_gpus():
    import fbx
    if fbx.device_count() < 2:
        pytest.skip("needs at least two GPUs (one process per GPU over RCCL)")


def test_two_rank_rccl_collectives_and_sharded_estimator(gpu, tmp_path):
    """SURVEY 8e on hardware: two processes, two GPUs, libfbx's own communicator; the gathered sharded result equals
    the single-GPU answer bit for bit (items are independent: the partition cannot change an item's arithmetic)."""
    _needs_two_gpus()
    from fbx import synthetic, tomography
    rd = tmp_path / "rdzv"
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29517",
                   FBX_RDZV_DIR=str(rd), HSA_ENABLE_IPC_MODE_LEGACY="0", FBX_RCCL_INIT_TIMEOUT="120")
        procs.append(subprocess.Popen([sys.executable, "-c", _TWO_RANK_WORKER, os.path.join(ROOT, "forest-benchmarking_amd"),
                                       str(tmp_path)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"RANK_OK {r}" in so, (so[-1000:], se[-3000:])
    design, _, e, c = synthetic.process_batch(2, "sic", 37)
    want = tomography.pgdb_process_estimate_batch(design, e, c)
    a, b = np.load(tmp_path / "full_0.npy"), np.load(tmp_path / "full_1.npy")
    assert np.array_equal(a, b) and np.array_equal(a, want)


def test_bench_two_gpus_over_rccl(gpu):
    _needs_two_gpus()
    line, full = _bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--total-batch", "4096", "--batch", "256", "--iters", "30")
    col = full["config"]["collectives"]
    assert col["backend"] == "rccl" and col["ranks"] == 2 and "rccl_failure" not in col
    assert len({d["pci_bus_id"] for d in col["rank_devices"]}) == 2 and sorted(d["rank"] for d in col["rank_devices"]) == [0, 1]
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["config"]["batch_per_gpu"] == 2048
