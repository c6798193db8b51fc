"""The N > 1 path on the hardware a test box has (ONE GPU): libfbx's RCCL communicator with a 1-rank world
(every collective of include/fbx.h's fbx_comm_* section), and bench.py's own rank spawner with two ranks
sharing the device -- RCCL refuses two ranks on one GPU, so that run exercises the launcher contract, the
rendezvous, the sharded workload and the recorded host fallback; the RCCL collectives between ranks are
covered by construction (same entry points) and by the driver's multi-GPU run."""
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
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_bench_spawns_its_own_ranks(gpu):
    """`python bench.py --gpus 2` with no launcher: two ranks, the configs[4]-shaped strong split (scaled down),
    the weak 1024-per-GPU leg (scaled down), one JSON line from rank 0."""
    line = _bench("--gpus", "2", "--oversubscribe", "--steps", "2", "--warmup", "1", "--total-batch", "600",
                  "--batch", "128", "--iters", "20")
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["steps"] == 2
    assert line["config"]["total_batch"] == 600 and line["config"]["batch_per_gpu"] == 300
    assert line["config"]["collectives"]["ranks"] == 2
    assert line["config"]["collectives"]["backend"] in ("rccl", "host-files")
    assert line["config"]["mean_outer_iters"] == 20.0 and line["value"] > 0
    assert line["per_gpu_1024"]["scaling"] == "weak" and line["per_gpu_1024"]["value"] > 0
    assert line["roofline"]["executed_flop"] > 0 and line["vs_baseline"] is None


def test_bench_single_rank_headline_only(gpu):
    line = _bench("--workload", "pgdb", "--steps", "2", "--warmup", "1", "--batch", "256", "--iters", "30", "--cpu-sample", "0")
    assert line["n_gpus"] == 1 and line["scaling"] == "weak" and line["config"]["batch_per_gpu"] == 256
    r = line["roofline"]
    assert r["bound"] == "mfma" and 0 < r["executed_frac"] < r["frac"] < 1.5 and r["kernel_ms"] <= line["ms_per_step"] * 1.05
    assert "secondary" not in line and "cpu_baseline" not in line
