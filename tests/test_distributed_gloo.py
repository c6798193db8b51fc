"""world_size-2 gloo run of the sharded batch path on CPU: contiguous partition, no data-path
collective, one all-gather of result slabs.  The per-shard compute is the CPU oracle (the HIP
library needs a GPU); what is under test is the N > 1 plumbing bench.py and fbx.parallel use."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, out_dir):
    for p in (os.path.join(ROOT, "forest-benchmarking_amd"), os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fbx import synthetic
    from fbx.parallel import reduce_summary, run_sharded, shard_bounds
    from fbx_oracle import design as od, estimators as oe
    design, _, e, c = synthetic.process_batch(1, "sic", 7)          # 7 items: ragged split 4 + 3
    o = od.process_design(1, "sic")

    def estimate(eb, cb):
        return np.array([oe.linear_inv_process_estimate(o, eb[i]) for i in range(eb.shape[0])]).reshape(-1, 4, 4)

    full, (lo, hi) = run_sharded(estimate, [e, c], rank, world, dist=dist)
    assert (lo, hi) == shard_bounds(7, rank, world)
    # max-over-ranks timing reduction as in bench.py
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.item() == world
    # whole-job summary: sum of the local item counts and traces, max of the local block length
    local = full[lo:hi]
    sums, maxima = reduce_summary([hi - lo, float(np.trace(local, axis1=1, axis2=2).real.sum())], [hi - lo], dist)
    assert sums[0] == 7 and abs(sums[1] - np.trace(full, axis1=1, axis2=2).real.sum()) < 1e-12 and maxima[0] == 4
    np.save(os.path.join(out_dir, f"full_{rank}.npy"), full)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a = np.load(tmp_path / "full_0.npy"); b = np.load(tmp_path / "full_1.npy")
    assert a.shape == (7, 4, 4) and np.array_equal(a, b)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from fbx import synthetic
    from fbx_oracle import design as od, estimators as oe
    _, _, e, _ = synthetic.process_batch(1, "sic", 7)
    o = od.process_design(1, "sic")
    want = np.array([oe.linear_inv_process_estimate(o, e[i]) for i in range(7)])
    assert np.allclose(a, want, atol=1e-14)
