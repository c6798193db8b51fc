"""Sharding of independent reconstructions over the GPUs of one node (one process per GPU).

The batch axis is embarrassingly parallel: rank g owns the contiguous block
``[g * ceil(B / G), min(B, (g + 1) * ceil(B / G)))`` of items, runs the same single-GPU entry
points on it and, when a caller wants the whole result everywhere, the slabs are exchanged with
one all-gather (RCCL over xGMI under the ``nccl`` backend; ``gloo`` on CPU for tests).  There is
no collective on the data path of the estimators themselves.
"""
from typing import Callable, Sequence, Tuple

import numpy as np


def shard_bounds(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block partition; trailing ranks may get an empty range."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("need 0 <= rank < world")
    per = -(-n_items // world)
    lo = min(n_items, rank * per)
    return lo, min(n_items, lo + per)


def run_sharded(fn: Callable[..., np.ndarray], arrays: Sequence[np.ndarray], rank: int, world: int,
                dist=None, gather: bool = True):
    """Apply ``fn(*[a[lo:hi] for a in arrays])`` to this rank's block of the leading axis.

    With ``gather`` and an initialised ``torch.distributed`` module in ``dist`` every rank gets the
    concatenated result of all ranks (padded all-gather of equally sized slabs); otherwise the local
    block and its bounds are returned."""
    n = arrays[0].shape[0]
    lo, hi = shard_bounds(n, rank, world)
    local = fn(*[a[lo:hi] for a in arrays])
    if not gather or dist is None or world == 1:
        return local, (lo, hi)
    import torch
    per = -(-n // world)
    slab = np.zeros((per,) + local.shape[1:], dtype=local.dtype)
    slab[: hi - lo] = local
    is_complex = np.iscomplexobj(slab)
    view = slab.view(np.float64) if is_complex else slab
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.from_numpy(np.ascontiguousarray(view)).to(dev)
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t)
    full = np.concatenate([o.cpu().numpy() for o in outs], axis=0)
    if is_complex:
        full = full.view(np.complex128)
    return full[:n], (lo, hi)


def reduce_summary(sums: Sequence[float], maxima: Sequence[float] = (), dist=None):
    """Whole-job summary scalars: element-wise SUM of ``sums`` (e.g. sum of
    fidelities, of iteration counts, number of items that hit the cap) and MAX of ``maxima`` (e.g.
    most halvings, slowest shard) over all ranks -- two all-reduces on vectors of a few doubles,
    the only other collective next to the optional all-gather.  Returns two numpy arrays; without an
    initialised ``dist`` the inputs are returned unchanged."""
    s = np.asarray(list(sums), dtype=np.float64)
    m = np.asarray(list(maxima), dtype=np.float64)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return s, m
    import torch
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    if s.size:
        t = torch.from_numpy(s.copy()).to(dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        s = t.cpu().numpy()
    if m.size:
        t = torch.from_numpy(m.copy()).to(dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        m = t.cpu().numpy()
    return s, m
