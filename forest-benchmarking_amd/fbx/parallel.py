"""Sharding of independent reconstructions over the GPUs of one node -- one process per GPU.

The batch axis is embarrassingly parallel (the reference's natural units are the
entries of ``get_results_by_qubit_groups``, observable_estimation.py:1145-1173, and the bootstrap
resamples of tomography.py:440-451): rank g owns the contiguous block
``[g * ceil(B / G), min(B, (g + 1) * ceil(B / G)))`` of items and runs the single-GPU entry points on
it.  There is NO collective on the data path of the estimators.  What ranks exchange around it goes
through libfbx's RCCL entry points (``fbx_comm_*``, RCCL over xGMI): an all-gather of result slabs
when a caller wants every rank to hold the whole result, an all-reduce of a summary vector of a few
doubles, a broadcast of design-sized constants.

No torch here: the communicator lives in libfbx.so.  A communicator object only needs ``rank``,
``world``, ``allgather(array)``, ``allreduce(vector, op)`` and ``barrier()``, so the CPU tests drive
the same partition / gather code through a gloo-backed stand-in (tests/test_distributed_gloo.py).
"""
import os
import threading
import time
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np


def shard_bounds(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block partition; trailing ranks may get an empty range."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("need 0 <= rank < world")
    per = -(-n_items // world)
    lo = min(n_items, rank * per)
    return lo, min(n_items, lo + per)


# ------------------------------------------------------------------------------------ rendezvous
class FileRendezvous:
    """Host-side hand-off between the ranks of ONE node through a private directory: every rank
    publishes a small payload under a tag (atomic rename) and reads everybody else's.  Used to pass
    the RCCL unique id from rank 0 to the others, and as the barrier of last resort when RCCL cannot
    be initialised (e.g. several test ranks sharing one GPU)."""

    def __init__(self, rank: int, world: int, directory: Optional[str] = None, timeout: float = 300.0):
        if directory is None:
            directory = os.environ.get("FBX_RDZV_DIR")
        if directory is None:
            # torchrun children share their parent (the elastic agent) and MASTER_PORT
            directory = os.path.join(os.environ.get("TMPDIR", "/tmp"),
                                     f"fbx_rdzv_{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}")
        os.makedirs(directory, exist_ok=True)
        self.dir, self.rank, self.world, self.timeout = directory, int(rank), int(world), timeout
        self._mine: List[str] = []
        self._seq = 0

    def allgather(self, payload: bytes, tag: Optional[str] = None) -> List[bytes]:
        if tag is None:
            tag = f"x{self._seq}"
            self._seq += 1
        path = os.path.join(self.dir, f"{tag}.{self.rank}")
        tmp = path + ".tmp"
        with open(tmp, "wb") as f:
            f.write(payload)
        os.replace(tmp, path)
        self._mine.append(path)
        out, deadline = [], time.monotonic() + self.timeout
        for r in range(self.world):
            p = os.path.join(self.dir, f"{tag}.{r}")
            delay = 1e-4
            while not os.path.exists(p):
                if time.monotonic() > deadline:
                    raise TimeoutError(f"rendezvous: rank {r} never published '{tag}' in {self.dir}")
                time.sleep(delay)
                delay = min(delay * 1.5, 2e-3)
            with open(p, "rb") as f:
                out.append(f.read())
        return out

    def barrier(self):
        self.allgather(b"")

    def close(self):
        # a rank's files may only go once every rank has read them: one last round, then everybody
        # deletes what it wrote (the directory goes with the last file)
        try:
            self.allgather(b"", tag="close")
            time.sleep(0.05)
        except TimeoutError:
            pass
        for p in self._mine:
            try:
                os.remove(p)
            except OSError:
                pass
        self._mine = []
        try:
            os.rmdir(self.dir)
        except OSError:
            pass


# ------------------------------------------------------------------------------------ communicators
class LocalComm:
    """world = 1: every collective is the identity."""
    rank, world, backend = 0, 1, "local"

    def allgather(self, arr):
        return np.asarray(arr)[None].copy()

    def allreduce(self, vec, op="sum"):
        return np.array(vec, dtype=np.float64, copy=True)

    def barrier(self):
        from . import _lib
        _lib.synchronize()

    def close(self):
        pass


class RcclComm:
    """libfbx's RCCL communicator (``fbx_comm_*``): one rank per process / GPU."""
    backend = "rccl"
    _OPS = {"sum": 0, "max": 1, "min": 2}

    def __init__(self, rank: int, world: int, unique_id: bytes):
        from . import _lib
        self._lib = _lib
        ident = (_lib.C.c_uint8 * _lib.COMM_ID_BYTES).from_buffer_copy(unique_id)
        _lib.check(_lib.lib().fbx_comm_init(ident, int(rank), int(world)))
        self.rank, self.world = int(rank), int(world)
        v = _lib.C.c_int(0)
        _lib.lib().fbx_comm_info(None, None, _lib.C.byref(v))
        self.rccl_version = v.value

    @staticmethod
    def new_unique_id() -> bytes:
        from . import _lib
        ident = (_lib.C.c_uint8 * _lib.COMM_ID_BYTES)()
        _lib.check(_lib.lib().fbx_comm_unique_id(ident))
        return bytes(ident)

    def allgather_dev(self, d_send, d_recv, bytes_per_rank: int):
        """Device pointers (DeviceBuffer.ptr); asynchronous on the library stream."""
        self._lib.check(self._lib.lib().fbx_comm_allgather_dev(d_send, d_recv, int(bytes_per_rank)))

    def allgather(self, arr):
        """Equal-shaped host array per rank -> stacked ``[world, *arr.shape]`` on every rank."""
        _lib = self._lib
        a = np.ascontiguousarray(arr)
        send = _lib.DeviceBuffer.from_array(a) if a.nbytes else None
        recv = _lib.DeviceBuffer(max(16, a.nbytes * self.world))
        if a.nbytes:
            self.allgather_dev(send.ptr, recv.ptr, a.nbytes)
        _lib.synchronize()
        out = recv.to_array(a.dtype, (self.world,) + a.shape) if a.nbytes else np.empty((self.world,) + a.shape, a.dtype)
        recv.free()
        if send is not None:
            send.free()
        return out

    def allreduce(self, vec, op="sum"):
        v = np.ascontiguousarray(np.array(vec, dtype=np.float64, copy=True))
        if v.size:
            self._lib.check(self._lib.lib().fbx_comm_allreduce_f64(self._lib.dptr(v), v.size, self._OPS[op]))
        return v

    def broadcast_dev(self, d_buf, nbytes: int, root: int = 0):
        self._lib.check(self._lib.lib().fbx_comm_broadcast_dev(d_buf, int(nbytes), int(root)))

    def barrier(self):
        self._lib.check(self._lib.lib().fbx_comm_barrier())

    def close(self):
        self._lib.check(self._lib.lib().fbx_comm_destroy())


class HostComm:
    """Fallback with the communicator interface on top of FileRendezvous (host memory only): for ranks
    that cannot form an RCCL communicator.  Never the product path on a multi-GPU node."""
    backend = "host-files"

    def __init__(self, rdzv: FileRendezvous):
        self._r = rdzv
        self.rank, self.world = rdzv.rank, rdzv.world

    def allgather(self, arr):
        a = np.ascontiguousarray(arr)
        parts = self._r.allgather(a.tobytes())
        return np.stack([np.frombuffer(p, dtype=a.dtype).reshape(a.shape) for p in parts])

    def allreduce(self, vec, op="sum"):
        g = self.allgather(np.asarray(vec, dtype=np.float64))
        return {"sum": g.sum(axis=0), "max": g.max(axis=0), "min": g.min(axis=0)}[op]

    def barrier(self):
        from . import _lib
        _lib.synchronize()
        self._r.barrier()

    def close(self):
        pass


def init_from_env(allow_host_fallback: bool = False, allow_oversubscribe: bool = False):
    """The launcher contract of ``torchrun`` / ``bench.py``'s own spawner: RANK, LOCAL_RANK, WORLD_SIZE
    (and MASTER_PORT for the rendezvous directory) in the environment.  Selects GPU LOCAL_RANK, forms
    the RCCL communicator (rank 0's unique id travels through a FileRendezvous) and returns
    ``(comm, rendezvous)``; world 1 gives a LocalComm and no rendezvous.  ``allow_host_fallback``: when
    the ranks cannot form an RCCL communicator, return a HostComm (with ``.failure`` saying why)
    instead of raising; ``allow_oversubscribe`` (testing) additionally lets ranks share a GPU."""
    from . import _lib
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    ndev = _lib.device_count()
    if ndev < 1:
        _lib.set_device(0)                                   # raises: no device, no fallback
    if local >= ndev and not (allow_host_fallback and allow_oversubscribe):
        raise _lib.FbxError(_lib.FBX_ERR_NO_DEVICE, f"LOCAL_RANK {local} but only {ndev} GPU(s) visible")
    _lib.set_device(local % ndev)
    if world == 1:
        return LocalComm(), None
    rdzv = FileRendezvous(rank, world)
    try:
        ident = RcclComm.new_unique_id() if rank == 0 else b""
        err = b""
    except Exception as exc:                                 # rank 0 could not even load RCCL
        ident, err = b"", str(exc).encode()
    ids = rdzv.allgather(ident + b"|" + err, tag="rccl_id")
    ident, _, err = ids[0].rpartition(b"|")
    failure = err.decode() if err else None
    comm = None
    if failure is None and (ndev >= world or not allow_host_fallback):
        # ncclCommInitRank is a collective: it blocks for as long as a peer is missing.  It runs in a
        # helper thread so that a rank whose peer died reports a failure instead of hanging the job.
        box: dict = {}

        def _init():
            try:
                box["comm"] = RcclComm(rank, world, ident)
            except Exception as exc:                         # noqa: BLE001 -- reported below
                box["error"] = str(exc)

        limit = float(os.environ.get("FBX_RCCL_INIT_TIMEOUT", "180"))
        worker = threading.Thread(target=_init, name="fbx-rccl-init", daemon=True)
        worker.start()
        worker.join(limit)
        if worker.is_alive():
            failure = f"ncclCommInitRank did not return within {limit:.0f} s"
        else:
            comm, failure = box.get("comm"), box.get("error")
    elif failure is None:
        failure = f"{world} ranks share {ndev} GPU(s): RCCL needs one GPU per rank"
    # all ranks must agree on the transport
    oks = rdzv.allgather(b"1" if comm is not None else b"0", tag="rccl_ok")
    if all(o == b"1" for o in oks):
        return comm, rdzv
    if comm is not None:
        comm.close()
    if not allow_host_fallback:
        raise _lib.FbxError(_lib.FBX_ERR_RCCL, failure or "another rank failed to initialise RCCL")
    host = HostComm(rdzv)
    host.failure = failure or "another rank failed to initialise RCCL"
    return host, rdzv


# ------------------------------------------------------------------------------------ sharded runs
def run_sharded(fn: Callable[..., np.ndarray], arrays: Sequence[np.ndarray], comm=None, gather: bool = True):
    """Apply ``fn(*[a[lo:hi] for a in arrays])`` to this rank's block of the leading axis.

    With ``gather`` and a communicator of more than one rank every rank gets the concatenated result
    of all ranks (one all-gather of equally sized, zero-padded slabs); otherwise the local block.
    Returns ``(result, (lo, hi))``."""
    comm = comm or LocalComm()
    n = arrays[0].shape[0]
    lo, hi = shard_bounds(n, comm.rank, comm.world)
    local = np.asarray(fn(*[a[lo:hi] for a in arrays]))
    if not gather or comm.world == 1:
        return local, (lo, hi)
    per = -(-n // comm.world)
    slab = np.zeros((per,) + local.shape[1:], dtype=local.dtype)
    slab[: hi - lo] = local
    full = comm.allgather(slab)
    return full.reshape((-1,) + local.shape[1:])[:n], (lo, hi)


def reduce_summary(sums: Sequence[float], maxima: Sequence[float] = (), comm=None):
    """Whole-job summary scalars: element-wise SUM of ``sums`` (e.g. sum of fidelities,
    of iteration counts, number of items that hit the cap) and MAX of ``maxima`` (e.g. most halvings,
    slowest shard) over all ranks -- two all-reduces on vectors of a few doubles.  Returns two numpy
    arrays; without a communicator the inputs are returned unchanged."""
    s = np.asarray(list(sums), dtype=np.float64)
    m = np.asarray(list(maxima), dtype=np.float64)
    if comm is None or comm.world == 1:
        return s, m
    if s.size:
        s = comm.allreduce(s, "sum")
    if m.size:
        m = comm.allreduce(m, "max")
    return s, m
