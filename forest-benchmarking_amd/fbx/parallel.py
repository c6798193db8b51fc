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
    be initialised (e.g. several test ranks sharing one GPU).

    A directory can outlive a failed attempt (torchrun restarts its workers with the same parent and
    MASTER_PORT; FBX_RDZV_DIR may be reused), so nothing in it is trusted by name alone: rank 0 opens a
    GENERATION -- it removes whatever the directory held and publishes a fresh random nonce under the one
    fixed name ``gen`` -- and every other file name carries that nonce.  A rank that still sees the
    previous attempt's ``gen`` joins a generation nobody else is in; every ``join_timeout`` without an ack it
    re-reads ``gen`` and joins again once the nonce has changed (a late ack alone is not a reason to leave: rank 0
    acknowledges only after the LAST rank has joined).  ``close`` is acknowledged: files go only after every rank
    has said it will read no more."""

    def __init__(self, rank: int, world: int, directory: Optional[str] = None, timeout: float = 300.0,
                 join_timeout: float = 5.0):
        if directory is None:
            directory = os.environ.get("FBX_RDZV_DIR")
        if directory is None:
            # torchrun children share their parent (the elastic agent) and MASTER_PORT; the restart count
            # separates the attempts of one agent
            directory = os.path.join(os.environ.get("TMPDIR", "/tmp"),
                                     "fbx_rdzv_%s_%s_%d_r%s" % (os.getuid(), os.environ.get("MASTER_PORT", "0"), os.getppid(),
                                                                os.environ.get("TORCHELASTIC_RESTART_COUNT", "0")))
        os.makedirs(directory, mode=0o700, exist_ok=True)
        st = os.stat(directory)
        if st.st_uid != os.getuid() or (st.st_mode & 0o022):
            raise PermissionError(f"rendezvous directory {directory} is not private to this user")
        self.dir, self.rank, self.world, self.timeout = directory, int(rank), int(world), timeout
        self._mine: List[str] = []
        self._seq = 0
        self.gen = self._open_generation(join_timeout)

    # -- generation handshake
    def _open_generation(self, join_timeout: float) -> str:
        """Rank 0: purge, publish a fresh ``gen`` nonce, collect every rank's join TOKEN, echo the tokens in an
        ``ack``.  Rank r: read ``gen``, publish a fresh random token, and accept the generation only if rank 0's
        ack echoes that token -- files of a dead attempt can look complete, but cannot contain a token drawn now."""
        gen_path = os.path.join(self.dir, "gen")
        deadline = time.monotonic() + self.timeout
        if self.rank == 0:
            for name in os.listdir(self.dir):                 # leftovers of an earlier attempt
                try:
                    os.remove(os.path.join(self.dir, name))
                except OSError:
                    pass
            self.gen = os.urandom(8).hex()
            tmp = gen_path + ".tmp0"
            with open(tmp, "w") as f:
                f.write(self.gen)
            os.replace(tmp, gen_path)
            self._publish(b"rank0", "join")
            tokens = self._collect("join", self.timeout)
            self._publish(b",".join(tokens), "ack")
            return self.gen
        # A late ack is NOT a stale generation: rank 0 acks only once EVERY rank has joined, so with three or more ranks
        # an early rank may wait for a straggler far longer than ``join_timeout``.  The join file and its token stay in
        # place while ``gen`` is unchanged; the rank re-joins only when ``gen`` has changed (rank 0 of a new attempt
        # purged the directory), when its files were purged under it, or when the ack of this generation does not echo
        # its token (a complete-looking dead attempt).  Only such a generation is skipped from then on.
        dead = set()
        last_error = None                                      # a PERSISTENT OSError (ENOSPC, EACCES, a stale NFS handle) must end in
        delay = 1e-3                                           # an error at the deadline, not in a rank spinning at 100 % CPU
        while True:
            if time.monotonic() > deadline:
                if last_error is not None:
                    raise last_error
                raise TimeoutError(f"rendezvous: could not join a generation in {self.dir}")
            if last_error is not None:                         # back off between attempts that failed on the file system
                time.sleep(delay)
                delay = min(delay * 2, 0.1)
            self.gen = self._read(gen_path, deadline, skip=dead)
            token = os.urandom(8).hex().encode()
            self._mine = []
            try:
                self._publish(token, "join")
            except OSError as exc:                             # rank 0 purged the directory under us: read ``gen`` again
                last_error = exc
                continue
            while True:
                try:
                    ack = self._collect("ack", join_timeout, ranks=(0,))[0].split(b",")
                    if len(ack) == self.world and ack[self.rank] == token:
                        return self.gen
                    dead.add(self.gen)                         # an ack that cannot be about this join: a dead attempt
                    last_error = None
                    break
                except TimeoutError:
                    pass
                except OSError as exc:
                    last_error = exc
                    break
                if time.monotonic() > deadline:
                    raise TimeoutError(f"rendezvous: rank 0 never acknowledged generation {self.gen} in {self.dir}")
                try:
                    with open(gen_path, "r") as f:
                        current = f.read()
                except OSError:
                    current = ""                               # mid-purge: the new ``gen`` is about to appear
                if current != self.gen or not os.path.exists(self._mine[-1]):
                    break                                      # a new generation (or our join file is gone): join again

    def _read(self, path: str, deadline: float, skip=()) -> str:
        delay = 1e-4
        while True:
            try:
                with open(path, "r") as f:
                    val = f.read()
                if val and val not in skip:
                    return val
            except (FileNotFoundError, PermissionError):
                pass
            if time.monotonic() > deadline:
                raise TimeoutError(f"rendezvous: {path} never appeared")
            time.sleep(delay)
            delay = min(delay * 1.5, 2e-3)

    def _publish(self, payload: bytes, tag: str) -> str:
        path = os.path.join(self.dir, f"{self.gen}.{tag}.{self.rank}")
        tmp = path + ".tmp"
        with open(tmp, "wb") as f:
            f.write(payload)
        os.replace(tmp, path)
        self._mine.append(path)
        return path

    def _collect(self, tag: str, timeout: float, ranks=None) -> List[bytes]:
        out, deadline = [], time.monotonic() + timeout
        for r in (range(self.world) if ranks is None else ranks):
            p = os.path.join(self.dir, f"{self.gen}.{tag}.{r}")
            delay = 1e-4
            while True:
                try:
                    with open(p, "rb") as f:              # (no exists()-then-open window)
                        out.append(f.read())
                    break
                except FileNotFoundError:
                    pass
                if time.monotonic() > deadline:
                    raise TimeoutError(f"rendezvous: rank {r} never published '{tag}' in {self.dir}")
                time.sleep(delay)
                delay = min(delay * 1.5, 2e-3)
        return out

    def _exchange(self, payload: bytes, tag: str, timeout: float) -> List[bytes]:
        self._publish(payload, tag)
        return self._collect(tag, timeout)

    def allgather(self, payload: bytes, tag: Optional[str] = None) -> List[bytes]:
        if tag is None:
            tag = f"x{self._seq}"
            self._seq += 1
        return self._exchange(payload, tag, self.timeout)

    def barrier(self):
        self.allgather(b"")

    def close(self):
        """Acknowledged: "closing" is a full round (every rank has read everything it will ever read, so data
        files may go); a rank then publishes "closed" -- its promise to read nothing more, not even the
        "closing" markers -- and leaves without waiting.  Rank 0 waits for every "closed" and removes what is
        left, directory included.  No rank sleeps and hopes; no rank polls for a file a peer already deleted."""
        try:
            self._exchange(b"", "closing", self.timeout)
        except TimeoutError:
            self._mine = []
            return
        for p in self._mine[:-1]:                             # all but my "closing" marker, which peers may still read
            try:
                os.remove(p)
            except OSError:
                pass
        self._mine = []
        self._publish(b"", "closed")
        self._mine = []
        if self.rank == 0:
            try:
                self._collect("closed", self.timeout)
            except TimeoutError:
                return
            for name in os.listdir(self.dir):
                try:
                    os.remove(os.path.join(self.dir, name))
                except OSError:
                    pass
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

    def __init__(self, rank: int, world: int, unique_id: bytes, timeout: Optional[float] = None):
        """Collective.  ``timeout`` (seconds; default FBX_RCCL_INIT_TIMEOUT or 180): a missing peer makes the
        call fail with FbxError(FBX_ERR_RCCL) instead of blocking for ever (fbx_comm_init_timeout)."""
        from . import _lib
        self._lib = _lib
        ident = (_lib.C.c_uint8 * _lib.COMM_ID_BYTES).from_buffer_copy(unique_id)
        if timeout is None:
            _lib.check(_lib.lib().fbx_comm_init(ident, int(rank), int(world)))
        else:
            _lib.check(_lib.lib().fbx_comm_init_timeout(ident, int(rank), int(world), float(timeout)))
        v = _lib.C.c_int(0)
        _lib.lib().fbx_comm_info(None, None, _lib.C.byref(v))
        self.rccl_version = v.value
        # rank / world / device as the communicator itself reports them (not the launcher's environment)
        q = self.query()
        self.rank, self.world, self.device = q["rank"], q["world"], q["device"]
        if (self.rank, self.world) != (int(rank), int(world)):
            _lib.lib().fbx_comm_destroy()                     # (or the next init fails with "a communicator already exists")
            raise _lib.FbxError(_lib.FBX_ERR_RCCL, f"communicator reports rank {self.rank} of {self.world}, asked for {rank} of {world}")

    def query(self) -> dict:
        C = self._lib.C
        r, w, d = C.c_int(-1), C.c_int(-1), C.c_int(-1)
        self._lib.check(self._lib.lib().fbx_comm_query(C.byref(r), C.byref(w), C.byref(d)))
        return {"rank": r.value, "world": w.value, "device": d.value}

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
        # ncclCommInitRank is a collective; fbx_comm_init_timeout bounds the wait for a missing peer inside the
        # library (no Python helper thread, no lock held meanwhile)
        try:
            comm = RcclComm(rank, world, ident, timeout=float(os.environ.get("FBX_RCCL_INIT_TIMEOUT", "180")))
        except Exception as exc:                             # noqa: BLE001 -- reported to every rank below
            failure = str(exc)
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
