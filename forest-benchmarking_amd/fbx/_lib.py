"""ctypes binding of libfbx.so (include/fbx.h).  No compute happens in Python."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("FBX_LIBRARY", os.path.join(os.path.dirname(_HERE), "libfbx.so"))

FBX_OK, FBX_ERR_BAD_ARG, FBX_ERR_HIP, FBX_ERR_NO_DEVICE, FBX_ERR_UNSUPPORTED, FBX_ERR_NOMEM, FBX_ERR_RCCL = range(7)
COMM_ID_BYTES = 128
COMM_SUM, COMM_MAX, COMM_MIN = range(3)
KIND_STATE, KIND_PROCESS = 0, 1
MODE_CONVERGE, MODE_FIXED = 0, 1
MODE_LS_REFERENCE = 0x100       # flag: the line search of tomography.py:575-585 taken literally (include/fbx.h)
REP_KRAUS, REP_CHOI, REP_SUPEROP, REP_PAULI_LIOUVILLE, REP_CHI = range(5)
PROJ_CP, PROJ_TP, PROJ_TNI, PROJ_PHYSICAL_TP, PROJ_PHYSICAL_TNI = range(5)
RAND_GINIBRE, RAND_UNITARY, RAND_STATE_VECTOR, RAND_GINIBRE_STATE, RAND_BURES_STATE = range(5)


class FbxError(RuntimeError):
    """A non-argument failure inside libfbx (HIP error, no device, unsupported size)."""

    def __init__(self, code, message):
        super().__init__(f"libfbx error {code}: {message}")
        self.code = code


def library_path() -> str:
    return _LIB_PATH


_lib = None

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_u8p = C.POINTER(C.c_uint8)
_vp = C.c_void_p
_i64 = C.c_int64

# name -> argtypes; every symbol include/fbx.h declares (tests/test_abi.py checks the list)
PROTOTYPES = {
    "fbx_version": [],
    "fbx_last_error": [],
    "fbx_device_count": [C.POINTER(C.c_int)],
    "fbx_set_device": [C.c_int],
    "fbx_set_devices": [C.POINTER(C.c_int), C.c_int],
    "fbx_device_name": [C.c_char_p, C.c_size_t, C.POINTER(C.c_int)],
    "fbx_device_id": [C.POINTER(C.c_int), C.c_char_p, C.c_size_t],
    "fbx_synchronize": [],
    "fbx_release_workspace": [],
    "fbx_comm_unique_id": [_u8p],
    "fbx_comm_init": [_u8p, C.c_int, C.c_int],
    "fbx_comm_init_timeout": [_u8p, C.c_int, C.c_int, C.c_double],
    "fbx_comm_query": [C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "fbx_comm_info": [C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "fbx_comm_destroy": [],
    "fbx_comm_allgather_dev": [_vp, _vp, C.c_size_t],
    "fbx_comm_broadcast_dev": [_vp, C.c_size_t, C.c_int],
    "fbx_comm_allreduce_f64_dev": [_vp, _vp, C.c_size_t, C.c_int],
    "fbx_comm_allreduce_f64": [_dp, C.c_size_t, C.c_int],
    "fbx_comm_barrier": [],
    "fbx_malloc": [C.POINTER(_vp), C.c_size_t],
    "fbx_free": [_vp],
    "fbx_host_alloc": [C.POINTER(_vp), C.c_size_t],
    "fbx_host_free": [_vp],
    "fbx_memcpy_h2d": [_vp, _vp, C.c_size_t],
    "fbx_memcpy_d2h": [_vp, _vp, C.c_size_t],
    "fbx_timer_begin": [],
    "fbx_timer_end": [_dp],
    "fbx_design_create": [C.c_int, C.c_int, C.c_int, _u8p, _u8p, _dp, C.POINTER(_vp)],
    "fbx_design_destroy": [_vp],
    "fbx_design_info": [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                        C.POINTER(C.c_int)],
    "fbx_pgdb_process": [_vp, _i64, _dp, _dp, C.c_int, C.c_int, C.c_int, _dp, _ip, _ip, _ip, _dp, _ip],
    "fbx_pgdb_process_dev": [_vp, _i64, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp],
    "fbx_pgdb_process_ex": [_vp, _i64, _dp, _dp, C.c_int, C.c_int, C.c_int, C.c_double, _dp, _ip, _ip, _ip, _dp, _ip, _ip, C.c_int],
    "fbx_pgdb_cost_grad": [_vp, _i64, _dp, _dp, C.c_double, _dp, _dp],
    "fbx_pgdb_cost_grad_dev": [_vp, _i64, _vp, _vp, C.c_double, _vp, _vp],
    "fbx_pgdb_process_ex_dev": [_vp, _i64, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_double, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int],
    "fbx_linv_process": [_vp, _i64, _dp, _dp],
    "fbx_linv_state": [_vp, _i64, _dp, _dp],
    "fbx_mle_state": [_vp, _i64, _dp, _dp, C.c_double, C.c_double, C.c_double, C.c_double,
                      C.c_int, _dp, _ip, _ip],
    "fbx_r_operator": [_vp, _i64, _dp, _dp, _dp],
    "fbx_state_log_likelihood": [_vp, _i64, _dp, _dp, _dp, _dp],
    "fbx_convert": [C.c_int, C.c_int, C.c_int, _i64, _dp, C.c_int, _dp],
    "fbx_kraus_sweep": [C.c_int, _i64, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp],
    "fbx_kraus_sweep_dev": [C.c_int, _i64, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp],
    "fbx_proj_choi": [C.c_int, C.c_int, _i64, _dp, _dp, _ip],
    "fbx_proj_state_physical": [C.c_int, _i64, _dp, _dp],
    "fbx_apply_choi": [C.c_int, _i64, _dp, _dp, _dp],
    "fbx_process_fidelity": [C.c_int, _i64, _dp, _dp, _dp, _dp],
    "fbx_state_measures": [C.c_int, _i64, _dp, _dp, _dp, _dp, _dp, _dp],
    "fbx_eigh": [C.c_int, _i64, _dp, _dp, _dp],
    "fbx_shots_to_moments": [C.c_int, _i64, _i64, _u8p, _u8p, _dp, C.c_int, _dp, _dp],
    "fbx_shots_to_moments_dev": [C.c_int, _i64, _i64, _vp, _vp, _vp, C.c_int, _vp, _vp],
    "fbx_calibrate_expectations": [_i64, _i64, _dp, _dp, _ip, _i64, _dp, _dp, _dp, _dp],
    "fbx_calibrate_expectations_dev": [_i64, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp],
    "fbx_kraus_pairs": [C.c_int, _i64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp],
    "fbx_kraus_pairs_dev": [C.c_int, _i64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp],
    "fbx_pauli_twirl_chi": [_i64, C.c_int, _dp, _dp],
    "fbx_pauli_twirl_chi_dev": [_i64, C.c_int, _vp, _vp],
    "fbx_dfe_estimate": [C.c_int, C.c_int, _i64, _i64, _dp, _dp, _dp, _dp],
    "fbx_convert_dev": [C.c_int, C.c_int, C.c_int, _i64, _vp, C.c_int, _vp],
    "fbx_proj_choi_dev": [C.c_int, C.c_int, _i64, _vp, _vp, _vp],
    "fbx_process_fidelity_dev": [C.c_int, _i64, _vp, _vp, _vp, _vp],
    "fbx_linv_process_dev": [_vp, _i64, _vp, _vp],
    "fbx_linv_state_dev": [_vp, _i64, _vp, _vp],
    "fbx_mle_state_dev": [_vp, _i64, _vp, _vp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, _vp, _vp, _vp],
    "fbx_r_operator_dev": [_vp, _i64, _vp, _vp, _vp],
    "fbx_state_log_likelihood_dev": [_vp, _i64, _vp, _vp, _vp, _vp],
    "fbx_proj_state_physical_dev": [C.c_int, _i64, _vp, _vp],
    "fbx_apply_choi_dev": [C.c_int, _i64, _vp, _vp, _vp],
    "fbx_state_measures_dev": [C.c_int, _i64, _vp, _vp, _vp, _vp, _vp, _vp],
    "fbx_eigh_dev": [C.c_int, _i64, _vp, _vp, _vp],
    "fbx_choi2kraus": [C.c_int, _i64, _dp, C.c_double, _dp, _ip],
    "fbx_choi2kraus_dev": [C.c_int, _i64, _vp, C.c_double, _vp, _vp],
    "fbx_convert_general": [C.c_int, C.c_int, C.c_int, _i64, _dp, C.c_int, _dp],
    "fbx_convert_general_dev": [C.c_int, C.c_int, C.c_int, _i64, _vp, C.c_int, _vp],
    "fbx_partial_trace": [C.c_int, C.c_int, C.c_int, _i64, _dp, _dp],
    "fbx_partial_trace_dev": [C.c_int, C.c_int, C.c_int, _i64, _vp, _vp],
    "fbx_matmul": [C.c_int, _i64, _dp, C.c_int, _dp, _dp, C.c_int, _dp],
    "fbx_matmul_dev": [C.c_int, _i64, _vp, C.c_int, _vp, _vp, C.c_int, _vp],
    "fbx_set_option": [C.c_char_p, C.c_double],
    "fbx_get_option": [C.c_char_p, _dp],
    "fbx_pauli_vector": [C.c_int, _i64, _dp, _dp],
    "fbx_pauli_vector_dev": [C.c_int, _i64, _vp, _vp],
    "fbx_random_operators": [C.c_int, C.c_int, C.c_int, _i64, C.c_uint64, _i64, _dp],
    "fbx_random_operators_dev": [C.c_int, C.c_int, C.c_int, _i64, C.c_uint64, _i64, _vp],
    "fbx_random_kraus": [C.c_int, _i64, C.c_int, C.c_uint64, _i64, _dp],
    "fbx_random_kraus_dev": [C.c_int, _i64, C.c_int, C.c_uint64, _i64, _vp],
    "fbx_beta_resample": [_i64, _i64, _dp, _dp, C.c_double, C.c_uint64, _dp],
    "fbx_beta_resample_dev": [_i64, _i64, _vp, _vp, C.c_double, C.c_uint64, _vp, _vp],
}


def lib():
    """The loaded library; raises (loudly) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise FbxError(-1, f"{_LIB_PATH} is missing: build it with "
                               f"`python -c 'import __graft_entry__ as g; g.build()'` "
                               f"(there is no CPU fallback)")
        handle = C.CDLL(_LIB_PATH)
        for name, argtypes in PROTOTYPES.items():
            fn = getattr(handle, name)
            fn.argtypes = argtypes
            fn.restype = C.c_char_p if name == "fbx_last_error" else C.c_int
        _lib = handle
    return _lib


def check(code):
    """Map an ABI return code to the reference's exception types."""
    if code == FBX_OK:
        return
    msg = lib().fbx_last_error().decode("utf-8", "replace")
    if code == FBX_ERR_BAD_ARG:
        raise ValueError(msg)
    raise FbxError(code, msg)


def device_count() -> int:
    n = C.c_int(0)
    check(lib().fbx_device_count(C.byref(n)))
    return n.value


def set_device(idx: int):
    """Select the GPU of this process (one process per GPU).  Designs created on another device go
    stale (the library refuses them), so the shim's design cache is dropped."""
    global _device_list
    check(lib().fbx_set_device(int(idx)))
    from . import design as _design
    _design._design_cache.clear()
    if _device_list and _device_list[0] != int(idx):
        check(lib().fbx_set_devices(None, 0))                # the list named another primary device: back to one device
        _device_list = ()


_device_list: tuple = ()


def set_devices(ids) -> tuple:
    """One call, several GPUs (fbx_set_devices): ``ids`` = ``'all'`` or a sequence of device ordinals; ids[0] becomes the
    process's device, and with more than one entry the host-pointer batch entry points (``pgdb_process_estimate_batch``,
    ``kraus_sweep``) split their batch into contiguous blocks, one per entry, on worker threads inside the library.  An
    entry may repeat (two workers share that GPU).  A one-entry list restores the single-device behaviour.  Returns the list."""
    global _device_list
    ids = tuple(range(device_count())) if isinstance(ids, str) and ids == "all" else tuple(int(i) for i in ids)
    if not ids:
        raise ValueError("set_devices: empty device list")
    if ids == _device_list:
        return ids
    before = C.c_int(-1)
    lib().fbx_device_id(C.byref(before), None, 0)
    arr = (C.c_int * len(ids))(*ids)
    check(lib().fbx_set_devices(arr, len(ids)))
    if before.value != ids[0]:
        from . import design as _design
        _design._design_cache.clear()
    _device_list = ids
    return ids


def release_workspace():
    """Give the calling thread's cached device workspaces / staging pool back (fbx_release_workspace)."""
    check(lib().fbx_release_workspace())


def set_option(name: str, value: float) -> None:
    """Process-wide tunable (fbx_set_option; include/fbx.h): 'pgdb_eig_rel_tol', 'pgdb3_eig_rel_tol'."""
    check(lib().fbx_set_option(name.encode(), float(value)))


def get_option(name: str) -> float:
    v = C.c_double(0.0)
    check(lib().fbx_get_option(name.encode(), C.byref(v)))
    return v.value


class option:
    """``with _lib.option('pgdb_eig_rel_tol', 0.0): ...`` -- set a tunable for a block, then restore it."""

    def __init__(self, name: str, value: float):
        self.name, self.value = name, value

    def __enter__(self):
        self.old = get_option(self.name)
        set_option(self.name, self.value)
        return self

    def __exit__(self, *exc):
        set_option(self.name, self.old)
        return False


def device_name():
    buf = C.create_string_buffer(256)
    cus = C.c_int(0)
    check(lib().fbx_device_name(buf, 256, C.byref(cus)))
    return buf.value.decode(), cus.value


def device_id():
    """(ordinal, PCI bus id) of the device this process selected."""
    buf = C.create_string_buffer(64)
    o = C.c_int(-1)
    check(lib().fbx_device_id(C.byref(o), buf, 64))
    return o.value, buf.value.decode()


def synchronize():
    check(lib().fbx_synchronize())


# ------------------------------------------------------------------ numpy <-> pointer helpers
def f64(a, shape=None):
    """C-contiguous float64 view/copy (complex128 arrays are viewed as interleaved pairs)."""
    a = np.ascontiguousarray(a)
    if np.iscomplexobj(a):
        a = np.ascontiguousarray(a, dtype=np.complex128).view(np.float64)
    else:
        a = np.ascontiguousarray(a, dtype=np.float64)
    return a


def c128(a):
    return np.ascontiguousarray(a, dtype=np.complex128)


def dptr(a):
    return None if a is None else a.ctypes.data_as(_dp)


def iptr(a):
    return None if a is None else a.ctypes.data_as(_ip)


def eigh_batch(a, eigenvectors=True):
    """numpy.linalg.eigh semantics (lower triangle, ascending) for stacked [B, N, N], any N in 1..1024: up to
    64 in LDS (the hot sizes are the powers of two; other sizes are zero-padded inside fbx_eigh), above that
    with the matrix in HBM (4- and 5-qubit Choi matrices; slow but on the device).  Larger matrices:
    FbxError(FBX_ERR_UNSUPPORTED) -- there is no host fallback."""
    a = c128(a)
    a = a.reshape((-1,) + a.shape[-2:])
    B, N = a.shape[0], a.shape[-1]
    if a.shape[-2] != N:
        raise ValueError("matrices must be square")
    if N > 1024:
        raise FbxError(FBX_ERR_UNSUPPORTED, f"fbx_eigh handles N <= 1024 (got {N}): matrices of more than 5 qubits "
                                            f"are outside this build, and there is no host fallback")
    w = np.empty((B, N))
    v = np.empty((B, N, N), dtype=np.complex128) if eigenvectors else None
    check(lib().fbx_eigh(N, B, dptr(a.view(np.float64)), dptr(w), dptr(v.view(np.float64)) if eigenvectors else None))
    return (w, v) if eigenvectors else w


def matmul_batch(a, b, conj_t_a=False, conj_t_b=False, scale=None):
    """op(a) diag(scale) op(b) for stacked [B, N, N] complex matrices on the device (fbx_matmul), N <= 1024."""
    a, b = c128(a), c128(b)
    a = a.reshape((-1,) + a.shape[-2:]); b = b.reshape((-1,) + b.shape[-2:])
    if a.shape != b.shape or a.shape[-1] != a.shape[-2]:
        raise ValueError("operands must be stacks of square matrices of one shape")
    B, N = a.shape[0], a.shape[-1]
    sc = None if scale is None else np.ascontiguousarray(np.asarray(scale, dtype=np.float64).reshape(B, N))
    out = np.empty((B, N, N), dtype=np.complex128)
    check(lib().fbx_matmul(N, B, dptr(a.view(np.float64)), int(bool(conj_t_a)), dptr(sc), dptr(b.view(np.float64)),
                           int(bool(conj_t_b)), dptr(out.view(np.float64))))
    return out


class _PinnedBlock:
    """Owner of one fbx_host_alloc block; freed when the last numpy view of it goes away."""

    def __init__(self, nbytes):
        p = _vp()
        check(lib().fbx_host_alloc(C.byref(p), int(nbytes)))
        self.ptr, self.nbytes = p, int(nbytes)

    def __del__(self):
        try:
            if self.ptr is not None:
                lib().fbx_host_free(self.ptr)
                self.ptr = None
        except Exception:
            pass


def pinned_empty(shape, dtype=np.float64):
    """A numpy array in page-locked host memory (fbx_host_alloc): hand such arrays to the host-pointer entry points and
    the transfers run at the full PCIe rate, overlapped with the kernels (fbx_pgdb_process pipelines them)."""
    dt = np.dtype(dtype)
    n = int(np.prod(shape)) * dt.itemsize
    blk = _PinnedBlock(max(n, 16))
    buf = (C.c_char * max(n, 16)).from_address(blk.ptr.value)
    arr = np.frombuffer(buf, dtype=dt, count=int(np.prod(shape))).reshape(shape)
    buf._owner = blk
    return arr


def pinned_copy(a):
    out = pinned_empty(np.shape(a), np.asarray(a).dtype)
    out[...] = a
    return out


class DeviceBuffer:
    """A caller-owned HBM allocation (for batches that stay resident between calls)."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        p = _vp()
        check(lib().fbx_malloc(C.byref(p), self.nbytes))
        self.ptr = p

    @classmethod
    def from_array(cls, arr):
        arr = np.ascontiguousarray(arr)
        buf = cls(arr.nbytes)
        check(lib().fbx_memcpy_h2d(buf.ptr, arr.ctypes.data_as(_vp), arr.nbytes))
        return buf

    def to_array(self, dtype, shape):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        check(lib().fbx_memcpy_d2h(out.ctypes.data_as(_vp), self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr is not None:
            lib().fbx_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
