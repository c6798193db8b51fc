"""The C-ABI library loads and exports every symbol include/fbx.h declares (no GPU needed)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "fbx.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fbx_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    import fbx
    path = fbx.library_path()
    if not os.path.exists(path):
        import __graft_entry__
        __graft_entry__.build()
    return ctypes.CDLL(path)


def test_every_declared_symbol_is_exported(lib):
    names = declared_symbols()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_binding_covers_the_header():
    from fbx import _lib
    assert sorted(_lib.PROTOTYPES) == declared_symbols()
    assert _lib.lib().fbx_version() >= 100


def test_no_device_fails_loudly_not_silently():
    """Without a GPU every compute entry point must report FBX_ERR_NO_DEVICE -- never fall back."""
    import numpy as np
    import fbx
    from fbx import _lib, tomography
    from fbx.design import state_design
    if fbx.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(fbx.FbxError) as ei:
        tomography.linear_inv_state_estimate_batch(state_design(1), np.zeros((1, 3)))
    assert ei.value.code == _lib.FBX_ERR_NO_DEVICE
    with pytest.raises(fbx.FbxError):
        from fbx import operator_tools as ot
        ot.kraus2choi(np.eye(2))


def test_argument_errors_map_to_value_error():
    """Bad arguments are rejected before any device work (ValueError like the reference)."""
    import numpy as np
    from fbx import _lib
    lib = _lib.lib()
    out = np.zeros(8)
    rc = lib.fbx_convert(_lib.REP_CHOI, _lib.REP_CHOI, 1, 1, _lib.dptr(out), 0, _lib.dptr(out))
    assert rc == _lib.FBX_ERR_BAD_ARG
    with pytest.raises(ValueError):
        _lib.check(rc)
    rc = lib.fbx_mle_state(None, 1, None, None, 0.1, 0.0, 0.0, 1e-9, 10, None, None, None)
    assert rc == _lib.FBX_ERR_BAD_ARG
    h = ctypes.c_void_p()
    rc = lib.fbx_design_create(6, 0, 1, None, None, None, ctypes.byref(h))           # state designs: 1..5 qubits
    assert rc == _lib.FBX_ERR_BAD_ARG and b"n_qubits" in lib.fbx_last_error()
    rc = lib.fbx_design_create(4, 1, 1, None, None, None, ctypes.byref(h))           # process designs: 1..3
    assert rc == _lib.FBX_ERR_BAD_ARG and b"n_qubits" in lib.fbx_last_error()


def test_options_are_validated_without_a_device(lib):
    """fbx_set_option / fbx_get_option: every option the header documents has its default, rejects values outside its range
    (a bad-argument code and a message, nothing changed) and reads back what was set."""
    lib.fbx_set_option.argtypes = [ctypes.c_char_p, ctypes.c_double]
    lib.fbx_get_option.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_double)]
    lib.fbx_last_error.restype = ctypes.c_char_p
    v = ctypes.c_double()
    defaults = {b"pgdb_eig_rel_tol": 1e-8, b"pgdb3_eig_rel_tol": 1e-7, b"eigh_cooperative": 1.0, b"pgdb_packed_1q": 1.0,
                b"pgdb_host_chunk": 4096.0, b"pgdb_pieces": 8.0, b"pgdb1_binned": 1.0}
    header = open(HEADER).read()
    for name, want in defaults.items():
        assert ('"%s"' % name.decode()) in header
        assert lib.fbx_get_option(name, ctypes.byref(v)) == 0 and v.value == want, name
    for name, bad in ((b"pgdb_pieces", 0.0), (b"pgdb_pieces", 65.0), (b"pgdb_pieces", 2.5), (b"pgdb1_binned", 3.0),
                      (b"pgdb_packed_1q", -1.0), (b"pgdb_host_chunk", 3.0), (b"pgdb_eig_rel_tol", 1.0), (b"no_such_option", 1.0)):
        assert lib.fbx_set_option(name, bad) != 0, (name, bad)
        assert lib.fbx_last_error()
    for name, ok in ((b"pgdb_pieces", 16.0), (b"pgdb1_binned", 2.0)):
        assert lib.fbx_set_option(name, ok) == 0
        assert lib.fbx_get_option(name, ctypes.byref(v)) == 0 and v.value == ok
        assert lib.fbx_set_option(name, defaults[name]) == 0
