import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "forest-benchmarking_amd"), os.path.join(ROOT, "oracle"),
          os.path.join(ROOT, "tests", "golden"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)
sys.dont_write_bytecode = True


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


def pytest_collection_modifyitems(config, items):
    ref_ok = os.path.isdir("/root/reference/forest/benchmarking")
    skip_ref = pytest.mark.skip(reason="/root/reference not present (GPU box)")
    for item in items:
        if "reference" in item.keywords and not ref_ok:
            item.add_marker(skip_ref)


@pytest.fixture(scope="session")
def gpu():
    """The loaded library bound to device 0; fails loudly if the HIP path is unavailable."""
    import fbx
    from fbx import _lib
    assert os.path.exists(fbx.library_path()), "libfbx.so missing: run __graft_entry__.build()"
    n = fbx.device_count()
    assert n >= 1, "no HIP device visible: -m gpu tests need a real MI355X"
    _lib.set_device(0)
    return _lib
