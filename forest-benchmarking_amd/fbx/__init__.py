"""fbx -- MI355X-native tomography reconstruction behind forest-benchmarking's own API.

Host-side mirror of the reference's interface for the hot path
(forest/benchmarking/tomography.py estimators, operator_tools, distance_measures): same
function names, argument meaning, return types and error behaviour, plus ``*_batch``
variants that take SoA arrays for many experiments sharing one design.  All arithmetic runs
in hand-written HIP kernels inside ``libfbx.so`` (C ABI in ``include/fbx.h``) reached through
ctypes; there is no CPU fallback -- importing the compute entry points without the library or
without a GPU raises.
"""
from . import _lib  # noqa: F401
from ._lib import FbxError, library_path, device_count  # noqa: F401

__version__ = "0.1.0"
