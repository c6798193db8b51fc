"""Device-resident throughput of fbx_shots_to_moments_dev: bitstrings -> +-1 products -> mean / variance.
usage: python scripts/shots_time.py [n_qubits] [n_settings] [n_shots]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "forest-benchmarking_amd"))
import numpy as np
from fbx import _lib
_lib.set_device(0)
lib = _lib.lib()
for n, S, shots in ([(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]))] if len(sys.argv) > 3 else
                    [(2, 540 * 256, 1000), (2, 540 * 64, 10000), (3, 4032 * 16, 1000), (3, 4032 * 16, 1003), (8, 65536, 1000), (5, 65536, 1000), (6, 65536, 1000), (7, 32768, 1001), (9, 32768, 1000)]):
    rs = np.random.RandomState(1)
    bits = rs.randint(0, 2, size=(S, shots, n)).astype(np.uint8)
    mask = rs.randint(0, 2, size=(S, n)).astype(np.uint8); mask[:, 0] |= (mask.sum(1) == 0)
    d_bits, d_mask = _lib.DeviceBuffer.from_array(bits), _lib.DeviceBuffer.from_array(mask)
    d_mean, d_var = _lib.DeviceBuffer(S * 8), _lib.DeviceBuffer(S * 8)
    ms = ctypes.c_double()
    best = 1e9
    for rep in range(6):
        _lib.check(lib.fbx_timer_begin())
        _lib.check(lib.fbx_shots_to_moments_dev(n, S, shots, d_bits.ptr, d_mask.ptr, None, 0, d_mean.ptr, d_var.ptr))
        _lib.check(lib.fbx_timer_end(ctypes.byref(ms)))
        if rep: best = min(best, ms.value)
    mean = d_mean.to_array(np.float64, (S,))
    prod = np.where(mask[:, None, :] != 0, 1 - 2 * bits.astype(np.int8), 1).prod(axis=2)
    ok = np.array_equal(mean[:256], prod[:256].mean(axis=1))
    print(f"n={n} settings={S} shots={shots}: {best:.3f} ms  {bits.nbytes / best / 1e6:.0f} GB/s  {S / best / 1e3:.1f} M settings/s  exact={ok}")
    for b in (d_bits, d_mask, d_mean, d_var): b.free()
