"""fbx_eigh on a batch of 64 x 64 Hermitian matrices: time and residual (FBX_LIBRARY selects the build)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "forest-benchmarking_amd"))
from fbx import _lib
rs = np.random.RandomState(0); B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
g = rs.randn(B, 64, 64) + 1j * rs.randn(B, 64, 64); h = g + g.conj().transpose(0, 2, 1)
for rep in range(3):
    t = time.time(); w, v = _lib.eigh_batch(h); dt = time.time() - t
print(os.path.basename(os.environ.get("FBX_LIBRARY", "libfbx.so")), "eigh 64x64 x", B, "%.1f ms (host call, transfers inside)" % (1e3 * dt), "residual %.2e" % np.abs(h @ v - v * w[:, None, :]).max(),
      "ascending", bool((np.diff(w, axis=1) >= 0).all()))
