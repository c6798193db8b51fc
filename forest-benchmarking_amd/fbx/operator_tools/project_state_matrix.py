"""project_state_matrix_to_physical on MI355X (operator_tools/project_state_matrix.py:6-52)."""
import numpy as np

from .. import _lib


def project_state_matrix_to_physical_batch(rho) -> np.ndarray:
    x = _lib.c128(rho)
    x = x.reshape((-1,) + x.shape[-2:])
    d = x.shape[-1]
    if x.shape[-2] != d:
        raise ValueError("state matrices must be square")
    n = int(round(np.log2(d)))
    if 2 ** n != d or d > 8:
        return _project_general(x)
    out = np.empty_like(x)
    _lib.check(_lib.lib().fbx_proj_state_physical(n, x.shape[0], _lib.dptr(x.view(np.float64)),
                                                  _lib.dptr(out.view(np.float64))))
    return out


def _project_general(x) -> np.ndarray:
    """Any dimension up to 1024 (a qutrit, 4 and 5 qubits): the same algorithm on the generic device primitives --
    ``fbx_eigh`` of rho / tr(rho), the redistribution of the negative eigenvalues over the d numbers of the spectrum
    (project_state_matrix.py:37-48; control flow, done here), ``fbx_matmul`` for V diag(lambda') V^H.  Already
    physical inputs come back rescaled but otherwise untouched, as in the reference (:32-33)."""
    x = x / np.trace(x, axis1=1, axis2=2)[:, None, None]
    w, v = _lib.eigh_batch(x)
    d = x.shape[-1]
    new = np.array(w)
    for b in range(x.shape[0]):
        if w[b].min() >= 0:
            continue
        ev = w[b][::-1]                                    # descending
        i, acc = d, 0.0
        while ev[i - 1] + acc / float(i) < 0:
            acc += ev[i - 1]
            i -= 1
        out = np.zeros(d)
        out[:i] = ev[:i] + acc / float(i)
        new[b] = out[::-1]
    rebuilt = _lib.matmul_batch(v, v, conj_t_b=True, scale=new)
    keep = w.min(axis=1) >= 0
    rebuilt[keep] = x[keep]
    return rebuilt


def project_state_matrix_to_physical(rho: np.ndarray) -> np.ndarray:
    """Closest (2-norm) trace-one PSD matrix, Smolin-Gambetta-Smith."""
    return project_state_matrix_to_physical_batch(np.asarray(rho)[None])[0]
