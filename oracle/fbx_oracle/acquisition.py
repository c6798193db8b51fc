"""Shots -> observable moments -- CPU restatement of observable_estimation.py:804-853, :1052-1090.

TEST INFRASTRUCTURE (see package docstring).  Pinned by the exact values of the reference's tests
(tests/test_observable_estimation.py:521-550) and, in the build container, against the reference."""
import numpy as np
from scipy.stats import beta


def shots_to_obs_moments(bitarray, obs_mask, coeff=1.0, use_beta_dist_unbiased_prior=False):
    """obs_mask[q] != 0 where the observable acts on column q of the [n_shots, n_qubits] bit array."""
    idxs = [i for i, m in enumerate(obs_mask) if m]
    if len(idxs) == 0:
        return coeff, 0
    obs_strings = np.asarray(bitarray)[:, idxs]
    obs_vals = np.prod(1 - 2 * obs_strings.astype(np.int64), axis=1)
    if use_beta_dist_unbiased_prior:
        n_minus, n_plus = np.bincount(obs_vals == 1, minlength=2)
        m, v = beta.mean(n_plus + 1, n_minus + 1), beta.var(n_plus + 1, n_minus + 1)
        return (2 * m - 1) * coeff, 4 * v * coeff ** 2
    obs_vals = coeff * obs_vals
    return np.mean(obs_vals).item(), np.var(obs_vals).item() / len(bitarray)


def ratio_variance(a, var_a, b, var_b):
    """observable_estimation.py:1052-1090."""
    return var_a / b ** 2 + (a ** 2 * var_b) / b ** 4


def estimate_dfe(expectations, std_errs, n_qubits, kind):
    """direct_fidelity_estimation.py:286-307 on plain arrays: (mean fidelity, standard error).

    state:   F = 1/d + (1 - 1/d) mean(e)
    process: average gate fidelity from the Choi-state fidelity p = 1/d^2 + (1 - 1/d^2) mean(e),
             F = (d^2 p + d) / (d^2 + d)
    The error bar is the root of the summed squared standard errors over m, times dF/dmean(e)."""
    d = 2.0 ** n_qubits
    e = np.asarray(expectations, dtype=float)
    s = np.asarray(std_errs, dtype=float)
    rms = np.sqrt(np.sum(s * s)) / e.size
    which = kind.lower()
    if which == "state":
        slope = 1.0 - 1.0 / d
        return 1.0 / d + slope * e.mean(), slope * rms
    if which == "process":
        slope = 1.0 - 1.0 / d ** 2
        p = 1.0 / d ** 2 + slope * e.mean()
        return (d * d * p + d) / (d * d + d), d / (d + 1.0) * slope * rms
    raise ValueError("Kind can only be 'state' or 'process'.")


# --------------------------------------------------------------------------------------------------
# Counter-based bootstrap resampling.  The reference draws e' = 2 Beta(n+ + prior, n- + prior) - 1
# from numpy's global stream (tomography.py:378-409); the device version cannot share that stream, so
# it defines its own generator.  This is the CPU statement of exactly that generator (checker only):
# Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11; known-answer
# vectors of the Random123 distribution are in tests/test_resample_cpu.py), counter = (element index
# low, high, draw number, 0), key = (seed low, high); per attempt one block -> two 32-bit uniforms for
# a Box-Muller normal and one 53-bit uniform for the Marsaglia-Tsang acceptance test.
# --------------------------------------------------------------------------------------------------
_M0, _M1, _W0, _W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
GAMMA_MAX_TRIES = 64


def philox4x32_10(counter, key):
    """counter [..., 4] uint32, key [..., 2] uint32 -> [..., 4] uint32"""
    c = [np.asarray(counter[..., j], dtype=np.uint64) for j in range(4)]
    k0 = np.asarray(key[..., 0], dtype=np.uint64)
    k1 = np.asarray(key[..., 1], dtype=np.uint64)
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(_M0) * c[0]
        p1 = np.uint64(_M1) * c[2]
        hi0, lo0 = p0 >> np.uint64(32), p0 & mask
        hi1, lo1 = p1 >> np.uint64(32), p1 & mask
        c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
        k0 = (k0 + np.uint64(_W0)) & mask
        k1 = (k1 + np.uint64(_W1)) & mask
    return np.stack(c, axis=-1).astype(np.uint32)


def _draw(idx, draw, seed):
    ctr = np.stack([idx & 0xFFFFFFFF, idx >> 32, draw, np.zeros_like(idx)], axis=-1).astype(np.uint32)
    key = np.broadcast_to(np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint32), idx.shape + (2,))
    x = philox4x32_10(ctr, key).astype(np.uint64)
    u1 = (x[..., 0].astype(np.float64) + 0.5) * 2.0 ** -32
    u2 = (x[..., 1].astype(np.float64) + 0.5) * 2.0 ** -32
    u3 = ((((x[..., 2] >> np.uint64(5)) << np.uint64(26)) | (x[..., 3] >> np.uint64(6))).astype(np.float64) + 0.5) * 2.0 ** -53
    return u1, u2, u3


def _gamma(a, idx, draw, seed):
    """Marsaglia-Tsang for every element; `draw` (int64 array) is advanced in place."""
    a = a.copy()
    boost = np.ones_like(a)
    small = a < 1.0
    if small.any():
        _, _, u3 = _draw(idx[small], draw[small], seed)
        boost[small] = u3 ** (1.0 / a[small])
        a[small] += 1.0
        draw[small] += 1
    d = a - 1.0 / 3.0
    c = 1.0 / np.sqrt(9.0 * d)
    out = boost * d
    todo = np.ones(a.shape, dtype=bool)
    for _ in range(GAMMA_MAX_TRIES):
        if not todo.any():
            break
        w = np.nonzero(todo)[0]
        u1, u2, u3 = _draw(idx[w], draw[w], seed)
        draw[w] += 1
        x = np.sqrt(-2.0 * np.log(u1)) * np.cos(6.283185307179586476925 * u2)
        v = 1.0 + c[w] * x
        ok = v > 0.0
        v3 = np.where(ok, v, 1.0) ** 3
        acc = ok & (np.log(u3) < 0.5 * x * x + d[w] - d[w] * v3 + d[w] * np.log(v3))
        out[w[acc]] = boost[w[acc]] * d[w[acc]] * v3[acc]
        todo[w[acc]] = False
    return out


def beta_resample(expectations, total_counts, n_resamples, prior_counts=1.0, seed=0):
    """[n_resamples, *expectations.shape] resampled expectations of the counter-based generator."""
    e = np.asarray(expectations, dtype=np.float64)
    cnt = np.broadcast_to(np.asarray(total_counts, dtype=np.float64), e.shape)
    n = e.size
    idx = np.arange(n * n_resamples, dtype=np.int64)
    ef, cf = np.tile(e.ravel(), n_resamples), np.tile(cnt.ravel(), n_resamples)
    n_plus = ((ef + 1.0) / 2.0) * cf
    a, b = n_plus + prior_counts, (cf - n_plus) + prior_counts
    good = (a > 0) & (b > 0)
    out = np.full(idx.shape, np.nan)
    draw = np.zeros(good.sum(), dtype=np.int64)
    ga = _gamma(a[good], idx[good], draw, seed)
    gb = _gamma(b[good], idx[good], draw, seed)
    out[good] = 2.0 * (ga / (ga + gb)) - 1.0
    return out.reshape((n_resamples,) + e.shape)


# --------------------------------------------------------------------------------------------------
# Device random operators (csrc/fbx_random.hip): the same Philox stream -- counter = (item id low, high,
# element index, stream tag), key = seed -- and the reference's arithmetic after the normals
# (operator_tools/random_operators.py:21-157).
# --------------------------------------------------------------------------------------------------
def ginibre_entries(seed, item, n_elems, tag=0):
    """n_elems complex standard normals of item `item` (Box-Muller on the first two words of each block)."""
    elem = np.arange(n_elems, dtype=np.int64)
    ctr = np.stack([np.full_like(elem, item & 0xFFFFFFFF), np.full_like(elem, item >> 32), elem,
                    np.full_like(elem, tag)], axis=-1).astype(np.uint32)
    key = np.broadcast_to(np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint32), (n_elems, 2))
    x = philox4x32_10(ctr, key).astype(np.float64)
    u1, u2 = (x[:, 0] + 0.5) * 2.0 ** -32, (x[:, 1] + 0.5) * 2.0 ** -32
    mag = np.sqrt(-2.0 * np.log(u1))
    ang = 6.283185307179586476925 * u2
    return mag * np.cos(ang) + 1j * mag * np.sin(ang)


def ginibre_matrix(seed, item, dim, k, tag=0):
    return ginibre_entries(seed, item, dim * k, tag).reshape(dim, k)


def haar_unitary(seed, item, dim, tag=0):
    """random_operators.py:49-72 on the device's Ginibre matrix."""
    z = ginibre_matrix(seed, item, dim, dim, tag)
    q, r = np.linalg.qr(z)
    dr = np.diagonal(r)
    return q @ (np.diag(dr) / np.absolute(dr))


def ginibre_state(seed, item, dim, rank):
    a = ginibre_matrix(seed, item, dim, rank)
    m = a @ a.conj().T
    return m / np.trace(m)


def bures_state(seed, item, dim):
    a = ginibre_matrix(seed, item, dim, dim, 0)
    u = haar_unitary(seed, item, dim, 1)
    w = np.eye(dim) + u
    p = w @ (a @ a.conj().T) @ w.conj().T
    return p / np.trace(p)


def random_kraus(seed, item, dim, k):
    """K_j = G_j S^{-1/2}; kraus2choi of it is the reference's rand_map_with_BCSZ_dist construction
    (random_operators.py:149-157) for X with columns vec(G_j)."""
    g = ginibre_entries(seed, item, k * dim * dim).reshape(k, dim, dim)
    s = sum(x.conj().T @ x for x in g)
    w, v = np.linalg.eigh(s)
    s_inv_half = (v / np.sqrt(w)) @ v.conj().T
    return np.array([x @ s_inv_half for x in g])
