// fbx_pgdb1_core.hpp -- single-qubit PGDB process tomography, ONE LANE per reconstruction.
//
// A single-qubit Choi matrix is 4 x 4: the wave-per-item layout of fbx_pgdb.hip uses 4 of a wavefront's 64
// lanes for it.  Here every lane owns a whole reconstruction -- 64 per wavefront -- and everything of it lives
// in that lane's registers: Hermitian matrices as 4 real diagonal + 6 complex upper entries (16 doubles),
// the 4 x 4 eigenvectors, the Pauli-Liouville coefficients.  There is no cross-lane traffic at all; the
// design (settings grouped by input state, Bloch vectors) is shared by the batch, so every index into it is
// wave-uniform and its loads are scalar.  The only per-lane table is the normalised counts n+- (LDS, one
// column per lane, conflict-free).
//
// Reference functions restated (file:line under forest/benchmarking/):
//   pgdb_process_estimate            tomography.py:542-594
//   _extract_from_results / _cost / _grad_cost   tomography.py:494-539, 597-633 (the dense A is never formed:
//                                    p(s, P, +-) = (T[s][0] +- coef T[s][P]) / (2 d^2), T[s][i] = sum_j R_ij c_j(s))
//   proj_choi_to_physical (Dykstra), _completely_positive, _trace_preserving, _trace_non_increasing
//                                    operator_tools/project_superoperators.py:19-144
//
// The routines are plain per-thread C++ (static indices only, so that every array is a register file): the
// same source is compiled for the host by tests/host_harness (test infrastructure, never loaded by the
// package) to check the algebra against the oracle without a GPU.
#pragma once
#include <cstdint>
#include <cmath>

// profiling hooks (defined by fbx_pgdb1.hip in -DFBX_PHASE_TIMERS builds only: build.py --profile): wall cycles between marks, wave-level trip counts
#ifndef P1_PROF_BEGIN
#define P1_PROF_BEGIN
#define P1_PROF(k)
#define P1_COUNT(k)
#define P1_COUNT_LANES(k, n)
#endif

#ifdef FBX_PGDB1_HOST
#define FBX_P1 inline
namespace fbx {
inline double p1_log(double x) { return std::log(x); }
inline double p1_rsqrt(double x) { return 1.0 / std::sqrt(x); }
inline double p1_rcp(double x) { return 1.0 / x; }
#else
#define FBX_P1 __device__ __forceinline__
namespace fbx {
__device__ __forceinline__ double p1_log(double x) { return fast_log_pos(x); }
__device__ __forceinline__ double p1_rsqrt(double x) { return fast_rsqrt(x); }
__device__ __forceinline__ double p1_rcp(double x) {          // 1 / x for normal positive x, two Newton steps on v_rcp_f64
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    return fma(fma(-x, r, 1.0), r, r);
}
#endif

constexpr double P1_EPS = 1e-6;          // probability clip, tomography.py:597,613,631
constexpr double P1_GAMMA = 0.3;         // tomography.py:567
constexpr double P1_STOP = 1e-10;        // tomography.py:589
constexpr double P1_ALPHA_MIN = 1e-15;   // tomography.py:584
constexpr double P1_JTOL2 = 1e-26;       // off-norm^2 / norm^2 of the eigensolver (the library's FBX_JACOBI_TOL2)
constexpr int P1_MAX_SWEEPS = 40;
constexpr int P1_MAX_DYKSTRA = 100000;   // the reference has no cap; never binding (a NaN ends the loop by itself)

// ---- Hermitian 4 x 4: d[r] = A[r][r], (re, im)[u(r, c)] = A[r][c] for r < c
struct H4 { double d[4]; double re[6]; double im[6]; };
struct V4 { double re[4][4]; double im[4][4]; };

FBX_P1 constexpr int h4u(int r, int c) { return r * (7 - r) / 2 + (c - r - 1); }

template <int R, int C>
FBX_P1 void h4_get(const H4& A, double& xr, double& xi) {
    if constexpr (R == C) { xr = A.d[R]; xi = 0.0; }
    else if constexpr (R < C) { xr = A.re[h4u(R, C)]; xi = A.im[h4u(R, C)]; }
    else { xr = A.re[h4u(C, R)]; xi = -A.im[h4u(C, R)]; }
}
template <int R, int C>
FBX_P1 void h4_set(H4& A, double xr, double xi) {
    static_assert(R != C, "off-diagonal entries only");
    if constexpr (R < C) { A.re[h4u(R, C)] = xr; A.im[h4u(R, C)] = xi; }
    else { A.re[h4u(C, R)] = xr; A.im[h4u(C, R)] = -xi; }
}
FBX_P1 H4 h4_zero() {
    H4 r;
#pragma unroll
    for (int k = 0; k < 4; ++k) r.d[k] = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) { r.re[k] = 0.0; r.im[k] = 0.0; }
    return r;
}
FBX_P1 H4 h4_axpy(const H4& a, double s, const H4& b) {       // a + s b
    H4 r;
#pragma unroll
    for (int k = 0; k < 4; ++k) r.d[k] = fma(s, b.d[k], a.d[k]);
#pragma unroll
    for (int k = 0; k < 6; ++k) { r.re[k] = fma(s, b.re[k], a.re[k]); r.im[k] = fma(s, b.im[k], a.im[k]); }
    return r;
}
FBX_P1 H4 h4_sub(const H4& a, const H4& b) {
    H4 r;
#pragma unroll
    for (int k = 0; k < 4; ++k) r.d[k] = a.d[k] - b.d[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) { r.re[k] = a.re[k] - b.re[k]; r.im[k] = a.im[k] - b.im[k]; }
    return r;
}
FBX_P1 H4 h4_add(const H4& a, const H4& b) {
    H4 r;
#pragma unroll
    for (int k = 0; k < 4; ++k) r.d[k] = a.d[k] + b.d[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) { r.re[k] = a.re[k] + b.re[k]; r.im[k] = a.im[k] + b.im[k]; }
    return r;
}
// <a, b> = tr(a^H b): real for Hermitian operands
FBX_P1 double h4_dot(const H4& a, const H4& b) {
    double dg = 0.0, off = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) dg = fma(a.d[k], b.d[k], dg);
#pragma unroll
    for (int k = 0; k < 6; ++k) off = fma(a.re[k], b.re[k], fma(a.im[k], b.im[k], off));
    return fma(2.0, off, dg);
}
FBX_P1 double h4_norm2(const H4& a) { return h4_dot(a, a); }

// ---- 4 x 4 Hermitian eigendecomposition: cyclic Jacobi in the parallel order (0,1)(2,3) (0,2)(1,3) (0,3)(1,2) --
// the two rotations of a round touch disjoint pivots, so their reciprocal-square-root chains interleave.
// U = [[c, sigma], [-conj(sigma), c]] on the (p, q) plane with  delta = (a_qq - a_pp) / 2,  r = sqrt(delta^2 + |a_pq|^2),
// w = 1 / (|delta| + r),  c = 1 / sqrt(1 + |a_pq|^2 w^2),  sigma = sgn(delta) c w a_pq:  no division by |a_pq|, no
// branch except for an exactly vanishing pivot.
template <int P, int Q, int K>
FBX_P1 void p1_rot_offdiag(H4& A, double c, double sr, double si) {
    double xr, xi, yr, yi;
    h4_get<K, P>(A, xr, xi); h4_get<K, Q>(A, yr, yi);
    const double nxr = fma(c, xr, -fma(sr, yr, si * yi));
    const double nxi = fma(c, xi, -fma(sr, yi, -(si * yr)));
    const double nyr = fma(c, yr, fma(sr, xr, -(si * xi)));
    const double nyi = fma(c, yi, fma(sr, xi, si * xr));
    h4_set<K, P>(A, nxr, nxi); h4_set<K, Q>(A, nyr, nyi);
}
template <int P, int Q>
FBX_P1 void p1_rotate(H4& A, V4& V) {
    static_assert(P < Q, "pivot above the diagonal");
    const double ar = A.re[h4u(P, Q)], ai = A.im[h4u(P, Q)];
    const double g2 = fma(ar, ar, ai * ai);
    if (!(g2 > 0.0)) return;                           // (also leaves a NaN pivot alone)
    const double delta = 0.5 * (A.d[Q] - A.d[P]);
    const double r2 = fma(delta, delta, g2);
    const double r = r2 * p1_rsqrt(r2);
    const double w = p1_rcp(fabs(delta) + r);
    const double c = p1_rsqrt(fma(g2 * w, w, 1.0));
    const double k = copysign(c * w, delta);
    const double sr = k * ar, si = k * ai;
    const double tg = copysign(g2 * w, delta);         // t |a_pq|
    A.d[P] -= tg; A.d[Q] += tg;
    A.re[h4u(P, Q)] = 0.0; A.im[h4u(P, Q)] = 0.0;
    constexpr int K1 = (P != 0 && Q != 0) ? 0 : ((P != 1 && Q != 1) ? 1 : 2);
    constexpr int K2 = 6 - P - Q - K1;
    p1_rot_offdiag<P, Q, K1>(A, c, sr, si);
    p1_rot_offdiag<P, Q, K2>(A, c, sr, si);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {                   // V <- V U
        const double xr = V.re[kk][P], xi = V.im[kk][P], yr = V.re[kk][Q], yi = V.im[kk][Q];
        V.re[kk][P] = fma(c, xr, -fma(sr, yr, si * yi));
        V.im[kk][P] = fma(c, xi, -fma(sr, yi, -(si * yr)));
        V.re[kk][Q] = fma(c, yr, fma(sr, xr, -(si * xi)));
        V.im[kk][Q] = fma(c, yi, fma(sr, xi, si * xr));
    }
}
// A is destroyed (its diagonal ends up as the eigenvalues); V = eigenvectors (columns).  Returns the sweeps.
// `warm`: V holds a unitary that nearly diagonalises A (the basis of the previous decomposition -- consecutive Dykstra
// iterates and consecutive outer iterations project nearby matrices): A <- V^H A V first, then the sweeps continue
// from there (1-2 instead of 4-6).  The sweeps run to the same off-norm either way; a basis is only a starting guess.
FBX_P1 void p1_into_basis(H4& A, const V4& V) {
    double tr[4][4], ti[4][4];                              // T = A V
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            double sr = 0.0, si = 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                double ar, ai;
                if (r == k) { ar = A.d[r]; ai = 0.0; }
                else if (r < k) { ar = A.re[h4u(r, k)]; ai = A.im[h4u(r, k)]; }
                else { ar = A.re[h4u(k, r)]; ai = -A.im[h4u(k, r)]; }
                sr = fma(ar, V.re[k][c], fma(-ai, V.im[k][c], sr));
                si = fma(ar, V.im[k][c], fma(ai, V.re[k][c], si));
            }
            tr[r][c] = sr; ti[r][c] = si;
        }
#pragma unroll
    for (int r = 0; r < 4; ++r) {                           // (V^H T)[r][c] = sum_k conj(V[k][r]) T[k][c], upper triangle
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) s = fma(V.re[k][r], tr[k][r], fma(V.im[k][r], ti[k][r], s));
        A.d[r] = s;
#pragma unroll
        for (int c = r + 1; c < 4; ++c) {
            double sr = 0.0, si = 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                sr = fma(V.re[k][r], tr[k][c], fma(V.im[k][r], ti[k][c], sr));
                si = fma(V.re[k][r], ti[k][c], fma(-V.im[k][r], tr[k][c], si));
            }
            A.re[h4u(r, c)] = sr; A.im[h4u(r, c)] = si;
        }
    }
}
FBX_P1 int p1_eigh(H4& A, V4& V, bool warm = false) {
    if (warm) p1_into_basis(A, V);
    else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) { V.re[r][c] = r == c ? 1.0 : 0.0; V.im[r][c] = 0.0; }
    }
    int sweeps = 0;
    for (; sweeps < P1_MAX_SWEEPS; ++sweeps) {
        P1_COUNT(9); P1_COUNT_LANES(10, 1);
        double off = 0.0, dg = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) off = fma(A.re[k], A.re[k], fma(A.im[k], A.im[k], off));
#pragma unroll
        for (int k = 0; k < 4; ++k) dg = fma(A.d[k], A.d[k], dg);
        if (!(2.0 * off > P1_JTOL2 * fma(2.0, off, dg))) break;
        p1_rotate<0, 1>(A, V); p1_rotate<2, 3>(A, V);
        p1_rotate<0, 2>(A, V); p1_rotate<1, 3>(A, V);
        p1_rotate<0, 3>(A, V); p1_rotate<1, 2>(A, V);
    }
    return sweeps;
}

// The basis a lane carries from one decomposition to the next.  Every rotation costs ~1e-16 of unitarity: the chain is
// dropped (cold start from the identity) once it has absorbed P1_CHAIN_SWEEPS sweeps (the rule of the 2-qubit kernel's
// stored bases, FBX_BASIS_CHAIN_SWEEPS in fbx_pgdb.hip).
constexpr int P1_CHAIN_SWEEPS = 200;
struct P1Basis { V4 V; int chain; bool valid; };

// ---- CP projection (project_superoperators.py:19-34): V diag(max(lambda, 0)) V^H of the (Hermitian) argument
FBX_P1 H4 p1_proj_cp(const H4& x, int& sweeps, int& terms, P1Basis& basis) {
    H4 A = x;
    V4& V = basis.V;
    if (basis.chain >= P1_CHAIN_SWEEPS) { basis.valid = false; basis.chain = 0; }
    const int sw = p1_eigh(A, V, basis.valid);
    basis.valid = true; basis.chain += sw;
    sweeps += sw;
    double lam[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { lam[k] = A.d[k] < 0.0 ? 0.0 : A.d[k]; terms += A.d[k] > 0.0 ? 1 : 0; }
    H4 out;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) s = fma(lam[k], fma(V.re[r][k], V.re[r][k], V.im[r][k] * V.im[r][k]), s);
        out.d[r] = s;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = r + 1; c < 4; ++c) {
            double sr = 0.0, si = 0.0;                 // sum_k lam_k V[r][k] conj(V[c][k])
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double pr = fma(V.re[r][k], V.re[c][k], V.im[r][k] * V.im[c][k]);
                const double pi = fma(V.im[r][k], V.re[c][k], -(V.re[r][k] * V.im[c][k]));
                sr = fma(lam[k], pr, sr); si = fma(lam[k], pi, si);
            }
            out.re[h4u(r, c)] = sr; out.im[h4u(r, c)] = si;
        }
    return out;
}

// ---- the correction a TP / TNI projection subtracts: kron(corr / d, I_d) with a Hermitian 2 x 2 `corr`
// (index = 2 * input + output: entries (0,0) (1,1) <- c00, (2,2) (3,3) <- c11, (0,2) (1,3) <- c01)
struct Corr2 { double c00, c11, c01r, c01i; };
FBX_P1 H4 p1_tp_change(const Corr2& c) {             // -kron(corr / d, I_d)
    H4 r = h4_zero();
    r.d[0] = r.d[1] = -(0.5 * c.c00); r.d[2] = r.d[3] = -(0.5 * c.c11);
    r.re[h4u(0, 2)] = r.re[h4u(1, 3)] = -(0.5 * c.c01r);
    r.im[h4u(0, 2)] = r.im[h4u(1, 3)] = -(0.5 * c.c01i);
    return r;
}
// partial trace over the output space (calculational.py:5-35, keep=[0], dims=[2, 2])
FBX_P1 Corr2 p1_partial_trace(const H4& x) {
    Corr2 p;
    p.c00 = x.d[0] + x.d[1]; p.c11 = x.d[2] + x.d[3];
    p.c01r = x.re[h4u(0, 2)] + x.re[h4u(1, 3)]; p.c01i = x.im[h4u(0, 2)] + x.im[h4u(1, 3)];
    return p;
}
// project_superoperators.py:62-84 (TP) and :37-59 (TNI: eigenvalues of the partial trace above 1 clamped to 1)
FBX_P1 Corr2 p1_tp_correction(const H4& x, bool trace_preserving) {
    Corr2 pt = p1_partial_trace(x);
    if (trace_preserving) { pt.c00 -= 1.0; pt.c11 -= 1.0; return pt; }
    // 2 x 2 Hermitian eigendecomposition in closed form (the rotation of p1_rotate)
    const double g2 = fma(pt.c01r, pt.c01r, pt.c01i * pt.c01i);
    double l0 = pt.c00, l1 = pt.c11, c = 1.0, sr = 0.0, si = 0.0;
    if (g2 > 0.0) {
        const double delta = 0.5 * (pt.c11 - pt.c00);
        const double r2 = fma(delta, delta, g2);
        const double r = r2 * p1_rsqrt(r2);
        const double w = p1_rcp(fabs(delta) + r);
        c = p1_rsqrt(fma(g2 * w, w, 1.0));
        const double k = copysign(c * w, delta);
        sr = k * pt.c01r; si = k * pt.c01i;
        const double tg = copysign(g2 * w, delta);
        l0 -= tg; l1 += tg;
    }
    l0 = l0 > 1.0 ? 1.0 : l0; l1 = l1 > 1.0 ? 1.0 : l1;
    // projection = U diag(l) U^H,  U = [[c, sigma], [-conj(sigma), c]]
    const double s2 = fma(sr, sr, si * si);
    const double p00 = fma(l0 * c, c, l1 * s2), p11 = fma(l0, s2, l1 * c * c);
    const double p01r = (l1 - l0) * c * sr, p01i = (l1 - l0) * c * si;
    Corr2 out;
    out.c00 = pt.c00 - p00; out.c11 = pt.c11 - p11; out.c01r = pt.c01r - p01r; out.c01i = pt.c01i - p01i;
    return out;
}

// ---- Dykstra (project_superoperators.py:87-144), carried with two matrices: u = pre_CP and p = old_CP_change;
// old_TP_change is the 2 x 2 correction of the last TP / TNI projection, last_CP_projection only enters through the
// scalar <old_CP_change, last_CP_projection>, last_state = u + p (the form of proj_physical_blk_compact, fbx_choi.hpp)
FBX_P1 H4 p1_proj_physical(const H4& x, bool trace_preserving, int& iters, int& sweeps, int& terms, P1Basis& basis) {
    H4 u = x, p = h4_zero(), new_state = x;
    Corr2 qold; qold.c00 = qold.c11 = qold.c01r = qold.c01i = 0.0;
    double c0 = 0.0;
    for (int it = 0; it < P1_MAX_DYKSTRA; ++it) {
        ++iters;
        P1_COUNT(7); P1_COUNT_LANES(8, 1);
        const H4 cp = p1_proj_cp(u, sweeps, terms, basis);
        const H4 new_cp = h4_sub(cp, u);
        const double s1 = h4_norm2(h4_sub(new_cp, p));
        const double pc = h4_dot(p, cp), nc = h4_dot(new_cp, cp);
        const H4 last_state = h4_add(u, p);
        const H4 old_tp = p1_tp_change(qold);
        const H4 pre_tp = h4_sub(cp, old_tp);
        const Corr2 q = p1_tp_correction(pre_tp, trace_preserving);
        const H4 new_tp = p1_tp_change(q);
        new_state = h4_add(pre_tp, new_tp);
        const double s2 = h4_norm2(h4_sub(new_tp, old_tp));
        const double i1 = h4_dot(old_tp, h4_sub(new_state, last_state));
        const double i2 = pc - c0;
        const double crit = s1 + s2 + 2.0 * fabs(i1) + 2.0 * fabs(i2);
        if (!(crit >= 1e-4)) break;                   // converged -- or not finite: never spin
        c0 = nc; p = new_cp; qold = q;
        u = h4_sub(new_state, new_cp);
    }
    return new_state;
}

// ---- Choi <-> Pauli-Liouville coefficients, R_ij = (1/d) tr[(P_j^T (x) P_i) E]: radix-2 butterflies over the two tensor
// sites (input qubit = high bit of the row / column index, output qubit = low bit), as pauli_site_stage of fbx_choi.hpp
FBX_P1 void p1_choi_to_pauli(const H4& E, double (&R)[16]) {
    double mr[4][4], mi[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (r == c) { mr[r][c] = E.d[r]; mi[r][c] = 0.0; }
            else if (r < c) { mr[r][c] = E.re[h4u(r, c)]; mi[r][c] = E.im[h4u(r, c)]; }
            else { mr[r][c] = E.re[h4u(c, r)]; mi[r][c] = -E.im[h4u(c, r)]; }
        }
#pragma unroll
    for (int stage = 0; stage < 2; ++stage) {
        const int hb = stage == 0 ? 2 : 1, lb = stage == 0 ? 1 : 2;     // stride of the site's bit / of the other bit
        const double ys = stage == 0 ? -1.0 : 1.0;                      // -i for the input site (P_j^T), +i for the output site
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int r0 = a * lb, r1 = a * lb + hb, c0 = b * lb, c1 = b * lb + hb;
                const double c00r = mr[r0][c0], c00i = mi[r0][c0], c11r = mr[r1][c1], c11i = mi[r1][c1];
                const double c01r = mr[r0][c1], c01i = mi[r0][c1], c10r = mr[r1][c0], c10i = mi[r1][c0];
                mr[r0][c0] = c00r + c11r; mi[r0][c0] = c00i + c11i;       // I
                mr[r1][c1] = c00r - c11r; mi[r1][c1] = c00i - c11i;       // Z
                mr[r0][c1] = c01r + c10r; mi[r0][c1] = c01i + c10i;       // X
                const double dr = c01r - c10r, di = c01i - c10i;          // Y = +-i (c01 - c10)
                mr[r1][c0] = -ys * di; mi[r1][c0] = ys * dr;
            }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) R[i * 4 + j] = 0.5 * mr[2 * (j >> 1) + (i >> 1)][2 * (j & 1) + (i & 1)];
}
FBX_P1 H4 p1_pauli_to_choi(const double (&R)[16]) {
    double mr[4][4], mi[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { mr[2 * (j >> 1) + (i >> 1)][2 * (j & 1) + (i & 1)] = 2.0 * R[i * 4 + j]; mi[2 * (j >> 1) + (i >> 1)][2 * (j & 1) + (i & 1)] = 0.0; }
#pragma unroll
    for (int stage = 0; stage < 2; ++stage) {
        const int hb = stage == 0 ? 1 : 2, lb = stage == 0 ? 2 : 1;     // output site first, then the input site
        const double ys = stage == 0 ? 1.0 : -1.0;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int r0 = a * lb, r1 = a * lb + hb, c0 = b * lb, c1 = b * lb + hb;
                const double Ir = mr[r0][c0], Ii = mi[r0][c0], Zr = mr[r1][c1], Zi = mi[r1][c1];
                const double Xr = mr[r0][c1], Xi = mi[r0][c1], Yr = mr[r1][c0], Yi = mi[r1][c0];
                const double yr = -ys * Yi, yi = ys * Yr;                 // s i Y
                mr[r0][c0] = 0.5 * (Ir + Zr); mi[r0][c0] = 0.5 * (Ii + Zi);
                mr[r1][c1] = 0.5 * (Ir - Zr); mi[r1][c1] = 0.5 * (Ii - Zi);
                mr[r0][c1] = 0.5 * (Xr - yr); mi[r0][c1] = 0.5 * (Xi - yi);
                mr[r1][c0] = 0.5 * (Xr + yr); mi[r1][c0] = 0.5 * (Xi + yi);
            }
    }
    H4 out;
#pragma unroll
    for (int r = 0; r < 4; ++r) out.d[r] = mr[r][r];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = r + 1; c < 4; ++c) { out.re[h4u(r, c)] = mr[r][c]; out.im[h4u(r, c)] = mi[r][c]; }
    return out;
}

// ---- the likelihood.  `Des` has the members of DesignDev that are used here (m, S, unit_coefs, sp, coef, sptr, Ct);
// `NT` gives the lane's normalised counts of grouped setting g: nt.plus(g), nt.minus(g).
FBX_P1 double p1_pick(const double (&T)[4], int p) { return p == 1 ? T[1] : (p == 2 ? T[2] : (p == 3 ? T[3] : T[0])); }

template <class Des>
FBX_P1 void p1_state_row(const Des& des, int s, const double (&R)[16], double (&T)[4]) {
    const double* c = des.Ct + s * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) T[i] = fma(R[i * 4 + 3], c[3], fma(R[i * 4 + 2], c[2], fma(R[i * 4 + 1], c[1], R[i * 4] * c[0])));
}
// negative log-likelihood of the channel with Pauli-Liouville coefficients R (tomography.py:597-614)
template <class Des, class NT>
FBX_P1 double p1_cost(const Des& des, const NT& nt, const double (&R)[16]) {
    double acc = 0.0;
    for (int s = 0; s < des.S; ++s) {
        double T[4];
        p1_state_row(des, s, R, T);
        for (int g = des.sptr[s]; g < des.sptr[s + 1]; ++g) {
            const int p = (int)(des.sp[g] & 0xffffu);
            const double cf = des.unit_coefs ? 1.0 : des.coef[g];
            const double ex = cf * p1_pick(T, p);
            double pp = (T[0] + ex) * 0.125, pm = (T[0] - ex) * 0.125;     // 1 / (2 d^2)
            pp = pp < P1_EPS ? P1_EPS : pp; pm = pm < P1_EPS ? P1_EPS : pm;
            acc -= fma(nt.plus(g), p1_log(pp), nt.minus(g) * p1_log(pm));
        }
    }
    return acc;
}
// gradient (tomography.py:617-633) as a Hermitian matrix
template <class Des, class NT>
FBX_P1 H4 p1_gradient(const Des& des, const NT& nt, const double (&R)[16]) {
    double Rg[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) Rg[k] = 0.0;
    for (int s = 0; s < des.S; ++s) {
        double T[4];
        p1_state_row(des, s, R, T);
        double w0 = 0.0, w1 = 0.0, w2 = 0.0, w3 = 0.0;
        for (int g = des.sptr[s]; g < des.sptr[s + 1]; ++g) {
            const int p = (int)(des.sp[g] & 0xffffu);
            const double cf = des.unit_coefs ? 1.0 : des.coef[g];
            const double ex = cf * p1_pick(T, p);
            double pp = (T[0] + ex) * 0.125, pm = (T[0] - ex) * 0.125;
            pp = pp < P1_EPS ? P1_EPS : pp; pm = pm < P1_EPS ? P1_EPS : pm;
            const double ep = nt.plus(g) / pp, em = nt.minus(g) / pm;
            const double wsum = 0.5 * (ep + em), wdif = cf * 0.5 * (ep - em);
            w0 += wsum;
            w0 += p == 0 ? wdif : 0.0; w1 += p == 1 ? wdif : 0.0; w2 += p == 2 ? wdif : 0.0; w3 += p == 3 ? wdif : 0.0;
        }
        const double* c = des.Ct + s * 4;
        const double w[4] = {w0, w1, w2, w3};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) Rg[i * 4 + j] = fma(w[i], c[j], Rg[i * 4 + j]);
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) Rg[k] *= -0.25;        // -1 / d^2
    return p1_pauli_to_choi(Rg);
}

// ---- small steps of the line search.  cost(est + alpha upd) - cost(est) = -sum_o n_o log1p(alpha r_o), r_o = pu_o / pe_o:
// once alpha max|r_o| < 2^-3 the sum is the series sum_k c_k alpha^k S_k with the power sums S_k = sum_o n_o r_o^k (k <= 16:
// remainder < 2^-51 / 17 per unit of n, below the rounding of the exact evaluation), reduced ONCE per outer iteration --
// every further halving is a 16-term Horner step instead of 2 m logarithms, and the long halving runs of stalled iterations
// (up to 50) cost less than one evaluation.  Outcomes at the clip, or moving by more than their own size over a full step,
// are kept out of the sums and evaluated exactly (two register slots per lane; a lane with more of them stays with full
// evaluations).  The acceptance test is then made on the difference itself -- the rule of the 2-qubit kernel
// (fbx_pgdb.hip `rejected()`): the noise-free limit of the reference's `new_cost > old_cost + change`.
constexpr int P1_NS = 16;
constexpr double P1_SMALL_STEP = 0x1p-3;
struct P1Line {
    double S[P1_NS];
    double rmax;
    double fpe[2], fpu[2], fn[2], fbase[2];
    int nflag;
    bool ok;
};
template <class Des, class NT>
FBX_P1 void p1_line_prepare(const Des& des, const NT& nt, const double (&Re)[16], const double (&Ru)[16], P1Line& L) {
#pragma unroll
    for (int k = 0; k < P1_NS; ++k) L.S[k] = 0.0;
    L.rmax = 0.0; L.nflag = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) { L.fpe[i] = 1.0; L.fpu[i] = 0.0; L.fn[i] = 0.0; L.fbase[i] = 0.0; }
    for (int s = 0; s < des.S; ++s) {
        double Te[4], Tu[4];
        p1_state_row(des, s, Re, Te);
        p1_state_row(des, s, Ru, Tu);
        for (int g = des.sptr[s]; g < des.sptr[s + 1]; ++g) {
            const int p = (int)(des.sp[g] & 0xffffu);
            const double cf = des.unit_coefs ? 1.0 : des.coef[g];
            const double exe = cf * p1_pick(Te, p), exu = cf * p1_pick(Tu, p);
#pragma unroll
            for (int sg = 0; sg < 2; ++sg) {
                const double pe = (sg ? Te[0] - exe : Te[0] + exe) * 0.125, pu = (sg ? Tu[0] - exu : Tu[0] + exu) * 0.125;
                const double nn = sg ? nt.minus(g) : nt.plus(g);
                const bool flag = pe < 2.0 * P1_EPS || fabs(pu) > pe;
                double x = 0.0;
                if (flag) {
                    if (L.nflag == 0) { L.fpe[0] = pe; L.fpu[0] = pu; L.fn[0] = nn; }
                    else if (L.nflag == 1) { L.fpe[1] = pe; L.fpu[1] = pu; L.fn[1] = nn; }
                    ++L.nflag;
                } else x = pu * p1_rcp(pe);
                L.rmax = fmax(L.rmax, fabs(x));
                double t = nn * x;
#pragma unroll
                for (int k = 0; k < P1_NS; ++k) { L.S[k] += t; t *= x; }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < P1_NS; ++k) L.S[k] *= ((k & 1) ? -1.0 : 1.0) / (double)(k + 1);
#pragma unroll
    for (int i = 0; i < 2; ++i) L.fbase[i] = L.fn[i] * p1_log(L.fpe[i] < P1_EPS ? P1_EPS : L.fpe[i]);
    L.ok = (L.rmax == L.rmax) && L.nflag <= 2;
}
// sum_o n_o [log p_o(alpha) - log p_o(0)] = old_cost - new_cost for a step in the small regime
FBX_P1 double p1_line_small(const P1Line& L, double alpha) {
    double q = L.S[P1_NS - 1];
#pragma unroll
    for (int k = P1_NS - 2; k >= 0; --k) q = fma(alpha, q, L.S[k]);
    double acc = alpha * q;
    // the exact slots: empty ones (n = 0, log 1 - 0) add nothing -- and a stalled line search is 50 of these steps while the
    // other lanes of the wavefront wait, so a wavefront without any flagged outcome skips the two logarithms altogether
    if (L.nflag > 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const double pa = fma(alpha, L.fpu[i], L.fpe[i]);
            acc += L.fn[i] * p1_log(pa < P1_EPS ? P1_EPS : pa) - L.fbase[i];
        }
    }
    return acc;
}

// ---- one reconstruction's running state
struct P1State {
    H4 est;
    P1Basis basis;
    double old_cost, new_cost;
    double last_decrease;        // old_cost - new_cost of the last outer iteration (a scheduling hint, never part of the result)
    int iters, dyk, backtracks, sweeps, terms, ls_full, ls_sums;
};
template <class Des, class NT>
FBX_P1 void p1_begin(const Des& des, const NT& nt, P1State& st) {
    st.est = h4_zero();
#pragma unroll
    for (int k = 0; k < 4; ++k) st.est.d[k] = 0.5;      // I_D / d, tomography.py:564
    st.iters = st.dyk = st.backtracks = st.sweeps = st.terms = 0;
    st.basis.valid = false; st.basis.chain = 0;
    double R[16];
    p1_choi_to_pauli(st.est, R);
    st.old_cost = p1_cost(des, nt, R);                    // tomography.py:565
    st.new_cost = st.old_cost;
    st.last_decrease = 1.0;
    st.ls_full = 1; st.ls_sums = 0;
}
// One outer iteration (tomography.py:570-592).  Returns true when the reconstruction is finished.
template <class Des, class NT>
FBX_P1 bool p1_outer_iteration(const Des& des, const NT& nt, P1State& st, bool trace_preserving, int mode, int max_iters,
                               int& dyk_this, int& bt_this) {
    if (mode == 1 /* FBX_MODE_FIXED */ && st.iters >= max_iters) { dyk_this = 0; bt_this = 0; return true; }
    const int dyk_before = st.dyk, bt_before = st.backtracks;
    P1_PROF_BEGIN;
    P1_COUNT(5); P1_COUNT_LANES(6, 1);
    double Re[16];
    p1_choi_to_pauli(st.est, Re);
    const H4 grad = p1_gradient(des, nt, Re);
    const H4 x = h4_axpy(st.est, -(8.0 / 3.0), grad);          // est - gradient / mu, mu = 3 / (2 d^2)
    P1_PROF(0);
    const H4 proj = p1_proj_physical(x, trace_preserving, st.dyk, st.sweeps, st.terms, st.basis);
    P1_PROF(1);
    const H4 upd = h4_sub(proj, st.est);
    const double ipr = h4_dot(upd, grad);
    double Ru[16];
    p1_choi_to_pauli(upd, Ru);
    // backtracking line search (tomography.py:575-585); the coefficients are linear in the estimate
    double alpha = 1.0, change = P1_GAMMA * ipr;
    double new_cost;
    {
        double Ra[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) Ra[k] = Re[k] + Ru[k];
        new_cost = p1_cost(des, nt, Ra);
        ++st.ls_full;
    }
    P1_PROF(2);
    bool ls_exact = false, prepared = false;
    double ls_diff = 0.0;
    P1Line line;
    line.ok = false; line.rmax = 0.0;
    while (ls_exact ? (ls_diff > change) : (new_cost > st.old_cost + change)) {
        alpha *= 0.5; change *= 0.5;
        P1_COUNT(11); P1_COUNT_LANES(12, 1);
        if (!prepared) { P1_COUNT(13); p1_line_prepare(des, nt, Re, Ru, line); prepared = true; ++st.ls_sums; }
        if (line.ok && alpha * line.rmax < P1_SMALL_STEP) {
            const double acc = p1_line_small(line, alpha);
            ls_exact = true; ls_diff = -acc;
            new_cost = st.old_cost - acc;
        } else {
            P1_COUNT(14); P1_COUNT_LANES(15, 1);
            double Ra[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) Ra[k] = fma(alpha, Ru[k], Re[k]);
            new_cost = p1_cost(des, nt, Ra);
            ls_exact = false;
            ++st.ls_full;
        }
        ++st.backtracks;
        if (alpha < P1_ALPHA_MIN) break;
    }
    P1_PROF(3);
    st.est = h4_axpy(st.est, alpha, upd);                     // tomography.py:588
    st.new_cost = new_cost;
    ++st.iters;
    dyk_this = st.dyk - dyk_before; bt_this = st.backtracks - bt_before;
    bool done = false;
    if (mode == 0 /* FBX_MODE_CONVERGE */) {
        if (!(st.old_cost - new_cost >= P1_STOP)) done = true;        // tomography.py:589; a NaN cost also ends the loop
        if (max_iters > 0 && st.iters >= max_iters) done = true;
    } else if (st.iters >= max_iters) done = true;
    st.last_decrease = st.old_cost - new_cost;
    st.old_cost = new_cost;
    return done;
}

}  // namespace fbx
