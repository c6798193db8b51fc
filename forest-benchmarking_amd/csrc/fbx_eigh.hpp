// fbx_eigh.hpp -- wavefront-resident Hermitian eigensolver for the 4^n x 4^n (n <= 2) and
// 2^n x 2^n matrices of the tomography path: cyclic two-sided Jacobi with a round-robin
// (Brent-Luk style) parallel ordering.  The matrix lives in LDS; lane (I, J) of an
// (N/2) x (N/2) lane grid owns the 2x2 block of rows {p_I, q_I} x columns {p_J, q_J} of the
// current pairing, so one round applies N/2 disjoint rotations with every lane busy.
//
// Replaces scipy.linalg.eigh / numpy.linalg.eigh at
//   operator_tools/project_superoperators.py:30,52,165; project_state_matrix.py:27;
//   calculational.py:85; superoperator_transformations.py:334.
// Parity is on V f(Lambda) V^H (basis independent), never on eigenvectors.
#pragma once
#include "fbx_common.hpp"

namespace fbx {

// 2x2 complex block owned by a lane: element e = 2a + b is (row 2I+a, col 2J+b)
struct Blk { double re[4], im[4]; };

__device__ __forceinline__ Blk blk_zero() {
    Blk r;
#pragma unroll
    for (int e = 0; e < 4; ++e) { r.re[e] = 0.0; r.im[e] = 0.0; }
    return r;
}
__device__ __forceinline__ Blk blk_sub(const Blk& a, const Blk& b) {
    Blk r;
#pragma unroll
    for (int e = 0; e < 4; ++e) { r.re[e] = a.re[e] - b.re[e]; r.im[e] = a.im[e] - b.im[e]; }
    return r;
}
__device__ __forceinline__ Blk blk_axpy(const Blk& a, double s, const Blk& b) {   // a + s*b
    Blk r;
#pragma unroll
    for (int e = 0; e < 4; ++e) { r.re[e] = a.re[e] + s * b.re[e]; r.im[e] = a.im[e] + s * b.im[e]; }
    return r;
}
// sum_e |a_e|^2 (per lane partial)
__device__ __forceinline__ double blk_norm2(const Blk& a) {
    double s = 0.0;
#pragma unroll
    for (int e = 0; e < 4; ++e) s += a.re[e] * a.re[e] + a.im[e] * a.im[e];
    return s;
}
// sum_e conj(a_e) * b_e (per lane partial)
__device__ __forceinline__ void blk_dotc(const Blk& a, const Blk& b, double& re, double& im) {
    re = 0.0; im = 0.0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        re += a.re[e] * b.re[e] + a.im[e] * b.im[e];
        im += a.re[e] * b.im[e] - a.im[e] * b.re[e];
    }
}

template <int N, int LD>
__device__ __forceinline__ void blk_store(cplx* M, int lane, const Blk& v) {
    constexpr int NB = N / 2;
    if (lane < NB * NB) {
        const int I = lane / NB, J = lane % NB;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            cplx c; c.re = v.re[e]; c.im = v.im[e];
            M[(2 * I + (e >> 1)) * LD + 2 * J + (e & 1)] = c;
        }
    }
}
template <int N, int LD>
__device__ __forceinline__ Blk blk_load(const cplx* M, int lane) {
    constexpr int NB = N / 2;
    Blk v = blk_zero();
    if (lane < NB * NB) {
        const int I = lane / NB, J = lane % NB;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            cplx c = M[(2 * I + (e >> 1)) * LD + 2 * J + (e & 1)];
            v.re[e] = c.re; v.im[e] = c.im;
        }
    }
    return v;
}
// conjugate-transpose block of the matrix staged in M: element e -> conj(M[col][row])
template <int N, int LD>
__device__ __forceinline__ Blk blk_load_adjoint(const cplx* M, int lane) {
    constexpr int NB = N / 2;
    Blk v = blk_zero();
    if (lane < NB * NB) {
        const int I = lane / NB, J = lane % NB;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            cplx c = M[(2 * J + (e & 1)) * LD + 2 * I + (e >> 1)];
            v.re[e] = c.re; v.im[e] = -c.im;
        }
    }
    return v;
}

// round-robin (circle method) pairing of N players: round r in [0, N-1), pair k in [0, N/2)
template <int N>
__device__ __forceinline__ void rr_pair(int r, int k, int& p, int& q) {
    constexpr int M = N - 1;
    if (k == 0) { p = N - 1; q = r; }
    else {
        p = r + k; if (p >= M) p -= M;
        q = r - k; if (q < 0) q += M;
    }
}

constexpr int FBX_JACOBI_MAX_SWEEPS = 40;
constexpr double FBX_JACOBI_TOL2 = 1e-26;   // stop when off(A)^2 <= TOL2 * ||A||_F^2

// In-LDS Hermitian eigendecomposition.  On entry M holds the (Hermitian) matrix; on exit the
// diagonal of M holds the eigenvalues and the columns of V the eigenvectors.  `rot` is
// 4*(N/2) doubles of LDS scratch.  All 64 lanes of the wave must call; (N/2)^2 do the work.
// Returns the number of sweeps performed.
template <int N, int LD>
__device__ int jacobi_eigh_lds(cplx* M, cplx* V, double* rot, int lane) {
    constexpr int NB = N / 2, NACT = NB * NB;
    static_assert(NACT <= 64, "one wavefront per matrix");
    const bool act = lane < NACT;
    const int I = act ? lane / NB : 0, J = act ? lane % NB : 0;
    if (act) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int r = 2 * I + (e >> 1), c = 2 * J + (e & 1);
            cplx v; v.re = (r == c) ? 1.0 : 0.0; v.im = 0.0;
            V[r * LD + c] = v;
        }
    }
    __syncthreads();
    int sweep = 0;
    for (; sweep < FBX_JACOBI_MAX_SWEEPS; ++sweep) {
        double o2 = 0.0, n2 = 0.0;
        if (act) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 2 * I + (e >> 1), c = 2 * J + (e & 1);
                const cplx v = M[r * LD + c];
                const double a2 = v.re * v.re + v.im * v.im;
                n2 += a2;
                if (r != c) o2 += a2;
            }
        }
        o2 = uniform(wave_sum(o2));
        n2 = uniform(wave_sum(n2));
        if (!(o2 > FBX_JACOBI_TOL2 * n2)) break;
        for (int r = 0; r < N - 1; ++r) {
            if (act && I == J) {
                int p, q;
                rr_pair<N>(r, I, p, q);
                const double a = M[p * LD + p].re, dq = M[q * LD + q].re;
                const cplx b = M[p * LD + q];
                const double beta = b.re * b.re + b.im * b.im;
                double c = 1.0, sr = 0.0, si = 0.0;
                if (beta > 1e-300) {
                    const double delta = 0.5 * (dq - a);
                    const double w = fabs(delta) + sqrt(delta * delta + beta);
                    const double iw = 1.0 / w;
                    c = 1.0 / sqrt(1.0 + beta * iw * iw);
                    const double f = (delta >= 0.0 ? iw : -iw) * c;
                    sr = f * b.re; si = f * b.im;
                }
                rot[4 * I + 0] = c; rot[4 * I + 1] = sr; rot[4 * I + 2] = si;
            }
            __syncthreads();
            if (act) {
                int pI, qI, pJ, qJ;
                rr_pair<N>(r, I, pI, qI);
                rr_pair<N>(r, J, pJ, qJ);
                const double cI = rot[4 * I], sIr = rot[4 * I + 1], sIi = rot[4 * I + 2];
                const double cJ = rot[4 * J], sJr = rot[4 * J + 1], sJi = rot[4 * J + 2];
                cplx m00 = M[pI * LD + pJ], m01 = M[pI * LD + qJ];
                cplx m10 = M[qI * LD + pJ], m11 = M[qI * LD + qJ];
                // columns: u' = c u - conj(s) v ; v' = s u + c v
                cplx t00, t01, t10, t11;
                t00.re = cJ * m00.re - (sJr * m01.re + sJi * m01.im);
                t00.im = cJ * m00.im - (sJr * m01.im - sJi * m01.re);
                t01.re = cJ * m01.re + (sJr * m00.re - sJi * m00.im);
                t01.im = cJ * m01.im + (sJr * m00.im + sJi * m00.re);
                t10.re = cJ * m10.re - (sJr * m11.re + sJi * m11.im);
                t10.im = cJ * m10.im - (sJr * m11.im - sJi * m11.re);
                t11.re = cJ * m11.re + (sJr * m10.re - sJi * m10.im);
                t11.im = cJ * m11.im + (sJr * m10.im + sJi * m10.re);
                // rows: u' = c u - s v ; v' = conj(s) u + c v
                m00.re = cI * t00.re - (sIr * t10.re - sIi * t10.im);
                m00.im = cI * t00.im - (sIr * t10.im + sIi * t10.re);
                m10.re = cI * t10.re + (sIr * t00.re + sIi * t00.im);
                m10.im = cI * t10.im + (sIr * t00.im - sIi * t00.re);
                m01.re = cI * t01.re - (sIr * t11.re - sIi * t11.im);
                m01.im = cI * t01.im - (sIr * t11.im + sIi * t11.re);
                m11.re = cI * t11.re + (sIr * t01.re + sIi * t01.im);
                m11.im = cI * t11.im + (sIr * t01.im - sIi * t01.re);
                if (I == J) {   // the annihilated pair: exact zeros, real diagonal
                    m01.re = m01.im = 0.0; m10.re = m10.im = 0.0;
                    m00.im = 0.0; m11.im = 0.0;
                }
                M[pI * LD + pJ] = m00; M[pI * LD + qJ] = m01;
                M[qI * LD + pJ] = m10; M[qI * LD + qJ] = m11;
                // eigenvector accumulation: rows 2I, 2I+1 of V, columns pJ, qJ
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int row = 2 * I + a;
                    const cplx u = V[row * LD + pJ], v = V[row * LD + qJ];
                    cplx un, vn;
                    un.re = cJ * u.re - (sJr * v.re + sJi * v.im);
                    un.im = cJ * u.im - (sJr * v.im - sJi * v.re);
                    vn.re = cJ * v.re + (sJr * u.re - sJi * u.im);
                    vn.im = cJ * v.im + (sJr * u.im + sJi * u.re);
                    V[row * LD + pJ] = un; V[row * LD + qJ] = vn;
                }
            }
            __syncthreads();
        }
    }
    return sweep;
}

// block (I, J) of sum_k lam[k] v_k v_k^H for the eigenvectors in V; terms with lam[k] == 0
// are skipped (wave-uniform branch).
template <int N, int LD>
__device__ __forceinline__ Blk reconstruct_blk(const cplx* V, const double* lam, int lane) {
    constexpr int NB = N / 2;
    Blk out = blk_zero();
    const bool act = lane < NB * NB;
    const int I = act ? lane / NB : 0, J = act ? lane % NB : 0;
    for (int k = 0; k < N; ++k) {
        const double l = lam[k];
        if (l == 0.0) continue;
        const cplx r0 = V[(2 * I) * LD + k], r1 = V[(2 * I + 1) * LD + k];
        const cplx c0 = V[(2 * J) * LD + k], c1 = V[(2 * J + 1) * LD + k];
        const double w0r = l * r0.re, w0i = l * r0.im, w1r = l * r1.re, w1i = l * r1.im;
        // w * conj(c)
        out.re[0] += w0r * c0.re + w0i * c0.im; out.im[0] += w0i * c0.re - w0r * c0.im;
        out.re[1] += w0r * c1.re + w0i * c1.im; out.im[1] += w0i * c1.re - w0r * c1.im;
        out.re[2] += w1r * c0.re + w1i * c0.im; out.im[2] += w1i * c0.re - w1r * c0.im;
        out.re[3] += w1r * c1.re + w1i * c1.im; out.im[3] += w1i * c1.re - w1r * c1.im;
    }
    if (!act) out = blk_zero();
    return out;
}

}  // namespace fbx
