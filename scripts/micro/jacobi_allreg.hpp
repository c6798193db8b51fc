// EXPERIMENT of round 5 (scripts/micro/jacobi_bench.hip -DFBX_JACOBI_ALLREG includes this file): the single-wavefront 16 x 16 solver
// with BOTH blocks in registers for the whole decomposition.  On top of jacobi_vdpp.hpp (eigenvector columns through DPP): the
// matrix block's columns move through the same DPP row shifts, its rows -- lane +- 8, which crosses DPP rows -- through
// ds_bpermute_b32 (16 per round: the LDS crossbar, no bank write), and the pivots never exist in LDS: after the exchange the
// diagonal lane (K, K) holds the complete pivot block of pair K, computes its rotation, and every lane fetches the rotation of its
// column pair from lane (J, J) and of its row pair from lane (I, I) (12 ds_bpermute_b32).  LDS per round: 28 ds_bpermute_b32
// instead of 4 ds_write_b128 + 7 reads + 6 ds_bpermute_b32.  Same rotations on the same data: bit-identical.
#pragma once
namespace fbx {
template <int N>
__device__ int jacobi_eigh_wave_allreg(cplx* __restrict__ Ms, cplx* __restrict__ Vs, int lane, bool init_identity,
                                       double expect_n2 = -1.0, double tol2 = FBX_JACOBI_TOL2) {
    constexpr int NB = N / 2, LS = NB * NB, PS = sys_plane<N>();
    static_assert(N == 16 && LS == 64, "every lane of the wavefront owns one 2x2 block; a block row is half a DPP row");
    lane = FBX_LOCAL(lane);
    const int I = lane / NB, J = lane % NB;
    const int me = lane;
    const bool firstJ = J == 0, lastJ = J == NB - 1, firstI = I == 0, lastI = I == NB - 1, diag = I == J;
    const int up = (lane + 64 - NB) & 63, down = (lane + NB) & 63;        // lanes (I - 1, J) and (I + 1, J)
    const int col_src = J * NB + J, row_src = I * NB + I;
    cplx m00 = Ms[0 * PS + me], m01 = Ms[1 * PS + me], m10 = Ms[2 * PS + me], m11 = Ms[3 * PS + me];
    cplx v0p, v0q, v1p, v1q;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        cplx v;
        if (init_identity) { v.re = (2 * I + (e >> 1) == 2 * J + (e & 1)) ? 1.0 : 0.0; v.im = 0.0; }
        else v = Vs[e * PS + me];
        if (e == 0) v0p = v; else if (e == 1) v0q = v; else if (e == 2) v1p = v; else v1q = v;
    }
    // the tournament permutation along a block row (columns): slot t_J -> t_{J+1} (t_0 stays, t_last -> b_last), b_J -> b_{J-1} (b_0 -> t_1)
    auto colperm = [&](double& p, double& q) __attribute__((always_inline)) {
        const double right = firstJ ? q : p;
        const double from_left = dpp_shift<0x111>(right);          // row_shr:1
        const double from_right = dpp_shift<0x101>(q);             // row_shl:1
        const double pn = firstJ ? p : from_left;
        const double qn = lastJ ? p : from_right;
        p = pn; q = qn;
    };
    // the same along a block column (rows): lanes (I -+ 1, J) are 8 lanes away, across DPP rows -> ds_bpermute
    auto rowperm = [&](double& p, double& q) __attribute__((always_inline)) {
        const double below = firstI ? q : p;                       // what this lane hands to the block row below it
        const double from_up = __shfl(below, up);
        const double from_down = __shfl(q, down);
        const double pn = firstI ? p : from_up;
        const double qn = lastI ? p : from_down;
        p = pn; q = qn;
    };
    double pc = 1.0, psr = 0.0, psi = 0.0;
    bool pending = false;
    int sweep = 0;
    double n2 = 0.0;
    for (; sweep < FBX_JACOBI_MAX_SWEEPS; ++sweep) {
        {
            const double q00 = m00.re * m00.re + m00.im * m00.im, q01 = m01.re * m01.re + m01.im * m01.im;
            const double q10 = m10.re * m10.re + m10.im * m10.im, q11 = m11.re * m11.re + m11.im * m11.im;
            double a_all = 0.0, o2 = 0.0;
            a_all += q00; a_all += q01; a_all += q10; a_all += q11;          // (the order of jacobi_eigh_wave's loop)
            if (!diag) o2 += q00;
            o2 += q01; o2 += q10;
            if (!diag) o2 += q11;
            o2 = uniform(wave_sum(o2));
            if (sweep == 0) n2 = uniform(wave_sum(a_all));
            if (sweep == 0 && expect_n2 >= 0.0 && !(fabs(n2 - expect_n2) <= FBX_BASIS_NORM_TOL * expect_n2)) return -1;
            if (!(o2 > tol2 * n2)) break;
        }
        for (int r = 0; r < N - 1; ++r) {
            // every lane rotates "its own block as a pivot"; only the diagonal lanes' result is fetched
            const JRot rot = jacobi_rotation(m00.re, m11.re, m01.re, m01.im);
            const double cJ = __shfl(rot.c, col_src), sJr = __shfl(rot.sr, col_src), sJi = __shfl(rot.si, col_src);
            const double cI = __shfl(rot.c, row_src), sIr = __shfl(rot.sr, row_src), sIi = __shfl(rot.si, row_src);
            if (pending) {
                jacobi_apply_v(pc, psr, psi, v0p, v0q, v1p, v1q);
                colperm(v0p.re, v0q.re); colperm(v0p.im, v0q.im); colperm(v1p.re, v1q.re); colperm(v1p.im, v1q.im);
            }
            jacobi_apply_m(cI, sIr, sIi, cJ, sJr, sJi, m00, m01, m10, m11);
            if (diag) {
                m01.re = m01.im = 0.0; m10.re = m10.im = 0.0;
                m00.im = 0.0; m11.im = 0.0;
            }
            // seats: columns, then rows
            colperm(m00.re, m01.re); colperm(m00.im, m01.im); colperm(m10.re, m11.re); colperm(m10.im, m11.im);
            rowperm(m00.re, m10.re); rowperm(m00.im, m10.im); rowperm(m01.re, m11.re); rowperm(m01.im, m11.im);
            pc = cJ; psr = sJr; psi = sJi; pending = true;
        }
    }
    if (pending) {
        jacobi_apply_v(pc, psr, psi, v0p, v0q, v1p, v1q);
        colperm(v0p.re, v0q.re); colperm(v0p.im, v0q.im); colperm(v1p.re, v1q.re); colperm(v1p.im, v1q.im);
    }
    Ms[0 * PS + me] = m00; Ms[1 * PS + me] = m01; Ms[2 * PS + me] = m10; Ms[3 * PS + me] = m11;
    Vs[0 * PS + me] = v0p; Vs[1 * PS + me] = v0q; Vs[2 * PS + me] = v1p; Vs[3 * PS + me] = v1q;
    FBX_WAVE_SYNC();
    return sweep;
}
}  // namespace fbx
