// EXPERIMENT of round 5 (scripts/micro/jacobi_bench.hip -DFBX_JACOBI_REGPIVOT includes this file): the review's "next-round
// pivots delivered through registers instead of an LDS round trip" for the single-wavefront 16 x 16 solver.
//
// In jacobi_eigh_wave (csrc/fbx_eigh.hpp) the dependent chain of a round is
//     seat writes (4 ds_write_b128) -> pivot read -> rotation (two v_rsq_f64) -> ds_bpermute of the row rotation -> 2 x 2 update.
// Here the pivot never goes through LDS.  After the tournament permutation the pivot of the new pair K is
//     a' = rotated diagonal of one old pair, d' = rotated diagonal of another, b' = ONE entry of ONE lane's updated block:
//     K = 0: lane (0,1) entry (0,1), a' = an_0, d' = dn_1        K = 1: lane (0,2) entry (1,1), a' = dn_0, d' = dn_2
//     2 <= K <= 6: lane (K-1,K+1) entry (0,1), a' = an_{K-1}, d' = dn_{K+1}        K = 7: lane (6,7) entry (0,0), a' = an_6, d' = an_7
// so that lane computes the NEXT rotation straight from its registers (a', d' are the closed-form rotated diagonals an / dn of
// the rotation records it already holds; the diagonal blocks written to LDS carry the same values, so registers and LDS
// agree) and every lane fetches its row and column rotation of the next round from the two "pivot lanes" with ds_bpermute:
//     2 x 2 update -> rotation -> ds_bpermute (c, s of row and column pair, an of the row pair, dn of the column pair, one
//     exception value for lanes (0,2) / (6,7): 9 doubles = 18 ds_bpermute_b32) -> 2 x 2 update.
// One LDS hop per round instead of two, at the price of 12 more ds_bpermute_b32 and ~8 selects.  The blocks themselves still
// travel through their LDS seats (those reads are off the chain).  NOT bit-identical to jacobi_eigh_wave (an / dn of the
// rotation formula instead of the diagonal the 2 x 2 update produces); same tolerance, same sweeps.
#pragma once
namespace fbx {
template <int N>
__device__ int jacobi_eigh_wave_regpivot(cplx* __restrict__ Ms, cplx* __restrict__ Vs, int lane, bool init_identity,
                                         double expect_n2 = -1.0, double tol2 = FBX_JACOBI_TOL2) {
    constexpr int NB = N / 2, LS = NB * NB, PS = sys_plane<N>();
    static_assert(N == 16 && LS == 64, "every lane of the wavefront owns one 2x2 block");
    const int I = lane / NB, J = lane % NB;
    const int me = lane;
    const bool diag = I == J;
    int wm[4], wv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int sa = jacobi_seat<N>(2 * I + (e >> 1)), sb = jacobi_seat<N>(2 * J + (e & 1));
        wm[e] = ((sa & 1) * 2 + (sb & 1)) * PS + (sa >> 1) * NB + (sb >> 1);
        wv[e] = ((e >> 1) * 2 + (sb & 1)) * PS + I * NB + (sb >> 1);
    }
    auto pivlane = [](int k) { return k == 0 ? 1 : (k == 1 ? 2 : (k == NB - 1 ? (NB - 2) * NB + NB - 1 : (k - 1) * NB + k + 1)); };
    const int src_col = pivlane(J), src_row = pivlane(I);
    const bool ex1 = lane == 2, ex7 = lane == (NB - 2) * NB + NB - 1;        // lanes (0,2) and (6,7)
    const int src_x = ex1 ? pivlane(0) : lane;                                // (6,7) takes an_7 from itself = pivlane(7)
    const bool put_dn = lane == pivlane(0);                                   // what a lane offers as the exception value
    const int dJ = J * NB + J;
    const int diag_lane = (lane & 63) - J + I;

    cplx v0p, v0q, v1p, v1q;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        cplx v;
        if (init_identity) { v.re = (2 * I + (e >> 1) == jacobi_seat<N>(2 * J + (e & 1))) ? 1.0 : 0.0; v.im = 0.0; }
        else v = Vs[wv[e]];
        if (e == 0) v0p = v; else if (e == 1) v0q = v; else if (e == 2) v1p = v; else v1q = v;
    }
    double pc = 1.0, psr = 0.0, psi = 0.0;
    int sweep = 0;
    double n2 = 0.0;
    // rotations of this lane's row and column pair for the coming round (+ the rotated diagonals the next pivot needs)
    double cI = 1.0, sIr = 0.0, sIi = 0.0, anI = 0.0, cJ = 1.0, sJr = 0.0, sJi = 0.0, dnJ = 0.0, xv = 0.0;
    bool primed = false;
    for (; sweep < FBX_JACOBI_MAX_SWEEPS; ++sweep) {
        {
            double o2 = 0.0, a_all = 0.0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const cplx v = Ms[e * PS + me];
                const double a2 = v.re * v.re + v.im * v.im;
                a_all += a2;
                if (!(diag && (e == 0 || e == 3))) o2 += a2;
            }
            o2 = uniform(wave_sum(o2));
            if (sweep == 0) n2 = uniform(wave_sum(a_all));
            if (sweep == 0 && expect_n2 >= 0.0 && !(fabs(n2 - expect_n2) <= FBX_BASIS_NORM_TOL * expect_n2)) return -1;
            if (!(o2 > tol2 * n2)) break;
        }
        if (!primed) {          // first round of the decomposition: the pivots come from LDS, as in jacobi_eigh_wave
            const double aJ = Ms[0 * PS + dJ].re, dJ_ = Ms[3 * PS + dJ].re;
            const cplx bJ = Ms[1 * PS + dJ];
            const JRot rJ = jacobi_rotation(aJ, dJ_, bJ.re, bJ.im);
            cJ = rJ.c; sJr = rJ.sr; sJi = rJ.si; dnJ = rJ.dn;
            cI = __shfl(rJ.c, diag_lane); sIr = __shfl(rJ.sr, diag_lane); sIi = __shfl(rJ.si, diag_lane); anI = __shfl(rJ.an, diag_lane);
            const double dn0 = __shfl(rJ.dn, 0);                 // lane (0,0): dn of pair 0, for lane (0,2)
            xv = ex1 ? dn0 : rJ.an;                              // (6,7): an_7 is its own column rotation's
            primed = true;
        }
        for (int r = 0; r < N - 1; ++r) {
            cplx m00 = Ms[0 * PS + me], m01 = Ms[1 * PS + me];
            cplx m10 = Ms[2 * PS + me], m11 = Ms[3 * PS + me];
            jacobi_apply_v(pc, psr, psi, v0p, v0q, v1p, v1q);
            Vs[wv[0]] = v0p; Vs[wv[1]] = v0q; Vs[wv[2]] = v1p; Vs[wv[3]] = v1q;
            v0p = Vs[0 * PS + me]; v0q = Vs[1 * PS + me]; v1p = Vs[2 * PS + me]; v1q = Vs[3 * PS + me];
            jacobi_apply_m(cI, sIr, sIi, cJ, sJr, sJi, m00, m01, m10, m11);
            if (diag) {         // the annihilated pair: exact zeros, closed-form rotated diagonal (anI == an_J on a diagonal lane)
                m01.re = m01.im = 0.0; m10.re = m10.im = 0.0;
                m00.re = anI; m00.im = 0.0; m11.re = dnJ; m11.im = 0.0;
            }
            // the next pivot, on the eight pivot lanes (anything elsewhere: never fetched)
            const double pa = ex1 ? xv : anI, pd = ex7 ? xv : dnJ;
            const double pbr = ex1 ? m11.re : (ex7 ? m00.re : m01.re), pbi = ex1 ? m11.im : (ex7 ? m00.im : m01.im);
            const JRot nr = jacobi_rotation(pa, pd, pbr, pbi);
            Ms[wm[0]] = m00; Ms[wm[1]] = m01; Ms[wm[2]] = m10; Ms[wm[3]] = m11;
            pc = cJ; psr = sJr; psi = sJi;
            cJ = __shfl(nr.c, src_col); sJr = __shfl(nr.sr, src_col); sJi = __shfl(nr.si, src_col); dnJ = __shfl(nr.dn, src_col);
            cI = __shfl(nr.c, src_row); sIr = __shfl(nr.sr, src_row); sIi = __shfl(nr.si, src_row); anI = __shfl(nr.an, src_row);
            xv = __shfl(put_dn ? nr.dn : nr.an, src_x);
        }
    }
    jacobi_apply_v(pc, psr, psi, v0p, v0q, v1p, v1q);
    Vs[wv[0]] = v0p; Vs[wv[1]] = v0q; Vs[wv[2]] = v1p; Vs[wv[3]] = v1q;
    FBX_WAVE_SYNC();
    return sweep;
}
}  // namespace fbx
