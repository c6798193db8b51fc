// Round-5 experiment (measured, NOT adopted): the published-rotation scheme of the 64 x 64 solver (csrc/fbx_eigh64.hpp) on the
// single-wavefront 16 x 16 solver.  jacobi_bench -DFBX_JACOBI_PUB16, one MI355X, cycles per round: 1215 against 1029 (the library's
// loop) at one wavefront per SIMD, 1706 against 1423 at two; max-ILP scheduling 1104 / 1203 against 960 / 1080.  With ONE wavefront
// per matrix nothing is saved -- the wavefront issues one rotation per round either way -- and the order  read -> apply ->
// evaluate -> write  puts the two reciprocal-square-root chains BEHIND the block update on the chain of the round, where the
// library's loop starts them as soon as the pivot block has arrived and overlaps them with the block loads.
// (Without the compiler barrier behind the record stores the next round's record loads were hoisted above them: the records
// travel between lanes, which the per-thread memory model does not see.)
#pragma once
#include "fbx_eigh.hpp"
namespace fbx {
// PUBLISHED ROTATIONS for the single-wavefront solver (round 5; the 64 x 64 solver's scheme, fbx_eigh64.hpp).  The form below this
// one evaluates the rotation of the lane's COLUMN pair from the pivot block and fetches the one of its ROW pair from the lane
// that evaluated it (three doubles through ds_bpermute) -- a second trip through the LDS pipe on the chain of every round.  Here
// the lane that holds next round's pivot entry at the end of this round -- block (0,1), (0,2), (K-1,K+1), (NB-2,NB-1) -- evaluates
// that rotation from its own updated entry and the rotated diagonals of the two records it has just applied, and writes the record
// {c, s, a', d'} with its block; everybody reads two records with its block.  One evaluation per lane as before (the wavefront
// issues the masked branch once), no ds_bpermute, no pivot loads.  Records: Vs entries 0..NB-1 of planes 0 / 1 / 2 (the
// eigenvectors are in registers during the solve; the final store overwrites the records).  The rotated diagonal is the closed
// form a + u, d - u of the rotation (as in fbx_eigh64.hpp and the two-worker form), not the result of applying the rotations to
// the pivot block: rounding-level different from the form below.
template <int N>
__device__ int jacobi_eigh_wave_pub16(cplx* __restrict__ Ms, cplx* __restrict__ Vs, int lane, bool init_identity,
                                double expect_n2 = -1.0, double tol2 = FBX_JACOBI_TOL2) {
    constexpr int NB = N / 2, LS = NB * NB, PS = sys_plane<N>();
    static_assert(N == 16 && LS == 64, "every lane of the wavefront owns one 2x2 block; a block row is half a DPP row");
    lane = FBX_LOCAL(lane);
    const int I = lane / NB, J = lane % NB;
    const int me = lane;
    int wm[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int sa = jacobi_seat<N>(2 * I + (e >> 1)), sb = jacobi_seat<N>(2 * J + (e & 1));
        wm[e] = ((sa & 1) * 2 + (sb & 1)) * PS + (sa >> 1) * NB + (sb >> 1);
    }
    const bool first = J == 0, last = J == NB - 1, diag = I == J;
    // which of next round's pairs this block holds the pivot entry of: the tournament permutation forms pair 0 = {top 0, bottom 1},
    // pair 1 = {bottom 0, bottom 2}, pair K = {top K-1, bottom K+1}, pair NB-1 = {top NB-2, top NB-1}
    const bool k_first = I == 0 && J == 1, k_second = I == 0 && J == 2, k_last = I == NB - 2 && J == NB - 1;
    const bool owner = k_first || k_second || k_last || (J == I + 2 && I >= 1);
    const int Kn = k_first ? 0 : k_second ? 1 : k_last ? NB - 1 : I + 1;
    cplx v0p, v0q, v1p, v1q;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        cplx v;
        if (init_identity) { v.re = (2 * I + (e >> 1) == 2 * J + (e & 1)) ? 1.0 : 0.0; v.im = 0.0; }
        else v = Vs[e * PS + me];
        if (e == 0) v0p = v; else if (e == 1) v0q = v; else if (e == 2) v1p = v; else v1q = v;
    }
    auto permute = [&](double& p, double& q) __attribute__((always_inline)) {
        const double right = first ? q : p;
        const double from_left = dpp_shift<0x111>(right);          // row_shr:1
        const double from_right = dpp_shift<0x101>(q);             // row_shl:1
        const double pn = first ? p : from_left;
        const double qn = last ? p : from_right;
        p = pn; q = qn;
    };
    auto publish = [&](int K, const JRot& n) __attribute__((always_inline)) {
        cplx e0, e1, e2;
        e0.re = n.c; e0.im = n.sr; e1.re = n.si; e1.im = n.an; e2.re = n.dn; e2.im = 0.0;
        Vs[K] = e0; Vs[PS + K] = e1; Vs[2 * PS + K] = e2;
    };
    double pc = 1.0, psr = 0.0, psi = 0.0;          // rotation whose eigenvector update is still pending
    bool pending = false;
    int sweep = 0;
    double n2 = 0.0;
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
        if (!pending && diag) {                     // the records of the very first round, from the pivots as they stand
            const JRot n = jacobi_rotation(Ms[0 * PS + me].re, Ms[3 * PS + me].re, Ms[1 * PS + me].re, Ms[1 * PS + me].im);
            publish(J, n);
        }
        FBX_WAVE_SYNC();
        for (int r = 0; r < N - 1; ++r) {
            const cplx i0 = Vs[I], i1 = Vs[PS + I];            // (c, Re s), (Im s, a') of the row pair
            const cplx j0 = Vs[J], j1 = Vs[PS + J];            // ... of the column pair
            cplx m00 = Ms[0 * PS + me], m01 = Ms[1 * PS + me];
            cplx m10 = Ms[2 * PS + me], m11 = Ms[3 * PS + me];
            double dnI = 0.0, dnJ = 0.0;
            dnI = Vs[2 * PS + I].re; dnJ = Vs[2 * PS + J].re;
            if (pending) {
                jacobi_apply_v(pc, psr, psi, v0p, v0q, v1p, v1q);
                permute(v0p.re, v0q.re); permute(v0p.im, v0q.im); permute(v1p.re, v1q.re); permute(v1p.im, v1q.im);
            }
            jacobi_apply_m(i0.re, i0.im, i1.re, j0.re, j0.im, j1.re, m00, m01, m10, m11);
            if (diag) {     // the annihilated pair: exact zeros, the rotated diagonal in closed form
                m01.re = m01.im = 0.0; m10.re = m10.im = 0.0;
                m00.re = j1.im; m00.im = 0.0; m11.re = dnJ; m11.im = 0.0;
            }
            {   // (evaluated by every lane, stored by the eight that hold a pivot entry: straight-line code that the scheduler can
                //  interleave with the eigenvector update above -- behind a branch the two reciprocal-square-root chains run alone)
                const double a = k_second ? dnI : i1.im;           // pair 1 takes the BOTTOM of pair 0
                const double d = k_last ? j1.im : dnJ;             // the last pair takes the TOP of the last pair
                const cplx b = k_second ? m11 : k_last ? m00 : m01;
                const JRot nn = jacobi_rotation(a, d, b.re, b.im);
                if (owner) publish(Kn, nn);
            }
            Ms[wm[0]] = m00; Ms[wm[1]] = m01; Ms[wm[2]] = m10; Ms[wm[3]] = m11;
            FBX_WAVE_SYNC();                        // (the records travel between lanes: the next round's loads stay behind these stores)
            pc = j0.re; psr = j0.im; psi = j1.re; pending = true;
        }
    }
    if (pending) {
        jacobi_apply_v(pc, psr, psi, v0p, v0q, v1p, v1q);
        permute(v0p.re, v0q.re); permute(v0p.im, v0q.im); permute(v1p.re, v1q.re); permute(v1p.im, v1q.im);
    }
    if (init_identity || pending) { Vs[0 * PS + me] = v0p; Vs[1 * PS + me] = v0q; Vs[2 * PS + me] = v1p; Vs[3 * PS + me] = v1q; }
    FBX_WAVE_SYNC();
    return sweep;
}
}  // namespace fbx
