// EXPERIMENT of round 5 (scripts/micro/jacobi_bench.hip -DFBX_JACOBI_VDPP includes this file): the single-wavefront 16 x 16 solver
// with the EIGENVECTOR block exchanged through DPP row shifts instead of LDS.  The tournament permutation only moves eigenvector
// COLUMNS between neighbouring lanes of a block row (top column of pair J to pair J + 1, bottom column to pair J - 1, three
// exceptions at the ends of the ring), so four ds_write_b128 + four ds_read_b128 per round become 40 32-bit VALU instructions
// (16 v_mov_dpp + 24 v_cndmask).  Same rotations applied to the same data: bit-identical to jacobi_eigh_wave.  Aimed at the
// two-waves-per-SIMD kernel, where the round is bound by the LDS pipe (8 b128 writes = 104 of its 160 LDS cycles); a lone
// wavefront is issue-bound and pays for the extra instructions.
#pragma once
namespace fbx {
template <int N>
__device__ int jacobi_eigh_wave_vdpp(cplx* __restrict__ Ms, cplx* __restrict__ Vs, int lane, bool init_identity,
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
    const int dJ = J * NB + J;
    const int src_lane = (lane & 63) - J + I;
    const bool first = J == 0, second = J == 1, last = J == NB - 1;
    (void)second;
    // eigenvector block in registers, NORMAL layout: v0p = V[2I][t_J], v0q = V[2I][b_J], v1p = V[2I+1][t_J], v1q = V[2I+1][b_J]
    cplx v0p, v0q, v1p, v1q;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        cplx v;
        if (init_identity) { v.re = (2 * I + (e >> 1) == 2 * J + (e & 1)) ? 1.0 : 0.0; v.im = 0.0; }
        else v = Vs[e * PS + me];
        if (e == 0) v0p = v; else if (e == 1) v0q = v; else if (e == 2) v1p = v; else v1q = v;
    }
    // the tournament permutation of the columns, in registers: top column t_J -> t_{J+1} (t_0 stays, t_last -> b_last),
    // bottom column b_J -> b_{J-1} (b_0 -> t_1)
    auto permute = [&](double& p, double& q) __attribute__((always_inline)) {
        const double right = first ? q : p;                       // what this lane hands to its right neighbour
        const double from_left = dpp_shift<0x111>(right);          // row_shr:1
        const double from_right = dpp_shift<0x101>(q);             // row_shl:1
        const double pn = first ? p : from_left;
        const double qn = last ? p : from_right;
        p = pn; q = qn;
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
                if (!(I == J && (e == 0 || e == 3))) o2 += a2;
            }
            o2 = uniform(wave_sum(o2));
            if (sweep == 0) n2 = uniform(wave_sum(a_all));
            if (sweep == 0 && expect_n2 >= 0.0 && !(fabs(n2 - expect_n2) <= FBX_BASIS_NORM_TOL * expect_n2)) return -1;
            if (!(o2 > tol2 * n2)) break;
        }
        for (int r = 0; r < N - 1; ++r) {
            const double aJ = Ms[0 * PS + dJ].re, dJ_ = Ms[3 * PS + dJ].re;
            const cplx bJ = Ms[1 * PS + dJ];
            cplx m00 = Ms[0 * PS + me], m01 = Ms[1 * PS + me];
            cplx m10 = Ms[2 * PS + me], m11 = Ms[3 * PS + me];
            if (pending) {          // the previous round's eigenvector update, then its seat permutation -- independent of the chain below
                jacobi_apply_v(pc, psr, psi, v0p, v0q, v1p, v1q);
                permute(v0p.re, v0q.re); permute(v0p.im, v0q.im); permute(v1p.re, v1q.re); permute(v1p.im, v1q.im);
            }
            const JRot rJ = jacobi_rotation(aJ, dJ_, bJ.re, bJ.im);
            JRot rI;
            rI.c = __shfl(rJ.c, src_lane); rI.sr = __shfl(rJ.sr, src_lane); rI.si = __shfl(rJ.si, src_lane);
            jacobi_apply_m(rI.c, rI.sr, rI.si, rJ.c, rJ.sr, rJ.si, m00, m01, m10, m11);
            if (I == J) {
                m01.re = m01.im = 0.0; m10.re = m10.im = 0.0;
                m00.im = 0.0; m11.im = 0.0;
            }
            Ms[wm[0]] = m00; Ms[wm[1]] = m01; Ms[wm[2]] = m10; Ms[wm[3]] = m11;
            pc = rJ.c; psr = rJ.sr; psi = rJ.si; pending = true;
        }
    }
    if (pending) {
        jacobi_apply_v(pc, psr, psi, v0p, v0q, v1p, v1q);
        permute(v0p.re, v0q.re); permute(v0p.im, v0q.im); permute(v1p.re, v1q.re); permute(v1p.im, v1q.im);
    }
    Vs[0 * PS + me] = v0p; Vs[1 * PS + me] = v0q; Vs[2 * PS + me] = v1p; Vs[3 * PS + me] = v1q;
    FBX_WAVE_SYNC();
    return sweep;
}
}  // namespace fbx
