// EXPERIMENT of round 5 (scripts/micro/jacobi_bench.hip -DFBX_JACOBI_H2 includes this file): the single-wavefront 16 x 16 solver using the
// HERMITIAN SYMMETRY of the work matrix without idling a lane.  jacobi_eigh_wave lets lane (I, J) and lane (J, I) both compute their
// whole 2 x 2 block -- conjugate transposes of each other.  Here only the upper block triangle exists; its block (I, J), I < J, has TWO
// workers: lane (I, J) computes column 0 of the updated block, lane (J, I) column 1 -- the same instruction stream (the second worker
// reads the block with its columns swapped and rotates with -conj(s): one sign flip), 24 fp64 instructions instead of 48, two
// ds_write_b128 per lane instead of four.  Every entry is written once, to the seat the tournament permutation assigns it or,
// conjugated, to the mirrored seat, whichever lies in the stored triangle (the 64 x 64 solver of csrc/fbx_eigh64.hpp does the same
// across wavefronts).  The diagonal lanes place the rotated diagonal (closed form an / dn of the rotation) -- the entry a rotation
// annihilates is never stored: the two workers that would read it read a zero cell instead.  Eigenvector block as in
// jacobi_eigh_wave (DPP).  NOT bit-identical to jacobi_eigh_wave (there the two triangles evolve separately and agree to rounding).
#pragma once
namespace fbx {
template <int N>
__device__ int jacobi_eigh_wave_h2(cplx* __restrict__ Ms, cplx* __restrict__ Vs, int lane, bool init_identity,
                                   double expect_n2 = -1.0, double tol2 = FBX_JACOBI_TOL2) {
    constexpr int NB = N / 2, LS = NB * NB, PS = sys_plane<N>();
    static_assert(N == 16 && LS == 64, "every lane of the wavefront owns one eigenvector block");
    lane = FBX_LOCAL(lane);
    const int Ir = lane / NB, Jc = lane % NB;
    const int me = lane;
    const bool diag = Ir == Jc, wb = Ir > Jc;                  // wb: second worker of the upper block (Jc, Ir)
    const int I = wb ? Jc : Ir, J = wb ? Ir : Jc;              // the upper block this lane works on (diag: its own)
    const int col = wb ? 1 : 0;                                // the column of that block it computes
    constexpr int ZERO = 0 * PS + (NB - 1) * NB + 0;           // a cell of the (unused) lower triangle that holds 0
    // reads: x0 / y0 = own column (rows 0 / 1), x1 / y1 = the other column; entries a rotation has annihilated read the zero cell
    int rd_true[4], rd_zs[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int a = e >> 1, b = (e & 1) ^ col;               // e = 2 * row + (0: own column, 1: other column)
        rd_true[e] = (a * 2 + b) * PS + I * NB + J;
        bool zero = false;
        if (!diag) {
            if (I == 0 && J == 1) zero = a == 0 && b == 0;
            else if (J == I + 2) zero = a == 1 && b == 0;
            else if (I == NB - 2 && J == NB - 1) zero = a == 1 && b == 1;
        }
        rd_zs[e] = zero ? ZERO : rd_true[e];
    }
    // writes: entry (row a, own column) of the upper block -> its seat, or conjugated to the mirrored seat
    int wr[2]; int cmask[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        int r2, c2;
        if (diag) { r2 = jacobi_seat<N>(2 * Ir + a); c2 = r2; }
        else { r2 = jacobi_seat<N>(2 * I + a); c2 = jacobi_seat<N>(2 * J + col); }
        const int I2 = r2 >> 1, J2 = c2 >> 1, a2 = r2 & 1, b2 = c2 & 1;
        const bool flipped = I2 > J2 || (I2 == J2 && a2 > b2);
        wr[a] = flipped ? (b2 * 2 + a2) * PS + J2 * NB + I2 : (a2 * 2 + b2) * PS + I2 * NB + J2;
        cmask[a] = flipped ? (int)0x80000000 : 0;
    }
    // rotations: computed from the pivot of pair `pl`, fetched from the diagonal lane of pair `pf`
    const int pl = wb ? Ir : Jc, pf = wb ? Jc : Ir;
    const int dP = pl * NB + pl;
    const int src_lane = pf * NB + pf;
    const int smask = wb ? (int)0x80000000 : 0;                // second worker: s -> -conj(s)
    auto flip = [](double x, int mask) __attribute__((always_inline)) -> double {
        return __hiloint2double(__double2hiint(x) ^ mask, __double2loint(x));
    };
    const bool firstJ = Jc == 0, lastJ = Jc == NB - 1;
    cplx v0p, v0q, v1p, v1q;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        cplx v;
        if (init_identity) { v.re = (2 * Ir + (e >> 1) == 2 * Jc + (e & 1)) ? 1.0 : 0.0; v.im = 0.0; }
        else v = Vs[e * PS + me];
        if (e == 0) v0p = v; else if (e == 1) v0q = v; else if (e == 2) v1p = v; else v1q = v;
    }
    auto permute = [&](double& p, double& q) __attribute__((always_inline)) {
        const double right = firstJ ? q : p;
        const double from_left = dpp_shift<0x111>(right);
        const double from_right = dpp_shift<0x101>(q);
        const double pn = firstJ ? p : from_left;
        const double qn = lastJ ? p : from_right;
        p = pn; q = qn;
    };
    if (lane == 0) { cplx z; z.re = 0.0; z.im = 0.0; Ms[ZERO] = z; }
    FBX_WAVE_SYNC();
    int ra0 = rd_true[0], ra1 = rd_true[1], ra2 = rd_true[2], ra3 = rd_true[3];
    double pc = 1.0, psr = 0.0, psi = 0.0;
    bool pending = false;
    int sweep = 0;
    double n2 = 0.0;
    for (; sweep < FBX_JACOBI_MAX_SWEEPS; ++sweep) {
        {
            const cplx x0 = Ms[ra0], y0 = Ms[ra2], x1 = Ms[ra1];
            double o2, dg = 0.0;
            if (diag) { o2 = x1.re * x1.re + x1.im * x1.im; dg = x0.re * x0.re + Ms[3 * PS + dP].re * Ms[3 * PS + dP].re; }   // (0,1) entry; a, d
            else o2 = (x0.re * x0.re + x0.im * x0.im) + (y0.re * y0.re + y0.im * y0.im);
            o2 = 2.0 * uniform(wave_sum(o2));
            if (sweep == 0) n2 = o2 + uniform(wave_sum(dg));
            if (sweep == 0 && expect_n2 >= 0.0 && !(fabs(n2 - expect_n2) <= FBX_BASIS_NORM_TOL * expect_n2)) return -1;
            if (!(o2 > tol2 * n2)) break;
        }
        for (int r = 0; r < N - 1; ++r) {
            const double aP = Ms[0 * PS + dP].re, dP_ = Ms[3 * PS + dP].re;
            const cplx bP = Ms[1 * PS + dP];
            const cplx x0 = Ms[ra0], x1 = Ms[ra1], y0 = Ms[ra2], y1 = Ms[ra3];
            ra0 = rd_zs[0]; ra1 = rd_zs[1]; ra2 = rd_zs[2]; ra3 = rd_zs[3];
            if (pending) {
                jacobi_apply_v(pc, psr, psi, v0p, v0q, v1p, v1q);
                permute(v0p.re, v0q.re); permute(v0p.im, v0q.im); permute(v1p.re, v1q.re); permute(v1p.im, v1q.im);
            }
            const JRot rL = jacobi_rotation(aP, dP_, bP.re, bP.im);               // pair pl
            const double fc = __shfl(rL.c, src_lane), fsr = __shfl(rL.sr, src_lane), fsi = __shfl(rL.si, src_lane);   // pair pf
            // column rotation = pair J (the locally computed one), row rotation = pair I (the fetched one)
            const double cJ = rL.c, sJr = flip(rL.sr, smask), sJi = rL.si;
            cplx t0, t1, n0, n1;
            t0.re = cJ * x0.re - (sJr * x1.re + sJi * x1.im);
            t0.im = cJ * x0.im - (sJr * x1.im - sJi * x1.re);
            t1.re = cJ * y0.re - (sJr * y1.re + sJi * y1.im);
            t1.im = cJ * y0.im - (sJr * y1.im - sJi * y1.re);
            n0.re = fc * t0.re - (fsr * t1.re - fsi * t1.im);
            n0.im = fc * t0.im - (fsr * t1.im + fsi * t1.re);
            n1.re = fc * t1.re + (fsr * t0.re + fsi * t0.im);
            n1.im = fc * t1.im + (fsr * t0.im - fsi * t0.re);
            if (diag) { n0.re = rL.an; n0.im = 0.0; n1.re = rL.dn; n1.im = 0.0; }
            n0.im = flip(n0.im, cmask[0]); n1.im = flip(n1.im, cmask[1]);
            Ms[wr[0]] = n0; Ms[wr[1]] = n1;
            // the eigenvector block rotates with its own column pair Jc: the local rotation, for the second worker the fetched one
            pc = wb ? fc : rL.c; psr = wb ? fsr : rL.sr; psi = wb ? fsi : rL.si; pending = true;
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
