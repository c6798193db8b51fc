// EXPERIMENT of round 3, measured and not adopted (scripts/micro/jacobi_bench.hip -DFBX_JACOBI_CHAIN_FIRST includes this file).
// Idea: put only the ONE block entry that feeds the next pivot on the dependent chain of a Jacobi round, fetch the next
// pivot's diagonal through DPP row shifts, and issue the rest of the round behind the pivot read.  Correct (residual
// 4e-14, same sweeps), and at one wavefront per CU 1036 cycles per round against 931 for the plain loop with the same
// select-free rotation (it needs ~15 more instructions per round, and a lone wavefront pays ~6.5 cycles per instruction
// whatever their order); with four wavefronts per CU -- the benchmark's occupancy -- 1163 against 1181: the round is bound
// by instruction issue and by the shared LDS pipe, not by the latency chain this re-ordering shortens.
#pragma once
// Single-wavefront solver, round 3: only the ONE entry of the updated 2 x 2 block that feeds the next pivot is on
// the dependent chain of a round.
// The next pivot of pair K is { a' , d' , b' }: its diagonal a', d' are rotated diagonal entries (an, dn) of the
// neighbouring pairs -- every lane of column J has computed (an_J, dn_J) itself, so they arrive through two DPP row
// shifts, no LDS -- and b' is, for every pair, element (0,1) of exactly one lane's updated block, except for pairs 1
// and N/2-1 whose b' is element (1,1) of lane (0,2) and element (0,0) of lane (N/2-2, N/2-1) (the two turn-arounds of
// the tournament).  Those two lanes keep their block with rows resp. columns swapped in registers -- a swapped pair
// is rotated by (c, -conj(s)) instead of (c, s), i.e. one sign flip -- so that EVERY lane's chain is
//   rotation -> fetch the row rotation (ds_bpermute; the column stage t = m R_J runs meanwhile) -> n01 = (R_I^H t)_01
//   -> LDS write -> read of the next pivot's b',
// and the other three entries of the block, the eigenvector update, the seat writes and the read-back of the new
// blocks are issued BEHIND the b' read, where they fill its turn-around instead of preceding it (the round-2 loop
// had all of them in front of the next pivot read: 1220 cycles per round against ... here, scripts/micro/jacobi_bench).
// The diagonal blocks carry (an, dn) from the rotation formula itself (exact zeros off the diagonal).
// The arithmetic of every other entry is the expression jacobi_apply_m uses (a sign flip and its inverse are exact).
namespace fbx {
template <int N>
__device__ int jacobi_eigh_wave_chain_first(cplx* __restrict__ Ms, cplx* __restrict__ Vs, int lane, bool init_identity,
                                double expect_n2 = -1.0, double tol2 = FBX_JACOBI_TOL2) {
    constexpr int NB = N / 2, LS = NB * NB, PS = sys_plane<N>();
    static_assert(LS == 64 && NB >= 4, "every lane of the wavefront owns one 2x2 block");
    const int I = lane / NB, J = lane % NB;
    const int me = lane;
    const bool diag = I == J;
    const bool flipI = I == 0 && J == 2;                  // provides b' of pair 1 from its element (1,1): rows swapped
    const bool flipJ = I == NB - 2 && J == NB - 1;        // provides b' of pair NB-1 from its element (0,0): columns swapped
    int wmr[4], rdm[4], wv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {                          // e: REGISTER slot; ea: the matrix element it holds
        const int ea = e ^ (flipI ? 2 : 0) ^ (flipJ ? 1 : 0);
        const int sa = jacobi_seat<N>(2 * I + (ea >> 1)), sb = jacobi_seat<N>(2 * J + (ea & 1));
        wmr[e] = ((sa & 1) * 2 + (sb & 1)) * PS + (sa >> 1) * NB + (sb >> 1);
        rdm[e] = ea * PS + me;
        const int vb = jacobi_seat<N>(2 * J + (e & 1));
        wv[e] = ((e >> 1) * 2 + (vb & 1)) * PS + I * NB + (vb >> 1);
    }
    const int dJ = J * NB + J;
    const int src_lane = (lane & 63) - J + I;
    // sign masks of the swapped lanes: s_eff = -conj(s) = (-sr, +si)
    const int sgnI = flipI ? (int)0x80000000 : 0, sgnJ = flipJ ? (int)0x80000000 : 0;
    auto flip = [](double x, int mask) __attribute__((always_inline)) -> double {
        return __hiloint2double(__double2hiint(x) ^ mask, __double2loint(x));
    };
    if (init_identity) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            cplx v; v.re = (2 * I + (e >> 1) == 2 * J + (e & 1)) ? 1.0 : 0.0; v.im = 0.0;
            Vs[e * PS + me] = v;
        }
    }
    FBX_WAVE_SYNC();
    int sweep = 0;
    double n2 = 0.0;
    for (; sweep < FBX_JACOBI_MAX_SWEEPS; ++sweep) {
        {
            const cplx m00 = Ms[rdm[0]], m01 = Ms[rdm[1]], m10 = Ms[rdm[2]], m11 = Ms[rdm[3]];
            double o2 = 0.0;
            const double q00 = m00.re * m00.re + m00.im * m00.im, q01 = m01.re * m01.re + m01.im * m01.im;
            const double q10 = m10.re * m10.re + m10.im * m10.im, q11 = m11.re * m11.re + m11.im * m11.im;
            const double a_all = (q00 + q01) + (q10 + q11);
            o2 = diag ? q01 + q10 : a_all;                 // (a diagonal lane is never one of the two swapped lanes)
            o2 = uniform(wave_sum(o2));
            if (sweep == 0) n2 = uniform(wave_sum(a_all));
            if (sweep == 0 && expect_n2 >= 0.0 && !(fabs(n2 - expect_n2) <= FBX_BASIS_NORM_TOL * expect_n2)) return -1;
#ifdef FBX_ABL
            if (sweep >= 6) break;
#else
            if (!(o2 > tol2 * n2)) break;
#endif
        }
        double aJ = Ms[0 * PS + dJ].re, dJ_ = Ms[3 * PS + dJ].re;
        cplx bJ = Ms[1 * PS + dJ];
        // (nothing may be pending on entry to the loop: its header would otherwise wait for ALL LDS traffic -- the
        // entry path has the pivot read last in the queue, the back edge has it first with seven writes behind)
        __builtin_amdgcn_s_waitcnt(0xc07f);            // lgkmcnt(0)
        for (int r = 0; r < N - 1; ++r) {
            // ---- chain: this round's rotations.  The first use of b' waits for that one LDS read (the seven seat
            // writes behind it may still be in flight); the read-back of the blocks is issued only now, so that
            // the wait can name b' alone, and completes under the rsqrt chain.
            const double beta = fma(bJ.re, bJ.re, bJ.im * bJ.im);
            __builtin_amdgcn_sched_barrier(0);
            const cplx m00 = Ms[rdm[0]], m01 = Ms[rdm[1]], m10 = Ms[rdm[2]], m11 = Ms[rdm[3]];
            cplx v0p = Vs[0 * PS + me], v0q = Vs[1 * PS + me], v1p = Vs[2 * PS + me], v1q = Vs[3 * PS + me];
            __builtin_amdgcn_sched_barrier(0);
#if defined(FBX_ABL) && (FBX_ABL & 4)
            JRot rJ; rJ.c = 1.0 - beta; rJ.sr = bJ.re; rJ.si = bJ.im; rJ.an = aJ; rJ.dn = dJ_;
#else
            const JRot rJ = jacobi_rotation_beta(aJ, dJ_, bJ.re, bJ.im, beta);
#endif
            const double rIc = __shfl(rJ.c, src_lane), rIsr = __shfl(rJ.sr, src_lane), rIsi = __shfl(rJ.si, src_lane);
            // ---- column stage (needs no row rotation: it overlaps the bpermute)
            const double cJ = rJ.c, sJr = flip(rJ.sr, sgnJ), sJi = rJ.si;
            cplx t00, t01, t10, t11;
            t01.re = cJ * m01.re + (sJr * m00.re - sJi * m00.im);
            t01.im = cJ * m01.im + (sJr * m00.im + sJi * m00.re);
            t11.re = cJ * m11.re + (sJr * m10.re - sJi * m10.im);
            t11.im = cJ * m11.im + (sJr * m10.im + sJi * m10.re);
            t00.re = cJ * m00.re - (sJr * m01.re + sJi * m01.im);
            t00.im = cJ * m00.im - (sJr * m01.im - sJi * m01.re);
            t10.re = cJ * m10.re - (sJr * m11.re + sJi * m11.im);
            t10.im = cJ * m10.im - (sJr * m11.im - sJi * m11.re);
            // ---- chain: the one entry the next pivot needs, to its seat, and the pivot read behind it
            const double cI = rIc, sIr = flip(rIsr, sgnI), sIi = rIsi;
            cplx n01;
            n01.re = cI * t01.re - (sIr * t11.re - sIi * t11.im);
            n01.im = cI * t01.im - (sIr * t11.im + sIi * t11.re);
            if (diag) { n01.re = 0.0; n01.im = 0.0; }
            Ms[wmr[1]] = n01;
            bJ = Ms[1 * PS + dJ];
            {   // a' / d' of the next pivot: rotated diagonals of the neighbouring pairs
                const double an_l = dpp_shift<0x111>(rJ.an), dn_l = dpp_shift<0x111>(rJ.dn);   // row_shr:1 -- from pair J - 1
                const double dn_r = dpp_shift<0x101>(rJ.dn);                                    // row_shl:1 -- from pair J + 1
                aJ = J == 0 ? rJ.an : (J == 1 ? dn_l : an_l);
                dJ_ = J == NB - 1 ? rJ.an : dn_r;
            }
            __builtin_amdgcn_sched_barrier(0);             // everything below stays behind the pivot read
            // ---- behind the pivot read: the rest of the block, the eigenvectors, seats, read-back
            cplx n00, n10, n11;
#if defined(FBX_ABL) && (FBX_ABL & 2)
            n00 = t00; n10 = t10; n11 = t11;
#else
            n00.re = cI * t00.re - (sIr * t10.re - sIi * t10.im);
            n00.im = cI * t00.im - (sIr * t10.im + sIi * t10.re);
            n10.re = cI * t10.re + (sIr * t00.re + sIi * t00.im);
            n10.im = cI * t10.im + (sIr * t00.im - sIi * t00.re);
            n11.re = cI * t11.re + (sIr * t01.re + sIi * t01.im);
            n11.im = cI * t11.im + (sIr * t01.im - sIi * t01.re);
#endif
            if (diag) { n00.re = rJ.an; n00.im = 0.0; n10.re = 0.0; n10.im = 0.0; n11.re = rJ.dn; n11.im = 0.0; }
            Ms[wmr[0]] = n00; Ms[wmr[2]] = n10; Ms[wmr[3]] = n11;
#if !(defined(FBX_ABL) && (FBX_ABL & 1))
            jacobi_apply_v(rJ.c, rJ.sr, rJ.si, v0p, v0q, v1p, v1q);
#endif
#if defined(FBX_ABL) && (FBX_ABL & 16)
            (void)wv; asm volatile("" :: "v"(v0p.re), "v"(v0q.re), "v"(v1p.re), "v"(v1q.re));
#else
            Vs[wv[0]] = v0p; Vs[wv[1]] = v0q; Vs[wv[2]] = v1p; Vs[wv[3]] = v1q;
#endif
        }
        FBX_WAVE_SYNC();
    }
    FBX_WAVE_SYNC();
    return sweep;
}
}  // namespace fbx
