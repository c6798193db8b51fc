// fbx_eigh64.hpp -- the 64 x 64 Hermitian eigensolver of the 3-qubit kernels (1024 threads, one 2 x 2 block per thread on a
// 32 x 32 grid), round 4: the same cyclic two-sided Jacobi in the Brent-Luk systolic form as jacobi_eigh_simple<64, 1024>
// (fbx_eigh.hpp) with the two things that bounded its round removed (LDS WRITES, 78 B/clk/CU, and two
// workgroup barriers per round):
//
//   * the eigenvector block never touches LDS.  A wavefront owns two block rows (lanes 0-31 = row 2w, 32-63 = row 2w + 1),
//     and the tournament permutation only moves eigenvector COLUMNS -- the top column of pair J goes to pair J + 1, the bottom
//     one to pair J - 1, with the three exceptions at the ends of the ring -- i.e. between neighbouring lanes of the same
//     wavefront: two full-wave DPP shifts (wave_shr:1 / wave_shl:1) and three selects per 32-bit register instead of four
//     ds_write_b128 + four ds_read_b128 per thread and round;
//   * the work matrix is double buffered in Ms / Vs (Vs is free now): a round reads buffer `cur` and writes the permuted blocks
//     to the other one, so nothing separates its reads from its writes and ONE barrier per round is left.
//
// Interface as jacobi_eigh_simple: on entry Ms holds the Hermitian matrix (element-major layout, sys_pos<64>), Vs the starting
// basis unless init_identity; on exit Ms is diagonal, Vs holds the eigenvectors.  Replaces scipy.linalg.eigh at
// operator_tools/project_superoperators.py:30 for D = 64.  Parity is on V f(Lambda) V^H, never on eigenvectors.
#pragma once
#include "fbx_eigh.hpp"

namespace fbx {

// value of the previous / next lane of the wavefront (lane 0 / 63 keep their own)
__device__ __forceinline__ double dpp_prev_lane(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);      // wave_shr:1
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dpp_next_lane(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xf, 0xf, false);      // wave_shl:1
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// one component (re or im of one row) of the eigenvector block's two columns through the tournament permutation:
//   new top[J] = old top[J - 1] (J >= 2), old bottom[0] (J = 1), old top[0] (J = 0)
//   new bottom[J] = old bottom[J + 1] (J <= 30), old top[31] (J = 31)
// (Measured and dropped: the shift and the end-of-ring select fused into one v_cndmask_b32_dpp per register through inline
// assembly -- 48 instead of 80 instructions per round for the two blocks, and 6 % SLOWER: the assembly blocks pin the order of
// 16 instructions and need their own s_nop for the DPP read hazard, which the scheduler otherwise hides.  Round 5, on the
// published-rotation solver whose round is bound by vector issue: all sixteen register halves of a thread in ONE block, VCC = the
// lane mask of the ring ends, 48 instructions instead of 80 -- 2222-2272 against 1865 cycles per round: the block's 48 operands live
// at once cost the 128-register thread 244 B of scratch.)
__device__ __forceinline__ void seat_shift(double& top, double& bot, bool first, bool last) {
    const double send = first ? bot : top;             // pair 0 hands its BOTTOM column to pair 1 and keeps its top one
    const double from_prev = dpp_prev_lane(send);
    const double from_next = dpp_next_lane(bot);
    const double t_old = top;
    top = first ? t_old : from_prev;
    bot = last ? t_old : from_next;
}

// Round 5: the same permutation for a thread that holds TWO neighbouring column pairs (2 tau, 2 tau + 1) of a ring whose 32 pairs
// are the sixteen lanes of ONE DPP row.  Of the four columns (A, B) = (top, bottom) of pair 2 tau and (C, D) of pair 2 tau + 1, two
// stay in the thread (A -> top of 2 tau + 1, D -> bottom of 2 tau: register renaming) and two travel (C to the right neighbour's
// pair 2 tau + 2, B to the left neighbour's pair 2 tau - 1) -- and a DPP ROW shift has the ring's two ends built in: the lane
// without a source keeps its `old` operand, which is exactly what the ends do (lane 0: the top of pair 0 stays; lane 15: the top of
// pair 31 becomes its bottom).  4 DPP moves + 2 selects (the bottom of pair 0 becomes the top of pair 1) per FOUR doubles, against
// 2 x (4 + 6) with seat_shift: 24 instead of 80 instructions per thread and round.
__device__ __forceinline__ double dpp_row_shr_old(double old, double src) {        // lane i <- src of lane i - 1 of its row of 16; lane 0 <- old
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), 0x111, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), 0x111, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dpp_row_shl_old(double old, double src) {        // lane i <- src of lane i + 1; lane 15 <- old
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), 0x101, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), 0x101, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void ring_shift(double& A, double& B, double& C, double& D, bool first) {
    const double nA = dpp_row_shr_old(A, C);           // top of pair 2 tau <- top of pair 2 tau - 1 (left neighbour's C); pair 0 keeps its top
    const double nD = dpp_row_shl_old(C, B);           // bottom of pair 2 tau + 1 <- bottom of pair 2 tau + 2; pair 31: its own top
    const double nC = first ? B : A;                   // top of pair 2 tau + 1 <- top of pair 2 tau; pair 1: the bottom of pair 0
    A = nA; B = D; C = nC; D = nD;
}

// Where the entry (slot r, slot c), r != c, of the Hermitian work matrix is STORED: only the upper block triangle is kept
// (block row <= block column) and, inside a diagonal block, the entry (even slot, odd slot).  Returns the offset in the
// element-major layout; `conj` says that the stored value is the conjugate of M[r][c].
__device__ __forceinline__ int herm_store_pos(int r, int c, bool& conj) {
    constexpr int N = 64, PS = sys_plane<N>();
    const int R = r >> 1, C = c >> 1;
    conj = R > C || (R == C && (r & 1));
    const int rr = conj ? c : r, cc = conj ? r : c;
    const int pl = (rr & 1) * 2 + (cc & 1);
    return pl * PS + sys_pos<N>(rr >> 1, cc >> 1, pl);
}

// jacobi_eigh64: roles.  The Hermitian symmetry halves the matrix work and the eigenvector work has none to offer, so the two
// are given to different wavefronts (two of each per SIMD):
//   * wavefronts 0-7, MATRIX role: 496 threads own one strictly-upper block (Iu, Ju) each and apply the rotations of its row
//     pair and of its column pair; every entry is written once, to the seat the tournament permutation
//     assigns it or to the mirrored seat (conjugated), whichever lies in the stored triangle.
//   * wavefronts 8-15, EIGENVECTOR role: thread (I, J), I < 16, owns the eigenvector blocks (I, J) and (I + 16, J) -- same
//     column pair, one rotation for both -- in registers, exchanged with the neighbouring lanes as above.
// Two sets of LDS addresses (one per buffer) alternate between consecutive rounds, so a round contains no address arithmetic.
// The two roles are two separate loop nests behind a SCALAR branch (the wavefront index is wave-uniform), with the same sequence
// of workgroup barriers -- two per convergence test, one per round: their registers overlap instead of adding up (as one loop
// body with both roles the solver needed all 128 registers of a 1024-thread workgroup and, behind its call boundary in the
// 3-qubit kernels, saved 49 callee-saved registers to scratch per decomposition).
// Rounds 1-4 (-DFBX_EIGH64_LOCAL_ROTATIONS keeps that form for A/B builds): every thread evaluated the rotations it applies from
// the pivot blocks -- no exchange, no second barrier, and 2 x 496 + 512 evaluations of 32 rotations per round.
namespace eigh64 {
constexpr int N = 64, NB = 32, PS = sys_plane<N>(), NUP = NB * (NB - 1) / 2, NT = 1024;

#ifndef FBX_EIGH64_LOCAL_ROTATIONS
// PUBLISHED ROTATIONS (round 5).  The rotation of a pair is evaluated ONCE per round instead of by every thread that applies it
// (an evaluation is ~30 vector instructions with two reciprocal-square-root chains: 60 of the 106 instructions of a
// matrix-role thread's round and 30 of the 160 of an eigenvector-role thread's).  No second barrier: the pivot of NEXT round's
// pair K' is made of two diagonal entries and one off-diagonal entry that all exist in ONE matrix-role thread at the end of
// this round -- the tournament permutation (jacobi_seat) forms
//   pair 0 = {top 0, bottom 1}, pair 1 = {bottom 0, bottom 2}, pair K = {top K-1, bottom K+1} (2 <= K <= 30), pair 31 = {top 30, top 31},
// so the thread of block (0,1), (0,2), (K-1,K+1), (30,31) holds the updated entry b' in its registers and the rotated diagonals
// a' (of its row pair) and d' (of its column pair) in the two records it has just applied.  That thread evaluates the next
// rotation and publishes the record {c, s, a'', d''} in the buffer the round writes (entries 528 + K of planes 0 / 1 / 2, see the private
// layout below: (c, Re s) / (Im s, a'') / (d'', -)).  Everybody reads two records instead of two pivot
// blocks.  Same inputs, same function: BIT-IDENTICAL to the locally computed rotations (micro/jacobi64_bench prints a hash).
// What is sparse goes to ONE wavefront per role: a branch that a few lanes of EVERY wavefront take costs every wavefront its
// whole instruction stream, and an LDS instruction with four active lanes a full issue.  The 32 publishing blocks are threads
// 0-31 (wavefront 0 evaluates rotations, wavefronts 1-7 never enter that branch); the 32 threads of eigenvector row I = 0
// (wavefront 8) place the rotated diagonals and the annihilated entries of all pairs.  Measured, cycles per round in isolation
// (256 workgroups, profiles/r05/jacobi64_published.txt): 2615 local rotations; 2570 published; 2165 + publishing blocks in one
// wavefront; 1938 + diagonals placed by one wavefront.  Without the eigenvector role's arithmetic the matrix chain alone takes
// 1252.  Then (round 5, later): the eigenvector role with two column pairs per thread and its rings in DPP rows (ring_shift:
// 24 instead of 80 exchange instructions per thread and round) -> 1864 with the owner-indexed layout -> 1749; the eigenvector role one
// round BEHIND plus s_setprio for the matrix role -> 1691 (either alone: 1725 / 1764; on the heavier eigenvector role of the earlier
// steps the lag lost, 2019 against 1938).  Measured and dropped: DPP shifts without an `old` operand (the compiler does not fold
// them into the selects); the fused shift + select as inline assembly; 24-byte record loads.
// PRIVATE LAYOUT (round 5).  sys_pos<64> was laid out for an LDS that serves a b128 access in groups of 8 consecutive lanes on 8
// bank groups.  gfx950 serves ds_read_b128 in four groups of SIXTEEN lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same
// + 32 -- on sixteen 16-byte slots (MI355X_MICROARCH.md, LDS), and ds_write_b128 in eight groups of 8 consecutive lanes on eight:
// counted with that model (tests/test_host_logic.py) a round of the solver spent 322 of its 890 LDS-array cycles in bank conflicts
// (SQ_LDS_BANK_CONFLICT: 345 of 958), 218 of them on reads.  Inside the solver the work matrix is therefore stored by OWNER: entry x
// of every plane belongs to the matrix-role thread x (x < 496: its strictly-upper block), to the diagonal block of pair x - 496
// (x < 528) or to the record of pair x - 528 -- every wavefront reads its blocks from 64 consecutive entries, the eigenvector
// role its records from 32 consecutive ones: no conflict for any lane grouping.  (The bottom planes are rotated by 2 inside
// each aligned group of 8, as in sys_pos: at the ends of the tournament ring two lanes of a store group write the same block
// column in the two planes.)  The caller's layout is left at the first round and restored for the results.
__device__ __forceinline__ int priv_slot(int x, int e) { return (x & ~7) | ((x + 2 * (e & 1)) & 7); }
// matrix-role thread of the strictly-upper block (R, C) -- the inverse of the enumeration in matrix_role
__device__ __forceinline__ int thread_of(int R, int C) {
    if (R == 0 && C == 1) return 0;
    if (R == 0 && C == 2) return 1;
    if (R == NB - 2 && C == NB - 1) return NB - 1;
    if (C == R + 2 && R >= 1) return R + 1;
    const int u = R <= 15 ? 32 * R + (C - R - 1) : 32 * (30 - R) + C;      // index in the full enumeration (rows i and 30 - i share a run of 32)
    const int r = u / NB, c = u % NB;
    const int before = r == 0 ? 0 : 3 + 2 * (r - 1);                      // publishing blocks in earlier runs
    const int within = r == 0 ? (c > 0) + (c > 1) : r < 15 ? (c > 1) + (c > 32 - r) : (c > 1);
    return NB + u - before - within;
}
// offset (plane included) of entry (slot r, slot c) of the Hermitian work matrix in the private layout; `conj` as herm_store_pos
__device__ __forceinline__ int priv_store_pos(int r, int c, bool& conj) {
    const int R = r >> 1, C = c >> 1;
    conj = R > C || (R == C && (r & 1));
    const int rr = conj ? c : r, cc = conj ? r : c;
    const int pl = (rr & 1) * 2 + (cc & 1);
    const int x = (rr >> 1) == (cc >> 1) ? NUP + (rr >> 1) : thread_of(rr >> 1, cc >> 1);
    return pl * PS + priv_slot(x, pl);
}
__device__ __forceinline__ int rec_pos(int K, int e) { return e * PS + priv_slot(NUP + NB + K, e); }
#endif

// the loop nest both roles run: `test(o2, n2)` adds the role's share of the off-diagonal / total norm, `round(rd, wr)` is one
// round reading the buffer described by the first address set and writing the one described by the second (barrier NOT included)
template <int NR, int NW, class Test, class Round>
__device__ __forceinline__ int sweeps(int (&ra)[NR], int (&wa)[NW], int delta, double* red, double tol2, Test test, Round round) {
    int rb[NR], wb[NW];
#pragma unroll
    for (int k = 0; k < NR; ++k) rb[k] = ra[k] + delta;
#pragma unroll
    for (int k = 0; k < NW; ++k) wb[k] = wa[k] + delta;
    int sweep = 0;
    for (; sweep < FBX_JACOBI_MAX_SWEEPS; ++sweep) {
        double o2 = 0.0, n2 = 0.0;
        test(ra, o2, n2);
        block_sum2<NT>(o2, n2, red);
        o2 = uniform(o2); n2 = uniform(n2);
        if (!(o2 > tol2 * n2)) break;
        for (int r = 0; r < (N - 2) / 2; ++r) {
            round(ra, wb); FBX_BLOCK_SYNC();
            round(rb, wa); FBX_BLOCK_SYNC();
        }
        round(ra, wb); FBX_BLOCK_SYNC();
        // 63 rounds: the matrix is in the other buffer now -- the two address sets change places
#pragma unroll
        for (int k = 0; k < NR; ++k) { const int x = ra[k]; ra[k] = rb[k]; rb[k] = x; }
#pragma unroll
        for (int k = 0; k < NW; ++k) { const int x = wa[k]; wa[k] = wb[k]; wb[k] = x; }
    }
    return sweep;
}

__device__ __forceinline__ double flip_sign(double x, unsigned mask) {
    return __hiloint2double(__double2hiint(x) ^ (int)mask, __double2loint(x));
}

#ifndef FBX_EIGH64_LOCAL_ROTATIONS
__device__ __forceinline__ int matrix_role(cplx* Ms, int delta, int t, double* red, double tol2) {
    // the 32 blocks that publish a record are threads 0-31 (thread K' publishes pair K'); the others follow in this order with
    // those 32 left out: rows i and 30 - i have 32 strictly-upper blocks between them, a run of 32 indices takes those two rows
    // (i < 15), the last one the 16 blocks of row 15 -- runs of consecutive column pairs, as the conflict-free layout wants them
    int Iu, Ju;
    const bool mrole = t < NUP;
    if (t < NB) {
        Iu = t == 0 ? 0 : t == 1 ? 0 : t == NB - 1 ? NB - 2 : t - 1;
        Ju = t == 0 ? 1 : t == 1 ? 2 : t == NB - 1 ? NB - 1 : t + 1;
    } else {
        int u = t - NB;                                     // index among the non-publishing blocks -> index in the full enumeration
        // (publishing blocks in the full enumeration, ascending: (0,1), (0,2), (30,31); rows r and 30 - r: c = 1 and c = 32 - r; row 15: c = 1)
        if (0 <= u) ++u;
        if (1 <= u) ++u;
        if (31 <= u) ++u;
#pragma unroll
        for (int rr = 1; rr < 15; ++rr) { if (32 * rr + 1 <= u) ++u; if (32 * rr + 32 - rr <= u) ++u; }
        if (32 * 15 + 1 <= u) ++u;
        const int r = u / NB, c = u % NB;
        const bool lower_row = c >= NB - 1 - r;
        Iu = mrole ? (lower_row ? 30 - r : r) : 0;
        Ju = mrole ? (lower_row ? Iu + 1 + (c - (NB - 1 - r)) : r + 1 + c) : 1;
    }
    static_assert(NUP == 15 * NB + 16, "496 strictly-upper blocks");
    // which of next round's pairs this block holds the pivot entry of (see the head of the namespace), and where in the block
    const bool k_first = Iu == 0 && Ju == 1, k_second = Iu == 0 && Ju == 2, k_last = Iu == NB - 2 && Ju == NB - 1;
    const bool owner = mrole && (k_first || k_second || k_last || (Ju == Iu + 2 && Iu >= 1));
    const int Kn = k_first ? 0 : k_second ? 1 : k_last ? NB - 1 : Iu + 1;
    // own block (planes b = 0 / 1), record of the row pair, record of the column pair (entries 0 / 1; entry 2 = entry 0 + 2 planes);
    // the four seats and the record this thread publishes
    int ra[6], wa[6];
    unsigned sgm[4];                                        // sign bit to flip on the imaginary part of an entry stored mirrored
    // (the solver works in Vs first: the caller's matrix is copied there, into the private layout -- `delta` added here, taken
    //  away again by the address sets of the other buffer)
    ra[0] = delta + priv_slot(t, 0); ra[1] = delta + priv_slot(t, 1);
    ra[2] = delta + rec_pos(Iu, 0); ra[3] = delta + rec_pos(Iu, 1);
    ra[4] = delta + rec_pos(Ju, 0); ra[5] = delta + rec_pos(Ju, 1);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        bool cj;
        wa[e] = delta + priv_store_pos(jacobi_seat<N>(2 * Iu + (e >> 1)), jacobi_seat<N>(2 * Ju + (e & 1)), cj);
        sgm[e] = cj ? 0x80000000u : 0u;
    }
    wa[4] = delta + rec_pos(Kn, 0); wa[5] = delta + rec_pos(Kn, 1);
    if (mrole) {                                            // caller's layout -> private layout, in the other buffer
#pragma unroll
        for (int e = 0; e < 4; ++e) Ms[e * PS + ra[e & 1]] = Ms[e * PS + sys_pos<N>(Iu, Ju, e)];
    }
    FBX_BLOCK_SYNC();
    auto test = [&](const int (&rd)[6], double& o2, double& n2) __attribute__((always_inline)) {
        if (mrole) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const cplx v = Ms[e * PS + rd[e & 1]];
                o2 += 2.0 * (v.re * v.re + v.im * v.im);
            }
            n2 = o2;
        }
    };
    auto round = [&](const int (&rd)[6], const int (&wr)[6]) __attribute__((always_inline)) {
        if (mrole) {
            const cplx i0 = Ms[rd[2]], i1 = Ms[rd[3]];      // (c, Re s), (Im s, a') of the row pair
            const cplx j0 = Ms[rd[4]], j1 = Ms[rd[5]];      // ... of the column pair
            cplx m00 = Ms[0 * PS + rd[0]], m01 = Ms[1 * PS + rd[1]];
            cplx m10 = Ms[2 * PS + rd[0]], m11 = Ms[3 * PS + rd[1]];
            jacobi_apply_m(i0.re, i0.im, i1.re, j0.re, j0.im, j1.re, m00, m01, m10, m11);
            const cplx b = k_second ? m11 : k_last ? m00 : m01;    // (a publishing block's pivot entry is never stored mirrored)
            m00.im = flip_sign(m00.im, sgm[0]); m01.im = flip_sign(m01.im, sgm[1]);
            m10.im = flip_sign(m10.im, sgm[2]); m11.im = flip_sign(m11.im, sgm[3]);
            Ms[wr[0]] = m00; Ms[wr[1]] = m01; Ms[wr[2]] = m10; Ms[wr[3]] = m11;
            if (owner) {    // (behind the block's stores: they are on their way while the two reciprocal-square-root chains run)
                const double dI = Ms[2 * PS + rd[2]].re, dJ = Ms[2 * PS + rd[4]].re;
                const double a = k_second ? dI : i1.im;     // pair 1 takes the BOTTOM of pair 0
                const double d = k_last ? j1.im : dJ;       // pair 31 takes the TOP of pair 31
                const JRot n = jacobi_rotation(a, d, b.re, b.im);
                cplx e0, e1, e2;
                e0.re = n.c; e0.im = n.sr; e1.re = n.si; e1.im = n.an; e2.re = n.dn; e2.im = 0.0;
                Ms[wr[4]] = e0; Ms[wr[5]] = e1; Ms[2 * PS + wr[4]] = e2;
            }
        }
    };
    // (the matrix role is the chain of the round -- loads, update, stores, barrier; the eigenvector role below runs one round BEHIND, out
    //  of registers: with the priority here its arithmetic fills the time the matrix role waits for LDS.  1749 -> 1691 cycles per round
    //  together, 1725 / 1764 each alone)
    __builtin_amdgcn_s_setprio(3);
    const int sweep = sweeps<6, 6>(ra, wa, -delta, red, tol2, test, round);
    __builtin_amdgcn_s_setprio(0);
    return sweep;
}

__device__ __forceinline__ int vector_role(cplx* Ms, cplx* Vs, int delta, int tv, bool init_identity, double* red, double tol2) {
    // thread (R, tau): rows 2R, 2R + 1 of the eigenvector matrix, column pairs 2 tau and 2 tau + 1 (blocks b = 0, 1); the sixteen
    // lanes of a DPP row are one ring of 32 pairs (ring_shift above), a wavefront holds four row pairs
    const int R = tv / 16, tau = tv % 16;
    const bool first = tau == 0;
    // the 32 threads of the first two rings (half a wavefront) also place the rotated diagonal and the annihilated entry of pair Kd
    const bool diag = R < 2;
    const int Kd = 2 * tau + (R & 1);
    // pivot block of pair Kd (convergence test only), the records of the thread's two pairs, the record of pair Kd; the three seats
    // (the sixteen lanes of a ring read every second record: 2-way bank conflicts on these four loads -- off the round's chain, this
    //  role runs a round behind; a second, conflict-free copy of the records cost the publishing wavefront three more stores ON
    //  the chain: 1691 against 1656 cycles per round)
    int ra[7], wa[3];
    {
        const int sa = jacobi_seat<N>(2 * Kd), sd = jacobi_seat<N>(2 * Kd + 1);
        bool cj;
        ra[0] = delta + priv_slot(NUP + Kd, 0); ra[1] = delta + priv_slot(NUP + Kd, 1);
        ra[2] = delta + rec_pos(2 * tau, 0); ra[3] = delta + rec_pos(2 * tau, 1);
        ra[4] = delta + rec_pos(2 * tau + 1, 0); ra[5] = delta + rec_pos(2 * tau + 1, 1);
        ra[6] = delta + rec_pos(Kd, 0);
        wa[0] = delta + 3 * (sa & 1) * PS + priv_slot(NUP + (sa >> 1), sa & 1);
        wa[1] = delta + 3 * (sd & 1) * PS + priv_slot(NUP + (sd >> 1), sd & 1);
        wa[2] = delta + priv_store_pos(sa, sd, cj);
    }
    // eigenvector blocks (R, 2 tau + b) in registers: v0 = row 2R, v1 = row 2R + 1; p = top column of the pair, q = bottom column
    cplx v0p[2], v0q[2], v1p[2], v1q[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int J = 2 * tau + b;
        if (init_identity) {
            v0p[b].re = (R == J) ? 1.0 : 0.0; v0p[b].im = 0.0; v1q[b] = v0p[b];
            v0q[b].re = 0.0; v0q[b].im = 0.0; v1p[b] = v0q[b];
        } else {
            v0p[b] = Vs[0 * PS + sys_pos<N>(R, J, 0)]; v0q[b] = Vs[1 * PS + sys_pos<N>(R, J, 1)];
            v1p[b] = Vs[2 * PS + sys_pos<N>(R, J, 0)]; v1q[b] = Vs[3 * PS + sys_pos<N>(R, J, 1)];
        }
    }
    FBX_BLOCK_SYNC();                                       // Vs is a matrix buffer from here on -- the one the solver starts in
    // the pivot block of pair Kd into the private layout, and the record of the first round from it (visible behind
    // the barriers of the first convergence test)
    if (diag) {
        const cplx a0 = Ms[0 * PS + sys_pos<N>(Kd, Kd, 0)], b0 = Ms[1 * PS + sys_pos<N>(Kd, Kd, 1)], d0 = Ms[3 * PS + sys_pos<N>(Kd, Kd, 1)];
        Ms[0 * PS + ra[0]] = a0; Ms[1 * PS + ra[1]] = b0; Ms[3 * PS + ra[1]] = d0;
        const JRot n = jacobi_rotation(a0.re, d0.re, b0.re, b0.im);
        cplx e0, e1, e2;
        e0.re = n.c; e0.im = n.sr; e1.re = n.si; e1.im = n.an; e2.re = n.dn; e2.im = 0.0;
        Ms[ra[6]] = e0; Ms[delta + rec_pos(Kd, 1)] = e1; Ms[2 * PS + ra[6]] = e2;
    }
    FBX_BLOCK_SYNC();
    double ev_a = 0.0, ev_d = 0.0;                          // (diag threads) the pair's diagonal at the last convergence test
    auto test = [&](const int (&rd)[7], double& o2, double& n2) __attribute__((always_inline)) {
        if (diag) {
            ev_a = Ms[0 * PS + rd[0]].re; ev_d = Ms[3 * PS + rd[1]].re;
            const cplx b = Ms[1 * PS + rd[1]];
            o2 = 2.0 * (b.re * b.re + b.im * b.im);
            n2 = o2 + ev_a * ev_a + ev_d * ev_d;
        }
    };
    // ONE ROUND BEHIND: the eigenvectors depend on nothing but the records, so this role applies the rotations of round r - 1 (in
    // registers since the last round) while the matrix role works on round r, and fetches round r's records afterwards.
    // Same operations in the same order on every number.
    double pr[6] = {1.0, 0.0, 0.0, 1.0, 0.0, 0.0};
    bool pending = false;
    auto flush = [&]() __attribute__((always_inline)) {
        jacobi_apply_v(pr[0], pr[1], pr[2], v0p[0], v0q[0], v1p[0], v1q[0]);
        jacobi_apply_v(pr[3], pr[4], pr[5], v0p[1], v0q[1], v1p[1], v1q[1]);
        ring_shift(v0p[0].re, v0q[0].re, v0p[1].re, v0q[1].re, first); ring_shift(v0p[0].im, v0q[0].im, v0p[1].im, v0q[1].im, first);
        ring_shift(v1p[0].re, v1q[0].re, v1p[1].re, v1q[1].re, first); ring_shift(v1p[0].im, v1q[0].im, v1p[1].im, v1q[1].im, first);
    };
    auto round = [&](const int (&rd)[7], const int (&wr)[3]) __attribute__((always_inline)) {
        // (the records of this round are only applied in the next: they are loaded BEHIND the arithmetic -- before the barrier, the
        //  next round overwrites them -- so that right after the barrier the LDS belongs to the matrix role: 1653 -> 1566 cycles per round)
        if (pending) flush();
        __builtin_amdgcn_sched_barrier(0);
        const cplx x0 = Ms[rd[2]], x1 = Ms[rd[3]];         // pair 2 tau: (c, Re s), (Im s, a')
        const cplx y0 = Ms[rd[4]], y1 = Ms[rd[5]];         // pair 2 tau + 1
        if (diag) {
            cplx a; a.re = (R & 1) ? y1.im : x1.im; a.im = 0.0; cplx d; d.re = Ms[2 * PS + rd[6]].re; d.im = 0.0; cplx z; z.re = 0.0; z.im = 0.0;
            Ms[wr[0]] = a; Ms[wr[1]] = d; Ms[wr[2]] = z;
        }
        pr[0] = x0.re; pr[1] = x0.im; pr[2] = x1.re; pr[3] = y0.re; pr[4] = y0.im; pr[5] = y1.re; pending = true;
    };
    const int sweep = sweeps<7, 3>(ra, wa, -delta, red, tol2, test, round);
    if (pending) flush();
    if (sweep == FBX_JACOBI_MAX_SWEEPS && diag) { ev_a = Ms[0 * PS + ra[0]].re; ev_d = Ms[3 * PS + ra[1]].re; }
    FBX_BLOCK_SYNC();
    // results where the callers expect them: the eigenvalues on the diagonal of Ms, the eigenvectors in Vs (both buffers are
    // free: every thread's last read lies behind a barrier)
    if (diag) {
        cplx a; a.re = ev_a; a.im = 0.0; cplx d; d.re = ev_d; d.im = 0.0;
        Ms[0 * PS + sys_pos<N>(Kd, Kd, 0)] = a; Ms[3 * PS + sys_pos<N>(Kd, Kd, 1)] = d;
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int J = 2 * tau + b;
        Vs[0 * PS + sys_pos<N>(R, J, 0)] = v0p[b]; Vs[1 * PS + sys_pos<N>(R, J, 1)] = v0q[b];
        Vs[2 * PS + sys_pos<N>(R, J, 0)] = v1p[b]; Vs[3 * PS + sys_pos<N>(R, J, 1)] = v1q[b];
    }
    return sweep;
}
#else
__device__ __forceinline__ int matrix_role(cplx* Ms, int delta, int t, double* red, double tol2) {
    // rows i and 30 - i have 32 strictly-upper blocks between them: half-wavefront r = t / 32 takes those two rows (r < 15),
    // the last one the 16 blocks of row 15 -- runs of consecutive column pairs, as the conflict-free layout wants them
    const int r = t / NB, c = t % NB;
    const bool mrole = r < 15 || c < 16;
    const bool lower_row = c >= NB - 1 - r;                // the second row of the pair (r < 15 only)
    const int Iu = mrole ? (lower_row ? 30 - r : r) : 0;
    const int Ju = mrole ? (lower_row ? Iu + 1 + (c - (NB - 1 - r)) : r + 1 + c) : 1;
    static_assert(NUP == 15 * NB + 16, "496 strictly-upper blocks");
    // own block (planes b = 0 / 1), pivot block of the row pair, pivot block of the column pair; the four seats
    int ra[6], wa[4];
    unsigned sgm[4];                                        // sign bit to flip on the imaginary part of an entry stored mirrored
    ra[0] = sys_pos<N>(Iu, Ju, 0); ra[1] = sys_pos<N>(Iu, Ju, 1);
    ra[2] = sys_pos<N>(Iu, Iu, 0); ra[3] = sys_pos<N>(Iu, Iu, 1);
    ra[4] = sys_pos<N>(Ju, Ju, 0); ra[5] = sys_pos<N>(Ju, Ju, 1);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        bool cj;
        wa[e] = herm_store_pos(jacobi_seat<N>(2 * Iu + (e >> 1)), jacobi_seat<N>(2 * Ju + (e & 1)), cj);
        sgm[e] = cj ? 0x80000000u : 0u;
    }
    auto test = [&](const int (&rd)[6], double& o2, double& n2) __attribute__((always_inline)) {
        if (mrole) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const cplx v = Ms[e * PS + rd[e & 1]];
                o2 += 2.0 * (v.re * v.re + v.im * v.im);
            }
            n2 = o2;
        }
    };
    auto round = [&](const int (&rd)[6], const int (&wr)[4]) __attribute__((always_inline)) {
        if (mrole) {
            const double aI = Ms[0 * PS + rd[2]].re, dI_ = Ms[3 * PS + rd[3]].re; const cplx bI = Ms[1 * PS + rd[3]];
            const double aJ = Ms[0 * PS + rd[4]].re, dJ_ = Ms[3 * PS + rd[5]].re; const cplx bJ = Ms[1 * PS + rd[5]];
            cplx m00 = Ms[0 * PS + rd[0]], m01 = Ms[1 * PS + rd[1]];
            cplx m10 = Ms[2 * PS + rd[0]], m11 = Ms[3 * PS + rd[1]];
            const JRot rI = jacobi_rotation(aI, dI_, bI.re, bI.im);
            const JRot rJ = jacobi_rotation(aJ, dJ_, bJ.re, bJ.im);
            jacobi_apply_m(rI.c, rI.sr, rI.si, rJ.c, rJ.sr, rJ.si, m00, m01, m10, m11);
            m00.im = flip_sign(m00.im, sgm[0]); m01.im = flip_sign(m01.im, sgm[1]);
            m10.im = flip_sign(m10.im, sgm[2]); m11.im = flip_sign(m11.im, sgm[3]);
            Ms[wr[0]] = m00; Ms[wr[1]] = m01; Ms[wr[2]] = m10; Ms[wr[3]] = m11;
        }
    };
    return sweeps<6, 4>(ra, wa, delta, red, tol2, test, round);
}

__device__ __forceinline__ int vector_role(cplx* Ms, cplx* Vs, int delta, int tv, bool init_identity, double* red, double tol2) {
    const int I = tv / NB, J = tv % NB;
    const bool first = J == 0, last = J == NB - 1;
    const bool diag = I == (J & 15);                       // row pair I (J < 16) or I + 16 (J >= 16) is the column pair
    // pivot block of the column pair; seats of the rotated diagonal and of the annihilated entry
    int ra[2], wa[3];
    {
        const int sa = jacobi_seat<N>(2 * J), sd = jacobi_seat<N>(2 * J + 1);
        bool cj;
        ra[0] = sys_pos<N>(J, J, 0); ra[1] = sys_pos<N>(J, J, 1);
        wa[0] = 3 * (sa & 1) * PS + sys_pos<N>(sa >> 1, sa >> 1, sa & 1);
        wa[1] = 3 * (sd & 1) * PS + sys_pos<N>(sd >> 1, sd >> 1, sd & 1);
        wa[2] = herm_store_pos(sa, sd, cj);
    }
    // eigenvector blocks in registers: rows 2I, 2I + 1 (block 0) and 2I + 32, 2I + 33 (block 1); columns = slots 2J (p), 2J + 1 (q)
    cplx v0p[2], v0q[2], v1p[2], v1q[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int Ik = I + 16 * k;
        if (init_identity) {
            v0p[k].re = (Ik == J) ? 1.0 : 0.0; v0p[k].im = 0.0; v1q[k] = v0p[k];
            v0q[k].re = 0.0; v0q[k].im = 0.0; v1p[k] = v0q[k];
        } else {
            v0p[k] = Vs[0 * PS + sys_pos<N>(Ik, J, 0)]; v0q[k] = Vs[1 * PS + sys_pos<N>(Ik, J, 1)];
            v1p[k] = Vs[2 * PS + sys_pos<N>(Ik, J, 0)]; v1q[k] = Vs[3 * PS + sys_pos<N>(Ik, J, 1)];
        }
    }
    FBX_BLOCK_SYNC();                                       // Vs is the second matrix buffer from here on
    double ev_a = 0.0, ev_d = 0.0;                          // (diag threads) the pair's diagonal at the last convergence test
    auto test = [&](const int (&rd)[2], double& o2, double& n2) __attribute__((always_inline)) {
        if (diag) {
            ev_a = Ms[0 * PS + rd[0]].re; ev_d = Ms[3 * PS + rd[1]].re;
            const cplx b = Ms[1 * PS + rd[1]];
            o2 = 2.0 * (b.re * b.re + b.im * b.im);
            n2 = o2 + ev_a * ev_a + ev_d * ev_d;
        }
    };
    auto round = [&](const int (&rd)[2], const int (&wr)[3]) __attribute__((always_inline)) {
        const double aV = Ms[0 * PS + rd[0]].re, dV = Ms[3 * PS + rd[1]].re;
        const cplx bV = Ms[1 * PS + rd[1]];
        const JRot rV = jacobi_rotation(aV, dV, bV.re, bV.im);
        if (diag) {
            cplx a; a.re = rV.an; a.im = 0.0; cplx d; d.re = rV.dn; d.im = 0.0; cplx z; z.re = 0.0; z.im = 0.0;
            Ms[wr[0]] = a; Ms[wr[1]] = d; Ms[wr[2]] = z;
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            jacobi_apply_v(rV.c, rV.sr, rV.si, v0p[k], v0q[k], v1p[k], v1q[k]);
            seat_shift(v0p[k].re, v0q[k].re, first, last); seat_shift(v0p[k].im, v0q[k].im, first, last);
            seat_shift(v1p[k].re, v1q[k].re, first, last); seat_shift(v1p[k].im, v1q[k].im, first, last);
        }
    };
    const int sweep = sweeps<2, 3>(ra, wa, delta, red, tol2, test, round);
    if (sweep == FBX_JACOBI_MAX_SWEEPS && diag) { ev_a = Ms[0 * PS + ra[0]].re; ev_d = Ms[3 * PS + ra[1]].re; }
    FBX_BLOCK_SYNC();
    // results where the callers expect them: the eigenvalues on the diagonal of Ms, the eigenvectors in Vs (both buffers are
    // free: every thread's last read lies behind a barrier)
    if (diag) {
        cplx a; a.re = ev_a; a.im = 0.0; cplx d; d.re = ev_d; d.im = 0.0;
        Ms[0 * PS + sys_pos<N>(J, J, 0)] = a; Ms[3 * PS + sys_pos<N>(J, J, 1)] = d;
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int Ik = I + 16 * k;
        Vs[0 * PS + sys_pos<N>(Ik, J, 0)] = v0p[k]; Vs[1 * PS + sys_pos<N>(Ik, J, 1)] = v0q[k];
        Vs[2 * PS + sys_pos<N>(Ik, J, 0)] = v1p[k]; Vs[3 * PS + sys_pos<N>(Ik, J, 1)] = v1q[k];
    }
    return sweep;
}
#endif
}  // namespace eigh64

template <int NT = 1024>
__device__ int jacobi_eigh64(cplx* Ms, cplx* Vs, int t, bool init_identity, double* red, double tol2 = FBX_JACOBI_TOL2) {
    static_assert(NT == 1024, "sixteen wavefronts");
    const int delta = (int)(Vs - Ms);
    int sweep;
    // (a scalar branch: both sides execute the same sequence of workgroup barriers)
    if (__builtin_amdgcn_readfirstlane(t) < 512) {
        FBX_BLOCK_SYNC();                                   // (the eigenvector role's barrier behind its read of Vs)
        sweep = eigh64::matrix_role(Ms, delta, t, red, tol2);
        FBX_BLOCK_SYNC();                                   // (its barrier in front of the results)
    } else {
        sweep = eigh64::vector_role(Ms, Vs, delta, t - 512, init_identity, red, tol2);
    }
    FBX_BLOCK_SYNC();
    return sweep;
}

// the workgroup solver of the generic kernels (fbx_eigh, the into-chi conversions): the role-split solver above for 64 x 64 on sixteen
// wavefronts, jacobi_eigh_simple otherwise
template <int N, int NT>
__device__ __forceinline__ int jacobi_eigh_block(cplx* Ms, cplx* Vs, int t, bool init_identity, double* red) {
    if constexpr (N == 64 && NT == 1024) return jacobi_eigh64<1024>(Ms, Vs, t, init_identity, red);
    else return jacobi_eigh_simple<N, NT>(Ms, Vs, t, init_identity, red);
}

}  // namespace fbx
