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

// ---------------------------------------------------------------------------------------
// Systolic (Brent-Luk) Jacobi.  The matrix lives in "slot space": slot 2k / 2k+1 are the two
// members of pair k, so the pivots of a round are always the diagonal 2x2 blocks and lane
// (I, J) always rotates its own block with the rotations of pairs I (rows) and J (columns).
// After each round the slots are permuted by a fixed tournament permutation pi (player 0 stays,
// the others advance one seat), which is applied on the way through LDS: every lane writes its
// four updated entries to the seats pi assigns them and reads its new block back.  All LDS
// addresses are per-lane constants (no index arithmetic in the loop) and the element-major
// layout  Ms[e * LS + lane]  (e = 2a + b the position inside the 2x2 block) makes the b128
// accesses conflict-free.  After N-1 rounds (one sweep) pi^(N-1) = identity, so slot == index
// again whenever the convergence test runs.
template <int N>
__device__ __forceinline__ int jacobi_seat(int s) {       // pi: slot -> next slot
    constexpr int NB = N / 2;
    if (NB == 1) return s;
    const int k = s >> 1;
    if ((s & 1) == 0) {                                    // top row: t_k
        if (k == 0) return 0;
        if (k == NB - 1) return 2 * (NB - 1) + 1;          // t_last -> b_last
        return 2 * (k + 1);                                // t_k -> t_{k+1}
    }
    if (k == 0) return 2;                                  // b_0 -> t_1
    return 2 * (k - 1) + 1;                                // b_k -> b_{k-1}
}

// Plane stride of the element-major layout: plane e = 2a + b (position inside the 2 x 2 block) starts at
// e * sys_plane<N>() and holds one entry per block, block (I, J) at I * NB + J.  For N = 16 the planes
// are 66 entries apart instead of 64: the tournament permutation sends two of the eight blocks a
// ds_write_b128 lane group (8 consecutive lanes = one block row) owns to the same pair, into the two
// different b-planes; with a stride that is 2 (mod 8) sixteen-byte slots those two land on different
// banks, so the permuted writes of a Jacobi round are conflict-free (they were 2-way in every lane group:
// 36 % of the kernel's LDS cycles were bank-conflict cycles, profiles/r02).
template <int N>
__host__ __device__ constexpr int sys_plane() { return N == 16 ? 66 : (N / 2) * (N / 2); }
// number of cplx entries a matrix in this layout occupies
template <int N>
__host__ __device__ constexpr int sys_elems() { return 4 * sys_plane<N>(); }

// Position of block (I, K) inside plane e = 2a + b.  N <= 32: row-major, I * NB + K.  N = 64 (round 3): the column
// index is permuted inside its aligned group of eight so that EVERY b128 access of a Jacobi round is conflict-free --
// a b128 LDS instruction is served in groups of 8 consecutive lanes = 8 consecutive column pairs of one block row, and
// the tournament permutation sends such a group to the pairs {0, 2..8}, {9..16}, {17..24}, {25..31 top, 31 bottom}
// (top entries) resp. {1 top, 0..6 bottom}, {7..14}, {15..22}, {23..30} (bottom entries).  With the row-major layout
// pairs 0 and 8, and the top / bottom copies of pairs 31 and 1, share a bank group: 2-way conflicts in 3 of 8 lane
// groups, +37 % write passes (profiles/r02: 27 % of the LDS cycles of the 3-qubit kernel).  Writing a column pair K at
// residue rho(K) (mod 8) -- K mod 8, with residues 0 and 1 exchanged for K >= 8 -- and rotating the residues of the
// bottom planes (b = 1) by 2 makes every one of those sets, the aligned read groups and the pivot reads distinct mod 8
// (checked exhaustively by tests/test_host_logic.py::test_jacobi64_layout_is_conflict_free).
template <int N>
__host__ __device__ __forceinline__ constexpr int sys_pos(int I, int K, int e) {
    constexpr int NB = N / 2;
    if constexpr (N == 64) {
        const int r = K & 7;
        const int rho = (K >= 8 && r < 2) ? (r ^ 1) : r;
        return I * NB + ((K & ~7) | ((rho + 2 * (e & 1)) & 7));
    }
    return I * NB + K;
}
// element (r, c) of an N x N matrix in the element-major block layout
template <int N>
__device__ __forceinline__ int sys_index(int r, int c) {
    constexpr int PS = sys_plane<N>();
    const int e = (r & 1) * 2 + (c & 1);
    return e * PS + sys_pos<N>(r >> 1, c >> 1, e);
}
// linear entry index (0 .. N*N-1, plane-major as stored without padding) -> offset in the layout
template <int N>
__device__ __forceinline__ int sys_linear(int idx) {
    constexpr int LS = (N / 2) * (N / 2), PS = sys_plane<N>();
    return (idx / LS) * PS + (idx % LS);
}

template <int N>
__device__ __forceinline__ void sys_store(cplx* Ms, int lane, const Blk& v) {
    constexpr int NB = N / 2, LS = NB * NB, PS = sys_plane<N>();
    if (lane < LS) {
        const int I = lane / NB, J = lane % NB;
#pragma unroll
        for (int e = 0; e < 4; ++e) { cplx c; c.re = v.re[e]; c.im = v.im[e]; Ms[e * PS + sys_pos<N>(I, J, e)] = c; }
    }
}
template <int N>
__device__ __forceinline__ Blk sys_load(const cplx* Ms, int lane) {
    constexpr int NB = N / 2, LS = NB * NB, PS = sys_plane<N>();
    Blk v = blk_zero();
    if (lane < LS) {
        const int I = lane / NB, J = lane % NB;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const cplx c = Ms[e * PS + sys_pos<N>(I, J, e)]; v.re[e] = c.re; v.im[e] = c.im; }
    }
    return v;
}
// conjugate-transpose block: element (a, b) of block (I, J) <- conj of element (b, a) of block (J, I)
template <int N>
__device__ __forceinline__ Blk sys_load_adjoint(const cplx* Ms, int lane) {
    constexpr int NB = N / 2, LS = NB * NB, PS = sys_plane<N>();
    Blk v = blk_zero();
    if (lane < LS) {
        const int I = lane / NB, J = lane % NB;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int et = (e & 1) * 2 + (e >> 1);
            const cplx c = Ms[et * PS + sys_pos<N>(J, I, et)];
            v.re[e] = c.re; v.im[e] = -c.im;
        }
    }
    return v;
}

constexpr int FBX_JACOBI_MAX_SWEEPS = 40;
#define FBX_JACOBI_TOL2_VALUE 1e-26
constexpr double FBX_JACOBI_TOL2 = FBX_JACOBI_TOL2_VALUE;
#define FBX_BASIS_NORM_TOL 1e-9      // relative change of ||.||_F^2 under a stored basis above which it is discarded

// Rotation helpers ----------------------------------------------------------------------------
// Rotation R = [[c, s], [-conj(s), c]] that diagonalises the Hermitian pivot [[a, b], [conj(b), d]]
// (R^H . R): with delta = (d - a)/2, h = sqrt(delta^2 + |b|^2), q = |delta|/h:
//   c = sqrt((1 + q)/2),  s = sign(delta) * b / (2 h c) = f b.
// Two reciprocal square roots, no divisions, no branches (selects only); c^2 + |s|^2 = 1 to
// rounding.  Also returns the rotated diagonal  a' = a + u, d' = d - u,
// u = |s|^2 (d - a) - 2 c f |b|^2.
struct JRot { double c, sr, si, an, dn; };
__device__ __forceinline__ JRot jacobi_rotation_beta(double a, double d, double br, double bi, double beta);
__device__ __forceinline__ JRot jacobi_rotation(double a, double d, double br, double bi) {
    return jacobi_rotation_beta(a, d, br, bi, fma(br, br, bi * bi));
}
__device__ __forceinline__ JRot jacobi_rotation_beta(double a, double d, double br, double bi, double beta) {
    const double delta = 0.5 * (d - a);
    // |delta| + 1e-150 (absorbed by any |delta| > 1e-134, so every ordinary rotation is bit-identical to the form
    // above) keeps h2 > 0: for b == 0 the formulas then give q = 1, c = 1, s = f b = 0, u = 0 by themselves -- the
    // identity rotation without a single select.  (b so small that |b|^2 underflows while delta == 0 exactly:
    // c = 1, |s| <= 1e150 |b| < 1e-10, unitary to 1e-20.)
    const double ad = fabs(delta) + 1e-150;
    const double h2 = fma(ad, ad, beta);
    const double ih = fast_rsqrt(h2);
    const double x = fma(0.5 * ad, ih, 0.5);           // (1 + q) / 2 in [0.5, 1]
    const double ic = fast_rsqrt(x);
    const double f = (delta >= 0.0 ? 0.5 : -0.5) * ih * ic;
    const double fb = f * beta;
    const double u = fma(f * fb, 2.0 * delta, -2.0 * (x * ic) * fb);
    JRot r;
    r.c = x * ic; r.sr = f * br; r.si = f * bi; r.an = a + u; r.dn = d - u;
    return r;
}

// 2x2 block update  m <- R_I^H m R_J  and eigenvector columns  v <- v R_J
__device__ __forceinline__ void jacobi_apply_m(double cI, double sIr, double sIi, double cJ, double sJr,
                                               double sJi, cplx& m00, cplx& m01, cplx& m10, cplx& m11) {
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
}
// eigenvector accumulation: rows 2I, 2I+1 of V, column pair J:  v <- v R_J
__device__ __forceinline__ void jacobi_apply_v(double cJ, double sJr, double sJi, cplx& v0p, cplx& v0q,
                                               cplx& v1p, cplx& v1q) {
    cplx u0, w0, u1, w1;
    u0.re = cJ * v0p.re - (sJr * v0q.re + sJi * v0q.im);
    u0.im = cJ * v0p.im - (sJr * v0q.im - sJi * v0q.re);
    w0.re = cJ * v0q.re + (sJr * v0p.re - sJi * v0p.im);
    w0.im = cJ * v0q.im + (sJr * v0p.im + sJi * v0p.re);
    u1.re = cJ * v1p.re - (sJr * v1q.re + sJi * v1q.im);
    u1.im = cJ * v1p.im - (sJr * v1q.im - sJi * v1q.re);
    w1.re = cJ * v1q.re + (sJr * v1p.re - sJi * v1p.im);
    w1.im = cJ * v1q.im + (sJr * v1p.im + sJi * v1p.re);
    v0p = u0; v0q = w0; v1p = u1; v1q = w1;
}

// per-pair record {c, s, rotated diagonal}; small LDS scratch reserved for publishing rotations
struct __attribute__((aligned(16))) JRec { double c, sr, si, an, dn, pad; };

// In-LDS Hermitian eigendecomposition A = V diag(w) V^H.  On entry Ms holds the Hermitian matrix
// in the element-major layout (sys_store); on exit Ms is diagonal (eigenvalue k at
// sys_index(k, k)) and Vs holds the eigenvectors as columns in the same layout.  All 64 lanes of
// the wave must call; (N/2)^2 of them work.  Returns the number of sweeps.
//
// Every lane derives the two rotations it needs from the pivot blocks in LDS.
// Single-wavefront version with the eigenvector update software-pipelined one round behind the
// matrix update: v <- v R_J of round r does not feed the next pivots, so it is issued inside the
// dependent rsqrt chain of round r + 1, where the wave would otherwise idle.  LDS instructions of
// one wavefront execute in program order, so no barrier or wait separates the rounds.
//
// Round 5: the EIGENVECTOR block never touches LDS inside the sweeps.  The tournament permutation only moves eigenvector
// COLUMNS between neighbouring lanes of a block row -- the top column of pair J goes to pair J + 1 (t_0 stays, t_last becomes
// b_last), the bottom column to pair J - 1 (b_0 becomes t_1) -- and a block row is eight consecutive lanes, half a DPP row:
// two row shifts and three selects per 64-bit value (40 32-bit VALU instructions per round) replace four ds_write_b128 +
// four ds_read_b128.  A round was 160 cycles of LDS pipe, 104 of them its eight 16-byte stores (13 cycles per wave
// instruction on MI355X, a third of the read rate), shared by the four or eight wavefronts of a CU.  Measured in isolation
// (scripts/micro/run_jacobi_vdpp.sh, profiles/r05): 1180 -> 1017 cycles per round at one wavefront per SIMD, 1635 -> 1433 at
// two, 2595 -> 2096 at four.  The same rotations are applied to the same data: results are bit-identical to the LDS form
// (-DFBX_JACOBI_V_THROUGH_LDS keeps it for A/Bs).
// The solver exists in two forms selected by a TEMPLATE TAG at the call site (fbx_choi.hpp: ChoiLds<..., TWO_WORKERS>), not by a
// per-file macro: jacobi_eigh_wave<N, true> below (two workers per upper block; only the diagonal of Ms is meaningful on exit) and
// jacobi_eigh_wave<N, false> (every lane its full block).  Two names for two post-conditions -- no ODR hazard under -fgpu-rdc / LTO.
template <int N, bool TWO_WORKERS = false> struct JacobiWave;
// Round 5, second step (the one-wavefront-per-SIMD PGDB kernels, whose ChoiLds carries the tag): the
// HERMITIAN SYMMETRY of the work matrix, used without idling a lane.  In the form below this one lane (I, J) and lane (J, I) both
// compute their whole 2 x 2 block -- conjugate transposes of each other.
// Here only the upper block triangle exists; its block (I, J), I < J, has TWO workers: lane (I, J) computes column 0 of the updated
// block, lane (J, I) column 1 -- the same instruction stream (the second worker reads the block with its columns swapped and rotates
// with -conj(s): one sign flip), 24 fp64 instructions instead of 48 and two ds_write_b128 per lane instead of four.  Every entry is
// written once, to the seat the tournament permutation assigns it or, conjugated, to the mirrored seat, whichever lies in the stored
// triangle (as the 64 x 64 solver of fbx_eigh64.hpp does across wavefronts).  The diagonal lanes place the rotated diagonal (closed
// form an / dn of the rotation); the entry a rotation annihilates is never stored: the two workers that would read it read a zero
// cell instead (from the second round of a decomposition on).  On exit the diagonal of Ms holds the eigenvalues; the rest of Ms is
// work space.  Each entry an upper lane computed before is computed with the same expression now; what changes is that the lower
// triangle no longer evolves on its own (it agreed with the upper one to rounding), so results differ from the full-block form in the
// last bits.  Measured in isolation (scripts/micro/jacobi_h2.hpp): 1064 -> 937 cycles per round at one wavefront per SIMD with the
// max-ILP scheduling of fbx_pgdb.hip, 1429 -> 1396 at two with the default scheduling of fbx_pgdb_lean.hip.  In the kernels (same-box
// A/B): B = 1024 fixed-100 12.38 -> 12.10 ms, to convergence 10.25 -> 9.75 ms -- and 8192 items on the register-capped two-waves
// kernel 51.2 -> 53.7 ms (its dozen extra per-lane constants are spilled), which is why only fbx_pgdb.hip uses it.
// (Measured and dropped: an annealed slot assignment for the entries between rounds -- scripts/micro/h2_layout_anneal.py: 21 / 25
// LDS-array cycles per round for the two stores / four loads instead of the 60 / 45 the mirrored seats cost in the plane layout built
// for full blocks, i.e. the 49.8 % bank conflicts of profiles/r05 gone -- changes nothing in isolation (940 vs 941 cycles per round:
// the conflicts are not on the dependent chain) and costs the kernel its table look-ups: B = 1024 fixed-100 12.17 -> 12.52 ms.)
template <int N>
struct JacobiWave<N, true> {
static __device__ int run(cplx* __restrict__ Ms, cplx* __restrict__ Vs, int lane, bool init_identity,
                          double expect_n2 = -1.0, double tol2 = FBX_JACOBI_TOL2) {
    constexpr int NB = N / 2, LS = NB * NB, PS = sys_plane<N>();
    static_assert(N == 16 && LS == 64, "every lane of the wavefront owns one eigenvector block");
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
};
template <int N>
struct JacobiWave<N, false> {
static __device__ int run(cplx* __restrict__ Ms, cplx* __restrict__ Vs, int lane, bool init_identity,
                          double expect_n2 = -1.0, double tol2 = FBX_JACOBI_TOL2) {
    constexpr int NB = N / 2, LS = NB * NB, PS = sys_plane<N>();
    static_assert(N == 16 && LS == 64, "every lane of the wavefront owns one 2x2 block; a block row is half a DPP row");
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
    const bool first = J == 0, last = J == NB - 1;
    // eigenvector block in registers, in the matrix's layout: v0p = V[2I][t_J], v0q = V[2I][b_J], v1p = V[2I+1][t_J], v1q = V[2I+1][b_J]
    cplx v0p, v0q, v1p, v1q;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        cplx v;
        if (init_identity) { v.re = (2 * I + (e >> 1) == 2 * J + (e & 1)) ? 1.0 : 0.0; v.im = 0.0; }
        else v = Vs[e * PS + me];
        if (e == 0) v0p = v; else if (e == 1) v0q = v; else if (e == 2) v1p = v; else v1q = v;
    }
    // the tournament permutation of the columns (jacobi_seat), in registers
    auto permute = [&](double& p, double& q) __attribute__((always_inline)) {
        const double right = first ? q : p;                       // what this lane hands to its right neighbour (b_0 -> t_1)
        const double from_left = dpp_shift<0x111>(right);          // row_shr:1
        const double from_right = dpp_shift<0x101>(q);             // row_shl:1
        const double pn = first ? p : from_left;                   // t_0 stays
        const double qn = last ? p : from_right;                   // t_last -> b_last
        p = pn; q = qn;
    };
    double pc = 1.0, psr = 0.0, psi = 0.0;          // rotation whose eigenvector update is still pending
    bool pending = false;
    int sweep = 0;
    double n2 = 0.0;                                // ||A||_F^2: invariant under the rotations, reduced once
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
            // expect_n2 >= 0: the matrix was brought into a basis loaded from memory; a unitary similarity
            // keeps ||.||_F^2 (here to FBX_BASIS_NORM_TOL), a damaged basis does not -> tell the caller (-1)
            if (sweep == 0 && expect_n2 >= 0.0 && !(fabs(n2 - expect_n2) <= FBX_BASIS_NORM_TOL * expect_n2)) return -1;
            if (!(o2 > tol2 * n2)) break;
        }
        for (int r = 0; r < N - 1; ++r) {
            const double aJ = Ms[0 * PS + dJ].re, dJ_ = Ms[3 * PS + dJ].re;
            const cplx bJ = Ms[1 * PS + dJ];
            cplx m00 = Ms[0 * PS + me], m01 = Ms[1 * PS + me];
            cplx m10 = Ms[2 * PS + me], m11 = Ms[3 * PS + me];
            // the previous round's eigenvector update and its seat permutation -- independent of the chain below
            if (pending) {
                jacobi_apply_v(pc, psr, psi, v0p, v0q, v1p, v1q);
                permute(v0p.re, v0q.re); permute(v0p.im, v0q.im); permute(v1p.re, v1q.re); permute(v1p.im, v1q.im);
            }
            const JRot rJ = jacobi_rotation(aJ, dJ_, bJ.re, bJ.im);
            JRot rI;
            rI.c = __shfl(rJ.c, src_lane); rI.sr = __shfl(rJ.sr, src_lane); rI.si = __shfl(rJ.si, src_lane);
            jacobi_apply_m(rI.c, rI.sr, rI.si, rJ.c, rJ.sr, rJ.si, m00, m01, m10, m11);
            if (I == J) {   // the annihilated pair: exact zeros, real diagonal (see jacobi_eigh_simple)
                m01.re = m01.im = 0.0; m10.re = m10.im = 0.0;
                m00.im = 0.0; m11.im = 0.0;
            }
            Ms[wm[0]] = m00; Ms[wm[1]] = m01; Ms[wm[2]] = m10; Ms[wm[3]] = m11;
            pc = rJ.c; psr = rJ.sr; psi = rJ.si; pending = true;
        }
    }
    if (pending) {                                  // flush the last pending update
        jacobi_apply_v(pc, psr, psi, v0p, v0q, v1p, v1q);
        permute(v0p.re, v0q.re); permute(v0p.im, v0q.im); permute(v1p.re, v1q.re); permute(v1p.im, v1q.im);
    }
    // (an identity start always stores; a warm start that needed no rotation leaves the caller's basis where it is)
    if (init_identity || pending) { Vs[0 * PS + me] = v0p; Vs[1 * PS + me] = v0q; Vs[2 * PS + me] = v1p; Vs[3 * PS + me] = v1q; }
    FBX_WAVE_SYNC();
    return sweep;
}
};
template <int N, bool TWO_WORKERS = false>
__device__ int jacobi_eigh_wave(cplx* __restrict__ Ms, cplx* __restrict__ Vs, int lane, bool init_identity,
                                                double expect_n2 = -1.0, double tol2 = FBX_JACOBI_TOL2) {
    return JacobiWave<N, TWO_WORKERS>::run(Ms, Vs, lane, init_identity, expect_n2, tol2);
}

template <int N, int NT = 64>
__device__ int jacobi_eigh_simple(cplx* Ms, cplx* Vs, int lane, bool init_identity = true,
                                  double* red = nullptr, double tol2 = FBX_JACOBI_TOL2) {
    constexpr int NB = N / 2, LS = NB * NB, PS = sys_plane<N>();
    static_assert(LS <= NT, "one lane per 2x2 block");
    if constexpr (NT <= 64 && N == 16) { (void)red; return jacobi_eigh_wave<N>(Ms, Vs, lane, init_identity, -1.0, tol2); }
    const bool act = lane < LS;
    const int I = act ? lane / NB : 0, J = act ? lane % NB : 0;
    // own block of plane e: me0 for the planes with b = 0 (e = 0, 2), me1 for b = 1 (they differ only for N = 64)
    const int me0 = sys_pos<N>(I, J, 0), me1 = sys_pos<N>(I, J, 1);
    int wm[4], wv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int sa = jacobi_seat<N>(2 * I + (e >> 1)), sb = jacobi_seat<N>(2 * J + (e & 1));
        const int pm = (sa & 1) * 2 + (sb & 1), pv = (e >> 1) * 2 + (sb & 1);
        wm[e] = pm * PS + sys_pos<N>(sa >> 1, sb >> 1, pm);
        wv[e] = pv * PS + sys_pos<N>(I, sb >> 1, pv);
    }
    const int dI0 = sys_pos<N>(I, I, 0), dI1 = sys_pos<N>(I, I, 1);    // pivot block of the row pair (two-chain variant only)
    const int dJ0 = sys_pos<N>(J, J, 0), dJ1 = sys_pos<N>(J, J, 1);    // pivot block of the column pair: planes 0 / 1, 3
    const int src_lane = (lane & 63) - J + I;              // lane (I, I): same row, NB | 64 keeps rows inside a wave
    (void)dI0; (void)dI1;
    if (act && init_identity) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            cplx v; v.re = (2 * I + (e >> 1) == 2 * J + (e & 1)) ? 1.0 : 0.0; v.im = 0.0;
            Vs[e * PS + ((e & 1) ? me1 : me0)] = v;
        }
    }
    if constexpr (NT > 64) FBX_BLOCK_SYNC(); else FBX_WAVE_SYNC();
    int sweep = 0;
    for (; sweep < FBX_JACOBI_MAX_SWEEPS; ++sweep) {
        {
            double o2 = 0.0, n2 = 0.0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const cplx v = Ms[e * PS + ((e & 1) ? me1 : me0)];
                const double a2 = v.re * v.re + v.im * v.im;
                n2 += a2;
                if (!(I == J && (e == 0 || e == 3))) o2 += a2;
            }
            if (!act) { o2 = 0.0; n2 = 0.0; }
            block_sum2<NT>(o2, n2, red);
            o2 = uniform(o2); n2 = uniform(n2);
            if (!(o2 > tol2 * n2)) break;
        }
        for (int r = 0; r < N - 1; ++r) {
            const double aJ = Ms[0 * PS + dJ0].re, dJ_ = Ms[3 * PS + dJ1].re;
            const cplx bJ = Ms[1 * PS + dJ1];
            cplx m00 = Ms[0 * PS + me0], m01 = Ms[1 * PS + me1];
            cplx m10 = Ms[2 * PS + me0], m11 = Ms[3 * PS + me1];
            cplx v0p = Vs[0 * PS + me0], v0q = Vs[1 * PS + me1];
            cplx v1p = Vs[2 * PS + me0], v1q = Vs[3 * PS + me1];
            // everything read before anyone overwrites it.  A single wavefront needs no barrier and no
            // wait here or after the writes: its LDS instructions execute in program order, so the
            // reads above see the previous round and the next round's reads see the writes below.
            if constexpr (NT > 64) FBX_BLOCK_SYNC();
            const JRot rJ = jacobi_rotation(aJ, dJ_, bJ.re, bJ.im);
            // the row rotation (pair I) is the column rotation of the lane (I, I) of this block row,
            // which sits in the same wavefront: fetch it instead of computing a second chain
            JRot rI;
            rI.c = __shfl(rJ.c, src_lane); rI.sr = __shfl(rJ.sr, src_lane); rI.si = __shfl(rJ.si, src_lane);
            jacobi_apply_m(rI.c, rI.sr, rI.si, rJ.c, rJ.sr, rJ.si, m00, m01, m10, m11);
            jacobi_apply_v(rJ.c, rJ.sr, rJ.si, v0p, v0q, v1p, v1q);
            if (I == J) {   // the annihilated pair: exact zeros, real diagonal
                m01.re = m01.im = 0.0; m10.re = m10.im = 0.0;
                // keep the diagonal the rotation itself produced (consistent with the rest of the
                // rows / columns to rounding); only its exactly-zero imaginary part is enforced
                m00.im = 0.0; m11.im = 0.0;
            }
            if (act) {
                Ms[wm[0]] = m00; Ms[wm[1]] = m01; Ms[wm[2]] = m10; Ms[wm[3]] = m11;
                Vs[wv[0]] = v0p; Vs[wv[1]] = v0q; Vs[wv[2]] = v1p; Vs[wv[3]] = v1q;
            }
            if constexpr (NT > 64) FBX_BLOCK_SYNC();
        }
        if constexpr (NT <= 64) FBX_WAVE_SYNC();
    }
    return sweep;
}

// Warm start: replace the matrix in Ms by V^H Ms V for the (unitary) V already in Vs, so that the
// sweeps start from a nearly diagonal matrix when V diagonalises a nearby matrix.  `Ts` is 4*LS
// cplx of scratch.  All three arrays use the element-major block layout.
template <int N>
__device__ void jacobi_rotate_into_basis(cplx* Ms, const cplx* Vs, cplx* Ts, int lane) {
    constexpr int NB = N / 2, LS = NB * NB, PS = sys_plane<N>();
    const bool act = lane < LS;
    const int I = act ? lane / NB : 0, J = act ? lane % NB : 0;
    cplx t[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { t[e].re = 0.0; t[e].im = 0.0; }
    for (int kb = 0; kb < NB; ++kb) {
#pragma unroll
        for (int ke = 0; ke < 2; ++ke) {
            const cplx h0 = Ms[(0 + ke) * PS + sys_pos<N>(I, kb, ke)], h1 = Ms[(2 + ke) * PS + sys_pos<N>(I, kb, ke)];   // H[2I+a][k]
            const cplx v0 = Vs[(ke * 2 + 0) * PS + sys_pos<N>(kb, J, 0)], v1 = Vs[(ke * 2 + 1) * PS + sys_pos<N>(kb, J, 1)]; // V[k][2J+b]
            t[0].re += h0.re * v0.re - h0.im * v0.im; t[0].im += h0.re * v0.im + h0.im * v0.re;
            t[1].re += h0.re * v1.re - h0.im * v1.im; t[1].im += h0.re * v1.im + h0.im * v1.re;
            t[2].re += h1.re * v0.re - h1.im * v0.im; t[2].im += h1.re * v0.im + h1.im * v0.re;
            t[3].re += h1.re * v1.re - h1.im * v1.im; t[3].im += h1.re * v1.im + h1.im * v1.re;
        }
    }
    if (act) {
#pragma unroll
        for (int e = 0; e < 4; ++e) Ts[e * PS + sys_pos<N>(I, J, e)] = t[e];
    }
    FBX_WAVE_SYNC();
#pragma unroll
    for (int e = 0; e < 4; ++e) { t[e].re = 0.0; t[e].im = 0.0; }
    for (int kb = 0; kb < NB; ++kb) {
#pragma unroll
        for (int ke = 0; ke < 2; ++ke) {
            const cplx u0 = Vs[(ke * 2 + 0) * PS + sys_pos<N>(kb, I, 0)], u1 = Vs[(ke * 2 + 1) * PS + sys_pos<N>(kb, I, 1)]; // V[k][2I+a]
            const cplx w0 = Ts[(ke * 2 + 0) * PS + sys_pos<N>(kb, J, 0)], w1 = Ts[(ke * 2 + 1) * PS + sys_pos<N>(kb, J, 1)]; // T[k][2J+b]
            // conj(u) * w
            t[0].re += u0.re * w0.re + u0.im * w0.im; t[0].im += u0.re * w0.im - u0.im * w0.re;
            t[1].re += u0.re * w1.re + u0.im * w1.im; t[1].im += u0.re * w1.im - u0.im * w1.re;
            t[2].re += u1.re * w0.re + u1.im * w0.im; t[2].im += u1.re * w0.im - u1.im * w0.re;
            t[3].re += u1.re * w1.re + u1.im * w1.im; t[3].im += u1.re * w1.im - u1.im * w1.re;
        }
    }
    FBX_WAVE_SYNC();
    if (act) {
        if (I == J) { t[0].im = 0.0; t[3].im = 0.0; }
#pragma unroll
        for (int e = 0; e < 4; ++e) Ms[e * PS + sys_pos<N>(I, J, e)] = t[e];
    }
    FBX_WAVE_SYNC();
}

// The same warm-start transform for N = 16 on one full wavefront, on the fp64 matrix cores.  The two
// 16 x 16 x 16 complex products are four real v_mfma_f64_16x16x4_f64 chains each (lane l feeds
// A[m = l & 15][k = l >> 4] and B[k = l >> 4][n = l & 15]; accumulator r is D[(l >> 4) + 4 r][l & 15]).
// With T = H V accumulated first, the operands of M' = V^H T are already in place: conj(V[k][i]) is
// the V element loaded as B operand of the first product, and T[4 k4 + (l >> 4)][l & 15] is
// accumulator k4 of this lane.  8 LDS loads + 4 stores per lane instead of 128 + 8 for the
// register-blocked VALU form, which is LDS-bandwidth bound with four wavefronts per CU.
typedef double fbx_v4d __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void jacobi_rotate_into_basis_mfma16(cplx* Ms, const cplx* Vs, int lane) {
    constexpr int NB = 8, PS = sys_plane<16>();
    const int c = lane & 15, g = lane >> 4;
    auto at = [](int r, int cc) { return ((r & 1) * 2 + (cc & 1)) * PS + (r >> 1) * NB + (cc >> 1); };
    cplx h[4], v[4];
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
        const int k = 4 * k4 + g;
        h[k4] = Ms[at(c, k)];            // H[i = c][k]
        v[k4] = Vs[at(k, c)];            // V[k][j = c]  (= V[k][i = c] of the second product)
    }
    fbx_v4d tre = {0.0, 0.0, 0.0, 0.0}, tim = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {     // T = H V
        tre = __builtin_amdgcn_mfma_f64_16x16x4f64(h[k4].re, v[k4].re, tre, 0, 0, 0);
        tim = __builtin_amdgcn_mfma_f64_16x16x4f64(h[k4].re, v[k4].im, tim, 0, 0, 0);
        tre = __builtin_amdgcn_mfma_f64_16x16x4f64(-h[k4].im, v[k4].im, tre, 0, 0, 0);
        tim = __builtin_amdgcn_mfma_f64_16x16x4f64(h[k4].im, v[k4].re, tim, 0, 0, 0);
    }
    fbx_v4d mre = {0.0, 0.0, 0.0, 0.0}, mim = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {     // M' = V^H T: conj(v) * t
        mre = __builtin_amdgcn_mfma_f64_16x16x4f64(v[k4].re, tre[k4], mre, 0, 0, 0);
        mim = __builtin_amdgcn_mfma_f64_16x16x4f64(v[k4].re, tim[k4], mim, 0, 0, 0);
        mre = __builtin_amdgcn_mfma_f64_16x16x4f64(v[k4].im, tim[k4], mre, 0, 0, 0);
        mim = __builtin_amdgcn_mfma_f64_16x16x4f64(-v[k4].im, tre[k4], mim, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = g + 4 * r;
        cplx o; o.re = mre[r]; o.im = row == c ? 0.0 : mim[r];
        Ms[at(row, c)] = o;
    }
    FBX_WAVE_SYNC();
}

template <int N, int NT = 64>
__device__ __forceinline__ int jacobi_eigh_lds(cplx* Ms, cplx* Vs, JRec* rec, int lane,
                                               bool init_identity = true, double* red = nullptr) {
    (void)rec;      // (a variant that publishes next-round rotations through `rec` measured no faster)
    return jacobi_eigh_simple<N, NT>(Ms, Vs, lane, init_identity, red);
}

// block (I, J) of sum_k lam[k] v_k v_k^H for the eigenvectors in Vs (element-major layout);
// terms with lam[k] == 0 are skipped (wave-uniform branch).
template <int N>
__device__ __forceinline__ Blk reconstruct_blk(const cplx* Vs, const double* lam, int lane) {
    constexpr int NB = N / 2, LS = NB * NB, PS = sys_plane<N>();
    Blk out = blk_zero();
    const bool act = lane < LS;
    const int I = act ? lane / NB : 0, J = act ? lane % NB : 0;
    // lane k of every wavefront keeps lam[k]; the non-zero ones are walked through a ballot mask and
    // v_readlane, so the loop has no LDS load + branch on its critical path
    const int wl = lane & 63;
    const double mine = wl < N ? lam[wl] : 0.0;
    unsigned long long todo = __ballot(mine != 0.0);
    while (todo) {
        const int k = __builtin_ctzll(todo);
        todo &= todo - 1;
        const double l = readlane_f64(mine, k);
        const int kb = k >> 1, ke = k & 1;
        const int rk = sys_pos<N>(I, kb, ke), ck = sys_pos<N>(J, kb, ke);      // column k of the block rows I and J
        const cplx r0 = Vs[(0 + ke) * PS + rk], r1 = Vs[(2 + ke) * PS + rk];
        const cplx c0 = Vs[(0 + ke) * PS + ck], c1 = Vs[(2 + ke) * PS + ck];
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
