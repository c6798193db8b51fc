// fbx_pgdb3.hip -- 3-qubit (64 x 64 Choi) projected-gradient process tomography
// (BASELINE config 4).  Same algorithm and the same device routines as the 2-qubit kernel
// (fbx_pgdb.hip), re-mapped to ONE 1024-THREAD WORKGROUP PER RECONSTRUCTION: the 64 x 64 matrices
// are held as one 2x2 block per thread on a 32 x 32 thread grid, the systolic Jacobi runs with all
// 16 wavefronts (63 rounds per sweep, two workgroup barriers per round), and the 160 KiB of LDS are
// used to the byte:
//     [ Ms 64 KiB | Vs 64 KiB (aliased by the row-major staging matrix Mw) | R 32 KiB ]
// with the prediction tables T[s][i] / the gradient accumulator W[i][s] (up to 108 KiB for the
// Pauli in-basis) overlaying Ms + Vs between projections, and the small Dykstra scratch overlaying R.
// The Bloch matrix C (D x S doubles, 108 KiB for the Pauli in-basis) is read from L2.
//
// Reference: tomography.py:542-633, operator_tools/project_superoperators.py:19-144.
// no thread of these kernels reads global memory another thread of the same launch wrote (basis store and parked state
// are per thread, tables go through LDS): barriers order LDS only and leave global traffic in flight (fbx_common.hpp)
#ifndef FBX_FULL_BARRIERS          // (-DFBX_FULL_BARRIERS: every barrier a full __syncthreads() -- libfbx_fullbar.so, the reference build of
#define FBX_LDS_ONLY_BARRIERS      //  tests/test_barriers_gpu.py, which must reproduce this one bit for bit)
#endif
#include "fbx_choi.hpp"
#include "fbx_eigh64.hpp"
#include <cstdlib>

namespace fbx {

#ifdef FBX_DIAGNOSTICS
extern long long* g_phase_out;      // fbx_pgdb.hip (diagnostics builds only)
#define FBX_PHASE_OUT3(b0) (g_phase_out ? g_phase_out + (b0) * 8 : nullptr)
#else
#define FBX_PHASE_OUT3(b0) ((long long*)nullptr)
#endif
#define FBX3_BASIS_STEP 3e-2           // outer step below which Dykstra iteration j starts from the previous call's basis j.  The 2-qubit kernel's
                                       // 1e-3 (FBX_BASIS_STEP) is too timid here: same-box A/B (scripts/ab_time3.py, 256 items) 331.6 ms at
                                       // 1e-3 / 3e-2, 324.4 at 1e-2 / 1e-1, 317.2 at 3e-2 / 3e-1, 317.1 at 1e-1 / 1 -- a starting basis only
                                       // changes how many sweeps a decomposition needs, never the tolerance it runs to
#define FBX3_BASIS_WRITE_STEP 3e-1     // outer step below which every basis is written back (10 x the threshold above, as in the 2-qubit kernel)
#define FBX3_BASIS_CHAIN_SWEEPS 216   // as FBX_BASIS_CHAIN_SWEEPS of the 2-qubit kernel (fbx_pgdb.hip)

namespace p3 {
constexpr int NQ = 3, d = 8, D = 64, NB = 32, NT = 1024, LD = 64, LDs = d + 1;
constexpr double EPS = 1e-6, GAMMA = 0.3, STOP = 1e-10, ALPHA_MIN = 1e-15;

struct Lds {
    cplx* Ms;        // [4096] Jacobi work matrix (element-major)
    cplx* Vs;        // [4096] eigenvectors
    cplx* Mw;        // = Vs: row-major 64 x 64 staging (partial trace, Pauli butterflies)
    double* T;       // = Ms..: prediction table [S][64] / gradient accumulator W[64][S]
    double* Rt;      // [4096] Pauli coefficients, TRANSPOSED: Rt[j * 64 + i] = R[i][j]
    // overlay on Rt (alive only while R is dead):
    cplx* pt; cplx* pts; cplx* ptV; cplx* ptold; double* lam; double* red;
    PhaseClock* pc;  // diagnostics (-DFBX_PHASE_TIMERS)
    int terms = 0;   // work accounting: eigenvalue terms rebuilt by the CP projections
    double jtol2 = FBX_JACOBI_TOL2;   // eigensolver tolerance of the CP projections (see fbx_pgdb.hip, FBX_JTOL_REL)
    __device__ void carve(char* p) {
        Ms = (cplx*)p; Vs = Ms + D * D; Mw = Vs; T = (double*)p;
        Rt = (double*)(p + 2 * sizeof(cplx) * D * D);
        char* q = (char*)Rt;
        pt = (cplx*)q; q += sizeof(cplx) * d * LDs;
        pts = (cplx*)q; q += sizeof(cplx) * d * d;
        ptV = (cplx*)q; q += sizeof(cplx) * d * d;
        ptold = (cplx*)q; q += sizeof(cplx) * d * LDs;
        lam = (double*)q; q += sizeof(double) * D;
        red = (double*)q;
    }
    static constexpr size_t bytes() { return 2 * sizeof(cplx) * D * D + sizeof(double) * D * D; }
};

__device__ __forceinline__ double bsum(double v, Lds& L) { return block_sum<NT>(v, L.red); }
// K <= 8 sums over the workgroup with ONE barrier pair instead of K (every thread gets all totals).  Second stage (round 5): lane
// 16 k + w of every wavefront reads the partial sum of wavefront w for value k (two loads per lane for K > 4) and the sixteen
// partials are added by four DPP steps inside the row -- instead of 16 K broadcast loads and additions per thread (128 LDS
// loads per thread and call for the Dykstra stopping functional: 17 k cycles of the ~420 k of a Dykstra iteration).
template <int K>
__device__ __forceinline__ void bsum_multi(double (&v)[K], Lds& L) {
    static_assert((K <= 4 || K == 8) && NT / 64 == 16, "sixteen wavefronts; the second-stage load of values 4 .. K - 1 indexes with a mask: K x 16 must be a power of two");
    const int lane = threadIdx.x & 63;
    double mine = 0.0;                                  // lane k < K publishes value k of this wavefront
#pragma unroll
    for (int k = 0; k < K; ++k) { const double w = wave_sum(v[k]); mine = lane == k ? w : mine; }
    FBX_BLOCK_SYNC();                                   // earlier readers of `red` are done
    if (lane < K) L.red[lane * (NT / 64) + (threadIdx.x >> 6)] = mine;
    FBX_BLOCK_SYNC();
    double a = L.red[lane], b = K > 4 ? L.red[(64 + lane) & (K * (NT / 64) - 1)] : 0.0;
    a += dpp_permute<0xB1>(a); a += dpp_permute<0x4E>(a); a += dpp_permute<0x141>(a); a += dpp_permute<0x140>(a);
    if constexpr (K > 4) { b += dpp_permute<0xB1>(b); b += dpp_permute<0x4E>(b); b += dpp_permute<0x141>(b); b += dpp_permute<0x140>(b); }
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = k < 4 ? readlane_f64(a, 16 * k) : readlane_f64(b, 16 * (k - 4));
}

// ---- warm start: Ms <- V^H Ms V for the eigenvectors V of the previous projection (still in Vs).
// LDS is full, but the intermediate product T = Ms V can take the place of Ms itself once every
// thread holds its block of it (a first version sent T through an L2 scratch: 1 MB of L2 reads per
// decomposition).  `Tg` only says that warm starts are enabled.
__device__ void rotate_into_basis(Lds& L, cplx* Tg, int t) {
    (void)Tg;
    const int I = t / NB, J = t % NB;
    cplx acc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { acc[e].re = 0.0; acc[e].im = 0.0; }
    for (int kb = 0; kb < NB; ++kb) {
#pragma unroll
        for (int ke = 0; ke < 2; ++ke) {
            const cplx h0 = L.Ms[sys_index<D>(2 * I, 2 * kb + ke)], h1 = L.Ms[sys_index<D>(2 * I + 1, 2 * kb + ke)];
            const cplx v0 = L.Vs[sys_index<D>(2 * kb + ke, 2 * J)], v1 = L.Vs[sys_index<D>(2 * kb + ke, 2 * J + 1)];
            acc[0].re += h0.re * v0.re - h0.im * v0.im; acc[0].im += h0.re * v0.im + h0.im * v0.re;
            acc[1].re += h0.re * v1.re - h0.im * v1.im; acc[1].im += h0.re * v1.im + h0.im * v1.re;
            acc[2].re += h1.re * v0.re - h1.im * v0.im; acc[2].im += h1.re * v0.im + h1.im * v0.re;
            acc[3].re += h1.re * v1.re - h1.im * v1.im; acc[3].im += h1.re * v1.im + h1.im * v1.re;
        }
    }
    // T = Ms V replaces Ms in place: H is dead once every thread has its block of the product
    FBX_BLOCK_SYNC();
#pragma unroll
    for (int e = 0; e < 4; ++e) L.Ms[sys_index<D>(2 * I + (e >> 1), 2 * J + (e & 1))] = acc[e];
    FBX_BLOCK_SYNC();
#pragma unroll
    for (int e = 0; e < 4; ++e) { acc[e].re = 0.0; acc[e].im = 0.0; }
    for (int kb = 0; kb < NB; ++kb) {
#pragma unroll
        for (int ke = 0; ke < 2; ++ke) {
            const cplx u0 = L.Vs[sys_index<D>(2 * kb + ke, 2 * I)], u1 = L.Vs[sys_index<D>(2 * kb + ke, 2 * I + 1)];
            const cplx w0 = L.Ms[sys_index<D>(2 * kb + ke, 2 * J)], w1 = L.Ms[sys_index<D>(2 * kb + ke, 2 * J + 1)];
            acc[0].re += u0.re * w0.re + u0.im * w0.im; acc[0].im += u0.re * w0.im - u0.im * w0.re;
            acc[1].re += u0.re * w1.re + u0.im * w1.im; acc[1].im += u0.re * w1.im - u0.im * w1.re;
            acc[2].re += u1.re * w0.re + u1.im * w0.im; acc[2].im += u1.re * w0.im - u1.im * w0.re;
            acc[3].re += u1.re * w1.re + u1.im * w1.im; acc[3].im += u1.re * w1.im - u1.im * w1.re;
        }
    }
    FBX_BLOCK_SYNC();                                   // every thread is done reading Ms
    if (I == J) { acc[0].im = 0.0; acc[3].im = 0.0; }
#pragma unroll
    for (int e = 0; e < 4; ++e) L.Ms[sys_index<D>(2 * I + (e >> 1), 2 * J + (e & 1))] = acc[e];
    FBX_BLOCK_SYNC();
}

// The same change of basis on the fp64 matrix cores: wavefront w owns the 16 x 16 output tile (w >> 2, w & 3) of
// each of the two 64^3 complex products, 16 k-steps of four v_mfma_f64_16x16x4_f64 (re.re, -im.im, re.im, im.re).
// Lane l feeds A[m = l & 15][k = l >> 4] and B[k = l >> 4][n = l & 15]; accumulator r is D[(l >> 4) + 4 r][l & 15].
// Both A operands are read TRANSPOSED so that the 16 lanes of a group walk along a row of the layout:
// H[m][k] = conj(H[k][m]) (H is exactly Hermitian: it was just symmetrised) and (V^H)[m][k] = conj(V[k][m]).
// 2 x 16 x 2 b128 loads and 8 stores per lane instead of 2 x 64 x 4 loads; the products themselves take
// 128 x 32 cycles per wavefront (the VALU form: 68.7 k cycles per decomposition, this one: see DESIGN.md 4.4).
__device__ void rotate_into_basis_mfma(Lds& L, int t) {
    typedef double v4d __attribute__((ext_vector_type(4)));
    t = opaque(t);
    const int w = t >> 6, l = t & 63, tm = w >> 2, tn = w & 3, g = l >> 4, c = l & 15;
    const int colA = 16 * tm + c, colB = 16 * tn + c;
    v4d tre = {0.0, 0.0, 0.0, 0.0}, tim = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (int ks = 0; ks < 16; ++ks) {                   // T = H V
        const int k = 4 * ks + g;
        const cplx h = L.Ms[sys_index<D>(k, colA)];     // conj(H[colA][k])
        const cplx v = L.Vs[sys_index<D>(k, colB)];     // V[k][colB]
        tre = __builtin_amdgcn_mfma_f64_16x16x4f64(h.re, v.re, tre, 0, 0, 0);
        tim = __builtin_amdgcn_mfma_f64_16x16x4f64(h.re, v.im, tim, 0, 0, 0);
        tre = __builtin_amdgcn_mfma_f64_16x16x4f64(h.im, v.im, tre, 0, 0, 0);      // -(-im) im
        tim = __builtin_amdgcn_mfma_f64_16x16x4f64(-h.im, v.re, tim, 0, 0, 0);
    }
    FBX_BLOCK_SYNC();                                    // H is dead: T takes its place
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        cplx o; o.re = tre[r]; o.im = tim[r];
        L.Ms[sys_index<D>(16 * tm + g + 4 * r, colB)] = o;
    }
    FBX_BLOCK_SYNC();
    // M' = V^H T is Hermitian and the eigensolver reads its upper block triangle only: the ten tiles (tm2 <= tn2) go to wavefronts
    // 0-9 -- three tiles on two of the matrix-core pipes, two on the others, instead of four on each.  The six tiles below the
    // diagonal keep T (proj_cp's norm check counts the upper tiles twice instead).
    const bool upper = w < 10;
    const int tm2 = w < 4 ? 0 : w < 7 ? 1 : w < 9 ? 2 : 3, tn2 = w < 4 ? w : w < 7 ? w - 3 : w < 9 ? w - 5 : 3;
    const int colA2 = 16 * tm2 + c, colB2 = 16 * tn2 + c;
    v4d mre = {0.0, 0.0, 0.0, 0.0}, mim = {0.0, 0.0, 0.0, 0.0};
    if (upper) {
#pragma unroll 4
        for (int ks = 0; ks < 16; ++ks) {               // M' = V^H T
            const int k = 4 * ks + g;
            const cplx v = L.Vs[sys_index<D>(k, colA2)];    // conj((V^H)[colA2][k])
            const cplx q = L.Ms[sys_index<D>(k, colB2)];    // T[k][colB2]
            mre = __builtin_amdgcn_mfma_f64_16x16x4f64(v.re, q.re, mre, 0, 0, 0);
            mim = __builtin_amdgcn_mfma_f64_16x16x4f64(v.re, q.im, mim, 0, 0, 0);
            mre = __builtin_amdgcn_mfma_f64_16x16x4f64(v.im, q.im, mre, 0, 0, 0);
            mim = __builtin_amdgcn_mfma_f64_16x16x4f64(-v.im, q.re, mim, 0, 0, 0);
        }
    }
    FBX_BLOCK_SYNC();                                    // every wavefront is done reading T
    if (upper) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * tm2 + g + 4 * r;
            cplx o; o.re = mre[r]; o.im = row == colB2 ? 0.0 : mim[r];
            L.Ms[sys_index<D>(row, colB2)] = o;
        }
    }
    FBX_BLOCK_SYNC();
}

// The 64 x 64 eigensolver is INLINED into the kernels (behind a call boundary -- tried in round 4, when its per-thread address constants
// were hoisted out of the Dykstra loop and spilled -- the role-split solver needs more than the 80 caller-saved registers and saved 49
// more to scratch per call: 296 against 274 ms per 256 reconstructions).
__device__ __forceinline__
int jacobi64(cplx* Ms, cplx* Vs, int t, bool init_identity, double* red, double tol2) {
    t = opaque(t);
    return jacobi_eigh64<NT>(Ms, Vs, t, init_identity, red, tol2);     // fbx_eigh64.hpp
}

// ---- V diag(lam) V^H for the 64 x 64 eigenvectors in Vs: the block of thread (I, J) is sum_k lam_k V[2I + a][k] conj(V[2J + b][k]).
// The first factor is the same for the 32 lanes of a block row (a broadcast read); the second walks DOWN a column of the
// layout -- a stride of 32 entries, 8-way bank conflicts on two of the four b128 reads of every term: the generic
// reconstruct_blk was LDS-bound at ~1300 cycles per eigenvalue term (34 k cycles per call, 5.8 % of the kernel).  Here V is
// first copied TRANSPOSED into Ms (dead once the eigenvalues are in L.lam), block (J, I) at J * 32 + (I ^ J): the XOR makes the
// eight blocks a lane group writes AND the eight it later reads fall on eight different bank groups, so both factors are
// conflict-free reads.  Same terms in the same order: bit-identical to reconstruct_blk<64>.
__device__ Blk reconstruct64(Lds& L, int t) {
    constexpr int PS = sys_plane<D>();
    t = opaque(t);
    const int I = t / NB, J = t % NB;
#pragma unroll
    for (int e = 0; e < 4; ++e) {                          // V[2I + a][2J + b] -> VT[2J + b][2I + a]
        const int a = e >> 1, b = e & 1;
        L.Ms[(b * 2 + a) * PS + J * NB + (I ^ J)] = L.Vs[e * PS + sys_pos<D>(I, J, e)];
    }
    FBX_BLOCK_SYNC();
    Blk out = blk_zero();
    const int wl = t & 63;
    const double mine = L.lam[wl];
    unsigned long long todo = __ballot(mine != 0.0);
    while (todo) {
        const int k = __builtin_ctzll(todo);
        todo &= todo - 1;
        const double l = readlane_f64(mine, k);
        const int kb = k >> 1, ke = k & 1;
        const int rk = sys_pos<D>(I, kb, ke);
        const cplx r0 = L.Vs[(0 + ke) * PS + rk], r1 = L.Vs[(2 + ke) * PS + rk];
        const cplx c0 = L.Ms[(ke * 2 + 0) * PS + kb * NB + (J ^ kb)], c1 = L.Ms[(ke * 2 + 1) * PS + kb * NB + (J ^ kb)];
        const double w0r = l * r0.re, w0i = l * r0.im, w1r = l * r1.re, w1i = l * r1.im;
        out.re[0] += w0r * c0.re + w0i * c0.im; out.im[0] += w0i * c0.re - w0r * c0.im;
        out.re[1] += w0r * c1.re + w0i * c1.im; out.im[1] += w0i * c1.re - w0r * c1.im;
        out.re[2] += w1r * c0.re + w1i * c0.im; out.im[2] += w1i * c0.re - w1r * c0.im;
        out.re[3] += w1r * c1.re + w1i * c1.im; out.im[3] += w1i * c1.re - w1r * c1.im;
    }
    return out;
}

// ---- CP projection (project_superoperators.py:19-34)
// Hermitised copy of x into Ms (element-major layout); returns ||h||_F^2's per-thread part when asked
__device__ __forceinline__ double hermitise_into_ms(const Blk& x, Lds& L, int t, bool want_norm) {
    t = opaque(t);
    FBX_BLOCK_SYNC();
    // the transposed copy goes through a layout of its own: block (I, J) of plane e at e * PS + I * 32 + (I ^ J), so that the
    // eight blocks a lane group writes (eight column pairs of one row) AND the eight it reads back (eight ROW pairs of one column:
    // a stride of 32 entries in the solver's layout, 8-way bank conflicts on every b128 read) fall on eight different bank groups
    Blk xa;
    {
        constexpr int PS = sys_plane<D>();
        const int I = t / NB, J = t % NB, sw = I ^ J;
#pragma unroll
        for (int e = 0; e < 4; ++e) { cplx c; c.re = x.re[e]; c.im = x.im[e]; L.Ms[e * PS + I * NB + sw] = c; }
        FBX_BLOCK_SYNC();
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int et = (e & 1) * 2 + (e >> 1);
            const cplx c = L.Ms[et * PS + J * NB + sw];
            xa.re[e] = c.re; xa.im[e] = -c.im;
        }
    }
    Blk h;
#pragma unroll
    for (int e = 0; e < 4; ++e) { h.re[e] = 0.5 * (x.re[e] + xa.re[e]); h.im[e] = 0.5 * (x.im[e] + xa.im[e]); }
    FBX_BLOCK_SYNC();
    sys_store<D>(L.Ms, t, h);
    FBX_BLOCK_SYNC();
    return want_norm ? blk_norm2(h) : 0.0;
}
__device__ __forceinline__ Blk unpark_blk(const cplx* g, int t);
// (`upark`: the caller's two matrices are read back from their parking slab behind the eigensolver, see proj_physical)
__device__ __forceinline__ Blk proj_cp(const Blk& x_, Lds& L, int t_, int& sweeps, bool warm = false, cplx* Tg = nullptr,
                       bool check_basis = false, const cplx* upark = nullptr, Blk* u_back = nullptr, Blk* p_back = nullptr) {
    const Blk x = x_;
    int t = t_;
    // (the Hermitised matrix lives in LDS only: a register copy kept for the rare rejected basis would stay live
    // across the eigensolver -- 16 of 128 registers; the rejection path rebuilds it from x instead)
    double n2[2] = {0.0, 0.0};
    n2[0] = hermitise_into_ms(x, L, t, warm && check_basis);
    PH_STOP(*L.pc, 2);
    if (warm) {
        // a basis loaded from the store is only trusted if the change of basis kept ||.||_F^2 (unitary
        // similarity); otherwise the matrix is restored and the decomposition starts from the identity
        // (same guard as proj_cp_blk, fbx_choi.hpp)
        (void)Tg; rotate_into_basis_mfma(L, t);
        if (check_basis) {
            {   // (the matrix-core form leaves the tiles below the diagonal unwritten: the upper ones count twice)
                const int ti = t / NB / 8, tj = t % NB / 8;      // tile of entry position t (the layout permutes column pairs inside aligned groups of 8 only)
                const double wgt = ti < tj ? 2.0 : ti == tj ? 1.0 : 0.0;
                double acc2 = 0.0;
#pragma unroll
                for (int e = 0; e < 4; ++e) { const cplx v = L.Ms[e * NT + t]; acc2 = fma(v.re, v.re, fma(v.im, v.im, acc2)); }
                n2[1] = wgt * acc2;
            }
            bsum_multi<2>(n2, L);
            if (!(fabs(n2[1] - n2[0]) <= FBX_BASIS_NORM_TOL * n2[0])) {
                (void)hermitise_into_ms(x, L, t, false);
                warm = false;
            }
        }
    }
    PH_STOP(*L.pc, 6);
    if (upark) __asm__ volatile("" ::: "memory");       // nothing of the caller's parked matrices stays in registers across the solver
    sweeps += jacobi64(L.Ms, L.Vs, t, !warm, L.red, L.jtol2);
    t = opaque(t);
    if (upark) {                                        // requested here, consumed behind the reconstruction below
        __asm__ volatile("" ::: "memory");
        *u_back = unpark_blk(upark, t); *p_back = unpark_blk(upark + 4 * NT, t);
    }
    PH_STOP(*L.pc, 0);
    if (t < D) {
        const double l = L.Ms[sys_index<D>(t, t)].re;
        L.lam[t] = l < 0.0 ? 0.0 : l;
    }
    FBX_BLOCK_SYNC();
    // work accounting: eigenvalue terms the reconstruction walks (thread 0 reports it; every wavefront holds lam[lane])
    L.terms += __popcll(__ballot(L.lam[t & 63] != 0.0));
    const Blk out = reconstruct64(L, t);
    PH_STOP(*L.pc, 1);
    return out;
}

// ---- partial trace over the output space into L.pt (calculational.py:5-35); stages x through Mw
__device__ void partial_trace_out(const Blk& x, Lds& L, int t) {
    t = opaque(t);
    // staged row-major through Ms (dead between the reconstruction and the next projection), NOT
    // through Mw = Vs: the eigenvectors in Vs must survive for the warm start of the next projection
    cplx* St = L.Ms;
    FBX_BLOCK_SYNC();
    blk_store<D, LD>(St, t, x);
    FBX_BLOCK_SYNC();
    if (t < d * d) {
        const int i = t / d, ip = t % d;
        cplx s; s.re = 0.0; s.im = 0.0;
#pragma unroll
        for (int o = 0; o < d; ++o) { const cplx v = St[(i * d + o) * LD + ip * d + o]; s.re += v.re; s.im += v.im; }
        L.pt[i * LDs + ip] = s;
    }
    FBX_BLOCK_SYNC();
}
__device__ __forceinline__ Blk subtract_kron_pt(const Blk& x, const Lds& L, int t) {
    Blk r = x;
    t = opaque(t);
    const int I = t / NB, J = t % NB;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int row = 2 * I + (e >> 1), col = 2 * J + (e & 1);
        if ((row % d) == (col % d)) { const cplx c = L.pt[(row / d) * LDs + (col / d)]; r.re[e] -= c.re / d; r.im[e] -= c.im / d; }
    }
    return r;
}
__device__ Blk proj_tp(const Blk& x, Lds& L, int t) {                 // project_superoperators.py:62-84
    partial_trace_out(x, L, t);
    if (t < d) L.pt[t * LDs + t].re -= 1.0;
    FBX_BLOCK_SYNC();
    return subtract_kron_pt(x, L, t);
}
__device__ Blk proj_tni(const Blk& x, Lds& L, int t, int& sweeps) {   // project_superoperators.py:37-59
    partial_trace_out(x, L, t);
    const Blk ptb = blk_load<d, LDs>(L.pt, t);
    const Blk pta = blk_load_adjoint<d, LDs>(L.pt, t);
    Blk h;
#pragma unroll
    for (int e = 0; e < 4; ++e) { h.re[e] = 0.5 * (ptb.re[e] + pta.re[e]); h.im[e] = 0.5 * (ptb.im[e] + pta.im[e]); }
    FBX_BLOCK_SYNC();
    sys_store<d>(L.pts, t, h);
    FBX_BLOCK_SYNC();
    sweeps += jacobi_eigh_simple<d, NT>(L.pts, L.ptV, t, true, L.red);
    if (t < d) { const double l = L.pts[sys_index<d>(t, t)].re; L.lam[t] = l > 1.0 ? 1.0 : l; }
    FBX_BLOCK_SYNC();
    const Blk proj = reconstruct_blk<d>(L.ptV, L.lam, t);
    FBX_BLOCK_SYNC();
    blk_store<d, LDs>(L.pt, t, blk_sub(ptb, proj));
    FBX_BLOCK_SYNC();
    return subtract_kron_pt(x, L, t);
}

// -DFBX3_DYK_DETAIL (diagnostics, with FBX_PHASE_TIMERS): the phases of a Dykstra iteration take over the timer slots of
// the outer phases -- 3 basis load, 7 basis write-back, 4 TP / TNI projection, 5 stopping functional (2 then = Hermitise only)
#define PH3(ph)
// one stored basis (64 KiB, the layout of Vs) from HBM / L2 into LDS without passing through registers: every lane moves
// four 16-byte entries; the LDS address of a global_load_lds is wave-uniform base + lane * 16
// s_waitcnt vmcnt(0) (gfx9 encoding: vmcnt = bits 3:0 and 15:14, expcnt 6:4 and lgkmcnt 11:8 left at their maxima): the consumer side
// of basis_fetch -- global_load_lds is a VMEM load whose completion is counted by vmcnt, not by the LDS counter
__device__ __forceinline__ void wait_global_loads() { __builtin_amdgcn_s_waitcnt(0x0070); }
// Ownership: thread t moves entries e * NT + t (e = 0..3) of a stored basis in BOTH directions -- write-back (store_index below) and
// fetch -- so no thread ever reads global memory another thread wrote, which is what lets the barriers order LDS only.
__device__ __forceinline__ int store_index(int e, int t) { return e * NT + t; }
__device__ __forceinline__ void basis_fetch(const cplx* g, cplx* Vs, int t) {
    typedef __attribute__((address_space(1))) const void* gptr;
    typedef __attribute__((address_space(3))) void* lptr;
#pragma unroll
    for (int e = 0; e < 4; ++e)
        __builtin_amdgcn_global_load_lds((gptr)(g + store_index(e, t)), (lptr)(Vs + store_index(e, t & ~63)), 16, 0, 0);
}
// block of kron(C, I_d) for the d x d matrix C staged in LDS (leading dimension LDs)
__device__ __forceinline__ Blk kron_id_blk(const cplx* C, int t) {
    Blk r = blk_zero();
    t = opaque(t);
    const int I = t / NB, J = t % NB;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int row = 2 * I + (e >> 1), col = 2 * J + (e & 1);
        if ((row % d) == (col % d)) { const cplx c = C[(row / d) * LDs + (col / d)]; r.re[e] = c.re; r.im[e] = c.im; }
    }
    return r;
}

// ---- Dykstra (project_superoperators.py:87-144), carried with TWO matrices per thread.
// The reference's loop state is (last_state, old_CP_change, old_TP_change, last_CP_projection).  Here:
//   * old_TP_change = new_state - pre_TP = -kron(corr / d, I_d) with corr the d x d correction the TP / TNI projection
//     subtracted (proj_tp / proj_tni leave it in L.pt): kept as that 8 x 8 matrix in LDS (L.ptold), rebuilt per entry
//     where it is used;
//   * last_CP_projection only enters through <old_CP_change, CP_projection - last_CP_projection>: the second half is
//     the scalar <new_CP_change, CP_projection> of the previous iteration (one more term of the block reduction);
//   * last_state = pre_CP + old_CP_change.
// What stays live in registers across the eigensolver is pre_CP and old_CP_change -- 32 of the 128 registers a
// thread of a 1024-thread workgroup has, next to the solver's 80 (rounds 1-2 carried four matrices + the caller's
// estimate, gradient and counts: 1.4 KB of scratch per lane and 235 GB of spill traffic per 256-item launch).
// Differences to the literal expressions are rounding-level terms of the stopping functional (threshold 1e-4).
// `upark` (optional, [2][4][1024] entries per workgroup, HBM / L2): pre_CP and old_CP_change are written there in front of the
// eigensolver and read back behind it -- explicitly, coalesced, the loads issued before the reconstruction that covers their
// latency -- instead of being spilled around it by the compiler (the role-split solver of fbx_eigh64.hpp wants the registers).
__device__ __forceinline__ void park_blk(cplx* g, int t, const Blk& b) {
    fbx_global_cplx_ptr dst = (fbx_global_cplx_ptr)g;
#pragma unroll
    for (int e = 0; e < 4; ++e) dst[e * NT + t] = fbx_v2d{b.re[e], b.im[e]};
}
__device__ __forceinline__ Blk unpark_blk(const cplx* g, int t) {
    fbx_global_cplx_ptr src = (fbx_global_cplx_ptr)const_cast<cplx*>(g);
    Blk b;
#pragma unroll
    for (int e = 0; e < 4; ++e) { const fbx_v2d w = src[e * NT + t]; b.re[e] = w.x; b.im[e] = w.y; }
    return b;
}
__device__ __forceinline__ Blk proj_physical(const Blk& x, bool tp, Lds& L, int t_, int& iters, int& sweeps, cplx* Tg,
                             BasisStore* store = nullptr, cplx* upark = nullptr) {
    int t = t_;
    Blk u = x;                       // pre_CP = last_state - old_CP_change
    Blk p = blk_zero();              // old_CP_change
    Blk new_state = x;
    double c0r = 0.0, c0i = 0.0;     // <old_CP_change, last_CP_projection>
    bool pf_pending = false;         // the next iteration's stored basis is on its way from HBM / L2 into Vs
    FBX_BLOCK_SYNC();
    if (t < d * LDs) { cplx z; z.re = 0.0; z.im = 0.0; L.ptold[t] = z; }     // old_TP_change = 0
    int it = 0;
    for (; it < 100000; ++it) {
        ++iters;
        // consecutive Dykstra iterates are close: start from the previous eigenvectors; when the
        // outer step was small, from the basis the previous call found at the same Dykstra
        // iteration (BasisStore, fbx_choi.hpp) -- the first projection always, it has no other
        bool warm = it > 0 && Tg != nullptr;
        bool from_slot = false;
        if (store && Tg && it < store->nprev && (it == 0 || store->use_prev)) {
            from_slot = true;
            if (!pf_pending) { FBX_BLOCK_SYNC(); basis_fetch(store->g + (size_t)it * D * D, L.Vs, t); }
            wait_global_loads();                      // this wave's share of the basis has landed in LDS
            FBX_BLOCK_SYNC();
            pf_pending = false;
            warm = true;
        }
        PH3(3);
        t = opaque(t);
        if (upark) { park_blk(upark, t, u); park_blk(upark + 4 * NT, t, p); }
        const Blk cp = proj_cp(u, L, t, sweeps, warm, Tg, from_slot, upark, &u, &p);
        if (store && it < store->cap && (it == 0 || store->write_all)) {      // write-back policy: BasisStore, fbx_choi.hpp
            fbx_global_cplx_ptr dst = (fbx_global_cplx_ptr)(store->g + (size_t)it * D * D);
#pragma unroll
            for (int e = 0; e < 4; ++e) { const cplx v = L.Vs[store_index(e, t)]; dst[store_index(e, t)] = fbx_v2d{v.re, v.im}; }
        }
        PH3(7);
        double red8[8];
        const Blk new_cp = blk_sub(cp, u);
        red8[0] = blk_norm2(blk_sub(new_cp, p));                           // ||new_CP_change - old_CP_change||^2
        blk_dotc(p, cp, red8[4], red8[5]);                                 // <old_CP_change, CP_projection>
        blk_dotc(new_cp, cp, red8[6], red8[7]);                            // next iteration's <old_CP_change, last_CP_projection>
        const Blk last_state = blk_axpy(u, 1.0, p);
        // pre_TP = CP_projection - old_TP_change = CP_projection + kron(corr_old / d, I)
        const Blk old_tp = blk_axpy(blk_zero(), -1.0 / d, kron_id_blk(L.ptold, t));
        const Blk pre_tp = blk_sub(cp, old_tp);
        new_state = tp ? proj_tp(pre_tp, L, t) : proj_tni(pre_tp, L, t, sweeps);
        PH3(4);
        const Blk new_tp = blk_axpy(blk_zero(), -1.0 / d, kron_id_blk(L.pt, t));
        red8[1] = blk_norm2(blk_sub(new_tp, old_tp));                      // ||new_TP_change - old_TP_change||^2
        blk_dotc(old_tp, blk_sub(new_state, last_state), red8[2], red8[3]); // <old_TP_change, state_change>
        // (measured and dropped, round 5: contracting the state change in two halves -- (new_CP_change - old_CP_change) in front
        //  of the TP projection, the Kronecker-structured TP half behind it -- so that only new_CP_change crosses the projection:
        //  200.5 against 197.4 ms per 256 reconstructions)
        bsum_multi<8>(red8, L);
        PH3(5);
        const double i2r = red8[4] - c0r, i2i = red8[5] - c0i;
        const double crit = red8[0] + red8[1] + 2.0 * sqrt(red8[2] * red8[2] + red8[3] * red8[3]) + 2.0 * sqrt(i2r * i2r + i2i * i2i);
        if (!(crit >= 1e-4)) { ++it; break; }        // converged -- or not finite (NaN input): never spin
        c0r = red8[6]; c0i = red8[7];
        p = new_cp;
        u = blk_sub(new_state, new_cp);
        if (t < d * LDs) L.ptold[t] = L.pt[t];       // (bsum_multi's barriers separate this from the readers above; the next
                                                     //  reader is behind the barriers of proj_cp)
        // The eigenvectors in Vs are dead when the next decomposition starts from a stored basis (reconstruction and
        // write-back have read them, behind the barriers above): that basis is requested NOW, straight into LDS
        // (global_load_lds_dwordx4), and arrives while the next iteration Hermitises its matrix.
        if (store && Tg && it + 1 < store->nprev && store->use_prev) { basis_fetch(store->g + (size_t)(it + 1) * D * D, L.Vs, t); pf_pending = true; }
    }
    if (store) {
        const int written = store->write_all ? it : 1;
        store->nprev = written < store->cap ? written : store->cap;
    }
    return new_state;
}

// ---- Choi (in registers) -> transposed Pauli coefficients Rt (uses Mw = Vs as staging)
__device__ void choi_to_pauli(const Blk& x, Lds& L, int t) {
    t = opaque(t);
    FBX_BLOCK_SYNC();
    blk_store<D, LD>(L.Mw, t, x);
    FBX_BLOCK_SYNC();
#pragma unroll
    for (int s = NQ - 1; s >= 0; --s) { pauli_site_stage<NQ, false, LD>(L.Mw, t, 2 * NQ + NQ + s, NQ + s, -1.0); FBX_BLOCK_SYNC(); }
#pragma unroll
    for (int s = NQ - 1; s >= 0; --s) { pauli_site_stage<NQ, false, LD>(L.Mw, t, 2 * NQ + s, s, +1.0); FBX_BLOCK_SYNC(); }
    for (int idx = t; idx < D * D; idx += NT) {
        const int i = idx % D, j = idx / D;           // idx = j * 64 + i: coalesced Rt writes
        int row, col;
        pauli_coeff_position<NQ>(i, j, row, col);
        L.Rt[idx] = L.Mw[row * LD + col].re / d;
    }
    FBX_BLOCK_SYNC();
}
// ---- transposed Pauli coefficients Rt -> Choi block (Mw = Vs as scratch)
__device__ Blk pauli_to_choi(Lds& L, int t) {
    t = opaque(t);
    FBX_BLOCK_SYNC();
    for (int idx = t; idx < D * D; idx += NT) {
        const int i = idx % D, j = idx / D;
        int row, col;
        pauli_coeff_position<NQ>(i, j, row, col);
        cplx v; v.re = L.Rt[idx] * d; v.im = 0.0;
        L.Mw[row * LD + col] = v;
    }
    FBX_BLOCK_SYNC();
#pragma unroll
    for (int s = 0; s < NQ; ++s) { pauli_site_stage<NQ, true, LD>(L.Mw, t, 2 * NQ + s, s, +1.0); FBX_BLOCK_SYNC(); }
#pragma unroll
    for (int s = 0; s < NQ; ++s) { pauli_site_stage<NQ, true, LD>(L.Mw, t, 2 * NQ + NQ + s, NQ + s, -1.0); FBX_BLOCK_SYNC(); }
    const Blk out = blk_load<D, LD>(L.Mw, t);
    FBX_BLOCK_SYNC();
    return out;
}
// ---- T[s][i] = sum_j R[i][j] C[j][s]: the dense basis-change GEMM of the 3-qubit path
// ([S x 64] = C^T [S x 64] . R^T [64 x 64]) on the fp64 matrix cores.  v_mfma_f64_16x16x4_f64:
// lane l feeds A[m = l & 15][k = l >> 4] and B[k = l >> 4][n = l & 15]; its four accumulators are
// D[row = (l >> 4) + 4 r][col = l & 15].  B = Rt is already [k][n] row-major in LDS; A = C^T comes
// from L2 (C is [64][S]).  16 waves share the ceil(S/16) x 4 output tiles.
typedef double v4d __attribute__((ext_vector_type(4)));
__device__ void predict_table(const DesignDev& des, Lds& L, int t) {
    const int S = des.S;
    t = opaque(t);
    const int lane = t & 63, wave = t >> 6;
    const int mtiles = (S + 15) / 16;
    FBX_BLOCK_SYNC();
    for (int tile = wave; tile < mtiles * 4; tile += NT / 64) {
        const int ms = tile / 4, ni = tile % 4;
        const int srow = ms * 16 + (lane & 15);
        v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
        for (int kk = 0; kk < D / 4; ++kk) {
            const int k = 4 * kk + (lane >> 4);
            const double a = srow < S ? des.C[k * S + srow] : 0.0;
            const double b = L.Rt[k * D + ni * 16 + (lane & 15)];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int srw = ms * 16 + (lane >> 4) + 4 * r;
            if (srw < S) L.T[srw * D + ni * 16 + (lane & 15)] = acc[r];
        }
    }
    FBX_BLOCK_SYNC();
}
// ---- Rt[j][i] = -(1/d^2) sum_s W[i][s] C[j][s]  ([64 x 64] = W [64 x S] . C^T [S x 64]), one
// 16 x 16 output tile per wave; W from LDS, C from L2.  S is padded with zero terms to a multiple of 4.
__device__ void gradient_coefficients(const DesignDev& des, Lds& L, const double* W, int t) {
    const int S = des.S;
    t = opaque(t);
    const int lane = t & 63, wave = t >> 6;
    const int mi = wave / 4, nj = wave % 4;
    v4d acc = {0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < S; k0 += 4) {
        const int k = k0 + (lane >> 4);
        const double a = k < S ? W[(mi * 16 + (lane & 15)) * S + k] : 0.0;
        const double b = k < S ? des.C[(nj * 16 + (lane & 15)) * S + k] : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    FBX_BLOCK_SYNC();                                   // Rt overlays the partial arrays read above
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = mi * 16 + (lane >> 4) + 4 * r, j = nj * 16 + (lane & 15);
        L.Rt[j * D + i] = -acc[r] / (double)(d * d);
    }
    FBX_BLOCK_SYNC();
}
}  // namespace p3

// Per-workgroup slab in HBM / L2 for what must survive the projection but is not touched by it -- the estimate, the
// gradient's Pauli coefficients, the model probabilities of the estimate and the normalised counts: written once and
// read once (counts: twice) per OUTER iteration with coalesced 8 / 16-byte accesses, instead of living in registers the
// compiler then spills around every one of the ~10 eigendecompositions of the projection.
template <int MAXJ>
struct Park3 {
    static constexpr size_t doubles() { return 2 * (size_t)p3::D * p3::D + (size_t)p3::D * p3::D + 4 * (size_t)MAXJ * p3::NT + 4 * (size_t)p3::D * p3::D; }
    double* base;
    __device__ cplx* est() const { return (cplx*)base; }                              // [4][1024] blocks, element-major
    __device__ double* rg() const { return base + 2 * p3::D * p3::D; }                // [4096] gradient coefficients (Rt layout)
    __device__ double* pp() const { return rg() + p3::D * p3::D; }                    // [2 MAXJ][1024] probabilities of the estimate
    __device__ double* nn() const { return pp() + 2 * MAXJ * p3::NT; }                // [2 MAXJ][1024] normalised counts
    __device__ cplx* dyk() const { return (cplx*)(nn() + 2 * MAXJ * p3::NT); }       // [2][4][1024] Dykstra's two matrices around the eigensolver
};

template <int MAXJ>
__global__ void __launch_bounds__(1024)
pgdb3_kernel(DesignDev des, long long B, const double* __restrict__ expect, const double* __restrict__ counts,
             int trace_preserving, int mode, int max_iters, double* __restrict__ choi_out,
             int* __restrict__ iters_out, int* __restrict__ dykstra_out, int* __restrict__ backtracks_out,
             double* __restrict__ cost_out, cplx* __restrict__ scratch, long long* __restrict__ phase_out, int* __restrict__ sweeps_out,
             cplx* __restrict__ basis_scratch, int basis_cap, int* __restrict__ trace_out, int trace_iters,
             double* __restrict__ park_base) {
    using namespace p3;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Lds L; L.carve(smem);
    PhaseClock pc; pc.reset(); L.pc = &pc;
    const int t = threadIdx.x;
    const long long item = blockIdx.x;
    const int m = des.m, S = des.S;
    const bool unit_coefs = des.unit_coefs != 0;
    Park3<MAXJ> park; park.base = park_base + (size_t)blockIdx.x * Park3<MAXJ>::doubles();

    // ---- data: n+-[k] = counts * (1 +- e)/2 / grand_total   (tomography.py:528-538), parked
    {
        double npl[MAXJ], nmi[MAXJ];
        double tot = 0.0;
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) {
            const int g = t * MAXJ + j;
            npl[j] = 0.0; nmi[j] = 0.0;
            if (g < m) {
                const int k = des.order[g];
                const double e = expect[item * m + k], c = counts[item * m + k];
                const double plus = (1.0 + e) / 2.0;
                npl[j] = c * plus; nmi[j] = c * (1.0 - plus);
                tot += c;
            }
        }
        tot = bsum(tot, L);
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) { park.nn()[(2 * j) * NT + t] = npl[j] / tot; park.nn()[(2 * j + 1) * NT + t] = nmi[j] / tot; }
    }
    const double half_dd = 0.5 / (double)(d * d), inv_mu = (2.0 * d * d) / 3.0;

    // model probabilities of this thread's outcomes from the prediction table in LDS
    auto probs_of = [&](int j, double& pp, double& pm, double missing) __attribute__((always_inline)) {
        const int g = t * MAXJ + j;
        pp = missing; pm = missing;
        if (g < m) {
            const uint32_t w = des.sp[g];
            const int s = w >> 16, p = w & 0xffff;
            const double cf = unit_coefs ? 1.0 : des.coef[g];
            const double tr = L.T[s * D], ex = cf * L.T[s * D + p];
            pp = (tr + ex) * half_dd; pm = (tr - ex) * half_dd;
        }
    };

    Blk est = blk_zero();
    { const int I = t / NB, J = t % NB; if (I == J) { est.re[0] = 1.0 / d; est.re[3] = 1.0 / d; } }
    int iters = 0, dyk = 0, backtracks = 0, sweeps = 0, cost_evals = 0, chain_start = 0;
    double old_cost = 0.0, new_cost = 0.0;
    bool have_cost = false;
    BasisStore basis;
    basis.g = basis_scratch ? basis_scratch + (size_t)blockIdx.x * basis_cap * D * D : nullptr;
    basis.cap = basis_cap; basis.nprev = 0; basis.use_prev = false; basis.write_all = false;
    double outer_step = 1.0;

    PH_START(pc);
    while (true) {
        if (mode == FBX_MODE_FIXED && iters >= max_iters) break;
        const int dyk_before = dyk, bt_before = backtracks;       // per-iteration trace (fbx_pgdb_process_ex)
        choi_to_pauli(est, L, t);
        PH_STOP(pc, 3);
        predict_table(des, L, t);
        PH_STOP(pc, 7);

        // ---- probabilities of the estimate (parked for the line search), initial cost, and the gradient
        // (tomography.py:617-633): W[i][s] = sum over the settings of state s.
        // Threads own CONTIGUOUS runs of the state-grouped settings, so the identity row W[0][s]
        // (one term per setting of the state) is reduced deterministically: a run that lies inside
        // one thread is written directly, the first / last run of every thread goes to a partial
        // array that one thread per state adds up in thread order.  The W[p][s] cells receive one
        // term each (LDS atomics only matter for designs that repeat a setting).
        double ep[MAXJ], em[MAXJ];                     // eta = n / clip(p) of this thread's outcomes
        {
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < MAXJ; ++j) {
                double pp, pm;
                probs_of(j, pp, pm, 1.0);
                park.pp()[(2 * j) * NT + t] = pp; park.pp()[(2 * j + 1) * NT + t] = pm;
                pp = pp < EPS ? EPS : pp; pm = pm < EPS ? EPS : pm;
                const double np_ = park.nn()[(2 * j) * NT + t], nm_ = park.nn()[(2 * j + 1) * NT + t];
                if (!have_cost && t * MAXJ + j < m) acc -= np_ * fast_log_pos(pp) + nm_ * fast_log_pos(pm);
                ep[j] = np_ / pp; em[j] = nm_ / pm;
            }
            if (!have_cost) { old_cost = bsum(acc, L); have_cost = true; ++cost_evals; }   // tomography.py:565
        }
        PH_STOP(pc, 5);
        FBX_BLOCK_SYNC();                              // T fully consumed
        double* W = L.T;
        double* pfirst = L.Rt + 768;                  // overlays R (dead here), past the small scratch
        double* plast = pfirst + NT;
        int* sfirst = (int*)(plast + NT);
        int* slast = sfirst + NT;
        for (int idx = t; idx < D * S; idx += NT) W[idx] = 0.0;
        sfirst[t] = -1; slast[t] = -1; pfirst[t] = 0.0; plast[t] = 0.0;
        FBX_BLOCK_SYNC();
        {
            int run_state = -1, first_state = -1; double run = 0.0; bool first_done = false;
#pragma unroll
            for (int j = 0; j < MAXJ; ++j) {
                const int g = t * MAXJ + j;
                if (g < m) {
                    const uint32_t w = des.sp[g];
                    const int s = w >> 16, p = w & 0xffff;
                    const double cf = unit_coefs ? 1.0 : des.coef[g];
                    atomicAdd(&W[p * S + s], cf * 0.5 * (ep[j] - em[j]));
                    if (s != run_state) {
                        if (run_state >= 0) {                      // flush the finished run
                            if (!first_done) { pfirst[t] = run; sfirst[t] = run_state; first_done = true; }
                            else atomicAdd(&W[run_state], run);    // interior run: sole contributor
                        }
                        run_state = s; run = 0.0;
                        if (first_state < 0) first_state = s;
                    }
                    run += 0.5 * (ep[j] + em[j]);
                }
            }
            if (run_state >= 0) {
                if (!first_done) { pfirst[t] = run; sfirst[t] = run_state; }
                else { plast[t] = run; slast[t] = run_state; }
            }
        }
        FBX_BLOCK_SYNC();
        for (int s = t; s < S; s += NT) {
            const int g0 = des.sptr[s], g1 = des.sptr[s + 1];
            if (g1 > g0) {
                double acc = 0.0;
                for (int tt = g0 / MAXJ; tt <= (g1 - 1) / MAXJ; ++tt) {
                    if (sfirst[tt] == s) acc += pfirst[tt];
                    if (slast[tt] == s) acc += plast[tt];
                }
                W[s] += acc;
            }
        }
        FBX_BLOCK_SYNC();
        gradient_coefficients(des, L, W, t);
        // the gradient is needed once more after the projection, for <update, gradient>: as its Pauli coefficients
        // (the transform is unitary up to the factor d: <E1, E2> = sum_ij R1_ij R2_ij), parked
        for (int idx = t; idx < D * D; idx += NT) park.rg()[idx] = L.Rt[idx];
        PH_STOP(pc, 4);
        Blk x;
        {
            const Blk grad = pauli_to_choi(L, t);
            x = blk_axpy(est, -inv_mu, grad);
        }
        PH_STOP(pc, 3);
#pragma unroll
        for (int e = 0; e < 4; ++e) { cplx v; v.re = est.re[e]; v.im = est.im[e]; park.est()[e * NT + t] = v; }

        // bounds the accumulated loss of unitarity of the chained bases: a cold restart once the chains have
        // absorbed 54 sweeps per slot (what 16 converging iterations apply; see fbx_pgdb.hip)
        if (iters == 0 || sweeps - chain_start >= FBX3_BASIS_CHAIN_SWEEPS * (basis.nprev > 0 ? basis.nprev : 1)) { basis.nprev = 0; chain_start = sweeps; }
        basis.use_prev = outer_step < FBX3_BASIS_STEP;
        basis.write_all = outer_step < FBX3_BASIS_WRITE_STEP;
        { const double tr_ = des.eig_rel_tol * outer_step; L.jtol2 = fmax(FBX_JACOBI_TOL2, tr_ * tr_); }   // as in fbx_pgdb.hip
        const Blk proj = proj_physical(x, trace_preserving != 0, L, t, dyk, sweeps,
                                       scratch,
                                       basis.g ? &basis : nullptr, nullptr);
#pragma unroll
        for (int e = 0; e < 4; ++e) { const cplx v = park.est()[e * NT + t]; est.re[e] = v.re; est.im[e] = v.im; }
        const Blk upd = blk_sub(proj, est);
        PH_STOP(pc, 2);

        choi_to_pauli(upd, L, t);
        PH_STOP(pc, 3);
        double ipr = 0.0;                             // <update, gradient> = sum_ij R^upd_ij R^grad_ij
        for (int idx = t; idx < D * D; idx += NT) ipr = fma(L.Rt[idx], park.rg()[idx], ipr);
        predict_table(des, L, t);
        ipr = bsum(ipr, L);                           // (after the table product: the reduction scratch overlays Rt)
        PH_STOP(pc, 7);

        // ---- backtracking line search (tomography.py:575-585)
        double alpha = 1.0;
        {
            double npl[MAXJ], nmi[MAXJ], pep[MAXJ], pem[MAXJ], pup[MAXJ], pum[MAXJ];
#pragma unroll
            for (int j = 0; j < MAXJ; ++j) {
                probs_of(j, pup[j], pum[j], 0.0);
                pep[j] = park.pp()[(2 * j) * NT + t]; pem[j] = park.pp()[(2 * j + 1) * NT + t];
                npl[j] = park.nn()[(2 * j) * NT + t]; nmi[j] = park.nn()[(2 * j + 1) * NT + t];
            }
            auto cost_at = [&](double a) -> double {
                double acc = 0.0;
                auto slot = [&](int j) __attribute__((always_inline)) {
                    if (t * MAXJ + j < m) {
                        double pp = fma(a, pup[j], pep[j]), pm = fma(a, pum[j], pem[j]);
                        pp = pp < EPS ? EPS : pp; pm = pm < EPS ? EPS : pm;
                        acc -= npl[j] * fast_log_pos(pp) + nmi[j] * fast_log_pos(pm);
                    }
                };
                // (beyond four slots the six slot arrays do not fit the 128 registers of a thread: they live in scratch and the passes
                //  over them are loops -- same terms in the same order)
                if constexpr (MAXJ <= 4) {
#pragma unroll
                    for (int j = 0; j < MAXJ; ++j) slot(j);
                } else {
#pragma unroll 1
                    for (int j = 0; j < MAXJ; ++j) slot(j);
                }
                return bsum(acc, L);
            };
            // Round 6: once two halvings have failed -- the long runs of a stalled iteration, ~50 halvings each -- the NEXT KL halvings
            // are evaluated in ONE pass over the thread's outcomes (one read of the six per-slot arrays, which live in scratch for the
            // Pauli in-basis: 1.4 KB per thread = 0.7 MB per workgroup through L2 / HBM per evaluation, what bounded this phase) and the
            // loop below walks them in order.  Every cost is the same sum in the same order as cost_at's (per-thread terms by slot, the
            // wavefront reduction, the sixteen partials added in wavefront order): the accepted step and every count are
            // BIT-IDENTICAL to the one-at-a-time loop; what changes is that up to KL - 1 evaluations behind the accepted one are wasted.
            constexpr int KL = 4;       // (eight: 168.2 against 162.4 ms for the Pauli in-basis -- sixteen logarithm chains in flight spill; two: 163.8)
            auto cost_ladder = [&](double a0, double (&out)[KL]) __attribute__((always_inline)) {
                double acc[KL];
#pragma unroll
                for (int k = 0; k < KL; ++k) acc[k] = 0.0;
                auto slot = [&](int j) __attribute__((always_inline)) {
                    if (t * MAXJ + j < m) {
                        const double ep_ = pep[j], em_ = pem[j], up_ = pup[j], um_ = pum[j], np_ = npl[j], nm_ = nmi[j];
                        double a = a0;
#pragma unroll
                        for (int k = 0; k < KL; ++k) {
                            double pp = fma(a, up_, ep_), pm = fma(a, um_, em_);
                            pp = pp < EPS ? EPS : pp; pm = pm < EPS ? EPS : pm;
                            acc[k] -= np_ * fast_log_pos(pp) + nm_ * fast_log_pos(pm);
                            a *= 0.5;
                            __builtin_amdgcn_sched_barrier(0);      // one step's two logarithms at a time: interleaved chains spill
                        }
                    }
                };
                if constexpr (MAXJ <= 4) {                      // register-resident slot arrays: static indices
#pragma unroll
                    for (int j = 0; j < MAXJ; ++j) slot(j);
                } else {                                        // slot arrays in scratch anyway: a loop
#pragma unroll 1
                    for (int j = 0; j < MAXJ; ++j) slot(j);
                }
#pragma unroll
                for (int k = 0; k < KL; ++k) acc[k] = wave_sum(acc[k]);
                FBX_BLOCK_SYNC();                               // earlier readers of `red` are done
                if ((t & 63) == 0) {
#pragma unroll
                    for (int k = 0; k < KL; ++k) L.red[k * (NT / 64) + (t >> 6)] = acc[k];
                }
                FBX_BLOCK_SYNC();
#pragma unroll
                for (int k = 0; k < KL; ++k) {
                    double sum = 0.0;
#pragma unroll
                    for (int w = 0; w < NT / 64; ++w) sum += L.red[k * (NT / 64) + w];
                    out[k] = sum;
                }
            };
            new_cost = cost_at(alpha); ++cost_evals;
            double change = GAMMA * alpha * ipr;
            int fails = 0;
            while (new_cost > old_cost + change) {
                if (fails >= 2) {
                    double lad[KL];
                    cost_ladder(0.5 * alpha, lad); cost_evals += KL;
                    bool stop = false;
#pragma unroll
                    for (int k = 0; k < KL; ++k) {
                        if (!stop) {
                            alpha *= 0.5; change *= 0.5; new_cost = lad[k]; ++backtracks;
                            stop = alpha < ALPHA_MIN || !(new_cost > old_cost + change);
                        }
                    }
                    if (stop) break;
                    continue;
                }
                alpha *= 0.5; change *= 0.5;
                new_cost = cost_at(alpha); ++cost_evals;
                ++backtracks; ++fails;
                if (alpha < ALPHA_MIN) break;
            }
        }
        est = blk_axpy(est, alpha, upd);
        outer_step = alpha * sqrt(bsum(blk_norm2(upd), L));
        if (trace_out && iters < trace_iters && t == 0) {
            int* tr = trace_out + ((size_t)item * trace_iters + iters) * 2;
            tr[0] = dyk - dyk_before; tr[1] = backtracks - bt_before;
        }
        ++iters;
        PH_STOP(pc, 5);
        if (mode == FBX_MODE_CONVERGE) {
            if (!(old_cost - new_cost >= STOP)) break;          // tomography.py:589; a NaN cost also ends the loop
            if (max_iters > 0 && iters >= max_iters) break;
        }
        old_cost = new_cost;
    }
    {
        const int I = t / NB, J = t % NB;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = 2 * I + (e >> 1), col = 2 * J + (e & 1);
            double* o = choi_out + ((item * D + row) * D + col) * 2;
            o[0] = est.re[e]; o[1] = est.im[e];
        }
    }
#ifdef FBX_PHASE_TIMERS
    if (t == 0 && phase_out) for (int i = 0; i < FBX_NPHASE; ++i) phase_out[item * FBX_NPHASE + i] = pc.acc[i];
#endif
    if (t == 0) {
        if (iters_out) iters_out[item] = iters;
        if (dykstra_out) dykstra_out[item] = dyk;
        if (backtracks_out) backtracks_out[item] = backtracks;
        if (cost_out) cost_out[item] = have_cost ? new_cost : 0.0;
        if (sweeps_out) {     // work_out[4]: Jacobi sweeps, eigenvalue terms rebuilt, cost evaluations, 0
            sweeps_out[4 * item] = sweeps; sweeps_out[4 * item + 1] = L.terms;
            sweeps_out[4 * item + 2] = cost_evals; sweeps_out[4 * item + 3] = 0;
        }
    }
}

template <int MAXJ>
static int launch3(const fbx_design* des, int64_t B, const double* e, const double* c, int tp, int mode,
                   int max_iters, double* choi, int32_t* it, int32_t* dy, int32_t* bt, double* cost, int32_t* sw,
                   const PgdbExtras& ex) {
    const size_t lds = p3::Lds::bytes();
    if ((size_t)des->dev.S * p3::D * sizeof(double) > 2 * sizeof(cplx) * p3::D * p3::D) {
        set_error("fbx_pgdb_process: too many distinct input states for the 3-qubit kernel");
        return FBX_ERR_UNSUPPORTED;
    }
    auto kern = pgdb3_kernel<MAXJ>;
    FBX_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // per-workgroup store of Dykstra eigenvector bases (BASIS_CAP x 64 KiB); batches are processed in
    // chunks so the store stays bounded (512 workgroups = 768 MiB)
    constexpr int64_t CHUNK = 512;
#define FBX_BASIS_CAP3 24
    constexpr int BASIS_CAP = FBX_BASIS_CAP3;   // Dykstra iterations per projection with a stored basis (64 KiB each)
    // (a grow-only workspace of the calling thread, released by fbx_release_workspace)
    void* w = nullptr;
    const size_t in_flight = (size_t)(B < CHUNK ? B : CHUNK);
    const size_t basis_bytes = sizeof(cplx) * p3::D * p3::D * in_flight * BASIS_CAP;
    { const int rc = workspace(WS_PGDB3_BASIS, basis_bytes + sizeof(double) * Park3<MAXJ>::doubles() * in_flight, &w); if (rc) return rc; }
    cplx* scratch = (cplx*)w;
    cplx* basis = scratch;                   // (`scratch` itself only tells the kernel that warm starts are on)
    double* park = (double*)((char*)w + basis_bytes);
    const size_t m = des->dev.m, DD = (size_t)p3::D * p3::D;
    DesignDev dev = des->dev;
    dev.eig_rel_tol = ex.eig_rel_tol >= 0.0 ? ex.eig_rel_tol : option_pgdb_eig_rel_tol(3);
    for (int64_t b0 = 0; b0 < B; b0 += CHUNK) {
        const int64_t nb = B - b0 < CHUNK ? B - b0 : CHUNK;
        hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(1024), lds, ex.launch_stream ? ex.launch_stream : stream(), dev, (long long)nb,
                           e + b0 * m, c + b0 * m, tp, mode, max_iters, choi + b0 * DD * 2, it ? it + b0 : nullptr,
                           dy ? dy + b0 : nullptr, bt ? bt + b0 : nullptr, cost ? cost + b0 : nullptr, scratch,
                           FBX_PHASE_OUT3(b0), sw ? sw + 4 * b0 : nullptr, basis, BASIS_CAP,
                           ex.trace ? ex.trace + (size_t)b0 * ex.trace_iters * 2 : nullptr, ex.trace_iters, park);
    }
    FBX_HIP(hipGetLastError());
    return FBX_OK;
}

// _cost / _grad_cost (tomography.py:597-633) for 3 qubits as a function of their own (fbx_pgdb_cost_grad; the 1- / 2-qubit form and
// the reasons are in fbx_pgdb.hip): one evaluation with the device functions of pgdb3_kernel -- choi_to_pauli, the matrix-core
// table product, the run-wise deterministic identity row of W, gradient_coefficients, pauli_to_choi.  Any number of settings:
// thread t owns the contiguous run [t MJ, (t + 1) MJ) of the state-grouped settings, MJ = ceil(m / 1024) at run time; eta = n / p
// waits in `eta` (HBM, [2 m] per item) between the pass that consumes the table and the pass that fills W in its place.
__global__ void __launch_bounds__(1024)
pgdb3_cost_grad_kernel(DesignDev des, long long B, const double* __restrict__ nvec, const double* __restrict__ choi_in, double eps,
                       double* __restrict__ cost_out, double* __restrict__ grad_out, double* __restrict__ eta) {
    using namespace p3;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Lds L; L.carve(smem);
    PhaseClock pc; pc.reset(); L.pc = &pc;
    const int t = threadIdx.x;
    const long long item = blockIdx.x;
    const int m = des.m, S = des.S, MJ = (m + NT - 1) / NT;
    const bool unit_coefs = des.unit_coefs != 0;
    const double half_dd = 0.5 / (double)(d * d);
    const int I = t / NB, J = t % NB;
    Blk est;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int row = 2 * I + (e >> 1), col = 2 * J + (e & 1);
        const double* o = choi_in + ((item * D + row) * D + col) * 2;
        est.re[e] = o[0]; est.im[e] = o[1];
    }
    choi_to_pauli(est, L, t);
    predict_table(des, L, t);
    const double* nv = nvec + item * 2 * m;
    double* et = eta + item * 2 * m;
    double acc = 0.0;
    for (int j = 0; j < MJ; ++j) {
        const int g = t * MJ + j;
        if (g < m) {
            const uint32_t w = des.sp[g];
            const int s = w >> 16, p = w & 0xffff, k = des.order[g];
            const double cf = unit_coefs ? 1.0 : des.coef[g];
            const double tr = L.T[s * D], ex = cf * L.T[s * D + p];
            double pp = (tr + ex) * half_dd, pm = (tr - ex) * half_dd;
            pp = pp < eps ? eps : pp; pm = pm < eps ? eps : pm;
            const double np_ = nv[2 * k], nm_ = nv[2 * k + 1];
            acc -= np_ * fast_log_pos(pp) + nm_ * fast_log_pos(pm);
            et[2 * g] = np_ / pp; et[2 * g + 1] = nm_ / pm;
        }
    }
    acc = bsum(acc, L);
    if (t == 0 && cost_out) cost_out[item] = acc;
    if (!grad_out) return;
    FBX_BLOCK_SYNC();                              // T fully consumed
    double* W = L.T;
    double* pfirst = L.Rt + 768;                  // overlays R (dead here), past the small scratch
    double* plast = pfirst + NT;
    int* sfirst = (int*)(plast + NT);
    int* slast = sfirst + NT;
    for (int idx = t; idx < D * S; idx += NT) W[idx] = 0.0;
    sfirst[t] = -1; slast[t] = -1; pfirst[t] = 0.0; plast[t] = 0.0;
    FBX_BLOCK_SYNC();
    {
        int run_state = -1; double run = 0.0; bool first_done = false;
        for (int j = 0; j < MJ; ++j) {
            const int g = t * MJ + j;
            if (g < m) {
                const uint32_t w = des.sp[g];
                const int s = w >> 16, p = w & 0xffff;
                const double cf = unit_coefs ? 1.0 : des.coef[g];
                const double ep = et[2 * g], em = et[2 * g + 1];      // (written by this very thread)
                atomicAdd(&W[p * S + s], cf * 0.5 * (ep - em));
                if (s != run_state) {
                    if (run_state >= 0) {
                        if (!first_done) { pfirst[t] = run; sfirst[t] = run_state; first_done = true; }
                        else atomicAdd(&W[run_state], run);
                    }
                    run_state = s; run = 0.0;
                }
                run += 0.5 * (ep + em);
            }
        }
        if (run_state >= 0) {
            if (!first_done) { pfirst[t] = run; sfirst[t] = run_state; }
            else { plast[t] = run; slast[t] = run_state; }
        }
    }
    FBX_BLOCK_SYNC();
    for (int s = t; s < S; s += NT) {
        const int g0 = des.sptr[s], g1 = des.sptr[s + 1];
        if (g1 > g0) {
            double a = 0.0;
            for (int tt = g0 / MJ; tt <= (g1 - 1) / MJ; ++tt) {
                if (sfirst[tt] == s) a += pfirst[tt];
                if (slast[tt] == s) a += plast[tt];
            }
            W[s] += a;
        }
    }
    FBX_BLOCK_SYNC();
    gradient_coefficients(des, L, W, t);
    const Blk grad = pauli_to_choi(L, t);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int row = 2 * I + (e >> 1), col = 2 * J + (e & 1);
        double* o = grad_out + ((item * D + row) * D + col) * 2;
        o[0] = grad.re[e]; o[1] = grad.im[e];
    }
}

int pgdb3_cost_grad_launch(const fbx_design* des, int64_t B, const double* nvec, const double* choi, double eps, double* cost,
                           double* grad) {
    const size_t lds = p3::Lds::bytes();
    if ((size_t)des->dev.S * p3::D * sizeof(double) > 2 * sizeof(cplx) * p3::D * p3::D) {
        set_error("fbx_pgdb_cost_grad: too many distinct input states for the 3-qubit kernel");
        return FBX_ERR_UNSUPPORTED;
    }
    DevBuf eta;
    { const int rc = eta.alloc(sizeof(double) * 2 * (size_t)B * des->dev.m); if (rc) return rc; }
    FBX_HIP(hipFuncSetAttribute((const void*)pgdb3_cost_grad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(pgdb3_cost_grad_kernel, dim3((unsigned)B), dim3(1024), lds, stream(), des->dev, (long long)B, nvec, choi, eps, cost,
                       grad, eta.as<double>());
    FBX_HIP(hipGetLastError());
    return FBX_OK;
}

// ---- 3-qubit Choi projections (fbx_proj_choi) and linear inversion (fbx_linv_process) on the same
// 1024-thread building blocks
__global__ void __launch_bounds__(1024)
proj_choi3_kernel(int kind, long long B, const double* __restrict__ in, double* __restrict__ out,
                  int* __restrict__ iters_out) {
    using namespace p3;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Lds L; L.carve(smem);
    const int t = threadIdx.x;
    const long long item = blockIdx.x;
    const int I = t / NB, J = t % NB;
    Blk x;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int row = 2 * I + (e >> 1), col = 2 * J + (e & 1);
        const double* p = in + ((item * D + row) * D + col) * 2;
        x.re[e] = p[0]; x.im[e] = p[1];
    }
    int iters = 0, sweeps = 0;
    Blk y;
    if (kind == FBX_PROJ_CP) y = proj_cp(x, L, t, sweeps);
    else if (kind == FBX_PROJ_TP) y = proj_tp(x, L, t);
    else if (kind == FBX_PROJ_TNI) y = proj_tni(x, L, t, sweeps);
    else y = proj_physical(x, kind == FBX_PROJ_PHYSICAL_TP, L, t, iters, sweeps, nullptr);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int row = 2 * I + (e >> 1), col = 2 * J + (e & 1);
        double* p = out + ((item * D + row) * D + col) * 2;
        p[0] = y.re[e]; p[1] = y.im[e];
    }
    if (t == 0 && iters_out) iters_out[item] = iters;
}

__global__ void __launch_bounds__(1024)
linv_process3_kernel(DesignDev des, long long B, const double* __restrict__ expect, double* __restrict__ out) {
    using namespace p3;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Lds L; L.carve(smem);
    const int t = threadIdx.x;
    const long long item = blockIdx.x;
    for (int idx = t; idx < D * D; idx += NT) {       // Rt[j * 64 + i] = R[i][j] (+ the identity term)
        const int i = idx % D, j = idx / D;
        double acc = 0.0;
        for (int g = des.pptr[i]; g < des.pptr[i + 1]; ++g)
            acc += expect[item * des.m + des.porder[g]] * des.pinvT[(size_t)g * D + j];
        L.Rt[idx] = acc + ((idx == 0) ? 1.0 : 0.0);
    }
    FBX_BLOCK_SYNC();
    const Blk c = pauli_to_choi(L, t);
    const int I = t / NB, J = t % NB;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int row = 2 * I + (e >> 1), col = 2 * J + (e & 1);
        double* p = out + ((item * D + row) * D + col) * 2;
        p[0] = c.re[e]; p[1] = c.im[e];
    }
}

int proj_choi3_launch(int kind, int64_t B, const double* d_in, double* d_out, int32_t* d_iters) {
    const size_t lds = p3::Lds::bytes();
    FBX_HIP(hipFuncSetAttribute((const void*)proj_choi3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(proj_choi3_kernel, dim3((unsigned)B), dim3(1024), lds, stream(), kind, (long long)B, d_in, d_out, d_iters);
    FBX_HIP(hipGetLastError());
    return FBX_OK;
}
int linv_process3_launch(const fbx_design* des, int64_t B, const double* d_expect, double* d_out) {
    const size_t lds = p3::Lds::bytes();
    FBX_HIP(hipFuncSetAttribute((const void*)linv_process3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(linv_process3_kernel, dim3((unsigned)B), dim3(1024), lds, stream(), des->dev, (long long)B, d_expect, d_out);
    FBX_HIP(hipGetLastError());
    return FBX_OK;
}

// called from fbx_pgdb.hip's dispatcher
int pgdb3_dispatch(const fbx_design* des, int64_t B, const double* e, const double* c, int tp, int mode,
                   int max_iters, double* choi, int32_t* it, int32_t* dy, int32_t* bt, double* cost, int32_t* sw,
                   const PgdbExtras& ex) {
    const int m = des->dev.m;
    mode &= 0xff;          // FBX_MODE_LS_REFERENCE: this kernel's line search always evaluates the full cost with the rounded test
    if (m <= 4096) return launch3<4>(des, B, e, c, tp, mode, max_iters, choi, it, dy, bt, cost, sw, ex);
    if (m <= 14336) return launch3<14>(des, B, e, c, tp, mode, max_iters, choi, it, dy, bt, cost, sw, ex);
    // Merged / repeated datasets (the reference takes any result list, tomography.py:494-539): the same kernel with 32 / 64 outcome
    // slots per thread.  Their per-slot arrays no longer fit 128 registers and live in scratch: slow (a few times the resident
    // instantiations per outer iteration outside the projection) and complete up to 65 536 settings.
    if (m <= 32768) return launch3<32>(des, B, e, c, tp, mode, max_iters, choi, it, dy, bt, cost, sw, ex);
    if (m <= 65536) return launch3<64>(des, B, e, c, tp, mode, max_iters, choi, it, dy, bt, cost, sw, ex);
    set_error("fbx_pgdb_process: 3-qubit designs are limited to 65536 settings");
    return FBX_ERR_UNSUPPORTED;
}

}  // namespace fbx
