// fbx_state.hip -- batched state-tomography estimators and state measures on d x d (d = 2^n,
// n <= 3) density matrices.  One 64-lane wavefront per item; lane t < d*d owns matrix entry
// (t / d, t % d); Pauli expectations and the R operator are evaluated through the sparsity of
// the Pauli matrices (P_p[r][c] != 0 iff c = r ^ x_p), never by building d x d operators.
//
// Reference functions (file:line under forest/benchmarking/):
//   linear_inv_state_estimate        tomography.py:130-165
//   iterative_mle_state_estimate     tomography.py:168-270   (+ _R, :273-338)
//   state_log_likelihood             tomography.py:341-375
//   project_state_matrix_to_physical operator_tools/project_state_matrix.py:6-52
//   purity / fidelity / trace_distance / hilbert_schmidt_ip   distance_measures.py:14-114,198-216
//   sqrtm_psd                        operator_tools/calculational.py:77-91
#include "fbx_eigh64.hpp"
#include <hip/hip_cooperative_groups.h>
#include <cfloat>
#include <type_traits>
#include <cstdlib>
#include <algorithm>
#include <vector>
#include <cstring>

namespace fbx {

template <int NQ>
struct StateLds {
    static constexpr int d = 1 << NQ, D = d * d;
    cplx *rho, *U, *tmp, *aux;     // [d*d] row-major
    cplx *Ms, *Vs;                 // [d*d] Jacobi layout
    JRec* rec;                     // [d/2 + 1]
    double *w, *r, *lam;           // [D], [D], [d]
    double *hs, *hd;               // [m] per-setting scratch
    __host__ __device__ static constexpr size_t bytes(int m) {
        return sizeof(cplx) * 6 * D + sizeof(JRec) * (d / 2 + 1) + sizeof(double) * (2 * D + d + 2 * (size_t)m) + 64;
    }
    // The per-setting scratch (16 bytes per setting) is staged in LDS up to 64 KiB; a design beyond that -- a state-tomography
    // dataset repeated or merged some 60 times over -- takes the streamed form of r_operator_elem, which needs none.
    __host__ __device__ static constexpr bool staged(int m) { return bytes(m) <= 64 * 1024; }
    __host__ __device__ static constexpr size_t launch_bytes(int m) { return bytes(staged(m) ? m : 0); }
    __device__ void carve(char* p, int m) {
        rho = (cplx*)p; p += sizeof(cplx) * D;  U = (cplx*)p; p += sizeof(cplx) * D;
        tmp = (cplx*)p; p += sizeof(cplx) * D;  aux = (cplx*)p; p += sizeof(cplx) * D;
        Ms = (cplx*)p; p += sizeof(cplx) * D;   Vs = (cplx*)p; p += sizeof(cplx) * D;
        rec = (JRec*)p; p += sizeof(JRec) * (d / 2 + 1);
        w = (double*)p; p += sizeof(double) * D; r = (double*)p; p += sizeof(double) * D;
        lam = (double*)p; p += sizeof(double) * d;
        hs = (double*)p; p += sizeof(double) * m; hd = (double*)p;
    }
};

// P_p[row][row ^ x] = i^{ny} (-1)^{popc((row ^ x) & z)}
template <int NQ>
__device__ __forceinline__ void pauli_entry(int x, int z, int ny, int row, int& col, int& ph, int& neg) {
    col = row ^ x; ph = ny & 3; neg = __popc(col & z) & 1;
}

// r[p] = Re tr(P_p rho) for every Pauli index p (lanes p < D)
template <int NQ>
__device__ void pauli_expectations(const cplx* rho, double* r, int lane) {
    constexpr int d = 1 << NQ, D = d * d;
    if (lane < D) {
        int x, z, ny; pauli_masks<NQ>(lane, x, z, ny);
        double acc = 0.0;
#pragma unroll
        for (int row = 0; row < d; ++row) {
            int col, ph, neg; pauli_entry<NQ>(x, z, ny, row, col, ph, neg);
            const cplx v = rho[col * d + row];                 // rho[col][row]
            const double t = (ph == 0) ? v.re : (ph == 1) ? -v.im : (ph == 2) ? -v.re : v.im;
            acc += neg ? -t : t;
        }
        r[lane] = acc;
    }
}

// element (row, col) of  w0 * I + sum_p w[p] P_p
template <int NQ>
__device__ __forceinline__ cplx pauli_synthesis(const double* w, double w0, int row, int col) {
    constexpr int d = 1 << NQ;
    const int x = row ^ col;
    double re = (row == col) ? w0 : 0.0, im = 0.0;
#pragma unroll
    for (int z = 0; z < d; ++z) {
        const int p = pauli_index<NQ>(x, z);
        const int ph = __popc(x & z) & 3, neg = __popc(col & z) & 1;
        double v = w[p]; v = neg ? -v : v;
        if (ph == 0) re += v; else if (ph == 1) im += v; else if (ph == 2) re -= v; else im -= v;
    }
    cplx o; o.re = re; o.im = im; return o;
}

// Round 6: the two Pauli passes of the iterative-MLE loop with their per-lane index and sign logic evaluated ONCE per reconstruction.
// The state kernels are bound by vector-instruction issue (profiles/r06: 46-48 % of their vector instructions are fp64 arithmetic,
// the rest phase selects, index arithmetic and reductions), and what a lane selects in pauli_expectations / pauli_synthesis depends on
// nothing but the lane: lane p reads the SAME component (real or imaginary, by the number of Y factors of its Pauli) of d fixed
// entries with fixed signs; entry (row, col) adds d fixed weights with fixed signs to its real or imaginary part.  As tables: one
// 8-byte LDS read + one FMA with a +-1 constant per term (expectations), one read + two FMAs with constants in {0, +-1} (synthesis).
// A product with +-1 is exact and a term with coefficient 0 leaves a non-negative-zero accumulator untouched: BIT-IDENTICAL to the
// select forms above (tests/test_state_gpu.py holds both against the reference fixtures; scripts/compare_libs.py the hashes).
template <int NQ>
struct PauliTables {
    static constexpr int d = 1 << NQ;
    int eoff[d]; double esg[d];             // r[p] = sum_row esg[row] * ((double*)rho)[eoff[row]]
    int sp[d]; double sa[d], sb[d];         // entry: re += sa[z] * w[sp[z]], im += sb[z] * w[sp[z]]
    // lane `p` of the expectation pass, entry (row, col) of the synthesis pass
    __device__ __forceinline__ void init(int p, int row, int col) {
        int x, z, ny; pauli_masks<NQ>(p, x, z, ny);
#pragma unroll
        for (int r = 0; r < d; ++r) {
            int c, ph, neg; pauli_entry<NQ>(x, z, ny, r, c, ph, neg);
            const double s = neg ? -1.0 : 1.0;
            eoff[r] = 2 * (c * d + r) + (ph & 1);                               // ph 0 / 2: real part, 1 / 3: imaginary part
            esg[r] = (ph == 0 || ph == 3) ? s : -s;
        }
        const int xs = row ^ col;
#pragma unroll
        for (int zz = 0; zz < d; ++zz) {
            sp[zz] = pauli_index<NQ>(xs, zz);
            const int ph = __popc(xs & zz) & 3, neg = __popc(col & zz) & 1;
            const double s = neg ? -1.0 : 1.0;
            sa[zz] = ph == 0 ? s : ph == 2 ? -s : 0.0;
            sb[zz] = ph == 1 ? s : ph == 3 ? -s : 0.0;
        }
    }
    __device__ __forceinline__ double expectation(const cplx* rho) const {
        const double* f = reinterpret_cast<const double*>(rho);
        double acc = 0.0;
#pragma unroll
        for (int r = 0; r < d; ++r) acc = fma(esg[r], f[eoff[r]], acc);
        return acc;
    }
    __device__ __forceinline__ cplx synthesis(const double* w, double w0, bool on_diagonal) const {
        double re = on_diagonal ? w0 : 0.0, im = 0.0;
#pragma unroll
        for (int zz = 0; zz < d; ++zz) { const double v = w[sp[zz]]; re = fma(sa[zz], v, re); im = fma(sb[zz], v, im); }
        cplx o; o.re = re; o.im = im; return o;
    }
};

// The same tables with the signs as BIT MASKS (one 32-bit word per pass instead of d doubles) and the synthesis terms ordered by the part
// they feed: what the 3-qubit kernel uses -- with the double-valued coefficients above it holds 236 registers (two wavefronts per SIMD),
// with these ~110 (four).  Entry (row, col), x = row ^ col: the weights with even popc(x & z) feed the real part, those with odd parity the
// imaginary part -- for x != 0 half of the z each, in ascending z inside each half; on the diagonal (x = 0) all d feed the real part.  The
// list holds the real-part terms first, in ascending z, then the others: the first d / 2 terms always go to the real accumulator, the
// second half continues the SAME accumulator on the diagonal and starts the imaginary one elsewhere, so every sum keeps the order of the
// select form (bit-identical).  A sign flip is one v_xor on the high word.
template <int NQ, bool FMA_SIGNS = false>
struct PauliTablesLean {
    static constexpr int d = 1 << NQ;
    int eoff[d]; unsigned esign;            // bit r: the expectation term of row r enters negated
    int sp[d]; unsigned ssign;              // bit k: synthesis term k enters negated
    double esg[FMA_SIGNS ? d : 1], ssg[FMA_SIGNS ? d : 1];      // FMA_SIGNS: the same signs as +-1.0 (one FMA per term instead of xor + add)
    __device__ __forceinline__ void init(int p, int row, int col) {
        int x, z, ny; pauli_masks<NQ>(p, x, z, ny);
        esign = 0u;
#pragma unroll
        for (int r = 0; r < d; ++r) {
            int c, ph, neg; pauli_entry<NQ>(x, z, ny, r, c, ph, neg);
            eoff[r] = 2 * (c * d + r) + (ph & 1);
            const bool minus = (ph == 0 || ph == 3) ? neg != 0 : neg == 0;
            esign |= (minus ? 1u : 0u) << r;
        }
        const int xs = row ^ col;
        ssign = 0u;
#pragma unroll
        for (int q = 0; q < d; ++q) sp[q] = 0;
        int k = 0;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
            for (int zz = 0; zz < d; ++zz) {
                const int ph = __popc(xs & zz) & 3, neg = __popc(col & zz) & 1;
                if ((ph & 1) == pass) {
                    const bool minus = (ph >= 2) != (neg != 0);
                    // (k is a compile-time-unknown position: the writes below are selects over the d slots, once per reconstruction)
                    const int pidx = pauli_index<NQ>(xs, zz);
#pragma unroll
                    for (int q = 0; q < d; ++q) sp[q] = (q == k) ? pidx : sp[q];
                    ssign |= (minus ? 1u : 0u) << k;
                    ++k;
                }
            }
        }
        if constexpr (FMA_SIGNS) {
#pragma unroll
            for (int q = 0; q < d; ++q) { esg[q] = ((esign >> q) & 1u) ? -1.0 : 1.0; ssg[q] = ((ssign >> q) & 1u) ? -1.0 : 1.0; }
        }
    }
    static __device__ __forceinline__ double signed_(double v, unsigned bits, int k) {
        return __hiloint2double(__double2hiint(v) ^ (int)(((bits >> k) & 1u) << 31), __double2loint(v));
    }
    __device__ __forceinline__ double expectation(const cplx* rho) const {
        const double* f = reinterpret_cast<const double*>(rho);
        double acc = 0.0;
#pragma unroll
        for (int r = 0; r < d; ++r) { if constexpr (FMA_SIGNS) acc = fma(esg[r], f[eoff[r]], acc); else acc += signed_(f[eoff[r]], esign, r); }
        return acc;
    }
    __device__ __forceinline__ cplx synthesis(const double* w, double w0, bool on_diagonal) const {
        double re = on_diagonal ? w0 : 0.0;
#pragma unroll
        for (int k = 0; k < d / 2; ++k) { if constexpr (FMA_SIGNS) re = fma(ssg[k], w[sp[k]], re); else re += signed_(w[sp[k]], ssign, k); }
        double acc = on_diagonal ? re : 0.0;                  // the diagonal continues its real sum, every other entry starts the imaginary one
#pragma unroll
        for (int k = d / 2; k < d; ++k) { if constexpr (FMA_SIGNS) acc = fma(ssg[k], w[sp[k]], acc); else acc += signed_(w[sp[k]], ssign, k); }
        cplx o; o.re = on_diagonal ? acc : re; o.im = on_diagonal ? 0.0 : acc; return o;
    }
};

// One setting of the design held by a lane for the whole reconstruction (designs of at most 64
// settings -- every state-tomography design of the reference has 4^n - 1 <= 63): Pauli index,
// coefficient and measured expectation are fetched from HBM once instead of once per iteration.
struct LaneSetting { int p; double cf, e; bool valid; };
template <int NQ>
__device__ __forceinline__ LaneSetting load_lane_setting(const DesignDev& des, const double* __restrict__ e, int lane) {
    LaneSetting s; s.valid = des.m <= 64 && lane < des.m; s.p = 0; s.cf = 1.0; s.e = 0.0;
    if (s.valid) { s.p = des.sp[lane] & 0xffff; s.cf = des.unit_coefs ? 1.0 : des.coef[lane]; s.e = e[des.order[lane]]; }
    return s;
}

// R operator of tomography.py:273-338 for the state in L.rho; result element of this lane.
// the register-resident-settings form of r_operator_elem below (designs of at most 64 settings) with the Pauli passes as tables: what
// the iterative-MLE loop runs.  Same operations in the same order.
template <int NQ, class Tables>
__device__ __forceinline__ cplx r_operator_tab(const DesignDev& des, StateLds<NQ>& L, int lane, const LaneSetting& mine,
                                               const Tables& tab) {
    constexpr int d = 1 << NQ, D = d * d;
    const int m = des.m;
    if (lane < D) { L.r[lane] = tab.expectation(L.rho); L.w[lane] = 0.0; }
    FBX_WAVE_SYNC();
    double s0 = 0.0;
    if (mine.valid) {
        const double pe = mine.cf * L.r[mine.p];
        const double gp = ((1.0 + mine.e) * 0.5) / ((1.0 + pe) * 0.5 + DBL_MIN);
        const double gm = ((1.0 - mine.e) * 0.5) / ((1.0 - pe) * 0.5 + DBL_MIN);
        s0 = 0.5 * (gp + gm);
        atomicAdd(&L.w[mine.p], mine.cf * 0.5 * (gp - gm));
    }
    s0 = wave_sum(s0);
    FBX_WAVE_SYNC();
    if (lane < D) L.w[lane] = L.w[lane] / m;
    FBX_WAVE_SYNC();
    cplx out; out.re = 0.0; out.im = 0.0;
    if (lane < D) out = tab.synthesis(L.w, s0 / m + 0.0, lane / d == lane % d);
    return out;
}

template <int NQ>
__device__ cplx r_operator_elem(const DesignDev& des, const double* __restrict__ e, StateLds<NQ>& L, int lane,
                                const LaneSetting* mine = nullptr) {
    constexpr int d = 1 << NQ, D = d * d;
    const int m = des.m;
    pauli_expectations<NQ>(L.rho, L.r, lane);
    if (mine && m <= 64) {                    // register-resident settings, weights through LDS atomics
        if (lane < D) L.w[lane] = 0.0;
        FBX_WAVE_SYNC();
        double s0 = 0.0;
        if (mine->valid) {
            const double pe = mine->cf * L.r[mine->p];
            const double gp = ((1.0 + mine->e) * 0.5) / ((1.0 + pe) * 0.5 + DBL_MIN);
            const double gm = ((1.0 - mine->e) * 0.5) / ((1.0 - pe) * 0.5 + DBL_MIN);
            s0 = 0.5 * (gp + gm);
            atomicAdd(&L.w[mine->p], mine->cf * 0.5 * (gp - gm));
        }
        s0 = wave_sum(s0);
        FBX_WAVE_SYNC();
        if (lane < D) L.w[lane] = L.w[lane] / m;
        FBX_WAVE_SYNC();
        cplx out; out.re = 0.0; out.im = 0.0;
        if (lane < D) out = pauli_synthesis<NQ>(L.w, s0 / m + 0.0, lane / d, lane % d);
        return out;
    }
    if (!StateLds<NQ>::staged(m)) {
        // Streamed (any number of settings; the reference loops over whatever list it is given, tomography.py:326-336): every lane
        // walks its settings straight from HBM and adds their weights to w[p] with LDS atomics (one wavefront: the same order
        // in every run) -- no per-setting scratch.
        if (lane < D) L.w[lane] = 0.0;
        FBX_WAVE_SYNC();
        double s0 = 0.0;
        for (int g = lane; g < m; g += 64) {
            const int p = des.sp[g] & 0xffff;
            const double cf = des.unit_coefs ? 1.0 : des.coef[g];
            const double me = e[des.order[g]], pe = cf * L.r[p];
            const double gp = ((1.0 + me) * 0.5) / ((1.0 + pe) * 0.5 + DBL_MIN);
            const double gm = ((1.0 - me) * 0.5) / ((1.0 - pe) * 0.5 + DBL_MIN);
            s0 += 0.5 * (gp + gm);
            atomicAdd(&L.w[p], cf * 0.5 * (gp - gm));
        }
        s0 = wave_sum(s0);
        FBX_WAVE_SYNC();
        if (lane < D) L.w[lane] = L.w[lane] / m;
        FBX_WAVE_SYNC();
        cplx out; out.re = 0.0; out.im = 0.0;
        if (lane < D) out = pauli_synthesis<NQ>(L.w, s0 / m + 0.0, lane / d, lane % d);
        return out;
    }
    FBX_WAVE_SYNC();
    double s0 = 0.0;
    for (int g = lane; g < m; g += 64) {
        const int p = des.sp[g] & 0xffff;
        const double cf = des.unit_coefs ? 1.0 : des.coef[g];
        const double me = e[des.order[g]], pe = cf * L.r[p];
        const double gp = ((1.0 + me) * 0.5) / ((1.0 + pe) * 0.5 + DBL_MIN);
        const double gm = ((1.0 - me) * 0.5) / ((1.0 - pe) * 0.5 + DBL_MIN);
        L.hs[g] = 0.5 * (gp + gm);
        L.hd[g] = cf * 0.5 * (gp - gm);
        s0 += 0.5 * (gp + gm);
    }
    s0 = wave_sum(s0);
    FBX_WAVE_SYNC();
    if (lane < D) {
        double acc = 0.0;
        for (int g = 0; g < m; ++g) if ((int)(des.sp[g] & 0xffff) == lane) acc += L.hd[g];
        L.w[lane] = acc / m;
    }
    FBX_WAVE_SYNC();
    cplx out; out.re = 0.0; out.im = 0.0;
    if (lane < D) out = pauli_synthesis<NQ>(L.w, s0 / m + 0.0, lane / d, lane % d);
    // identity-observable settings contribute through w[0] as well as through s0: P_0 = I
    return out;
}

// Hermitian function of a d x d matrix staged row-major in `src`: out = V f(lambda) V^H with
// f selected by `fn` (0: log, 1: pseudo-inverse, 2: sqrt(max(.,0))); eigenvalues left in L.lam
template <int NQ>
__device__ void herm_function(const cplx* src, cplx* dst, int fn, StateLds<NQ>& L, int lane, bool lower_only) {
    constexpr int d = 1 << NQ, NB = d / 2;
    Blk h = blk_zero();
    if (lane < NB * NB) {
        const int I = lane / NB, J = lane % NB;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int r = 2 * I + (e >> 1), c = 2 * J + (e & 1);
            if (lower_only) {                       // scipy / numpy eigh read the lower triangle
                if (r > c) { h.re[e] = src[r * d + c].re; h.im[e] = src[r * d + c].im; }
                else if (r < c) { h.re[e] = src[c * d + r].re; h.im[e] = -src[c * d + r].im; }
                else { h.re[e] = src[r * d + c].re; h.im[e] = 0.0; }
            } else {
                const cplx a = src[r * d + c], b = src[c * d + r];
                h.re[e] = 0.5 * (a.re + b.re); h.im[e] = 0.5 * (a.im - b.im);
            }
        }
    }
    FBX_WAVE_SYNC();
    sys_store<d>(L.Ms, lane, h);
    FBX_WAVE_SYNC();
    jacobi_eigh_lds<d>(L.Ms, L.Vs, L.rec, lane);
    double lmax = 0.0;
    if (lane < d) lmax = fabs(L.Ms[sys_index<d>(lane, lane)].re);
    lmax = wave_max(lmax);
    if (lane < d) {
        const double l = L.Ms[sys_index<d>(lane, lane)].re;
        double f;
        if (fn == 0) f = log(l);
        else if (fn == 1) f = (fabs(l) > d * DBL_EPSILON * lmax) ? 1.0 / l : 0.0;   // scipy pinv cut-off
        else f = sqrt(l > 0.0 ? l : 0.0);
        L.lam[lane] = f;
    }
    FBX_WAVE_SYNC();
    const Blk o = reconstruct_blk<d>(L.Vs, L.lam, lane);
    blk_store<d, d>(dst, lane, o);
    FBX_WAVE_SYNC();
}

// tmp2 = A * B (row-major d x d), one output element per lane
template <int NQ>
__device__ __forceinline__ cplx matmul_elem(const cplx* A, const cplx* Bm, int lane) {
    constexpr int d = 1 << NQ;
    cplx o; o.re = 0.0; o.im = 0.0;
    if (lane < d * d) {
        const int r = lane / d, c = lane % d;
#pragma unroll
        for (int k = 0; k < d; ++k) {
            const cplx a = A[r * d + k], b = Bm[k * d + c];
            o.re += a.re * b.re - a.im * b.im;
            o.im += a.re * b.im + a.im * b.re;
        }
    }
    return o;
}

// ---------------------------------------------------------------------------------------------
// PLAIN: no entropy penalty, no hedging, a design of at most 64 settings (every state design of the reference) -- the variants'
// code (two Hermitian matrix functions through the Jacobi) is compiled out
template <int NQ, bool PLAIN, class Tables>
__device__ __forceinline__ void
mle_state_body(char* smem, const DesignDev& des, long long B, const double* __restrict__ expect, const double* __restrict__ counts,
               double epsilon, double entropy_penalty, double beta, double tol, int maxiter,
               double* __restrict__ rho_out, int* __restrict__ iters_out, int* __restrict__ hit_out) {
    constexpr int d = 1 << NQ, D = d * d;
    StateLds<NQ> L; L.carve(smem, StateLds<NQ>::staged(des.m) ? des.m : 0);
    const int lane = threadIdx.x;
    const long long item = blockIdx.x;
    const double* e = expect + item * des.m;
    const bool act = lane < D;
    const int row = act ? lane / d : 0, col = act ? lane % d : 0;
    double num_meas = 0.0;
    for (int g = lane; g < des.m; g += 64) num_meas += counts[item * des.m + g];
    num_meas = wave_sum(num_meas);
    cplx rho; rho.re = (act && row == col) ? 1.0 / d : 0.0; rho.im = 0.0;
    if (act) L.rho[lane] = rho;
    FBX_WAVE_SYNC();
    const LaneSetting mine = load_lane_setting<NQ>(des, e, lane);
    Tables tab; tab.init(act ? lane : 0, row, col);
    int iteration = 1, hit = 0;
    while (true) {
        if (iteration >= maxiter) { hit = 1; break; }            // tomography.py:244-246
        cplx T;                                                    // R(rho)
        if constexpr (PLAIN) T = r_operator_tab<NQ>(des, L, lane, mine, tab);
        else T = des.m <= 64 ? r_operator_tab<NQ>(des, L, lane, mine, tab) : r_operator_elem<NQ>(des, e, L, lane, &mine);
        if (act && row == col) T.re -= 1.0;                        // Tk = R - I
        if (!PLAIN && entropy_penalty > 0.0) {                     // tomography.py:252-254
            herm_function<NQ>(L.rho, L.aux, 0, L, lane, false);    // logm(rho)
            cplx lg; lg.re = 0.0; lg.im = 0.0;
            if (act) lg = L.aux[lane];
            const cplx rl = matmul_elem<NQ>(L.rho, L.aux, lane);   // rho @ logm(rho)
            double tr_re = (act && row == col) ? rl.re : 0.0, tr_im = (act && row == col) ? rl.im : 0.0;
            tr_re = wave_sum(tr_re); tr_im = wave_sum(tr_im);
            if (act && row == col) { lg.re -= tr_re; lg.im -= tr_im; }
            T.re -= entropy_penalty * lg.re; T.im -= entropy_penalty * lg.im;
        }
        if (!PLAIN && beta > 0.0) {                                // tomography.py:257-260
            T.re *= num_meas / 2; T.im *= num_meas / 2;
            herm_function<NQ>(L.rho, L.aux, 1, L, lane, false);    // pinv(rho)
            cplx pi; pi.re = 0.0; pi.im = 0.0;
            if (act) pi = L.aux[lane];
            if (act && row == col) pi.re -= d;
            T.re += beta * pi.re / 2; T.im += beta * pi.im / 2;
        }
        cplx Um; Um.re = epsilon * T.re + ((act && row == col) ? 1.0 : 0.0); Um.im = epsilon * T.im;
        FBX_WAVE_SYNC();
        if (act) L.U[lane] = Um;
        FBX_WAVE_SYNC();
        const cplx t1 = matmul_elem<NQ>(L.rho, L.U, lane);         // rho U
        if (act) L.tmp[lane] = t1;
        FBX_WAVE_SYNC();
        cplx nr = matmul_elem<NQ>(L.U, L.tmp, lane);               // U rho U
        double tr_re = (act && row == col) ? nr.re : 0.0, tr_im = (act && row == col) ? nr.im : 0.0;
        tr_re = wave_sum(tr_re); tr_im = wave_sum(tr_im);
        {   // complex division by the trace
            const double den = tr_re * tr_re + tr_im * tr_im;
            const double qr = (nr.re * tr_re + nr.im * tr_im) / den, qi = (nr.im * tr_re - nr.re * tr_im) / den;
            nr.re = qr; nr.im = qi;
        }
        double diff = act ? (nr.re - rho.re) * (nr.re - rho.re) + (nr.im - rho.im) * (nr.im - rho.im) : 0.0;
        diff = uniform(wave_sum(diff));
        rho = nr;
        FBX_WAVE_SYNC();
        if (act) L.rho[lane] = rho;
        FBX_WAVE_SYNC();
        if (sqrt(diff) < tol) break;
        ++iteration;
    }
    if (act) { rho_out[(item * D + lane) * 2] = rho.re; rho_out[(item * D + lane) * 2 + 1] = rho.im; }
    if (lane == 0) { if (iters_out) iters_out[item] = iteration; if (hit_out) hit_out[item] = hit; }
}

template <int NQ>
__global__ void __launch_bounds__(64)
mle_state_kernel(DesignDev des, long long B, const double* __restrict__ expect, const double* __restrict__ counts,
                 double epsilon, double entropy_penalty, double beta, double tol, int maxiter,
                 double* __restrict__ rho_out, int* __restrict__ iters_out, int* __restrict__ hit_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    mle_state_body<NQ, false, PauliTables<NQ>>(smem, des, B, expect, counts, epsilon, entropy_penalty, beta, tol, maxiter, rho_out, iters_out, hit_out);
}

// The plain 3-qubit reconstruction (what bench.py --workload mle_state3 times): the body without the variants and with the ordered
// tables (PauliTablesLean: d indices + d signs per pass instead of d + 2 d coefficients), held to 128 registers = FOUR wavefronts per
// SIMD -- with the tables of PauliTables it needs 236 registers (two wavefronts), and two wavefronts do not fill the vector pipe
// (VALU active 0.40 per wave).  Same box, bit-identical: 22.7 -> 17.4 ms per 2^18 reconstructions with the signs as bit masks
// (94 registers), 17.0 ms with the signs as +-1.0 constants of an FMA (110 registers) -- the latter is what runs.
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4)))
mle_state_plain3_kernel(DesignDev des, long long B, const double* __restrict__ expect, const double* __restrict__ counts,
                        double epsilon, double tol, int maxiter, double* __restrict__ rho_out, int* __restrict__ iters_out,
                        int* __restrict__ hit_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    mle_state_body<3, true, PauliTablesLean<3, true>>(smem, des, B, expect, counts, epsilon, 0.0, 0.0, tol, maxiter, rho_out, iters_out, hit_out);
}

// ---- plain diluted MLE (no entropy penalty, no hedging) for 1 and 2 qubits with SEVERAL items per
// wavefront: a d x d state needs d^2 = 4 / 16 lanes, so 16 / 4 reconstructions share a wave, each in
// its own group of D consecutive lanes with its own slice of LDS.  Needs m <= D settings (every
// state-tomography design of the reference has 4^n - 1); groups that have converged idle until the
// last one of the wave has.  Same arithmetic per item as mle_state_kernel.
template <int D>
__device__ __forceinline__ double group_sum(double v) {          // sum over the D lanes of a group, in every lane
    v += dpp_permute<0xB1>(v);                                   // quad_perm [1,0,3,2]
    v += dpp_permute<0x4E>(v);                                   // quad_perm [2,3,0,1]
    if constexpr (D == 16) { v += dpp_permute<0x141>(v); v += dpp_permute<0x140>(v); }
    return v;
}

template <int NQ>
__global__ void __launch_bounds__(64)
mle_state_packed_kernel(DesignDev des, long long B, const double* __restrict__ expect, double epsilon, double tol,
                        int maxiter, double* __restrict__ rho_out, int* __restrict__ iters_out, int* __restrict__ hit_out) {
    constexpr int d = 1 << NQ, D = d * d, G = 64 / D;
    static_assert(D == 4 || D == 16, "one or two qubits");
    __shared__ cplx s_rho[G * D], s_U[G * D], s_tmp[G * D];
    __shared__ double s_r[G * D], s_w[G * D];
    const int lane = threadIdx.x, sub = lane / D, t = lane % D;
    const int row = t / d, col = t % d;
    const long long item = (long long)blockIdx.x * G + sub;
    const bool valid = item < B;
    const int m = des.m;
    cplx* rho_l = s_rho + sub * D; cplx* U_l = s_U + sub * D; cplx* tmp_l = s_tmp + sub * D;
    double* r_l = s_r + sub * D; double* w_l = s_w + sub * D;
    // this lane's setting (t < m <= D)
    const bool has = valid && t < m;
    int sp = 0; double cf = 1.0, me = 0.0;
    if (has) { sp = des.sp[t] & 0xffff; cf = des.unit_coefs ? 1.0 : des.coef[t]; me = expect[item * m + des.order[t]]; }
    cplx rho; rho.re = (row == col) ? 1.0 / d : 0.0; rho.im = 0.0;
    rho_l[t] = rho;
    FBX_WAVE_SYNC();
    PauliTables<NQ> tab; tab.init(t, row, col);
    int iteration = 1, hit = 0;
    bool running = valid;
    while (__ballot(running)) {
        if (running && iteration >= maxiter) { hit = 1; running = false; }     // tomography.py:244-246
        if (!__ballot(running)) break;
        // ---- R(rho)  (tomography.py:273-338)
        r_l[t] = tab.expectation(rho_l);
        w_l[t] = 0.0;
        FBX_WAVE_SYNC();
        double s0 = 0.0;
        if (has) {
            const double pe = cf * r_l[sp];
            const double gp = ((1.0 + me) * 0.5) / ((1.0 + pe) * 0.5 + DBL_MIN);
            const double gm = ((1.0 - me) * 0.5) / ((1.0 - pe) * 0.5 + DBL_MIN);
            s0 = 0.5 * (gp + gm);
            atomicAdd(&w_l[sp], cf * 0.5 * (gp - gm));
        }
        s0 = group_sum<D>(s0);
        FBX_WAVE_SYNC();
        w_l[t] = w_l[t] / m;
        FBX_WAVE_SYNC();
        cplx T = tab.synthesis(w_l, s0 / m + 0.0, row == col);
        if (row == col) T.re -= 1.0;                                           // Tk = R - I
        cplx Um; Um.re = epsilon * T.re + ((row == col) ? 1.0 : 0.0); Um.im = epsilon * T.im;
        FBX_WAVE_SYNC();
        U_l[t] = Um;
        FBX_WAVE_SYNC();
        const cplx t1 = matmul_elem<NQ>(rho_l, U_l, t);                         // rho U
        tmp_l[t] = t1;
        FBX_WAVE_SYNC();
        cplx nr = matmul_elem<NQ>(U_l, tmp_l, t);                               // U rho U
        const double tr_re = group_sum<D>((row == col) ? nr.re : 0.0), tr_im = group_sum<D>((row == col) ? nr.im : 0.0);
        {
            const double den = tr_re * tr_re + tr_im * tr_im;
            const double qr = (nr.re * tr_re + nr.im * tr_im) / den, qi = (nr.im * tr_re - nr.re * tr_im) / den;
            nr.re = qr; nr.im = qi;
        }
        const double diff = group_sum<D>((nr.re - rho.re) * (nr.re - rho.re) + (nr.im - rho.im) * (nr.im - rho.im));
        FBX_WAVE_SYNC();
        if (running) { rho = nr; rho_l[t] = rho; }
        FBX_WAVE_SYNC();
        if (running) {
            if (sqrt(diff) < tol) running = false; else ++iteration;
        }
    }
    if (valid) {
        rho_out[(item * D + t) * 2] = rho.re; rho_out[(item * D + t) * 2 + 1] = rho.im;
        if (t == 0) { if (iters_out) iters_out[item] = iteration; if (hit_out) hit_out[item] = hit; }
    }
}

template <int NQ>
__global__ void __launch_bounds__(64)
r_operator_kernel(DesignDev des, long long B, const double* __restrict__ rho_in, const double* __restrict__ expect,
                  double* __restrict__ r_out) {
    constexpr int d = 1 << NQ, D = d * d;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    StateLds<NQ> L; L.carve(smem, StateLds<NQ>::staged(des.m) ? des.m : 0);
    const int lane = threadIdx.x;
    const long long item = blockIdx.x;
    if (lane < D) { L.rho[lane].re = rho_in[(item * D + lane) * 2]; L.rho[lane].im = rho_in[(item * D + lane) * 2 + 1]; }
    FBX_WAVE_SYNC();
    const cplx R = r_operator_elem<NQ>(des, expect + item * des.m, L, lane);
    if (lane < D) { r_out[(item * D + lane) * 2] = R.re; r_out[(item * D + lane) * 2 + 1] = R.im; }
}

// log-likelihood (log10), tomography.py:341-375
template <int NQ>
__global__ void __launch_bounds__(64)
loglik_kernel(DesignDev des, long long B, const double* __restrict__ rho_in, const double* __restrict__ expect,
              const double* __restrict__ counts, double* __restrict__ ll_out) {
    constexpr int d = 1 << NQ, D = d * d;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    StateLds<NQ> L; L.carve(smem, des.m);
    const int lane = threadIdx.x;
    const long long item = blockIdx.x;
    if (lane < D) { L.rho[lane].re = rho_in[(item * D + lane) * 2]; L.rho[lane].im = rho_in[(item * D + lane) * 2 + 1]; }
    FBX_WAVE_SYNC();
    pauli_expectations<NQ>(L.rho, L.r, lane);
    FBX_WAVE_SYNC();
    double ll = 0.0;
    for (int g = lane; g < des.m; g += 64) {
        const int k = des.order[g], p = des.sp[g] & 0xffff;
        const double cf = des.unit_coefs ? 1.0 : des.coef[g];
        const double n = counts[item * des.m + k], me = expect[item * des.m + k], pe = cf * L.r[p];
        const double pp = (1.0 + pe) / 2, pm = (1.0 - pe) / 2;
        if (pp > 0.0) ll += n * (1.0 + me) / 2 * log10(pp);
        if (pm > 0.0) ll += n * (1.0 - me) / 2 * log10(pm);
    }
    ll = wave_sum(ll);
    if (lane == 0) ll_out[item] = ll;
}

// linear inversion, tomography.py:130-165: per Pauli a least-squares coefficient, then synthesis
template <int NQ>
__global__ void __launch_bounds__(64)
linv_state_kernel(DesignDev des, long long B, const double* __restrict__ expect, double* __restrict__ rho_out) {
    constexpr int d = 1 << NQ, D = d * d;
    __shared__ double w[D];
    const int lane = threadIdx.x;
    const long long item = blockIdx.x;
    if (lane < D) {
        double num = 0.0, den = 0.0;
        for (int g = 0; g < des.m; ++g) {
            if ((int)(des.sp[g] & 0xffff) != lane) continue;
            const double cf = des.unit_coefs ? 1.0 : des.coef[g];
            num += cf * expect[item * des.m + des.order[g]];
            den += cf * cf;
        }
        w[lane] = den > 0.0 ? num / (den * d) : 0.0;     // pinv of orthogonal rows c_k vec(P)^H
    }
    FBX_WAVE_SYNC();
    if (lane < D) {
        const cplx v = pauli_synthesis<NQ>(w, 1.0 / d, lane / d, lane % d);   // + I/d, tomography.py:165
        rho_out[(item * D + lane) * 2] = v.re; rho_out[(item * D + lane) * 2 + 1] = v.im;
    }
}

// project_state_matrix_to_physical, project_state_matrix.py:6-52
template <int NQ>
__global__ void __launch_bounds__(64)
proj_state_kernel(long long B, const double* __restrict__ rho_in, double* __restrict__ out) {
    constexpr int d = 1 << NQ, D = d * d;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    StateLds<NQ> L; L.carve(smem, 1);
    const int lane = threadIdx.x;
    const long long item = blockIdx.x;
    const bool act = lane < D;
    cplx v; v.re = 0.0; v.im = 0.0;
    if (act) { v.re = rho_in[(item * D + lane) * 2]; v.im = rho_in[(item * D + lane) * 2 + 1]; }
    double tr_re = (act && lane / d == lane % d) ? v.re : 0.0, tr_im = (act && lane / d == lane % d) ? v.im : 0.0;
    tr_re = wave_sum(tr_re); tr_im = wave_sum(tr_im);
    const double den = tr_re * tr_re + tr_im * tr_im;
    cplx q; q.re = (v.re * tr_re + v.im * tr_im) / den; q.im = (v.im * tr_re - v.re * tr_im) / den;
    if (act) L.rho[lane] = q;
    FBX_WAVE_SYNC();
    // eigh (lower triangle, like scipy.linalg.eigh)
    constexpr int NB = d / 2;
    Blk h = blk_zero();
    if (lane < NB * NB) {
        const int I = lane / NB, J = lane % NB;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int r = 2 * I + (e >> 1), c = 2 * J + (e & 1);
            if (r > c) { h.re[e] = L.rho[r * d + c].re; h.im[e] = L.rho[r * d + c].im; }
            else if (r < c) { h.re[e] = L.rho[c * d + r].re; h.im[e] = -L.rho[c * d + r].im; }
            else { h.re[e] = L.rho[r * d + c].re; h.im[e] = 0.0; }
        }
    }
    sys_store<d>(L.Ms, lane, h);
    FBX_WAVE_SYNC();
    jacobi_eigh_lds<d>(L.Ms, L.Vs, L.rec, lane);
    __shared__ int physical;
    if (lane == 0) {
        double ev[d]; int idx[d];
        double mn = 1e300;
        for (int k = 0; k < d; ++k) { ev[k] = L.Ms[sys_index<d>(k, k)].re; idx[k] = k; mn = fmin(mn, ev[k]); }
        physical = mn >= 0.0;
        // descending order
        for (int a = 0; a < d; ++a) for (int b = a + 1; b < d; ++b)
            if (ev[b] > ev[a]) { double t = ev[a]; ev[a] = ev[b]; ev[b] = t; int u = idx[a]; idx[a] = idx[b]; idx[b] = u; }
        int i = d; double acc = 0.0;
        while (i > 0 && ev[i - 1] + acc / (double)i < 0.0) { acc += ev[i - 1]; --i; }
        for (int j = 0; j < d; ++j) L.lam[idx[j]] = (j < i) ? ev[j] + acc / (double)i : 0.0;
    }
    FBX_WAVE_SYNC();
    cplx o = q;
    if (!physical) {
        const Blk pb = reconstruct_blk<d>(L.Vs, L.lam, lane);
        blk_store<d, d>(L.tmp, lane, pb);
        FBX_WAVE_SYNC();
        if (act) o = L.tmp[lane];
    }
    if (act) { out[(item * D + lane) * 2] = o.re; out[(item * D + lane) * 2 + 1] = o.im; }
}

// purity / fidelity / trace distance / Hilbert-Schmidt inner product
template <int NQ>
__global__ void __launch_bounds__(64)
state_measures_kernel(long long B, const double* __restrict__ rho_in, const double* __restrict__ sig_in,
                      double* __restrict__ purity, double* __restrict__ fidelity, double* __restrict__ tdist,
                      double* __restrict__ hsip) {
    constexpr int d = 1 << NQ, D = d * d;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    StateLds<NQ> L; L.carve(smem, 1);
    const int lane = threadIdx.x;
    const long long item = blockIdx.x;
    const bool act = lane < D;
    cplx a, b; a.re = a.im = b.re = b.im = 0.0;
    if (act) {
        a.re = rho_in[(item * D + lane) * 2]; a.im = rho_in[(item * D + lane) * 2 + 1];
        b.re = sig_in[(item * D + lane) * 2]; b.im = sig_in[(item * D + lane) * 2 + 1];
        L.rho[lane] = a; L.U[lane] = b;
    }
    FBX_WAVE_SYNC();
    if (purity) {                                  // Re tr(rho rho)
        double p = 0.0;
        if (act) { const cplx t = L.rho[(lane % d) * d + lane / d]; p = a.re * t.re - a.im * t.im; }
        p = wave_sum(p);
        if (lane == 0) purity[item] = p;
    }
    if (hsip) {                                    // Re tr(A^H B)
        double p = act ? a.re * b.re + a.im * b.im : 0.0;
        p = wave_sum(p);
        if (lane == 0) hsip[item] = p;
    }
    if (tdist) {                                   // 0.5 * max_c sum_r |rho - sigma|[r][c]
        if (act) { const double dr = a.re - b.re, di = a.im - b.im; L.r[lane] = sqrt(dr * dr + di * di); }
        FBX_WAVE_SYNC();
        double cs = 0.0;
        if (lane < d) for (int r = 0; r < d; ++r) cs += L.r[r * d + lane];
        cs = wave_max(cs);
        if (lane == 0) tdist[item] = 0.5 * cs;
        FBX_WAVE_SYNC();
    }
    if (fidelity) {                                // (tr sqrtm_psd(sqrt_rho sigma sqrt_rho))^2
        herm_function<NQ>(L.rho, L.aux, 2, L, lane, true);            // sqrt_rho
        const cplx t1 = matmul_elem<NQ>(L.aux, L.U, lane);            // sqrt_rho sigma
        if (act) L.tmp[lane] = t1;
        FBX_WAVE_SYNC();
        const cplx t2 = matmul_elem<NQ>(L.tmp, L.aux, lane);          // ... sqrt_rho
        FBX_WAVE_SYNC();
        if (act) L.tmp[lane] = t2;
        FBX_WAVE_SYNC();
        herm_function<NQ>(L.tmp, L.aux, 2, L, lane, true);            // lam = sqrt(max(mu, 0))
        double s = (lane < d) ? L.lam[lane] : 0.0;
        s = wave_sum(s);
        if (lane == 0) fidelity[item] = s * s;
    }
}

// generic batched eigh (lower triangle read, ascending eigenvalues, eigenvectors as columns).
// One workgroup of NT = max(64, (N/2)^2) threads per matrix: one wavefront up to 16 x 16, four for
// 32 x 32, sixteen for 64 x 64 (128 KiB of LDS for the matrix and the eigenvectors).
template <int N, int NT>
__global__ void __launch_bounds__(NT)
eigh_kernel(long long B, const double* __restrict__ a, double* __restrict__ w_out, double* __restrict__ v_out) {
    constexpr int NB = N / 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx* Ms = (cplx*)smem;
    cplx* Vs = Ms + sys_elems<N>();
    double* lam = (double*)(Vs + sys_elems<N>());
    double* red = lam + N;
    int* pos = (int*)(red + 64);
    const int lane = threadIdx.x;
    const long long item = blockIdx.x;
    const double* src = a + item * (long long)N * N * 2;
    Blk h = blk_zero();
    if (lane < NB * NB) {
        const int I = lane / NB, J = lane % NB;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int r = 2 * I + (e >> 1), c = 2 * J + (e & 1);
            if (r > c) { h.re[e] = src[2 * (r * N + c)]; h.im[e] = src[2 * (r * N + c) + 1]; }
            else if (r < c) { h.re[e] = src[2 * (c * N + r)]; h.im[e] = -src[2 * (c * N + r) + 1]; }
            else { h.re[e] = src[2 * (r * N + c)]; h.im[e] = 0.0; }
        }
    }
    sys_store<N>(Ms, lane, h);
    __syncthreads();
    jacobi_eigh_block<N, NT>(Ms, Vs, lane, true, red);
    if (lane < N) lam[lane] = Ms[sys_index<N>(lane, lane)].re;
    __syncthreads();
    if (lane < N) {                     // rank of eigenvalue `lane` in ascending order (stable)
        int rank = 0;
        for (int j = 0; j < N; ++j) rank += (lam[j] < lam[lane]) || (lam[j] == lam[lane] && j < lane);
        pos[lane] = rank;
        w_out[item * N + rank] = lam[lane];
    }
    __syncthreads();
    if (v_out) {
        for (int idx = lane; idx < N * N; idx += NT) {
            const int r = idx / N, k = idx % N;
            const cplx v = Vs[sys_index<N>(r, k)];
            double* o = v_out + ((item * N + r) * N + pos[k]) * 2;
            o[0] = v.re; o[1] = v.im;
        }
    }
}

// ---- Hermitian eigendecomposition for 64 < N <= 1024 (4- and 5-qubit Choi matrices, padded odd sizes): the same
// systolic two-sided Jacobi with the matrix and the eigenvectors in HBM / L2 instead of LDS.  One 1024-thread
// workgroup per matrix; a round = (a) the N/2 rotations from the pivot blocks into an LDS table, (b) every 2 x 2
// block rotated and written to the seats the tournament permutation assigns it -- from the `cur` copies into the
// `nxt` copies, so no entry is overwritten before it is read -- and the copies swap.  Correctness first: this
// serves validators, choi2kraus and sqrtm of large operators, not a benchmark (one CU per matrix, ~20 us per round).
__device__ __forceinline__ int jacobi_seat_rt(int NB, int s) {           // jacobi_seat<N> with N at run time
    if (NB == 1) return s;
    const int k = s >> 1;
    if ((s & 1) == 0) {
        if (k == 0) return 0;
        if (k == NB - 1) return 2 * (NB - 1) + 1;
        return 2 * (k + 1);
    }
    if (k == 0) return 2;
    return 2 * (k - 1) + 1;
}

__global__ void __launch_bounds__(1024)
eigh_big_kernel(int N, long long B, const double* __restrict__ a, double* __restrict__ w_out, double* __restrict__ v_out,
                cplx* __restrict__ work) {
    constexpr int NT = 1024;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* rot = (double*)smem;                 // [N/2][4]: c, sr, si, -
    double* lam = rot + 2 * N;                   // [N]
    int* pos = (int*)(lam + N);                  // [N]
    double* red = (double*)(pos + N);            // [64]
    const int t = threadIdx.x, NB = N / 2;
    const long long item = blockIdx.x;
    const size_t NN = (size_t)N * N;
    cplx* M0 = work + (size_t)item * 4 * NN;
    cplx* M1 = M0 + NN; cplx* V0 = M1 + NN; cplx* V1 = V0 + NN;
    const double* src = a + item * (long long)NN * 2;
    for (size_t idx = t; idx < NN; idx += NT) {            // numpy eigh: the lower triangle defines the matrix
        const int r = (int)(idx / N), c = (int)(idx % N);
        cplx h, v;
        if (r > c) { h.re = src[2 * idx]; h.im = src[2 * idx + 1]; }
        else if (r < c) { h.re = src[2 * ((size_t)c * N + r)]; h.im = -src[2 * ((size_t)c * N + r) + 1]; }
        else { h.re = src[2 * idx]; h.im = 0.0; }
        v.re = r == c ? 1.0 : 0.0; v.im = 0.0;
        M0[idx] = h; V0[idx] = v;
    }
    __syncthreads();
    cplx *Mc = M0, *Mn = M1, *Vc = V0, *Vn = V1;
    for (int sweep = 0; sweep < FBX_JACOBI_MAX_SWEEPS; ++sweep) {
        double o2 = 0.0, n2 = 0.0;
        for (size_t idx = t; idx < NN; idx += NT) {
            const cplx v = Mc[idx];
            const double a2 = v.re * v.re + v.im * v.im;
            n2 += a2;
            if (idx / N != idx % N) o2 += a2;
        }
        block_sum2<NT>(o2, n2, red);
        if (!(o2 > FBX_JACOBI_TOL2 * n2)) break;
        for (int r = 0; r < N - 1; ++r) {
            for (int p = t; p < NB; p += NT) {
                const cplx b = Mc[(size_t)(2 * p) * N + 2 * p + 1];
                const JRot q = jacobi_rotation(Mc[(size_t)(2 * p) * N + 2 * p].re, Mc[(size_t)(2 * p + 1) * N + 2 * p + 1].re, b.re, b.im);
                rot[4 * p] = q.c; rot[4 * p + 1] = q.sr; rot[4 * p + 2] = q.si;
            }
            __syncthreads();
            for (int blk = t; blk < NB * NB; blk += NT) {
                const int I = blk / NB, J = blk % NB;
                const size_t r0 = (size_t)(2 * I) * N + 2 * J, r1 = r0 + N;
                cplx m00 = Mc[r0], m01 = Mc[r0 + 1], m10 = Mc[r1], m11 = Mc[r1 + 1];
                cplx v0p = Vc[r0], v0q = Vc[r0 + 1], v1p = Vc[r1], v1q = Vc[r1 + 1];
                const double cJ = rot[4 * J], sJr = rot[4 * J + 1], sJi = rot[4 * J + 2];
                jacobi_apply_m(rot[4 * I], rot[4 * I + 1], rot[4 * I + 2], cJ, sJr, sJi, m00, m01, m10, m11);
                jacobi_apply_v(cJ, sJr, sJi, v0p, v0q, v1p, v1q);
                if (I == J) { m01.re = m01.im = 0.0; m10.re = m10.im = 0.0; m00.im = 0.0; m11.im = 0.0; }
                const size_t ra = (size_t)jacobi_seat_rt(NB, 2 * I) * N, rb = (size_t)jacobi_seat_rt(NB, 2 * I + 1) * N;
                const int ca = jacobi_seat_rt(NB, 2 * J), cb = jacobi_seat_rt(NB, 2 * J + 1);
                Mn[ra + ca] = m00; Mn[ra + cb] = m01; Mn[rb + ca] = m10; Mn[rb + cb] = m11;
                const size_t va = (size_t)(2 * I) * N, vb = va + N;                  // eigenvector ROWS stay, columns move
                Vn[va + ca] = v0p; Vn[va + cb] = v0q; Vn[vb + ca] = v1p; Vn[vb + cb] = v1q;
            }
            __syncthreads();
            cplx* q = Mc; Mc = Mn; Mn = q; q = Vc; Vc = Vn; Vn = q;
        }
    }
    // (after whole sweeps the seats are the indices again)
    for (int k = t; k < N; k += NT) lam[k] = Mc[(size_t)k * N + k].re;
    __syncthreads();
    for (int k = t; k < N; k += NT) {               // rank of eigenvalue k in ascending order (stable)
        int rank = 0;
        for (int j = 0; j < N; ++j) rank += (lam[j] < lam[k]) || (lam[j] == lam[k] && j < k);
        pos[k] = rank;
        w_out[item * N + rank] = lam[k];
    }
    __syncthreads();
    if (v_out) {
        for (size_t idx = t; idx < NN; idx += NT) {
            const int r = (int)(idx / N), k = (int)(idx % N);
            const cplx v = Vc[idx];
            double* o = v_out + ((item * N + r) * N + pos[k]) * 2;
            o[0] = v.re; o[1] = v.im;
        }
    }
}

// ---- the same decomposition with ONE matrix spread over the chip: a cooperative launch (every workgroup resident),
// each workgroup takes a share of the 2 x 2 blocks of a round, computes the round's N/2 rotations for itself (they
// are cheap and every block needs two of them), and a grid-wide barrier separates the rounds.  One CU moves the
// 4 N^2 x 16 bytes of a round at ~85 GB/s; the chip moves them at L2 / HBM speed, so the barrier (a few microseconds)
// becomes the cost of a round.  hipLaunchCooperativeKernel refuses a grid that cannot be co-resident, in which case
// (or with fbx_set_option("eigh_cooperative", 0)) the single-workgroup kernel above takes over.
__global__ void __launch_bounds__(256)
eigh_coop_kernel(int N, const double* __restrict__ a, double* __restrict__ w_out, double* __restrict__ v_out,
                 cplx* __restrict__ work, double* __restrict__ partial) {
    namespace cg = cooperative_groups;
    cg::grid_group grid = cg::this_grid();
    constexpr int NT = 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* rot = (double*)smem;                 // [N/2][4]
    double* red = rot + 2 * N;                   // [16]
    const int t = threadIdx.x, NB = N / 2, G = gridDim.x, g = blockIdx.x;
    const size_t NN = (size_t)N * N;
    cplx* M0 = work; cplx* M1 = M0 + NN; cplx* V0 = M1 + NN; cplx* V1 = V0 + NN;
    for (size_t idx = (size_t)g * NT + t; idx < NN; idx += (size_t)G * NT) {
        const int r = (int)(idx / N), c = (int)(idx % N);
        cplx h, v;
        if (r > c) { h.re = a[2 * idx]; h.im = a[2 * idx + 1]; }
        else if (r < c) { h.re = a[2 * ((size_t)c * N + r)]; h.im = -a[2 * ((size_t)c * N + r) + 1]; }
        else { h.re = a[2 * idx]; h.im = 0.0; }
        v.re = r == c ? 1.0 : 0.0; v.im = 0.0;
        M0[idx] = h; V0[idx] = v;
    }
    grid.sync();
    cplx *Mc = M0, *Mn = M1, *Vc = V0, *Vn = V1;
    for (int sweep = 0; sweep < FBX_JACOBI_MAX_SWEEPS; ++sweep) {
        double o2 = 0.0, n2 = 0.0;
        for (size_t idx = (size_t)g * NT + t; idx < NN; idx += (size_t)G * NT) {
            const cplx v = Mc[idx];
            const double a2 = v.re * v.re + v.im * v.im;
            n2 += a2;
            if (idx / N != idx % N) o2 += a2;
        }
        block_sum2<NT>(o2, n2, red);
        if (t == 0) { partial[2 * g] = o2; partial[2 * g + 1] = n2; }
        grid.sync();
        o2 = 0.0; n2 = 0.0;
        for (int k = 0; k < G; ++k) { o2 += partial[2 * k]; n2 += partial[2 * k + 1]; }      // same order in every workgroup
        grid.sync();                                  // `partial` is rewritten at the next sweep
        if (!(o2 > FBX_JACOBI_TOL2 * n2)) break;
        for (int r = 0; r < N - 1; ++r) {
            for (int p = t; p < NB; p += NT) {
                const cplx b = Mc[(size_t)(2 * p) * N + 2 * p + 1];
                const JRot q = jacobi_rotation(Mc[(size_t)(2 * p) * N + 2 * p].re, Mc[(size_t)(2 * p + 1) * N + 2 * p + 1].re, b.re, b.im);
                rot[4 * p] = q.c; rot[4 * p + 1] = q.sr; rot[4 * p + 2] = q.si;
            }
            __syncthreads();
            for (int blk = g * NT + t; blk < NB * NB; blk += G * NT) {
                const int I = blk / NB, J = blk % NB;
                const size_t r0 = (size_t)(2 * I) * N + 2 * J, r1 = r0 + N;
                cplx m00 = Mc[r0], m01 = Mc[r0 + 1], m10 = Mc[r1], m11 = Mc[r1 + 1];
                cplx v0p = Vc[r0], v0q = Vc[r0 + 1], v1p = Vc[r1], v1q = Vc[r1 + 1];
                const double cJ = rot[4 * J], sJr = rot[4 * J + 1], sJi = rot[4 * J + 2];
                jacobi_apply_m(rot[4 * I], rot[4 * I + 1], rot[4 * I + 2], cJ, sJr, sJi, m00, m01, m10, m11);
                jacobi_apply_v(cJ, sJr, sJi, v0p, v0q, v1p, v1q);
                if (I == J) { m01.re = m01.im = 0.0; m10.re = m10.im = 0.0; m00.im = 0.0; m11.im = 0.0; }
                const size_t ra = (size_t)jacobi_seat_rt(NB, 2 * I) * N, rb = (size_t)jacobi_seat_rt(NB, 2 * I + 1) * N;
                const int ca = jacobi_seat_rt(NB, 2 * J), cb = jacobi_seat_rt(NB, 2 * J + 1);
                Mn[ra + ca] = m00; Mn[ra + cb] = m01; Mn[rb + ca] = m10; Mn[rb + cb] = m11;
                const size_t va = (size_t)(2 * I) * N, vb = va + N;
                Vn[va + ca] = v0p; Vn[va + cb] = v0q; Vn[vb + ca] = v1p; Vn[vb + cb] = v1q;
            }
            grid.sync();
            cplx* q = Mc; Mc = Mn; Mn = q; q = Vc; Vc = Vn; Vn = q;
        }
    }
    // eigenvalues ascending, eigenvectors as columns in that order (ranks recomputed by every thread that needs one)
    for (size_t idx = (size_t)g * NT + t; idx < (size_t)N; idx += (size_t)G * NT) {
        const int k = (int)idx;
        const double lk = Mc[(size_t)k * N + k].re;
        int rank = 0;
        for (int j = 0; j < N; ++j) { const double lj = Mc[(size_t)j * N + j].re; rank += (lj < lk) || (lj == lk && j < k); }
        w_out[rank] = lk;
        partial[2 * G + k] = (double)rank;            // column k goes to column `rank`
    }
    grid.sync();
    if (v_out) {
        for (size_t idx = (size_t)g * NT + t; idx < NN; idx += (size_t)G * NT) {
            const int r = (int)(idx / N), k = (int)(idx % N);
            const cplx v = Vc[idx];
            double* o = v_out + ((size_t)r * N + (int)partial[2 * G + k]) * 2;
            o[0] = v.re; o[1] = v.im;
        }
    }
}

static int launch_eigh_coop(int N, int64_t B, const double* da, double* dw, double* dv, bool* done) {
    *done = false;
    if (!option_eigh_cooperative()) return FBX_OK;
    int dev = current_device(), coop = 0, cus = 0;
    if (hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, dev) != hipSuccess || !coop) { (void)hipGetLastError(); return FBX_OK; }
    FBX_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const size_t lds = sizeof(double) * (2 * (size_t)N + 16);
    int per_cu = 0;
    FBX_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, eigh_coop_kernel, 256, lds));
    if (per_cu < 1) return FBX_OK;
    const long long blocks_needed = ((long long)(N / 2) * (N / 2) + 255) / 256;
    long long G = std::min<long long>(blocks_needed, (long long)cus * std::min(per_cu, 2));
    if (G < 2) return FBX_OK;
    const size_t NN = (size_t)N * N;
    void* w = nullptr;
    { const int rc = workspace(WS_CONVERT, 4 * NN * sizeof(cplx) + sizeof(double) * (2 * (size_t)G + N), &w); if (rc) return rc; }
    cplx* work = (cplx*)w;
    double* partial = (double*)(work + 4 * NN);
    for (int64_t b = 0; b < B; ++b) {
        const double* a = da + b * NN * 2;
        double* wo = dw + b * N;
        double* vo = dv ? dv + b * NN * 2 : nullptr;
        int n_arg = N;
        void* args[] = {&n_arg, (void*)&a, (void*)&wo, (void*)&vo, (void*)&work, (void*)&partial};
        const hipError_t e = hipLaunchCooperativeKernel((const void*)eigh_coop_kernel, dim3((unsigned)G), dim3(256), args, (unsigned)lds, stream());
        if (e != hipSuccess) { (void)hipGetLastError(); if (b == 0) return FBX_OK; return hip_fail(e, "hipLaunchCooperativeKernel", __FILE__, __LINE__); }
    }
    *done = true;
    return FBX_OK;
}

static int launch_eigh_big(int N, int64_t B, const double* da, double* dw, double* dv) {
    const size_t lds = sizeof(double) * (2 * (size_t)N + N + 64) + sizeof(int) * N;
    const size_t per_item = 4 * (size_t)N * N * sizeof(cplx);
    const int64_t chunk = (int64_t)std::max<size_t>(1, std::min<size_t>((size_t)B, ((size_t)1 << 30) / per_item));
    void* w = nullptr;
    { const int rc = workspace(WS_CONVERT, per_item * (size_t)chunk, &w); if (rc) return rc; }
    for (int64_t b0 = 0; b0 < B; b0 += chunk) {
        const int64_t nb = B - b0 < chunk ? B - b0 : chunk;
        hipLaunchKernelGGL(eigh_big_kernel, dim3((unsigned)nb), dim3(1024), lds, stream(), N, (long long)nb,
                           da + b0 * (size_t)N * N * 2, dw + b0 * N, dv ? dv + b0 * (size_t)N * N * 2 : nullptr, (cplx*)w);
    }
    FBX_HIP(hipGetLastError());
    return FBX_OK;
}

// ---- out = op(A) diag(s) op(B) for stacks of N x N complex matrices, N up to 1024: the products around the large
// eigensolver (V f(lambda) V^H of sqrtm_psd, calculational.py:77-91; sqrt(rho) sigma sqrt(rho) of fidelity,
// distance_measures.py:64-84).  16 x 16 output tiles staged through LDS; a utility, not a tuned GEMM.
__global__ void __launch_bounds__(256)
matmul_kernel(int N, long long B, const double* __restrict__ a, int conj_t_a, const double* __restrict__ scale,
              const double* __restrict__ b, int conj_t_b, double* __restrict__ out) {
    __shared__ cplx As[16][17], Bs[16][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int tiles = (N + 15) / 16;
    const long long item = blockIdx.x / (tiles * tiles);
    const int tile = (int)(blockIdx.x % (tiles * tiles)), row0 = (tile / tiles) * 16, col0 = (tile % tiles) * 16;
    const double* pa = a + item * (long long)N * N * 2;
    const double* pb = b + item * (long long)N * N * 2;
    double re = 0.0, im = 0.0;
    for (int k0 = 0; k0 < N; k0 += 16) {
        {   // As[ty][tx] = op(A)[row0 + ty][k0 + tx] * s[k0 + tx];  Bs[ty][tx] = op(B)[k0 + ty][col0 + tx]
            const int r = row0 + ty, k = k0 + tx;
            cplx v; v.re = 0.0; v.im = 0.0;
            if (r < N && k < N) {
                const long long idx = conj_t_a ? (long long)k * N + r : (long long)r * N + k;
                v.re = pa[2 * idx]; v.im = conj_t_a ? -pa[2 * idx + 1] : pa[2 * idx + 1];
                if (scale) { const double sc = scale[item * N + k]; v.re *= sc; v.im *= sc; }
            }
            As[ty][tx] = v;
            const int kk = k0 + ty, c = col0 + tx;
            cplx w; w.re = 0.0; w.im = 0.0;
            if (kk < N && c < N) {
                const long long idx = conj_t_b ? (long long)c * N + kk : (long long)kk * N + c;
                w.re = pb[2 * idx]; w.im = conj_t_b ? -pb[2 * idx + 1] : pb[2 * idx + 1];
            }
            Bs[ty][tx] = w;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const cplx x = As[ty][k], y = Bs[k][tx];
            re += x.re * y.re - x.im * y.im;
            im += x.re * y.im + x.im * y.re;
        }
        __syncthreads();
    }
    const int r = row0 + ty, c = col0 + tx;
    if (r < N && c < N) {
        double* o = out + (item * (long long)N * N + (long long)r * N + c) * 2;
        o[0] = re; o[1] = im;
    }
}

template <int N>
static int launch_eigh(int64_t B, const double* da, double* dw, double* dv) {
    constexpr int NT = (N / 2) * (N / 2) > 64 ? (N / 2) * (N / 2) : 64;
    const size_t lds = 2 * sizeof(cplx) * sys_elems<N>() + sizeof(double) * (N + 64) + sizeof(int) * N;
    auto kern = eigh_kernel<N, NT>;
    FBX_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)B), dim3(NT), lds, stream(), (long long)B, da, dw, dv);
    FBX_HIP(hipGetLastError());
    return FBX_OK;
}

}  // namespace fbx

using namespace fbx;


// =====================================================================================================
// 4 and 5 qubits (16 x 16 and 32 x 32 density matrices, 255 / 1023 settings): ONE WORKGROUP OF d*d THREADS per state
// (256 / 1024), thread t owns matrix entry (t / d, t % d) as a lane does above, the same device routines
// (pauli_expectations, pauli_synthesis, the block Jacobi) with workgroup barriers and workgroup sums.  A 16 x 16
// density matrix has the size of a 2-qubit Choi matrix: its Hermitian functions (logm for the entropy penalty,
// pinv for hedging) run on the solver of the 2- / 3-qubit process kernels (jacobi_eigh_simple<d, NT>).
// Settings are held one per thread (m <= d*d for every design of the reference); weights of the R operator
// are summed with LDS atomics (several settings on the same Pauli operator -- not in the reference's designs --
// would add in an order that differs between runs at rounding level).
template <int NQ>
struct StateBigLds {
    static constexpr int d = 1 << NQ, D = d * d, NT = D;
    cplx *rho, *U, *tmp, *aux, *Ms, *Vs;
    double *w, *r, *lam, *red;
    static constexpr size_t bytes() { return sizeof(cplx) * (4 * (size_t)D + 2 * (size_t)sys_elems<d>()) + sizeof(double) * (2 * (size_t)D + d + 2 * (NT / 64) + 8); }
    __device__ void carve(char* p) {
        rho = (cplx*)p; p += sizeof(cplx) * D;  U = (cplx*)p; p += sizeof(cplx) * D;
        tmp = (cplx*)p; p += sizeof(cplx) * D;  aux = (cplx*)p; p += sizeof(cplx) * D;
        Ms = (cplx*)p; p += sizeof(cplx) * sys_elems<d>();  Vs = (cplx*)p; p += sizeof(cplx) * sys_elems<d>();
        w = (double*)p; p += sizeof(double) * D; r = (double*)p; p += sizeof(double) * D;
        lam = (double*)p; p += sizeof(double) * d; red = (double*)p;
    }
};

template <int NQ>
__device__ __forceinline__ double big_sum(double v, StateBigLds<NQ>& L) { return block_sum<StateBigLds<NQ>::NT>(v, L.red); }

// R operator (tomography.py:273-338) of the state in L.rho; element of this thread.  Ends behind a barrier.
template <int NQ>
__device__ cplx r_operator_big(const DesignDev& des, const double* __restrict__ e, StateBigLds<NQ>& L, int t) {
    constexpr int d = 1 << NQ, NT = d * d;
    const int m = des.m;
    pauli_expectations<NQ>(L.rho, L.r, t);
    L.w[t] = 0.0;
    __syncthreads();
    double s0 = 0.0;
    for (int g = t; g < m; g += NT) {
        const int p = des.sp[g] & 0xffff;
        const double cf = des.unit_coefs ? 1.0 : des.coef[g];
        const double me = e[des.order[g]], pe = cf * L.r[p];
        const double gp = ((1.0 + me) * 0.5) / ((1.0 + pe) * 0.5 + DBL_MIN);
        const double gm = ((1.0 - me) * 0.5) / ((1.0 - pe) * 0.5 + DBL_MIN);
        s0 += 0.5 * (gp + gm);
        atomicAdd(&L.w[p], cf * 0.5 * (gp - gm));
    }
    s0 = big_sum<NQ>(s0, L);                            // (two barriers: the atomics above are complete behind them)
    L.w[t] = L.w[t] / m;
    __syncthreads();
    const cplx out = pauli_synthesis<NQ>(L.w, s0 / m + 0.0, t / d, t % d);
    __syncthreads();
    return out;
}

// dst = V f(lambda) V^H of the Hermitian part of `src` (row-major); fn as herm_function
template <int NQ>
__device__ void herm_function_big(const cplx* src, cplx* dst, int fn, StateBigLds<NQ>& L, int t) {
    constexpr int d = 1 << NQ, NB = d / 2, NT = d * d;
    Blk h = blk_zero();
    if (t < NB * NB) {
        const int I = t / NB, J = t % NB;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int r = 2 * I + (e >> 1), c = 2 * J + (e & 1);
            const cplx a = src[r * d + c], b = src[c * d + r];
            h.re[e] = 0.5 * (a.re + b.re); h.im[e] = 0.5 * (a.im - b.im);
        }
    }
    __syncthreads();
    sys_store<d>(L.Ms, t, h);
    __syncthreads();
    jacobi_eigh_simple<d, NT>(L.Ms, L.Vs, t, true, L.red);
    __syncthreads();
    double lmax = 0.0;
    if (t < d) lmax = fabs(L.Ms[sys_index<d>(t, t)].re);
    lmax = wave_max(lmax);                                   // d <= 32: the diagonal sits in the first wavefront
    if (t < d) {
        const double l = L.Ms[sys_index<d>(t, t)].re;
        double f;
        if (fn == 0) f = log(l);
        else if (fn == 1) f = (fabs(l) > d * DBL_EPSILON * lmax) ? 1.0 / l : 0.0;
        else f = sqrt(l > 0.0 ? l : 0.0);
        L.lam[t] = f;
    }
    __syncthreads();
    const Blk o = reconstruct_blk<d>(L.Vs, L.lam, t);
    blk_store<d, d>(dst, t, o);
    __syncthreads();
}

template <int NQ>
__device__ __forceinline__ cplx matmul_big(const cplx* A, const cplx* Bm, int t) {
    constexpr int d = 1 << NQ;
    const int r = t / d, c = t % d;
    cplx o; o.re = 0.0; o.im = 0.0;
#pragma unroll 8
    for (int k = 0; k < d; ++k) {
        const cplx a = A[r * d + k], b = Bm[k * d + c];
        o.re += a.re * b.re - a.im * b.im;
        o.im += a.re * b.im + a.im * b.re;
    }
    return o;
}

// iterative_mle_state_estimate, tomography.py:168-270: statement for statement the loop of mle_state_kernel
template <int NQ>
__global__ void __launch_bounds__(1 << (2 * NQ))
mle_state_big_kernel(DesignDev des, long long B, const double* __restrict__ expect, const double* __restrict__ counts,
                     double epsilon, double entropy_penalty, double beta, double tol, int maxiter,
                     double* __restrict__ rho_out, int* __restrict__ iters_out, int* __restrict__ hit_out) {
    constexpr int d = 1 << NQ, D = d * d, NT = D;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    StateBigLds<NQ> L; L.carve(smem);
    const int t = threadIdx.x;
    const long long item = blockIdx.x;
    const double* e = expect + item * des.m;
    const int row = t / d, col = t % d;
    double num_meas = 0.0;
    for (int g = t; g < des.m; g += NT) num_meas += counts[item * des.m + g];
    num_meas = big_sum<NQ>(num_meas, L);
    cplx rho; rho.re = (row == col) ? 1.0 / d : 0.0; rho.im = 0.0;
    L.rho[t] = rho;
    __syncthreads();
    int iteration = 1, hit = 0;
    while (true) {
        if (iteration >= maxiter) { hit = 1; break; }
        cplx T = r_operator_big<NQ>(des, e, L, t);
        if (row == col) T.re -= 1.0;
        if (entropy_penalty > 0.0) {
            herm_function_big<NQ>(L.rho, L.aux, 0, L, t);          // logm(rho)
            cplx lg = L.aux[t];
            const cplx rl = matmul_big<NQ>(L.rho, L.aux, t);
            double tr_re = (row == col) ? rl.re : 0.0, tr_im = (row == col) ? rl.im : 0.0;
            tr_re = big_sum<NQ>(tr_re, L); tr_im = big_sum<NQ>(tr_im, L);
            if (row == col) { lg.re -= tr_re; lg.im -= tr_im; }
            T.re -= entropy_penalty * lg.re; T.im -= entropy_penalty * lg.im;
        }
        if (beta > 0.0) {
            T.re *= num_meas / 2; T.im *= num_meas / 2;
            herm_function_big<NQ>(L.rho, L.aux, 1, L, t);          // pinv(rho)
            cplx pi = L.aux[t];
            if (row == col) pi.re -= d;
            T.re += beta * pi.re / 2; T.im += beta * pi.im / 2;
        }
        cplx Um; Um.re = epsilon * T.re + ((row == col) ? 1.0 : 0.0); Um.im = epsilon * T.im;
        __syncthreads();
        L.U[t] = Um;
        __syncthreads();
        const cplx t1 = matmul_big<NQ>(L.rho, L.U, t);
        L.tmp[t] = t1;
        __syncthreads();
        cplx nr = matmul_big<NQ>(L.U, L.tmp, t);
        double tr_re = (row == col) ? nr.re : 0.0, tr_im = (row == col) ? nr.im : 0.0;
        tr_re = big_sum<NQ>(tr_re, L); tr_im = big_sum<NQ>(tr_im, L);
        {
            const double den = tr_re * tr_re + tr_im * tr_im;
            const double qr = (nr.re * tr_re + nr.im * tr_im) / den, qi = (nr.im * tr_re - nr.re * tr_im) / den;
            nr.re = qr; nr.im = qi;
        }
        double diff = (nr.re - rho.re) * (nr.re - rho.re) + (nr.im - rho.im) * (nr.im - rho.im);
        diff = big_sum<NQ>(diff, L);
        rho = nr;
        __syncthreads();
        L.rho[t] = rho;
        __syncthreads();
        if (sqrt(diff) < tol) break;
        ++iteration;
    }
    rho_out[(item * D + t) * 2] = rho.re; rho_out[(item * D + t) * 2 + 1] = rho.im;
    if (t == 0) { if (iters_out) iters_out[item] = iteration; if (hit_out) hit_out[item] = hit; }
}

// op 0: R operator of a given state, 1: log-likelihood (log10) of a given state, 2: linear inversion
template <int NQ>
__global__ void __launch_bounds__(1 << (2 * NQ))
state_big_kernel(int op, DesignDev des, long long B, const double* __restrict__ rho_in, const double* __restrict__ expect,
                 const double* __restrict__ counts, double* __restrict__ out) {
    constexpr int d = 1 << NQ, D = d * d, NT = D;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    StateBigLds<NQ> L; L.carve(smem);
    const int t = threadIdx.x;
    const long long item = blockIdx.x;
    const int m = des.m;
    if (op == 2) {                                         // tomography.py:130-165, as linv_state_kernel
        double* num = L.w; double* den = L.r;
        num[t] = 0.0; den[t] = 0.0;
        __syncthreads();
        for (int g = t; g < m; g += NT) {
            const int p = des.sp[g] & 0xffff;
            const double cf = des.unit_coefs ? 1.0 : des.coef[g];
            atomicAdd(&num[p], cf * expect[item * m + des.order[g]]);
            atomicAdd(&den[p], cf * cf);
        }
        __syncthreads();
        const double wv = den[t] > 0.0 ? num[t] / (den[t] * d) : 0.0;
        __syncthreads();
        L.w[t] = wv;
        __syncthreads();
        const cplx v = pauli_synthesis<NQ>(L.w, 1.0 / d, t / d, t % d);
        out[(item * D + t) * 2] = v.re; out[(item * D + t) * 2 + 1] = v.im;
        return;
    }
    L.rho[t].re = rho_in[(item * D + t) * 2]; L.rho[t].im = rho_in[(item * D + t) * 2 + 1];
    __syncthreads();
    if (op == 0) {
        const cplx R = r_operator_big<NQ>(des, expect + item * m, L, t);
        out[(item * D + t) * 2] = R.re; out[(item * D + t) * 2 + 1] = R.im;
        return;
    }
    pauli_expectations<NQ>(L.rho, L.r, t);
    __syncthreads();
    double ll = 0.0;
    for (int g = t; g < m; g += NT) {
        const int k = des.order[g], p = des.sp[g] & 0xffff;
        const double cf = des.unit_coefs ? 1.0 : des.coef[g];
        const double n = counts[item * m + k], me = expect[item * m + k], pe = cf * L.r[p];
        const double pp = (1.0 + pe) / 2, pm = (1.0 - pe) / 2;
        if (pp > 0.0) ll += n * (1.0 + me) / 2 * log10(pp);
        if (pm > 0.0) ll += n * (1.0 - me) / 2 * log10(pm);
    }
    ll = big_sum<NQ>(ll, L);
    if (t == 0) out[item] = ll;
}

template <int NQ>
static int launch_state_big(int op, const fbx_design* des, int64_t B, const double* rho, const double* e, const double* c, double* out) {
    const size_t lds = StateBigLds<NQ>::bytes();
    FBX_HIP(hipFuncSetAttribute((const void*)state_big_kernel<NQ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(state_big_kernel<NQ>, dim3((unsigned)B), dim3(1 << (2 * NQ)), lds, stream(), op, des->dev, (long long)B, rho, e, c, out);
    FBX_HIP(hipGetLastError());
    return FBX_OK;
}
static int state_big(int op, const fbx_design* des, int64_t B, const double* rho, const double* e, const double* c, double* out) {
    return des->dev.n == 4 ? launch_state_big<4>(op, des, B, rho, e, c, out) : launch_state_big<5>(op, des, B, rho, e, c, out);
}
template <int NQ>
static int launch_mle_big(const fbx_design* des, int64_t B, const double* e, const double* c, double epsilon, double entropy_penalty,
                          double beta, double tol, int maxiter, double* rho, int32_t* it, int32_t* hit) {
    const size_t lds = StateBigLds<NQ>::bytes();
    FBX_HIP(hipFuncSetAttribute((const void*)mle_state_big_kernel<NQ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(mle_state_big_kernel<NQ>, dim3((unsigned)B), dim3(1 << (2 * NQ)), lds, stream(), des->dev, (long long)B, e, c,
                       epsilon, entropy_penalty, beta, tol, maxiter, rho, it, hit);
    FBX_HIP(hipGetLastError());
    return FBX_OK;
}

namespace {
struct HostIO {
    std::vector<DevBuf*> bufs;
    ~HostIO() { for (auto* b : bufs) delete b; }
    template <class T> int in(const T* host, size_t count, T** dev) {
        auto* b = new DevBuf(); bufs.push_back(b);
        int rc = b->alloc(sizeof(T) * count);
        if (rc) return rc;
        if (host && count) {
            hipError_t e = hipMemcpyAsync(b->p, host, sizeof(T) * count, hipMemcpyHostToDevice, stream());
            if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(H2D)", __FILE__, __LINE__);
        }
        *dev = b->as<T>();
        return FBX_OK;
    }
    template <class T> int out(size_t count, T** dev) {
        auto* b = new DevBuf(); bufs.push_back(b);
        int rc = b->alloc(sizeof(T) * count);
        if (rc) return rc;
        *dev = b->as<T>();
        return FBX_OK;
    }
    template <class T> int back(T* host, const T* dev, size_t count) {
        if (!host || !count) return FBX_OK;
        hipError_t e = hipMemcpyAsync(host, dev, sizeof(T) * count, hipMemcpyDeviceToHost, stream());
        if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(D2H)", __FILE__, __LINE__);
        return FBX_OK;
    }
    int sync() { FBX_HIP(hipStreamSynchronize(stream())); return FBX_OK; }
};
#define FBX_TRY(x) do { int _rc = (x); if (_rc) return _rc; } while (0)

#define FBX_DISPATCH_NQ(n, KERNEL, lds, B, ...)                                                    \
    do {                                                                                           \
        if ((n) == 1) hipLaunchKernelGGL(KERNEL<1>, dim3((unsigned)(B)), dim3(64), (lds), stream(), __VA_ARGS__); \
        else if ((n) == 2) hipLaunchKernelGGL(KERNEL<2>, dim3((unsigned)(B)), dim3(64), (lds), stream(), __VA_ARGS__); \
        else hipLaunchKernelGGL(KERNEL<3>, dim3((unsigned)(B)), dim3(64), (lds), stream(), __VA_ARGS__); \
    } while (0)

size_t state_lds(int n, int m) {
    return n == 1 ? StateLds<1>::launch_bytes(m) : n == 2 ? StateLds<2>::launch_bytes(m) : StateLds<3>::launch_bytes(m);
}
int check_state_design(const fbx_design* des, const char* who) {
    { const int rc = check_design(des, who); if (rc) return rc; }
    if (des->dev.kind != FBX_KIND_STATE) { set_error(std::string(who) + ": needs a state design"); return FBX_ERR_BAD_ARG; }
    return FBX_OK;
}
}  // namespace


// ---- Pauli-Liouville vector of a state: c2p vec(rho) = tr[P_k rho] / d, k in the order of
// itertools.product('IXYZ', repeat=n) (first letter = most significant qubit), the input of
// plotting/state_process.py:10-87 (computational2pauli_basis_matrix, superoperator_transformations.py:413-424).
// One thread per coefficient: P_k has one non-zero per row, P_k[i][i ^ x] = prod_q (I, X: 1; Y: -i / +i for row
// bit 0 / 1; Z: +1 / -1), so tr[P_k rho] = sum_i P_k[i][i ^ x] rho[i ^ x][i].
__global__ void __launch_bounds__(256)
pauli_vector_kernel(int n, long long total, const double* __restrict__ rho, double* __restrict__ out) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int d = 1 << n, DD = d * d;
    const long long item = gid / DD;
    const int k = (int)(gid % DD);
    int x = 0, z = 0, ny = 0;                   // X/Y positions, Z/Y positions, number of Y
    for (int q = 0; q < n; ++q) {
        const int op = (k >> (2 * (n - 1 - q))) & 3, bit = 1 << (n - 1 - q);
        if (op == 1 || op == 2) x |= bit;
        if (op == 2 || op == 3) z |= bit;
        ny += op == 2;
    }
    const double* r = rho + item * (long long)DD * 2;
    double re = 0.0, im = 0.0;
    for (int i = 0; i < d; ++i) {
        const int j = i ^ x;
        // Y on row bit b contributes -i (b = 0) or +i (b = 1) = -i * (-1)^b; Z contributes (-1)^b
        const double sgn = (__popc(i & z) & 1) ? -1.0 : 1.0;
        const double vr = r[2 * (j * d + i)], vi = r[2 * (j * d + i) + 1];
        re += sgn * vr; im += sgn * vi;
    }
    // times (-i)^ny
    double o;
    switch (ny & 3) { case 0: o = re; break; case 1: o = im; break; case 2: o = -re; break; default: o = -im; }
    out[gid] = o / d;
}

extern "C" {

// Every estimator / measure below comes as a device-pointer form (`_dev`: checks + launch on the library
// stream, no synchronisation) and a host-pointer form (staging buffers, H2D, the `_dev` form, D2H, sync).

int fbx_linv_state_dev(const fbx_design* design, int64_t B, const double* d_expect, double* d_rho_out) {
    FBX_TRY(check_state_design(design, "fbx_linv_state"));
    FBX_REQUIRE(B >= 0 && (B == 0 || (d_expect && d_rho_out)), "fbx_linv_state: bad batch / NULL buffer");
    FBX_TRY(ensure_device());
    if (B == 0) return FBX_OK;
    if (design->dev.n >= 4) return state_big(2, design, B, nullptr, d_expect, nullptr, d_rho_out);
    FBX_DISPATCH_NQ(design->dev.n, linv_state_kernel, 0, B, design->dev, (long long)B, d_expect, d_rho_out);
    FBX_HIP(hipGetLastError());
    return FBX_OK;
}

int fbx_linv_state(const fbx_design* design, int64_t B, const double* expect, double* rho_out) {
    FBX_TRY(check_state_design(design, "fbx_linv_state"));
    FBX_REQUIRE(B >= 0 && (B == 0 || (expect && rho_out)), "fbx_linv_state: bad batch / NULL buffer");
    FBX_TRY(ensure_device());
    if (B == 0) return FBX_OK;
    const size_t m = design->dev.m, D = design->dev.D;
    HostIO io; double *de, *dr;
    FBX_TRY(io.in(expect, m * B, &de)); FBX_TRY(io.out(D * 2 * B, &dr));
    FBX_TRY(fbx_linv_state_dev(design, B, de, dr));
    FBX_TRY(io.back(rho_out, dr, D * 2 * B));
    return io.sync();
}

int fbx_mle_state_dev(const fbx_design* design, int64_t B, const double* d_expect, const double* d_counts,
                      double epsilon, double entropy_penalty, double beta, double tol, int maxiter,
                      double* d_rho_out, int32_t* d_iters_out, int32_t* d_hit_max_out) {
    FBX_TRY(check_state_design(design, "fbx_mle_state"));
    FBX_REQUIRE(!(entropy_penalty != 0.0 && beta != 0.0),
                "One can't sensibly do entropy penalty and hedging. Do one or the other but not both.");
    FBX_REQUIRE(B >= 0 && (B == 0 || (d_expect && d_counts && d_rho_out)), "fbx_mle_state: bad batch / NULL buffer");
    FBX_TRY(ensure_device());
    if (B == 0) return FBX_OK;
    const int n = design->dev.n; const size_t m = design->dev.m, D = design->dev.D;
    if (n == 4) return launch_mle_big<4>(design, B, d_expect, d_counts, epsilon, entropy_penalty, beta, tol, maxiter, d_rho_out, d_iters_out, d_hit_max_out);
    if (n == 5) return launch_mle_big<5>(design, B, d_expect, d_counts, epsilon, entropy_penalty, beta, tol, maxiter, d_rho_out, d_iters_out, d_hit_max_out);
    const size_t lds = state_lds(n, (int)m);           // (designs beyond 64 KiB of per-setting staging take the streamed form: StateLds::staged)
    const bool packed = entropy_penalty == 0.0 && beta == 0.0 && n <= 2 && m <= D;
    if (packed && n == 1)
        hipLaunchKernelGGL(mle_state_packed_kernel<1>, dim3((unsigned)((B + 15) / 16)), dim3(64), 0, stream(), design->dev,
                           (long long)B, d_expect, epsilon, tol, maxiter, d_rho_out, d_iters_out, d_hit_max_out);
    else if (packed)
        hipLaunchKernelGGL(mle_state_packed_kernel<2>, dim3((unsigned)((B + 3) / 4)), dim3(64), 0, stream(), design->dev,
                           (long long)B, d_expect, epsilon, tol, maxiter, d_rho_out, d_iters_out, d_hit_max_out);
    else if (n == 3 && entropy_penalty == 0.0 && beta == 0.0 && m <= 64) {
        FBX_HIP(hipFuncSetAttribute((const void*)mle_state_plain3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(mle_state_plain3_kernel, dim3((unsigned)B), dim3(64), lds, stream(), design->dev, (long long)B, d_expect, d_counts,
                           epsilon, tol, maxiter, d_rho_out, d_iters_out, d_hit_max_out);
    } else
        FBX_DISPATCH_NQ(n, mle_state_kernel, lds, B, design->dev, (long long)B, d_expect, d_counts, epsilon, entropy_penalty,
                        beta, tol, maxiter, d_rho_out, d_iters_out, d_hit_max_out);
    FBX_HIP(hipGetLastError());
    return FBX_OK;
}

int fbx_mle_state(const fbx_design* design, int64_t B, const double* expect, const double* counts,
                  double epsilon, double entropy_penalty, double beta, double tol, int maxiter,
                  double* rho_out, int32_t* iters_out, int32_t* hit_max_out) {
    FBX_TRY(check_state_design(design, "fbx_mle_state"));
    FBX_REQUIRE(!(entropy_penalty != 0.0 && beta != 0.0),
                "One can't sensibly do entropy penalty and hedging. Do one or the other but not both.");
    FBX_REQUIRE(B >= 0 && (B == 0 || (expect && counts && rho_out)), "fbx_mle_state: bad batch / NULL buffer");
    FBX_TRY(ensure_device());
    if (B == 0) return FBX_OK;
    const size_t m = design->dev.m, D = design->dev.D;
    HostIO io; double *de, *dc, *dr; int32_t *di, *dh;
    FBX_TRY(io.in(expect, m * B, &de)); FBX_TRY(io.in(counts, m * B, &dc));
    FBX_TRY(io.out(D * 2 * B, &dr)); FBX_TRY(io.out((size_t)B, &di)); FBX_TRY(io.out((size_t)B, &dh));
    FBX_TRY(fbx_mle_state_dev(design, B, de, dc, epsilon, entropy_penalty, beta, tol, maxiter, dr, di, dh));
    FBX_TRY(io.back(rho_out, dr, D * 2 * B)); FBX_TRY(io.back(iters_out, di, (size_t)B));
    FBX_TRY(io.back(hit_max_out, dh, (size_t)B));
    return io.sync();
}

int fbx_r_operator_dev(const fbx_design* design, int64_t B, const double* d_rho, const double* d_expect, double* d_r_out) {
    FBX_TRY(check_state_design(design, "fbx_r_operator"));
    FBX_REQUIRE(B >= 0 && (B == 0 || (d_rho && d_expect && d_r_out)), "fbx_r_operator: bad batch / NULL buffer");
    FBX_TRY(ensure_device());
    if (B == 0) return FBX_OK;
    const int n = design->dev.n;
    if (n >= 4) return state_big(0, design, B, d_rho, d_expect, nullptr, d_r_out);
    const size_t lds = state_lds(n, (int)design->dev.m);
    FBX_DISPATCH_NQ(n, r_operator_kernel, lds, B, design->dev, (long long)B, d_rho, d_expect, d_r_out);
    FBX_HIP(hipGetLastError());
    return FBX_OK;
}

int fbx_r_operator(const fbx_design* design, int64_t B, const double* rho, const double* expect, double* r_out) {
    FBX_TRY(check_state_design(design, "fbx_r_operator"));
    FBX_REQUIRE(B >= 0 && (B == 0 || (rho && expect && r_out)), "fbx_r_operator: bad batch / NULL buffer");
    FBX_TRY(ensure_device());
    if (B == 0) return FBX_OK;
    const size_t m = design->dev.m, D = design->dev.D;
    HostIO io; double *dr, *de, *dout;
    FBX_TRY(io.in(rho, D * 2 * B, &dr)); FBX_TRY(io.in(expect, m * B, &de)); FBX_TRY(io.out(D * 2 * B, &dout));
    FBX_TRY(fbx_r_operator_dev(design, B, dr, de, dout));
    FBX_TRY(io.back(r_out, dout, D * 2 * B));
    return io.sync();
}

int fbx_state_log_likelihood_dev(const fbx_design* design, int64_t B, const double* d_rho, const double* d_expect,
                                 const double* d_counts, double* d_ll_out) {
    FBX_TRY(check_state_design(design, "fbx_state_log_likelihood"));
    FBX_REQUIRE(B >= 0 && (B == 0 || (d_rho && d_expect && d_counts && d_ll_out)), "fbx_state_log_likelihood: bad batch / NULL buffer");
    FBX_TRY(ensure_device());
    if (B == 0) return FBX_OK;
    const int n = design->dev.n;
    if (n >= 4) return state_big(1, design, B, d_rho, d_expect, d_counts, d_ll_out);
    FBX_DISPATCH_NQ(n, loglik_kernel, state_lds(n, 1), B, design->dev, (long long)B, d_rho, d_expect, d_counts, d_ll_out);
    FBX_HIP(hipGetLastError());
    return FBX_OK;
}

int fbx_state_log_likelihood(const fbx_design* design, int64_t B, const double* rho, const double* expect,
                             const double* counts, double* ll_out) {
    FBX_TRY(check_state_design(design, "fbx_state_log_likelihood"));
    FBX_REQUIRE(B >= 0 && (B == 0 || (rho && expect && counts && ll_out)), "fbx_state_log_likelihood: bad batch / NULL buffer");
    FBX_TRY(ensure_device());
    if (B == 0) return FBX_OK;
    const size_t m = design->dev.m, D = design->dev.D;
    HostIO io; double *dr, *de, *dc, *dout;
    FBX_TRY(io.in(rho, D * 2 * B, &dr)); FBX_TRY(io.in(expect, m * B, &de)); FBX_TRY(io.in(counts, m * B, &dc));
    FBX_TRY(io.out((size_t)B, &dout));
    FBX_TRY(fbx_state_log_likelihood_dev(design, B, dr, de, dc, dout));
    FBX_TRY(io.back(ll_out, dout, (size_t)B));
    return io.sync();
}

int fbx_matmul_dev(int N, int64_t B, const double* d_a, int conj_t_a, const double* d_scale, const double* d_b, int conj_t_b,
                   double* d_out) {
    FBX_REQUIRE(N >= 1 && N <= 1024, "fbx_matmul: N must be in 1..1024");
    FBX_REQUIRE(B >= 0 && (B == 0 || (d_a && d_b && d_out)), "fbx_matmul: bad batch / NULL buffer");
    FBX_TRY(ensure_device());
    if (B == 0) return FBX_OK;
    const long long tiles = (N + 15) / 16;
    FBX_REQUIRE(B * tiles * tiles < (1LL << 31), "fbx_matmul: batch too large for one launch");
    hipLaunchKernelGGL(matmul_kernel, dim3((unsigned)(B * tiles * tiles)), dim3(256), 0, stream(), N, (long long)B, d_a, conj_t_a,
                       d_scale, d_b, conj_t_b, d_out);
    FBX_HIP(hipGetLastError());
    return FBX_OK;
}

int fbx_matmul(int N, int64_t B, const double* a, int conj_t_a, const double* scale, const double* b, int conj_t_b, double* out) {
    FBX_REQUIRE(N >= 1 && N <= 1024, "fbx_matmul: N must be in 1..1024");
    FBX_REQUIRE(B >= 0 && (B == 0 || (a && b && out)), "fbx_matmul: bad batch / NULL buffer");
    FBX_TRY(ensure_device());
    if (B == 0) return FBX_OK;
    const size_t nn = (size_t)N * N * 2 * B;
    HostIO io; double *da, *db, *ds = nullptr, *dout;
    FBX_TRY(io.in(a, nn, &da)); FBX_TRY(io.in(b, nn, &db));
    if (scale) FBX_TRY(io.in(scale, (size_t)N * B, &ds));
    FBX_TRY(io.out(nn, &dout));
    FBX_TRY(fbx_matmul_dev(N, B, da, conj_t_a, ds, db, conj_t_b, dout));
    FBX_TRY(io.back(out, dout, nn));
    return io.sync();
}

int fbx_eigh_dev(int N, int64_t B, const double* d_a, double* d_w_out, double* d_v_out) {
    FBX_REQUIRE(N == 2 || N == 4 || N == 8 || N == 16 || N == 32 || N == 64 || (N > 64 && N <= 1024 && N % 2 == 0),
                "fbx_eigh_dev: N must be a power of two in 2..64 or an even number in 66..1024");
    FBX_REQUIRE(B >= 0 && (B == 0 || (d_a && d_w_out)), "fbx_eigh: bad batch / NULL buffer");
    FBX_TRY(ensure_device());
    if (B == 0) return FBX_OK;
    if (N > 64) {
        // few large matrices: one at a time over the whole chip; many: one CU each
        // (measured: one CU per matrix 14 / 108 / 1270 / 9300 ms for N = 128 / 256 / 512 / 1024, whatever the batch up to
        // the number of CUs; the whole chip on one matrix 6 / 25 / 200 / 820 ms each)
        const int64_t coop_up_to = N >= 768 ? 10 : N >= 384 ? 5 : 3;
        if (N >= 128 && B <= coop_up_to) { bool done = false; FBX_TRY(launch_eigh_coop(N, B, d_a, d_w_out, d_v_out, &done)); if (done) return FBX_OK; }
        return launch_eigh_big(N, B, d_a, d_w_out, d_v_out);
    }
    switch (N) {
        case 2: FBX_TRY(launch_eigh<2>(B, d_a, d_w_out, d_v_out)); break;
        case 4: FBX_TRY(launch_eigh<4>(B, d_a, d_w_out, d_v_out)); break;
        case 8: FBX_TRY(launch_eigh<8>(B, d_a, d_w_out, d_v_out)); break;
        case 16: FBX_TRY(launch_eigh<16>(B, d_a, d_w_out, d_v_out)); break;
        case 32: FBX_TRY(launch_eigh<32>(B, d_a, d_w_out, d_v_out)); break;
        default: FBX_TRY(launch_eigh<64>(B, d_a, d_w_out, d_v_out)); break;
    }
    FBX_HIP(hipGetLastError());
    return FBX_OK;
}

int fbx_eigh(int N, int64_t B, const double* a, double* w_out, double* v_out) {
    FBX_REQUIRE(N >= 1 && N <= 1024, "fbx_eigh: N must be in 1..1024");
    FBX_REQUIRE(B >= 0 && (B == 0 || (a && w_out)), "fbx_eigh: bad batch / NULL buffer");
    FBX_TRY(ensure_device());
    if (B == 0) return FBX_OK;
    int Np = 2;
    while (Np < N) Np *= 2;
    if (N > 64) Np = N + (N & 1);           // the HBM-resident solver takes any even size
    if (Np == N) {
        const size_t nn = (size_t)N * N * 2 * B;
        HostIO io; double *da, *dw, *dv = nullptr;
        FBX_TRY(io.in(a, nn, &da)); FBX_TRY(io.out((size_t)N * B, &dw));
        if (v_out) FBX_TRY(io.out(nn, &dv));
        FBX_TRY(fbx_eigh_dev(N, B, da, dw, dv));
        FBX_TRY(io.back(w_out, dw, (size_t)N * B)); FBX_TRY(io.back(v_out, dv, nn));
        return io.sync();
    }
    // Any other size (e.g. a qutrit's 3 x 3, a 9 x 9 Choi matrix): embedded in the next power of two
    // with zero rows / columns.  The padding coordinates are decoupled and stay so exactly (a pivot
    // with a zero off-diagonal entry gets the identity rotation), so their eigenvectors come back as
    // unit vectors on the padding coordinates and are dropped here; the rest is the decomposition
    // of the N x N matrix, still ascending.
    const size_t np2 = (size_t)Np * Np;
    std::vector<double> ap(np2 * 2 * B, 0.0), wp((size_t)Np * B), vp(np2 * 2 * B);
    for (int64_t b = 0; b < B; ++b)
        for (int r = 0; r < N; ++r)
            memcpy(&ap[(b * np2 + (size_t)r * Np) * 2], &a[((size_t)b * N * N + (size_t)r * N) * 2], sizeof(double) * 2 * N);
    {
        HostIO io; double *da, *dw, *dv;
        FBX_TRY(io.in(ap.data(), ap.size(), &da)); FBX_TRY(io.out(wp.size(), &dw)); FBX_TRY(io.out(vp.size(), &dv));
        FBX_TRY(fbx_eigh_dev(Np, B, da, dw, dv));
        FBX_TRY(io.back(wp.data(), dw, wp.size())); FBX_TRY(io.back(vp.data(), dv, vp.size()));
        FBX_TRY(io.sync());
    }
    for (int64_t b = 0; b < B; ++b) {
        int kept = 0;
        for (int k = 0; k < Np; ++k) {
            bool padding = false;
            for (int r = N; r < Np && !padding; ++r) {
                const double* e = &vp[(b * np2 + (size_t)r * Np + k) * 2];
                padding = e[0] != 0.0 || e[1] != 0.0;
            }
            if (padding) continue;
            if (kept < N) {
                w_out[b * N + kept] = wp[b * Np + k];
                if (v_out)
                    for (int r = 0; r < N; ++r) {
                        v_out[((size_t)b * N * N + (size_t)r * N + kept) * 2] = vp[(b * np2 + (size_t)r * Np + k) * 2];
                        v_out[((size_t)b * N * N + (size_t)r * N + kept) * 2 + 1] = vp[(b * np2 + (size_t)r * Np + k) * 2 + 1];
                    }
            }
            ++kept;
        }
        if (kept != N) { set_error("fbx_eigh: internal error separating the padding of a non-power-of-two matrix"); return FBX_ERR_HIP; }
    }
    return FBX_OK;
}

// ---- choi2kraus for a batch (superoperator_transformations.py:325-336): fbx_eigh_dev + one assembling kernel.
// One workgroup per item.  Eigenpair k is kept when |lambda_k| > tol (the reference's test); its operator is
// sqrt(lambda_k) unvec(v_k) -- numpy's scimath square root, i sqrt(|lambda|) for a negative eigenvalue -- with the phase of v_k
// fixed so that its first component above 1e-12 ||v_k|| is real and positive (the convention of the host form this replaces,
// fbx/operator_tools/superoperator_transformations.py: what LAPACK hands the reference on the operators its tests compare
// entry by entry).  unvec is column stacking: K[r][c] = v[c d + r].  Kept operators are packed at the front of the item's D
// slots in ascending eigenvalue order (the order of the reference's list), the other slots are zeroed.
__global__ void __launch_bounds__(256)
kraus_assemble_kernel(int D, int d, long long B, const double* __restrict__ w, const double* __restrict__ V, double tol,
                      double* __restrict__ out, int* __restrict__ count) {
    extern __shared__ __attribute__((aligned(16))) char kraus_smem[];
    double* fre = (double*)kraus_smem;
    double* fim = fre + D;
    int* keep = (int*)(fim + D);
    int* pos = keep + D;
    const long long b = blockIdx.x;
    const int tid = threadIdx.x;
    const cplx* Vb = (const cplx*)V + (size_t)b * D * D;
    const double* wb = w + (size_t)b * D;
    for (int k = tid; k < D; k += 256) {
        const double ev = wb[k];
        const bool kp = fabs(ev) > tol;
        double fr = 0.0, fi = 0.0;
        if (kp) {
            double n2 = 0.0;
            for (int i = 0; i < D; ++i) { const cplx x = Vb[(size_t)i * D + k]; n2 = fma(x.re, x.re, fma(x.im, x.im, n2)); }
            const double thr = 1e-12 * sqrt(n2);
            double pr = 1.0, pi = 0.0;
            for (int i = 0; i < D; ++i) {
                const cplx x = Vb[(size_t)i * D + k];
                const double a = sqrt(fma(x.re, x.re, x.im * x.im));
                if (a > thr) { pr = x.re / a; pi = -x.im / a; break; }       // |x| / x
            }
            const double sq = sqrt(fabs(ev));
            if (ev >= 0.0) { fr = pr * sq; fi = pi * sq; } else { fr = -pi * sq; fi = pr * sq; }    // i (pr + i pi)
        }
        keep[k] = kp ? 1 : 0; fre[k] = fr; fim[k] = fi;
    }
    __syncthreads();
    for (int k = tid; k < D; k += 256) { int c = 0; for (int j = 0; j < k; ++j) c += keep[j]; pos[k] = c; }
    __syncthreads();
    const int total = pos[D - 1] + keep[D - 1];
    cplx* ob = (cplx*)out + (size_t)b * D * D;
    for (int idx = tid; idx < D * D; idx += 256) {
        const int i = idx / D, k = idx % D;                    // k fastest: coalesced reads of V[i][.]
        if (!keep[k]) continue;
        const cplx x = Vb[idx];
        const int r = i % d, c = i / d;
        cplx o; o.re = fre[k] * x.re - fim[k] * x.im; o.im = fre[k] * x.im + fim[k] * x.re;
        ob[((size_t)pos[k] * d + r) * d + c] = o;
    }
    for (int idx = total * D + tid; idx < D * D; idx += 256) { cplx z; z.re = 0.0; z.im = 0.0; ob[idx] = z; }
    if (tid == 0 && count) count[b] = total;
}

int fbx_choi2kraus_dev(int n_qubits, int64_t B, const double* d_choi, double tol, double* d_kraus_out, int32_t* d_count_out) {
    FBX_REQUIRE(n_qubits >= 1 && n_qubits <= 5, "fbx_choi2kraus: n_qubits must be 1..5");
    FBX_REQUIRE(B >= 0 && (B == 0 || (d_choi && d_kraus_out)), "fbx_choi2kraus: bad batch / NULL buffer");
    FBX_REQUIRE(tol >= 0.0, "fbx_choi2kraus: negative tolerance");
    FBX_TRY(ensure_device());
    if (B == 0) return FBX_OK;
    const int d = 1 << n_qubits, D = d * d;
    // the eigenvectors of a block of items at a time: 16 MiB per 5-qubit item
    const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(B, ((int64_t)1 << 28) / ((int64_t)D * D * 16)));
    DevBuf dw, dv;
    FBX_TRY(dw.alloc(sizeof(double) * D * (size_t)chunk));
    FBX_TRY(dv.alloc(sizeof(double) * 2 * D * D * (size_t)chunk));
    for (int64_t b0 = 0; b0 < B; b0 += chunk) {
        const int64_t nb = std::min(chunk, B - b0);
        FBX_TRY(fbx_eigh_dev(D, nb, d_choi + (size_t)b0 * D * D * 2, dw.as<double>(), dv.as<double>()));
        hipLaunchKernelGGL(kraus_assemble_kernel, dim3((unsigned)nb), dim3(256), (size_t)D * 24, stream(), D, d, (long long)nb,
                           dw.as<double>(), dv.as<double>(), tol, d_kraus_out + (size_t)b0 * D * D * 2,
                           d_count_out ? d_count_out + b0 : nullptr);
        FBX_HIP(hipGetLastError());
    }
    return FBX_OK;
}

int fbx_choi2kraus(int n_qubits, int64_t B, const double* choi, double tol, double* kraus_out, int32_t* count_out) {
    FBX_REQUIRE(n_qubits >= 1 && n_qubits <= 5, "fbx_choi2kraus: n_qubits must be 1..5");
    FBX_REQUIRE(B >= 0 && (B == 0 || (choi && kraus_out)), "fbx_choi2kraus: bad batch / NULL buffer");
    FBX_TRY(ensure_device());
    if (B == 0) return FBX_OK;
    const size_t D = (size_t)1 << (2 * n_qubits), nn = D * D * 2 * (size_t)B;
    HostIO io; double *dc, *dk; int32_t* dn;
    FBX_TRY(io.in(choi, nn, &dc)); FBX_TRY(io.out(nn, &dk)); FBX_TRY(io.out((size_t)B, &dn));
    FBX_TRY(fbx_choi2kraus_dev(n_qubits, B, dc, tol, dk, dn));
    FBX_TRY(io.back(kraus_out, dk, nn)); FBX_TRY(io.back(count_out, dn, (size_t)B));
    return io.sync();
}

int fbx_proj_state_physical_dev(int n_qubits, int64_t B, const double* d_rho, double* d_out) {
    FBX_REQUIRE(n_qubits >= 1 && n_qubits <= 3, "fbx_proj_state_physical: n_qubits must be 1..3");
    FBX_REQUIRE(B >= 0 && (B == 0 || (d_rho && d_out)), "fbx_proj_state_physical: bad batch / NULL buffer");
    FBX_TRY(ensure_device());
    if (B == 0) return FBX_OK;
    FBX_DISPATCH_NQ(n_qubits, proj_state_kernel, state_lds(n_qubits, 1), B, (long long)B, d_rho, d_out);
    FBX_HIP(hipGetLastError());
    return FBX_OK;
}

int fbx_proj_state_physical(int n_qubits, int64_t B, const double* rho, double* out) {
    FBX_REQUIRE(n_qubits >= 1 && n_qubits <= 3, "fbx_proj_state_physical: n_qubits must be 1..3");
    FBX_REQUIRE(B >= 0 && (B == 0 || (rho && out)), "fbx_proj_state_physical: bad batch / NULL buffer");
    FBX_TRY(ensure_device());
    if (B == 0) return FBX_OK;
    const size_t d = (size_t)1 << n_qubits, D = d * d;
    HostIO io; double *dr, *dout;
    FBX_TRY(io.in(rho, D * 2 * B, &dr)); FBX_TRY(io.out(D * 2 * B, &dout));
    FBX_TRY(fbx_proj_state_physical_dev(n_qubits, B, dr, dout));
    FBX_TRY(io.back(out, dout, D * 2 * B));
    return io.sync();
}

int fbx_state_measures_dev(int n_qubits, int64_t B, const double* d_rho, const double* d_sigma, double* d_purity_out,
                           double* d_fidelity_out, double* d_trace_dist_out, double* d_hs_ip_out) {
    FBX_REQUIRE(n_qubits >= 1 && n_qubits <= 3, "fbx_state_measures: n_qubits must be 1..3");
    FBX_REQUIRE(B >= 0 && (B == 0 || (d_rho && d_sigma)), "fbx_state_measures: bad batch / NULL buffer");
    FBX_TRY(ensure_device());
    if (B == 0) return FBX_OK;
    FBX_DISPATCH_NQ(n_qubits, state_measures_kernel, state_lds(n_qubits, 1), B, (long long)B, d_rho, d_sigma, d_purity_out,
                    d_fidelity_out, d_trace_dist_out, d_hs_ip_out);
    FBX_HIP(hipGetLastError());
    return FBX_OK;
}

int fbx_state_measures(int n_qubits, int64_t B, const double* rho, const double* sigma, double* purity_out,
                       double* fidelity_out, double* trace_dist_out, double* hs_ip_out) {
    FBX_REQUIRE(n_qubits >= 1 && n_qubits <= 3, "fbx_state_measures: n_qubits must be 1..3");
    FBX_REQUIRE(B >= 0 && (B == 0 || (rho && sigma)), "fbx_state_measures: bad batch / NULL buffer");
    FBX_TRY(ensure_device());
    if (B == 0) return FBX_OK;
    const size_t d = (size_t)1 << n_qubits, D = d * d;
    HostIO io; double *dr, *ds, *dp = nullptr, *df = nullptr, *dt = nullptr, *dh = nullptr;
    FBX_TRY(io.in(rho, D * 2 * B, &dr)); FBX_TRY(io.in(sigma, D * 2 * B, &ds));
    if (purity_out) FBX_TRY(io.out((size_t)B, &dp));
    if (fidelity_out) FBX_TRY(io.out((size_t)B, &df));
    if (trace_dist_out) FBX_TRY(io.out((size_t)B, &dt));
    if (hs_ip_out) FBX_TRY(io.out((size_t)B, &dh));
    FBX_TRY(fbx_state_measures_dev(n_qubits, B, dr, ds, dp, df, dt, dh));
    FBX_TRY(io.back(purity_out, dp, (size_t)B)); FBX_TRY(io.back(fidelity_out, df, (size_t)B));
    FBX_TRY(io.back(trace_dist_out, dt, (size_t)B)); FBX_TRY(io.back(hs_ip_out, dh, (size_t)B));
    return io.sync();
}

int fbx_pauli_vector_dev(int n_qubits, int64_t B, const double* d_rho, double* d_out) {
    FBX_REQUIRE(n_qubits >= 1 && n_qubits <= 5, "fbx_pauli_vector: n_qubits must be 1..5");
    FBX_REQUIRE(B >= 0 && (B == 0 || (d_rho && d_out)), "fbx_pauli_vector: bad batch / NULL buffer");
    FBX_TRY(ensure_device());
    if (B == 0) return FBX_OK;
    const long long total = (long long)B << (2 * n_qubits);
    hipLaunchKernelGGL(pauli_vector_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream(), n_qubits, total, d_rho, d_out);
    FBX_HIP(hipGetLastError());
    return FBX_OK;
}

int fbx_pauli_vector(int n_qubits, int64_t B, const double* rho, double* out) {
    FBX_REQUIRE(n_qubits >= 1 && n_qubits <= 5, "fbx_pauli_vector: n_qubits must be 1..5");
    FBX_REQUIRE(B >= 0 && (B == 0 || (rho && out)), "fbx_pauli_vector: bad batch / NULL buffer");
    FBX_TRY(ensure_device());
    if (B == 0) return FBX_OK;
    const size_t DD = (size_t)1 << (2 * n_qubits);
    HostIO io; double *dr, *dout;
    FBX_TRY(io.in(rho, DD * 2 * B, &dr)); FBX_TRY(io.out(DD * B, &dout));
    FBX_TRY(fbx_pauli_vector_dev(n_qubits, B, dr, dout));
    FBX_TRY(io.back(out, dout, DD * B));
    return io.sync();
}

}  // extern "C"
