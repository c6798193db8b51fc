// fbx_pgdb_body.hpp -- the 1- / 2-qubit PGDB reconstruction of one wavefront (pgdb_body) and its LDS layout, shared by the two
// translation units that instantiate it: fbx_pgdb.hip (the one-wavefront-per-SIMD kernel, the launchers and the C ABI) and
// fbx_pgdb_lean.hip (the two-wavefronts-per-SIMD kernel).  Two units because they are COMPILED DIFFERENTLY: the instruction
// scheduler's max-ILP strategy is worth 3 % on the lone wavefront of the first kernel and costs the register-starved second one
// half its speed (build.py: per-file flags).
#pragma once
#include "fbx_choi.hpp"
#include <cstdlib>
#include <vector>
#define FBX_LEAN_MIN_BATCH 1025     // batch size from which the two-waves-per-SIMD kernel is used (2 qubits): one reconstruction more than
                                    // the chip has SIMDs.  The one-wave kernel is the faster chain (1024 items: 12.7 against 15.3 ms), but item
                                    // 1025 starts a second round on it (1152 items: 19.8 against 16.1 ms; scripts/lean_crossover.py; round 3's
                                    // crossover, before the pieces, lay between 1100 and 1280)
#define FBX_PACKED_1Q_MIN_BATCH 8192  // single-qubit batches from which the lane-per-item kernel is used (fbx_pgdb1.hip)
#define FBX_BASIS_CHAIN_SWEEPS 216  // Jacobi sweeps a chain of stored bases may accumulate (per slot, on average) before a cold restart
#define FBX_BASIS_STEP 1e-3
#define FBX_BASIS_WRITE_STEP 3e-2     // outer step below which every Dykstra basis is written back (fbx_choi.hpp BasisStore)

namespace fbx {

constexpr double PGDB_EPS = 1e-6;     // probability clip, tomography.py:597,613,631
constexpr double PGDB_GAMMA = 0.3;    // tomography.py:567
constexpr double PGDB_STOP = 1e-10;   // tomography.py:589
constexpr double PGDB_ALPHA_MIN = 1e-15;  // tomography.py:584
#define FBX_SMALL_STEP_LIMIT 0x1p-3      // alpha * max |pu / pe| below which the line search uses the power-sum series

// LEAN (2 waves per SIMD at large batches): 19.4 KB instead of 39 KB per reconstruction, so that eight wavefronts share a CU's
// 160 KB.  The Bloch matrix is read through L2 (DesignDev::Ct); the normalised counts n+- live in REGISTERS (2 MAXJ doubles per
// lane) and the model probabilities of an outcome are recomputed from the two prediction tables wherever they are used (four
// ds_read_b64 and four flops per setting -- the one-wave kernel keeps them in 8 MAXJ registers); the Pauli-coefficient matrix Rb
// aliases the tail of Vs, which is dead whenever Rb is live (outside a projection; the transforms stage through the front of
// Ms + Vs), and the gradient is parked in Tupd alone while the projection runs.  Round 3's form kept ONE table and the counts in
// an L2 workspace: 72 + registers of per-outcome probabilities, 216 spilled registers and ~9 KB of L2 reads per cost evaluation.
template <int NQ, bool LEAN = false>
struct PgdbLds {
    ChoiLds<NQ, LEAN, !LEAN> choi;      // the one-wave body: two workers per upper block in the 16 x 16 eigensolver (fbx_choi.hpp)
    double* Rb;     // [D*D]  Pauli coefficients (one matrix at a time), TRANSPOSED: Rb[j * D + i] = R[i][j]   (LEAN: inside Vs)
    double* Test;   // [S*D]  predicted tr(P_i E(rho_s)) for the current estimate
    double* Tupd;   // [S*D]  same for the update direction; reused as Wt[S][D] in the gradient
    double* Cl;     // [S*D]  Bloch coefficients of the input states, one state per row: Cl[s * D + j] = C[j][s]   (LEAN: none)
    double* Ln;     // [2*ceil(m/64)][64]  normalised counts n+ / n- of this lane's outcomes (row 2 j + sign): item
                    // constants that are only read by the cost / gradient passes -- kept here, not in 36 registers (LEAN: registers)
    double* park;   // [2*D*D] the gradient block of every lane while the projection runs: Rb + Tupd (adjacent, both dead
                    // meanwhile; Tupd is sized for at least D states); LEAN: Tupd alone (sized for at least 2 D states)
    static constexpr int D = ChoiLds<NQ>::D;
    static constexpr size_t tupd_rows(int S) { return LEAN ? (S > 2 * D ? S : 2 * D) : (S > D ? S : D); }
    static size_t bytes(int S, int m) {
        const size_t Su = tupd_rows(S);
        const size_t base = (ChoiLds<NQ, LEAN>::bytes() + 15) & ~(size_t)15;
        if (LEAN) return base + sizeof(double) * ((Su + (size_t)S) * D) + 64;
        return base + sizeof(double) * ((size_t)D * D + (Su + 2 * (size_t)S) * D + 2 * (size_t)((m + 63) / 64) * 64) + 64;
    }
    // every pointer is a plain offset from the start of the dynamic LDS segment (no conditional
    // layout), so the compiler keeps them in the LDS address space (ds_* instead of flat_*)
    __device__ void carve(char* p, int S, int m) {
        char* q = p;
        choi.carve(q);
        // (rounding the POINTER up through an integer cast would turn everything behind it into
        // generic-address-space pointers: flat_load / flat_store instead of ds_read / ds_write)
        constexpr size_t aligned = (ChoiLds<NQ, LEAN>::bytes() + 15) & ~(size_t)15;
        p += aligned;
        if constexpr (LEAN) {
            static_assert(sizeof(cplx) * D * (D + 1) + sizeof(double) * D * D <= 2 * sizeof(cplx) * sys_elems<D>(),
                          "the transforms' staging matrix and Rb must both fit into Ms + Vs");
            Rb = (double*)(choi.Vs + sys_elems<D>()) - D * D;
            Tupd = (double*)p; p += sizeof(double) * tupd_rows(S) * D;
            Test = (double*)p; p += sizeof(double) * S * D;
            Cl = nullptr; Ln = nullptr; park = Tupd;
        } else {
            Rb = (double*)p; p += sizeof(double) * D * D;
            Tupd = (double*)p; p += sizeof(double) * tupd_rows(S) * D;
            Test = (double*)p; p += sizeof(double) * S * D;
            Cl = (double*)p; p += sizeof(double) * D * S;
            Ln = (double*)p; p += sizeof(double) * 2 * ((m + 63) / 64) * 64;
            park = Rb;
        }
    }
};

// LEAN: the design's Bloch table [S][D] fits at the front of Ms + Vs, in front of Rb (2 qubits: up to 50 input states)
template <int NQ>
__host__ __device__ constexpr bool pgdb_ct_fits(int S) {
    constexpr int D = 1 << (2 * NQ);
    return sizeof(double) * D * (size_t)S + sizeof(double) * D * D <= 2 * sizeof(cplx) * sys_elems<D>();
}

// T[s][i] = sum_j R[i][j] * C[j][s].  Lane (i = lane % D, q = lane / D) keeps row i of R in registers
// (Rb is transposed, so the D loads are conflict-free across i) and walks the states s = q, q + 64/D,
// ...: per state D/2 broadcast 16-byte loads of the state's Bloch vector (Ct is [S][D]) and D FMAs, all
// straight-line code -- the D x S loop over (s, i) pairs with two 8-byte loads per FMA it replaces
// took 16k cycles per call for the 36-state design.
template <int NQ>
__device__ void predict_table(const double* Rb, const double* Ct, double* T, int S, int lane) {
    constexpr int D = ChoiLds<NQ>::D, STEP = 64 / D;
    const int i = lane % D, q = lane / D;
    double r[D];
#pragma unroll
    for (int j = 0; j < D; ++j) r[j] = Rb[j * D + i];
    for (int s = q; s < S; s += STEP) {
        const double2* c2 = reinterpret_cast<const double2*>(Ct + (size_t)s * D);
        double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
        for (int j = 0; j < D / 2; ++j) {
            const double2 c = c2[j];
            acc0 = fma(r[2 * j], c.x, acc0);
            acc1 = fma(r[2 * j + 1], c.y, acc1);
        }
        T[s * D + i] = acc0 + acc1;
    }
}

// record of a paused reconstruction (pieces, below): the estimate's blocks [8][64], then 13 scalars
constexpr int PGDB_REC_EST = 8 * 64, PGDB_REC = PGDB_REC_EST + 16;

// TPC: the kind of projection as a COMPILE-TIME constant (1 = trace preserving, 0 = trace non-increasing) or -1 = the run-time
// argument.  With both kinds compiled in, the trace-non-increasing branch (a d x d eigendecomposition of the partial trace inside every
// Dykstra iteration) costs the register-capped two-waves kernel 35 spilled registers it never uses in a trace-preserving run: its
// instantiations take the kind as a template argument (fbx_pgdb_lean.hip; 8192 items 50.9 -> 49.0 ms, 65 536 items 402 -> 387 ms, same box,
// bit-identical).  The one-wave kernels (no spills either way, same time) keep the run-time form.  (With the registers that freed, the
// two-workers form of the 16 x 16 Jacobi was tried in this kernel again: 48.9 against 49.0 ms -- not adopted.)
template <int NQ, int MAXJ, bool LEAN, bool PIECES = false, int TPC = -1>
__device__ __forceinline__ void
pgdb_body(char* smem, long long item_, const DesignDev& des, long long B, const double* __restrict__ expect,
          const double* __restrict__ counts, int trace_preserving, int mode, int max_iters,
          double* __restrict__ choi_out, int* __restrict__ iters_out,
          int* __restrict__ dykstra_out, int* __restrict__ backtracks_out,
          double* __restrict__ cost_out, int* __restrict__ work_out,
          long long* __restrict__ phase_out, cplx* __restrict__ basis_scratch, int basis_cap,
          double* __restrict__ ncounts, int* __restrict__ trace_out, int trace_iters,
          double* __restrict__ rec = nullptr, int piece_stop = 0x7fffffff, bool resume = false) {
    constexpr int d = 1 << NQ, D = d * d, LD = D + 1, NB = D / 2, NACT = NB * NB;
    // STREAM (LEAN with MAXJ = 0; round 5): designs of ANY number of settings -- merged or repeated datasets (the reference loops
    // over whatever result list it is given, tomography.py:494-539).  No per-slot register arrays at all: the number of outcome
    // slots per lane is a run-time value, the normalised counts and design words of slot j are read from HBM / L2 where they
    // are used (the item's slice of the ncounts workspace, the design's table), the probabilities come from the two tables as
    // in every LEAN instantiation.  Slow (every pass over the outcomes streams 20 bytes per setting through L2) and
    // complete; designs that fit the register-resident instantiations never get here (pgdb_dispatch).
    constexpr bool STREAM = LEAN && MAXJ == 0;
    static_assert(!STREAM || !PIECES, "the streamed instantiation runs whole reconstructions");
    constexpr int MJ = STREAM ? 1 : MAXJ;           // extent of the per-slot register arrays
    // The passes over a lane's MAXJ outcome slots (cost, gradient weights, clip detection, power sums) are straight-line code in
    // the one-wave kernel and LOOPS in the lean one (the slot index is wave-uniform: register arrays are indexed through
    // s_set_gpr_idx): unrolled they are 6.4 KB of code per slot -- 58 of the 91 KB of the 540-setting instantiation, against a
    // 64 KB instruction cache that eight wavefronts in eight different phases share.
    constexpr int SLOT_UNROLL = LEAN ? 1 : MAXJ;
    int lane = threadIdx.x & 63;
    const long long item = item_;
    const int m = des.m, S = des.S;
    const int nslots = STREAM ? (m + 63) / 64 : MAXJ;       // outcome slots per lane (a compile-time constant unless STREAM)
    PgdbLds<NQ, LEAN> L;
    L.carve(smem, S, 64 * MAXJ);

    // Bloch coefficients, one state per row.  One-wave kernel: an LDS copy for the whole reconstruction.  LEAN: a TRANSIENT copy at
    // the front of Ms + Vs (where the transforms stage: dead as soon as Rb holds the coefficients), re-read from the design's
    // [S][D] table (4.6 KB, shared by the batch: L1 / L2 hits, five 16-byte loads per lane) in front of each table product --
    // the products then run from LDS like the one-wave kernel's (through L2 they were latency-bound loops: round 3's lean kernel
    // spent 3.7 x the one-wave kernel's cycles in the transform / table phases, profiles/r04/phase_*.txt).  pgdb_lean_eligible()
    // keeps designs whose table does not fit beside Rb (more than 50 states) on the one-wave kernel.
    const double* Ct;
    // (STREAM: a design whose Bloch table does not fit beside Rb -- more than 50 input states for two qubits -- reads it through L2)
    const bool ct_staged = !STREAM || pgdb_ct_fits<NQ>(S);
    if constexpr (STREAM) Ct = ct_staged ? (const double*)L.choi.Ms : des.Ct;
    else if constexpr (LEAN) Ct = (const double*)L.choi.Ms;
    else {
        for (int idx = lane; idx < D * S; idx += 64) L.Cl[(idx % S) * D + idx / S] = des.C[idx];     // des.C is [D][S]
        Ct = L.Cl;
    }
    auto stage_ct = [&]() __attribute__((always_inline)) {       // LEAN: des.Ct -> front of Ms + Vs (both sides 16-byte aligned)
        if (LEAN && ct_staged) {
            typedef __attribute__((address_space(1))) const fbx_v2d* gptr;
            const gptr src = (gptr)des.Ct;
            fbx_v2d* dst = (fbx_v2d*)L.choi.Ms;
            for (int idx = lane; idx < (D / 2) * S; idx += 64) dst[idx] = src[idx];
        }
    };

    // ---- data: n+-[k] = counts * (1 +- e)/2 / grand_total   (tomography.py:528-538)
    double tot = 0.0;
    // LEAN: the normalised counts and the design words of this lane's MAXJ slots are REGISTER copies that live from the
    // gradient to the projection and from the projection to the end of the line search -- not across the projection, where the
    // Dykstra state and the eigensolver need the registers: load_slots() re-reads them (coalesced rows of the item's slice of
    // an L2-resident workspace; the design's own table) in front of the gradient pass and in front of the line search.
    double nreg_p[LEAN ? MJ : 1], nreg_m[LEAN ? MJ : 1];
    const __amdgpu_buffer_rsrc_t nc_rsrc = buf_rsrc(ncounts, LEAN ? 2u * (unsigned)nslots * 64u * 8u : 0u);      // (raw buffer rows: fbx_common.hpp)
    const __amdgpu_buffer_rsrc_t sp_rsrc = buf_rsrc(des.sp, 4u * (unsigned)m);                   // settings beyond m read as 0
    if constexpr (STREAM) {        // two passes: the grand total (same order of additions as below), then the normalised counts
        for (int j = 0; j < nslots; ++j) {
            const int g = lane + 64 * j;
            if (g < m) tot += counts[item * m + des.order[g]];
        }
        tot = uniform(wave_sum(tot));
        for (int j = 0; j < nslots; ++j) {
            const int g = lane + 64 * j;
            double np_ = 0.0, nm_ = 0.0;
            if (g < m) {
                const int k = des.order[g];
                const double e = expect[item * m + k], c = counts[item * m + k];
                const double plus = (1.0 + e) / 2.0;
                np_ = c * plus; nm_ = c * (1.0 - plus);
            }
            buf_store_f64(nc_rsrc, 8u * lane, 512u * (2 * j), np_ / tot); buf_store_f64(nc_rsrc, 8u * lane, 512u * (2 * j + 1), nm_ / tot);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (this wavefront reads them back through the same L2)
    } else {
        double npl[MJ], nmi[MJ];
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) {
            const int g = lane + 64 * j;
            npl[j] = 0.0; nmi[j] = 0.0;
            if (g < m) {
                const int k = des.order[g];
                const double e = expect[item * m + k], c = counts[item * m + k];
                const double plus = (1.0 + e) / 2.0;
                npl[j] = c * plus; nmi[j] = c * (1.0 - plus);
                tot += c;
            }
        }
        tot = uniform(wave_sum(tot));
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) {
            if constexpr (LEAN) {
                buf_store_f64(nc_rsrc, 8u * lane, 512u * (2 * j), npl[j] / tot); buf_store_f64(nc_rsrc, 8u * lane, 512u * (2 * j + 1), nmi[j] / tot);
            } else { L.Ln[(2 * j) * 64 + lane] = npl[j] / tot; L.Ln[(2 * j + 1) * 64 + lane] = nmi[j] / tot; }
        }
    }
    if constexpr (LEAN) {        // the update direction's table is read (times alpha = 0) by the very first cost evaluation
        for (int idx = lane; idx < S * D; idx += 64) L.Tupd[idx] = 0.0;
    }
    FBX_WAVE_SYNC();
    // normalised counts of slot j
    auto counts_of = [&](int j, double& np_, double& nm_) __attribute__((always_inline)) {
        if constexpr (STREAM) { np_ = buf_load_f64(nc_rsrc, 8u * lane, 512u * (2 * j)); nm_ = buf_load_f64(nc_rsrc, 8u * lane, 512u * (2 * j + 1)); }
        else if constexpr (LEAN) { np_ = nreg_p[j]; nm_ = nreg_m[j]; }
        else { np_ = L.Ln[(2 * j) * 64 + lane]; nm_ = L.Ln[(2 * j + 1) * 64 + lane]; }
    };

    const double half_dd = 0.5 / (double)(d * d);      // 1 / (2 d^2)
    const double inv_mu = (2.0 * d * d) / 3.0;          // 1 / mu, mu = 3 / (2 d^2)

    // per-setting design words: in registers for the whole reconstruction (LEAN: see load_slots)
    uint32_t spw[MJ];
    if constexpr (!STREAM) {
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) {
            const int g = lane + 64 * j;
            spw[j] = g < m ? des.sp[g] : 0u;
        }
    }
    auto spw_of = [&](int j) __attribute__((always_inline)) -> uint32_t {
        if constexpr (STREAM) return buf_load_u32(sp_rsrc, 4u * lane, 256u * j); else return spw[j];
    };
    auto load_slots = [&]() __attribute__((always_inline)) {
        if constexpr (LEAN && !STREAM) {
            // (the compiler must neither keep the previous copies alive across the projection nor hoist these loads above it)
            __asm__ volatile("" ::: "memory");
#pragma unroll
            for (int j = 0; j < MAXJ; ++j) {
                nreg_p[j] = buf_load_f64(nc_rsrc, 8u * lane, 512u * (2 * j)); nreg_m[j] = buf_load_f64(nc_rsrc, 8u * lane, 512u * (2 * j + 1));
                spw[j] = buf_load_u32(sp_rsrc, 4u * lane, 256u * j);
            }
        }
    };
    const bool unit_coefs = des.unit_coefs != 0;      // wave-uniform: coefficients re-read only when needed
    // model probabilities of an owned setting's two outcomes from a prediction table (`missing` for lanes without one)
    auto table_probs = [&](const double* T, int j, double missing, double& a, double& b) __attribute__((always_inline)) {
        const int g = lane + 64 * j;
        a = missing; b = missing;
        if (g < m) {
            const uint32_t dw = spw_of(j);
            const int s = dw >> 16, p = dw & 0xffff;
            const double cf = unit_coefs ? 1.0 : des.coef[g];
            const double tr = T[s * D], ex = cf * T[s * D + p];
            a = (tr + ex) * half_dd; b = (tr - ex) * half_dd;
        }
    };
    // model probabilities of the current estimate (pe) and of the update direction (pu), per owned
    // setting and outcome: p(alpha) = pe + alpha * pu.  One-wave kernel: register arrays, so a line-search step
    // touches no memory.  LEAN: recomputed from the two tables at every use (same expressions, same bits).
    double pep[LEAN ? 1 : MJ], pem[LEAN ? 1 : MJ], pup[LEAN ? 1 : MJ], pum[LEAN ? 1 : MJ];
    if constexpr (!LEAN) {
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) { pep[j] = pem[j] = 1.0; pup[j] = pum[j] = 0.0; }
    }
    // (every slot is assigned -- `missing` for lanes without an outcome -- so that the arrays are dead
    // between two calls and do not occupy registers across the projection)
    auto load_probs = [&](const double* T, double* pp, double* pm, double missing) __attribute__((always_inline)) {
        if constexpr (!LEAN) {
#pragma unroll
            for (int j = 0; j < MAXJ; ++j) table_probs(T, j, missing, pp[j], pm[j]);
        }
    };
    auto pe_of = [&](int j, double& a, double& b) __attribute__((always_inline)) {
        if constexpr (LEAN) table_probs(L.Test, j, 1.0, a, b); else { a = pep[j]; b = pem[j]; }
    };
    auto pu_of = [&](int j, double& a, double& b) __attribute__((always_inline)) {
        if constexpr (LEAN) table_probs(L.Tupd, j, 0.0, a, b); else { a = pup[j]; b = pum[j]; }
    };
    // negative log-likelihood at est + alpha * update (tomography.py:597-614)
    auto cost_at = [&](double alpha) __attribute__((always_inline)) -> double {
        double acc = 0.0;
#pragma unroll SLOT_UNROLL
        for (int j = 0; j < (STREAM ? nslots : MAXJ); ++j) {
            const int g = lane + 64 * j;
            if (g < m) {
                double ep_, em_, up_, um_;
                pe_of(j, ep_, em_); pu_of(j, up_, um_);
                double pp = fma(alpha, up_, ep_), pm = fma(alpha, um_, em_);
                pp = pp < PGDB_EPS ? PGDB_EPS : pp;
                pm = pm < PGDB_EPS ? PGDB_EPS : pm;
                double np_, nm_;
                counts_of(j, np_, nm_);
                acc -= np_ * fast_log_pos(pp) + nm_ * fast_log_pos(pm);
            }
        }
        return uniform(wave_sum(acc));
    };

    // ---- initial estimate I_D / d (tomography.py:564) in block layout
    Blk est = blk_zero();
    if (lane < NACT) {
        const int I = lane / NB, J = lane % NB;
        if (I == J) { est.re[0] = 1.0 / d; est.re[3] = 1.0 / d; }
    }
    FBX_WAVE_SYNC();

    int iters = 0, dyk = 0, backtracks = 0, sweeps = 0;
    int ls_full = 0, ls_sums = 0;          // work accounting: full cost evaluations / power-sum reductions
    // eigenvector bases of the previous outer iteration's Dykstra run (fbx_choi.hpp BasisStore)
    BasisStore basis;
    basis.g = basis_scratch ? basis_scratch + (size_t)item * basis_cap * D * D : nullptr;
    basis.cap = basis_cap; basis.nprev = 0; basis.use_prev = false; basis.write_all = false;
    int chain_start = 0;                           // value of `sweeps` at the last cold start of the stored bases
    double outer_step = 1.0;                       // alpha * ||update||_F of the previous outer iteration
    PhaseClock pc; pc.reset(); L.choi.pc = &pc;
    PH_START(pc);
    double old_cost = 0.0, new_cost = 0.0;
    bool have_cost = false;
    // A reconstruction cut into PIECES of outer iterations (fbx_pgdb_lean.hip, pgdb_lean_pieces_kernel): everything the loop
    // carries from one outer iteration to the next -- the estimate, the counters, the state of the basis store (whose bases stay
    // where they are, in the item's slice of the workspace), the previous step and cost -- is in `rec` (PGDB_REC doubles) between
    // two pieces; the arithmetic of an iteration does not know where the previous one ran, so the result is bit-identical.
    bool paused = false;
    if (PIECES && resume) {
        if (rec[PGDB_REC_EST + 12] != 0.0) return;        // finished in an earlier piece (converge mode): results are out
#pragma unroll
        for (int e = 0; e < 4; ++e) { est.re[e] = rec[(2 * e) * 64 + lane]; est.im[e] = rec[(2 * e + 1) * 64 + lane]; }
        const double* sc = rec + PGDB_REC_EST;
        iters = (int)sc[0]; dyk = (int)sc[1]; backtracks = (int)sc[2]; sweeps = (int)sc[3]; ls_full = (int)sc[4]; ls_sums = (int)sc[5];
        basis.nprev = (int)sc[6]; chain_start = (int)sc[7]; L.choi.terms = (int)sc[8];
        outer_step = sc[9]; old_cost = sc[10]; new_cost = sc[11];
        have_cost = true;
    }

    while (true) {
        if (mode == FBX_MODE_FIXED && iters >= max_iters) break;
        if (PIECES && iters >= piece_stop) { paused = true; break; }
        const int dyk_before = dyk, bt_before = backtracks;       // per-iteration trace (fbx_pgdb_process_ex)
        // A stored basis is the product of all rotations applied to its chain since the last cold start, and every
        // rotation costs ~1e-16 of unitarity: the chains are dropped once they have absorbed FBX_BASIS_CHAIN_SWEEPS
        // sweeps per slot (216 sweeps x 120 rotations: < 3e-12 even if every rounding error had the same sign; the
        // norm test of the warm start discards anything beyond 1e-9 anyway.  Round 1 dropped them every 16
        // iterations whatever had happened, i.e. also in the stalled iterations, whose frozen decompositions apply
        // no rotation at all, and a cold restart costs ~10 sweeps more than a warm projection).  The first basis of this iteration's
        // projection is requested now, so that it arrives behind the gradient.
        if (iters == 0 ||
            sweeps - chain_start >= FBX_BASIS_CHAIN_SWEEPS * (basis.nprev > 0 ? basis.nprev : 1)) {
            basis.nprev = 0; chain_start = sweeps;
        }
        if (basis.g && basis.nprev > 0) basis.template prefetch<D * D>(0, lane);
        // ---- prediction table of the current estimate
        FBX_WAVE_SYNC();
        blk_store<D, LD>(L.choi.Mw, lane, est);
        FBX_WAVE_SYNC();
        choi_to_pauli_real<NQ>(L.choi.Mw, L.Rb, lane);
        FBX_WAVE_SYNC();
        stage_ct();                                 // (LEAN) stays in place for the gradient product below
        FBX_WAVE_SYNC();
        PH_STOP(pc, 3);
        predict_table<NQ>(L.Rb, Ct, L.Test, S, lane);
        load_slots();
        FBX_WAVE_SYNC();
        PH_STOP(pc, 7);
        load_probs(L.Test, pep, pem, 1.0);
        if (!have_cost) { old_cost = cost_at(0.0); have_cost = true; ++ls_full; }   // tomography.py:565

        // ---- gradient (tomography.py:617-633): eta = n / clip(p); per input state s the weights of the
        // Pauli components, Wt[s][0] = sum (eta+ + eta-)/2 and Wt[s][p] = coef (eta+ - eta-)/2, added
        // straight from the registers of the lanes that own the settings (LDS fp64 atomics; one
        // wavefront per item, so the order of the additions is the same in every run)
        double* Wt = L.Tupd;                        // [S][D]
        for (int idx = lane; idx < D * S; idx += 64) Wt[idx] = 0.0;
        FBX_WAVE_SYNC();
#pragma unroll SLOT_UNROLL
        for (int j = 0; j < (STREAM ? nslots : MAXJ); ++j) {
            const int g = lane + 64 * j;
            if (g < m) {
                double ep_, em_;
                pe_of(j, ep_, em_);
                const double pp = ep_ < PGDB_EPS ? PGDB_EPS : ep_;
                const double pm = em_ < PGDB_EPS ? PGDB_EPS : em_;
                double np_, nm_;
                counts_of(j, np_, nm_);
                const double ep = np_ / pp, em = nm_ / pm;
                const double cf = unit_coefs ? 1.0 : des.coef[g];
                const uint32_t dw = spw_of(j);
                const int st = dw >> 16, p = dw & 0xffff;
                atomicAdd(&Wt[st * D], 0.5 * (ep + em));
                atomicAdd(&Wt[st * D + p], cf * 0.5 * (ep - em));
            }
        }
        FBX_WAVE_SYNC();
        // R-coefficients of the gradient: Rg_ij = -(1/d^2) sum_s W[i][s] C[j][s].  Lane (i = lane % D,
        // jq = lane / D) owns the outputs j = JB jq .. JB jq + JB - 1: per state one conflict-free load of
        // Wt[s][i] and JB broadcast coefficients of the state's Bloch vector.
        {
            constexpr int JB = (D * D + 63) / 64;
            const int i = lane % D, j0 = (lane / D) * JB;
            if (j0 < D) {
                double acc[JB];
#pragma unroll
                for (int r = 0; r < JB; ++r) acc[r] = 0.0;
                for (int st = 0; st < S; ++st) {
                    const double w = Wt[st * D + i];
#pragma unroll
                    for (int r = 0; r < JB; ++r) acc[r] = fma(w, Ct[st * D + j0 + r], acc[r]);
                }
#pragma unroll
                for (int r = 0; r < JB; ++r) L.Rb[(j0 + r) * D + i] = -acc[r] / (double)(d * d);
            }
        }
        FBX_WAVE_SYNC();
        Blk x;
        {
            const Blk grad = pauli_real_to_choi_blk<NQ>(L.Rb, L.choi.Mw, lane);
            PH_STOP(pc, 4);
            // ---- projected step (tomography.py:572)
            x = blk_axpy(est, -inv_mu, grad);
            // the gradient is needed again after the projection (inner product with the update): parked in
            // Rb + Tupd (LEAN: Tupd), not in 16 registers across the Dykstra loop
            FBX_WAVE_SYNC();
            if (lane < NACT) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { L.park[(2 * e) * NACT + lane] = grad.re[e]; L.park[(2 * e + 1) * NACT + lane] = grad.im[e]; }
            }
            FBX_WAVE_SYNC();
        }
        // below a step of 1e-3 the previous run's trajectory is closer to this one than consecutive
        // Dykstra iterates are to each other (those stop at ~1e-2)
        basis.use_prev = outer_step < FBX_BASIS_STEP;
        // Inexact projections while the iteration is far from its fixed point: the eigensolver of the CP
        // projections stops at an off-diagonal norm of des.eig_rel_tol (default FBX_JTOL_REL; fbx_set_option) x the previous outer step (relative to
        // ||H||_F), never looser than that and never tighter than the 1e-13 it uses everywhere else.  What the
        // reconstruction V diag(M)+ V^H drops is of the size of that off-diagonal part, i.e. 1e-8 of the
        // distance the estimate still moves per iteration (-10 % time, parity survey unchanged).
        { const double tr_ = des.eig_rel_tol * outer_step; L.choi.jtol2 = fmax(FBX_JACOBI_TOL2, tr_ * tr_); }
        basis.write_all = outer_step < FBX_BASIS_WRITE_STEP;
        const Blk proj = proj_physical_blk<NQ>(x, TPC < 0 ? trace_preserving != 0 : TPC != 0, L.choi, lane, dyk, sweeps, 100000,
                                               &basis);
        const Blk upd = blk_sub(proj, est);
        Blk grad = blk_zero();
        if (lane < NACT) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { grad.re[e] = L.park[(2 * e) * NACT + lane]; grad.im[e] = L.park[(2 * e + 1) * NACT + lane]; }
        }
        double ipr, ipi;
        blk_dotc(upd, grad, ipr, ipi);
        ipr = uniform(wave_sum(ipr));
        PH_STOP(pc, 2);

        // ---- prediction tables for the line search
        FBX_WAVE_SYNC();
        blk_store<D, LD>(L.choi.Mw, lane, upd);
        FBX_WAVE_SYNC();
        choi_to_pauli_real<NQ>(L.choi.Mw, L.Rb, lane);
        FBX_WAVE_SYNC();
        stage_ct();
        FBX_WAVE_SYNC();
        predict_table<NQ>(L.Rb, Ct, L.Tupd, S, lane);
        load_slots();
        FBX_WAVE_SYNC();
        load_probs(L.Test, pep, pem, 1.0);      // again: not kept in registers across the projection
        load_probs(L.Tupd, pup, pum, 0.0);
        PH_STOP(pc, 3);
        // ---- backtracking line search (tomography.py:575-585)
        // Small steps: cost(alpha) = cost(0) - sum n log1p(alpha pu / pe), and cost(0) is old_cost.
        // Once alpha |pu / pe| < 2^-9 for every outcome (and nothing sits at the clip), log1p is a
        // degree-6 polynomial to < 1e-17 relative -- 8 instructions per outcome instead of ~40.  The
        // long halving runs of stalled iterations live here.
        auto ratio = [](double pu, double pe) __attribute__((always_inline)) -> double {
            double ip = __builtin_amdgcn_rcp(pe);
            ip = fma(fma(-pe, ip, 1.0), ip, ip);
            return (pe < 2.0 * PGDB_EPS || fabs(pu) > pe) ? 0.0 : pu * ip;
        };
        // outcomes evaluated exactly instead (compact list below): at the clip, or moving by more than
        // their own size over a full step -- with those out, |pu / pe| <= 1 and the polynomial takes over
        // from alpha = 2^-9 on whatever the design
        auto exact = [](double pu, double pe) __attribute__((always_inline)) -> bool {
            return pe < 2.0 * PGDB_EPS || fabs(pu) > pe;
        };
        double rmax = 0.0;                   // max |pu / pe| over the outcomes that are not listed
        uint32_t near_clip = 0u;             // wave-uniform: bit 2j / 2j+1 = some lane's +/- outcome of slot j is near the clip
#pragma unroll SLOT_UNROLL
        for (int j = 0; j < (STREAM ? nslots : MAXJ); ++j) {
            double ep_, em_, up_, um_;
            pe_of(j, ep_, em_); pu_of(j, up_, um_);
            rmax = fmax(rmax, fmax(fabs(ratio(up_, ep_)), fabs(ratio(um_, em_))));
            // (STREAM: more slots than the mask has bit pairs -- one flag for all, every slot is searched below)
            if (__ballot(exact(up_, ep_))) near_clip |= STREAM ? 1u : 1u << (2 * j);
            if (__ballot(exact(um_, em_))) near_clip |= STREAM ? 1u : 2u << (2 * j);
        }
        rmax = uniform(wave_max(rmax));
        const bool small_ok = rmax == rmax;
        // The near-clip outcomes are few and scattered over lanes and slots: they are compacted into
        // one entry per lane (through LDS -- the Jacobi work area is idle during the line search), so
        // an evaluation costs one clipped log per lane instead of one per flagged slot and sign.
        // the list lives in the Jacobi work area (Ms + Vs, idle during the line search): 4 rows of CL_MAX doubles
        constexpr int CL_MAX = (2 * sizeof(cplx) * D * D) / (4 * sizeof(double)) < 64 ? (int)((2 * sizeof(cplx) * D * D) / (4 * sizeof(double))) : 64;
        static_assert(4 * CL_MAX * sizeof(double) <= 2 * sizeof(cplx) * D * D, "compact list must fit into Ms + Vs");
        double clip_pe = 1.0, clip_pu = 0.0, clip_n = 0.0;       // this lane's entry of the compact list
        bool clip_listed = false;
        double clip_base = 0.0;
        int n_clip = 0;
        if (near_clip) {
            double* cl = (double*)L.choi.Ms;                     // [4][CL_MAX]: pe, pu, n, n log(clip(pe))
            const unsigned long long below = (1ull << lane) - 1ull;
            FBX_WAVE_SYNC();
#pragma unroll SLOT_UNROLL
            for (int j = 0; j < (STREAM ? nslots : MAXJ); ++j) {
#pragma unroll
                for (int sg = 0; sg < 2; ++sg) {
                    if (STREAM || (near_clip & ((1u + sg) << (2 * j)))) {
                        double np_, nm_, ep_, em_, up_, um_;
                        counts_of(j, np_, nm_);
                        pe_of(j, ep_, em_); pu_of(j, up_, um_);
                        const double pe = sg ? em_ : ep_, pu = sg ? um_ : up_, nn = sg ? nm_ : np_;
                        const bool f = exact(pu, pe);
                        const unsigned long long mk = __ballot(f);
                        const int pos = n_clip + __popcll(mk & below);
                        if (f && pos < CL_MAX) { cl[pos] = pe; cl[CL_MAX + pos] = pu; cl[2 * CL_MAX + pos] = nn; }
                        n_clip += __popcll(mk);
                    }
                }
            }
            FBX_WAVE_SYNC();
            clip_listed = n_clip <= CL_MAX;
            if (clip_listed) {
                if (lane < n_clip) { clip_pe = cl[lane]; clip_pu = cl[CL_MAX + lane]; clip_n = cl[2 * CL_MAX + lane]; }
                clip_base = clip_n * fast_log_pos(clip_pe < PGDB_EPS ? PGDB_EPS : clip_pe);
                if (lane < n_clip) cl[3 * CL_MAX + lane] = clip_base;       // for the one-pass ladder below
            }                                                    // more than CL_MAX of them: full evaluations only
            FBX_WAVE_SYNC();
        }
        auto clipped_log = [](double p) __attribute__((always_inline)) -> double { return fast_log_pos(p < PGDB_EPS ? PGDB_EPS : p); };
        // sum_o n_o log1p(alpha r_o) = sum_k c_k alpha^k S_k with the power sums S_k = sum_o n_o r_o^k,
        // c_k = (-1)^(k+1) / k, reduced ONCE per outer iteration (on the first small step): every further
        // halving is a Horner evaluation (+ the clipped logs of the listed outcomes) instead of a pass
        // over all outcomes.  Degree 16 below alpha rmax = 2^-3: remainder < 2^-51 / 17 per unit of n,
        // below the rounding of the exact evaluation.
        constexpr int NS = 16;
        double Sk[NS];
        bool have_sums = false;
        auto small_regime = [&](double alpha) __attribute__((always_inline)) -> bool {
            // (des.ls_reference: FBX_MODE_LS_REFERENCE of the call -- every halving a full cost sum, the rounded test below)
            return !des.ls_reference && small_ok && (near_clip == 0u || clip_listed) && alpha * rmax < FBX_SMALL_STEP_LIMIT;
        };
        auto series = [&](double alpha) __attribute__((always_inline)) -> double {      // alpha may differ per lane
            double q = Sk[NS - 1];
#pragma unroll
            for (int k = NS - 2; k >= 0; --k) q = fma(alpha, q, Sk[k]);
            return alpha * q;
        };
        // The acceptance test of tomography.py:578 is `new_cost > old_cost + change`.  In the small-step regime the
        // cost DIFFERENCE is known exactly (the series), and the test is made on it: `new - old > change`.  The
        // two forms differ only where |new - old| and |change| are below the rounding of the cost itself -- the
        // stalled iterations past convergence, where the projection's inexactness makes the direction an
        // ASCENT direction (new - old = alpha <update, gradient> > change > 0 for every alpha): there the
        // reference's rounded test is decided by the noise of its cost sums (it ends up halving 47-50 times
        // per iteration), the rounded test on an exact difference would accept as soon as both
        // sides vanish against the cost (~11 halvings, a 3e-8 step along an ascent direction, every stalled
        // iteration), and the exact test rejects down to alpha < 1e-15 like the reference's late iterations:
        // the estimate then stays where the reference's stays, to rounding.  -DFBX_LS_ROUNDED restores the
        // rounded form (round-1 behaviour).
        bool ls_exact = false;
        double ls_diff = 0.0;
        auto cost_step = [&](double alpha) __attribute__((always_inline)) -> double {
            ls_exact = false;
            if (!small_regime(alpha)) { ++ls_full; return cost_at(alpha); }
            if (!have_sums) {
#pragma unroll
                for (int k = 0; k < NS; ++k) Sk[k] = 0.0;
#pragma unroll SLOT_UNROLL
                for (int j = 0; j < (STREAM ? nslots : MAXJ); ++j) {
                    double np_, nm_, ep_, em_, up_, um_;
                    counts_of(j, np_, nm_);
                    pe_of(j, ep_, em_); pu_of(j, up_, um_);
#pragma unroll
                    for (int sg = 0; sg < 2; ++sg) {
                        const double x = sg ? ratio(um_, em_) : ratio(up_, ep_);   // recomputed: not kept live
                        double t = (sg ? nm_ : np_) * x;
#pragma unroll
                        for (int k = 0; k < NS; ++k) { Sk[k] += t; t *= x; }
                    }
                }
#pragma unroll
                for (int k = 0; k < NS; ++k) Sk[k] = uniform(wave_sum(Sk[k])) * ((k & 1) ? -1.0 : 1.0) / (double)(k + 1);
                have_sums = true; ++ls_sums;
            }
            double acc = series(alpha);
            if (near_clip)                   // the listed outcomes: exact difference of clipped logs
                acc += uniform(wave_sum(clip_n * clipped_log(fma(alpha, clip_pu, clip_pe)) - clip_base));
            ls_exact = true; ls_diff = -acc;
            return old_cost - acc;
        };
        auto rejected = [&](double change_) __attribute__((always_inline)) -> bool {
            return ls_exact ? (ls_diff > change_) : (new_cost > old_cost + change_);
        };
        double alpha = 1.0;
        new_cost = cost_step(alpha);
        double change = PGDB_GAMMA * alpha * ipr;
        int small_fails = 0;
        while (rejected(change)) {
            // Two series evaluations in a row have failed: this is one of the long halving runs of a
            // stalled iteration.  The rest of the ladder alpha 2^-L, L = 1, 2, ... is evaluated in ONE pass,
            // lane L taking its own alpha (same series; the listed outcomes, one per lane so far, are
            // walked from their LDS list by every lane), and the first L that the sequential loop would
            // have stopped at -- sufficient decrease, or alpha below the floor -- is taken.
            if (small_fails >= 2) {
                const double a_l = __builtin_ldexp(alpha, -lane), c_l = __builtin_ldexp(change, -lane);
                double acc = series(a_l);
                if (near_clip) {
                    const double* cl = (const double*)L.choi.Ms;
                    for (int e = 0; e < n_clip; ++e) {
                        const double pe = cl[e], pu = cl[CL_MAX + e], nn = cl[2 * CL_MAX + e];
                        acc += nn * clipped_log(fma(a_l, pu, pe)) - cl[3 * CL_MAX + e];
                    }
                }
                const double val = old_cost - acc;
                const bool rej_l = -acc > c_l;
                const unsigned long long stop = __ballot(lane >= 1 && (a_l < PGDB_ALPHA_MIN || !rej_l));
                const int Ls = __builtin_ctzll(stop);          // alpha <= 1: lane 50 is below the floor at the latest
                alpha = __builtin_ldexp(alpha, -Ls); change = __builtin_ldexp(change, -Ls);
                new_cost = uniform(__shfl(val, Ls));
                backtracks += Ls;
                break;
            }
            alpha *= 0.5;
            change *= 0.5;
            if (small_regime(alpha)) ++small_fails;
            new_cost = cost_step(alpha);
            ++backtracks;
            if (alpha < PGDB_ALPHA_MIN) break;
        }
        PH_STOP(pc, 5);
        est = blk_axpy(est, alpha, upd);            // tomography.py:588
        outer_step = alpha * sqrt(uniform(wave_sum(blk_norm2(upd))));
        if (trace_out && iters < trace_iters && lane == 0) {     // Dykstra iterations and halvings of THIS outer iteration
            int* tr = trace_out + ((size_t)item * trace_iters + iters) * 2;
            tr[0] = dyk - dyk_before; tr[1] = backtracks - bt_before;
        }
        ++iters;
        if (mode == FBX_MODE_CONVERGE) {
            if (!(old_cost - new_cost >= PGDB_STOP)) break;      // tomography.py:589; a NaN cost also ends the loop
            if (max_iters > 0 && iters >= max_iters) break;
        }
        old_cost = new_cost;
    }

    if (PIECES && paused) {                             // the next piece of this item continues from here
#pragma unroll
        for (int e = 0; e < 4; ++e) { rec[(2 * e) * 64 + lane] = est.re[e]; rec[(2 * e + 1) * 64 + lane] = est.im[e]; }
        if (lane == 0) {
            double* sc = rec + PGDB_REC_EST;
            sc[0] = iters; sc[1] = dyk; sc[2] = backtracks; sc[3] = sweeps; sc[4] = ls_full; sc[5] = ls_sums;
            sc[6] = basis.nprev; sc[7] = chain_start; sc[8] = L.choi.terms;
            sc[9] = outer_step; sc[10] = old_cost; sc[11] = new_cost; sc[12] = 0.0;
        }
        return;
    }
    if (PIECES && lane == 0) rec[PGDB_REC_EST + 12] = 1.0;
    // ---- write back
    if (lane < NACT) {
        const int I = lane / NB, J = lane % NB;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = 2 * I + (e >> 1), col = 2 * J + (e & 1);
            double* o = choi_out + ((item * D + row) * D + col) * 2;
            o[0] = est.re[e]; o[1] = est.im[e];
        }
    }
    if (lane == 0) {
        if (iters_out) iters_out[item] = iters;
        if (dykstra_out) dykstra_out[item] = dyk;
        if (backtracks_out) backtracks_out[item] = backtracks;
        if (cost_out) cost_out[item] = have_cost ? new_cost : 0.0;
        if (work_out) {     // Jacobi sweeps, eigenvalue terms rebuilt, full cost evaluations, power-sum reductions
            work_out[4 * item] = sweeps; work_out[4 * item + 1] = L.choi.terms;
            work_out[4 * item + 2] = ls_full; work_out[4 * item + 3] = ls_sums;
        }
    }
#ifdef FBX_PHASE_TIMERS
    if (lane == 0 && phase_out) for (int i = 0; i < FBX_NPHASE; ++i) phase_out[item * FBX_NPHASE + i] = pc.acc[i];
#endif
}

// The ticket loop of the pieces kernels (pgdb_lean_pieces_kernel, fbx_pgdb_lean.hip: the protocol is described there;
// pgdb_pieces_kernel, fbx_pgdb.hip): persistent workgroups draw (item, piece) tickets from `queue` until none is left.
template <int NQ, int MAXJ, bool LEAN, int TPC = -1>
__device__ __forceinline__ void
pgdb_pieces_run(char* smem, const DesignDev& des, long long B, const double* __restrict__ expect, const double* __restrict__ counts,
                int trace_preserving, int mode, int max_iters, double* __restrict__ choi_out, int* __restrict__ iters_out,
                int* __restrict__ dykstra_out, int* __restrict__ backtracks_out, double* __restrict__ cost_out, int* __restrict__ work_out,
                long long* __restrict__ phase_out, cplx* __restrict__ basis_scratch, int basis_cap, double* __restrict__ ncounts,
                int* __restrict__ trace_out, int trace_iters, int pieces, int piece_iters, int* __restrict__ queue,
                int* __restrict__ flags, double* __restrict__ recs) {
    const long long total = (long long)pieces * B;
    for (;;) {
        long long e = 0;
        if ((threadIdx.x & 63) == 0) e = (long long)__hip_atomic_fetch_add((unsigned*)queue, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        e = __builtin_amdgcn_readfirstlane((int)e);
        if (e >= total) break;
        const int piece = (int)(e / B);
        const long long item = e % B;
        if (piece > 0) {
            // the previous piece of this item has been published (bounded: a lost flag must not hang the device)
            // The bound follows the length of a piece: ~8000 polls (>= 1 us each) per outer iteration of the predecessor + a
            // fixed 2^21 (~2-4 s), i.e. > 10 ms per iteration where one costs 0.15 ms (0.5 ms on a device shared by several
            // launches); the host keeps piece_iters <= 512 (launch_pgdb).  Beyond it the flag is lost -- a protocol error, not
            // a slow neighbour -- and the launch is aborted rather than left spinning.
            long long spins = 0;
            const long long spin_limit = (1ll << 21) + ((long long)piece_iters << 13);
            while (__hip_atomic_load(&flags[item], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < piece) {
                __builtin_amdgcn_s_sleep(32);
                if (++spins > spin_limit) __builtin_trap();
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        const int stop = piece + 1 < pieces ? (piece + 1) * piece_iters : 0x7fffffff;
        pgdb_body<NQ, MAXJ, LEAN, true, TPC>(smem, item, des, B, expect, counts, trace_preserving, mode, max_iters, choi_out, iters_out,
                                  dykstra_out, backtracks_out, cost_out, work_out, phase_out, basis_scratch, basis_cap,
                                  LEAN ? ncounts + (size_t)blockIdx.x * 2 * MAXJ * 64 : nullptr, trace_out, trace_iters,
                                  recs + (size_t)item * PGDB_REC, stop, piece > 0);
        if (piece + 1 < pieces) {
            // publish: the record and the bases written by this piece, then the flag
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if ((threadIdx.x & 63) == 0) __hip_atomic_store(&flags[item], piece + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __builtin_amdgcn_s_barrier();                      // (one wavefront: orders this piece's LDS accesses before the next one's)
    }
}

// what a launcher hands to either kernel
struct PgdbLaunch {
    DesignDev dev; long long nb; const double* e; const double* c; int tp, mode, max_iters; double* choi;
    int *it, *dy, *bt; double* cost; int* sw; long long* phase; cplx* basis; int basis_cap; double* ncounts; int* trace; int trace_iters;
    // pieces (the two-waves kernel only; pieces <= 1: one reconstruction per workgroup from start to end): `piece_iters` outer
    // iterations per piece, `queue` the ticket counter, `flags` one progress flag per item, `recs` [nb][PGDB_REC]
    int pieces = 0, piece_iters = 0; int* queue = nullptr; int* flags = nullptr; double* recs = nullptr;
};
// fbx_pgdb_lean.hip: the two-wavefronts-per-SIMD kernel for 2 qubits, MAXJ in {4, 9, 16}
size_t pgdb_lean_lds(int maxj, int S);
bool pgdb_lean_eligible(int S);      // the design's Bloch table fits beside Rb in the Jacobi work area
int pgdb_lean_launch(int maxj, size_t lds, hipStream_t st, const PgdbLaunch& a);
// the streamed instantiations (any number of settings; 1 or 2 qubits): LDS bytes and launch
size_t pgdb_stream_lds(int nq, int S);
int pgdb_stream_launch(int nq, size_t lds, hipStream_t st, const PgdbLaunch& a);

}  // namespace fbx
