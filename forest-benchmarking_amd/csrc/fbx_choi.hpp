// fbx_choi.hpp -- device routines on D x D (D = 4^n) Choi matrices held as one 2x2 block per
// lane (fbx_eigh.hpp): CP / TP / TNI projections, Dykstra's alternating projection and the
// Choi <-> Pauli-coefficient transforms used by the PGDB gradient.
//
// Reference functions restated on the device (file:line under forest/benchmarking/):
//   proj_choi_to_completely_positive   operator_tools/project_superoperators.py:19-34
//   proj_choi_to_trace_non_increasing  operator_tools/project_superoperators.py:37-59
//   proj_choi_to_trace_preserving      operator_tools/project_superoperators.py:62-84
//   proj_choi_to_physical (Dykstra)    operator_tools/project_superoperators.py:87-144
//   partial_trace                      operator_tools/calculational.py:5-35
#pragma once
#include "fbx_eigh.hpp"


namespace fbx {

// LDS work area shared by the routines below (carved by the kernel).  LEAN (the 2-waves-per-SIMD PGDB
// variant, 16.5 KB per reconstruction instead of 39 KB): no separate staging matrix -- the Pauli transforms
// stage through Ms and spill into Vs (both dead outside a projection), the partial trace inside a projection
// through Ms alone (dead once the eigenvalues are read) with an unpadded leading dimension.
// TWO_WORKERS: the CP projections of this work area use the two-workers-per-upper-block form of the 16 x 16 eigensolver
// (fbx_eigh.hpp, JacobiWave<16, true>) -- set by the one-wavefront-per-SIMD PGDB body, whose lone wavefront gains 2-5 % from it; every
// other user keeps the full-block form (the register-capped two-waves kernel is 5 % slower with it).
template <int NQ, bool LEAN = false, bool TWO_WORKERS = false>
struct ChoiLds {
    static constexpr int d = 1 << NQ, D = d * d, LD = D + 1, LDs = d + 1;
    static constexpr int LDpt = LEAN ? D : LD;     // leading dimension of the partial-trace staging
    static constexpr bool lean = LEAN;
    static constexpr bool two_workers = TWO_WORKERS && D == 16;
    cplx* Mw;      // [D * LD]   row-major staging matrix (Pauli transforms); LEAN: = Ms, running into Vs
    cplx* Mpt;     // partial-trace staging: Mw, or Ms (LEAN)
    cplx* Ts;      // scratch of the generic warm-start basis change (D != 16): Mw, or (LEAN, where Mw IS Ms) a block of its own
    cplx* Ms;      // [sys_elems<D>()]  Jacobi work matrix, element-major block layout (fbx_eigh.hpp)
    cplx* Vs;      // [sys_elems<D>()]  eigenvectors, same layout
    double* lam;   // [D]
    JRec* rec;     // [D / 2 + 1] rotation records (scratch)
    cplx* pt;      // [d * LDs]  partial trace (d x d), row-major
    cplx* pts;     // [d * d]    partial trace in the Jacobi layout (TNI only)
    cplx* ptV;     // [d * d]    its eigenvectors (TNI only)
    cplx* ptold;   // [d * LDs]  the previous Dykstra iteration's TP / TNI correction (old_TP_change as a d x d matrix)
    PhaseClock* pc = nullptr;   // diagnostics (FBX_PHASE_TIMERS builds)
    int terms = 0;              // work accounting: eigenvalue terms rebuilt by the CP projections (wave-uniform)
    double jtol2 = FBX_JACOBI_TOL2;   // off-norm^2 / norm^2 at which the CP projections' eigensolver stops
    static constexpr size_t bytes() {
        static_assert(!LEAN || 2 * sys_elems<D>() >= D * LD, "the transforms' staging matrix must fit into Ms + Vs");
        return sizeof(cplx) * ((LEAN ? 0 : D * LD) + (LEAN && D != 16 ? sys_elems<D>() : 0) + 2 * sys_elems<D>() + d * LDs + 2 * d * d + d * LDs) + sizeof(double) * D + sizeof(JRec) * (D / 2 + 1);
    }
    __device__ void carve(char*& p) {
        if constexpr (!LEAN) { Mw = (cplx*)p; p += sizeof(cplx) * D * LD; }
        Ms = (cplx*)p; p += sizeof(cplx) * sys_elems<D>();
        Vs = (cplx*)p; p += sizeof(cplx) * sys_elems<D>();
        if constexpr (LEAN) { Mw = Ms; Mpt = Ms; } else { Mpt = Mw; }
        if constexpr (LEAN && D != 16) { Ts = (cplx*)p; p += sizeof(cplx) * sys_elems<D>(); } else { Ts = Mw; }
        pt = (cplx*)p; p += sizeof(cplx) * d * LDs;
        pts = (cplx*)p; p += sizeof(cplx) * d * d;
        ptV = (cplx*)p; p += sizeof(cplx) * d * d;
        ptold = (cplx*)p; p += sizeof(cplx) * d * LDs;
        rec = (JRec*)p; p += sizeof(JRec) * (D / 2 + 1);
        lam = (double*)p; p += sizeof(double) * D;
    }
};

// ---- CP projection: Hermitise, eigh, clamp negative eigenvalues, rebuild -----------------
// project_superoperators.py:19-34.  `x` need not be Hermitian.  `sweeps` accumulates Jacobi
// sweeps (diagnostics).
template <int NQ, class LdsT>
__device__ Blk proj_cp_blk(const Blk& x, LdsT& L, int lane, int& sweeps, bool warm = false,
                       bool check_basis = false) {
    constexpr int D = LdsT::D;
    FBX_WAVE_SYNC();                       // previous readers of Ms / Vs are done
    sys_store<D>(L.Ms, lane, x);
    FBX_WAVE_SYNC();
    const Blk xa = sys_load_adjoint<D>(L.Ms, lane);
    Blk h;
#pragma unroll
    for (int e = 0; e < 4; ++e) { h.re[e] = 0.5 * (x.re[e] + xa.re[e]); h.im[e] = 0.5 * (x.im[e] + xa.im[e]); }
    FBX_WAVE_SYNC();
    sys_store<D>(L.Ms, lane, h);
    FBX_WAVE_SYNC();
    PH_STOP(*L.pc, 2);
    // warm start: the eigenvectors of the previous projection (still in Vs) nearly diagonalise
    // this matrix, because consecutive Dykstra iterates are close
    // A basis that came from the HBM store is only trusted after the change of basis: a unitary
    // similarity preserves the Frobenius norm, anything else (a damaged or torn slot) does not -- then
    // the matrix is restored and the decomposition starts from the identity.  For the single-wavefront
    // 16 x 16 solver the test rides on its first off-norm reduction (one extra wave reduction here, which
    // overlaps the matrix-core products).
    double hn2 = -1.0;
    if (warm && check_basis) hn2 = uniform(wave_sum(blk_norm2(h)));
    if (warm) {
        if constexpr (D == 16) jacobi_rotate_into_basis_mfma16(L.Ms, L.Vs, lane);
        else
        {
            jacobi_rotate_into_basis<D>(L.Ms, L.Vs, L.Ts, lane);
        }
    }
    int sw;
    if constexpr (D == 16) {
        sw = jacobi_eigh_wave<D, LdsT::two_workers>(L.Ms, L.Vs, lane, !warm, hn2, L.jtol2);
    } else
    {
        if (warm && check_basis) {
            constexpr int LS = (D / 2) * (D / 2), PS = sys_plane<D>();
            double mn2 = 0.0;
            if (lane < LS) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const cplx v = L.Ms[e * PS + lane]; mn2 = fma(v.re, v.re, fma(v.im, v.im, mn2)); }
            }
            mn2 = uniform(wave_sum(mn2));
            sw = (fabs(mn2 - hn2) <= FBX_BASIS_NORM_TOL * hn2) ? jacobi_eigh_lds<D>(L.Ms, L.Vs, L.rec, lane, false) : -1;
        } else sw = jacobi_eigh_lds<D>(L.Ms, L.Vs, L.rec, lane, !warm);
    }
    if (sw < 0) {                              // rejected basis: cold start on the restored matrix
        FBX_WAVE_SYNC();
        sys_store<D>(L.Ms, lane, h);
        FBX_WAVE_SYNC();
        sw = jacobi_eigh_lds<D>(L.Ms, L.Vs, L.rec, lane, true);
#ifdef FBX_DEBUG_REJECT
        sweeps += 1000000;
#endif
    }
    sweeps += sw;
    PH_STOP(*L.pc, 0);
    {
        const double l = lane < D ? L.Ms[sys_index<D>(lane, lane)].re : 0.0;
        if (lane < D) L.lam[lane] = l < 0.0 ? 0.0 : l;
        L.terms += __popcll(__ballot(l > 0.0));
    }
    FBX_WAVE_SYNC();
    const Blk out = reconstruct_blk<D>(L.Vs, L.lam, lane);
    PH_STOP(*L.pc, 1);
    return out;
}

// ---- partial trace over the output space into L.pt (d x d): calculational.py:5-35 with
// keep=[0], dims=[d, d].  Stages `x` through Mw.
template <int NQ, class LdsT>
__device__ void partial_trace_out(const Blk& x, LdsT& L, int lane) {
    constexpr int d = LdsT::d, D = LdsT::D, LD = LdsT::LDpt, LDs = LdsT::LDs;
    FBX_WAVE_SYNC();
    blk_store<D, LD>(L.Mpt, lane, x);
    FBX_WAVE_SYNC();
    if (lane < d * d) {
        const int i = lane / d, ip = lane % d;
        cplx s; s.re = 0.0; s.im = 0.0;
#pragma unroll
        for (int o = 0; o < d; ++o) {
            const cplx v = L.Mpt[(i * d + o) * LD + ip * d + o];
            s.re += v.re; s.im += v.im;
        }
        L.pt[i * LDs + ip] = s;
    }
    FBX_WAVE_SYNC();
}

// subtract kron(corr / d, I_d) where corr (d x d) is in L.pt
template <int NQ, class LdsT>
__device__ __forceinline__ Blk subtract_kron_pt(const Blk& x, const LdsT& L, int lane) {
    constexpr int d = LdsT::d, D = LdsT::D, LDs = LdsT::LDs, NB = D / 2;
    Blk r = x;
    if (lane < NB * NB) {
        const int I = lane / NB, J = lane % NB;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = 2 * I + (e >> 1), col = 2 * J + (e & 1);
            if ((row % d) == (col % d)) {
                const cplx c = L.pt[(row / d) * LDs + (col / d)];
                r.re[e] -= c.re / d; r.im[e] -= c.im / d;
            }
        }
    }
    return r;
}

// ---- TP projection: project_superoperators.py:62-84
template <int NQ, class LdsT>
__device__ Blk proj_tp_blk(const Blk& x, LdsT& L, int lane) {
    constexpr int d = LdsT::d, LDs = LdsT::LDs;
    partial_trace_out<NQ>(x, L, lane);
    if (lane < d) L.pt[lane * LDs + lane].re -= 1.0;       // pt - I
    FBX_WAVE_SYNC();
    return subtract_kron_pt<NQ>(x, L, lane);
}

// ---- TNI projection: project_superoperators.py:37-59 (d x d eigh of the partial trace,
// eigenvalues above 1 clamped to 1)
template <int NQ, class LdsT>
__device__ Blk proj_tni_blk(const Blk& x, LdsT& L, int lane, int& sweeps) {
    constexpr int d = LdsT::d, LDs = LdsT::LDs;
    partial_trace_out<NQ>(x, L, lane);
    // keep pt in registers, Hermitise a copy for the eigensolver
    const Blk ptb = blk_load<d, LDs>(L.pt, lane);
    const Blk pta = blk_load_adjoint<d, LDs>(L.pt, lane);
    Blk h;
#pragma unroll
    for (int e = 0; e < 4; ++e) { h.re[e] = 0.5 * (ptb.re[e] + pta.re[e]); h.im[e] = 0.5 * (ptb.im[e] + pta.im[e]); }
    FBX_WAVE_SYNC();
    sys_store<d>(L.pts, lane, h);
    FBX_WAVE_SYNC();
    sweeps += jacobi_eigh_lds<d>(L.pts, L.ptV, L.rec, lane);
    if (lane < d) {
        const double l = L.pts[sys_index<d>(lane, lane)].re;
        L.lam[lane] = l > 1.0 ? 1.0 : l;
    }
    FBX_WAVE_SYNC();
    const Blk proj = reconstruct_blk<d>(L.ptV, L.lam, lane);
    FBX_WAVE_SYNC();
    blk_store<d, LDs>(L.pt, lane, blk_sub(ptb, proj));      // pt - projection
    FBX_WAVE_SYNC();
    return subtract_kron_pt<NQ>(x, L, lane);
}

// ---- Dykstra: project_superoperators.py:87-144.  Stops on the Birgin-Raydan functional
// < 1e-4 (no iteration cap in the reference; `max_iter` is a safety net that is never the
// binding constraint in practice).  Returns the last TP / TNI iterate.
// `store` (optional) keeps, in HBM / L2, the eigenvector basis of EVERY Dykstra iteration of the
// previous call for the same item: successive PGDB iterations project nearby matrices along nearly
// the same Dykstra trajectory, so iteration j of this call starts its eigendecomposition from basis
// j of the previous call when the caller says the outer step was small (`use_prev`; the first
// projection, which has no basis of its own run to start from, always does).  A basis is only an
// initial guess -- every decomposition still runs to the same off-norm tolerance.
typedef double fbx_v2d __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) fbx_v2d* fbx_global_cplx_ptr;   // HBM: global_load / global_store, not flat_*
struct BasisStore {
    cplx* g;          // [cap][D * D] in the Jacobi layout, private to the item
    int cap;          // slots
    int nprev;        // slots that hold a basis of the previous call
    bool use_prev;    // outer step small: prefer the previous call's basis over the previous iteration's
    bool write_all;   // the next call is expected to want every slot (outer step within a small factor of the
                      // use_prev threshold): otherwise only slot 0 -- which every call starts from -- is written back
    // one basis in flight from HBM to registers (single-wavefront kernels, D <= 16: 16 B per lane and u)
    cplx pf[4];
    int pf_slot = -1; // slot the registers hold / are waiting for, -1: none
    template <int DD>
    __device__ __forceinline__ void prefetch(int slot, int lane) {
        const fbx_global_cplx_ptr src = (fbx_global_cplx_ptr)(g + (size_t)slot * DD);
#pragma unroll
        for (int u = 0; u < (DD + 63) / 64; ++u) {
            const int idx = lane + 64 * u;
            if (idx < DD) { const fbx_v2d w = src[idx]; pf[u].re = w.x; pf[u].im = w.y; }
        }
        pf_slot = slot;
    }
};

// block of -kron(C / d, I_d) for the d x d matrix C staged in LDS: what a TP / TNI projection adds to its argument
template <class LdsT>
__device__ __forceinline__ Blk tp_change_blk(const cplx* C, int lane) {
    constexpr int d = LdsT::d, D = LdsT::D, LDs = LdsT::LDs, NB = D / 2;
    Blk r = blk_zero();
    if (lane < NB * NB) {
        const int I = lane / NB, J = lane % NB;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = 2 * I + (e >> 1), col = 2 * J + (e & 1);
            if ((row % d) == (col % d)) { const cplx c = C[(row / d) * LDs + (col / d)]; r.re[e] = -(c.re / d); r.im[e] = -(c.im / d); }
        }
    }
    return r;
}

// Dykstra carried with TWO matrices per lane (the two-waves-per-SIMD kernels, 256 registers): pre_CP and old_CP_change
// stay in registers across the eigensolver; old_TP_change = -kron(corr / d, I_d) is kept as the d x d correction the TP /
// TNI projection subtracted (L.pt -> L.ptold), last_CP_projection only enters as the scalar <old_CP_change,
// last_CP_projection>, last_state = pre_CP + old_CP_change (see fbx_pgdb3.hip, where the same form removed the spills of
// the 3-qubit kernel's projection).  Same projections, same stopping rule; the stopping functional differs from the
// four-matrix form in rounding-level terms only (threshold 1e-4).
template <int NQ, class LdsT>
__device__ Blk proj_physical_blk_compact(const Blk& x, bool trace_preserving, LdsT& L, int lane,
                                         int& iters, int& sweeps, int max_iter, BasisStore* store) {
    constexpr int DD = LdsT::D * LdsT::D, d = LdsT::d, LDs = LdsT::LDs;
    Blk u = x, p = blk_zero(), new_state = x;
    double c0r = 0.0, c0i = 0.0;
    const bool have_store = store != nullptr && store->g != nullptr;
    constexpr int PF = (DD + 63) / 64;
    static_assert(PF <= 4, "BasisStore::pf holds one basis of at most 256 entries");
    FBX_WAVE_SYNC();
    if (lane < d * LDs) { cplx z; z.re = 0.0; z.im = 0.0; L.ptold[lane] = z; }
    int it = 0;
    for (; it < max_iter; ++it) {
        ++iters;
        bool warm = it > 0;
        bool from_slot = false;
        const int sweeps_before = sweeps;
        if (have_store && it < store->nprev && (it == 0 || store->use_prev)) {
            from_slot = true;
            FBX_WAVE_SYNC();
            if (store->pf_slot != it) store->template prefetch<DD>(it, lane);
#pragma unroll
            for (int q = 0; q < PF; ++q) {
                const int idx = lane + 64 * q;
                if (idx < DD) L.Vs[sys_linear<LdsT::D>(idx)] = store->pf[q];
            }
            store->pf_slot = -1;
            warm = true;
        }
        if (have_store && store->use_prev && it + 1 < store->nprev && it + 1 < store->cap)
            store->template prefetch<DD>(it + 1, lane);
        const Blk cp = proj_cp_blk<NQ>(u, L, lane, sweeps, warm, from_slot);
        if (have_store && it < store->cap && (it == 0 || store->write_all)) {
            if (!(from_slot && sweeps == sweeps_before)) {
                fbx_global_cplx_ptr dst = (fbx_global_cplx_ptr)(store->g + (size_t)it * DD);
#pragma unroll
                for (int q = 0; q < PF; ++q) {
                    const int idx = lane + 64 * q;
                    if (idx < DD) { const cplx w = L.Vs[sys_linear<LdsT::D>(idx)]; dst[idx] = fbx_v2d{w.re, w.im}; }
                }
            }
        }
        const Blk new_cp = blk_sub(cp, u);
        double s1 = blk_norm2(blk_sub(new_cp, p));
        double pcr, pci, ncr, nci;
        blk_dotc(p, cp, pcr, pci);
        blk_dotc(new_cp, cp, ncr, nci);
        const Blk last_state = blk_axpy(u, 1.0, p);
        const Blk old_tp = tp_change_blk<LdsT>(L.ptold, lane);
        const Blk pre_tp = blk_sub(cp, old_tp);
        new_state = trace_preserving ? proj_tp_blk<NQ>(pre_tp, L, lane) : proj_tni_blk<NQ>(pre_tp, L, lane, sweeps);
        const Blk new_tp = tp_change_blk<LdsT>(L.pt, lane);
        double s2 = blk_norm2(blk_sub(new_tp, old_tp));
        double i1r, i1i;
        blk_dotc(old_tp, blk_sub(new_state, last_state), i1r, i1i);
        s1 = wave_sum(s1); s2 = wave_sum(s2); i1r = wave_sum(i1r); i1i = wave_sum(i1i);
        pcr = wave_sum(pcr); pci = wave_sum(pci); ncr = wave_sum(ncr); nci = wave_sum(nci);
        const double i2r = pcr - c0r, i2i = pci - c0i;
        const double a1 = i1r * i1r + i1i * i1i, a2 = i2r * i2r + i2i * i2i;
        const double m1 = a1 > 1e-300 ? a1 * fast_rsqrt(a1) : 0.0, m2 = a2 > 1e-300 ? a2 * fast_rsqrt(a2) : 0.0;
        const double crit = uniform(s1 + s2 + 2.0 * m1 + 2.0 * m2);
        if (!(crit >= 1e-4)) { ++it; break; }
        c0r = ncr; c0i = nci;
        p = new_cp;
        u = blk_sub(new_state, new_cp);
        FBX_WAVE_SYNC();
        if (lane < d * LDs) L.ptold[lane] = L.pt[lane];
        FBX_WAVE_SYNC();
    }
    if (have_store) {
        const int written = store->write_all ? it : 1;
        store->nprev = written < store->cap ? written : store->cap;
        store->pf_slot = -1;
    }
    return new_state;
}

template <int NQ, class LdsT>
__device__ Blk proj_physical_blk(const Blk& x, bool trace_preserving, LdsT& L, int lane,
                                 int& iters, int& sweeps, int max_iter = 100000,
                                 BasisStore* store = nullptr) {
    if constexpr (LdsT::lean) return proj_physical_blk_compact<NQ>(x, trace_preserving, L, lane, iters, sweeps, max_iter, store);
    constexpr int DD = LdsT::D * LdsT::D;
    Blk old_cp = blk_zero(), old_tp = blk_zero(), last_cp = blk_zero();
    Blk last_state = x, new_state = x;
    // (the caller passes the address of a local BasisStore unconditionally -- a pointer chosen at run time
    // would keep the in-flight registers of the store in scratch memory)
    const bool have_store = store != nullptr && store->g != nullptr;
    constexpr int PF = (DD + 63) / 64;         // 16-byte loads per lane for one basis
    static_assert(PF <= 4, "BasisStore::pf holds one basis of at most 256 entries");
    int it = 0;
    for (; it < max_iter; ++it) {
        ++iters;
        const Blk pre_cp = blk_sub(last_state, old_cp);
        bool warm = it > 0;
        bool from_slot = false;
        const int sweeps_before = sweeps;
        if (have_store && it < store->nprev && (it == 0 || store->use_prev)) {
            from_slot = true;
            FBX_WAVE_SYNC();
            if (store->pf_slot != it) store->template prefetch<DD>(it, lane);      // nothing in flight: fetch now
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int idx = lane + 64 * u;
                if (idx < DD) L.Vs[sys_linear<LdsT::D>(idx)] = store->pf[u];
            }
            store->pf_slot = -1;
#ifdef FBX_DBG_CORRUPT_BASIS                   // test hook: damage every basis loaded for Dykstra iteration 1
            if (it == 1 && lane < 3) L.Vs[sys_linear<LdsT::D>(17 * lane)].re += 0.25;
#endif
            warm = true;
        }
        // the basis of the NEXT iteration is requested before this iteration's decomposition, so that
        // its HBM / L2 latency lies behind the Jacobi sweeps
        if (have_store && store->use_prev && it + 1 < store->nprev && it + 1 < store->cap)
            store->template prefetch<DD>(it + 1, lane);
        const Blk cp = proj_cp_blk<NQ>(pre_cp, L, lane, sweeps, warm, from_slot);
        if (have_store && it < store->cap && (it == 0 || store->write_all)) {
            // Write-back of the basis (4 KB per decomposition).  Slot j > 0 is only read by a call whose outer
            // step is below FBX_BASIS_STEP, and the outer step shrinks by ~1.2x per iteration: the slots are
            // written from FBX_BASIS_WRITE_STEP (30x the threshold; 4x and 10x measured 1.3 % slower at B = 1024, 30x is free) on -- before that only
            // slot 0.  (Writing every basis of every call was 0.9 GB of write-back per 1024-item launch.)
            if (!(from_slot && sweeps == sweeps_before)) {       // no sweep on the slot's own basis: nothing changed
                fbx_global_cplx_ptr dst = (fbx_global_cplx_ptr)(store->g + (size_t)it * DD);
#pragma unroll
                for (int u = 0; u < PF; ++u) {
                    const int idx = lane + 64 * u;
                    if (idx < DD) { const cplx w = L.Vs[sys_linear<LdsT::D>(idx)]; dst[idx] = fbx_v2d{w.re, w.im}; }
                }
            }
        }
        const Blk new_cp = blk_sub(cp, pre_cp);
        const Blk pre_tp = blk_sub(cp, old_tp);
        new_state = trace_preserving ? proj_tp_blk<NQ>(pre_tp, L, lane)
                                     : proj_tni_blk<NQ>(pre_tp, L, lane, sweeps);
        const Blk new_tp = blk_sub(new_state, pre_tp);
        PH_STOP(*L.pc, 6);
        double s1 = blk_norm2(blk_sub(new_cp, old_cp));
        double s2 = blk_norm2(blk_sub(new_tp, old_tp));
        double i1r, i1i, i2r, i2i;
        blk_dotc(old_tp, blk_sub(new_state, last_state), i1r, i1i);
        blk_dotc(old_cp, blk_sub(cp, last_cp), i2r, i2i);
        s1 = wave_sum(s1); s2 = wave_sum(s2);
        i1r = wave_sum(i1r); i1i = wave_sum(i1i); i2r = wave_sum(i2r); i2i = wave_sum(i2i);
        // |z| = |z|^2 rsqrt(|z|^2): the IEEE sqrt expansion is ~40 instructions, twice per iteration
        const double a1 = i1r * i1r + i1i * i1i, a2 = i2r * i2r + i2i * i2i;
        const double m1 = a1 > 1e-300 ? a1 * fast_rsqrt(a1) : 0.0, m2 = a2 > 1e-300 ? a2 * fast_rsqrt(a2) : 0.0;
        const double crit = uniform(s1 + s2 + 2.0 * m1 + 2.0 * m2);
        PH_STOP(*L.pc, 7);
        if (!(crit >= 1e-4)) { ++it; break; }        // converged -- or not finite (NaN input): never spin
        old_cp = new_cp; old_tp = new_tp; last_cp = cp; last_state = new_state;
    }
    if (have_store) {
        const int written = store->write_all ? it : 1;           // slots of this call that hold a basis now
        store->nprev = written < store->cap ? written : store->cap;
        store->pf_slot = -1;
    }
    return new_state;
}

// ---- Choi <-> Pauli-coefficient transforms --------------------------------------------
// R_ij = (1/d) tr[(P_j^T (x) P_i) E]  (real for Hermitian E; this is the Pauli-Liouville
// matrix of the channel, superoperator_transformations.py:364-371), and its inverse
// E = (1/d) sum_ij R_ij (P_j^T (x) P_i).
//
// Both are radix-2 butterflies over the 2n tensor sites, in place on the row-major complex matrix
// staged in Mw (leading dimension D + 1): a site pairs one row-index bit with one column-index
// bit, and its four entries (c00, c11, c01, c10) go to (I, Z, X, Y) = (c00 + c11, c00 - c11,
// c01 + c10, +-i (c01 - c10)); -i for input-qubit sites (P_j^T), +i for output-qubit sites (P_i).
// One quad per lane and stage: 4 LDS reads + 4 writes + 8 adds, no index arithmetic beyond two
// bit inserts -- against D terms per entry with per-term popcounts in the direct sum.
__device__ __forceinline__ int insert_zero_bits2(int v, int lo, int hi) {   // lo < hi bit positions
    int r = (v & ((1 << lo) - 1)) | ((v >> lo) << (lo + 1));
    r = (r & ((1 << hi) - 1)) | ((r >> hi) << (hi + 1));
    return r;
}
template <int NQ, bool INVERSE, int LDM = (1 << (2 * NQ)) + 1>
__device__ __forceinline__ void pauli_site_stage(cplx* M, int lane, int pbit, int qbit, double ysign) {
    constexpr int D = 1 << (2 * NQ), LD = LDM, NQUAD = D * D / 4;
    if (lane < NQUAD) {
        const int lo = pbit < qbit ? pbit : qbit, hi = pbit < qbit ? qbit : pbit;
        const int base = insert_zero_bits2(lane, lo, hi);
        const int i00 = base, i11 = base | (1 << pbit) | (1 << qbit), i01 = base | (1 << qbit), i10 = base | (1 << pbit);
        const int a00 = (i00 >> (2 * NQ)) * LD + (i00 & (D - 1)), a11 = (i11 >> (2 * NQ)) * LD + (i11 & (D - 1));
        const int a01 = (i01 >> (2 * NQ)) * LD + (i01 & (D - 1)), a10 = (i10 >> (2 * NQ)) * LD + (i10 & (D - 1));
        const cplx c00 = M[a00], c11 = M[a11], c01 = M[a01], c10 = M[a10];
        cplx o00, o11, o01, o10;
        if (!INVERSE) {
            o00.re = c00.re + c11.re; o00.im = c00.im + c11.im;          // I
            o11.re = c00.re - c11.re; o11.im = c00.im - c11.im;          // Z
            o01.re = c01.re + c10.re; o01.im = c01.im + c10.im;          // X
            const double dr = c01.re - c10.re, di = c01.im - c10.im;     // Y = +-i (c01 - c10)
            o10.re = -ysign * di; o10.im = ysign * dr;
        } else {                                                          // (I, Z, X, Y) at (00, 11, 01, 10)
            o00.re = 0.5 * (c00.re + c11.re); o00.im = 0.5 * (c00.im + c11.im);
            o11.re = 0.5 * (c00.re - c11.re); o11.im = 0.5 * (c00.im - c11.im);
            // c01 = (X - s i Y)/2, c10 = (X + s i Y)/2 ;  i Y = (-Y.im, Y.re)
            const double yr = -ysign * c10.im, yi = ysign * c10.re;       // s * i * Y
            o01.re = 0.5 * (c01.re - yr); o01.im = 0.5 * (c01.im - yi);
            o10.re = 0.5 * (c01.re + yr); o10.im = 0.5 * (c01.im + yi);
        }
        M[a00] = o00; M[a11] = o11; M[a01] = o01; M[a10] = o10;
    }
}
// (row, col) of the matrix entry that holds coefficient R[i][j] after the forward stages
template <int NQ>
__device__ __forceinline__ void pauli_coeff_position(int i, int j, int& row, int& col) {
    row = 0; col = 0;
#pragma unroll
    for (int t = 0; t < NQ; ++t) {                 // digit t (least significant first) = 2 rowbit + colbit
        const int di = (i >> (2 * t)) & 3, dj = (j >> (2 * t)) & 3;
        row |= ((di >> 1) & 1) << t; col |= (di & 1) << t;                    // output qubit -> low bits
        row |= ((dj >> 1) & 1) << (NQ + t); col |= (dj & 1) << (NQ + t);      // input qubit -> high bits
    }
}

// E staged in Mw (destroyed) -> real coefficients Rb[D*D]
template <int NQ>
__device__ void choi_to_pauli_real(cplx* Mw, double* Rb, int lane) {
    constexpr int d = 1 << NQ, D = d * d, LD = D + 1;
#pragma unroll
    for (int t = NQ - 1; t >= 0; --t) {            // input-qubit sites: row bit NQ + t, col bit NQ + t
        pauli_site_stage<NQ, false>(Mw, lane, 2 * NQ + NQ + t, NQ + t, -1.0);
        FBX_WAVE_SYNC();
    }
#pragma unroll
    for (int t = NQ - 1; t >= 0; --t) {            // output-qubit sites
        pauli_site_stage<NQ, false>(Mw, lane, 2 * NQ + t, t, +1.0);
        FBX_WAVE_SYNC();
    }
    for (int idx = lane; idx < D * D; idx += 64) {
        int row, col;
        pauli_coeff_position<NQ>(idx / D, idx % D, row, col);
        Rb[(idx % D) * D + idx / D] = Mw[row * LD + col].re / d;     // transposed: Rb[j * D + i] = R[i][j]
    }
}

// real coefficients Rb -> Choi block of this lane (Mw is scratch)
template <int NQ>
__device__ Blk pauli_real_to_choi_blk(const double* Rb, cplx* Mw, int lane) {
    constexpr int d = 1 << NQ, D = d * d, LD = D + 1;
    FBX_WAVE_SYNC();
    for (int idx = lane; idx < D * D; idx += 64) {
        int row, col;
        pauli_coeff_position<NQ>(idx / D, idx % D, row, col);
        cplx v; v.re = Rb[(idx % D) * D + idx / D] * d; v.im = 0.0;     // E = d * F^{-1}(R); Rb[j * D + i] = R[i][j]
        Mw[row * LD + col] = v;
    }
    FBX_WAVE_SYNC();
#pragma unroll
    for (int t = 0; t < NQ; ++t) {
        pauli_site_stage<NQ, true>(Mw, lane, 2 * NQ + t, t, +1.0);
        FBX_WAVE_SYNC();
    }
#pragma unroll
    for (int t = 0; t < NQ; ++t) {
        pauli_site_stage<NQ, true>(Mw, lane, 2 * NQ + NQ + t, NQ + t, -1.0);
        FBX_WAVE_SYNC();
    }
    const Blk out = blk_load<D, LD>(Mw, lane);
    FBX_WAVE_SYNC();
    return out;
}

}  // namespace fbx
