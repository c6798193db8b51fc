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

#ifndef FBX_WARM_START
#define FBX_WARM_START 1     // reuse the previous eigenvectors inside one Dykstra run (reset per run)
#endif

namespace fbx {

// LDS work area shared by the routines below (carved by the kernel)
template <int NQ>
struct ChoiLds {
    static constexpr int d = 1 << NQ, D = d * d, LD = D + 1, LDs = d + 1;
    cplx* Mw;      // [D * LD]   row-major staging matrix (partial trace, Pauli transforms)
    cplx* Ms;      // [D * D]    Jacobi work matrix, element-major block layout
    cplx* Vs;      // [D * D]    eigenvectors, same layout
    double* lam;   // [D]
    JRec* rec;     // [D / 2]    published rotations of the pipelined Jacobi
    cplx* pt;      // [d * LDs]  partial trace (d x d), row-major
    cplx* pts;     // [d * d]    partial trace in the Jacobi layout (TNI only)
    cplx* ptV;     // [d * d]    its eigenvectors (TNI only)
    PhaseClock* pc = nullptr;   // diagnostics (FBX_PHASE_TIMERS builds)
    static constexpr size_t bytes() {
        return sizeof(cplx) * (D * LD + 2 * D * D + d * LDs + 2 * d * d) + sizeof(double) * D + sizeof(JRec) * (D / 2 + 1);
    }
    __device__ void carve(char*& p) {
        Mw = (cplx*)p; p += sizeof(cplx) * D * LD;
        Ms = (cplx*)p; p += sizeof(cplx) * D * D;
        Vs = (cplx*)p; p += sizeof(cplx) * D * D;
        pt = (cplx*)p; p += sizeof(cplx) * d * LDs;
        pts = (cplx*)p; p += sizeof(cplx) * d * d;
        ptV = (cplx*)p; p += sizeof(cplx) * d * d;
        rec = (JRec*)p; p += sizeof(JRec) * (D / 2 + 1);
        lam = (double*)p; p += sizeof(double) * D;
    }
};

// ---- CP projection: Hermitise, eigh, clamp negative eigenvalues, rebuild -----------------
// project_superoperators.py:19-34.  `x` need not be Hermitian.  `sweeps` accumulates Jacobi
// sweeps (diagnostics).
template <int NQ>
__device__ Blk proj_cp_blk(const Blk& x, ChoiLds<NQ>& L, int lane, int& sweeps, bool warm = false) {
    constexpr int D = ChoiLds<NQ>::D;
    __syncthreads();                       // previous readers of Ms / Vs are done
    sys_store<D>(L.Ms, lane, x);
    __syncthreads();
    const Blk xa = sys_load_adjoint<D>(L.Ms, lane);
    Blk h;
#pragma unroll
    for (int e = 0; e < 4; ++e) { h.re[e] = 0.5 * (x.re[e] + xa.re[e]); h.im[e] = 0.5 * (x.im[e] + xa.im[e]); }
    __syncthreads();
    sys_store<D>(L.Ms, lane, h);
    __syncthreads();
    PH_STOP(*L.pc, 2);
    // warm start: the eigenvectors of the previous projection (still in Vs) nearly diagonalise
    // this matrix, because consecutive Dykstra iterates are close
    if (warm) jacobi_rotate_into_basis<D>(L.Ms, L.Vs, (cplx*)L.Mw, lane);
    sweeps += jacobi_eigh_lds<D>(L.Ms, L.Vs, L.rec, lane, !warm);
    PH_STOP(*L.pc, 0);
    if (lane < D) {
        const double l = L.Ms[sys_index<D>(lane, lane)].re;
        L.lam[lane] = l < 0.0 ? 0.0 : l;
    }
    __syncthreads();
    const Blk out = reconstruct_blk<D>(L.Vs, L.lam, lane);
    PH_STOP(*L.pc, 1);
    return out;
}

// ---- partial trace over the output space into L.pt (d x d): calculational.py:5-35 with
// keep=[0], dims=[d, d].  Stages `x` through Mw.
template <int NQ>
__device__ void partial_trace_out(const Blk& x, ChoiLds<NQ>& L, int lane) {
    constexpr int d = ChoiLds<NQ>::d, D = ChoiLds<NQ>::D, LD = ChoiLds<NQ>::LD, LDs = ChoiLds<NQ>::LDs;
    __syncthreads();
    blk_store<D, LD>(L.Mw, lane, x);
    __syncthreads();
    if (lane < d * d) {
        const int i = lane / d, ip = lane % d;
        cplx s; s.re = 0.0; s.im = 0.0;
#pragma unroll
        for (int o = 0; o < d; ++o) {
            const cplx v = L.Mw[(i * d + o) * LD + ip * d + o];
            s.re += v.re; s.im += v.im;
        }
        L.pt[i * LDs + ip] = s;
    }
    __syncthreads();
}

// subtract kron(corr / d, I_d) where corr (d x d) is in L.pt
template <int NQ>
__device__ __forceinline__ Blk subtract_kron_pt(const Blk& x, const ChoiLds<NQ>& L, int lane) {
    constexpr int d = ChoiLds<NQ>::d, D = ChoiLds<NQ>::D, LDs = ChoiLds<NQ>::LDs, NB = D / 2;
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
template <int NQ>
__device__ Blk proj_tp_blk(const Blk& x, ChoiLds<NQ>& L, int lane) {
    constexpr int d = ChoiLds<NQ>::d, LDs = ChoiLds<NQ>::LDs;
    partial_trace_out<NQ>(x, L, lane);
    if (lane < d) L.pt[lane * LDs + lane].re -= 1.0;       // pt - I
    __syncthreads();
    return subtract_kron_pt<NQ>(x, L, lane);
}

// ---- TNI projection: project_superoperators.py:37-59 (d x d eigh of the partial trace,
// eigenvalues above 1 clamped to 1)
template <int NQ>
__device__ Blk proj_tni_blk(const Blk& x, ChoiLds<NQ>& L, int lane, int& sweeps) {
    constexpr int d = ChoiLds<NQ>::d, LDs = ChoiLds<NQ>::LDs;
    partial_trace_out<NQ>(x, L, lane);
    // keep pt in registers, Hermitise a copy for the eigensolver
    const Blk ptb = blk_load<d, LDs>(L.pt, lane);
    const Blk pta = blk_load_adjoint<d, LDs>(L.pt, lane);
    Blk h;
#pragma unroll
    for (int e = 0; e < 4; ++e) { h.re[e] = 0.5 * (ptb.re[e] + pta.re[e]); h.im[e] = 0.5 * (ptb.im[e] + pta.im[e]); }
    __syncthreads();
    sys_store<d>(L.pts, lane, h);
    __syncthreads();
    sweeps += jacobi_eigh_lds<d>(L.pts, L.ptV, L.rec, lane);
    if (lane < d) {
        const double l = L.pts[sys_index<d>(lane, lane)].re;
        L.lam[lane] = l > 1.0 ? 1.0 : l;
    }
    __syncthreads();
    const Blk proj = reconstruct_blk<d>(L.ptV, L.lam, lane);
    __syncthreads();
    blk_store<d, LDs>(L.pt, lane, blk_sub(ptb, proj));      // pt - projection
    __syncthreads();
    return subtract_kron_pt<NQ>(x, L, lane);
}

// ---- Dykstra: project_superoperators.py:87-144.  Stops on the Birgin-Raydan functional
// < 1e-4 (no iteration cap in the reference; `max_iter` is a safety net that is never the
// binding constraint in practice).  Returns the last TP / TNI iterate.
template <int NQ>
__device__ Blk proj_physical_blk(const Blk& x, bool trace_preserving, ChoiLds<NQ>& L, int lane,
                                 int& iters, int& sweeps, int max_iter = 100000) {
    Blk old_cp = blk_zero(), old_tp = blk_zero(), last_cp = blk_zero();
    Blk last_state = x, new_state = x;
    for (int it = 0; it < max_iter; ++it) {
        ++iters;
        const Blk pre_cp = blk_sub(last_state, old_cp);
        const Blk cp = proj_cp_blk<NQ>(pre_cp, L, lane, sweeps, FBX_WARM_START && it > 0);
        const Blk new_cp = blk_sub(cp, pre_cp);
        const Blk pre_tp = blk_sub(cp, old_tp);
        new_state = trace_preserving ? proj_tp_blk<NQ>(pre_tp, L, lane)
                                     : proj_tni_blk<NQ>(pre_tp, L, lane, sweeps);
        const Blk new_tp = blk_sub(new_state, pre_tp);
        double s1 = blk_norm2(blk_sub(new_cp, old_cp));
        double s2 = blk_norm2(blk_sub(new_tp, old_tp));
        double i1r, i1i, i2r, i2i;
        blk_dotc(old_tp, blk_sub(new_state, last_state), i1r, i1i);
        blk_dotc(old_cp, blk_sub(cp, last_cp), i2r, i2i);
        s1 = wave_sum(s1); s2 = wave_sum(s2);
        i1r = wave_sum(i1r); i1i = wave_sum(i1i); i2r = wave_sum(i2r); i2i = wave_sum(i2i);
        const double crit = uniform(s1 + s2 + 2.0 * sqrt(i1r * i1r + i1i * i1i)
                                    + 2.0 * sqrt(i2r * i2r + i2i * i2i));
        if (crit < 1e-4) break;
        old_cp = new_cp; old_tp = new_tp; last_cp = cp; last_state = new_state;
    }
    return new_state;
}

// ---- Choi <-> Pauli-coefficient transforms --------------------------------------------
// R_ij = (1/d) tr[(P_j^T (x) P_i) E]  (real for Hermitian E; this is the Pauli-Liouville
// matrix of the channel, superoperator_transformations.py:364-371), and its inverse
// E = (1/d) sum_ij R_ij (P_j^T (x) P_i).  E is staged in Mw (complex, LD), R in `Rb`
// (real, row-major D x D).  Every lane handles D*D/64 entries.
template <int NQ>
__device__ void choi_to_pauli_real(const cplx* Mw, double* Rb, int lane) {
    constexpr int d = 1 << NQ, D = d * d, LD = D + 1;
    for (int idx = lane; idx < D * D; idx += 64) {
        const int i = idx / D, j = idx % D;
        int xi, zi, yi, xj, zj, yj;
        pauli_masks<NQ>(i, xi, zi, yi);
        pauli_masks<NQ>(j, xj, zj, yj);
        const int ph = (yi + yj) & 3;
        double acc = 0.0;
        for (int a = 0; a < d; ++a) {
#pragma unroll
            for (int b = 0; b < d; ++b) {
                const cplx v = Mw[((a ^ xj) * d + (b ^ xi)) * LD + a * d + b];
                const int sgn = (__popc(a & zj) + __popc((b ^ xi) & zi)) & 1;
                // real part of i^ph * v
                double t = (ph == 0) ? v.re : (ph == 1) ? -v.im : (ph == 2) ? -v.re : v.im;
                acc += sgn ? -t : t;
            }
        }
        Rb[idx] = acc / d;
    }
}

template <int NQ>
__device__ Blk pauli_real_to_choi_blk(const double* Rb, int lane) {
    constexpr int d = 1 << NQ, D = d * d, NB = D / 2;
    Blk out = blk_zero();
    if (lane < NB * NB) {
        const int I = lane / NB, J = lane % NB;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = 2 * I + (e >> 1), col = 2 * J + (e & 1);
            const int ap = row / d, bp = row % d, a = col / d, b = col % d;
            const int xj = a ^ ap, xi = b ^ bp;
            double re = 0.0, im = 0.0;
            for (int zj = 0; zj < d; ++zj) {
#pragma unroll
                for (int zi = 0; zi < d; ++zi) {
                    const int j = pauli_index<NQ>(xj, zj), i = pauli_index<NQ>(xi, zi);
                    const int ph = (__popc(xj & zj) + __popc(xi & zi)) & 3;
                    const int sgn = (__popc(ap & zj) + __popc(b & zi)) & 1;
                    double v = Rb[i * D + j];
                    v = sgn ? -v : v;
                    if (ph == 0) re += v; else if (ph == 1) im += v;
                    else if (ph == 2) re -= v; else im -= v;
                }
            }
            out.re[e] = re / d; out.im[e] = im / d;
        }
    }
    return out;
}

}  // namespace fbx
