// fbx_superop.hip -- batched superoperator algebra: representation changes, Choi projections,
// channel application, process fidelity.  One 64-lane wavefront per item; matrices are staged
// in LDS and every basis change uses the sparsity of the Pauli matrices (each vec(P_k) has d
// non-zero entries, all in {+-1, +-i}) instead of a dense D x D x D product.
//
// Reference functions (file:line under forest/benchmarking/):
//   operator_tools/superoperator_transformations.py:82-371   (pairwise conversions)
//   operator_tools/project_superoperators.py:19-144          (CP / TP / TNI / physical)
//   operator_tools/apply_superoperator.py:60-90              (apply_choi_matrix_2_state)
//   distance_measures.py:271-359                             (entanglement / process fidelity)
#include "fbx_choi.hpp"
#include "fbx_eigh64.hpp"
#include <cstdlib>
#include <algorithm>

namespace fbx {

// ---------------------------------------------------------------------------------------------
// device primitives on a D x D complex matrix, row-major with leading dimension LD, in LDS
// ---------------------------------------------------------------------------------------------

// vec(P_k)[c*d + r] = P_k[r][c]; non-zero iff c = r ^ x_k with value i^{ny} (-1)^{popc(c & z)}
// multiply v by i^ph
__device__ __forceinline__ cplx mul_iph(cplx v, int ph) {
    cplx o;
    switch (ph & 3) {
        case 0: o = v; break;
        case 1: o.re = -v.im; o.im = v.re; break;
        case 2: o.re = -v.re; o.im = -v.im; break;
        default: o.re = v.im; o.im = -v.re; break;
    }
    return o;
}

// out = scale * P2C^H in P2C  (superop -> Pauli-Liouville with scale 1/d; Choi -> chi with 1/d^2),
// P2C columns = vec(P_k): out[k][l] = scale * sum_{r,s} conj(vP_k[r]) in[r][s] vP_l[s]
template <int NQ, int NT = 64, int LD = (1 << (2 * NQ)) + 1>
__device__ void to_pauli_basis(const cplx* in, cplx* out, double scale, int lane) {
    constexpr int d = 1 << NQ, D = d * d;
    for (int idx = lane; idx < D * D; idx += NT) {
        const int k = idx / D, l = idx % D;
        int xk, zk, yk, xl, zl, yl;
        pauli_masks<NQ>(k, xk, zk, yk);
        pauli_masks<NQ>(l, xl, zl, yl);
        double re = 0.0, im = 0.0;
        for (int rk = 0; rk < d; ++rk) {          // row index of P_k's non-zero: (rk, ck = rk ^ xk)
            const int ck = rk ^ xk;
            const int sk = __popc(ck & zk) & 1;
#pragma unroll
            for (int rl = 0; rl < d; ++rl) {
                const int cl = rl ^ xl;
                const int sl = __popc(cl & zl) & 1;
                const cplx v = in[(ck * d + rk) * LD + cl * d + rl];
                // conj(i^yk) * i^yl = i^(yl - yk)
                const cplx w = mul_iph(v, (yl - yk) & 3);
                if (sk ^ sl) { re -= w.re; im -= w.im; } else { re += w.re; im += w.im; }
            }
        }
        cplx o; o.re = re * scale; o.im = im * scale;
        out[k * LD + l] = o;
    }
}

// out = scale * P2C in P2C^H: out[r][s] = scale * sum_{k,l} vP_k[r] in[k][l] conj(vP_l[s])
template <int NQ, int NT = 64, int LD = (1 << (2 * NQ)) + 1>
__device__ void from_pauli_basis(const cplx* in, cplx* out, double scale, int lane) {
    constexpr int d = 1 << NQ, D = d * d;
    for (int idx = lane; idx < D * D; idx += NT) {
        const int r = idx / D, s = idx % D;
        const int cr = r / d, rr = r % d, cs = s / d, rs = s % d;   // vec index = col * d + row
        const int xk = rr ^ cr, xl = rs ^ cs;
        double re = 0.0, im = 0.0;
        for (int zk = 0; zk < d; ++zk) {
            const int k = pauli_index<NQ>(xk, zk);
            const int yk = __popc(xk & zk), sk = __popc(cr & zk) & 1;
#pragma unroll
            for (int zl = 0; zl < d; ++zl) {
                const int l = pauli_index<NQ>(xl, zl);
                const int yl = __popc(xl & zl), sl = __popc(cs & zl) & 1;
                const cplx w = mul_iph(in[k * LD + l], (yk - yl) & 3);
                if (sk ^ sl) { re -= w.re; im -= w.im; } else { re += w.re; im += w.im; }
            }
        }
        cplx o; o.re = re * scale; o.im = im * scale;
        out[r * LD + s] = o;
    }
}

// Site-factored forms of the two transforms above: P2C factors over the qubits, so the change of basis
// is 2n in-place butterfly stages (one quad per thread and stage) and a bit-permuting copy instead of a
// D-term sum per entry.  Element index = row * D + col with row = (a_{n-1}..a_0 b_{n-1}..b_0) = vec
// index c*d + r; the stages pair (a_t, b_t) of the row (conj: -i) and of the column (+i); Pauli digit
// 2 a_t + b_t of label k is I, X, Y, Z, so entry [k][l] of the Pauli side sits at row site_index(k),
// column site_index(l).  The forward form destroys `in`.  NT threads, NT >= D*D/4 or a multiple loop.
template <int NQ>
__device__ __forceinline__ int site_index(int k) {
    int r = 0;
#pragma unroll
    for (int t = 0; t < NQ; ++t) r |= (((k >> (2 * t + 1)) & 1) << (NQ + t)) | (((k >> (2 * t)) & 1) << t);
    return r;
}
template <int NT>
__device__ __forceinline__ void sites_sync() { if constexpr (NT <= 64) FBX_WAVE_SYNC(); else __syncthreads(); }
template <int NQ, bool INVERSE, int NT, int LD>
__device__ __forceinline__ void site_stages(cplx* M, int t) {
    static_assert(NT >= (1 << (4 * NQ)) / 4, "one quad per thread");
#pragma unroll
    for (int q = NQ - 1; q >= 0; --q) { pauli_site_stage<NQ, INVERSE, LD>(M, t, 3 * NQ + q, 2 * NQ + q, -1.0); sites_sync<NT>(); }
#pragma unroll
    for (int q = NQ - 1; q >= 0; --q) { pauli_site_stage<NQ, INVERSE, LD>(M, t, NQ + q, q, +1.0); sites_sync<NT>(); }
}
template <int NQ, int NT = 64, int LD = (1 << (2 * NQ)) + 1>
__device__ void to_pauli_sites(cplx* in, cplx* out, double scale, int t) {
    constexpr int D = 1 << (2 * NQ);
    site_stages<NQ, false, NT, LD>(in, t);
    for (int idx = t; idx < D * D; idx += NT) {
        const int k = idx / D, l = idx % D;
        cplx v = in[site_index<NQ>(k) * LD + site_index<NQ>(l)];
        v.re *= scale; v.im *= scale;
        out[k * LD + l] = v;
    }
}
// P2C x P2C^H = D * (inverse of the forward stages)
template <int NQ, int NT = 64, int LD = (1 << (2 * NQ)) + 1>
__device__ void from_pauli_sites(const cplx* in, cplx* out, double scale, int t) {
    constexpr int D = 1 << (2 * NQ);
    const double s = scale * D;
    for (int idx = t; idx < D * D; idx += NT) {
        const int k = idx / D, l = idx % D;
        cplx v = in[k * LD + l];
        v.re *= s; v.im *= s;
        out[site_index<NQ>(k) * LD + site_index<NQ>(l)] = v;
    }
    sites_sync<NT>();
    site_stages<NQ, true, NT, LD>(out, t);
}

// choi <-> superop reshuffle (superoperator_transformations.py:267-277,351-361):
// out[(p,q)][(r,s)] = in[(s,q)][(r,p)]
template <int NQ, int NT = 64, int LD = (1 << (2 * NQ)) + 1>
__device__ void reshuffle(const cplx* in, cplx* out, int lane) {
    constexpr int d = 1 << NQ, D = d * d;
    for (int idx = lane; idx < D * D; idx += NT) {
        const int row = idx / D, col = idx % D;
        const int p = row / d, q = row % d, r = col / d, s = col % d;
        out[row * LD + col] = in[(s * d + q) * LD + r * d + p];
    }
}

// kraus -> choi (sum vec(K) vec(K)^H) or superop (sum conj(K) (x) K); K ops row-major d x d in HBM
template <int NQ, int NT = 64, int LD = (1 << (2 * NQ)) + 1>
__device__ void kraus_to(const double* __restrict__ kraus, int K, bool to_superop, cplx* out, cplx* kb,
                         int lane) {
    constexpr int d = 1 << NQ, D = d * d;
    for (int idx = lane; idx < K * D; idx += NT) { kb[idx].re = kraus[2 * idx]; kb[idx].im = kraus[2 * idx + 1]; }
    __syncthreads();
    for (int idx = lane; idx < D * D; idx += NT) {
        const int row = idx / D, col = idx % D;
        double re = 0.0, im = 0.0;
        for (int t = 0; t < K; ++t) {
            cplx a, b;     // out += a * b (a possibly conjugated below)
            if (to_superop) {   // kron(conj(K), K)[(i,k)][(j,l)] = conj(K[i][j]) K[k][l]
                const int i = row / d, k = row % d, j = col / d, l = col % d;
                a = kb[t * D + i * d + j]; a.im = -a.im;
                b = kb[t * D + k * d + l];
            } else {            // vec(K)[c*d + r] = K[r][c]; choi[row][col] = vK[row] conj(vK[col])
                a = kb[t * D + (row % d) * d + row / d];
                b = kb[t * D + (col % d) * d + col / d]; b.im = -b.im;
            }
            re += a.re * b.re - a.im * b.im;
            im += a.re * b.im + a.im * b.re;
        }
        cplx o; o.re = re; o.im = im;
        out[row * LD + col] = o;
    }
}

// matrix absolute value through the eigendecomposition, as choi2kraus -> kraus2choi does it
// (superoperator_transformations.py:325-336): numpy eigh reads the LOWER triangle; eigenvalues
// with |lambda| <= tol are dropped; sqrt of a negative eigenvalue is imaginary, so the rebuilt
// matrix is sum |lambda| v v^H.
template <int NQ>
__device__ void abs_via_eigh(const cplx* in, cplx* out, ChoiLds<NQ>& L, double tol, int lane) {
    constexpr int d = 1 << NQ, D = d * d, LD = D + 1, NB = D / 2;
    Blk h = blk_zero();
    if (lane < NB * NB) {
        const int I = lane / NB, J = lane % NB;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int r = 2 * I + (e >> 1), c = 2 * J + (e & 1);
            if (r > c) { const cplx v = in[r * LD + c]; h.re[e] = v.re; h.im[e] = v.im; }
            else if (r < c) { const cplx v = in[c * LD + r]; h.re[e] = v.re; h.im[e] = -v.im; }
            else { h.re[e] = in[r * LD + c].re; h.im[e] = 0.0; }
        }
    }
    __syncthreads();
    sys_store<D>(L.Ms, lane, h);
    __syncthreads();
    jacobi_eigh_lds<D>(L.Ms, L.Vs, L.rec, lane);
    if (lane < D) {
        const double l = fabs(L.Ms[sys_index<D>(lane, lane)].re);
        L.lam[lane] = l > tol ? l : 0.0;
    }
    __syncthreads();
    const Blk a = reconstruct_blk<D>(L.Vs, L.lam, lane);
    blk_store<D, LD>(out, lane, a);
    __syncthreads();
}

template <int NQ, int NT = 64, int LD = (1 << (2 * NQ)) + 1>
__device__ void load_matrix(const double* __restrict__ g, cplx* m, int lane) {
    constexpr int d = 1 << NQ, D = d * d;
    for (int idx = lane; idx < D * D; idx += NT) {
        cplx v; v.re = g[2 * idx]; v.im = g[2 * idx + 1];
        m[(idx / D) * LD + idx % D] = v;
    }
}
template <int NQ, int NT = 64, int LD = (1 << (2 * NQ)) + 1>
__device__ void store_matrix(const cplx* m, double* __restrict__ g, int lane) {
    constexpr int d = 1 << NQ, D = d * d;
    for (int idx = lane; idx < D * D; idx += NT) {
        const cplx v = m[(idx / D) * LD + idx % D];
        g[2 * idx] = v.re; g[2 * idx + 1] = v.im;
    }
}

// ---------------------------------------------------------------------------------------------
// fbx_convert
// ---------------------------------------------------------------------------------------------
// EIGH = false: the conversions that never pass through choi2kraus need no eigensolver arrays -- 10 KB instead of
// 23 KB of LDS per wavefront (2 qubits), i.e. 15 instead of 6 wavefronts per CU on a kernel that only waits for HBM.
template <int NQ, bool EIGH = true>
__global__ void __launch_bounds__(64)
convert_kernel(int from, int to, long long B, const double* __restrict__ in, int K, double* __restrict__ out) {
    constexpr int d = 1 << NQ, D = d * d, LD = D + 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* p = smem;
    ChoiLds<NQ> L;
    if constexpr (EIGH) {
        L.carve(p);
        p = smem + ((ChoiLds<NQ>::bytes() + 15) & ~(size_t)15);  // (no pointer -> integer -> pointer: keeps the LDS address space)
    }
    cplx* A = (cplx*)p; p += sizeof(cplx) * D * LD;
    cplx* Bm = (cplx*)p; p += sizeof(cplx) * D * LD;
    cplx* kb = (cplx*)p;
    const int lane = threadIdx.x;
    const long long item = blockIdx.x;
    cplx* cur = A; cplx* nxt = Bm;
    auto swap = [&]() { cplx* t = cur; cur = nxt; nxt = t; __syncthreads(); };
    const double inv_d = 1.0 / d;

    // stage 1: bring the input to Choi (or directly to the target when a shorter path exists)
    int rep = from;
    if (from == FBX_REP_KRAUS) {
        const bool sup = (to == FBX_REP_SUPEROP || to == FBX_REP_PAULI_LIOUVILLE);
        kraus_to<NQ>(in + item * (long long)K * D * 2, K, sup, cur, kb, lane);
        __syncthreads();
        rep = sup ? FBX_REP_SUPEROP : FBX_REP_CHOI;
    } else {
        load_matrix<NQ>(in + item * (long long)D * D * 2, cur, lane);
        __syncthreads();
    }
    // walk the representation graph: chi -> choi <-> superop <-> pauli-liouville, choi -> chi
    const bool kraus_chi = (from == FBX_REP_KRAUS && to == FBX_REP_CHI);
    while (rep != to) {
        if (rep == FBX_REP_CHI) {                       // chi2choi: p2c chi p2c^H
            from_pauli_sites<NQ>(cur, nxt, 1.0, lane); swap(); rep = FBX_REP_CHOI;
        } else if (rep == FBX_REP_CHOI) {
            if (to == FBX_REP_CHI) {
                if (!kraus_chi) {                       // through choi2kraus (eigh, |C|, tol 1e-9)
                    if constexpr (EIGH) { abs_via_eigh<NQ>(cur, nxt, L, 1e-9, lane); swap(); }
                }
                to_pauli_sites<NQ>(cur, nxt, inv_d * inv_d, lane); swap(); rep = FBX_REP_CHI;
            } else {
                reshuffle<NQ>(cur, nxt, lane); swap(); rep = FBX_REP_SUPEROP;
            }
        } else if (rep == FBX_REP_SUPEROP) {
            if (to == FBX_REP_PAULI_LIOUVILLE) {
                to_pauli_sites<NQ>(cur, nxt, inv_d, lane); swap(); rep = FBX_REP_PAULI_LIOUVILLE;
            } else {
                reshuffle<NQ>(cur, nxt, lane); swap(); rep = FBX_REP_CHOI;
            }
        } else {                                        // pauli-liouville -> superop
            from_pauli_sites<NQ>(cur, nxt, inv_d, lane); swap(); rep = FBX_REP_SUPEROP;
        }
    }
    store_matrix<NQ>(cur, out + item * (long long)D * D * 2, lane);
}

template <int NQ>
static int launch_convert(int from, int to, int64_t B, const double* in, int K, double* out) {
    constexpr int d = 1 << NQ, D = d * d, LD = D + 1;
    const bool eigh = to == FBX_REP_CHI && from != FBX_REP_KRAUS;
    const size_t lds = (eigh ? ChoiLds<NQ>::bytes() + 16 : 0) + sizeof(cplx) * (2 * D * LD + (size_t)(K > 0 ? K : 1) * D);
    if (lds > 160 * 1024) { set_error("fbx_convert: too many Kraus operators for LDS staging"); return FBX_ERR_UNSUPPORTED; }
    auto kern = eigh ? convert_kernel<NQ, true> : convert_kernel<NQ, false>;
    FBX_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)B), dim3(64), lds, stream(), from, to, (long long)B, in, K, out);
    FBX_HIP(hipGetLastError());
    return FBX_OK;
}

// ---- three qubits: 64 x 64 matrices, one 1024-thread workgroup per item.  The two ping-pong
// matrices (row-major, LD = 64) ARE the Jacobi work / eigenvector arrays of the |C| step: every
// hand-over goes through registers, so the aliasing is safe.  LDS: [A 64K | B 64K | Kraus + scratch 32K].
__global__ void __launch_bounds__(1024)
convert3_kernel(int from, int to, long long B, const double* __restrict__ in, int K, double* __restrict__ out) {
    constexpr int NQ = 3, d = 8, D = 64, LD = 64, NT = 1024, NB = 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx* A = (cplx*)smem;
    cplx* Bm = A + D * D;
    double* lam = (double*)(Bm + D * D);
    double* red = lam + D;
    cplx* kb = (cplx*)(red + 64);
    const int t = threadIdx.x;
    const long long item = blockIdx.x;
    cplx* cur = A; cplx* nxt = Bm;
    auto swap = [&]() { cplx* q = cur; cur = nxt; nxt = q; __syncthreads(); };
    const double inv_d = 1.0 / d;

    int rep = from;
    if (from == FBX_REP_KRAUS) {
        const bool sup = (to == FBX_REP_SUPEROP || to == FBX_REP_PAULI_LIOUVILLE);
        kraus_to<NQ, NT, LD>(in + item * (long long)K * D * 2, K, sup, cur, kb, t);
        __syncthreads();
        rep = sup ? FBX_REP_SUPEROP : FBX_REP_CHOI;
    } else {
        load_matrix<NQ, NT, LD>(in + item * (long long)D * D * 2, cur, t);
        __syncthreads();
    }
    const bool kraus_chi = (from == FBX_REP_KRAUS && to == FBX_REP_CHI);
    while (rep != to) {
        if (rep == FBX_REP_CHI) {
            from_pauli_sites<NQ, NT, LD>(cur, nxt, 1.0, t); swap(); rep = FBX_REP_CHOI;
        } else if (rep == FBX_REP_CHOI) {
            if (to == FBX_REP_CHI) {
                if (!kraus_chi) {       // |C| = sum |lambda| v v^H, as choi2kraus -> kraus2chi (tol 1e-9)
                    const int I = t / NB, J = t % NB;
                    Blk h;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {       // numpy eigh reads the lower triangle
                        const int r = 2 * I + (e >> 1), c = 2 * J + (e & 1);
                        if (r > c) { const cplx v = cur[r * LD + c]; h.re[e] = v.re; h.im[e] = v.im; }
                        else if (r < c) { const cplx v = cur[c * LD + r]; h.re[e] = v.re; h.im[e] = -v.im; }
                        else { h.re[e] = cur[r * LD + c].re; h.im[e] = 0.0; }
                    }
                    __syncthreads();
                    sys_store<D>(A, t, h);
                    __syncthreads();
                    jacobi_eigh_block<D, NT>(A, Bm, t, true, red);
                    if (t < D) {
                        const double l = fabs(A[sys_index<D>(t, t)].re);
                        lam[t] = l > 1e-9 ? l : 0.0;
                    }
                    __syncthreads();
                    const Blk a = reconstruct_blk<D>(Bm, lam, t);
                    __syncthreads();
                    blk_store<D, LD>(nxt, t, a);
                    swap();
                }
                to_pauli_sites<NQ, NT, LD>(cur, nxt, inv_d * inv_d, t); swap(); rep = FBX_REP_CHI;
            } else {
                reshuffle<NQ, NT, LD>(cur, nxt, t); swap(); rep = FBX_REP_SUPEROP;
            }
        } else if (rep == FBX_REP_SUPEROP) {
            if (to == FBX_REP_PAULI_LIOUVILLE) {
                to_pauli_sites<NQ, NT, LD>(cur, nxt, inv_d, t); swap(); rep = FBX_REP_PAULI_LIOUVILLE;
            } else {
                reshuffle<NQ, NT, LD>(cur, nxt, t); swap(); rep = FBX_REP_CHOI;
            }
        } else {
            from_pauli_sites<NQ, NT, LD>(cur, nxt, inv_d, t); swap(); rep = FBX_REP_SUPEROP;
        }
    }
    store_matrix<NQ, NT, LD>(cur, out + item * (long long)D * D * 2, t);
}

// ---------------------------------------------------------------------------------------------
// 4 and 5 qubits (256 x 256 / 1024 x 1024 superoperators): the same walk through the representation graph
// with the two work matrices of an item in HBM / L2 (1 MB / 16 MB each) instead of LDS -- one 1024-thread
// workgroup per item, every primitive looped over the entries (or the quads of a butterfly stage) with a
// workgroup barrier between the stages.  Conversions INTO chi from anything but Kraus operators go through
// a D x D eigendecomposition in the reference (choi2kraus) and are not offered beyond 3 qubits.
// ---------------------------------------------------------------------------------------------
template <int NQ, bool INVERSE, int NT>
__device__ void site_stages_big(cplx* M, int t) {
    constexpr int D = 1 << (2 * NQ), NQUAD = D * D / 4;
#pragma unroll 1
    for (int q = NQ - 1; q >= 0; --q) {
        for (int u = t; u < NQUAD; u += NT) pauli_site_stage<NQ, INVERSE, D>(M, u, 3 * NQ + q, 2 * NQ + q, -1.0);
        __syncthreads();
    }
#pragma unroll 1
    for (int q = NQ - 1; q >= 0; --q) {
        for (int u = t; u < NQUAD; u += NT) pauli_site_stage<NQ, INVERSE, D>(M, u, NQ + q, q, +1.0);
        __syncthreads();
    }
}
template <int NQ, int NT>
__device__ void to_pauli_big(cplx* in, cplx* out, double scale, int t) {        // destroys `in`
    constexpr int D = 1 << (2 * NQ);
    site_stages_big<NQ, false, NT>(in, t);
    for (int idx = t; idx < D * D; idx += NT) {
        cplx v = in[site_index<NQ>(idx / D) * D + site_index<NQ>(idx % D)];
        v.re *= scale; v.im *= scale;
        out[idx] = v;
    }
}
template <int NQ, int NT>
__device__ void from_pauli_big(const cplx* in, cplx* out, double scale, int t) {
    constexpr int D = 1 << (2 * NQ);
    const double s = scale * D;
    for (int idx = t; idx < D * D; idx += NT) {
        cplx v = in[idx];
        v.re *= s; v.im *= s;
        out[site_index<NQ>(idx / D) * D + site_index<NQ>(idx % D)] = v;
    }
    __syncthreads();
    site_stages_big<NQ, true, NT>(out, t);
}

template <int NQ>
__global__ void __launch_bounds__(1024)
convert_big_kernel(int from, int to, long long B, const double* __restrict__ in, int K, double* __restrict__ out,
                   cplx* __restrict__ work) {
    constexpr int d = 1 << NQ, D = d * d, LD = D, NT = 1024;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx* kb = (cplx*)smem;                                  // the Kraus operators of the item
    const int t = threadIdx.x;
    const long long item = blockIdx.x;
    cplx* cur = work + (size_t)item * 2 * D * D;
    cplx* nxt = cur + (size_t)D * D;
    auto swap = [&]() { cplx* q = cur; cur = nxt; nxt = q; __syncthreads(); };
    const double inv_d = 1.0 / d;
    int rep = from;
    if (from == FBX_REP_KRAUS) {
        const bool sup = (to == FBX_REP_SUPEROP || to == FBX_REP_PAULI_LIOUVILLE);
        kraus_to<NQ, NT, LD>(in + item * (long long)K * D * 2, K, sup, cur, kb, t);
        __syncthreads();
        rep = sup ? FBX_REP_SUPEROP : FBX_REP_CHOI;
    } else {
        load_matrix<NQ, NT, LD>(in + item * (long long)D * D * 2, cur, t);
        __syncthreads();
    }
    while (rep != to) {
        if (rep == FBX_REP_CHI) {
            from_pauli_big<NQ, NT>(cur, nxt, 1.0, t); swap(); rep = FBX_REP_CHOI;
        } else if (rep == FBX_REP_CHOI) {
            if (to == FBX_REP_CHI) {            // only from Kraus operators (checked on the host): the Choi matrix is PSD
                to_pauli_big<NQ, NT>(cur, nxt, inv_d * inv_d, t); swap(); rep = FBX_REP_CHI;
            } else {
                reshuffle<NQ, NT, LD>(cur, nxt, t); swap(); rep = FBX_REP_SUPEROP;
            }
        } else if (rep == FBX_REP_SUPEROP) {
            if (to == FBX_REP_PAULI_LIOUVILLE) {
                to_pauli_big<NQ, NT>(cur, nxt, inv_d, t); swap(); rep = FBX_REP_PAULI_LIOUVILLE;
            } else {
                reshuffle<NQ, NT, LD>(cur, nxt, t); swap(); rep = FBX_REP_CHOI;
            }
        } else {
            from_pauli_big<NQ, NT>(cur, nxt, inv_d, t); swap(); rep = FBX_REP_SUPEROP;
        }
    }
    store_matrix<NQ, NT, LD>(cur, out + item * (long long)D * D * 2, t);
}

template <int NQ>
static int convert_into_chi_big(int from, int64_t B, const double* in, double* out);

// psd_choi: the caller vouches that the Choi matrix on the way into chi is positive semidefinite (|C| = C): the kernel's
// linear basis change then IS choi2chi (it is what kraus -> chi runs)
template <int NQ>
static int launch_convert_big(int from, int to, int64_t B, const double* in, int K, double* out, bool psd_choi = false) {
    constexpr size_t d = (size_t)1 << NQ, D = d * d;
    if (to == FBX_REP_CHI && from != FBX_REP_KRAUS && !(psd_choi && from == FBX_REP_CHOI))
        return convert_into_chi_big<NQ>(from, B, in, out);
    const size_t lds = sizeof(cplx) * (size_t)(K > 0 ? K : 1) * D;
    if (lds > 160 * 1024) { set_error("fbx_convert: too many Kraus operators for LDS staging"); return FBX_ERR_UNSUPPORTED; }
    auto kern = convert_big_kernel<NQ>;
    FBX_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const size_t per_item = 2 * D * D * sizeof(cplx);
    const int64_t chunk = (int64_t)std::max<size_t>(1, std::min<size_t>((size_t)B, ((size_t)512 << 20) / per_item));
    void* w = nullptr;
    { const int rc = workspace(WS_CONVERT, per_item * (size_t)chunk, &w); if (rc) return rc; }
    const size_t in_item = (from == FBX_REP_KRAUS ? (size_t)K * D : D * D) * 2;
    for (int64_t b0 = 0; b0 < B; b0 += chunk) {
        const int64_t nb = B - b0 < chunk ? B - b0 : chunk;
        hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(1024), lds, stream(), from, to, (long long)nb, in + b0 * in_item, K,
                           out + b0 * D * D * 2, (cplx*)w);
    }
    FBX_HIP(hipGetLastError());
    return FBX_OK;
}

// Into chi from a Choi / superoperator / Pauli-Liouville matrix for 4 and 5 qubits (round 5).  The reference goes through
// choi2kraus -> kraus2chi (superoperator_transformations.py:241-250, 291-298, 339-348): chi of |C| = sum |lambda_i| v_i v_i^H
// over the eigenpairs with |lambda_i| > 1e-9.  Composed from the library's own primitives, everything resident: the walk to
// the Choi matrix, fbx_eigh_dev (the HBM-resident Jacobi: 25 ms per 256 x 256 matrix, 0.8 s per 1024 x 1024), |lambda| with
// the reference's cut, fbx_matmul_dev for V diag(|lambda|) V^H, and the kernel's linear basis change on that PSD matrix.
__global__ void __launch_bounds__(256) abs_cut_kernel(double* __restrict__ w, long long n, double tol) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { const double a = fabs(w[i]); w[i] = a > tol ? a : 0.0; }
}
template <int NQ>
static int convert_into_chi_big(int from, int64_t B, const double* in, double* out) {
    constexpr size_t D = (size_t)1 << (2 * NQ);
    const int64_t chunk = (int64_t)std::max<size_t>(1, std::min<size_t>((size_t)B, ((size_t)256 << 20) / (D * D * sizeof(cplx))));
    DevBuf c, v, w;
    { int rc; if ((rc = c.alloc(D * D * sizeof(cplx) * chunk)) || (rc = v.alloc(D * D * sizeof(cplx) * chunk)) || (rc = w.alloc(D * sizeof(double) * chunk))) return rc; }
    for (int64_t b0 = 0; b0 < B; b0 += chunk) {
        const int64_t nb = B - b0 < chunk ? B - b0 : chunk;
        const double* choi = in + (size_t)b0 * D * D * 2;
        if (from != FBX_REP_CHOI) { const int rc = launch_convert_big<NQ>(from, FBX_REP_CHOI, nb, choi, 0, c.as<double>()); if (rc) return rc; choi = c.as<double>(); }
        { const int rc = fbx_eigh_dev((int)D, nb, choi, w.as<double>(), v.as<double>()); if (rc) return rc; }
        hipLaunchKernelGGL(abs_cut_kernel, dim3((unsigned)((nb * D + 255) / 256)), dim3(256), 0, stream(), w.as<double>(), (long long)(nb * D), 1e-9);
        FBX_HIP(hipGetLastError());
        { const int rc = fbx_matmul_dev((int)D, nb, v.as<double>(), 0, w.as<double>(), v.as<double>(), 1, c.as<double>()); if (rc) return rc; }
        { const int rc = launch_convert_big<NQ>(FBX_REP_CHOI, FBX_REP_CHI, nb, c.as<double>(), 0, out + (size_t)b0 * D * D * 2, true); if (rc) return rc; }
    }
    return FBX_OK;
}

// ---------------------------------------------------------------------------------------------
// Any Hilbert-space dimension (qutrits, ...): the conversions that involve no operator basis --
// kraus2superop, kraus2choi, superop2choi, choi2superop (superoperator_transformations.py:100-182,267-277,351-361).
// One thread per output entry, straight from and to HBM.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
convert_general_kernel(int from, int to, int d, long long B, const double* __restrict__ in, int K, double* __restrict__ out) {
    const long long D = (long long)d * d, DD = D * D;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= B * DD) return;
    const long long item = gid / DD;
    const int row = (int)((gid % DD) / D), col = (int)(gid % D);
    double re = 0.0, im = 0.0;
    if (from == FBX_REP_KRAUS) {
        const double* k0 = in + item * K * D * 2;
        for (int t = 0; t < K; ++t) {
            const double* k = k0 + (long long)t * D * 2;
            int ia, ib;
            double sa = 1.0, sb = 1.0;
            if (to == FBX_REP_SUPEROP) { ia = (row / d) * d + col / d; sa = -1.0; ib = (row % d) * d + col % d; }   // conj(K[i][j]) K[k][l]
            else { ia = (row % d) * d + row / d; ib = (col % d) * d + col / d; sb = -1.0; }                         // vK[row] conj(vK[col])
            const double ar = k[2 * ia], ai = sa * k[2 * ia + 1], br = k[2 * ib], bi = sb * k[2 * ib + 1];
            re += ar * br - ai * bi; im += ar * bi + ai * br;
        }
    } else {        // the reshuffle, its own inverse: out[(p,q)][(r,s)] = in[(s,q)][(r,p)]
        const int p = row / d, q = row % d, r = col / d, s = col % d;
        const double* src = in + (item * DD + (long long)(s * d + q) * D + r * d + p) * 2;
        re = src[0]; im = src[1];
    }
    out[2 * gid] = re; out[2 * gid + 1] = im;
}

// ---- 3 qubits, the routes between Choi / superoperator / Pauli-Liouville: ONE 64 KB matrix in LDS instead of two,
// so that two workgroups share a CU and the HBM loads / stores of one overlap the butterfly stages of the other.
// The reshuffle rides on the global load (forward) or store (backward) as an index permutation, the bit-permuting
// copy of the site-factored transform on the other side.  ROUTE: 0 choi->PL, 1 PL->choi, 2 superop->PL, 3 PL->superop,
// 4 choi<->superop (pure permutation, no LDS).
template <int ROUTE>
__global__ void __launch_bounds__(1024)
convert3_fast_kernel(long long B, const double* __restrict__ in, double* __restrict__ out) {
    constexpr int NQ = 3, d = 8, D = 64, LD = 64, NT = 1024;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx* X = (cplx*)smem;
    const int t = threadIdx.x;
    const double inv_d = 1.0 / d;
    for (long long item = blockIdx.x; item < B; item += gridDim.x) {
        const double* src = in + item * (long long)D * D * 2;
        double* dst = out + item * (long long)D * D * 2;
        auto shuffled = [](int idx) {                      // entry (p,q),(r,s) <- entry (s,q),(r,p): its own inverse
            const int row = idx / D, col = idx % D;
            const int p = row / d, q = row % d, r = col / d, s_ = col % d;
            return (s_ * d + q) * D + r * d + p;
        };
        if (ROUTE == 4) {
            for (int idx = t; idx < D * D; idx += NT) { const int j = shuffled(idx); dst[2 * idx] = src[2 * j]; dst[2 * idx + 1] = src[2 * j + 1]; }
            continue;
        }
        __syncthreads();                                   // the previous item's readers of X are done
        if (ROUTE == 0 || ROUTE == 2) {                    // -> Pauli-Liouville: (reshuffled) load, stages, permuted scaled store
            for (int idx = t; idx < D * D; idx += NT) {
                const int j = ROUTE == 0 ? shuffled(idx) : idx;
                cplx v; v.re = src[2 * j]; v.im = src[2 * j + 1];
                X[idx] = v;
            }
            __syncthreads();
            site_stages<NQ, false, NT, LD>(X, t);
            for (int idx = t; idx < D * D; idx += NT) {
                const cplx v = X[site_index<NQ>(idx / D) * LD + site_index<NQ>(idx % D)];
                dst[2 * idx] = v.re * inv_d; dst[2 * idx + 1] = v.im * inv_d;
            }
        } else {                                           // Pauli-Liouville ->: permuted scaled load, inverse stages, (reshuffled) store
            const double sc = inv_d * D;
            for (int idx = t; idx < D * D; idx += NT) {
                cplx v; v.re = src[2 * idx] * sc; v.im = src[2 * idx + 1] * sc;
                X[site_index<NQ>(idx / D) * LD + site_index<NQ>(idx % D)] = v;
            }
            __syncthreads();
            site_stages<NQ, true, NT, LD>(X, t);
            for (int idx = t; idx < D * D; idx += NT) {
                const cplx v = X[ROUTE == 1 ? shuffled(idx) : idx];
                dst[2 * idx] = v.re; dst[2 * idx + 1] = v.im;
            }
        }
    }
}
template <int ROUTE>
static int launch_convert3_fast(int64_t B, const double* in, double* out) {
    const size_t lds = ROUTE == 4 ? 0 : sizeof(cplx) * 64 * 64;
    auto kern = convert3_fast_kernel<ROUTE>;
    FBX_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const unsigned grid = (unsigned)(B < 512 * 8 ? B : 512 * 8);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), lds, stream(), (long long)B, in, out);
    FBX_HIP(hipGetLastError());
    return FBX_OK;
}

template <int ROUTE> static int launch_convert3_regs(int64_t B, const double* in, double* out);     // further down: two stages per pass in registers
static int launch_sweep3_regs(int64_t B, int K, const double* kraus, const double* ptm_ref, double* choi, double* ptm, double* chi, double* fid);

static int launch_convert3(int from, int to, int64_t B, const double* in, int K, double* out) {
    constexpr size_t D = 64;
    const char* v1s = getenv("FBX_CONVERT3_V1");            // 1 = the one-stage-per-pass kernels (A/B, tests)
    const bool v1 = v1s && atoi(v1s) != 0;
    // from Kraus operators: the sweep kernel with one output (operators read once, the result written once, coalesced)
    if (!v1 && from == FBX_REP_KRAUS && K >= 1 && K <= 31 && (to == FBX_REP_CHOI || to == FBX_REP_PAULI_LIOUVILLE || to == FBX_REP_CHI))
        return launch_sweep3_regs(B, K, in, nullptr, to == FBX_REP_CHOI ? out : nullptr, to == FBX_REP_PAULI_LIOUVILLE ? out : nullptr,
                                  to == FBX_REP_CHI ? out : nullptr, nullptr);
    if (from == FBX_REP_CHOI && to == FBX_REP_PAULI_LIOUVILLE) return v1 ? launch_convert3_fast<0>(B, in, out) : launch_convert3_regs<0>(B, in, out);
    if (from == FBX_REP_PAULI_LIOUVILLE && to == FBX_REP_CHOI) return launch_convert3_fast<1>(B, in, out);
    if (from == FBX_REP_SUPEROP && to == FBX_REP_PAULI_LIOUVILLE) return v1 ? launch_convert3_fast<2>(B, in, out) : launch_convert3_regs<2>(B, in, out);
    if (from == FBX_REP_PAULI_LIOUVILLE && to == FBX_REP_SUPEROP) return v1 ? launch_convert3_fast<3>(B, in, out) : launch_convert3_regs<3>(B, in, out);
    if ((from == FBX_REP_CHOI && to == FBX_REP_SUPEROP) || (from == FBX_REP_SUPEROP && to == FBX_REP_CHOI)) return launch_convert3_fast<4>(B, in, out);
    const size_t lds = 2 * sizeof(cplx) * D * D + sizeof(double) * 128 + sizeof(cplx) * (size_t)(K > 0 ? K : 1) * D;
    if (lds > 160 * 1024) { set_error("fbx_convert: too many Kraus operators for LDS staging (3 qubits: at most 31)"); return FBX_ERR_UNSUPPORTED; }
    FBX_HIP(hipFuncSetAttribute((const void*)convert3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(convert3_kernel, dim3((unsigned)B), dim3(1024), lds, stream(), from, to, (long long)B, in, K, out);
    FBX_HIP(hipGetLastError());
    return FBX_OK;
}

// results of the sweep are written once and never re-read by the kernel: stream them past the caches
typedef double fbx_d2v __attribute__((ext_vector_type(2)));
#define FBX_STREAM_STORE(ptr, val) __builtin_nontemporal_store(fbx_d2v{(val).x, (val).y}, reinterpret_cast<fbx_d2v*>(ptr))

// ---------------------------------------------------------------------------------------------
// fused Kraus sweep (BASELINE config 3)
// ---------------------------------------------------------------------------------------------
template <int NQ>
__global__ void __launch_bounds__(64)
sweep_kernel(long long B, int K, const double* __restrict__ kraus, const double* __restrict__ ptm_ref,
             double* __restrict__ choi_out, double* __restrict__ ptm_out, double* __restrict__ chi_out,
             double* __restrict__ fid_out) {
    constexpr int d = 1 << NQ, D = d * d, LD = D + 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx* C = (cplx*)smem;                 // Choi
    cplx* S = C + D * LD;                  // superop, then chi
    cplx* P = S + D * LD;                  // Pauli-Liouville
    cplx* R = P + D * LD;                  // reference PTM
    cplx* kb = R + D * LD;
    const int lane = threadIdx.x;
    if (ptm_ref) load_matrix<NQ>(ptm_ref, R, lane);
    const double inv_d = 1.0 / d;
    for (long long item = blockIdx.x; item < B; item += gridDim.x) {
        __syncthreads();
        kraus_to<NQ>(kraus + item * (long long)K * D * 2, K, false, C, kb, lane);
        __syncthreads();
        if (choi_out) store_matrix<NQ>(C, choi_out + item * (long long)D * D * 2, lane);
        reshuffle<NQ>(C, S, lane);
        __syncthreads();
        to_pauli_basis<NQ>(S, P, inv_d, lane);
        __syncthreads();
        if (ptm_out) store_matrix<NQ>(P, ptm_out + item * (long long)D * D * 2, lane);
        if (fid_out && ptm_ref) {          // process_fidelity(ref, ptm): (d Fe + 1)/(d + 1), Fe = tr(ref^H ptm)/d^2
            double acc = 0.0;
            for (int idx = lane; idx < D * D; idx += 64) {
                const cplx a = R[(idx / D) * LD + idx % D], b = P[(idx / D) * LD + idx % D];
                acc += a.re * b.re + a.im * b.im;
            }
            acc = wave_sum(acc);
            if (lane == 0) fid_out[item] = (d * (acc / (double)(d * d)) + 1.0) / (d + 1.0);
        }
        if (chi_out) {                      // a Kraus set is CP: chi = c2p Choi c2p^H (= kraus2chi)
            to_pauli_basis<NQ>(C, S, inv_d * inv_d, lane);
            __syncthreads();
            store_matrix<NQ>(S, chi_out + item * (long long)D * D * 2, lane);
        }
    }
}


// ---------------------------------------------------------------------------------------------
// sweep2q_pair_kernel: the same pipeline with HALF the LDS traffic (round 1's one-item-per-wavefront kernel was LDS-bandwidth
// bound: ~100 KB per item).  A wavefront takes TWO Kraus sets; 16 lanes own one 16 x 16 matrix (A ->
// Pauli-Liouville and W -> chi of each item), 16 elements per lane, so that TWO butterfly stages run in
// registers per pass and one LDS transpose separates the two passes.  The passes are ordered so that the
// final registers of a lane are one column (PTM) / one column (chi) of the output in matrix order: the
// results go from registers to HBM in 256-byte runs, no gather through LDS.  The process fidelity uses
// tr(R_ref^H R) = tr(E_ref^H E) (the Pauli transform is unitary up to the factor d), so it is reduced from
// the Choi accumulators against the Choi form of the reference, before any transform.
// Element index = row * 16 + col (8 bits); lane-group roles: (lane >> 5) = item of the pair,
// (lane >> 4) & 1 = 0: A, 1: W; within the group, 4 index bits come from the lane and 4 from the register.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ constexpr int dep4(int v, int b3, int b2, int b1, int b0) {
    return (((v >> 3) & 1) << b3) | (((v >> 2) & 1) << b2) | (((v >> 1) & 1) << b1) | ((v & 1) << b0);
}
__device__ __forceinline__ constexpr int padded(int idx) { return idx + (idx >> 4); }     // (idx >> 4) * 17 + (idx & 15)

// two sites on the 16 registers of a lane: register index r = (p1 q1 p2 q2)
__device__ __forceinline__ void two_sites(cplx (&x)[16], double y1, double y2) {
    auto site = [](cplx& c00, cplx& c11, cplx& c01, cplx& c10, double ys) {
        cplx oi, oz, ox, oy;
        oi.re = c00.re + c11.re; oi.im = c00.im + c11.im;
        oz.re = c00.re - c11.re; oz.im = c00.im - c11.im;
        ox.re = c01.re + c10.re; ox.im = c01.im + c10.im;
        const double dr = c01.re - c10.re, di = c01.im - c10.im;
        oy.re = -ys * di; oy.im = ys * dr;
        c00 = oi; c11 = oz; c01 = ox; c10 = oy;
    };
#pragma unroll
    for (int cd = 0; cd < 4; ++cd) site(x[cd], x[12 | cd], x[4 | cd], x[8 | cd], y1);            // p1 = bit 3, q1 = bit 2
#pragma unroll
    for (int ab = 0; ab < 4; ++ab) site(x[ab << 2], x[(ab << 2) | 3], x[(ab << 2) | 1], x[(ab << 2) | 2], y2);   // p2 = bit 1, q2 = bit 0
}

__global__ void __launch_bounds__(64)
sweep2q_pair_kernel(long long B, int K, const double* __restrict__ kraus, const double* __restrict__ choi_ref,
                    double* __restrict__ choi_out, double* __restrict__ ptm_out, double* __restrict__ chi_out,
                    double* __restrict__ fid_out) {
    constexpr int D = 16, MAT = 16 * 17;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx* bufA = (cplx*)smem;              // [2][MAT] Choi of the item, then the A transpose
    cplx* bufW = bufA + 2 * MAT;           // [2][MAT] W transpose
    cplx* kbs = bufW + 2 * MAT;            // [2][K * 16] vec of the Kraus operators
    const int lane = threadIdx.x;
    const int h = lane >> 5, u = lane & 31, w = (lane >> 4) & 1, l = lane & 15;
    const int col = u & 15, row0 = (u >> 4) * 8;           // kraus2choi: this lane owns C[row0 .. row0 + 7][col]
    // LDS addresses of the 16 registers in the two passes (additive: lane part + register part, no carries)
    const int la1 = padded(w ? dep4(l, 7, 6, 5, 4) : dep4(l, 5, 4, 1, 0));
    const int la2 = padded(w ? dep4(l, 3, 2, 1, 0) : dep4(l, 7, 6, 3, 2));
    const int jcol = ((l >> 3) & 1) << 3 | ((l >> 1) & 1) << 2 | ((l >> 2) & 1) << 1 | (l & 1);   // output column of this lane
    cplx* mine = (w ? bufW : bufA) + h * MAT;              // where this lane's matrix is transposed
    const cplx* src = bufA + h * MAT;                      // the item's Choi matrix
    cplx* kb = kbs + h * K * D;
    // reference in Choi form, in the kraus2choi layout
    cplx ref[8];
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
        ref[rr].re = ref[rr].im = 0.0;
        if (choi_ref) { const double* q = choi_ref + 2 * ((row0 + rr) * D + col); ref[rr].re = q[0]; ref[rr].im = q[1]; }
    }
#define FBX_WAVE_FENCE() asm volatile("" ::: "memory")
    const int n_ld = (K * D + 31) / 32;                    // 16-byte loads per lane and item (K <= 16: at most 8)
    const long long n_pairs = (B + 1) / 2;
    double2 nxt[8];
    auto fetch = [&](long long pair) {
        const long long item = 2 * pair + h;
#pragma unroll
        for (int t8 = 0; t8 < 8; ++t8) {
            const int idx = u + 32 * t8;
            if (t8 < n_ld && idx < K * D && item < B)
                nxt[t8] = *reinterpret_cast<const double2*>(kraus + (item * (long long)K * D + idx) * 2);
        }
    };
#pragma unroll
    for (int t8 = 0; t8 < 8; ++t8) nxt[t8].x = nxt[t8].y = 0.0;
    if ((long long)blockIdx.x < n_pairs) fetch(blockIdx.x);
    for (long long pair = blockIdx.x; pair < n_pairs; pair += gridDim.x) {
        const long long item = 2 * pair + h;
        const bool live = item < B;
        // ---- Kraus operators -> LDS as vec(K_t)[c * 4 + r] = K_t[r][c]; next pair's operators from HBM meanwhile
#pragma unroll
        for (int t8 = 0; t8 < 8; ++t8) {
            const int idx = u + 32 * t8;
            if (t8 < n_ld && idx < K * D) {
                const int t = idx >> 4, rr = (idx >> 2) & 3, cc = idx & 3;
                cplx c; c.re = nxt[t8].x; c.im = nxt[t8].y;
                kb[t * D + cc * 4 + rr] = c;
            }
        }
        if (pair + gridDim.x < n_pairs) fetch(pair + gridDim.x);
        FBX_WAVE_FENCE();
        // ---- kraus2choi: C[row][col] = sum_t vK_t[row] conj(vK_t[col])
        cplx acc[8];
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) acc[rr].re = acc[rr].im = 0.0;
        for (int t = 0; t < K; ++t) {
            const cplx b = kb[t * D + col];
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                const cplx a = kb[t * D + row0 + rr];
                acc[rr].re += a.re * b.re + a.im * b.im;
                acc[rr].im += a.im * b.re - a.re * b.im;
            }
        }
        double fr = 0.0;
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            bufA[h * MAT + (row0 + rr) * 17 + col] = acc[rr];
            fr += ref[rr].re * acc[rr].re + ref[rr].im * acc[rr].im;
            if (choi_out && live) {
                double2 v; v.x = acc[rr].re; v.y = acc[rr].im;
                FBX_STREAM_STORE(reinterpret_cast<double2*>(choi_out + (item * D * D + (row0 + rr) * D + col) * 2), v);
            }
        }
        if (fid_out) {                                     // sum over the 32 lanes of the item
            fr += dpp_permute<0xB1>(fr); fr += dpp_permute<0x4E>(fr);
            fr += dpp_permute<0x141>(fr); fr += dpp_permute<0x140>(fr);
            const double tot = readlane_f64(fr, 0) + readlane_f64(fr, 16), tot1 = readlane_f64(fr, 32) + readlane_f64(fr, 48);
            if (u == 0 && live) fid_out[item] = (4.0 * ((h ? tot1 : tot) / 16.0) + 1.0) / 5.0;
        }
        FBX_WAVE_FENCE();
        // ---- pass 1: A sites (7,3),(6,2) [-i: input qubits]; W sites (3,1),(2,0) [+i]
        cplx x[16];
#pragma unroll
        for (int r = 0; r < 16; ++r)
            x[r] = src[la1 + (w ? padded(dep4(r, 3, 1, 2, 0)) : padded(dep4(r, 7, 3, 6, 2)))];
        two_sites(x, w ? +1.0 : -1.0, w ? +1.0 : -1.0);
#pragma unroll
        for (int r = 0; r < 16; ++r)
            mine[la1 + (w ? padded(dep4(r, 3, 1, 2, 0)) : padded(dep4(r, 7, 3, 6, 2)))] = x[r];
        FBX_WAVE_FENCE();
        // ---- pass 2: A sites (5,1),(4,0) [+i: output qubits]; W sites (7,5),(6,4) [-i]
#pragma unroll
        for (int r = 0; r < 16; ++r)
            x[r] = mine[la2 + (w ? padded(dep4(r, 7, 5, 6, 4)) : padded(dep4(r, 5, 1, 4, 0)))];
        two_sites(x, w ? -1.0 : +1.0, w ? -1.0 : +1.0);
        // ---- register r = output row i, lane = output column jcol: 256-byte runs straight to HBM
        double* dst = w ? chi_out : ptm_out;
        const double scale = w ? 0.0625 : 0.25;
        if (dst && live) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                double2 o; o.x = x[r].re * scale; o.y = x[r].im * scale;
                FBX_STREAM_STORE(reinterpret_cast<double2*>(dst + (item * D * D + r * D + jcol) * 2), o);
            }
        }
        FBX_WAVE_FENCE();
    }
#undef FBX_WAVE_FENCE
}

// persistent grid of the 2-qubit sweep: 8 wavefronts resident per CU, the rest queued
#define FBX_SWEEP_GRID (256 * 16)
static_assert(FBX_SWEEP_GRID > 0, "FBX_SWEEP_GRID must be positive");

template <int NQ>
static int launch_sweep(int64_t B, int K, const double* kraus, const double* ptm_ref, double* choi,
                        double* ptm, double* chi, double* fid) {
    constexpr int d = 1 << NQ, D = d * d, LD = D + 1;
    const size_t lds = sizeof(cplx) * (4 * D * LD + (size_t)K * D);
    if (lds > 160 * 1024) { set_error("fbx_kraus_sweep: too many Kraus operators"); return FBX_ERR_UNSUPPORTED; }
    if (NQ == 2 && K <= 16) {
        // reference in Choi form for the on-the-fly fidelity (one 16 x 16 conversion per call, into a
        // workspace of the calling thread)
        double* choi_ref = nullptr;
        if (ptm_ref) {
            void* w = nullptr;
            { const int rc = workspace(WS_SWEEP_REF, sizeof(cplx) * 256, &w); if (rc) return rc; }
            choi_ref = (double*)w;
            { const int rc = launch_convert<2>(FBX_REP_PAULI_LIOUVILLE, FBX_REP_CHOI, 1, ptm_ref, 0, choi_ref); if (rc) return rc; }
        }
        const size_t ldsp = sizeof(cplx) * (4 * 16 * 17 + 2 * (size_t)K * 16);
        const long long n_pairs = (B + 1) / 2;
        const long long cap = FBX_SWEEP_GRID;                     // 8 wavefronts resident per CU, the rest queued
        const unsigned gridp = (unsigned)(n_pairs < cap ? n_pairs : cap);
        hipLaunchKernelGGL(sweep2q_pair_kernel, dim3(gridp), dim3(64), ldsp, stream(), (long long)B, K, kraus,
                           (const double*)choi_ref, choi, ptm, chi, fid);
        FBX_HIP(hipGetLastError());
        return FBX_OK;
    }
    auto kern = sweep_kernel<NQ>;
    FBX_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const unsigned grid = (unsigned)(B < 256 * 8 ? B : 256 * 8);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64), lds, stream(), (long long)B, K, kraus, ptm_ref, choi, ptm, chi, fid);
    FBX_HIP(hipGetLastError());
    return FBX_OK;
}

// ---------------------------------------------------------------------------------------------
// fbx_proj_choi
// ---------------------------------------------------------------------------------------------
template <int NQ>
__device__ __forceinline__ void proj_choi_body(char* smem, int kind, long long B, const double* __restrict__ in,
                                               double* __restrict__ out, int* __restrict__ iters_out) {
    constexpr int d = 1 << NQ, D = d * d, LD = D + 1;
    char* p = smem;
    ChoiLds<NQ> L; L.carve(p);
    PhaseClock pc; pc.reset(); L.pc = &pc;
    const int lane = threadIdx.x;
    const long long item = blockIdx.x;
    load_matrix<NQ>(in + item * (long long)D * D * 2, L.Mw, lane);
    __syncthreads();
    const Blk x = blk_load<D, LD>(L.Mw, lane);
    __syncthreads();
    int iters = 0, sweeps = 0;
    Blk y;
    if (kind == FBX_PROJ_CP) y = proj_cp_blk<NQ>(x, L, lane, sweeps);
    else if (kind == FBX_PROJ_TP) y = proj_tp_blk<NQ>(x, L, lane);
    else if (kind == FBX_PROJ_TNI) y = proj_tni_blk<NQ>(x, L, lane, sweeps);
    else y = proj_physical_blk<NQ>(x, kind == FBX_PROJ_PHYSICAL_TP, L, lane, iters, sweeps);
    __syncthreads();
    blk_store<D, LD>(L.Mw, lane, y);
    __syncthreads();
    store_matrix<NQ>(L.Mw, out + item * (long long)D * D * 2, lane);
    if (lane == 0 && iters_out) iters_out[item] = iters;
}
// one wavefront per SIMD (every register the Dykstra state wants: 324 for two qubits) ...
template <int NQ>
__global__ void __launch_bounds__(64)
proj_choi_kernel(int kind, long long B, const double* __restrict__ in, double* __restrict__ out, int* __restrict__ iters_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    proj_choi_body<NQ>(smem, kind, B, in, out, iters_out);
}
// ... or two per SIMD (at most 256 registers) for batches that put several items on a SIMD anyway: two dependent
// Jacobi chains interleave, as in pgdb_lean_kernel.  Same arithmetic, same results.
template <int NQ>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
proj_choi_w2_kernel(int kind, long long B, const double* __restrict__ in, double* __restrict__ out, int* __restrict__ iters_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    proj_choi_body<NQ>(smem, kind, B, in, out, iters_out);
}

// ---------------------------------------------------------------------------------------------
// apply_choi_matrix_2_state: out[o][o'] = sum_{i,i'} rho[i'][i] C[(i',o)][(i,o')]
// ---------------------------------------------------------------------------------------------
__global__ void apply_choi_kernel(int d, long long B, const double* __restrict__ choi,
                                  const double* __restrict__ rho, double* __restrict__ out) {
    const long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    const int dd = d * d, D = dd;
    if (t >= B * dd) return;
    const long long item = t / dd;
    const int o = (int)(t % dd) / d, op = (int)(t % dd) % d;
    const double* C = choi + item * (long long)D * D * 2;
    const double* r = rho + item * (long long)dd * 2;
    double re = 0.0, im = 0.0;
    for (int ip = 0; ip < d; ++ip)
        for (int i = 0; i < d; ++i) {
            const double ar = r[2 * (ip * d + i)], ai = r[2 * (ip * d + i) + 1];
            const long long ci = ((long long)(ip * d + o) * D + (i * d + op)) * 2;
            const double br = C[ci], bi = C[ci + 1];
            re += ar * br - ai * bi; im += ar * bi + ai * br;
        }
    out[t * 2] = re; out[t * 2 + 1] = im;
}

// ---------------------------------------------------------------------------------------------
// entanglement / process fidelity: Fe = Re tr(A^H B) / d^2 ; Fp = (d Fe + 1) / (d + 1)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
process_fidelity_kernel(int d, long long B, const double* __restrict__ a, int a_batched, const double* __restrict__ b,
                        double* __restrict__ fe_out, double* __restrict__ fp_out) {
    const int lane = threadIdx.x;
    const int DD = d * d * d * d;
    for (long long item = blockIdx.x; item < B; item += gridDim.x) {
        const double* pa = a + (a_batched ? item : 0) * (long long)DD * 2;      // one shared reference or one per item
        const double* pb = b + item * (long long)DD * 2;
        double acc = 0.0;
        for (int idx = lane; idx < 2 * DD; idx += 64) acc += pa[idx] * pb[idx];
        acc = wave_sum(acc);
        if (lane == 0) {
            const double fe = acc / (double)(d * d);
            if (fe_out) fe_out[item] = fe;
            if (fp_out) fp_out[item] = (d * fe + 1.0) / (d + 1.0);
        }
    }
}

// Three qubits, fused (round 4): one 1024-thread workgroup walks its items through ONE 64 x 64 LDS matrix (68 KB with the
// Kraus operators: two workgroups per CU), kraus2superop -> six in-place butterfly stages -> Pauli-Liouville matrix out with the
// process fidelity reduced on the way, then kraus2choi -> Choi out -> the same stages -> chi out (a Kraus set is CP, so
// chi = c2p Choi c2p^H exactly as kraus2chi, superoperator_transformations.py:82-98).  The operators are read once (K KB), the
// three 64 KB results written once, coalesced 16 bytes per thread: 4 x 1024 + 3 x 65 536 + 8 algorithmic bytes per item for
// K = 4.  Replaces the composition of three general 64 x 64 conversions + a fidelity kernel behind fbx_kraus_sweep (each of
// which re-read the operators and kept two matrices in LDS).  Reference: superoperator_transformations.py:100-182, 339-371;
// distance_measures.py:315-360.
__global__ void __launch_bounds__(1024)
sweep3_kernel(long long B, int K, const double* __restrict__ kraus, const double* __restrict__ ptm_ref,
              double* __restrict__ choi_out, double* __restrict__ ptm_out, double* __restrict__ chi_out,
              double* __restrict__ fid_out) {
    constexpr int NQ = 3, d = 8, D = 64, LD = 64, NT = 1024;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx* X = (cplx*)smem;
    double* red = (double*)(X + D * D);
    cplx* kb = (cplx*)(red + 16);
    const int t = threadIdx.x;
    const double inv_d = 1.0 / d;
    for (long long item = blockIdx.x; item < B; item += gridDim.x) {
        __syncthreads();                                   // the previous item's readers of X / kb / red are done
        const double* kr = kraus + item * (long long)K * D * 2;
        for (int idx = t; idx < K * D; idx += NT) { kb[idx].re = kr[2 * idx]; kb[idx].im = kr[2 * idx + 1]; }
        __syncthreads();
        if (ptm_out || fid_out) {
            for (int idx = t; idx < D * D; idx += NT) {    // kron(conj(K), K)[(i,k)][(j,l)] = conj(K[i][j]) K[k][l]
                const int row = idx / D, col = idx % D;
                const int i = row / d, k = row % d, j = col / d, l = col % d;
                double re = 0.0, im = 0.0;
                for (int q = 0; q < K; ++q) {
                    const cplx a = kb[q * D + i * d + j], b = kb[q * D + k * d + l];
                    re += a.re * b.re + a.im * b.im;
                    im += a.re * b.im - a.im * b.re;
                }
                cplx o; o.re = re; o.im = im;
                X[idx] = o;
            }
            __syncthreads();
            site_stages<NQ, false, NT, LD>(X, t);
            double acc = 0.0;
            double* dst = ptm_out ? ptm_out + item * (long long)D * D * 2 : nullptr;
            for (int idx = t; idx < D * D; idx += NT) {
                cplx v = X[site_index<NQ>(idx / D) * LD + site_index<NQ>(idx % D)];
                v.re *= inv_d; v.im *= inv_d;
                if (dst) { dst[2 * idx] = v.re; dst[2 * idx + 1] = v.im; }
                if (fid_out) acc += ptm_ref[2 * idx] * v.re + ptm_ref[2 * idx + 1] * v.im;
            }
            if (fid_out) {
                acc = block_sum<NT>(acc, red);
                if (t == 0) fid_out[item] = (d * (acc / (double)(d * d)) + 1.0) / (d + 1.0);
            }
            __syncthreads();                               // X is rebuilt below
        }
        if (choi_out || chi_out) {
            double* dst = choi_out ? choi_out + item * (long long)D * D * 2 : nullptr;
            for (int idx = t; idx < D * D; idx += NT) {    // vec(K)[c d + r] = K[r][c]; choi[row][col] = vK[row] conj(vK[col])
                const int row = idx / D, col = idx % D;
                double re = 0.0, im = 0.0;
                for (int q = 0; q < K; ++q) {
                    const cplx a = kb[q * D + (row % d) * d + row / d], b = kb[q * D + (col % d) * d + col / d];
                    re += a.re * b.re + a.im * b.im;
                    im += a.im * b.re - a.re * b.im;
                }
                if (dst) { dst[2 * idx] = re; dst[2 * idx + 1] = im; }
                cplx o; o.re = re; o.im = im;
                X[idx] = o;
            }
            if (chi_out) {
                __syncthreads();
                site_stages<NQ, false, NT, LD>(X, t);
                double* cx = chi_out + item * (long long)D * D * 2;
                const double sc = inv_d * inv_d;
                for (int idx = t; idx < D * D; idx += NT) {
                    const cplx v = X[site_index<NQ>(idx / D) * LD + site_index<NQ>(idx % D)];
                    cx[2 * idx] = v.re * sc; cx[2 * idx + 1] = v.im * sc;
                }
            }
        }
    }
}

// Three qubits, fused, two butterfly stages per pass in REGISTERS (round 4, second form; the kernel above stays as the A/B and
// as the form for FBX_SWEEP3_V1=1).  The first form did one stage per pass through LDS: per transform and thread 28 ds_write_b128
// (13 cycles each) + 60 ds_read_b128, 19 k LDS-pipe cycles per item against 15 k cycles of HBM time per item and CU -- LDS-bound
// (3.5-3.8 TB/s, 31 % bank conflicts).  Here a workgroup is 256 threads, a thread holds a 4 x 4 sub-tile (16 entries = two bit
// pairs of the 12-bit element index, as the 2-qubit kernel's two_sites) and the six stages are three passes:
//   P1  row site 2 + column site 2   registers = row bits {5,2} x column bits {5,2}: the tile is BUILT here from the Kraus operators
//                                    (4 + 4 operator entries per Kraus operator for 16 products), the Choi matrix leaves from here
//   P2  column sites 1 and 0         registers = column bits {4,1,3,0}, in place
//   P3  row sites 1 and 0            registers = row bits {4,1,3,0} = the output row's low four bits, lanes = the 64 output
//                                    columns: every store instruction of a wavefront writes one whole 1 KB output row
// Two LDS round trips (32 writes + 32 reads of 16 B per thread) + 32 broadcast reads of the operators: 5 k LDS-pipe cycles per
// item.  Layout X[row][col ^ g(row)], g(row) = r1 | r3 << 1 | r0 << 2 | r4 << 3, with the thread bits of every pass assigned so
// that each ds_write_b128 lane group (8 contiguous lanes, 128-B bank period) and each ds_read_b128 lane group (the four
// non-contiguous 16-lane groups of MI355X_MICROARCH.md, 256-B period) touches distinct 16-byte slots:
//   P1 threads  l0 l1 l2 l3 l4 l5 w0 w1 -> c0 c1 r0 c3 c4 r1 r3 r4
//   P2 threads                          -> c2 r1 r3 c5 r4 r0 r2 r5
//   P3 threads                          -> c0 c3 c1 c4 c2 c5 r2 r5   (lane = output column: column bit t = l bit 2t, 3 + t = 2t + 1)
// The stages commute (each acts on its own pair of index bits), so the grouping by site changes rounding only.
__device__ __forceinline__ int s3_swz(int row) { return ((row >> 1) & 1) | (((row >> 3) & 1) << 1) | ((row & 1) << 2) | (((row >> 4) & 1) << 3); }
__device__ __forceinline__ int s3_addr(int row, int col) { return row * 64 + (col ^ s3_swz(row)); }

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
sweep3_regs_kernel(long long B, int K, const double* __restrict__ kraus, const double* __restrict__ ptm_ref,
                   double* __restrict__ choi_out, double* __restrict__ ptm_out, double* __restrict__ chi_out,
                   double* __restrict__ fid_out) {
    constexpr int d = 8, D = 64, NT = 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx* X = (cplx*)smem;
    double* red = (double*)(X + D * D);
    cplx* kb = (cplx*)(red + 16);
    const int t = threadIdx.x;
    auto bit = [](int v, int b) { return (v >> b) & 1; };
    // thread bits -> element bits of the three passes (see above); the register part is added per register below
    const int w0 = bit(t, 6), w1 = bit(t, 7);
    const int row1 = bit(t, 2) | bit(t, 5) << 1 | w0 << 3 | w1 << 4;                       // P1: r0 r1 r3 r4
    const int col1 = bit(t, 0) | bit(t, 1) << 1 | bit(t, 3) << 3 | bit(t, 4) << 4;           //     c0 c1 c3 c4
    const int row2 = bit(t, 5) | bit(t, 1) << 1 | w0 << 2 | bit(t, 2) << 3 | bit(t, 4) << 4 | w1 << 5;   // P2: all six row bits
    const int col2 = bit(t, 0) << 2 | bit(t, 3) << 5;                                        //     c2 c5
    const int row3 = w0 << 2 | w1 << 5;                                                      // P3: r2 r5
    const int col3 = bit(t, 0) | bit(t, 2) << 1 | bit(t, 4) << 2 | bit(t, 1) << 3 | bit(t, 3) << 4 | bit(t, 5) << 5;
    const int lcol = t & 63, krow = (t >> 6) * 16;                                           // output column / first output row of P3
    // LDS address of register r in a pass = the thread's base XOR a compile-time constant: the register bits are disjoint from
    // the thread bits, and the swizzle of a row depends on thread bits only (P1, P2) or on register bits only (P3).  The bases
    // are re-made opaque where they are used, so that the compiler keeps three of them across the item loop and not 48 addresses.
    const int base1 = s3_addr(row1, col1), base2 = s3_addr(row2, col2), base3 = row3 * 64 + col3;
    // register r = (b3 b2 b1 b0) of a pass -> its row / column offset
    auto reg_row1 = [](int r) { return ((r >> 3) & 1) << 5 | ((r >> 2) & 1) << 2; };          // P1: b3 = r5, b2 = r2
    auto reg_col1 = [](int r) { return ((r >> 1) & 1) << 5 | (r & 1) << 2; };                 //     b1 = c5, b0 = c2
    auto reg_col2 = [](int r) { return ((r >> 3) & 1) << 4 | ((r >> 2) & 1) << 1 | ((r >> 1) & 1) << 3 | (r & 1); };   // P2: c4 c1 c3 c0
    auto reg_row3 = [](int r) { return ((r >> 3) & 1) << 4 | ((r >> 2) & 1) << 1 | ((r >> 1) & 1) << 3 | (r & 1); };   // P3: r4 r1 r3 r0
    const double inv_d = 1.0 / d;
    const int n_ld = (K * D + NT - 1) / NT;                // operator entries per thread (K <= 31: at most 8)
    double2 nxt[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) nxt[q].x = nxt[q].y = 0.0;
    auto fetch = [&](long long item) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int idx = t + NT * q;
            if (q < n_ld && idx < K * D) nxt[q] = *reinterpret_cast<const double2*>(kraus + (item * (long long)K * D + idx) * 2);
        }
    };
    if ((long long)blockIdx.x < B) fetch(blockIdx.x);
    // passes 2 and 3 of one transform: X holds the tile after P1; `dst` gets the result times `scale`; returns the thread's share of
    // <ref, result> when asked
    auto finish = [&](double* __restrict__ dst, double scale, const double* __restrict__ ref) -> double {
        cplx x[16];
        const int b2 = opaque(base2);
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = X[b2 ^ reg_col2(r)];
        two_sites(x, +1.0, +1.0);                          // column sites 1, 0 (output qubits: +i)
#pragma unroll
        for (int r = 0; r < 16; ++r) X[b2 ^ reg_col2(r)] = x[r];
        __syncthreads();
        const int b3 = opaque(base3);
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = X[b3 ^ (reg_row3(r) * 64 + s3_swz(reg_row3(r)))];
        two_sites(x, -1.0, -1.0);                          // row sites 1, 0 (input qubits: -i)
        double acc = 0.0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long long o = ((long long)(krow + r) * D + lcol) * 2;
            double2 v; v.x = x[r].re * scale; v.y = x[r].im * scale;
            if (dst) FBX_STREAM_STORE(reinterpret_cast<double2*>(dst + o), v);
            if (ref) { const double2 q = *reinterpret_cast<const double2*>(ref + o); acc += q.x * v.x + q.y * v.y; }
        }
        return acc;
    };
    for (long long item = blockIdx.x; item < B; item += gridDim.x) {
        __syncthreads();                                   // the previous item's readers of X / kb / red are done
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int idx = t + NT * q;
            if (q < n_ld && idx < K * D) { cplx c; c.re = nxt[q].x; c.im = nxt[q].y; kb[idx] = c; }
        }
        if (item + gridDim.x < B) fetch(item + gridDim.x); // the next item's operators arrive behind this item's work
        __syncthreads();
        if (ptm_out || fid_out) {
            // P1 on kron(conj(K), K)[(i,k)][(j,l)] = conj(K[i][j]) K[k][l]: row = 8 i + k, col = 8 j + l
            cplx x[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) x[r].re = x[r].im = 0.0;
            const int i0 = row1 >> 3, k0 = row1 & 7, j0 = col1 >> 3, l0 = col1 & 7;      // bit 2 of each comes from the register
            for (int q = 0; q < K; ++q) {
                cplx a[2][2], b[2][2];
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int v = 0; v < 2; ++v) {
                        a[u][v] = kb[q * D + (i0 | u << 2) * d + (j0 | v << 2)];
                        b[u][v] = kb[q * D + (k0 | u << 2) * d + (l0 | v << 2)];
                    }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const cplx aa = a[(r >> 3) & 1][(r >> 1) & 1], bb = b[(r >> 2) & 1][r & 1];
                    x[r].re += aa.re * bb.re + aa.im * bb.im;
                    x[r].im += aa.re * bb.im - aa.im * bb.re;
                }
            }
            two_sites(x, -1.0, +1.0);                      // row site 2 (-i), column site 2 (+i)
            const int b1 = opaque(base1);
#pragma unroll
            for (int r = 0; r < 16; ++r) X[b1 ^ (reg_row1(r) * 64 + reg_col1(r))] = x[r];
            __syncthreads();
            double acc = finish(ptm_out ? ptm_out + item * (long long)D * D * 2 : nullptr, inv_d, fid_out ? ptm_ref : nullptr);
            if (fid_out) {
                acc = block_sum<NT>(acc, red);
                if (t == 0) fid_out[item] = (d * (acc / (double)(d * d)) + 1.0) / (d + 1.0);
            }
            __syncthreads();                               // X is rebuilt below
        }
        if (choi_out || chi_out) {
            // P1 on choi[row][col] = vK[row] conj(vK[col]), vK[8 c + r] = K[r][c]
            cplx x[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) x[r].re = x[r].im = 0.0;
            for (int q = 0; q < K; ++q) {
                cplx a[4], b[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int row = row1 | ((u >> 1) & 1) << 5 | (u & 1) << 2, col = col1 | ((u >> 1) & 1) << 5 | (u & 1) << 2;
                    a[u] = kb[q * D + (row % d) * d + row / d];
                    b[u] = kb[q * D + (col % d) * d + col / d];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const cplx aa = a[r >> 2], bb = b[r & 3];
                    x[r].re += aa.re * bb.re + aa.im * bb.im;
                    x[r].im += aa.im * bb.re - aa.re * bb.im;
                }
            }
            if (choi_out) {
                double* dst = choi_out + item * (long long)D * D * 2;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    double2 v; v.x = x[r].re; v.y = x[r].im;
                    FBX_STREAM_STORE(reinterpret_cast<double2*>(dst + ((long long)(row1 | reg_row1(r)) * D + (col1 | reg_col1(r))) * 2), v);
                }
            }
            if (chi_out) {
                two_sites(x, -1.0, +1.0);
                const int b1 = opaque(base1);
#pragma unroll
                for (int r = 0; r < 16; ++r) X[b1 ^ (reg_row1(r) * 64 + reg_col1(r))] = x[r];
                __syncthreads();
                (void)finish(chi_out + item * (long long)D * D * 2, inv_d * inv_d, nullptr);
            }
        }
    }
}

// The pairwise 3-qubit routes between Choi / superoperator / Pauli-Liouville in the same three register passes (round 4): 64 KB
// in, 64 KB out per item, 256-thread workgroups, one swizzled 64 KB tile.  ROUTE as convert3_fast_kernel: 0 choi->PL,
// 2 superop->PL, 3 PL->superop (route 1, PL->choi, keeps the one-stage-per-pass kernel: its result leaves reshuffled, i.e. as
// 16-byte pieces 8 KB apart from any register layout -- measured 7 x slower than a shuffle on the LDS side).  P1 loads the tile
// straight from HBM into its registers (route 0: reshuffled, scattered 16-byte reads that L2 absorbs; route 3: the Pauli index
// permuted to the site order, 256-byte runs), P3 stores whole rows; route 3 runs the inverse butterflies in the same order
// (the stages commute).
__device__ __forceinline__ void two_sites_inv(cplx (&x)[16], double y1, double y2) {
    auto site = [](cplx& c00, cplx& c11, cplx& c01, cplx& c10, double ys) {   // (I, Z, X, Y) at (00, 11, 01, 10)
        cplx o00, o11, o01, o10;
        o00.re = 0.5 * (c00.re + c11.re); o00.im = 0.5 * (c00.im + c11.im);
        o11.re = 0.5 * (c00.re - c11.re); o11.im = 0.5 * (c00.im - c11.im);
        const double yr = -ys * c10.im, yi = ys * c10.re;                      // s * i * Y
        o01.re = 0.5 * (c01.re - yr); o01.im = 0.5 * (c01.im - yi);
        o10.re = 0.5 * (c01.re + yr); o10.im = 0.5 * (c01.im + yi);
        c00 = o00; c11 = o11; c01 = o01; c10 = o10;
    };
#pragma unroll
    for (int cd = 0; cd < 4; ++cd) site(x[cd], x[12 | cd], x[4 | cd], x[8 | cd], y1);
#pragma unroll
    for (int ab = 0; ab < 4; ++ab) site(x[ab << 2], x[(ab << 2) | 3], x[(ab << 2) | 1], x[(ab << 2) | 2], y2);
}

template <int ROUTE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
convert3_regs_kernel(long long B, const double* __restrict__ in, double* __restrict__ out) {
    static_assert(ROUTE == 0 || ROUTE == 2 || ROUTE == 3, "route 1 stays with convert3_fast_kernel");
    constexpr int d = 8, D = 64;
    constexpr bool TO_PL = ROUTE != 3;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx* X = (cplx*)smem;
    const int t = threadIdx.x;
    auto bit = [](int v, int b) { return (v >> b) & 1; };
    const int w0 = bit(t, 6), w1 = bit(t, 7);
    const int row1 = bit(t, 2) | bit(t, 5) << 1 | w0 << 3 | w1 << 4;
    const int col1 = bit(t, 0) | bit(t, 1) << 1 | bit(t, 3) << 3 | bit(t, 4) << 4;
    const int row2 = bit(t, 5) | bit(t, 1) << 1 | w0 << 2 | bit(t, 2) << 3 | bit(t, 4) << 4 | w1 << 5;
    const int col2 = bit(t, 0) << 2 | bit(t, 3) << 5;
    const int row3 = w0 << 2 | w1 << 5;
    const int col3 = bit(t, 0) | bit(t, 2) << 1 | bit(t, 4) << 2 | bit(t, 1) << 3 | bit(t, 3) << 4 | bit(t, 5) << 5;
    const int lcol = t & 63, krow = (t >> 6) * 16;
    auto reg_row1 = [](int r) { return ((r >> 3) & 1) << 5 | ((r >> 2) & 1) << 2; };
    auto reg_col1 = [](int r) { return ((r >> 1) & 1) << 5 | (r & 1) << 2; };
    auto reg_col2 = [](int r) { return ((r >> 3) & 1) << 4 | ((r >> 2) & 1) << 1 | ((r >> 1) & 1) << 3 | (r & 1); };
    auto reg_row3 = [](int r) { return ((r >> 3) & 1) << 4 | ((r >> 2) & 1) << 1 | ((r >> 1) & 1) << 3 | (r & 1); };
    const int base1 = s3_addr(row1, col1), base2 = s3_addr(row2, col2), base3 = row3 * 64 + col3;
    // Pauli label of a tile row / column (inverse of site_index: label bit 2t = index bit t, 2t + 1 = index bit 3 + t)
    auto label = [](int x) { return (x & 1) | ((x >> 3) & 1) << 1 | ((x >> 1) & 1) << 2 | ((x >> 4) & 1) << 3 | ((x >> 2) & 1) << 4 | ((x >> 5) & 1) << 5; };
    // HBM index the tile entry (row, col) is loaded from
    auto load_index = [&](int row, int col) {
        if (ROUTE == 2) return row * D + col;
        if (ROUTE == 3) return label(row) * D + label(col);
        const int p = row / d, q = row % d, r = col / d, s_ = col % d;       // route 0: entry (p,q),(r,s) <- Choi entry (s,q),(r,p)
        return (s_ * d + q) * D + r * d + p;
    };
    const double inv_d = 1.0 / d;
    const double sc_in = TO_PL ? 1.0 : inv_d * D, sc_out = TO_PL ? inv_d : 1.0;
    auto sites = [](cplx (&x)[16], double y1, double y2) { if (TO_PL) two_sites(x, y1, y2); else two_sites_inv(x, y1, y2); };
    for (long long item = blockIdx.x; item < B; item += gridDim.x) {
        const double* src = in + item * (long long)D * D * 2;
        double* dst = out + item * (long long)D * D * 2;
        cplx x[16];
        __syncthreads();                                   // the previous item's readers of X are done
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const double2 v = *reinterpret_cast<const double2*>(src + 2 * load_index(row1 | reg_row1(r), col1 | reg_col1(r)));
            x[r].re = v.x * sc_in; x[r].im = v.y * sc_in;
        }
        sites(x, -1.0, +1.0);                              // row site 2 (-i), column site 2 (+i)
        const int b1 = opaque(base1);
#pragma unroll
        for (int r = 0; r < 16; ++r) X[b1 ^ (reg_row1(r) * 64 + reg_col1(r))] = x[r];
        __syncthreads();
        const int b2 = opaque(base2);
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = X[b2 ^ reg_col2(r)];
        sites(x, +1.0, +1.0);                              // column sites 1, 0
#pragma unroll
        for (int r = 0; r < 16; ++r) X[b2 ^ reg_col2(r)] = x[r];
        __syncthreads();
        const int b3 = opaque(base3);
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = X[b3 ^ (reg_row3(r) * 64 + s3_swz(reg_row3(r)))];
        sites(x, -1.0, -1.0);                              // row sites 1, 0
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            double2 v; v.x = x[r].re * sc_out; v.y = x[r].im * sc_out;
            // towards PL: Pauli row 16 w + r, column = lane; route 3: tile row / column as they are
            const long long o = TO_PL ? ((long long)(krow + r) * D + lcol) : ((long long)(row3 | reg_row3(r)) * D + col3);
            FBX_STREAM_STORE(reinterpret_cast<double2*>(dst + 2 * o), v);
        }
    }
}
static int launch_sweep3_regs(int64_t B, int K, const double* kraus, const double* ptm_ref, double* choi, double* ptm, double* chi, double* fid) {
    const size_t lds = sizeof(cplx) * 64 * 64 + sizeof(double) * 16 + sizeof(cplx) * (size_t)K * 64;
    FBX_HIP(hipFuncSetAttribute((const void*)sweep3_regs_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const unsigned grid = (unsigned)(B < 2048 ? B : 2048);
    hipLaunchKernelGGL(sweep3_regs_kernel, dim3(grid), dim3(256), lds, stream(), (long long)B, K, kraus, ptm_ref, choi, ptm, chi, fid);
    FBX_HIP(hipGetLastError());
    return FBX_OK;
}

template <int ROUTE>
static int launch_convert3_regs(int64_t B, const double* in, double* out) {
    const size_t lds = sizeof(cplx) * 64 * 64;
    auto kern = convert3_regs_kernel<ROUTE>;
    FBX_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const unsigned grid = (unsigned)(B < 2048 ? B : 2048);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream(), (long long)B, in, out);
    FBX_HIP(hipGetLastError());
    return FBX_OK;
}

// Three qubits, unfused (kept as the reference form: FBX_SWEEP3_COMPOSED=1 in the environment of a diagnostics build, and the
// fallback for more than 31 Kraus operators): the composition of the pairwise 64 x 64 conversions and the fidelity reduction.
static int launch_sweep3_composed(int64_t B, int K, const double* kraus, const double* ptm_ref, double* choi, double* ptm,
                         double* chi, double* fid) {
    constexpr size_t D = 64;
    DevBuf tmp;
    double* ptm_buf = ptm;
    if (fid && !ptm_buf) {
        const int rc = tmp.alloc(sizeof(cplx) * D * D * (size_t)B);
        if (rc) return rc;
        ptm_buf = tmp.as<double>();
    }
    int rc = FBX_OK;
    if (choi && (rc = launch_convert3(FBX_REP_KRAUS, FBX_REP_CHOI, B, kraus, K, choi))) return rc;
    if (ptm_buf && (rc = launch_convert3(FBX_REP_KRAUS, FBX_REP_PAULI_LIOUVILLE, B, kraus, K, ptm_buf))) return rc;
    if (chi && (rc = launch_convert3(FBX_REP_KRAUS, FBX_REP_CHI, B, kraus, K, chi))) return rc;
    if (fid) {
        const unsigned grid = (unsigned)(B < 8192 ? B : 8192);
        hipLaunchKernelGGL(process_fidelity_kernel, dim3(grid), dim3(64), 0, stream(), 8, (long long)B, ptm_ref, 0, ptm_buf,
                           (double*)nullptr, fid);
        FBX_HIP(hipGetLastError());
        if (tmp.p) FBX_HIP(hipStreamSynchronize(stream()));      // the scratch PTMs go away with `tmp`
    }
    return FBX_OK;
}

// ---------------------------------------------------------------------------------------------
// linear_inv_process_estimate (tomography.py:459-491): R[i][:] = pinv(Abar_i) e_i, Choi by the
// inverse Pauli transform, plus the explicit identity term I_D / d (tomography.py:491)
// ---------------------------------------------------------------------------------------------
// ITEMS experiments per wavefront: a row of the block pseudo-inverses (540 x 16 doubles for two qubits, more than the
// L1 holds) is loaded once and used for all of them -- one experiment per wavefront ran at the L2's pace (6.8e7 /s).
// Every experiment's sums run over the same settings in the same order as before.
template <int NQ, int ITEMS>
__global__ void __launch_bounds__(64)
linv_process_kernel(DesignDev des, long long B, const double* __restrict__ expect, double* __restrict__ out) {
    constexpr int d = 1 << NQ, D = d * d, NB = D / 2, PER = (D * D + 63) / 64;
    __shared__ double Rb[D * D];
    __shared__ cplx Mw[D * (D + 1)];
    const int lane = threadIdx.x;
    const long long first = (long long)blockIdx.x * ITEMS;
    double acc[PER][ITEMS];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int idx = lane + 64 * u;
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) acc[u][k] = 0.0;
        if (idx < D * D) {
            const int i = idx / D, j = idx % D;
            for (int g = des.pptr[i]; g < des.pptr[i + 1]; ++g) {
                const double p = des.pinvT[(size_t)g * D + j];
                const int col = des.porder[g];
#pragma unroll
                for (int k = 0; k < ITEMS; ++k)
                    if (first + k < B) acc[u][k] += expect[(first + k) * des.m + col] * p;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const long long item = first + k;
        if (item >= B) break;                                   // uniform
        __syncthreads();
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int idx = lane + 64 * u;
            if (idx < D * D) Rb[(idx % D) * D + idx / D] = acc[u][k] + ((idx == 0) ? 1.0 : 0.0);   // transposed, as pauli_real_to_choi_blk reads it
        }
        __syncthreads();
        const Blk c = pauli_real_to_choi_blk<NQ>(Rb, Mw, lane);
        if (lane < NB * NB) {
            const int I = lane / NB, J = lane % NB;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = 2 * I + (e >> 1), col = 2 * J + (e & 1);
                double* o = out + ((item * D + row) * D + col) * 2;
                o[0] = c.re[e]; o[1] = c.im[e];
            }
        }
    }
}

}  // namespace fbx

namespace fbx {     // fbx_pgdb3.hip
int proj_choi3_launch(int kind, int64_t B, const double* d_in, double* d_out, int32_t* d_iters);
int linv_process3_launch(const fbx_design* des, int64_t B, const double* d_expect, double* d_out);
}

using namespace fbx;

namespace {
struct HostIO {     // host <-> device staging for the host-pointer entry points
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
}  // namespace

// ---- Kraus bookkeeping for batches (operator_tools/compose_superoperators.py:7-44) and the Pauli twirl
// (channel_approximation.py:31-49).  Output operator p = j * K2 + l (the reference's list order: k1 outer,
// k2 inner) is kron(k2[l], k1[j]) for the tensor form, k2[l] . k1[j] for the composition; one output
// element per thread, inputs read through L2 (every input element is used K times).
namespace fbx {
__global__ void __launch_bounds__(256)
kraus_pairs_kernel(int tensor, long long B, int K2, int r2, int c2, int K1, int r1, int c1,
                   const cplx* __restrict__ k2, const cplx* __restrict__ k1, cplx* __restrict__ out) {
    const int ro = tensor ? r2 * r1 : r2, co = tensor ? c2 * c1 : c1;
    const long long per = (long long)K1 * K2 * ro * co, total = B * per;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const long long b = idx / per;
        long long rem = idx - b * per;
        const int p = (int)(rem / ((long long)ro * co)); rem -= (long long)p * ro * co;
        const int r = (int)(rem / co), c = (int)(rem % co);
        const int j = p / K2, l = p % K2;
        const cplx* A = k2 + ((size_t)b * K2 + l) * r2 * c2;
        const cplx* Bm = k1 + ((size_t)b * K1 + j) * r1 * c1;
        cplx o; o.re = 0.0; o.im = 0.0;
        if (tensor) {
            const cplx x = A[(r / r1) * c2 + (c / c1)], y = Bm[(r % r1) * c1 + (c % c1)];
            o.re = x.re * y.re - x.im * y.im; o.im = x.re * y.im + x.im * y.re;
        } else {
            for (int t = 0; t < c2; ++t) {
                const cplx x = A[r * c2 + t], y = Bm[t * c1 + c];
                o.re += x.re * y.re - x.im * y.im; o.im += x.re * y.im + x.im * y.re;
            }
        }
        out[idx] = o;
    }
}
__global__ void __launch_bounds__(256)
twirl_kernel(long long B, int D, const cplx* __restrict__ chi, cplx* __restrict__ out) {
    const long long total = B * D * D;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int e = (int)(idx % ((long long)D * D));
        cplx o; o.re = 0.0; o.im = 0.0;
        if (e / D == e % D) o = chi[idx];
        out[idx] = o;
    }
}
}  // namespace fbx

// ---- partial trace of an operator on A (x) B (calculational.py:5-35 with two subsystems): keep = 0 traces out B,
// keep = 1 traces out A.  Any dimensions; one thread per output entry.
__global__ void __launch_bounds__(256)
partial_trace2_kernel(int da, int db, int keep, long long B, const double* __restrict__ in, double* __restrict__ out) {
    const int n = keep == 0 ? da : db, m = keep == 0 ? db : da;
    const long long N = (long long)da * db;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= B * n * n) return;
    const long long item = gid / ((long long)n * n);
    const int r = (int)((gid / n) % n), c = (int)(gid % n);
    const double* src = in + item * N * N * 2;
    double re = 0.0, im = 0.0;
    for (int k = 0; k < m; ++k) {
        const long long row = keep == 0 ? (long long)r * db + k : (long long)k * db + r;
        const long long col = keep == 0 ? (long long)c * db + k : (long long)k * db + c;
        re += src[(row * N + col) * 2]; im += src[(row * N + col) * 2 + 1];
    }
    out[2 * gid] = re; out[2 * gid + 1] = im;
}

extern "C" {

int fbx_kraus_pairs_dev(int tensor, int64_t B, int K2, int rows2, int cols2, int K1, int rows1, int cols1,
                        const double* d_k2, const double* d_k1, double* d_out) {
    FBX_REQUIRE(B >= 0 && K1 >= 1 && K2 >= 1 && rows1 >= 1 && cols1 >= 1 && rows2 >= 1 && cols2 >= 1,
                "fbx_kraus_pairs: sizes must be positive");
    FBX_REQUIRE(tensor || cols2 == rows1, "fbx_kraus_pairs: composition needs cols(k2) == rows(k1)");
    FBX_REQUIRE(B == 0 || (d_k2 && d_k1 && d_out), "fbx_kraus_pairs: NULL buffer");
    FBX_TRY(ensure_device());
    if (B == 0) return FBX_OK;
    const long long ro = tensor ? (long long)rows2 * rows1 : rows2, co = tensor ? (long long)cols2 * cols1 : cols1;
    const long long total = (long long)B * K1 * K2 * ro * co, want = (total + 255) / 256;
    hipLaunchKernelGGL(kraus_pairs_kernel, dim3((unsigned)(want < 256 * 32 ? want : 256 * 32)), dim3(256), 0, stream(),
                       tensor, (long long)B, K2, rows2, cols2, K1, rows1, cols1, (const cplx*)d_k2, (const cplx*)d_k1, (cplx*)d_out);
    FBX_HIP(hipGetLastError());
    return FBX_OK;
}

int fbx_kraus_pairs(int tensor, int64_t B, int K2, int rows2, int cols2, int K1, int rows1, int cols1,
                    const double* k2, const double* k1, double* out) {
    FBX_REQUIRE(B >= 0 && K1 >= 1 && K2 >= 1 && rows1 >= 1 && cols1 >= 1 && rows2 >= 1 && cols2 >= 1,
                "fbx_kraus_pairs: sizes must be positive");
    FBX_REQUIRE(tensor || cols2 == rows1, "fbx_kraus_pairs: composition needs cols(k2) == rows(k1)");
    FBX_REQUIRE(B == 0 || (k2 && k1 && out), "fbx_kraus_pairs: NULL buffer");
    FBX_TRY(ensure_device());
    if (B == 0) return FBX_OK;
    const size_t n2 = (size_t)B * K2 * rows2 * cols2 * 2, n1 = (size_t)B * K1 * rows1 * cols1 * 2;
    const size_t ro = tensor ? (size_t)rows2 * rows1 : rows2, co = tensor ? (size_t)cols2 * cols1 : cols1;
    const size_t no = (size_t)B * K1 * K2 * ro * co * 2;
    HostIO io; double *d2, *d1, *dout;
    FBX_TRY(io.in(k2, n2, &d2)); FBX_TRY(io.in(k1, n1, &d1)); FBX_TRY(io.out(no, &dout));
    FBX_TRY(fbx_kraus_pairs_dev(tensor, B, K2, rows2, cols2, K1, rows1, cols1, d2, d1, dout));
    FBX_TRY(io.back(out, dout, no));
    return io.sync();
}

int fbx_pauli_twirl_chi_dev(int64_t B, int D, const double* d_chi, double* d_out) {
    FBX_REQUIRE(B >= 0 && D >= 1, "fbx_pauli_twirl_chi: bad size");
    FBX_REQUIRE(B == 0 || (d_chi && d_out), "fbx_pauli_twirl_chi: NULL buffer");
    FBX_TRY(ensure_device());
    if (B == 0) return FBX_OK;
    const long long total = (long long)B * D * D, want = (total + 255) / 256;
    hipLaunchKernelGGL(twirl_kernel, dim3((unsigned)(want < 256 * 32 ? want : 256 * 32)), dim3(256), 0, stream(),
                       (long long)B, D, (const cplx*)d_chi, (cplx*)d_out);
    FBX_HIP(hipGetLastError());
    return FBX_OK;
}

int fbx_pauli_twirl_chi(int64_t B, int D, const double* chi, double* out) {
    FBX_REQUIRE(B >= 0 && D >= 1, "fbx_pauli_twirl_chi: bad size");
    FBX_REQUIRE(B == 0 || (chi && out), "fbx_pauli_twirl_chi: NULL buffer");
    FBX_TRY(ensure_device());
    if (B == 0) return FBX_OK;
    const size_t n = (size_t)B * D * D * 2;
    HostIO io; double *dc, *dout;
    FBX_TRY(io.in(chi, n, &dc)); FBX_TRY(io.out(n, &dout));
    FBX_TRY(fbx_pauli_twirl_chi_dev(B, D, dc, dout));
    FBX_TRY(io.back(out, dout, n));
    return io.sync();
}

int fbx_linv_process_dev(const fbx_design* design, int64_t B, const double* d_expect, double* d_choi_out) {
    FBX_TRY(check_design(design, "fbx_linv_process"));
    FBX_REQUIRE(design->dev.kind == FBX_KIND_PROCESS, "fbx_linv_process: needs a process design");
    FBX_REQUIRE(B >= 0 && (B == 0 || (d_expect && d_choi_out)), "fbx_linv_process: bad batch / NULL buffer");
    FBX_TRY(ensure_device());
    if (B == 0) return FBX_OK;
    const int n = design->dev.n;
    if (n == 3) FBX_TRY(linv_process3_launch(design, B, d_expect, d_choi_out));
    else if (n == 1) hipLaunchKernelGGL((linv_process_kernel<1, 4>), dim3((unsigned)((B + 3) / 4)), dim3(64), 0, stream(), design->dev, (long long)B, d_expect, d_choi_out);
    else hipLaunchKernelGGL((linv_process_kernel<2, 4>), dim3((unsigned)((B + 3) / 4)), dim3(64), 0, stream(), design->dev, (long long)B, d_expect, d_choi_out);
    FBX_HIP(hipGetLastError());
    return FBX_OK;
}

int fbx_linv_process(const fbx_design* design, int64_t B, const double* expect, double* choi_out) {
    FBX_TRY(check_design(design, "fbx_linv_process"));
    FBX_REQUIRE(design->dev.kind == FBX_KIND_PROCESS, "fbx_linv_process: needs a process design");
    FBX_REQUIRE(B >= 0 && (B == 0 || (expect && choi_out)), "fbx_linv_process: bad batch / NULL buffer");
    FBX_TRY(ensure_device());
    if (B == 0) return FBX_OK;
    const size_t m = design->dev.m, D = design->dev.D;
    HostIO io; double *de, *dout;
    FBX_TRY(io.in(expect, m * B, &de)); FBX_TRY(io.out(D * D * 2 * B, &dout));
    FBX_TRY(fbx_linv_process_dev(design, B, de, dout));
    FBX_TRY(io.back(choi_out, dout, D * D * 2 * B));
    return io.sync();
}

static int convert_check(int from_rep, int to_rep, int n_qubits, int64_t B, const void* in, int K, const void* out) {
    FBX_REQUIRE(n_qubits >= 1 && n_qubits <= 5, "fbx_convert: n_qubits must be 1..5");
    FBX_REQUIRE(from_rep >= FBX_REP_KRAUS && from_rep <= FBX_REP_CHI, "fbx_convert: bad source representation");
    FBX_REQUIRE(to_rep >= FBX_REP_CHOI && to_rep <= FBX_REP_CHI, "fbx_convert: bad target representation (Kraus output is not offered)");
    FBX_REQUIRE(from_rep != to_rep, "fbx_convert: source and target representation are the same");
    FBX_REQUIRE(B >= 0 && (B == 0 || (in && out)), "fbx_convert: bad batch / NULL buffer");
    FBX_REQUIRE(from_rep != FBX_REP_KRAUS || K >= 1, "fbx_convert: need K >= 1 Kraus operators");
    return FBX_OK;
}

int fbx_convert_dev(int from_rep, int to_rep, int n_qubits, int64_t B, const double* d_in, int K, double* d_out) {
    FBX_TRY(convert_check(from_rep, to_rep, n_qubits, B, d_in, K, d_out));
    FBX_TRY(ensure_device());
    if (B == 0) return FBX_OK;
    auto dispatch = [&](int from, int to, const double* in, int k, double* out, bool psd) -> int {
        if (n_qubits == 5) return launch_convert_big<5>(from, to, B, in, k, out, psd);
        if (n_qubits == 4) return launch_convert_big<4>(from, to, B, in, k, out, psd);
        if (n_qubits == 3) return launch_convert3(from, to, B, in, k, out);
        if (n_qubits == 1) return launch_convert<1>(from, to, B, in, k, out);
        return launch_convert<2>(from, to, B, in, k, out);
    };
    auto dispatch_n = [&](int from, int to, int64_t nb, const double* in, int k, double* out, bool psd) -> int {
        if (n_qubits == 5) return launch_convert_big<5>(from, to, nb, in, k, out, psd);
        if (n_qubits == 4) return launch_convert_big<4>(from, to, nb, in, k, out, psd);
        if (n_qubits == 3) return launch_convert3(from, to, nb, in, k, out);
        if (n_qubits == 1) return launch_convert<1>(from, to, nb, in, k, out);
        return launch_convert<2>(from, to, nb, in, k, out);
    };
    const int rc = dispatch(from_rep, to_rep, d_in, K, d_out, false);
    // (from a Kraus set the launchers above return FBX_ERR_UNSUPPORTED for exactly one reason: K operators do not fit their LDS staging)
    if (rc != FBX_ERR_UNSUPPORTED || from_rep != FBX_REP_KRAUS) return rc;
    // More Kraus operators than the fused kernels stage in LDS (K x D x 16 B against 160 KiB: 40 operators for 4 qubits, 10 for
    // 5): the Choi matrix from the basis-free kernel, which takes any K (one thread per entry, operators read through L2),
    // then on from there -- the Choi matrix of a Kraus set is PSD, so the way into chi is the linear one.  In chunks of at
    // most 256 MiB of Choi matrices (a 5-qubit item is 16 MiB), like convert_into_chi_big.
    const size_t D = (size_t)1 << (2 * n_qubits), d = (size_t)1 << n_qubits;
    int rc2 = FBX_OK;
    if (to_rep == FBX_REP_CHOI) rc2 = fbx_convert_general_dev(FBX_REP_KRAUS, FBX_REP_CHOI, 1 << n_qubits, B, d_in, K, d_out);
    else {
        const size_t per_item = D * D * sizeof(cplx);
        const int64_t chunk = (int64_t)std::max<size_t>(1, std::min<size_t>((size_t)B, ((size_t)256 << 20) / per_item));
        DevBuf choi;
        FBX_TRY(choi.alloc(per_item * (size_t)chunk));
        for (int64_t b0 = 0; b0 < B && rc2 == FBX_OK; b0 += chunk) {
            const int64_t nb = std::min<int64_t>(chunk, B - b0);
            rc2 = fbx_convert_general_dev(FBX_REP_KRAUS, FBX_REP_CHOI, 1 << n_qubits, nb, d_in + (size_t)b0 * K * d * d * 2, K, choi.as<double>());
            if (rc2 == FBX_OK) rc2 = dispatch_n(FBX_REP_CHOI, to_rep, nb, choi.as<double>(), 0, d_out + (size_t)b0 * D * D * 2, true);
        }
    }
    if (rc2 == FBX_OK) set_error("");              // the fused path's "too many Kraus operators" is not this call's outcome
    return rc2;
}

static int convert_general_check(int from_rep, int to_rep, int dim, int64_t B, const void* in, int K, const void* out) {
    FBX_REQUIRE(dim >= 1 && dim <= 256, "fbx_convert_general: dim must be 1..256");
    const bool ok = (from_rep == FBX_REP_KRAUS && (to_rep == FBX_REP_SUPEROP || to_rep == FBX_REP_CHOI)) ||
                    (from_rep == FBX_REP_SUPEROP && to_rep == FBX_REP_CHOI) || (from_rep == FBX_REP_CHOI && to_rep == FBX_REP_SUPEROP);
    FBX_REQUIRE(ok, "fbx_convert_general: only kraus -> superop / choi and superop <-> choi are basis free");
    FBX_REQUIRE(B >= 0 && (B == 0 || (in && out)), "fbx_convert_general: bad batch / NULL buffer");
    FBX_REQUIRE(from_rep != FBX_REP_KRAUS || K >= 1, "fbx_convert_general: need K >= 1 Kraus operators");
    return FBX_OK;
}

int fbx_partial_trace_dev(int dim_a, int dim_b, int keep, int64_t B, const double* d_in, double* d_out) {
    FBX_REQUIRE(dim_a >= 1 && dim_b >= 1 && (long long)dim_a * dim_b <= 4096, "fbx_partial_trace: dimensions must be >= 1 with dim_a * dim_b <= 4096");
    FBX_REQUIRE(keep == 0 || keep == 1, "fbx_partial_trace: keep must be 0 (first subsystem) or 1 (second)");
    FBX_REQUIRE(B >= 0 && (B == 0 || (d_in && d_out)), "fbx_partial_trace: bad batch / NULL buffer");
    FBX_TRY(ensure_device());
    if (B == 0) return FBX_OK;
    const long long n = keep == 0 ? dim_a : dim_b, total = (long long)B * n * n;
    hipLaunchKernelGGL(partial_trace2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream(), dim_a, dim_b, keep,
                       (long long)B, d_in, d_out);
    FBX_HIP(hipGetLastError());
    return FBX_OK;
}

int fbx_partial_trace(int dim_a, int dim_b, int keep, int64_t B, const double* in, double* out) {
    FBX_REQUIRE(dim_a >= 1 && dim_b >= 1 && (long long)dim_a * dim_b <= 4096, "fbx_partial_trace: dimensions must be >= 1 with dim_a * dim_b <= 4096");
    FBX_REQUIRE(keep == 0 || keep == 1, "fbx_partial_trace: keep must be 0 (first subsystem) or 1 (second)");
    FBX_REQUIRE(B >= 0 && (B == 0 || (in && out)), "fbx_partial_trace: bad batch / NULL buffer");
    FBX_TRY(ensure_device());
    if (B == 0) return FBX_OK;
    const size_t N = (size_t)dim_a * dim_b, n = keep == 0 ? dim_a : dim_b;
    HostIO io; double *d_in, *d_out;
    FBX_TRY(io.in(in, N * N * 2 * B, &d_in)); FBX_TRY(io.out(n * n * 2 * B, &d_out));
    FBX_TRY(fbx_partial_trace_dev(dim_a, dim_b, keep, B, d_in, d_out));
    FBX_TRY(io.back(out, d_out, n * n * 2 * B));
    return io.sync();
}

int fbx_convert_general_dev(int from_rep, int to_rep, int dim, int64_t B, const double* d_in, int K, double* d_out) {
    FBX_TRY(convert_general_check(from_rep, to_rep, dim, B, d_in, K, d_out));
    FBX_TRY(ensure_device());
    if (B == 0) return FBX_OK;
    const long long total = (long long)B * dim * dim * dim * dim;
    hipLaunchKernelGGL(convert_general_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream(), from_rep, to_rep, dim,
                       (long long)B, d_in, K, d_out);
    FBX_HIP(hipGetLastError());
    return FBX_OK;
}

int fbx_convert_general(int from_rep, int to_rep, int dim, int64_t B, const double* in, int K, double* out) {
    FBX_TRY(convert_general_check(from_rep, to_rep, dim, B, in, K, out));
    FBX_TRY(ensure_device());
    if (B == 0) return FBX_OK;
    const size_t D = (size_t)dim * dim;
    const size_t n_in = (from_rep == FBX_REP_KRAUS ? (size_t)K * D : D * D) * 2 * B, n_out = D * D * 2 * B;
    HostIO io; double *d_in, *d_out;
    FBX_TRY(io.in(in, n_in, &d_in)); FBX_TRY(io.out(n_out, &d_out));
    FBX_TRY(fbx_convert_general_dev(from_rep, to_rep, dim, B, d_in, K, d_out));
    FBX_TRY(io.back(out, d_out, n_out));
    return io.sync();
}

int fbx_convert(int from_rep, int to_rep, int n_qubits, int64_t B, const double* in, int K, double* out) {
    FBX_TRY(convert_check(from_rep, to_rep, n_qubits, B, in, K, out));
    FBX_TRY(ensure_device());
    if (B == 0) return FBX_OK;
    const size_t d = (size_t)1 << n_qubits, D = d * d;
    const size_t n_in = (from_rep == FBX_REP_KRAUS ? (size_t)K * D : D * D) * 2 * B, n_out = D * D * 2 * B;
    HostIO io; double *d_in, *d_out;
    FBX_TRY(io.in(in, n_in, &d_in)); FBX_TRY(io.out(n_out, &d_out));
    FBX_TRY(fbx_convert_dev(from_rep, to_rep, n_qubits, B, d_in, K, d_out));
    FBX_TRY(io.back(out, d_out, n_out));
    return io.sync();
}

int fbx_kraus_sweep_dev(int n_qubits, int64_t B, int K, const double* d_kraus, const double* d_ptm_ref,
                        double* d_choi_out, double* d_ptm_out, double* d_chi_out, double* d_fid_out) {
    FBX_REQUIRE(n_qubits >= 1 && n_qubits <= 3, "fbx_kraus_sweep: n_qubits must be 1..3");
    FBX_REQUIRE(B >= 0 && K >= 1 && (B == 0 || d_kraus), "fbx_kraus_sweep: bad arguments");
    FBX_REQUIRE(!d_fid_out || d_ptm_ref, "fbx_kraus_sweep: fidelity output needs a reference PTM");
    FBX_TRY(ensure_device());
    if (B == 0) return FBX_OK;
    if (n_qubits == 3) {
        const size_t lds = sizeof(cplx) * 64 * 64 + sizeof(double) * 16 + sizeof(cplx) * (size_t)K * 64;
        if (lds > 80 * 1024) return launch_sweep3_composed(B, K, d_kraus, d_ptm_ref, d_choi_out, d_ptm_out, d_chi_out, d_fid_out);
        const unsigned grid = (unsigned)(B < 2048 ? B : 2048);      // two workgroups per CU, four rounds of the chip: persistent over the items
        const char* v1s = getenv("FBX_SWEEP3_V1");                   // 1 = the one-stage-per-pass form (A/B, tests)
        const bool v1 = v1s && atoi(v1s) != 0;
        if (v1) {
            FBX_HIP(hipFuncSetAttribute((const void*)sweep3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(sweep3_kernel, dim3(grid), dim3(1024), lds, stream(), (long long)B, K, d_kraus, d_ptm_ref, d_choi_out, d_ptm_out,
                               d_chi_out, d_fid_out);
        } else return launch_sweep3_regs(B, K, d_kraus, d_ptm_ref, d_choi_out, d_ptm_out, d_chi_out, d_fid_out);
        FBX_HIP(hipGetLastError());
        return FBX_OK;
    }
    if (n_qubits == 1) return launch_sweep<1>(B, K, d_kraus, d_ptm_ref, d_choi_out, d_ptm_out, d_chi_out, d_fid_out);
    return launch_sweep<2>(B, K, d_kraus, d_ptm_ref, d_choi_out, d_ptm_out, d_chi_out, d_fid_out);
}

int fbx_kraus_sweep(int n_qubits, int64_t B, int K, const double* kraus, const double* ptm_ref,
                    double* choi_out, double* ptm_out, double* chi_out, double* fid_out) {
    FBX_REQUIRE(n_qubits >= 1 && n_qubits <= 3, "fbx_kraus_sweep: n_qubits must be 1..3");
    FBX_REQUIRE(B >= 0 && K >= 1 && (B == 0 || kraus), "fbx_kraus_sweep: bad arguments");
    FBX_REQUIRE(!fid_out || ptm_ref, "fbx_kraus_sweep: fidelity output needs a reference PTM");
    FBX_TRY(ensure_device());
    if (B == 0) return FBX_OK;
    const size_t d = (size_t)1 << n_qubits, D = d * d, nm = D * D * 2 * B;
    // fbx_set_devices: contiguous blocks of the batch on the workers of the device list (the items are independent)
    if (device_list_size() > 1 && !in_device_worker() && B >= 2 * (int64_t)device_list_size()) {
        return run_on_devices([&](int g, int G) -> int {       // G: the list's length as run_on_devices read it, under its lock
            const int64_t per = (B + G - 1) / G;
            const int64_t lo = (int64_t)g * per < B ? (int64_t)g * per : B, nb = (B - lo < per ? B - lo : per);
            if (nb <= 0) return FBX_OK;
            const size_t om = (size_t)lo * D * D * 2;
            return fbx_kraus_sweep(n_qubits, nb, K, kraus + (size_t)lo * K * D * 2, ptm_ref, choi_out ? choi_out + om : nullptr,
                                   ptm_out ? ptm_out + om : nullptr, chi_out ? chi_out + om : nullptr, fid_out ? fid_out + lo : nullptr);
        });
    }
    HostIO io; double *dk, *dr = nullptr, *dc = nullptr, *dp = nullptr, *dx = nullptr, *df = nullptr;
    FBX_TRY(io.in(kraus, (size_t)K * D * 2 * B, &dk));
    if (ptm_ref) FBX_TRY(io.in(ptm_ref, D * D * 2, &dr));
    if (choi_out) FBX_TRY(io.out(nm, &dc));
    if (ptm_out) FBX_TRY(io.out(nm, &dp));
    if (chi_out) FBX_TRY(io.out(nm, &dx));
    if (fid_out) FBX_TRY(io.out((size_t)B, &df));
    FBX_TRY(fbx_kraus_sweep_dev(n_qubits, B, K, dk, dr, dc, dp, dx, df));
    FBX_TRY(io.back(choi_out, dc, nm)); FBX_TRY(io.back(ptm_out, dp, nm)); FBX_TRY(io.back(chi_out, dx, nm));
    FBX_TRY(io.back(fid_out, df, (size_t)B));
    return io.sync();
}

int fbx_proj_choi_dev(int proj_kind, int n_qubits, int64_t B, const double* d_choi, double* d_out, int32_t* d_iters_out) {
    FBX_REQUIRE(n_qubits >= 1 && n_qubits <= 3, "fbx_proj_choi: n_qubits must be 1..3");
    FBX_REQUIRE(proj_kind >= FBX_PROJ_CP && proj_kind <= FBX_PROJ_PHYSICAL_TNI, "fbx_proj_choi: bad projection kind");
    FBX_REQUIRE(B >= 0 && (B == 0 || (d_choi && d_out)), "fbx_proj_choi: bad batch / NULL buffer");
    FBX_TRY(ensure_device());
    if (B == 0) return FBX_OK;
    if (n_qubits == 3) {
        FBX_TRY(proj_choi3_launch(proj_kind, B, d_choi, d_out, d_iters_out));
    } else if (n_qubits == 1) {
        const size_t lds = ChoiLds<1>::bytes() + 64;
        hipLaunchKernelGGL(proj_choi_kernel<1>, dim3((unsigned)B), dim3(64), lds, stream(), proj_kind, (long long)B, d_choi, d_out, d_iters_out);
    } else {
        const size_t lds = ChoiLds<2>::bytes() + 64;
        if (B >= 2048)
            hipLaunchKernelGGL(proj_choi_w2_kernel<2>, dim3((unsigned)B), dim3(64), lds, stream(), proj_kind, (long long)B, d_choi, d_out, d_iters_out);
        else
            hipLaunchKernelGGL(proj_choi_kernel<2>, dim3((unsigned)B), dim3(64), lds, stream(), proj_kind, (long long)B, d_choi, d_out, d_iters_out);
    }
    FBX_HIP(hipGetLastError());
    return FBX_OK;
}

int fbx_proj_choi(int proj_kind, int n_qubits, int64_t B, const double* choi, double* out, int32_t* iters_out) {
    FBX_REQUIRE(n_qubits >= 1 && n_qubits <= 3, "fbx_proj_choi: n_qubits must be 1..3");
    FBX_REQUIRE(B >= 0 && (B == 0 || (choi && out)), "fbx_proj_choi: bad batch / NULL buffer");
    FBX_REQUIRE(proj_kind >= FBX_PROJ_CP && proj_kind <= FBX_PROJ_PHYSICAL_TNI, "fbx_proj_choi: bad projection kind");
    FBX_TRY(ensure_device());
    if (B == 0) return FBX_OK;
    const size_t d = (size_t)1 << n_qubits, D = d * d, nm = D * D * 2 * B;
    HostIO io; double *d_in, *d_out; int32_t* d_it;
    FBX_TRY(io.in(choi, nm, &d_in)); FBX_TRY(io.out(nm, &d_out)); FBX_TRY(io.out((size_t)B, &d_it));
    FBX_TRY(fbx_proj_choi_dev(proj_kind, n_qubits, B, d_in, d_out, d_it));
    FBX_TRY(io.back(out, d_out, nm)); FBX_TRY(io.back(iters_out, d_it, (size_t)B));
    return io.sync();
}

int fbx_apply_choi_dev(int n_qubits, int64_t B, const double* d_choi, const double* d_rho, double* d_out) {
    FBX_REQUIRE(n_qubits >= 1 && n_qubits <= 3, "fbx_apply_choi: n_qubits must be 1..3");
    FBX_REQUIRE(B >= 0 && (B == 0 || (d_choi && d_rho && d_out)), "fbx_apply_choi: bad batch / NULL buffer");
    FBX_TRY(ensure_device());
    if (B == 0) return FBX_OK;
    const size_t d = (size_t)1 << n_qubits, D = d * d;
    const long long total = (long long)B * D;
    hipLaunchKernelGGL(apply_choi_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream(), (int)d, (long long)B, d_choi, d_rho, d_out);
    FBX_HIP(hipGetLastError());
    return FBX_OK;
}

int fbx_apply_choi(int n_qubits, int64_t B, const double* choi, const double* rho, double* out) {
    FBX_REQUIRE(n_qubits >= 1 && n_qubits <= 3, "fbx_apply_choi: n_qubits must be 1..3");
    FBX_REQUIRE(B >= 0 && (B == 0 || (choi && rho && out)), "fbx_apply_choi: bad batch / NULL buffer");
    FBX_TRY(ensure_device());
    if (B == 0) return FBX_OK;
    const size_t d = (size_t)1 << n_qubits, D = d * d;
    HostIO io; double *dc, *dr, *dout;
    FBX_TRY(io.in(choi, D * D * 2 * B, &dc)); FBX_TRY(io.in(rho, D * 2 * B, &dr)); FBX_TRY(io.out(D * 2 * B, &dout));
    FBX_TRY(fbx_apply_choi_dev(n_qubits, B, dc, dr, dout));
    FBX_TRY(io.back(out, dout, D * 2 * B));
    return io.sync();
}

int fbx_process_fidelity_dev(int n_qubits, int64_t B, const double* d_ptm0, const double* d_ptm1, double* d_fe_out, double* d_fp_out) {
    FBX_REQUIRE(n_qubits >= 1 && n_qubits <= 5, "fbx_process_fidelity: n_qubits must be 1..5");
    FBX_REQUIRE(B >= 0 && (B == 0 || (d_ptm0 && d_ptm1)), "fbx_process_fidelity: bad batch / NULL buffer");
    FBX_TRY(ensure_device());
    if (B == 0) return FBX_OK;
    const unsigned grid = (unsigned)(B < 8192 ? B : 8192);
    hipLaunchKernelGGL(process_fidelity_kernel, dim3(grid), dim3(64), 0, stream(), 1 << n_qubits, (long long)B, d_ptm0, 1, d_ptm1, d_fe_out, d_fp_out);
    FBX_HIP(hipGetLastError());
    return FBX_OK;
}

int fbx_process_fidelity(int n_qubits, int64_t B, const double* ptm0, const double* ptm1, double* fe_out, double* fp_out) {
    FBX_REQUIRE(n_qubits >= 1 && n_qubits <= 5, "fbx_process_fidelity: n_qubits must be 1..5");
    FBX_REQUIRE(B >= 0 && (B == 0 || (ptm0 && ptm1)), "fbx_process_fidelity: bad batch / NULL buffer");
    FBX_TRY(ensure_device());
    if (B == 0) return FBX_OK;
    const size_t d = (size_t)1 << n_qubits, D = d * d, nm = D * D * 2 * B;
    HostIO io; double *da, *db, *dfe, *dfp;
    FBX_TRY(io.in(ptm0, nm, &da)); FBX_TRY(io.in(ptm1, nm, &db));
    FBX_TRY(io.out((size_t)B, &dfe)); FBX_TRY(io.out((size_t)B, &dfp));
    FBX_TRY(fbx_process_fidelity_dev(n_qubits, B, da, db, dfe, dfp));
    FBX_TRY(io.back(fe_out, dfe, (size_t)B)); FBX_TRY(io.back(fp_out, dfp, (size_t)B));
    return io.sync();
}

}  // extern "C"
