// fbx_random.hip -- random operators generated ON the device from a counter-based stream.
//
// Restates operator_tools/random_operators.py:21-157 for batches (file:line under forest/benchmarking/):
//   ginibre_matrix_complex      :21-46   N(0,1) + i N(0,1) entries
//   haar_rand_unitary           :49-72   QR of a Ginibre matrix with diag(R) made positive (Mezzadri)
//   haar_rand_state             :75-89   first column of a Haar unitary
//   ginibre_state_matrix        :92-112  A A^H / tr, A = Ginibre(dim, rank)
//   bures_measure_state_matrix  :115-132 (1 + U) A A^H (1 + U)^H / tr
//   rand_map_with_BCSZ_dist     :135-157 random CPTP map; here in Kraus form, K_j = G_j S^{-1/2},
//                                        S = sum_j G_j^H G_j (its Choi matrix kraus2choi(K) equals the
//                                        reference's (rho_in^{-1/2} (x) 1) X X^H (rho_in^{-1/2} (x) 1)
//                                        when the columns of X are the column-stacked G_j)
// The reference draws from numpy's global Mersenne-Twister stream; here item b owns the Philox4x32-10
// stream (key = seed, counter = (item id, element index, stream tag)), so the matrices depend only on
// (seed, item id): any launch shape, any split over GPUs, any first_item offset gives the same items.
// Parity with the reference is distributional; the arithmetic after the normals is the reference's.
#include "fbx_eigh.hpp"

namespace fbx {

__device__ __forceinline__ void philox_block(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int round = 0; round < 10; ++round) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
        const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
        c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

// one complex standard normal N(0,1) + i N(0,1): Box-Muller on the first two words of the block
// (item, element, tag); the two outputs of one Box-Muller pair are independent normals
__device__ __forceinline__ cplx ginibre_entry(unsigned long long seed, long long item, uint32_t elem, uint32_t tag) {
    uint32_t c[4] = {(uint32_t)item, (uint32_t)((unsigned long long)item >> 32), elem, tag};
    philox_block(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    const double u1 = ((double)c[0] + 0.5) * 0x1p-32, u2 = ((double)c[1] + 0.5) * 0x1p-32;
    const double mag = sqrt(-2.0 * log(u1));
    double sn, cs;
    sincos(6.283185307179586476925 * u2, &sn, &cs);
    cplx z; z.re = mag * cs; z.im = mag * sn;
    return z;
}

__global__ void __launch_bounds__(256)
ginibre_kernel(long long B, int elems, unsigned long long seed, long long first_item, uint32_t tag, cplx* __restrict__ out) {
    const long long total = B * elems;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const long long b = idx / elems;
        out[idx] = ginibre_entry(seed, first_item + b, (uint32_t)(idx - b * elems), tag);
    }
}

// ---- d x d helpers on one wavefront (row-major matrices in LDS, lane t < d*d owns entry (t / d, t % d))
template <int d>
struct RandLds {
    cplx *A, *U, *W, *T;          // [d*d] row-major work matrices
    cplx *Ms, *Vs;                // Jacobi layout
    double* lam;                  // [d]
    static constexpr size_t bytes() { return sizeof(cplx) * 6 * d * d + sizeof(double) * d + 16; }
    __device__ void carve(char* p) {
        A = (cplx*)p; U = A + d * d; W = U + d * d; T = W + d * d; Ms = T + d * d; Vs = Ms + d * d;
        lam = (double*)(Vs + d * d);
    }
};

// Q factor of the QR decomposition with positive diag(R) (= Q diag(R)/|diag(R)| of
// random_operators.py:68-72), in place on the row-major d x d matrix M: Gram-Schmidt, each column
// orthogonalised twice (the second pass restores orthogonality to rounding).
template <int d>
__device__ void haar_q_factor(cplx* M, int lane) {
    for (int j = 0; j < d; ++j) {
        for (int pass = 0; pass < 2; ++pass) {
            for (int k = 0; k < j; ++k) {          // remove the component along column k
                double pr = 0.0, pi = 0.0;
                if (lane < d) {
                    const cplx q = M[lane * d + k], a = M[lane * d + j];
                    pr = q.re * a.re + q.im * a.im; pi = q.re * a.im - q.im * a.re;     // conj(q) a
                }
                pr = wave_sum(pr); pi = wave_sum(pi);
                FBX_WAVE_SYNC();
                if (lane < d) {
                    const cplx q = M[lane * d + k];
                    cplx a = M[lane * d + j];
                    a.re -= pr * q.re - pi * q.im; a.im -= pr * q.im + pi * q.re;
                    M[lane * d + j] = a;
                }
                FBX_WAVE_SYNC();
            }
        }
        double n2 = 0.0;
        if (lane < d) { const cplx a = M[lane * d + j]; n2 = a.re * a.re + a.im * a.im; }
        n2 = wave_sum(n2);
        const double inv = 1.0 / sqrt(n2);
        FBX_WAVE_SYNC();
        if (lane < d) { cplx a = M[lane * d + j]; a.re *= inv; a.im *= inv; M[lane * d + j] = a; }
        FBX_WAVE_SYNC();
    }
}

// dst = V f(lambda) V^H of the Hermitian row-major matrix `src`; fn 0: lambda^{-1/2}
template <int d>
__device__ void herm_inv_sqrt(const cplx* src, cplx* dst, RandLds<d>& L, int lane) {
    constexpr int NB = d / 2;
    Blk h = blk_zero();
    if (lane < NB * NB) {
        const int I = lane / NB, J = lane % NB;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int r = 2 * I + (e >> 1), c = 2 * J + (e & 1);
            const cplx a = src[r * d + c], b = src[c * d + r];
            h.re[e] = 0.5 * (a.re + b.re); h.im[e] = 0.5 * (a.im - b.im);
        }
    }
    FBX_WAVE_SYNC();
    sys_store<d>(L.Ms, lane, h);
    FBX_WAVE_SYNC();
    jacobi_eigh_lds<d>(L.Ms, L.Vs, nullptr, lane);
    if (lane < d) L.lam[lane] = 1.0 / sqrt(L.Ms[sys_index<d>(lane, lane)].re);
    FBX_WAVE_SYNC();
    const Blk o = reconstruct_blk<d>(L.Vs, L.lam, lane);
    blk_store<d, d>(dst, lane, o);
    FBX_WAVE_SYNC();
}

// ---- CPTP Kraus sets (BCSZ): out[b][k] = G_k S^{-1/2}
template <int d>
__global__ void __launch_bounds__(64)
random_kraus_kernel(long long B, int K, unsigned long long seed, long long first_item, cplx* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    RandLds<d> L; L.carve(smem);
    cplx* G = (cplx*)(smem + ((RandLds<d>::bytes() + 15) & ~(size_t)15));       // [K][d][d]
    const int lane = threadIdx.x;
    for (long long b = blockIdx.x; b < B; b += gridDim.x) {
        FBX_WAVE_SYNC();
        for (int idx = lane; idx < K * d * d; idx += 64) G[idx] = ginibre_entry(seed, first_item + b, (uint32_t)idx, 0u);
        FBX_WAVE_SYNC();
        if (lane < d * d) {                                  // S[r][c] = sum_k sum_j conj(G_k[j][r]) G_k[j][c]
            const int r = lane / d, c = lane % d;
            double sr = 0.0, si = 0.0;
            for (int k = 0; k < K; ++k)
#pragma unroll
                for (int j = 0; j < d; ++j) {
                    const cplx x = G[(k * d + j) * d + r], y = G[(k * d + j) * d + c];
                    sr += x.re * y.re + x.im * y.im; si += x.re * y.im - x.im * y.re;
                }
            cplx s; s.re = sr; s.im = si;
            L.A[lane] = s;
        }
        FBX_WAVE_SYNC();
        herm_inv_sqrt<d>(L.A, L.W, L, lane);                 // W = S^{-1/2}
        for (int idx = lane; idx < K * d * d; idx += 64) {
            const int k = idx / (d * d), r = (idx / d) % d, c = idx % d;
            double orr = 0.0, oi = 0.0;
#pragma unroll
            for (int j = 0; j < d; ++j) {
                const cplx g = G[(k * d + r) * d + j], w = L.W[j * d + c];
                orr += g.re * w.re - g.im * w.im; oi += g.re * w.im + g.im * w.re;
            }
            cplx o; o.re = orr; o.im = oi;
            out[(size_t)b * K * d * d + idx] = o;
        }
    }
}

// ---- Haar unitaries / Haar states / Ginibre and Bures states, one wavefront per item
template <int d>
__global__ void __launch_bounds__(64)
random_dxd_kernel(int kind, long long B, int rank, unsigned long long seed, long long first_item, cplx* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    RandLds<d> L; L.carve(smem);
    const int lane = threadIdx.x;
    const int r = lane / d, c = lane % d;
    for (long long b = blockIdx.x; b < B; b += gridDim.x) {
        const long long item = first_item + b;
        FBX_WAVE_SYNC();
        if (kind == FBX_RAND_UNITARY || kind == FBX_RAND_STATE_VECTOR || kind == FBX_RAND_BURES_STATE) {
            // the unitary's Ginibre matrix: stream tag 0 for the plain kinds, tag 1 for Bures (A is drawn first)
            if (lane < d * d) L.U[lane] = ginibre_entry(seed, item, (uint32_t)lane, kind == FBX_RAND_BURES_STATE ? 1u : 0u);
            FBX_WAVE_SYNC();
            haar_q_factor<d>(L.U, lane);
        }
        if (kind == FBX_RAND_UNITARY) {
            if (lane < d * d) out[(size_t)b * d * d + lane] = L.U[lane];
            continue;
        }
        if (kind == FBX_RAND_STATE_VECTOR) {
            if (lane < d) out[(size_t)b * d + lane] = L.U[lane * d];
            continue;
        }
        // M = A A^H with A = Ginibre(d, rank) (rank = d for Bures)
        const int kk = kind == FBX_RAND_BURES_STATE ? d : rank;
        cplx mm; mm.re = 0.0; mm.im = 0.0;
        if (lane < d * d) {
            for (int j = 0; j < kk; ++j) {
                const cplx x = ginibre_entry(seed, item, (uint32_t)(r * kk + j), 0u);
                const cplx y = ginibre_entry(seed, item, (uint32_t)(c * kk + j), 0u);
                mm.re += x.re * y.re + x.im * y.im; mm.im += x.im * y.re - x.re * y.im;      // x conj(y)
            }
            if (r == c) mm.im = 0.0;
            L.A[lane] = mm;
        }
        FBX_WAVE_SYNC();
        if (kind == FBX_RAND_BURES_STATE) {                  // P = (1 + U) M (1 + U)^H
            if (lane < d * d) { cplx w = L.U[lane]; if (r == c) w.re += 1.0; L.W[lane] = w; }
            FBX_WAVE_SYNC();
            if (lane < d * d) {                              // T = W M
                double tr = 0.0, ti = 0.0;
#pragma unroll
                for (int j = 0; j < d; ++j) { const cplx w = L.W[r * d + j], m = L.A[j * d + c]; tr += w.re * m.re - w.im * m.im; ti += w.re * m.im + w.im * m.re; }
                cplx t; t.re = tr; t.im = ti; L.T[lane] = t;
            }
            FBX_WAVE_SYNC();
            if (lane < d * d) {                              // P = T W^H
                double pr = 0.0, pi = 0.0;
#pragma unroll
                for (int j = 0; j < d; ++j) { const cplx t = L.T[r * d + j], w = L.W[c * d + j]; pr += t.re * w.re + t.im * w.im; pi += t.im * w.re - t.re * w.im; }
                mm.re = pr; mm.im = r == c ? 0.0 : pi;
            }
        }
        double tr = (lane < d * d && r == c) ? mm.re : 0.0;
        tr = wave_sum(tr);
        if (lane < d * d) { cplx o; o.re = mm.re / tr; o.im = mm.im / tr; out[(size_t)b * d * d + lane] = o; }
    }
}

}  // namespace fbx

using namespace fbx;

extern "C" {

int fbx_random_operators_dev(int kind, int dim, int cols_or_rank, int64_t B, uint64_t seed, int64_t first_item,
                             double* d_out) {
    FBX_REQUIRE(kind >= FBX_RAND_GINIBRE && kind <= FBX_RAND_BURES_STATE, "fbx_random_operators: bad kind");
    FBX_REQUIRE(B >= 0 && (B == 0 || d_out), "fbx_random_operators: bad batch / NULL buffer");
    FBX_REQUIRE(first_item >= 0, "fbx_random_operators: negative first_item");
    if (kind == FBX_RAND_GINIBRE) {
        FBX_REQUIRE(dim >= 1 && cols_or_rank >= 1 && (long long)dim * cols_or_rank <= (1ll << 30),
                    "fbx_random_operators: Ginibre shape out of range");
    } else {
        FBX_REQUIRE(dim == 2 || dim == 4 || dim == 8, "fbx_random_operators: dim must be 2, 4 or 8 (1..3 qubits)");
        if (kind == FBX_RAND_GINIBRE_STATE) {
            FBX_REQUIRE(cols_or_rank >= 1, "fbx_random_operators: rank must be positive");
            FBX_REQUIRE(cols_or_rank <= dim, "The rank of the state matrix cannot exceed the dimension.");
        }
    }
    int rc = ensure_device();
    if (rc) return rc;
    if (B == 0) return FBX_OK;
    if (kind == FBX_RAND_GINIBRE) {
        const long long total = (long long)B * dim * cols_or_rank, want = (total + 255) / 256;
        hipLaunchKernelGGL(ginibre_kernel, dim3((unsigned)(want < 256 * 32 ? want : 256 * 32)), dim3(256), 0, stream(),
                           (long long)B, dim * cols_or_rank, (unsigned long long)seed, (long long)first_item, 0u, (cplx*)d_out);
    } else {
        const unsigned grid = (unsigned)(B < 256 * 32 ? B : 256 * 32);
        if (dim == 2) hipLaunchKernelGGL(random_dxd_kernel<2>, dim3(grid), dim3(64), RandLds<2>::bytes(), stream(), kind, (long long)B, cols_or_rank, (unsigned long long)seed, (long long)first_item, (cplx*)d_out);
        else if (dim == 4) hipLaunchKernelGGL(random_dxd_kernel<4>, dim3(grid), dim3(64), RandLds<4>::bytes(), stream(), kind, (long long)B, cols_or_rank, (unsigned long long)seed, (long long)first_item, (cplx*)d_out);
        else hipLaunchKernelGGL(random_dxd_kernel<8>, dim3(grid), dim3(64), RandLds<8>::bytes(), stream(), kind, (long long)B, cols_or_rank, (unsigned long long)seed, (long long)first_item, (cplx*)d_out);
    }
    FBX_HIP(hipGetLastError());
    return FBX_OK;
}

int fbx_random_kraus_dev(int n_qubits, int64_t B, int K, uint64_t seed, int64_t first_item, double* d_kraus_out) {
    FBX_REQUIRE(n_qubits >= 1 && n_qubits <= 3, "fbx_random_kraus: n_qubits must be 1..3");
    FBX_REQUIRE(K >= 1 && K <= 64, "fbx_random_kraus: 1 <= K <= 64 Kraus operators");
    FBX_REQUIRE(B >= 0 && (B == 0 || d_kraus_out), "fbx_random_kraus: bad batch / NULL buffer");
    FBX_REQUIRE(first_item >= 0, "fbx_random_kraus: negative first_item");
    int rc = ensure_device();
    if (rc) return rc;
    if (B == 0) return FBX_OK;
    const int d = 1 << n_qubits;
    const unsigned grid = (unsigned)(B < 256 * 32 ? B : 256 * 32);
    const size_t g = sizeof(cplx) * (size_t)K * d * d + 16;
    if (d == 2) hipLaunchKernelGGL(random_kraus_kernel<2>, dim3(grid), dim3(64), RandLds<2>::bytes() + g, stream(), (long long)B, K, (unsigned long long)seed, (long long)first_item, (cplx*)d_kraus_out);
    else if (d == 4) hipLaunchKernelGGL(random_kraus_kernel<4>, dim3(grid), dim3(64), RandLds<4>::bytes() + g, stream(), (long long)B, K, (unsigned long long)seed, (long long)first_item, (cplx*)d_kraus_out);
    else {
        FBX_HIP(hipFuncSetAttribute((const void*)random_kraus_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(RandLds<8>::bytes() + g)));
        hipLaunchKernelGGL(random_kraus_kernel<8>, dim3(grid), dim3(64), RandLds<8>::bytes() + g, stream(), (long long)B, K, (unsigned long long)seed, (long long)first_item, (cplx*)d_kraus_out);
    }
    FBX_HIP(hipGetLastError());
    return FBX_OK;
}

static size_t random_out_doubles(int kind, int dim, int cols_or_rank) {
    if (kind == FBX_RAND_GINIBRE) return 2 * (size_t)dim * cols_or_rank;
    if (kind == FBX_RAND_STATE_VECTOR) return 2 * (size_t)dim;
    return 2 * (size_t)dim * dim;
}

int fbx_random_operators(int kind, int dim, int cols_or_rank, int64_t B, uint64_t seed, int64_t first_item, double* out) {
    FBX_REQUIRE(B >= 0 && (B == 0 || out), "fbx_random_operators: bad batch / NULL buffer");
    int rc = ensure_device();
    if (rc) return rc;
    if (B == 0) return FBX_OK;
    FBX_REQUIRE(kind >= FBX_RAND_GINIBRE && kind <= FBX_RAND_BURES_STATE && dim >= 1 && cols_or_rank >= (kind == FBX_RAND_GINIBRE || kind == FBX_RAND_GINIBRE_STATE ? 1 : 0),
                "fbx_random_operators: bad kind / shape");
    const size_t n = random_out_doubles(kind, dim, cols_or_rank) * (size_t)B;
    DevBuf d;
    if ((rc = d.alloc(sizeof(double) * n))) return rc;
    rc = fbx_random_operators_dev(kind, dim, cols_or_rank, B, seed, first_item, d.as<double>());
    if (rc) return rc;
    FBX_HIP(hipMemcpyAsync(out, d.p, sizeof(double) * n, hipMemcpyDeviceToHost, stream()));
    FBX_HIP(hipStreamSynchronize(stream()));
    return FBX_OK;
}

int fbx_random_kraus(int n_qubits, int64_t B, int K, uint64_t seed, int64_t first_item, double* kraus_out) {
    FBX_REQUIRE(B >= 0 && (B == 0 || kraus_out), "fbx_random_kraus: bad batch / NULL buffer");
    FBX_REQUIRE(n_qubits >= 1 && n_qubits <= 3 && K >= 1 && K <= 64, "fbx_random_kraus: n_qubits must be 1..3, 1 <= K <= 64");
    int rc = ensure_device();
    if (rc) return rc;
    if (B == 0) return FBX_OK;
    const size_t n = 2 * (size_t)B * K * (1u << (2 * n_qubits));
    DevBuf d;
    if ((rc = d.alloc(sizeof(double) * n))) return rc;
    rc = fbx_random_kraus_dev(n_qubits, B, K, seed, first_item, d.as<double>());
    if (rc) return rc;
    FBX_HIP(hipMemcpyAsync(kraus_out, d.p, sizeof(double) * n, hipMemcpyDeviceToHost, stream()));
    FBX_HIP(hipStreamSynchronize(stream()));
    return FBX_OK;
}

}  // extern "C"
