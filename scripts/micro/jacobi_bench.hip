// Micro-benchmark of the in-LDS Jacobi eigensolver (diagnostics; not part of the library).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../forest-benchmarking_amd/csrc jacobi_bench.hip -o jacobi_bench
#include "fbx_eigh.hpp"
#ifdef FBX_JACOBI_CHAIN_FIRST
#include "jacobi_chain_first.hpp"      // round-3 experiment (measured, not adopted)
#endif
#ifdef FBX_JACOBI_H2
#include "jacobi_h2.hpp"               // round-5 experiment: Hermitian symmetry with two workers per upper block
#endif
#ifdef FBX_JACOBI_ALLREG
#include "jacobi_allreg.hpp"           // round-5 experiment: matrix AND eigenvector blocks in registers, exchange through DPP + ds_bpermute
#endif
#ifdef FBX_JACOBI_VDPP
#include "jacobi_vdpp.hpp"             // round-5 experiment: eigenvector exchange through DPP instead of LDS
#endif
#ifdef FBX_JACOBI_PUB16
#include "jacobi_pub16.hpp"            // round-5 experiment: rotations published through LDS by the lanes that hold next round's pivots
#endif
#ifdef FBX_JACOBI_REGPIVOT
#include "jacobi_regpivot.hpp"         // round-5 experiment: next-round pivots through registers (measured, not adopted)
#endif
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
namespace fbx { void set_error(const std::string&) {} int hip_fail(hipError_t, const char*, const char*, int) { return 2; } hipStream_t stream() { return 0; } int ensure_device() { return 0; } }
using namespace fbx;

template <int N>
__global__ void __launch_bounds__(64) k_eigh(const double* A, double* W, double* Vout, long long* cyc, int* sw, int reps) {
    constexpr int LD = N + 1;
    __shared__ cplx M[sys_elems<N>()];
    __shared__ cplx V[sys_elems<N>()];
    __shared__ JRec rot[N];
    const int lane = threadIdx.x, item = blockIdx.x;
    long long total = 0; int sweeps = 0;
    for (int rep = 0; rep < reps; ++rep) {
        for (int idx = lane; idx < N * N; idx += 64) {
            cplx c; c.re = A[(item * N * N + idx) * 2]; c.im = A[(item * N * N + idx) * 2 + 1];
            M[sys_index<N>(idx / N, idx % N)] = c;
        }
        __syncthreads();
        long long t0 = __builtin_readcyclecounter();
#ifdef FBX_JACOBI_CHAIN_FIRST
        sweeps += jacobi_eigh_wave_chain_first<N>(M, V, lane, true);
#elif defined(FBX_JACOBI_REGPIVOT)
        sweeps += jacobi_eigh_wave_regpivot<N>(M, V, lane, true);
#elif defined(FBX_JACOBI_PUB16)
        sweeps += jacobi_eigh_wave_pub16<N>(M, V, lane, true);
#elif defined(FBX_JACOBI_H2)
        sweeps += jacobi_eigh_wave_h2<N>(M, V, lane, true);
#elif defined(FBX_JACOBI_ALLREG)
        sweeps += jacobi_eigh_wave_allreg<N>(M, V, lane, true);
#elif defined(FBX_JACOBI_VDPP)
        sweeps += jacobi_eigh_wave_vdpp<N>(M, V, lane, true);
#else
        sweeps += jacobi_eigh_lds<N>(M, V, rot, lane);
#endif
        total += __builtin_readcyclecounter() - t0;
        __syncthreads();
    }
    if (lane < N) W[item * N + lane] = M[sys_index<N>(lane, lane)].re;
    for (int idx = lane; idx < N * N; idx += 64) {
        cplx c = V[sys_index<N>(idx / N, idx % N)];
        Vout[(item * N * N + idx) * 2] = c.re; Vout[(item * N * N + idx) * 2 + 1] = c.im;
    }
    if (lane == 0) { cyc[item] = total; sw[item] = sweeps; }
}

// EXPERIMENT (not in the library): two wavefronts per matrix -- wave 0 the matrix, wave 1 the eigenvectors
template <int N>
__device__ int jacobi_eigh_2w(cplx* Ms, cplx* Vs, volatile int* flag, int tid, bool init_identity = true) {
    constexpr int NB = N / 2, LS = NB * NB;
    const int role = uniform(tid >> 6), lane = tid & 63;
    const bool act = lane < LS;
    const int I = act ? lane / NB : 0, J = act ? lane % NB : 0;
    const int me = act ? lane : 0;
    int w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int sa = jacobi_seat<N>(2 * I + (e >> 1)), sb = jacobi_seat<N>(2 * J + (e & 1));
        w[e] = role == 0 ? ((sa & 1) * 2 + (sb & 1)) * LS + (sa >> 1) * NB + (sb >> 1)
                         : ((e >> 1) * 2 + (sb & 1)) * LS + I * NB + (sb >> 1);
    }
    const int dI = I * NB + I, dJ = J * NB + J;
    if (role == 1 && act && init_identity) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            cplx v; v.re = (2 * I + (e >> 1) == 2 * J + (e & 1)) ? 1.0 : 0.0; v.im = 0.0;
            Vs[e * LS + me] = v;
        }
    }
    int sweep = 0;
    for (;; ++sweep) {
        if (role == 0) {
            double o2 = 0.0, n2 = 0.0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const cplx v = Ms[e * LS + me];
                const double a2 = v.re * v.re + v.im * v.im;
                n2 += a2;
                if (!(I == J && (e == 0 || e == 3))) o2 += a2;
            }
            if (!act) { o2 = 0.0; n2 = 0.0; }
            o2 = wave_sum(o2); n2 = wave_sum(n2);
            if (lane == 0) *flag = (sweep < FBX_JACOBI_MAX_SWEEPS && o2 > FBX_JACOBI_TOL2 * n2) ? 1 : 0;
        }
        __syncthreads();
        const int go = uniform(*flag);
        if (!go) break;
        if (role == 0) {
            for (int r = 0; r < N - 1; ++r) {
                const double aI = Ms[0 * LS + dI].re, dI_ = Ms[3 * LS + dI].re;
                const cplx bI = Ms[1 * LS + dI];
                const double aJ = Ms[0 * LS + dJ].re, dJ_ = Ms[3 * LS + dJ].re;
                const cplx bJ = Ms[1 * LS + dJ];
                cplx m00 = Ms[0 * LS + me], m01 = Ms[1 * LS + me];
                cplx m10 = Ms[2 * LS + me], m11 = Ms[3 * LS + me];
                __syncthreads();
                const JRot rI = jacobi_rotation(aI, dI_, bI.re, bI.im);
                const JRot rJ = jacobi_rotation(aJ, dJ_, bJ.re, bJ.im);
                jacobi_apply_m(rI.c, rI.sr, rI.si, rJ.c, rJ.sr, rJ.si, m00, m01, m10, m11);
                if (I == J) { m01.re = m01.im = 0.0; m10.re = m10.im = 0.0; m00.im = 0.0; m11.im = 0.0; }
                if (act) { Ms[w[0]] = m00; Ms[w[1]] = m01; Ms[w[2]] = m10; Ms[w[3]] = m11; }
                __syncthreads();
            }
        } else {
            for (int r = 0; r < N - 1; ++r) {
                const double aJ = Ms[0 * LS + dJ].re, dJ_ = Ms[3 * LS + dJ].re;
                const cplx bJ = Ms[1 * LS + dJ];
                cplx v0p = Vs[0 * LS + me], v0q = Vs[1 * LS + me];
                cplx v1p = Vs[2 * LS + me], v1q = Vs[3 * LS + me];
                __syncthreads();
                const JRot rJ = jacobi_rotation(aJ, dJ_, bJ.re, bJ.im);
                jacobi_apply_v(rJ.c, rJ.sr, rJ.si, v0p, v0q, v1p, v1q);
                if (act) { Vs[w[0]] = v0p; Vs[w[1]] = v0q; Vs[w[2]] = v1p; Vs[w[3]] = v1q; }
                __syncthreads();
            }
        }
    }
    return sweep;
}

template <int N>
__global__ void __launch_bounds__(128) k_eigh2w(const double* A, double* W, double* Vout, long long* cyc, int* sw, int reps, int* simd) {
    __shared__ cplx M[sys_elems<N>()];
    __shared__ cplx V[sys_elems<N>()];
    __shared__ int flag;
    const int tid = threadIdx.x, item = blockIdx.x;
    if ((tid & 63) == 0) simd[item * 2 + (tid >> 6)] = __builtin_amdgcn_s_getreg(((16 - 1) << 11) | (0 << 6) | 4);
    long long total = 0; int sweeps = 0;
    for (int rep = 0; rep < reps; ++rep) {
        for (int idx = tid; idx < N * N; idx += 128) {
            cplx c; c.re = A[(item * N * N + idx) * 2]; c.im = A[(item * N * N + idx) * 2 + 1];
            M[sys_index<N>(idx / N, idx % N)] = c;
        }
        __syncthreads();
        long long t0 = __builtin_readcyclecounter();
        sweeps += jacobi_eigh_2w<N>(M, V, &flag, tid);
        total += __builtin_readcyclecounter() - t0;
        __syncthreads();
    }
    if (tid < N) W[item * N + tid] = M[sys_index<N>(tid, tid)].re;
    for (int idx = tid; idx < N * N; idx += 128) {
        cplx c = V[sys_index<N>(idx / N, idx % N)];
        Vout[(item * N * N + idx) * 2] = c.re; Vout[(item * N * N + idx) * 2 + 1] = c.im;
    }
    if (tid == 0) { cyc[item] = total; sw[item] = sweeps; }
}

int main(int argc, char** argv) {
    const bool two = argc > 2 && atoi(argv[2]) == 2;
    const int pad = argc > 3 ? atoi(argv[3]) : 0;      // extra dynamic LDS per block: 31744 / 11264 / 4400 / 1024 -> 1 / 2 / 3 / 4 waves per SIMD
    const int N = 16, B = argc > 1 ? atoi(argv[1]) : 1024, reps = 20;
    std::vector<double> A((size_t)B * N * N * 2);
    srand(1);
    for (int b = 0; b < B; ++b) {
        std::vector<double> g(N * N * 2);
        for (auto& x : g) x = (rand() / (double)RAND_MAX) - 0.5;
        for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) {
            A[((size_t)b * N * N + i * N + j) * 2] = g[(i * N + j) * 2] + g[(j * N + i) * 2];
            A[((size_t)b * N * N + i * N + j) * 2 + 1] = g[(i * N + j) * 2 + 1] - g[(j * N + i) * 2 + 1];
        }
    }
    double *dA, *dW, *dV; long long* dc; int* ds;
    hipMalloc(&dA, A.size() * 8); hipMalloc(&dW, B * N * 8); hipMalloc(&dV, A.size() * 8);
    hipMalloc(&dc, B * 8); hipMalloc(&ds, B * 4);
    int* dsimd; hipMalloc(&dsimd, B * 8);
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 2; ++it) {
        hipEventRecord(e0);
        if (two) hipLaunchKernelGGL(k_eigh2w<N>, dim3(B), dim3(128), 0, 0, dA, dW, dV, dc, ds, reps, dsimd);
        else hipLaunchKernelGGL(k_eigh<N>, dim3(B), dim3(64), pad, 0, dA, dW, dV, dc, ds, reps);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> c(B); std::vector<int> s(B); std::vector<double> W(B * N), V(A.size());
    hipMemcpy(c.data(), dc, B * 8, hipMemcpyDeviceToHost); hipMemcpy(s.data(), ds, B * 4, hipMemcpyDeviceToHost);
    hipMemcpy(W.data(), dW, B * N * 8, hipMemcpyDeviceToHost); hipMemcpy(V.data(), dV, A.size() * 8, hipMemcpyDeviceToHost);
    double csum = 0, ssum = 0; for (int b = 0; b < B; ++b) { csum += c[b]; ssum += s[b]; }
    // residual check on item 0: || A V - V W ||
    double res = 0;
    for (int i = 0; i < N; ++i) for (int k = 0; k < N; ++k) {
        double re = 0, im = 0;
        for (int j = 0; j < N; ++j) {
            double ar = A[(i * N + j) * 2], ai = A[(i * N + j) * 2 + 1], vr = V[(j * N + k) * 2], vi = V[(j * N + k) * 2 + 1];
            re += ar * vr - ai * vi; im += ar * vi + ai * vr;
        }
        re -= V[(i * N + k) * 2] * W[k]; im -= V[(i * N + k) * 2 + 1] * W[k];
        res = fmax(res, sqrt(re * re + im * im));
    }
#ifdef FBX_JACOBI_SEGTIME
    long long seg[4]; hipMemcpyFromSymbol(seg, HIP_SYMBOL(fbx::g_seg), sizeof seg);
    double tot = seg[0] + seg[1] + seg[2] + seg[3];
    printf("segments (block 0, both launches): read+wait %.0f%%  rotations %.0f%%  update %.0f%%  write+sync %.0f%%  (cycles/round: %.0f %.0f %.0f %.0f)\n",
           100 * seg[0] / tot, 100 * seg[1] / tot, 100 * seg[2] / tot, 100 * seg[3] / tot,
           seg[0] / (2.0 * s[0] * 15), seg[1] / (2.0 * s[0] * 15), seg[2] / (2.0 * s[0] * 15), seg[3] / (2.0 * s[0] * 15));
#endif
    if (two) {
        std::vector<int> sid(B * 2); hipMemcpy(sid.data(), dsimd, B * 8, hipMemcpyDeviceToHost);
        int same = 0; for (int b = 0; b < B; ++b) same += (((sid[2 * b] >> 4) & 3) == ((sid[2 * b + 1] >> 4) & 3));
        printf("HW_ID item0: wave0 %04x wave1 %04x; items with both waves on the same SIMD: %d / %d\n", sid[0], sid[1], same, B);
    }
    printf("pad=%d ", pad);
    printf("B=%d reps=%d kernel %.3f ms; per eigh: %.0f cycles, %.2f sweeps, %.0f cycles/round; residual %.2e; eigh/s %.3e\n",
           B, reps, ms, csum / B / reps, ssum / B / reps, csum / ssum / (N - 1), res, B * reps / (ms * 1e-3));
    return 0;
}
