// Micro-benchmark of the in-LDS Jacobi eigensolver (diagnostics; not part of the library).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../forest-benchmarking_amd/csrc jacobi_bench.hip -o jacobi_bench
#include "fbx_eigh.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
namespace fbx { void set_error(const std::string&) {} int hip_fail(hipError_t, const char*, const char*, int) { return 2; } hipStream_t stream() { return 0; } int ensure_device() { return 0; } }
using namespace fbx;

template <int N>
__global__ void __launch_bounds__(64) k_eigh(const double* A, double* W, double* Vout, long long* cyc, int* sw, int reps) {
    constexpr int LD = N + 1;
    __shared__ cplx M[N * N];
    __shared__ cplx V[N * N];
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
        sweeps += jacobi_eigh_lds<N>(M, V, rot, lane);
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

int main(int argc, char** argv) {
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
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 2; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_eigh<N>, dim3(B), dim3(64), 0, 0, dA, dW, dV, dc, ds, reps);
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
    printf("B=%d reps=%d kernel %.3f ms; per eigh: %.0f cycles, %.2f sweeps, %.0f cycles/round; residual %.2e; eigh/s %.3e\n",
           B, reps, ms, csum / B / reps, ssum / B / reps, csum / ssum / (N - 1), res, B * reps / (ms * 1e-3));
    return 0;
}
