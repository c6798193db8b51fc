// LDS throughput micro-benchmark (diagnostics): bytes per clock and CU for b128 / b64 reads and writes, contiguous
// and with the strides of the 64 x 64 Jacobi layout, at 16 wavefronts per CU.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 lds_rate.hip -o lds_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));
template <int OP>
__global__ void __launch_bounds__(1024) k(double* out, long long* cyc, int iters, int stride, int active) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    d2* L2 = (d2*)smem; double* L1 = (double*)smem;
    const int t = threadIdx.x;
    for (int i = t; i < 8192; i += 1024) { d2 v = {1.0 * i, 2.0}; L2[i] = v; }
    __syncthreads();
    d2 acc = {0.0, 0.0}; double a1 = 0.0;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        const int base = (t * stride + it * 64) & 1023;
        asm volatile("" ::: "memory");
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (OP == 0) { d2 v = L2[e * 1024 + base]; acc += v; }                       // b128 read
            else if (OP == 1) { d2 v = {acc.x + e, a1}; L2[e * 1024 + base] = v; }      // b128 write
            else if (OP == 2) { a1 += L1[e * 1024 + base]; }                            // b64 read
            else if (OP == 3) { L1[e * 1024 + base] = a1 + e; }                         // b64 write
            else if (OP == 5) { if ((t & 63) < active) { d2 v = {acc.x + e, a1}; L2[e * 1024 + base] = v; } }   // b128 write, part of the lanes
            else if (OP == 6) { if (((t & 63) % 4) < active / 16) { d2 v = {acc.x + e, a1}; L2[e * 1024 + base] = v; } }   // scattered active lanes
            else if (OP == 4) { d2 v = L2[e * 1024 + base]; acc += v; d2 w = {acc.y, 1.0}; L2[((e + 1) & 7) * 1024 + base] = w; }  // mixed
        }
        if (OP == 1 || OP == 3 || OP == 5 || OP == 6) { acc.x += 1.0; a1 += 1.0; }
    }
    __syncthreads();
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 1024 + t] = acc.x + acc.y + a1 + L1[t];
    if (t == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int OP> void run(const char* name, int stride, int bytes, int active = 64) {
    double* o; long long* c; (void)hipMalloc(&o, 256 * 1024 * 8); (void)hipMalloc(&c, 256 * 8);
    const int iters = 200;
    (void)hipFuncSetAttribute((const void*)k<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 16);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(1024), 8192 * 16, 0, o, c, iters, stride, active);
    (void)hipDeviceSynchronize();
    long long h[256]; (void)hipMemcpy(h, c, sizeof h, hipMemcpyDeviceToHost);
    double cy = 0; for (int i = 0; i < 256; ++i) cy += h[i]; cy /= 256;
    const double total = (double)iters * 8 * 1024 * bytes * (OP == 4 ? 2 : 1);
    printf("%-12s stride %3d: %.1f B/clk/CU (%.1f cycles per wave-instruction)\n", name, stride, total / cy, cy / (iters * 8.0 * 16 * (OP == 4 ? 2 : 1)));
    (void)hipFree(o); (void)hipFree(c);
}
int main() {
    for (int stride : {1, 2, 32, 33}) {
        run<0>("read b128", stride, 16); run<1>("write b128", stride, 16);
        run<2>("read b64", stride, 8); run<3>("write b64", stride, 8); run<4>("rd+wr b128", stride, 16);
    }
    for (int active : {64, 48, 32, 16}) { printf("active lanes %d (contiguous / every 4th group): ", active); run<5>("wr b128 part", 1, 16, active); run<6>("wr b128 scat", 1, 16, active); }
    return 0;
}
