// cost of a double-precision log / division per wave (diagnostics)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP, int CH>
__global__ void __launch_bounds__(64) k(double* out, long long* cyc, int iters) {
    double x[CH]; for (int i = 0; i < CH; ++i) x[i] = 0.3 + threadIdx.x * 1e-3 + 0.01 * i;
    double acc = 0;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            if (OP == 0) acc += log(x[i]);
            else if (OP == 1) acc += 1.0 / x[i];
            else if (OP == 2) acc += __builtin_amdgcn_rcp(x[i]);
            x[i] += 1e-6;
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int OP, int CH> void run(const char* n, int blocks) {
    double* o; long long* c; (void)hipMalloc(&o, blocks * 64 * 8); (void)hipMalloc(&c, blocks * 8);
    hipLaunchKernelGGL((k<OP, CH>), dim3(blocks), dim3(64), 0, 0, o, c, 1000); (void)hipDeviceSynchronize();
    long long h; (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    printf("%-6s chains=%d blocks=%d: %.1f cycles per call per wave\n", n, CH, blocks, (double)h / (1000.0 * CH));
}
int main() { for (int b : {1024, 4096}) { run<0, 1>("log", b); run<0, 4>("log", b); run<0, 9>("log", b); run<1, 1>("div", b); run<1, 4>("div", b); } return 0; }
