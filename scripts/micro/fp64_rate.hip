// fp64 VALU issue/latency micro-benchmark (diagnostics).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int CHAINS, int OP>
__global__ void __launch_bounds__(64) k(double* out, long long* cyc, int iters) {
    double x[CHAINS];
    for (int i = 0; i < CHAINS; ++i) x[i] = 1.0 + threadIdx.x * 1e-3 + i;
    const double a = 1.0000001, b = 1e-9;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < CHAINS; ++i) {
                if (OP == 0) x[i] = fma(x[i], a, b);
                else if (OP == 1) x[i] = x[i] * a;
                else if (OP == 2) x[i] = x[i] + b;
                else if (OP == 3) x[i] = __builtin_amdgcn_rsq(x[i]) + 1.0;
                else if (OP == 4) x[i] = __builtin_amdgcn_rcp(x[i]) + 1.0;
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    double s = 0; for (int i = 0; i < CHAINS; ++i) s += x[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int CHAINS, int OP> void run(const char* name, int blocks) {
    double* o; long long* c; hipMalloc(&o, blocks * 64 * 8); hipMalloc(&c, blocks * 8);
    const int iters = 2000;
    hipLaunchKernelGGL((k<CHAINS, OP>), dim3(blocks), dim3(64), 0, 0, o, c, iters);
    hipDeviceSynchronize();
    long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    double per = (double)h / (iters * 8.0 * CHAINS);
    printf("%-8s chains=%d blocks=%d : %.2f cycles per wave-instruction (%.1f per dependent step)\n", name, CHAINS, blocks, per, per * CHAINS);
    hipFree(o); hipFree(c);
}
int main() {
    for (int blocks : {1024, 4096}) {
        run<1, 0>("fma", blocks); run<2, 0>("fma", blocks); run<4, 0>("fma", blocks); run<8, 0>("fma", blocks);
        run<1, 1>("mul", blocks); run<4, 1>("mul", blocks); run<8, 1>("mul", blocks);
        run<1, 2>("add", blocks); run<8, 2>("add", blocks);
        run<1, 3>("rsq+add", blocks); run<4, 3>("rsq+add", blocks);
        run<1, 4>("rcp+add", blocks); run<4, 4>("rcp+add", blocks);
    }
    return 0;
}
