#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double* x, double* raw, double* rcp, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { raw[i] = __builtin_amdgcn_rsq(x[i]); rcp[i] = __builtin_amdgcn_rcp(x[i]); }
}
int main() {
    const int n = 1 << 20; std::vector<double> x(n), r(n), c(n);
    for (int i = 0; i < n; ++i) x[i] = 0.25 + 3.75 * (i + 0.5) / n;
    double *dx, *dr, *dc; (void)hipMalloc(&dx, n * 8); (void)hipMalloc(&dr, n * 8); (void)hipMalloc(&dc, n * 8);
    (void)hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dr, dc, n);
    (void)hipMemcpy(r.data(), dr, n * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(c.data(), dc, n * 8, hipMemcpyDeviceToHost);
    double e1 = 0, e2 = 0;
    for (int i = 0; i < n; ++i) { e1 = fmax(e1, fabs(r[i] * sqrt(x[i]) - 1)); e2 = fmax(e2, fabs(c[i] * x[i] - 1)); }
    printf("v_rsq_f64 max rel err %.3e (2^%.1f); v_rcp_f64 max rel err %.3e (2^%.1f)\n", e1, log2(e1), e2, log2(e2));
    return 0;
}
