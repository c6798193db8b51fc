#include "fbx_common.hpp"
#include <cstdio>
#include <cmath>
#include <vector>
namespace fbx { void set_error(const std::string&) {} int hip_fail(hipError_t, const char*, const char*, int) { return 2; } hipStream_t stream() { return 0; } int ensure_device() { return 0; } }
__global__ void k(const double* x, double* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = fbx::fast_rsqrt(x[i]);
}
int main() {
    const int n = 1 << 20; std::vector<double> x(n), r(n);
    for (int i = 0; i < n; ++i) x[i] = pow(10.0, -290.0 + 580.0 * (i + 0.5) / n);
    double *dx, *dr; (void)hipMalloc(&dx, n * 8); (void)hipMalloc(&dr, n * 8);
    (void)hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dr, n);
    (void)hipMemcpy(r.data(), dr, n * 8, hipMemcpyDeviceToHost);
    double e1 = 0; int worst = 0;
    for (int i = 0; i < n; ++i) { double e = fabs(r[i] * sqrt(x[i]) - 1); if (!(e <= e1)) { e1 = e; worst = i; } }
    printf("fast_rsqrt max rel err %.3e at x=%.3e (value %.6e)\n", e1, x[worst], r[worst]);
    for (double t : {1e-290, 1e-200, 1e-100, 1e-30, 0.5, 1.0, 12.0, 1e10, 1e100, 1e200}) {
        int i = (int)((log10(t) + 290.0) / 580.0 * n); if (i >= n) i = n - 1;
        printf("  x=%.3e  err %.3e\n", x[i], fabs(r[i] * sqrt(x[i]) - 1));
    }
    return 0;
}
