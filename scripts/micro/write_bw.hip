// Pure streaming-write and copy bandwidth (diagnostic): what the HBM path sustains for the sweep's traffic mix.
// hipcc --offload-arch=gfx950 -O3 write_bw.hip -o write_bw
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v2d __attribute__((ext_vector_type(2)));
__global__ void fill(v2d* p, size_t n, int nt) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const v2d v = {1.0, 2.0};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += stride) {
        if (nt) __builtin_nontemporal_store(v, p + i); else p[i] = v;
    }
}
__global__ void copy(const v2d* a, v2d* b, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += stride) b[i] = a[i];
}
int main() {
    const size_t bytes = (size_t)12 << 30, n = bytes / 16;
    v2d *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int nt = 0; nt < 2; ++nt) for (int grid : {2048, 4096, 16384}) {
        float best = 1e9;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0); hipLaunchKernelGGL(fill, dim3(grid), dim3(256), 0, 0, a, n, nt); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("fill nt=%d grid=%d: %.3f ms  %.0f GB/s\n", nt, grid, best, bytes / best / 1e6);
    }
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0); hipLaunchKernelGGL(copy, dim3(8192), dim3(256), 0, 0, a, b, n); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("copy: %.3f ms  %.0f GB/s (read + write)\n", best, 2.0 * bytes / best / 1e6);
    return 0;
}
