// fbx_pgdb1.hip -- packed single-qubit PGDB process tomography: 64 reconstructions per wavefront, one per lane.
//
// What the reference's own tests and notebook run (tests/test_process_tomography.py:72-112,
// docs/examples/tomography_process.ipynb) and what single-qubit bootstraps are made of.  The wave-per-item kernel
// (pgdb_kernel<1, MAXJ>, fbx_pgdb.hip) keeps 4 of 64 lanes busy on a 4 x 4 Choi matrix; here a lane runs a whole
// reconstruction out of its registers (fbx_pgdb1_core.hpp) -- no LDS staging, no cross-lane traffic, no barriers.
//
//   * Design data (settings grouped by input state, Bloch rows) is shared by the batch: every index into it is
//     wave-uniform, its loads are scalar (s_load), its branches scalar.
//   * The only per-lane table is the normalised counts n+- [2 m]: LDS, one 8-byte column per lane (bank = lane,
//     conflict-free), 1 KiB x m per wavefront.
//   * Lanes diverge in their trip counts (outer iterations, Dykstra iterations, sweeps, halvings): a lane whose
//     reconstruction has finished writes it out and takes the NEXT item of the batch from a global counter
//     (persistent lanes), so a wavefront stays full until the batch runs dry instead of waiting for its slowest lane.
//   * Results do not depend on which lane ran an item: all arithmetic is per lane, in a fixed order.
//
// Replaces (file:line under forest/benchmarking/): pgdb_process_estimate tomography.py:542-594 with _cost / _grad_cost
// :597-633 and proj_choi_to_physical operator_tools/project_superoperators.py:87-144, for n_qubits = 1.
#include "fbx_common.hpp"
#include "fbx_pgdb1_core.hpp"

namespace fbx {

struct LaneCounts {
    double* tab;     // LDS [2 m][64], this lane's column
    __device__ __forceinline__ double plus(int g) const { return tab[(2 * g) * 64]; }
    __device__ __forceinline__ double minus(int g) const { return tab[(2 * g + 1) * 64]; }
};

// n+-[g] = counts * (1 +- e) / 2 / grand_total   (tomography.py:528-538), grouped order
__device__ __forceinline__ void p1_load_counts(const DesignDev& des, long long item, const double* __restrict__ expect,
                                               const double* __restrict__ counts, double* tab) {
    const int m = des.m;
    const double* e = expect + item * m;
    const double* c = counts + item * m;
    double tot = 0.0;
    for (int k = 0; k < m; ++k) tot += c[k];
    for (int g = 0; g < m; ++g) {
        const int k = des.order[g];
        const double plus = (1.0 + e[k]) / 2.0;
        tab[(2 * g) * 64] = (c[k] * plus) / tot;
        tab[(2 * g + 1) * 64] = (c[k] * (1.0 - plus)) / tot;
    }
}

__global__ void __launch_bounds__(64)
pgdb1_packed_kernel(DesignDev des, long long B, const double* __restrict__ expect, const double* __restrict__ counts,
                    int trace_preserving, int mode, int max_iters, double* __restrict__ choi_out,
                    int* __restrict__ iters_out, int* __restrict__ dykstra_out, int* __restrict__ backtracks_out,
                    double* __restrict__ cost_out, int* __restrict__ work_out, int* __restrict__ trace_out, int trace_iters,
                    unsigned long long* __restrict__ next_item) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    LaneCounts nt;
    nt.tab = reinterpret_cast<double*>(smem) + lane;
    long long item = (long long)blockIdx.x * 64 + lane;          // the first gridDim.x * 64 items are handed out statically
    bool active = item < B;
    P1State st;
    if (active) { p1_load_counts(des, item, expect, counts, nt.tab); p1_begin(des, nt, st); }
    while (active) {
        int dyk_this, bt_this;
        const int it_before = st.iters;
        const bool done = p1_outer_iteration(des, nt, st, trace_preserving != 0, mode, max_iters, dyk_this, bt_this);
        if (trace_out && st.iters > it_before && it_before < trace_iters) {
            int* tr = trace_out + ((size_t)item * trace_iters + it_before) * 2;
            tr[0] = dyk_this; tr[1] = bt_this;
        }
        if (done) {
            double* o = choi_out + item * 32;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    double xr, xi;
                    if (r == c) { xr = st.est.d[r]; xi = 0.0; }
                    else if (r < c) { xr = st.est.re[h4u(r, c)]; xi = st.est.im[h4u(r, c)]; }
                    else { xr = st.est.re[h4u(c, r)]; xi = -st.est.im[h4u(c, r)]; }
                    o[(r * 4 + c) * 2] = xr; o[(r * 4 + c) * 2 + 1] = xi;
                }
            if (iters_out) iters_out[item] = st.iters;
            if (dykstra_out) dykstra_out[item] = st.dyk;
            if (backtracks_out) backtracks_out[item] = st.backtracks;
            if (cost_out) cost_out[item] = st.new_cost;
            if (work_out) {       // Jacobi sweeps, eigenvalue terms rebuilt, full cost evaluations, power-sum reductions
                work_out[4 * item] = st.sweeps; work_out[4 * item + 1] = st.terms;
                work_out[4 * item + 2] = st.ls_full; work_out[4 * item + 3] = st.ls_sums;
            }
            item = (long long)atomicAdd(next_item, 1ull);
            active = item < B;
            if (active) { p1_load_counts(des, item, expect, counts, nt.tab); p1_begin(des, nt, st); }
        }
    }
}

__global__ void pgdb1_set_counter(unsigned long long* p, unsigned long long v) { *p = v; }

// Largest design the packed kernel takes: the counts table is 1 KiB x m of LDS per wavefront
constexpr int PGDB1_MAX_M = 64;

bool pgdb1_eligible(const fbx_design* des) { return des->dev.n == 1 && des->dev.m <= PGDB1_MAX_M; }

int pgdb1_dispatch(const fbx_design* des, int64_t B, const double* e, const double* c, int tp, int mode, int max_iters,
                   double* choi, int32_t* it, int32_t* dy, int32_t* bt, double* cost, int32_t* sw, const PgdbExtras& ex) {
    const size_t lds = sizeof(double) * 2 * (size_t)des->dev.m * 64;
    FBX_HIP(hipFuncSetAttribute((const void*)pgdb1_packed_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // wavefronts in flight: what the chip holds at once (LDS- and register-limited), the rest of the batch through the
    // item counter.  Two waves per SIMD is the register limit (<= 256 VGPRs); 160 KiB of LDS per CU the other.
    // (queried once per device and LDS size: hipGetDeviceProperties costs ~100 us, this dispatch is also the B = 1 path)
    static thread_local int cached_dev = -1, cached_epoch = -1, cached_cus = 0, cached_per_cu = 0;
    static thread_local size_t cached_lds = 0;
    const int dev = current_device();
    if (cached_dev != dev || cached_epoch != device_epoch() || cached_lds != lds) {
        hipDeviceProp_t prop;
        FBX_HIP(hipGetDeviceProperties(&prop, dev < 0 ? 0 : dev));
        int per_cu = 0;
        FBX_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pgdb1_packed_kernel, 64, lds));
        cached_cus = prop.multiProcessorCount; cached_per_cu = per_cu < 1 ? 1 : per_cu;
        cached_dev = dev; cached_epoch = device_epoch(); cached_lds = lds;
    }
    const long long resident = (long long)cached_per_cu * cached_cus;
    const long long want = (B + 63) / 64;
    const long long grid = want < resident ? want : resident;
    void* w = nullptr;
    { const int rc = workspace(WS_PGDB1_COUNTER, 256, &w); if (rc) return rc; }
    // (the pipelined host entry point runs two stages at once on two streams: each has its own counter)
    unsigned long long* counter = (unsigned long long*)w + (ex.ws_offset ? 8 : 0);
    hipStream_t st = ex.launch_stream ? ex.launch_stream : stream();
    hipLaunchKernelGGL(pgdb1_set_counter, dim3(1), dim3(1), 0, st, counter, (unsigned long long)(grid * 64));
    DesignDev d = des->dev;
    hipLaunchKernelGGL(pgdb1_packed_kernel, dim3((unsigned)grid), dim3(64), lds, st, d, (long long)B, e, c, tp, mode, max_iters,
                       choi, it, dy, bt, cost, sw, ex.trace, ex.trace_iters, counter);
    FBX_HIP(hipGetLastError());
    return FBX_OK;
}

}  // namespace fbx
