// fbx_pgdb_lean.hip -- the two-wavefronts-per-SIMD form of the 2-qubit PGDB kernel (batches that put more than one reconstruction
// on a SIMD anyway: BASELINE configs[4], 8192 per GPU).  Its own translation unit so that it keeps the compiler's default
// instruction scheduling (fbx_pgdb_body.hpp).  Reference: tomography.py:542-633, operator_tools/project_superoperators.py:19-144.
#include "fbx_pgdb_body.hpp"

namespace fbx {

// The same reconstruction with the lean LDS layout (16.5 KB) and at most 256 registers: TWO wavefronts per
// SIMD, i.e. two dependent Jacobi chains interleaved on every SIMD -- for batches that put more than one
// reconstruction on a SIMD anyway (BASELINE configs[4]: 8192 per GPU).  Results are bit-identical to
// pgdb_kernel's (same arithmetic; only where operands are kept differs).
// WAVES reconstructions (wavefronts) per workgroup.  WAVES = 1: the Bloch matrix is read through L2.  WAVES = 4: the wavefronts
// share ONE LDS copy of Ct[S][D] at the start of the segment (4.6 KB for the 36-state design) -- the table every prediction /
// gradient product walks -- at the price of a workgroup that holds its LDS until its slowest reconstruction has finished
// (the fixed-iteration mode, whose reconstructions take similar times, uses it; a per-wavefront copy costs an eighth wavefront
// per CU: 4.5 % slower).  The only workgroup barrier is the one that publishes the copy.
template <int NQ, int MAXJ, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(2, 2)))
pgdb_lean_kernel(DesignDev des, long long B, const double* __restrict__ expect,
                 const double* __restrict__ counts, int trace_preserving, int mode, int max_iters,
                 double* __restrict__ choi_out, int* __restrict__ iters_out,
                 int* __restrict__ dykstra_out, int* __restrict__ backtracks_out,
                 double* __restrict__ cost_out, int* __restrict__ work_out,
                 long long* __restrict__ phase_out, cplx* __restrict__ basis_scratch, int basis_cap,
                 double* __restrict__ ncounts, int* __restrict__ trace_out, int trace_iters, int wave_lds_bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int D = 1 << (2 * NQ);
    if constexpr (WAVES == 1) {
        pgdb_body<NQ, MAXJ, true>(smem, nullptr, blockIdx.x, des, B, expect, counts, trace_preserving, mode, max_iters, choi_out, iters_out,
                                  dykstra_out, backtracks_out, cost_out, work_out, phase_out, basis_scratch, basis_cap,
                                  ncounts ? ncounts + (size_t)blockIdx.x * 2 * MAXJ * 64 : nullptr, trace_out, trace_iters);
    } else {
        double* ct = reinterpret_cast<double*>(smem);
        for (int idx = threadIdx.x; idx < des.S * D; idx += 64 * WAVES) ct[idx] = des.Ct[idx];
        __syncthreads();
        const int wave = threadIdx.x >> 6;
        const long long item = (long long)blockIdx.x * WAVES + wave;
        if (item >= B) return;
        const size_t ct_bytes = (sizeof(double) * (size_t)des.S * D + 15) & ~(size_t)15;
        pgdb_body<NQ, MAXJ, true, true>(smem + ct_bytes + (size_t)wave * wave_lds_bytes, ct, item, des, B, expect, counts, trace_preserving, mode,
                                  max_iters, choi_out, iters_out, dykstra_out, backtracks_out, cost_out, work_out, phase_out,
                                  basis_scratch, basis_cap, ncounts ? ncounts + (size_t)item * 2 * MAXJ * 64 : nullptr, trace_out, trace_iters);
    }
}


static constexpr int LEAN_WAVES = 4;       // wavefronts per workgroup of the shared-table experiment

template <int MAXJ>
static int lean_launch(size_t lds, size_t wave_lds, bool shared_table, hipStream_t st, const PgdbLaunch& a) {
#define FBX_LEAN_ARGS a.dev, a.nb, a.e, a.c, a.tp, a.mode, a.max_iters, a.choi, a.it, a.dy, a.bt, a.cost, a.sw, a.phase, a.basis, a.basis_cap, \
                      a.ncounts, a.trace, a.trace_iters, (int)wave_lds
#if FBX_LEAN_SHARED_TABLE
    if (shared_table) {
        FBX_HIP(hipFuncSetAttribute((const void*)pgdb_lean_kernel<2, MAXJ, LEAN_WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((pgdb_lean_kernel<2, MAXJ, LEAN_WAVES>), dim3((unsigned)((a.nb + LEAN_WAVES - 1) / LEAN_WAVES)), dim3(64 * LEAN_WAVES), lds, st, FBX_LEAN_ARGS);
        return FBX_OK;
    }
#endif
    (void)shared_table;
    FBX_HIP(hipFuncSetAttribute((const void*)pgdb_lean_kernel<2, MAXJ, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((pgdb_lean_kernel<2, MAXJ, 1>), dim3((unsigned)a.nb), dim3(64), lds, st, FBX_LEAN_ARGS);
#undef FBX_LEAN_ARGS
    return FBX_OK;
}

size_t pgdb_lean_lds(int maxj, int S, bool shared_table, size_t* wave_lds) {
    const size_t w = (PgdbLds<2, true>::bytes(S, 64 * maxj) + 15) & ~(size_t)15;
    *wave_lds = w;
    if (!shared_table) return w;
    return ((sizeof(double) * (size_t)S * 16 + 15) & ~(size_t)15) + LEAN_WAVES * w;
}

int pgdb_lean_launch(int maxj, size_t lds, size_t wave_lds, bool shared_table, hipStream_t st, const PgdbLaunch& a) {
    if (maxj == 4) return lean_launch<4>(lds, wave_lds, shared_table, st, a);
    if (maxj == 9) return lean_launch<9>(lds, wave_lds, shared_table, st, a);
    return lean_launch<16>(lds, wave_lds, shared_table, st, a);
}

}  // namespace fbx
