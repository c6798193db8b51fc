// fbx_pgdb_lean.hip -- the two-wavefronts-per-SIMD form of the 2-qubit PGDB kernel (batches that put more than one reconstruction
// on a SIMD anyway: BASELINE configs[4], 8192 per GPU).  Its own translation unit so that it keeps the compiler's default
// instruction scheduling (fbx_pgdb_body.hpp).  Reference: tomography.py:542-633, operator_tools/project_superoperators.py:19-144.
#include "fbx_pgdb_body.hpp"

namespace fbx {

// The same reconstruction with the lean LDS layout (19.4 KB for the 36-state design: eight wavefronts per CU) and at most 256
// registers: TWO wavefronts per SIMD, i.e. two dependent Jacobi chains interleaved on every SIMD.  Results are bit-identical to
// pgdb_kernel's (same arithmetic; only where operands are kept differs -- PgdbLds<NQ, true>).
template <int NQ, int MAXJ>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
pgdb_lean_kernel(DesignDev des, long long B, const double* __restrict__ expect,
                 const double* __restrict__ counts, int trace_preserving, int mode, int max_iters,
                 double* __restrict__ choi_out, int* __restrict__ iters_out,
                 int* __restrict__ dykstra_out, int* __restrict__ backtracks_out,
                 double* __restrict__ cost_out, int* __restrict__ work_out,
                 long long* __restrict__ phase_out, cplx* __restrict__ basis_scratch, int basis_cap,
                 double* __restrict__ ncounts, int* __restrict__ trace_out, int trace_iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // (MAXJ = 0, the streamed instantiation: the number of outcome slots per lane comes from the design)
    const size_t slots = MAXJ > 0 ? (size_t)MAXJ : (size_t)((des.m + 63) / 64);
    pgdb_body<NQ, MAXJ, true>(smem, blockIdx.x, des, B, expect, counts, trace_preserving, mode, max_iters, choi_out, iters_out,
                              dykstra_out, backtracks_out, cost_out, work_out, phase_out, basis_scratch, basis_cap,
                              ncounts + (size_t)blockIdx.x * 2 * slots * 64, trace_out, trace_iters);
}

// The same, in PIECES (launches of 2048 .. 32 768 reconstructions, i.e. 1 .. 16 per wave slot).  A launch of whole reconstructions
// ends when its slowest wave slot does: the items of a batch differ by +-20 % in time (max / mean 1.6), and with four items per
// slot the launch lasts 1.19 x the mean load (list scheduling of the measured per-item times, scripts/piece_study.py) -- the gap
// between 133 k/s at 8192 experiments and 155 k/s at 65 536.  Here 2048 persistent workgroups draw TICKETS from one counter;
// ticket e is piece e / nb (outer iterations [piece * W, (piece + 1) * W), the last piece open-ended) of item e % nb, so all
// first pieces are handed out before any second piece.  What a slot is stuck with at the end of the launch is a piece, not
// a reconstruction: 1.07 x the mean load with four pieces in the same model.  Between two pieces a reconstruction lives in
// its record (fbx_pgdb_body.hpp) and in its slice of the basis store; the piece that continues it may run on another XCD,
// whose L2 is not coherent with the producer's: the producer drains its stores, issues an agent-scope release fence (L2
// write-back) and publishes `piece + 1` in the item's progress flag; the consumer polls the flag (relaxed, s_sleep), issues an
// agent-scope acquire fence and reads.  A ticket's predecessor was drawn nb tickets earlier by a workgroup that is running, so the
// wait is almost never entered and cannot deadlock.  Results are bit-identical to the one-launch-per-reconstruction form.
// (TPC: the kind of projection at compile time -- fbx_pgdb_body.hpp)
template <int NQ, int MAXJ, int TPC>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
pgdb_lean_pieces_kernel(DesignDev des, long long B, const double* __restrict__ expect,
                        const double* __restrict__ counts, int trace_preserving, int mode, int max_iters,
                        double* __restrict__ choi_out, int* __restrict__ iters_out,
                        int* __restrict__ dykstra_out, int* __restrict__ backtracks_out,
                        double* __restrict__ cost_out, int* __restrict__ work_out,
                        long long* __restrict__ phase_out, cplx* __restrict__ basis_scratch, int basis_cap,
                        double* __restrict__ ncounts, int* __restrict__ trace_out, int trace_iters,
                        int pieces, int piece_iters, int* __restrict__ queue, int* __restrict__ flags, double* __restrict__ recs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    pgdb_pieces_run<NQ, MAXJ, true, TPC>(smem, des, B, expect, counts, trace_preserving, mode, max_iters, choi_out, iters_out, dykstra_out,
                                    backtracks_out, cost_out, work_out, phase_out, basis_scratch, basis_cap, ncounts, trace_out,
                                    trace_iters, pieces, piece_iters, queue, flags, recs);
}

template <int MAXJ, int TPC>
static int lean_pieces_launch(size_t lds, hipStream_t st, const PgdbLaunch& a) {
    FBX_HIP(hipFuncSetAttribute((const void*)pgdb_lean_pieces_kernel<2, MAXJ, TPC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    FBX_HIP(hipMemsetAsync(a.queue, 0, sizeof(int), st));
    FBX_HIP(hipMemsetAsync(a.flags, 0, sizeof(int) * (size_t)a.nb, st));
    const unsigned grid = (unsigned)(a.nb < 2048 ? a.nb : 2048);
    hipLaunchKernelGGL((pgdb_lean_pieces_kernel<2, MAXJ, TPC>), dim3(grid), dim3(64), lds, st, a.dev, a.nb, a.e, a.c, a.tp, a.mode, a.max_iters,
                       a.choi, a.it, a.dy, a.bt, a.cost, a.sw, a.phase, a.basis, a.basis_cap, a.ncounts, a.trace, a.trace_iters,
                       a.pieces, a.piece_iters, a.queue, a.flags, a.recs);
    return FBX_OK;
}

template <int MAXJ>
static int lean_launch(size_t lds, hipStream_t st, const PgdbLaunch& a) {
    if (a.pieces > 1) return a.tp ? lean_pieces_launch<MAXJ, 1>(lds, st, a) : lean_pieces_launch<MAXJ, 0>(lds, st, a);
    FBX_HIP(hipFuncSetAttribute((const void*)pgdb_lean_kernel<2, MAXJ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((pgdb_lean_kernel<2, MAXJ>), dim3((unsigned)a.nb), dim3(64), lds, st, a.dev, a.nb, a.e, a.c, a.tp, a.mode, a.max_iters,
                       a.choi, a.it, a.dy, a.bt, a.cost, a.sw, a.phase, a.basis, a.basis_cap, a.ncounts, a.trace, a.trace_iters);
    return FBX_OK;
}

// Designs beyond the register-resident instantiations (2 qubits: more than 1024 settings; 1 qubit: more than 256): the same
// kernel with MAXJ = 0 -- outcome slots streamed from HBM / L2 (fbx_pgdb_body.hpp, STREAM), whole reconstructions.
template <int NQ>
static int stream_launch(size_t lds, hipStream_t st, const PgdbLaunch& a) {
    FBX_HIP(hipFuncSetAttribute((const void*)pgdb_lean_kernel<NQ, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((pgdb_lean_kernel<NQ, 0>), dim3((unsigned)a.nb), dim3(64), lds, st, a.dev, a.nb, a.e, a.c, a.tp, a.mode, a.max_iters,
                       a.choi, a.it, a.dy, a.bt, a.cost, a.sw, a.phase, a.basis, a.basis_cap, a.ncounts, a.trace, a.trace_iters);
    return FBX_OK;
}
size_t pgdb_stream_lds(int nq, int S) {
    return ((nq == 1 ? PgdbLds<1, true>::bytes(S, 0) : PgdbLds<2, true>::bytes(S, 0)) + 15) & ~(size_t)15;
}
int pgdb_stream_launch(int nq, size_t lds, hipStream_t st, const PgdbLaunch& a) {
    return nq == 1 ? stream_launch<1>(lds, st, a) : stream_launch<2>(lds, st, a);
}

size_t pgdb_lean_lds(int maxj, int S) { return (PgdbLds<2, true>::bytes(S, 64 * maxj) + 15) & ~(size_t)15; }

// the transient copy of the Bloch table [S][16] and Rb (2 KB at the tail of Vs) share Ms + Vs
bool pgdb_lean_eligible(int S) {
    constexpr size_t room = 2 * sizeof(cplx) * sys_elems<16>() - sizeof(double) * 16 * 16;
    return sizeof(double) * 16 * (size_t)S <= room;
}

int pgdb_lean_launch(int maxj, size_t lds, hipStream_t st, const PgdbLaunch& a) {
    if (maxj == 4) return lean_launch<4>(lds, st, a);
    if (maxj == 9) return lean_launch<9>(lds, st, a);
    return lean_launch<16>(lds, st, a);
}

}  // namespace fbx
