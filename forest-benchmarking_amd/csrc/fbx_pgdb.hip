// fbx_pgdb.hip -- batched projected-gradient-descent-with-backtracking process tomography.
//
// One 64-lane wavefront owns one reconstruction for its whole life: the Choi estimate and
// the Dykstra state live in registers (one 2x2 block per lane), work matrices and the
// predicted-expectation tables in LDS; HBM is read once (expectations + counts) and written
// once (Choi + counters) -- plus the per-item store of Dykstra eigenvector bases (BasisStore,
// fbx_choi.hpp), working state in HBM/L2 that lets an outer iteration start its decompositions
// from the bases the previous one found.
//
// Replaces, for a batch that shares one design (file:line under forest/benchmarking/):
//   pgdb_process_estimate   tomography.py:542-594
//   _extract_from_results   tomography.py:494-539  (A never materialised: the Kronecker
//                           structure A_row = vec(rho_in (x) Pi^T)/d^2 is applied as
//                           T[s][i] = sum_j R_ij c_j(s), R = Pauli-Liouville form of E)
//   _cost / _grad_cost      tomography.py:597-633
//   proj_choi_to_physical   operator_tools/project_superoperators.py:87-144
#include "fbx_pgdb_body.hpp"
#include <cstdlib>

namespace fbx {

// The product kernel: one wavefront per SIMD (up to 512 registers, 39 KB of LDS) -- the fastest form while
// there are no more reconstructions in flight than SIMDs (B <= 1024 on 256 CUs).
template <int NQ, int MAXJ>
__global__ void __launch_bounds__(64)
pgdb_kernel(DesignDev des, long long B, const double* __restrict__ expect,
            const double* __restrict__ counts, int trace_preserving, int mode, int max_iters,
            double* __restrict__ choi_out, int* __restrict__ iters_out,
            int* __restrict__ dykstra_out, int* __restrict__ backtracks_out,
            double* __restrict__ cost_out, int* __restrict__ work_out,
            long long* __restrict__ phase_out, cplx* __restrict__ basis_scratch, int basis_cap,
            int* __restrict__ trace_out, int trace_iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    pgdb_body<NQ, MAXJ, false>(smem, blockIdx.x, des, B, expect, counts, trace_preserving, mode, max_iters, choi_out, iters_out,
                               dykstra_out, backtracks_out, cost_out, work_out, phase_out, basis_scratch, basis_cap, nullptr,
                               trace_out, trace_iters);
}

// The same kernel in PIECES (fbx_pgdb_lean.hip has the description): for launches that put more than one reconstruction on a
// SIMD but cannot use the two-waves kernel -- single-qubit designs on the wavefront-per-item kernel (1025 .. 8191 / 16 383
// experiments), 2-qubit designs with more than 50 input states.  1024 persistent workgroups.
template <int NQ, int MAXJ>
__global__ void __launch_bounds__(64)
pgdb_pieces_kernel(DesignDev des, long long B, const double* __restrict__ expect,
                   const double* __restrict__ counts, int trace_preserving, int mode, int max_iters,
                   double* __restrict__ choi_out, int* __restrict__ iters_out,
                   int* __restrict__ dykstra_out, int* __restrict__ backtracks_out,
                   double* __restrict__ cost_out, int* __restrict__ work_out,
                   long long* __restrict__ phase_out, cplx* __restrict__ basis_scratch, int basis_cap,
                   int* __restrict__ trace_out, int trace_iters,
                   int pieces, int piece_iters, int* __restrict__ queue, int* __restrict__ flags, double* __restrict__ recs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    pgdb_pieces_run<NQ, MAXJ, false>(smem, des, B, expect, counts, trace_preserving, mode, max_iters, choi_out, iters_out, dykstra_out,
                                     backtracks_out, cost_out, work_out, phase_out, basis_scratch, basis_cap, nullptr, trace_out,
                                     trace_iters, pieces, piece_iters, queue, flags, recs);
}

#ifdef FBX_DIAGNOSTICS
__global__ void debug_log_kernel(const double* x, double* out, long long n) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i < n) out[i] = fast_log_pos(x[i]);
}
#endif

// _cost / _grad_cost (tomography.py:597-633) as functions of their own: ONE evaluation of the negative log-likelihood and of its
// gradient at a given Choi matrix, with the very device functions the reconstruction kernels use -- Choi -> Pauli coefficients
// (choi_to_pauli_real), T = R C (predict_table), the clipped probabilities and fast_log_pos, the per-state weights through LDS
// atomics, R^G = -(W C^T) / d^2 and the inverse transform (pauli_real_to_choi_blk) -- so that the gradient, which the reconstruction
// never outputs, can be held against the oracle directly (tests/test_cost_grad_gpu.py).  `nvec` is the reference's `n`: row 2 k /
// 2 k + 1 = normalised +1 / -1 counts of result k.  Any number of settings (the outcome slots are walked from HBM).
template <int NQ>
__global__ void __launch_bounds__(64)
pgdb_cost_grad_kernel(DesignDev des, long long B, const double* __restrict__ nvec, const double* __restrict__ choi_in, double eps,
                      double* __restrict__ cost_out, double* __restrict__ grad_out) {
    constexpr int d = 1 << NQ, D = d * d, LD = D + 1, NB = D / 2, NACT = NB * NB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    const long long item = blockIdx.x;
    const int m = des.m, S = des.S;
    PgdbLds<NQ, false> L;
    L.carve(smem, S, 0);
    for (int idx = lane; idx < D * S; idx += 64) L.Cl[(idx % S) * D + idx / S] = des.C[idx];     // des.C is [D][S]
    Blk est = blk_zero();
    if (lane < NACT) {
        const int I = lane / NB, J = lane % NB;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = 2 * I + (e >> 1), col = 2 * J + (e & 1);
            const double* o = choi_in + ((item * D + row) * D + col) * 2;
            est.re[e] = o[0]; est.im[e] = o[1];
        }
    }
    FBX_WAVE_SYNC();
    blk_store<D, LD>(L.choi.Mw, lane, est);
    FBX_WAVE_SYNC();
    choi_to_pauli_real<NQ>(L.choi.Mw, L.Rb, lane);
    FBX_WAVE_SYNC();
    predict_table<NQ>(L.Rb, L.Cl, L.Test, S, lane);
    double* Wt = L.Tupd;                        // [S][D]
    for (int idx = lane; idx < D * S; idx += 64) Wt[idx] = 0.0;
    FBX_WAVE_SYNC();
    const double half_dd = 0.5 / (double)(d * d);
    const bool unit_coefs = des.unit_coefs != 0;
    const double* nv = nvec + item * 2 * m;
    double acc = 0.0;
    for (int g = lane; g < m; g += 64) {
        const uint32_t dw = des.sp[g];
        const int st = dw >> 16, p = dw & 0xffff, k = des.order[g];
        const double cf = unit_coefs ? 1.0 : des.coef[g];
        const double tr = L.Test[st * D], ex = cf * L.Test[st * D + p];
        double pp = (tr + ex) * half_dd, pm = (tr - ex) * half_dd;
        pp = pp < eps ? eps : pp; pm = pm < eps ? eps : pm;
        const double np_ = nv[2 * k], nm_ = nv[2 * k + 1];
        acc -= np_ * fast_log_pos(pp) + nm_ * fast_log_pos(pm);
        const double ep = np_ / pp, em = nm_ / pm;
        atomicAdd(&Wt[st * D], 0.5 * (ep + em));
        atomicAdd(&Wt[st * D + p], cf * 0.5 * (ep - em));
    }
    acc = uniform(wave_sum(acc));
    if (lane == 0 && cost_out) cost_out[item] = acc;
    FBX_WAVE_SYNC();
    if (!grad_out) return;
    {
        constexpr int JB = (D * D + 63) / 64;
        const int i = lane % D, j0 = (lane / D) * JB;
        if (j0 < D) {
            double a[JB];
#pragma unroll
            for (int r = 0; r < JB; ++r) a[r] = 0.0;
            for (int st = 0; st < S; ++st) {
                const double w = Wt[st * D + i];
#pragma unroll
                for (int r = 0; r < JB; ++r) a[r] = fma(w, L.Cl[st * D + j0 + r], a[r]);
            }
#pragma unroll
            for (int r = 0; r < JB; ++r) L.Rb[(j0 + r) * D + i] = -a[r] / (double)(d * d);
        }
    }
    FBX_WAVE_SYNC();
    const Blk grad = pauli_real_to_choi_blk<NQ>(L.Rb, L.choi.Mw, lane);
    if (lane < NACT) {
        const int I = lane / NB, J = lane % NB;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = 2 * I + (e >> 1), col = 2 * J + (e & 1);
            double* o = grad_out + ((item * D + row) * D + col) * 2;
            o[0] = grad.re[e]; o[1] = grad.im[e];
        }
    }
}

int pgdb3_cost_grad_launch(const fbx_design* des, int64_t B, const double* nvec, const double* choi, double eps, double* cost,
                           double* grad);     // fbx_pgdb3.hip

template <int NQ>
static int launch_cost_grad(const fbx_design* des, int64_t B, const double* nvec, const double* choi, double eps, double* cost,
                            double* grad) {
    const size_t lds = PgdbLds<NQ, false>::bytes(des->dev.S, 0);
    if (lds > 160 * 1024) {
        set_error("fbx_pgdb_cost_grad: too many distinct input states for the LDS-resident tables");
        return FBX_ERR_UNSUPPORTED;
    }
    FBX_HIP(hipFuncSetAttribute((const void*)pgdb_cost_grad_kernel<NQ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((pgdb_cost_grad_kernel<NQ>), dim3((unsigned)B), dim3(64), lds, stream(), des->dev, (long long)B, nvec, choi, eps,
                       cost, grad);
    FBX_HIP(hipGetLastError());
    return FBX_OK;
}

constexpr int BASIS_CAP = 32;              // Dykstra iterations per projection that get a stored basis

#ifdef FBX_DIAGNOSTICS
long long* g_phase_out = nullptr;          // diagnostics builds only: set by fbx_debug_set_phase_buffer (also read by fbx_pgdb3.hip)
#define FBX_PHASE_OUT(b0) (g_phase_out ? g_phase_out + (b0) * 8 : nullptr)
#else
#define FBX_PHASE_OUT(b0) ((long long*)nullptr)
#endif

template <int NQ, int MAXJ>
static int launch_pgdb(const fbx_design* des, int64_t B, const double* e, const double* c, int tp,
                       int mode, int max_iters, double* choi, int32_t* it, int32_t* dy, int32_t* bt,
                       double* cost, int32_t* sw, const PgdbExtras& ex) {
    const bool ls_reference = (mode & FBX_MODE_LS_REFERENCE) != 0;      // per call: the line search taken literally (include/fbx.h)
    mode &= 0xff;
    // batches that put several reconstructions on a SIMD take the lean two-waves-per-SIMD kernel (2 qubits)
    // MAXJ = 0: the STREAMED instantiation (fbx_pgdb_body.hpp) -- any number of settings, outcome slots read from HBM / L2
    constexpr bool STREAM = MAXJ == 0;
    const int slots = STREAM ? (des->dev.m + 63) / 64 : MAXJ;            // outcome slots per lane
    bool lean = STREAM || (NQ == 2 && (ex.total_batch > B ? ex.total_batch : B) >= FBX_LEAN_MIN_BATCH);
    size_t lds = STREAM ? pgdb_stream_lds(NQ, des->dev.S) : PgdbLds<NQ, false>::bytes(des->dev.S, 64 * MAXJ);      // Ln has one row pair per outcome slot of the kernel
    if constexpr (NQ == 2 && !STREAM) {
        lean = lean && pgdb_lean_eligible(des->dev.S);
        if (lean) lds = pgdb_lean_lds(MAXJ, des->dev.S);
    }
    if (lds > 160 * 1024) {
        set_error("fbx_pgdb_process: too many distinct input states for the prediction tables of the LDS-resident kernel");
        return FBX_ERR_UNSUPPORTED;
    }
    if constexpr (!STREAM) {
        if (!lean) FBX_HIP(hipFuncSetAttribute((const void*)pgdb_kernel<NQ, MAXJ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    // per-item store of Dykstra eigenvector bases (BASIS_CAP x D x D complex each = 128 KiB per 2-qubit
    // item): a grow-only workspace of the calling thread (released by fbx_release_workspace).  A single
    // outer iteration has no previous iteration to take a basis from: no store then.
    constexpr int D = 1 << (2 * NQ);
    // One launch for up to 65 536 reconstructions: the store is 128 KiB per 2-qubit item (8 GiB for 65 536 -- this part
    // has 288 GB), and a launch that holds the whole batch keeps every SIMD busy until the last items, where a
    // sequence of 8192-item launches idles at the end of each (the items of a launch differ by +-15 % in work).
    // When the device cannot give that much, the launch size is halved until the store fits.
    const bool want_basis = !(mode == FBX_MODE_FIXED && max_iters <= 1) && !(mode == FBX_MODE_CONVERGE && max_iters == 1);
    const size_t basis_item = want_basis ? sizeof(cplx) * D * D * BASIS_CAP : 0;
    const size_t counts_item = lean ? sizeof(double) * 2 * (size_t)slots * 64 : 0;      // normalised counts of the lean kernel (fbx_pgdb_body.hpp)
    int64_t CHUNK = 65536;
    char* wsp = nullptr;
    int64_t ws_items = ex.ws_items > 0 ? ex.ws_items : (B < CHUNK ? B : CHUNK);
    if (basis_item + counts_item) {
        for (;;) {
            // the pipelined host entry point says how many slots its stages need between them; everybody else min(B, CHUNK)
            const size_t total = (basis_item + counts_item) * (size_t)ws_items;
            void* w = nullptr;
            const int rc = workspace(WS_PGDB_BASIS, total, &w);
            if (rc == FBX_OK) { wsp = (char*)w; break; }
            // (the streamed instantiations keep 16 B x m of normalised counts per slot: a huge merged design may need fewer slots than 1024)
            if (rc != FBX_ERR_NOMEM || ex.ws_items > 0 || ws_items <= (STREAM ? 1 : 1024)) return rc;
            (void)hipGetLastError();
            ws_items /= 2; CHUNK = ws_items;
        }
    }
    const int64_t n_slots = ws_items;
    cplx* basis = basis_item ? (cplx*)wsp + (size_t)ex.ws_offset * D * D * BASIS_CAP : nullptr;
    double* ncounts = counts_item ? (double*)(wsp + basis_item * (size_t)n_slots) + (size_t)ex.ws_offset * 2 * (size_t)slots * 64 : nullptr;
    const size_t m = des->dev.m;
    DesignDev dev = des->dev;
    dev.eig_rel_tol = ex.eig_rel_tol >= 0.0 ? ex.eig_rel_tol : option_pgdb_eig_rel_tol(NQ);     // per call, else the process default
    dev.ls_reference = ls_reference ? 1 : 0;
    hipStream_t st = ex.launch_stream ? ex.launch_stream : stream();
    for (int64_t b0 = 0; b0 < B; b0 += CHUNK) {
        const int64_t nb = B - b0 < CHUNK ? B - b0 : CHUNK;
        PgdbLaunch a;
        a.dev = dev; a.nb = nb; a.e = e + b0 * m; a.c = c + b0 * m; a.tp = tp; a.mode = mode; a.max_iters = max_iters;
        a.choi = choi + b0 * D * D * 2; a.it = it ? it + b0 : nullptr; a.dy = dy ? dy + b0 : nullptr; a.bt = bt ? bt + b0 : nullptr;
        a.cost = cost ? cost + b0 : nullptr; a.sw = sw ? sw + 4 * b0 : nullptr; a.phase = FBX_PHASE_OUT(b0); a.basis = basis;
        a.basis_cap = BASIS_CAP; a.ncounts = ncounts; a.trace = ex.trace ? ex.trace + (size_t)b0 * ex.trace_iters * 2 : nullptr;
        a.trace_iters = ex.trace_iters;
        // more than one reconstruction per SIMD on the one-wave kernel: in pieces too -- except single-qubit batches to convergence,
        // whose outer iterations last microseconds (a piece's set-up and hand-over then cost what the tail saves: 8000 Pauli
        // experiments 8.6 -> 10.3 ms, while 100 fixed iterations of the SIC design gain 25-35 %; scripts/pieces_time_1q.py)
        const bool fat_pieces = !lean && nb > 1024 && (NQ == 2 || mode == FBX_MODE_FIXED);
        if constexpr (STREAM) {                             // whole reconstructions, one workgroup each
            const int rc = pgdb_stream_launch(NQ, lds, st, a);
            if (rc) return rc;
            continue;
        } else {
        {
            // the two-waves kernel runs its reconstructions in pieces (fbx_pgdb_lean.hip): fbx_set_option("pgdb_pieces"), 1 = whole
            // reconstructions.  FBX_LEAN_PIECES / FBX_LEAN_PIECE_ITERS (environment, experiments and tests) override it per call.
            if (lean || fat_pieces) {
                const char* pv = getenv("FBX_LEAN_PIECES");
                const char* wv = getenv("FBX_LEAN_PIECE_ITERS");
                int pieces = pv && *pv ? atoi(pv) : option_pgdb_pieces();     // default 8 (measured 2048 .. 65 536 experiments: 8 >= 4, 16; scripts/pieces_time.py)
                if (pieces > 64) pieces = 64;
                if (pieces > 1) {
                    const int span = mode == FBX_MODE_FIXED || max_iters > 0 ? max_iters : 64;      // to convergence: ~45 iterations on average
                    a.piece_iters = wv && *wv ? atoi(wv) : (span + pieces - 1) / pieces;
                    if (a.piece_iters < 1) a.piece_iters = 1;
                    // A consumer waits for its predecessor piece with a BOUNDED spin (pgdb_pieces_run) that scales with piece_iters
                    // and allows ~10 ms per outer iteration -- 50 x what one costs on a device that several launches share.  Long
                    // pieces buy nothing (the tail of a launch is one piece either way): a caller with a huge fixed iteration
                    // count gets more pieces (up to 64) and, beyond 512 iterations per piece, whole reconstructions.
                    // (an explicit FBX_LEAN_PIECE_ITERS is taken as given: no doubling)
                    while (!(wv && *wv) && a.piece_iters > 64 && pieces < 64 && (int64_t)pieces * 2 * nb < (int64_t)1 << 30) { pieces *= 2; a.piece_iters = (span + pieces - 1) / pieces; }
                    if (a.piece_iters > 512) pieces = 1;
                }
                if (pieces > 1) {
                    // per workspace slot a progress flag and a record; two ticket counters: the pipelined host entry point has one
                    // launch in flight on each of its two compute streams, on disjoint slot ranges (ws_offset 0 / > 0)
                    void* w = nullptr;
                    const size_t fbytes = (sizeof(int) * (size_t)n_slots + 255) & ~(size_t)255;
                    if (workspace(WS_PGDB_PIECES, 512 + fbytes + sizeof(double) * PGDB_REC * (size_t)n_slots, &w) == FBX_OK) {
                        a.pieces = pieces; a.queue = (int*)w + (ex.ws_offset ? 64 : 0);
                        a.flags = (int*)((char*)w + 512) + ex.ws_offset;
                        a.recs = (double*)((char*)w + 512 + fbytes) + (size_t)ex.ws_offset * PGDB_REC;
                    } else (void)hipGetLastError();                  // no room: whole reconstructions
                }
            }
        }
        if constexpr (NQ == 2) {
            if (lean) {
                const int rc = pgdb_lean_launch(MAXJ, lds, st, a);
                if (rc) return rc;
                continue;
            }
        }
        if (fat_pieces && a.pieces > 1) {
            FBX_HIP(hipFuncSetAttribute((const void*)pgdb_pieces_kernel<NQ, MAXJ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            FBX_HIP(hipMemsetAsync(a.queue, 0, sizeof(int), st));
            FBX_HIP(hipMemsetAsync(a.flags, 0, sizeof(int) * (size_t)a.nb, st));
            hipLaunchKernelGGL((pgdb_pieces_kernel<NQ, MAXJ>), dim3(1024), dim3(64), lds, st, a.dev, a.nb, a.e, a.c, a.tp, a.mode, a.max_iters,
                               a.choi, a.it, a.dy, a.bt, a.cost, a.sw, a.phase, a.basis, a.basis_cap, a.trace, a.trace_iters,
                               a.pieces, a.piece_iters, a.queue, a.flags, a.recs);
            continue;
        }
        hipLaunchKernelGGL((pgdb_kernel<NQ, MAXJ>), dim3((unsigned)nb), dim3(64), lds, st, a.dev, a.nb, a.e, a.c, a.tp, a.mode, a.max_iters,
                           a.choi, a.it, a.dy, a.bt, a.cost, a.sw, a.phase, a.basis, a.basis_cap, a.trace, a.trace_iters);
        }
    }
    FBX_HIP(hipGetLastError());
    return FBX_OK;
}

int pgdb3_dispatch(const fbx_design* des, int64_t B, const double* e, const double* c, int tp, int mode,
                   int max_iters, double* choi, int32_t* it, int32_t* dy, int32_t* bt, double* cost,
                   int32_t* sw, const PgdbExtras& ex);   // fbx_pgdb3.hip
bool pgdb1_eligible(const fbx_design* des);               // fbx_pgdb1.hip: the lane-per-item single-qubit kernel
int pgdb1_dispatch(const fbx_design* des, int64_t B, const double* e, const double* c, int tp, int mode,
                   int max_iters, double* choi, int32_t* it, int32_t* dy, int32_t* bt, double* cost,
                   int32_t* sw, const PgdbExtras& ex);

static int pgdb_dispatch(const fbx_design* des, int64_t B, const double* e, const double* c, int tp,
                         int mode, int max_iters, double* choi, int32_t* it, int32_t* dy,
                         int32_t* bt, double* cost, int32_t* sw, const PgdbExtras& ex) {
    const int n = des->dev.n, m = des->dev.m;
    if (n == 1) {
        // Large batches: 64 reconstructions per wavefront, one per lane (fbx_pgdb1.hip; the eigensolver at its full tolerance --
        // eig_rel_tol does not apply).  A lane is ~3x slower on one reconstruction than a wavefront and a launch lasts as long as
        // its slowest reconstruction (12-14 ms for the Pauli design, 5 ms for SIC, whatever the batch up to 65 536), while the
        // wave-per-item kernel below scales with the batch (1.0 / 0.76 us per reconstruction): the measured crossover
        // (scripts/pgdb1_crossover.py) is ~7000 experiments for the 12-setting SIC design and ~16 000 for the 18-setting Pauli
        // design.  fbx_set_option("pgdb_packed_1q", 0 | 2) forces one or the other.
        {
            const int packed = option_pgdb_packed_1q();
            const int64_t total = ex.total_batch > B ? ex.total_batch : B;
            const int64_t from = m <= 12 ? FBX_PACKED_1Q_MIN_BATCH : 2 * FBX_PACKED_1Q_MIN_BATCH;
            // the lane-per-item kernel always solves to full tolerance: a caller who passes an explicit eigensolver tolerance
            // (fbx_pgdb_process_ex, eig_rel_tol >= 0) gets the wavefront-per-item kernel, which honours it, whatever the batch size
            // -- so that an experiment's iterates do not depend on how many neighbours it is batched with (packed == 2 forces
            // the lane-per-item kernel all the same: a test / diagnostics setting, documented in include/fbx.h)
            const bool explicit_tol = ex.eig_rel_tol >= 0.0 || (mode & FBX_MODE_LS_REFERENCE);      // (the flag too: the wave kernel honours it)
            if (pgdb1_eligible(des) && (packed == 2 || (packed == 1 && total >= from && !explicit_tol)))
                return pgdb1_dispatch(des, B, e, c, tp, mode, max_iters, choi, it, dy, bt, cost, sw, ex);
        }
        if (m <= 64) return launch_pgdb<1, 1>(des, B, e, c, tp, mode, max_iters, choi, it, dy, bt, cost, sw, ex);
        if (m <= 256) return launch_pgdb<1, 4>(des, B, e, c, tp, mode, max_iters, choi, it, dy, bt, cost, sw, ex);
        // merged / repeated datasets (the reference takes any result list, tomography.py:494-539): outcome slots streamed from HBM
        return launch_pgdb<1, 0>(des, B, e, c, tp, mode, max_iters, choi, it, dy, bt, cost, sw, ex);
    } else if (n == 2) {
        if (m <= 256) return launch_pgdb<2, 4>(des, B, e, c, tp, mode, max_iters, choi, it, dy, bt, cost, sw, ex);
        if (m <= 576) return launch_pgdb<2, 9>(des, B, e, c, tp, mode, max_iters, choi, it, dy, bt, cost, sw, ex);
        if (m <= 1024) return launch_pgdb<2, 16>(des, B, e, c, tp, mode, max_iters, choi, it, dy, bt, cost, sw, ex);
        return launch_pgdb<2, 0>(des, B, e, c, tp, mode, max_iters, choi, it, dy, bt, cost, sw, ex);
    }
    else if (n == 3) {
        return pgdb3_dispatch(des, B, e, c, tp, mode, max_iters, choi, it, dy, bt, cost, sw, ex);
    }
    set_error("fbx_pgdb_process: process designs of 1 to 3 qubits");
    return FBX_ERR_UNSUPPORTED;
}

static int pgdb_check(const fbx_design* des, int64_t B, const void* e, const void* c, int mode,
                      int max_iters, const void* choi) {
    { const int rc = check_design(des, "fbx_pgdb_process"); if (rc) return rc; }
    FBX_REQUIRE(des->dev.kind == FBX_KIND_PROCESS, "fbx_pgdb_process: needs a process design");
    FBX_REQUIRE(B >= 0, "fbx_pgdb_process: negative batch");
    FBX_REQUIRE(B == 0 || (e && c && choi), "fbx_pgdb_process: NULL buffer");
    FBX_REQUIRE((mode & ~FBX_MODE_LS_REFERENCE) == FBX_MODE_CONVERGE || (mode & ~FBX_MODE_LS_REFERENCE) == FBX_MODE_FIXED, "fbx_pgdb_process: bad mode");
    FBX_REQUIRE(max_iters >= 0, "fbx_pgdb_process: negative max_iters");
    return FBX_OK;
}

}  // namespace fbx

using namespace fbx;

extern "C" {

#ifdef FBX_DIAGNOSTICS
// diagnostics builds only (libfbx_prof.so / libfbx_cor.so; not part of include/fbx.h, not in libfbx.so):
// device buffer of 8 int64 per item that a -DFBX_PHASE_TIMERS build fills with per-phase shader cycles
int fbx_debug_set_phase_buffer(long long* d_buf) { g_phase_out = d_buf; return FBX_OK; }

// the device natural log used by the PGDB line search, on host arrays
int fbx_debug_log(const double* x, double* out, int64_t n) {
    int rc = ensure_device();
    if (rc) return rc;
    DevBuf dx, dout;
    if ((rc = dx.alloc(sizeof(double) * n)) || (rc = dout.alloc(sizeof(double) * n))) return rc;
    FBX_HIP(hipMemcpyAsync(dx.p, x, sizeof(double) * n, hipMemcpyHostToDevice, stream()));
    hipLaunchKernelGGL(debug_log_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream(),
                       dx.as<double>(), dout.as<double>(), (long long)n);
    FBX_HIP(hipGetLastError());
    FBX_HIP(hipMemcpyAsync(out, dout.p, sizeof(double) * n, hipMemcpyDeviceToHost, stream()));
    FBX_HIP(hipStreamSynchronize(stream()));
    return FBX_OK;
}
#endif

static int check_extras(double eig_rel_tol, const void* trace, int trace_iters) {
    FBX_REQUIRE(eig_rel_tol < 0.0 || eig_rel_tol <= 1e-3, "fbx_pgdb_process_ex: eig_rel_tol must be negative (default) or in [0, 1e-3]");
    FBX_REQUIRE(eig_rel_tol == eig_rel_tol, "fbx_pgdb_process_ex: eig_rel_tol is NaN");
    FBX_REQUIRE(trace_iters >= 0 && (trace == nullptr || trace_iters > 0), "fbx_pgdb_process_ex: trace_out needs trace_iters > 0");
    return FBX_OK;
}

int fbx_pgdb_process_ex_dev(const fbx_design* design, int64_t B, const double* d_expect,
                            const double* d_counts, int trace_preserving, int mode, int max_iters,
                            double eig_rel_tol, double* d_choi_out, int32_t* d_iters_out, int32_t* d_dykstra_out,
                            int32_t* d_backtracks_out, double* d_cost_out, int32_t* d_work_out,
                            int32_t* d_trace_out, int trace_iters) {
    int rc = ensure_device();
    if (rc) return rc;
    rc = pgdb_check(design, B, d_expect, d_counts, mode, max_iters, d_choi_out);
    if (rc) return rc;
    rc = check_extras(eig_rel_tol, d_trace_out, trace_iters);
    if (rc) return rc;
    if (B == 0) return FBX_OK;
    PgdbExtras ex; ex.eig_rel_tol = eig_rel_tol; ex.trace = d_trace_out; ex.trace_iters = d_trace_out ? trace_iters : 0;
    if (ex.trace) FBX_HIP(hipMemsetAsync(ex.trace, 0, sizeof(int32_t) * 2 * (size_t)B * trace_iters, stream()));
    return pgdb_dispatch(design, B, d_expect, d_counts, trace_preserving, mode, max_iters,
                         d_choi_out, d_iters_out, d_dykstra_out, d_backtracks_out, d_cost_out, d_work_out, ex);
}

int fbx_pgdb_process_dev(const fbx_design* design, int64_t B, const double* d_expect,
                         const double* d_counts, int trace_preserving, int mode, int max_iters,
                         double* d_choi_out, int32_t* d_iters_out, int32_t* d_dykstra_out,
                         int32_t* d_backtracks_out, double* d_cost_out, int32_t* d_work_out) {
    return fbx_pgdb_process_ex_dev(design, B, d_expect, d_counts, trace_preserving, mode, max_iters, -1.0, d_choi_out,
                                   d_iters_out, d_dykstra_out, d_backtracks_out, d_cost_out, d_work_out, nullptr, 0);
}

// the host-pointer call on the calling thread's device; `total_batch` = size of the caller's whole batch when this is one
// device's block of it (kernels are chosen by the whole, so that the split does not change a single bit of any item)
static int pgdb_process_host(const fbx_design* design, int64_t B, const double* expect,
                             const double* counts, int trace_preserving, int mode, int max_iters, double eig_rel_tol,
                             double* choi_out, int32_t* iters_out, int32_t* dykstra_out,
                             int32_t* backtracks_out, double* cost_out, int32_t* work_out,
                             int32_t* trace_out, int trace_iters, int64_t total_batch);

int fbx_pgdb_process_ex(const fbx_design* design, int64_t B, const double* expect,
                        const double* counts, int trace_preserving, int mode, int max_iters, double eig_rel_tol,
                        double* choi_out, int32_t* iters_out, int32_t* dykstra_out,
                        int32_t* backtracks_out, double* cost_out, int32_t* work_out,
                        int32_t* trace_out, int trace_iters) {
    int rc = ensure_device();
    if (rc) return rc;
    rc = pgdb_check(design, B, expect, counts, mode, max_iters, choi_out);
    if (rc) return rc;
    rc = check_extras(eig_rel_tol, trace_out, trace_iters);
    if (rc) return rc;
    if (B == 0) return FBX_OK;
    // fbx_set_devices: contiguous blocks of the batch, one per entry of the device list, each on that entry's
    // worker thread with its own context and a replica of the design; no exchange between devices
    if (device_list_size() > 1 && !in_device_worker() && B >= 2 * (int64_t)device_list_size()) {
        const size_t m = design->dev.m, DD = (size_t)design->dev.D * design->dev.D;
        return run_on_devices([&](int g, int G) -> int {       // G: the list's length as run_on_devices read it, under its lock
            const int64_t per = (B + G - 1) / G;
            const int64_t lo = (int64_t)g * per < B ? (int64_t)g * per : B, nb = (B - lo < per ? B - lo : per);
            if (nb <= 0) return FBX_OK;
            int r = ensure_device();
            if (r) return r;
            const fbx_design* dg = design_on_this_device(design, &r);
            if (r) return r;
            return pgdb_process_host(dg, nb, expect + lo * m, counts + lo * m, trace_preserving, mode, max_iters, eig_rel_tol,
                                     choi_out + lo * 2 * DD, iters_out ? iters_out + lo : nullptr, dykstra_out ? dykstra_out + lo : nullptr,
                                     backtracks_out ? backtracks_out + lo : nullptr, cost_out ? cost_out + lo : nullptr,
                                     work_out ? work_out + 4 * lo : nullptr, trace_out ? trace_out + (size_t)lo * trace_iters * 2 : nullptr,
                                     trace_iters, B);
        });
    }
    return pgdb_process_host(design, B, expect, counts, trace_preserving, mode, max_iters, eig_rel_tol, choi_out, iters_out,
                             dykstra_out, backtracks_out, cost_out, work_out, trace_out, trace_iters, B);
}

static int pgdb_process_host(const fbx_design* design, int64_t B, const double* expect,
                             const double* counts, int trace_preserving, int mode, int max_iters, double eig_rel_tol,
                             double* choi_out, int32_t* iters_out, int32_t* dykstra_out,
                             int32_t* backtracks_out, double* cost_out, int32_t* work_out,
                             int32_t* trace_out, int trace_iters, int64_t total_batch) {
    int rc = FBX_OK;
    const size_t m = design->dev.m, D = design->dev.D;
    const size_t trace_bytes = trace_out ? sizeof(int32_t) * 2 * (size_t)B * trace_iters : 0;
    DevBuf de, dc, dchoi, dit, ddy, dbt, dcost, dsw, dtr;
    if ((rc = de.alloc(sizeof(double) * B * m)) || (rc = dc.alloc(sizeof(double) * B * m)) ||
        (rc = dchoi.alloc(sizeof(double) * 2 * B * D * D)) || (rc = dit.alloc(sizeof(int32_t) * B)) ||
        (rc = ddy.alloc(sizeof(int32_t) * B)) || (rc = dbt.alloc(sizeof(int32_t) * B)) ||
        (rc = dcost.alloc(sizeof(double) * B)) || (rc = dsw.alloc(sizeof(int32_t) * 4 * B)) ||
        (trace_bytes && (rc = dtr.alloc(trace_bytes))))
        return rc;
    PgdbExtras ex; ex.eig_rel_tol = eig_rel_tol; ex.trace = trace_bytes ? dtr.as<int32_t>() : nullptr; ex.trace_iters = trace_bytes ? trace_iters : 0;
    ex.total_batch = total_batch;
    if (ex.trace) FBX_HIP(hipMemsetAsync(ex.trace, 0, trace_bytes, stream()));
    // Page-locked caller buffers and more than one stage of work: H2D, kernels and D2H overlap on separate streams (SURVEY.md
    // 8d prices the path including both transfers).  What cannot be hidden is the H2D of the first stage and the D2H of the
    // last, and every boundary between stages costs a launch tail (a stage's slowest items run while SIMDs idle): so the plan
    // is a SMALL first stage (fbx_set_option("pgdb_host_chunk") items, default 4096 = two reconstructions per wave slot of the two-waves kernel, which runs them in pieces; at
    // most half the batch), a small last one, and everything in between in as few launches as the per-item workspace allows
    // (65 536 items each) on a second, HIGH-PRIORITY compute stream: the first stage's kernel covers the upload of the rest,
    // the bulk takes the SIMDs over as soon as it has arrived, and the last stage -- queued behind the first on the normal-
    // priority stream -- only fills what the bulk's tail leaves idle and is still computing while the bulk's results go
    // down.  (Round 3 first used equal stages: 8 boundaries for 65 536 items, 94-96 % of the resident rate.)
    const int64_t CH = option_pgdb_host_chunk();
    if (B > CH && host_pointer_is_pinned(expect, sizeof(double) * B * m) && host_pointer_is_pinned(counts, sizeof(double) * B * m) &&
        host_pointer_is_pinned(choi_out, sizeof(double) * 2 * B * D * D)) {
        hipStream_t s_in, s_out, s_c2;
        if ((rc = copy_streams(&s_in, &s_out, &s_c2))) return rc;
        // every failure inside the pipeline leaves through ONE door that waits for all four streams: copies and kernels may
        // already be queued on them, and the staging blocks go back to the thread's pool when this function returns
        auto drain = [&]() { (void)hipStreamSynchronize(s_in); (void)hipStreamSynchronize(stream()); (void)hipStreamSynchronize(s_c2); (void)hipStreamSynchronize(s_out); };
#define FBX_HIP_P(call) do { hipError_t _e = (call); if (_e != hipSuccess) { drain(); return hip_fail(_e, #call, __FILE__, __LINE__); } } while (0)
        struct Stage { int64_t b0, nb; bool second; int64_t ws_offset; };
        std::vector<Stage> plan;
        const int64_t MID = 65536;
        const bool two_streams = design->dev.n <= 2;          // (the 3-qubit launcher has one workspace: one compute stream)
        const int64_t first = CH < B / 2 ? CH : B / 2;
        const int64_t last = (B >= 4 * CH) ? CH : 0;           // queued BEHIND the first stage: only when the bulk outlasts both
        plan.push_back({0, first, false, 0});
        for (int64_t b0 = first; b0 < B - last; b0 += MID)
            plan.push_back({b0, (B - last - b0 < MID ? B - last - b0 : MID), two_streams, two_streams ? first : 0});
        if (last) plan.push_back({B - last, last, false, 0});
        const int64_t mid_slots = (B - last - first) < MID ? (B - last - first) : MID;
        const int64_t ws_total = two_streams ? first + mid_slots : (first > mid_slots ? first : mid_slots);
        const int nst = (int)plan.size();
        hipEvent_t* ev = nullptr;
        if ((rc = ordering_events(2 * nst + 1, &ev))) return rc;      // (nothing queued yet)
        // (the staging buffers may still be in use by earlier work of this thread's stream)
        FBX_HIP_P(hipEventRecord(ev[2 * nst], stream()));
        FBX_HIP_P(hipStreamWaitEvent(s_in, ev[2 * nst], 0));
        FBX_HIP_P(hipStreamWaitEvent(s_out, ev[2 * nst], 0));
        FBX_HIP_P(hipStreamWaitEvent(s_c2, ev[2 * nst], 0));
        for (int k = 0; k < nst; ++k) {                        // uploads in stage order, back to back
            const int64_t b0 = plan[k].b0, nb = plan[k].nb;
            FBX_HIP_P(hipMemcpyAsync(de.as<double>() + b0 * m, expect + b0 * m, sizeof(double) * nb * m, hipMemcpyHostToDevice, s_in));
            FBX_HIP_P(hipMemcpyAsync(dc.as<double>() + b0 * m, counts + b0 * m, sizeof(double) * nb * m, hipMemcpyHostToDevice, s_in));
            FBX_HIP_P(hipEventRecord(ev[2 * k], s_in));
        }
        for (int k = 0; k < nst; ++k) {
            const int64_t b0 = plan[k].b0, nb = plan[k].nb;
            hipStream_t s_k = plan[k].second ? s_c2 : stream();
            FBX_HIP_P(hipStreamWaitEvent(s_k, ev[2 * k], 0));
            PgdbExtras exk = ex;
            if (exk.trace) exk.trace += (size_t)b0 * exk.trace_iters * 2;
            exk.launch_stream = s_k; exk.ws_items = ws_total; exk.ws_offset = plan[k].ws_offset; exk.total_batch = total_batch > B ? total_batch : B;
            rc = pgdb_dispatch(design, nb, de.as<double>() + b0 * m, dc.as<double>() + b0 * m, trace_preserving, mode, max_iters,
                               dchoi.as<double>() + b0 * 2 * D * D, dit.as<int32_t>() + b0, ddy.as<int32_t>() + b0,
                               dbt.as<int32_t>() + b0, dcost.as<double>() + b0, dsw.as<int32_t>() + 4 * b0, exk);
            if (rc) break;
            FBX_HIP_P(hipEventRecord(ev[2 * k + 1], s_k));
            FBX_HIP_P(hipStreamWaitEvent(s_out, ev[2 * k + 1], 0));
            FBX_HIP_P(hipMemcpyAsync(choi_out + b0 * 2 * D * D, dchoi.as<double>() + b0 * 2 * D * D, sizeof(double) * 2 * nb * D * D,
                                   hipMemcpyDeviceToHost, s_out));
        }
        if (rc) {
            drain();
            // the pipeline wants workspace slots for the whole bulk at once; when the device cannot give that much the staged path
            // below runs the batch in launches that fit (it halves them until the store does)
            if (rc != FBX_ERR_NOMEM) return rc;
            (void)hipGetLastError();
            goto staged;
        }
        FBX_HIP_P(hipEventRecord(ev[2 * nst], s_c2));            // the small outputs below follow BOTH compute streams
        FBX_HIP_P(hipStreamWaitEvent(stream(), ev[2 * nst], 0));
        if (iters_out) FBX_HIP_P(hipMemcpyAsync(iters_out, dit.p, sizeof(int32_t) * B, hipMemcpyDeviceToHost, stream()));
        if (dykstra_out) FBX_HIP_P(hipMemcpyAsync(dykstra_out, ddy.p, sizeof(int32_t) * B, hipMemcpyDeviceToHost, stream()));
        if (backtracks_out) FBX_HIP_P(hipMemcpyAsync(backtracks_out, dbt.p, sizeof(int32_t) * B, hipMemcpyDeviceToHost, stream()));
        if (cost_out) FBX_HIP_P(hipMemcpyAsync(cost_out, dcost.p, sizeof(double) * B, hipMemcpyDeviceToHost, stream()));
        if (work_out) FBX_HIP_P(hipMemcpyAsync(work_out, dsw.p, sizeof(int32_t) * 4 * B, hipMemcpyDeviceToHost, stream()));
        if (trace_bytes) FBX_HIP_P(hipMemcpyAsync(trace_out, dtr.p, trace_bytes, hipMemcpyDeviceToHost, stream()));
        FBX_HIP_P(hipStreamSynchronize(stream()));
        FBX_HIP_P(hipStreamSynchronize(s_out));
        FBX_HIP_P(hipStreamSynchronize(s_in));
        return FBX_OK;
#undef FBX_HIP_P
    }
staged:
    FBX_HIP(hipMemcpyAsync(de.p, expect, sizeof(double) * B * m, hipMemcpyHostToDevice, stream()));
    FBX_HIP(hipMemcpyAsync(dc.p, counts, sizeof(double) * B * m, hipMemcpyHostToDevice, stream()));
    rc = pgdb_dispatch(design, B, de.as<double>(), dc.as<double>(), trace_preserving, mode, max_iters,
                       dchoi.as<double>(), dit.as<int32_t>(), ddy.as<int32_t>(), dbt.as<int32_t>(),
                       dcost.as<double>(), dsw.as<int32_t>(), ex);
    if (rc) { (void)hipStreamSynchronize(stream()); return rc; }
    FBX_HIP(hipMemcpyAsync(choi_out, dchoi.p, sizeof(double) * 2 * B * D * D, hipMemcpyDeviceToHost, stream()));
    if (iters_out) FBX_HIP(hipMemcpyAsync(iters_out, dit.p, sizeof(int32_t) * B, hipMemcpyDeviceToHost, stream()));
    if (dykstra_out) FBX_HIP(hipMemcpyAsync(dykstra_out, ddy.p, sizeof(int32_t) * B, hipMemcpyDeviceToHost, stream()));
    if (backtracks_out) FBX_HIP(hipMemcpyAsync(backtracks_out, dbt.p, sizeof(int32_t) * B, hipMemcpyDeviceToHost, stream()));
    if (cost_out) FBX_HIP(hipMemcpyAsync(cost_out, dcost.p, sizeof(double) * B, hipMemcpyDeviceToHost, stream()));
    if (work_out) FBX_HIP(hipMemcpyAsync(work_out, dsw.p, sizeof(int32_t) * 4 * B, hipMemcpyDeviceToHost, stream()));
    if (trace_bytes) FBX_HIP(hipMemcpyAsync(trace_out, dtr.p, trace_bytes, hipMemcpyDeviceToHost, stream()));
    FBX_HIP(hipStreamSynchronize(stream()));
    return FBX_OK;
}

// _cost / _grad_cost (tomography.py:597-633); include/fbx.h
int fbx_pgdb_cost_grad_dev(const fbx_design* design, int64_t B, const double* d_nvec, const double* d_choi_in, double eps,
                           double* d_cost_out, double* d_grad_out) {
    int rc = ensure_device();
    if (rc) return rc;
    { const int r = check_design(design, "fbx_pgdb_cost_grad"); if (r) return r; }
    FBX_REQUIRE(design->dev.kind == FBX_KIND_PROCESS, "fbx_pgdb_cost_grad: needs a process design");
    FBX_REQUIRE(B >= 0, "fbx_pgdb_cost_grad: negative batch");
    FBX_REQUIRE(B == 0 || (d_nvec && d_choi_in && (d_cost_out || d_grad_out)), "fbx_pgdb_cost_grad: NULL buffer");
    FBX_REQUIRE(eps >= 0.0, "fbx_pgdb_cost_grad: eps must be a non-negative number");      // (also rejects NaN)
    if (B == 0) return FBX_OK;
    switch (design->dev.n) {
        case 1: return launch_cost_grad<1>(design, B, d_nvec, d_choi_in, eps, d_cost_out, d_grad_out);
        case 2: return launch_cost_grad<2>(design, B, d_nvec, d_choi_in, eps, d_cost_out, d_grad_out);
        case 3: return pgdb3_cost_grad_launch(design, B, d_nvec, d_choi_in, eps, d_cost_out, d_grad_out);
    }
    set_error("fbx_pgdb_cost_grad: process designs of 1 to 3 qubits");
    return FBX_ERR_UNSUPPORTED;
}

int fbx_pgdb_cost_grad(const fbx_design* design, int64_t B, const double* nvec, const double* choi_in, double eps,
                       double* cost_out, double* grad_out) {
    int rc = ensure_device();
    if (rc) return rc;
    { const int r = check_design(design, "fbx_pgdb_cost_grad"); if (r) return r; }
    FBX_REQUIRE(B >= 0, "fbx_pgdb_cost_grad: negative batch");
    FBX_REQUIRE(B == 0 || (nvec && choi_in && (cost_out || grad_out)), "fbx_pgdb_cost_grad: NULL buffer");
    if (B == 0) return FBX_OK;
    const size_t m = design->dev.m, DD = (size_t)design->dev.D * design->dev.D;
    DevBuf dn, dchoi, dcost, dgrad;
    if ((rc = dn.alloc(sizeof(double) * 2 * B * m)) || (rc = dchoi.alloc(sizeof(double) * 2 * B * DD)) ||
        (rc = dcost.alloc(sizeof(double) * B)) || (grad_out && (rc = dgrad.alloc(sizeof(double) * 2 * B * DD))))
        return rc;
    FBX_HIP(hipMemcpyAsync(dn.p, nvec, sizeof(double) * 2 * B * m, hipMemcpyHostToDevice, stream()));
    FBX_HIP(hipMemcpyAsync(dchoi.p, choi_in, sizeof(double) * 2 * B * DD, hipMemcpyHostToDevice, stream()));
    rc = fbx_pgdb_cost_grad_dev(design, B, dn.as<double>(), dchoi.as<double>(), eps, dcost.as<double>(), grad_out ? dgrad.as<double>() : nullptr);
    if (rc) { (void)hipStreamSynchronize(stream()); return rc; }
    if (cost_out) FBX_HIP(hipMemcpyAsync(cost_out, dcost.p, sizeof(double) * B, hipMemcpyDeviceToHost, stream()));
    if (grad_out) FBX_HIP(hipMemcpyAsync(grad_out, dgrad.p, sizeof(double) * 2 * B * DD, hipMemcpyDeviceToHost, stream()));
    FBX_HIP(hipStreamSynchronize(stream()));
    return FBX_OK;
}

int fbx_pgdb_process(const fbx_design* design, int64_t B, const double* expect,
                     const double* counts, int trace_preserving, int mode, int max_iters,
                     double* choi_out, int32_t* iters_out, int32_t* dykstra_out,
                     int32_t* backtracks_out, double* cost_out, int32_t* work_out) {
    return fbx_pgdb_process_ex(design, B, expect, counts, trace_preserving, mode, max_iters, -1.0, choi_out, iters_out,
                               dykstra_out, backtracks_out, cost_out, work_out, nullptr, 0);
}

}  // extern "C"
