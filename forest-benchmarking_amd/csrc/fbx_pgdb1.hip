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
#ifdef FBX_PHASE_TIMERS
// profile build (scripts/pgdb1_phase_profile.py): wall cycles of a wavefront between the marks of an outer iteration
// [0] gradient, [1] projection, [2] update + first cost, [3] line search, [4] whole kernel per wavefront; trip counts as the
// WAVEFRONT runs them (odd slot) and summed over its lanes (the even slot that follows): 5/6 outer iterations, 7/8 Dykstra
// iterations, 9/10 Jacobi sweeps, 11/12 halvings, 13 power-sum passes, 14/15 full cost evaluations inside the line search.
// Accumulated in LDS (one wavefront per workgroup), flushed once per wavefront into the row of the launch's step.
constexpr int P1_PROF_STEPS = 160;
__device__ unsigned long long g_p1_prof[P1_PROF_STEPS][24];
__shared__ unsigned long long p1_prof_lds[24];
#define P1_LEADER() ((int)(__ffsll((long long)__ballot(1)) - 1) == (int)(threadIdx.x & 63))
#define P1_PROF_BEGIN unsigned long long p1_prof_t = clock64()
#define P1_PROF(k) do { const unsigned long long now_ = clock64(); if (P1_LEADER()) atomicAdd(&p1_prof_lds[k], now_ - p1_prof_t); p1_prof_t = now_; } while (0)
#define P1_COUNT(k) do { if (P1_LEADER()) atomicAdd(&p1_prof_lds[k], 1ull); } while (0)
#define P1_COUNT_LANES(k, n) do { const unsigned long long m_ = __ballot(1); if (P1_LEADER()) atomicAdd(&p1_prof_lds[k], (unsigned long long)__popcll(m_) * (n)); } while (0)
#define P1_PROF_KERNEL_BEGIN const unsigned long long p1_k0 = clock64(); unsigned long long p1_kt = p1_k0; if (threadIdx.x < 24) p1_prof_lds[threadIdx.x] = 0ull; __syncthreads()
#define P1_KMARK(k) do { __builtin_amdgcn_s_waitcnt(0); const unsigned long long now_ = clock64(); if (threadIdx.x == 0) atomicAdd(&p1_prof_lds[k], now_ - p1_kt); p1_kt = now_; } while (0)
#define P1_PROF_KERNEL_END(step) do { __syncthreads(); if (threadIdx.x < 24) { unsigned long long v_ = p1_prof_lds[threadIdx.x]; if (threadIdx.x == 4) v_ = clock64() - p1_k0; \
        atomicAdd(&g_p1_prof[(step) < P1_PROF_STEPS ? (step) : P1_PROF_STEPS - 1][threadIdx.x], v_); } } while (0)
#else
#define P1_PROF_KERNEL_BEGIN
#define P1_PROF_KERNEL_END(step)
#define P1_KMARK(k)
#endif
#include "fbx_pgdb1_core.hpp"
#include <cstdlib>
#include <cstdio>
#include <cstring>

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

// a finished reconstruction: the estimate as a full 4 x 4 complex matrix and the counters
__device__ __forceinline__ void p1_write_result(const P1State& st, long long item, double* __restrict__ choi_out, int* __restrict__ iters_out,
                                                int* __restrict__ dykstra_out, int* __restrict__ backtracks_out,
                                                double* __restrict__ cost_out, int* __restrict__ work_out) {
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
}

__global__ void __launch_bounds__(64)
pgdb1_packed_kernel(DesignDev des, long long B, const double* __restrict__ expect, const double* __restrict__ counts,
                    int trace_preserving, int mode, int max_iters, double* __restrict__ choi_out,
                    int* __restrict__ iters_out, int* __restrict__ dykstra_out, int* __restrict__ backtracks_out,
                    double* __restrict__ cost_out, int* __restrict__ work_out, int* __restrict__ trace_out, int trace_iters,
                    unsigned long long* __restrict__ next_item) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    P1_PROF_KERNEL_BEGIN;
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
            p1_write_result(st, item, choi_out, iters_out, dykstra_out, backtracks_out, cost_out, work_out);
            item = (long long)atomicAdd(next_item, 1ull);
            active = item < B;
            if (active) { p1_load_counts(des, item, expect, counts, nt.tab); p1_begin(des, nt, st); }
        }
    }
    P1_PROF_KERNEL_END(0);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Binned relaunch (large batches).  The persistent kernel above keeps a wavefront full, but a wavefront still runs, at every
// nesting level, the LONGEST trip count of its 64 lanes: an outer iteration's projection is 2-4 Dykstra iterations for 98 %
// of the lanes and 5 to several thousand for the rest, so with 64 random reconstructions per wavefront every iteration pays
// for a straggler (profile build, 2^20 Pauli experiments: 9.5 Dykstra trips per wavefront and outer iteration against 3.2
// per lane, 5.6 full cost evaluations inside the line search against 0.17).  An item's Dykstra count is strongly persistent
// from one outer iteration to the next (88 % equal, 98 % within one), hence:
//
//   * ONE outer iteration per launch.  Between launches a reconstruction's running state -- estimate, warm-start basis, costs,
//     counters: P1_NF = 56 doubles -- lives in HBM, in one of P1_NB BINS keyed by the Dykstra count of the iteration it has
//     just finished.  The next launch walks the bins: a wavefront takes 64 consecutive slots of ONE bin, so its lanes have
//     (nearly always) the same trip counts.
//   * Slots are grouped by 64, field-major inside a group: a wavefront loads its group with one coalesced 512-byte read per
//     field.  After its iteration a lane claims a slot in the bin of its new key (one atomicAdd per wavefront and distinct key)
//     and stores the state there; lanes of a wavefront mostly agree on the key, so the stores come in long runs too.
//   * Two sets of bin regions alternate (read one, fill the other); three count arrays rotate (read / fill / being cleared).
//     A region holds TWO bins, one growing up from its first slot and one growing down from its last: their counts never add
//     up to more than the batch, so a region of `batch` slots cannot overflow whatever the distribution of the keys.
//   * The normalised counts (2 m doubles, constant) are written once, item-major, by the first launch and gathered by every
//     later one (a lane reads 2 m consecutive doubles: whole cache lines, out of L2).
//   * Once few reconstructions are left (they no longer fill the chip: grouping buys nothing and every launch costs a full
//     iteration's latency) one last launch runs the remaining ones to completion out of registers, as the kernel above does.
// The arithmetic of a reconstruction is per lane and in a fixed order either way: results are bit-identical to the persistent
// kernel's (tests/test_pgdb1_gpu.py).  Measured: 2^20 experiments to convergence 64.6 -> 53 ms (Pauli),
// 43.4 -> 34.5 ms (SIC); 30 fixed iterations 500 -> 217 ms / 131 -> 46 ms.
constexpr int P1_NB = 8;                 // bins: Dykstra count of the last outer iteration 1 .. 7, 8+
constexpr int P1_NF = 56;                // doubles per slot
struct P1Bins {
    const double* cur; double* next;     // [P1_NB / 2][cap_slots / 64][P1_NF][64]
    const int* cur_count; int* next_count; int* clear_count;     // [P1_NB] each
    double* tab;                         // [n_items][2 m] normalised counts
    long long cap_slots;                 // slots of a region (a multiple of 64)
};

__device__ __forceinline__ double p1_pack(int lo, int hi) { return __hiloint2double(hi, lo); }
// offset (in doubles) of field 0 of logical slot `s` of bin `bin`: even bins fill their region upwards, odd bins downwards
__device__ __forceinline__ long long p1_slot_offset(const P1Bins& bins, int bin, long long s) {
    const long long phys = (bin & 1) ? bins.cap_slots - 1 - s : s;
    return (((long long)(bin >> 1) * (bins.cap_slots >> 6) + (phys >> 6)) * P1_NF) * 64 + (phys & 63);
}

__global__ void __launch_bounds__(64)
pgdb1_step_kernel(DesignDev des, long long first_item, long long n_items, const double* __restrict__ expect,
                  const double* __restrict__ counts, int trace_preserving, int mode, int max_iters, double* __restrict__ choi_out,
                  int* __restrict__ iters_out, int* __restrict__ dykstra_out, int* __restrict__ backtracks_out,
                  double* __restrict__ cost_out, int* __restrict__ work_out, int* __restrict__ trace_out, int trace_iters,
                  P1Bins bins, int first, int to_completion, int step) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    const int m = des.m;
    LaneCounts nt;
    nt.tab = reinterpret_cast<double*>(smem) + lane;
    if (blockIdx.x == 0 && lane < P1_NB) bins.clear_count[lane] = 0;
    // which 64 slots of which bin (wave-uniform)
    long long g = blockIdx.x;
    int bin = 0, nslots = 0;
    if (first) {
        const long long left = n_items - g * 64;
        nslots = left <= 0 ? 0 : (left < 64 ? (int)left : 64);
    } else {
        for (; bin < P1_NB; ++bin) {
            const int cnt = bins.cur_count[bin];
            const long long groups = (cnt + 63) / 64;
            if (g < groups) { const long long left = cnt - g * 64; nslots = left < 64 ? (int)left : 64; break; }
            g -= groups;
        }
    }
    if (nslots == 0) return;
    P1_PROF_KERNEL_BEGIN;
    const bool active = lane < nslots;
    P1State st;
    long long item = 0;                  // index inside this chunk of the batch
    if (first) {
        item = g * 64 + lane;
        if (active) {
            p1_load_counts(des, first_item + item, expect, counts, nt.tab);
            double* t = bins.tab + item * (2 * m);
            for (int k = 0; k < 2 * m; ++k) t[k] = nt.tab[k * 64];
            p1_begin(des, nt, st);
        }
    } else if (active) {
        const double* src = bins.cur + p1_slot_offset(bins, bin, g * 64 + lane);
        item = (long long)__double_as_longlong(src[55 * 64]);
        // (batches of 12 loads in flight: one load - wait - LDS store per trip would be 2 m memory round trips)
        const double* t = bins.tab + item * (2 * m);
        for (int k0 = 0; k0 < 2 * m; k0 += 12) {
            double v[12];
#pragma unroll
            for (int j = 0; j < 12; ++j) v[j] = k0 + j < 2 * m ? t[k0 + j] : 0.0;
#pragma unroll
            for (int j = 0; j < 12; ++j) if (k0 + j < 2 * m) nt.tab[(k0 + j) * 64] = v[j];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) st.est.d[k] = src[k * 64];
#pragma unroll
        for (int k = 0; k < 6; ++k) { st.est.re[k] = src[(4 + k) * 64]; st.est.im[k] = src[(10 + k) * 64]; }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) { st.basis.V.re[r][c] = src[(16 + r * 4 + c) * 64]; st.basis.V.im[r][c] = src[(32 + r * 4 + c) * 64]; }
        st.old_cost = src[48 * 64]; st.new_cost = src[49 * 64];
        double w;
        w = src[50 * 64]; st.iters = __double2loint(w); st.dyk = __double2hiint(w);
        w = src[51 * 64]; st.backtracks = __double2loint(w); st.sweeps = __double2hiint(w);
        w = src[52 * 64]; st.terms = __double2loint(w); st.ls_full = __double2hiint(w);
        w = src[53 * 64]; st.ls_sums = __double2loint(w); st.basis.chain = __double2hiint(w);
        w = src[54 * 64]; st.basis.valid = __double2loint(w) != 0;
    }
    P1_KMARK(16);
    bool done = false;
    int dyk_this = 0, bt_this = 0;
    if (active) {
        do {
            const int it_before = st.iters;
            done = p1_outer_iteration(des, nt, st, trace_preserving != 0, mode, max_iters, dyk_this, bt_this);
            if (trace_out && st.iters > it_before && it_before < trace_iters) {
                int* tr = trace_out + ((size_t)(first_item + item) * trace_iters + it_before) * 2;
                tr[0] = dyk_this; tr[1] = bt_this;
            }
        } while (to_completion && !done);
    }
    P1_KMARK(17);
    if (active && done) p1_write_result(st, first_item + item, choi_out, iters_out, dykstra_out, backtracks_out, cost_out, work_out);
    P1_KMARK(18);
    // a slot in the bin of the new key: one atomicAdd per wavefront and distinct key
    const bool emit = active && !done;
    const int key = (dyk_this < 1 ? 1 : (dyk_this > P1_NB ? P1_NB : dyk_this)) - 1;
    unsigned long long todo = __ballot(emit);
    int pos = 0;
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int k = __builtin_amdgcn_readlane(key, leader);
        const unsigned long long mk = __ballot(emit && key == k);
        int base = 0;
        if (lane == leader) base = atomicAdd(&bins.next_count[k], (int)__popcll(mk));
        base = __builtin_amdgcn_readlane(base, leader);
        if (emit && key == k) pos = base + (int)__popcll(mk & ((1ull << lane) - 1ull));
        todo &= ~mk;
    }
    P1_KMARK(19);
    if (emit) {
        double* dst = bins.next + p1_slot_offset(bins, key, pos);
#pragma unroll
        for (int k = 0; k < 4; ++k) dst[k * 64] = st.est.d[k];
#pragma unroll
        for (int k = 0; k < 6; ++k) { dst[(4 + k) * 64] = st.est.re[k]; dst[(10 + k) * 64] = st.est.im[k]; }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) { dst[(16 + r * 4 + c) * 64] = st.basis.V.re[r][c]; dst[(32 + r * 4 + c) * 64] = st.basis.V.im[r][c]; }
        dst[48 * 64] = st.old_cost; dst[49 * 64] = st.new_cost;
        dst[50 * 64] = p1_pack(st.iters, st.dyk);
        dst[51 * 64] = p1_pack(st.backtracks, st.sweeps);
        dst[52 * 64] = p1_pack(st.terms, st.ls_full);
        dst[53 * 64] = p1_pack(st.ls_sums, st.basis.chain);
        dst[54 * 64] = p1_pack(st.basis.valid ? 1 : 0, 0);
        dst[55 * 64] = __longlong_as_double(item);
    }
    P1_KMARK(20);
    P1_PROF_KERNEL_END(step);
}

__global__ void pgdb1_set_counter(unsigned long long* p, unsigned long long v) { *p = v; }

// Largest design the packed kernel takes: the counts table is 1 KiB x m of LDS per wavefront
constexpr int PGDB1_MAX_M = 64;

bool pgdb1_eligible(const fbx_design* des) { return des->dev.n == 1 && des->dev.m <= PGDB1_MAX_M; }

static long long env_ll(const char* name, long long dflt) { const char* v = getenv(name); return v && *v ? atoll(v) : dflt; }

// Binned relaunch of one chunk of the batch (items [first, first + n) of the caller's arrays), advanced ONE LAUNCH AT A TIME so that
// the chunks of a call can be interleaved on two streams (round 6, pgdb1_binned_run below).
struct P1Run {
    const fbx_design* des = nullptr; long long first = 0, n = 0; const double* e = nullptr; const double* c = nullptr;
    int tp = 0, mode = 0, max_iters = 0; double* choi = nullptr; int32_t *it = nullptr, *dy = nullptr, *bt = nullptr; double* cost = nullptr;
    int32_t* sw = nullptr; PgdbExtras ex; long long tail_items = 0; int check_every = 8;
    hipStream_t st = nullptr; size_t lds = 0;
    P1Bins bins; int* cnt = nullptr; double* region[2] = {nullptr, nullptr};
    long long active_bound = 0; int step = 0; bool done = false, reading = false;
    int* host_cnt = nullptr;              // [P1_NB], page-locked or plain: the destination of the asynchronous read-back

    static size_t workspace_bytes(long long n, int m) {
        const long long cap = (n + 63) / 64 * 64;
        const size_t region_doubles = (size_t)(P1_NB / 2) * cap * P1_NF;
        return (256 + ((size_t)n * 2 * m + 2 * region_doubles) * sizeof(double) + 255) & ~(size_t)255;
    }
    int start(void* w) {
        const int m = des->dev.m;
        lds = sizeof(double) * 2 * (size_t)m * 64;
        bins.cap_slots = (n + 63) / 64 * 64;
        const size_t region_doubles = (size_t)(P1_NB / 2) * bins.cap_slots * P1_NF, tab_doubles = (size_t)n * 2 * m;
        cnt = (int*)w;                                    // three count arrays of 16 ints
        bins.tab = (double*)((char*)w + 256);
        region[0] = bins.tab + tab_doubles; region[1] = region[0] + region_doubles;
        FBX_HIP(hipMemsetAsync(cnt, 0, 256, st));
        active_bound = n; step = 0; done = n <= 0; reading = false;
        return FBX_OK;
    }
    // enqueue the next launch (and, when due, the read-back of how many reconstructions are left)
    int advance() {
        if (done || reading) return FBX_OK;
        bins.cur = region[step & 1]; bins.next = region[(step + 1) & 1];
        bins.cur_count = cnt + 16 * (step % 3); bins.next_count = cnt + 16 * ((step + 1) % 3); bins.clear_count = cnt + 16 * ((step + 2) % 3);
        const bool tail = step > 0 && active_bound <= tail_items;
        const long long grid = (active_bound + 63) / 64 + (step == 0 ? 0 : P1_NB);
        hipLaunchKernelGGL(pgdb1_step_kernel, dim3((unsigned)grid), dim3(64), lds, st, des->dev, first, n, e, c, tp, mode, max_iters,
                           choi, it, dy, bt, cost, sw, ex.trace, ex.trace_iters, bins, step == 0 ? 1 : 0, tail ? 1 : 0, step);
        FBX_HIP(hipGetLastError());
        ++step;
        if (tail) { done = true; return FBX_OK; }
        if (mode == FBX_MODE_FIXED && step >= max_iters) { done = true; return FBX_OK; }        // every reconstruction runs exactly max_iters iterations
        // how many are left: read back every few launches (the grid shrinks with it), at every launch near the end
        if (step % check_every == 0 || active_bound <= 4 * tail_items) {
            FBX_HIP(hipMemcpyAsync(host_cnt, bins.next_count, sizeof(int) * P1_NB, hipMemcpyDeviceToHost, st));
            reading = true;
        }
        return FBX_OK;
    }
    // wait for a pending read-back (the launches of the OTHER stream keep the device busy meanwhile)
    int settle() {
        if (!reading) return FBX_OK;
        FBX_HIP(hipStreamSynchronize(st));
        long long a = 0;
        for (int k = 0; k < P1_NB; ++k) a += host_cnt[k];
        active_bound = a; reading = false;
        if (a == 0) done = true;
        return FBX_OK;
    }
};

// The binned relaunch of a whole call.  One launch per outer iteration leaves the chip half empty (round 5's counters: a wavefront
// resident on 57 % of the SIMD-cycles of the call -- every launch ends with its slowest wavefront, the last one with the serial
// floor of one lane): the batch is therefore cut into chunks that run as INDEPENDENT binned pipelines, two at a time on two
// streams (the calling thread's and its second compute stream), so that one pipeline's launches fill the other's tails.  Bins are
// per chunk; an item's arithmetic does not know which chunk or stream it ran in (bit-identical outputs, tests/test_pgdb1_gpu.py).
static int pgdb1_binned_run(const fbx_design* des, long long B, const double* e, const double* c, int tp, int mode, int max_iters,
                            double* choi, int32_t* it, int32_t* dy, int32_t* bt, double* cost, int32_t* sw, const PgdbExtras& ex,
                            long long chunk, long long tail_items, int check_every, int n_streams) {
    FBX_HIP(hipFuncSetAttribute((const void*)pgdb1_step_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(sizeof(double) * 2 * (size_t)des->dev.m * 64)));
    const int m = des->dev.m;
    const int lanes = n_streams > 1 && B > chunk ? 2 : 1;         // pipelines in flight
    hipStream_t sts[2] = {stream(), stream()};
    hipEvent_t* ev = nullptr;
    if (lanes == 2) {
        hipStream_t s_in, s_out;
        { const int rc = copy_streams(&s_in, &s_out, &sts[1]); if (rc) return rc; }
        { const int rc = ordering_events(2, &ev); if (rc) return rc; }
    }
    void* w = nullptr;
    const size_t per = P1Run::workspace_bytes(chunk < B ? chunk : B, m);
    // (no room for the bins: the caller falls back to the persistent kernel, which needs none)
    if (workspace(WS_PGDB1_BINS, per * lanes, &w) != FBX_OK) { (void)hipGetLastError(); return FBX_ERR_NOMEM; }
    // (page-locked: a read-back into pageable memory would hold the host until its stream has drained -- and with it the other pipeline's launches)
    static thread_local int* host_cnt_mem = nullptr;
    if (!host_cnt_mem) FBX_HIP(hipHostMalloc((void**)&host_cnt_mem, sizeof(int) * 2 * P1_NB, hipHostMallocDefault));
    int* host_cnt[2] = {host_cnt_mem, host_cnt_mem + P1_NB};
    if (lanes == 2) {                                              // the second stream starts behind whatever the caller queued on the first
        FBX_HIP(hipEventRecord(ev[0], sts[0]));
        FBX_HIP(hipStreamWaitEvent(sts[1], ev[0], 0));
    }
    P1Run run[2];
    long long next = 0;
    int rc = FBX_OK;
    auto feed = [&](int k) -> int {                                // the next chunk of the batch into pipeline k
        P1Run& r = run[k];
        r = P1Run();
        r.des = des; r.first = next; r.n = B - next < chunk ? B - next : chunk; r.e = e; r.c = c; r.tp = tp; r.mode = mode; r.max_iters = max_iters;
        r.choi = choi; r.it = it; r.dy = dy; r.bt = bt; r.cost = cost; r.sw = sw; r.ex = ex; r.tail_items = tail_items; r.check_every = check_every;
        r.st = sts[k]; r.host_cnt = host_cnt[k];
        next += r.n;
        return r.start((char*)w + per * k);
    };
    for (int k = 0; k < lanes && rc == FBX_OK; ++k) { run[k].done = true; if (next < B) rc = feed(k); }
    while (rc == FBX_OK) {
        bool any = false;
        for (int k = 0; k < lanes && rc == FBX_OK; ++k) {
            if (run[k].done && next < B) rc = feed(k);             // (stream order: the new chunk's first launch follows the old one's last)
            if (!run[k].done) { any = true; rc = run[k].advance(); }
        }
        for (int k = 0; k < lanes && rc == FBX_OK; ++k) rc = run[k].settle();
        if (!any) break;
    }
    if (lanes == 2) {                                              // the caller's stream continues behind the second one's last launch
        if (rc != FBX_OK) { (void)hipStreamSynchronize(sts[1]); return rc; }
        FBX_HIP(hipEventRecord(ev[1], sts[1]));
        FBX_HIP(hipStreamWaitEvent(sts[0], ev[1], 0));
    }
    return rc;
}

#ifdef FBX_PHASE_TIMERS
static int pgdb1_dispatch_(const fbx_design* des, int64_t B, const double* e, const double* c, int tp, int mode, int max_iters,
                   double* choi, int32_t* it, int32_t* dy, int32_t* bt, double* cost, int32_t* sw, const PgdbExtras& ex);
int pgdb1_dispatch(const fbx_design* des, int64_t B, const double* e, const double* c, int tp, int mode, int max_iters,
                   double* choi, int32_t* it, int32_t* dy, int32_t* bt, double* cost, int32_t* sw, const PgdbExtras& ex) {
    static unsigned long long z[P1_PROF_STEPS][24];
    memset(z, 0, sizeof(z));
    FBX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_p1_prof), z, sizeof(z)));
    const int rc = pgdb1_dispatch_(des, B, e, c, tp, mode, max_iters, choi, it, dy, bt, cost, sw, ex);
    FBX_HIP(hipStreamSynchronize(stream()));
    FBX_HIP(hipMemcpyFromSymbol(z, HIP_SYMBOL(g_p1_prof), sizeof(z)));
    for (int sidx = 0; sidx < P1_PROF_STEPS; ++sidx) {
        if (!z[sidx][4]) continue;
        fprintf(stderr, "P1PROF %d", sidx);
        for (int k = 0; k < 24; ++k) fprintf(stderr, " %llu", z[sidx][k]);
        fprintf(stderr, "\n");
    }
    return rc;
}
#define pgdb1_dispatch static pgdb1_dispatch_
#endif
int pgdb1_dispatch(const fbx_design* des, int64_t B, const double* e, const double* c, int tp, int mode, int max_iters,
                   double* choi, int32_t* it, int32_t* dy, int32_t* bt, double* cost, int32_t* sw, const PgdbExtras& ex) {
    mode &= 0xff;          // (a call with FBX_MODE_LS_REFERENCE only gets here with pgdb_packed_1q = 2, a diagnostics setting: the flag is dropped)
    {
        // Binned relaunch from the batch sizes at which it wins (scripts/pgdb1_binned_time.py; below them a launch per outer
        // iteration costs more than the grouping returns): to convergence 2^20 experiments (2^19 for designs of <= 12
        // settings), a fixed iteration count -- where nothing ever leaves the batch -- 2^17.  fbx_set_option("pgdb1_binned") = 0 never /
        // 1 by these rules / 2 always; environment, for experiments: FBX_P1_BINNED overrides it per call; FBX_P1_TAIL the number of reconstructions left at which the
        // last launch takes over; FBX_P1_CHUNK the largest number binned at once (its workspace is 3.5 KB + 16 m bytes per reconstruction).
        const long long binned = env_ll("FBX_P1_BINNED", option_pgdb1_binned()), chunk = env_ll("FBX_P1_CHUNK", 0), tail = env_ll("FBX_P1_TAIL", 8192),
                        every = env_ll("FBX_P1_CHECK", 8);
        const long long from = mode == FBX_MODE_FIXED ? (1 << 17) : (des->dev.m <= 12 ? (1 << 19) : (1 << 20));
        // (a stage of the pipelined host entry point shares the calling thread's workspaces with the stage on the other stream:
        // those stay with the persistent kernel)
        if (!ex.launch_stream && !(mode == FBX_MODE_FIXED && max_iters == 0) && (binned == 2 || (binned == 1 && B >= from))) {
            // chunks: FBX_P1_CHUNK, else HALF the batch (at least 2^18 items, at most 2^20): one pipeline per stream.  Measured at 2^20
            // experiments (scripts/pgdb1_streams_time.py, same box, every output bit-identical): a pipeline costs ~22 ms whatever its
            // size -- the launches of the long tail of iteration counts, the serial floor of the slowest lanes -- plus ~7.7 ms per 2^18
            // items, so MORE chunks than streams lose (4 chunks on one stream 121 ms, on two 74 ms against 53 ms for one), and two
            // pipelines that run side by side reach their latency-bound phases together: 53.1 -> 51.9 ms to convergence, 234 -> 212 ms
            // for 30 fixed iterations (Pauli), 34.5 -> 34.7 / 46.7 -> 47.2 ms (SIC).
            long long chunk_ = chunk > 0 ? chunk : (B + 1) / 2;
            if (chunk <= 0) chunk_ = chunk_ < (1 << 18) ? (1 << 18) : (chunk_ > (1 << 20) ? (1 << 20) : chunk_);
            chunk_ = chunk_ < 4096 ? 4096 : (chunk_ > (1 << 22) ? (1 << 22) : chunk_);
            const int rc = pgdb1_binned_run(des, B, e, c, tp, mode, max_iters, choi, it, dy, bt, cost, sw, ex, chunk_, tail, every < 1 ? 1 : (int)every,
                                            (int)env_ll("FBX_P1_STREAMS", 2));
            // out of device memory for the bins (before anything of the chunk was launched: results are per item, the chunks
            // already done stay valid and are simply recomputed): the persistent kernel below takes the whole batch
            if (rc != FBX_ERR_NOMEM) return rc;
        }
    }
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
