// fbx_common.hpp -- shared host/device helpers of libfbx (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <functional>
#include <map>
#include <mutex>
#include "../../include/fbx.h"

namespace fbx {

// ------------------------------------------------------------------ host: errors / per-thread context
// The library is re-entrant across host threads (ctypes drops the GIL during a call): every host
// thread that enters the library owns a context -- its HIP stream, its timer events, its cached
// device workspaces and its pool of staging buffers -- so two threads never share mutable state.
// The only process-wide state is the selected device (fbx_set_device, "one process per GPU") and
// the RCCL communicator (fbx_comm_*).
void set_error(const std::string& msg);
int hip_fail(hipError_t e, const char* what, const char* file, int line);
hipStream_t stream();    // the calling thread's stream (created on first use)
int device_epoch();      // increases whenever fbx_set_device selects a different device: cached device memory is stale then
int current_device();    // device the calling thread works on: the process's (fbx_set_device), or a device worker's own; -1 before the first use
// One call, several devices (fbx_set_devices; SURVEY.md 8b / 8e "host thread per device does H2D of its slab, launches, D2H"):
// the host-pointer batch entry points split their batch into contiguous blocks and hand block g to the long-lived worker thread
// of device list entry g (its own context: stream, staging pool, workspaces).
int device_list_size();                                                  // entries of fbx_set_devices' list, 0 or 1 = single device
bool in_device_worker();                                                 // the calling thread is one of those workers
int run_on_devices(const std::function<int(int, int)>& job);            // job(g, G) on worker g of the G-entry list (G read once, under the list's lock); all waited for; first failure returned
const fbx_design* design_on_this_device(const fbx_design* des, int* rc); // des itself, or its replica on the calling worker's device (created on first use)
double option_pgdb_eig_rel_tol(int n_qubits);   // fbx_set_option("pgdb_eig_rel_tol" / "pgdb3_eig_rel_tol")
bool option_eigh_cooperative();                 // fbx_set_option("eigh_cooperative")
int option_pgdb_pieces();                        // fbx_set_option("pgdb_pieces"): pieces per reconstruction of the two-waves 2-qubit kernel
int option_pgdb1_binned();                       // fbx_set_option("pgdb1_binned"): binned relaunch of the lane-per-item single-qubit kernel 0 / 1 / 2
int option_pgdb_packed_1q();                     // fbx_set_option("pgdb_packed_1q"): single-qubit PGDB on the lane-per-item kernel: 0 never, 1 large batches (default), 2 always
// Defaults of those options: the eigensolver of PGDB's CP projections stops at an off-diagonal norm of
// <this> x the previous outer step (relative to ||H||_F), never tighter than 1e-13; 0 = always 1e-13.
// Surveys against the oracle: 2 qubits, 704 items -- identical deviation histogram at 1e-8
// and at 0; 3 qubits, 66 items -- 9e-11 at 1e-7 and at 0 alike (7e-10 at 3e-7, 4.5e-10 on 18 items at 1e-6, 1.3e-8 at 1e-5).
#define FBX_JTOL_REL 1e-8
#define FBX3_JTOL_REL 1e-7
int ensure_device();     // FBX_OK or FBX_ERR_NO_DEVICE (message set)
int copy_streams(hipStream_t* in, hipStream_t* out, hipStream_t* compute2);   // the calling thread's H2D / D2H / second compute stream (created on first use)
int ordering_events(int n, hipEvent_t** out);            // >= n reusable events of the calling thread (no timing)
bool host_pointer_is_pinned(const void* p, size_t bytes);   // the whole range is page-locked (fbx_host_alloc / hipHostMalloc / hipHostRegister)
long long option_pgdb_host_chunk();                      // fbx_set_option("pgdb_host_chunk")

// per-call arguments of fbx_pgdb_process_ex[_dev] beyond those of fbx_pgdb_process
struct PgdbExtras {
    double eig_rel_tol = -1.0;     // < 0: the process default (fbx_set_option)
    int32_t* trace = nullptr;      // DEVICE [B][trace_iters][2]: Dykstra iterations and halvings of every outer iteration
    int trace_iters = 0;
    // set by the pipelined host-pointer entry point only: the stream a stage's kernels go to, and the stage's share of
    // the per-item workspace: `ws_items` slots in all, this stage's start at `ws_offset` (stages on different streams get disjoint ranges)
    hipStream_t launch_stream = nullptr;
    int64_t ws_items = 0, ws_offset = 0;
    int64_t total_batch = 0;       // size of the caller's whole batch when this launch is one stage of it: kernels are chosen by IT,
                                   // so that a result never depends on how a batch was cut into stages
};

// Named grow-only device workspaces of the calling thread (kept between calls; released by
// fbx_release_workspace).  A workspace only ever serves kernels on the calling thread's stream, so
// growing it (stream sync + hipFree + hipMalloc) cannot pull memory from under another thread's kernel.
enum WorkspaceSlot { WS_PGDB_BASIS = 0, WS_PGDB3_BASIS = 1, WS_SWEEP_REF = 2, WS_COMM = 3, WS_CONVERT = 4, WS_PGDB1_COUNTER = 5, WS_PGDB1_BINS = 6, WS_PGDB_PIECES = 7, WS_COUNT = 8 };
int workspace(WorkspaceSlot slot, size_t bytes, void** out);
// staging blocks of the host-pointer entry points: taken from / returned to the calling thread's pool
int pool_take(size_t bytes, void** out);
void pool_give(void* p);

#define FBX_HIP(call)                                                          \
    do {                                                                       \
        hipError_t _e = (call);                                                \
        if (_e != hipSuccess) return fbx::hip_fail(_e, #call, __FILE__, __LINE__); \
    } while (0)

#define FBX_REQUIRE(cond, msg)                                                 \
    do {                                                                       \
        if (!(cond)) { fbx::set_error(msg); return FBX_ERR_BAD_ARG; }          \
    } while (0)

// Device staging buffer of a host-pointer entry point.  The block comes from the calling thread's
// pool (no hipMalloc / hipFree per call once the pool is warm) and goes back when the entry point
// returns -- every such entry point synchronises its stream before it does.
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) pool_give(p); }
    int alloc(size_t bytes) { return pool_take(bytes ? bytes : 16, &p); }
    template <class T> T* as() { return reinterpret_cast<T*>(p); }
};

// ------------------------------------------------------------------ device-side design
struct DesignDev {
    int n, kind, m, S, d, D;
    int unit_coefs;          // 1 when every observable coefficient is exactly 1
    const int* order;        // [m] grouped position -> caller's setting index
    const uint32_t* sp;      // [m] (grouped) packed (state index << 16 | pauli index)
    const double* coef;      // [m] (grouped)
    const int* sptr;         // [S+1] grouped ranges per distinct input state
    const double* C;         // [D][S] Bloch coefficients c_j(s) = tr(P_j rho_s)
    const double* Ct;        // [S][D] the same, one state per row (read from L2 by the lean PGDB kernel)
    // linear inversion of process designs: settings grouped by observable, and the rows of the
    // block pseudo-inverses (data independent; replaces pinv of tomography.py:482-488)
    const int* porder;       // [m] position grouped by Pauli index -> caller's setting index
    const int* pptr;         // [D+1]
    const double* pinvT;     // [m][D]: R[i][:] = sum_{g in group i} e[porder[g]] * pinvT[g][:]
    double eig_rel_tol;      // PGDB only, filled in by the launcher: fbx_set_option("pgdb[3]_eig_rel_tol")
    int ls_reference;        // PGDB only, filled in by the launcher: FBX_MODE_LS_REFERENCE of the call's mode argument
};

}  // namespace fbx

struct fbx_design {
    fbx::DesignDev dev;
    // what fbx_design_create was given (replicas on other devices are created from it) and the replicas, by device
    int arg_n = 0, arg_kind = 0, arg_m = 0;
    std::vector<uint8_t> arg_in_labels, arg_paulis;
    std::vector<double> arg_coefs;
    mutable std::mutex replica_mu;
    mutable std::map<int, fbx_design*> replicas;
    int device = -1;         // device (and selection epoch) the slabs were allocated on
    int epoch = -1;
    void* slab = nullptr;    // one device allocation backing every pointer in dev
    void* slab2 = nullptr;   // linear-inversion tables (process designs)
    std::vector<double> C_host;
    std::vector<int> order_host;
    std::vector<uint32_t> sp_host;
    std::vector<double> coef_host;
};

namespace fbx {

int check_design(const fbx_design* des, const char* who);   // FBX_OK, or FBX_ERR_BAD_ARG for NULL / a design of another device

// ------------------------------------------------------------------ device helpers
#if defined(__HIPCC__)

// Optional per-phase cycle accounting (build with -DFBX_PHASE_TIMERS; diagnostics only).
#ifdef FBX_PHASE_TIMERS
#define FBX_NPHASE 8
struct PhaseClock {
    long long acc[FBX_NPHASE]; long long t0;
    __device__ void reset() { for (int i = 0; i < FBX_NPHASE; ++i) acc[i] = 0; }
    __device__ __forceinline__ void start() { t0 = __builtin_readcyclecounter(); }
    __device__ __forceinline__ void stop(int ph) { long long t = __builtin_readcyclecounter(); acc[ph] += t - t0; t0 = t; }
};
#define PH_START(pc) (pc).start()
#define PH_STOP(pc, ph) (pc).stop(ph)
#else
struct PhaseClock { __device__ void reset() {} };
#define PH_START(pc)
#define PH_STOP(pc, ph)
#endif

struct __attribute__((aligned(16))) cplx { double re, im; };

// Phase separator for code that runs as ONE wavefront per workgroup.  The LDS instructions of a
// wavefront execute in program order, so a write phase followed by a read phase needs neither
// s_barrier nor a wait -- only that the compiler keeps the order and does not carry LDS values in
// registers across the boundary.  __syncthreads() would add `s_waitcnt vmcnt(0) lgkmcnt(0)`: every
// outstanding LDS access AND every outstanding HBM load / store (the basis prefetch) completed first.
#define FBX_WAVE_SYNC() asm volatile("" ::: "memory")

// Workgroup barrier of the multi-wave kernels.  __syncthreads() is a release / acquire fence on ALL memory plus
// s_barrier: it waits for every outstanding global load and store of the wave (s_waitcnt vmcnt(0)) at every barrier.
// A translation unit whose threads never communicate through global memory inside a kernel (fbx_pgdb3.hip: the
// basis store and the parked state are private per thread) defines FBX_LDS_ONLY_BARRIERS before including the
// headers: the barrier then orders LDS only, and write-backs / prefetches stay in flight across it.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
#ifdef FBX_LDS_ONLY_BARRIERS
#define FBX_BLOCK_SYNC() fbx::lds_barrier()
#else
#define FBX_BLOCK_SYNC() __syncthreads()
#endif

// Wavefront reductions on the VALU: four DPP butterfly steps inside each row of 16 lanes (quad
// swaps, half-row and row mirrors -- no LDS crossbar round trips as with ds_bpermute), then the four
// row results through v_readlane.  Every lane gets the same value (already wave-uniform); the
// summation order is fixed.
template <int CTRL>
__device__ __forceinline__ double dpp_permute(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// row shifts: lanes without a source keep their own value (no zero "old" operand to materialise)
template <int CTRL>
__device__ __forceinline__ double dpp_shift(double v) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_permute<0xB1>(v);          // quad_perm [1,0,3,2]
    v += dpp_permute<0x4E>(v);          // quad_perm [2,3,0,1]
    v += dpp_permute<0x141>(v);         // row_half_mirror
    v += dpp_permute<0x140>(v);         // row_mirror: every lane holds the sum of its row of 16
    return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}
__device__ __forceinline__ double wave_max(double v) {
    v = fmax(v, dpp_permute<0xB1>(v));
    v = fmax(v, dpp_permute<0x4E>(v));
    v = fmax(v, dpp_permute<0x141>(v));
    v = fmax(v, dpp_permute<0x140>(v));
    return fmax(fmax(readlane_f64(v, 0), readlane_f64(v, 16)), fmax(readlane_f64(v, 32), readlane_f64(v, 48)));
}
// sum over all NT threads of the workgroup (every thread receives the same total, summed in one
// fixed order); `red` is NT/64 doubles of LDS scratch, unused for single-wave workgroups
template <int NT>
__device__ __forceinline__ double block_sum(double v, double* red) {
    v = wave_sum(v);
    if constexpr (NT > 64) {
        FBX_BLOCK_SYNC();                       // earlier readers of `red` are done
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
        FBX_BLOCK_SYNC();
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) s += red[w];
        return s;
    }
    return v;
}
// two sums at once: one barrier pair instead of two (a workgroup barrier of 16 wavefronts costs
// thousands of cycles once the waves have drifted apart)
template <int NT>
__device__ __forceinline__ void block_sum2(double& a, double& b, double* red) {
    a = wave_sum(a); b = wave_sum(b);
    if constexpr (NT > 64) {
        FBX_BLOCK_SYNC();
        if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = a; red[NT / 64 + (threadIdx.x >> 6)] = b; }
        FBX_BLOCK_SYNC();
        double sa = 0.0, sb = 0.0;
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) { sa += red[w]; sb += red[NT / 64 + w]; }
        a = sa; b = sb;
    }
}
// make a wave-uniform copy (lane 0's value) so branches on it are scalar
__device__ __forceinline__ double uniform(double v) {
    int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// The same value, but opaque to the optimiser: index arithmetic derived from it is recomputed where it is used (a few integer
// instructions) instead of being hoisted out of every enclosing loop as a "loop-invariant" per-thread constant -- dozens of
// them in the 1024-thread kernels, which have 128 registers: hoisted, they are spilled at kernel entry and re-read from
// scratch (HBM latency, one wait each) in front of every phase.
__device__ __forceinline__ int opaque(int v) { __asm__ volatile("" : "+v"(v)); return v; }

// Raw buffer access (buffer_load / buffer_store with an SGPR resource, ONE 32-bit VGPR byte offset and an SGPR / immediate row
// offset) for per-lane rows of a wave-private slice: no 64-bit per-lane address arithmetic, which the compiler otherwise hoists
// out of the outer loop (one VGPR pair per 4 KB of rows) and then spills.  Reads beyond `bytes` return zero.
typedef unsigned fbx_v2u __attribute__((ext_vector_type(2)));
typedef unsigned fbx_v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ double buf_load_f64(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    const fbx_v2u w = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    return __hiloint2double((int)w.y, (int)w.x);
}
__device__ __forceinline__ void buf_store_f64(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, double v) {
    fbx_v2u w; w.x = (unsigned)__double2loint(v); w.y = (unsigned)__double2hiint(v);
    __builtin_amdgcn_raw_buffer_store_b64(w, r, voff, soff, 0);
}
__device__ __forceinline__ unsigned buf_load_u32(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0);
}

// 1/sqrt(x) for normal positive x to ~1 ulp: hardware estimate (v_rsq_f64, ~2^-26) refined by
// one cubically convergent step  y (1 + e/2 + 3 e^2/8),  e = 1 - x y^2.
// Avoids the IEEE sqrt / divide expansions in the Jacobi rotation's dependent chain.
__device__ __forceinline__ double fast_rsqrt(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    const double e = fma(-x * y, y, 1.0);
    const double p = fma(0.375, e, 0.5);
    return fma(y * e, p, y);
}

// Natural logarithm of a positive, finite, normal double (the PGDB probabilities are clipped to
// [1e-6, ~1]) to < 1 ulp with ~35 instructions instead of the ~100 of the general libm entry:
// x = 2^k (1 + f), sqrt(1/2) <= 1 + f < sqrt(2);  s = f / (2 + f);
// log(1 + f) = f - f^2/2 + s (f^2/2 + R(s^2)),  R = minimax polynomial of Remez type (the classic
// argument reduction of W. Kahan / K. C. Ng used by most libm implementations).
__device__ __forceinline__ double fast_log_pos(double x) {
    int k = __builtin_amdgcn_frexp_exp(x);                 // x = m * 2^k, m in [0.5, 1)
    double m = __builtin_amdgcn_frexp_mant(x);
    const bool low = m < 0.70710678118654752440;
    m = low ? 2.0 * m : m;                                  // [sqrt(1/2), sqrt(2))
    k = low ? k - 1 : k;
    const double f = m - 1.0;
    const double den = 2.0 + f;
    double r = __builtin_amdgcn_rcp(den);
    r = fma(fma(-den, r, 1.0), r, r);
    r = fma(fma(-den, r, 1.0), r, r);
    double s = f * r;
    s = fma(fma(-den, s, f), r, s);                         // correctly rounded-ish quotient
    const double z = s * s, w = z * z;
    const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
    const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01),
                                     2.857142874366239149e-01), 6.666666666666735130e-01);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)k;
    return dk * 6.93147180369123816490e-01 - ((hfsq - fma(s, hfsq + R, dk * 1.90821492927058770002e-10)) - f);
}

// Pauli index (base-4 digits, qubit 0 most significant) -> x / z bit masks over the
// computational index (qubit 0 = most significant bit) and number of Y factors.
template <int NQ>
__device__ __forceinline__ void pauli_masks(int idx, int& x, int& z, int& ny) {
    x = 0; z = 0; ny = 0;
#pragma unroll
    for (int t = 0; t < NQ; ++t) {
        int code = (idx >> (2 * t)) & 3;
        int xb = (code == 1) | (code == 2);
        int zb = (code == 2) | (code == 3);
        x |= xb << t; z |= zb << t; ny += (code == 2);
    }
}
// inverse: masks -> Pauli index.  Digit t is (I, X, Y, Z) = (0, 1, 2, 3) for (x_t, z_t) = (0,0), (1,0), (1,1), (0,1): its high bit is
// z_t and its low bit x_t ^ z_t, so the index is the bit-interleave of z and x ^ z (two Morton spreads instead of a loop over the qubits)
__device__ __forceinline__ int spread_bits8(int v) {      // bit t -> bit 2 t, t < 8
    v = (v | (v << 4)) & 0x0F0F;
    v = (v | (v << 2)) & 0x3333;
    v = (v | (v << 1)) & 0x5555;
    return v;
}
template <int NQ>
__device__ __forceinline__ int pauli_index(int x, int z) {
    static_assert(NQ <= 8, "spread_bits8");
    return (spread_bits8(z) << 1) | spread_bits8(x ^ z);
}
#endif

}  // namespace fbx
