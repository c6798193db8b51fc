// fbx_comm.hip -- the multi-GPU side of libfbx: one process per GPU, RCCL over xGMI.
//
// The reconstruction path shards on the batch axis with no exchange during compute (SURVEY.md 8e;
// natural unit in the reference: one entry of get_results_by_qubit_groups,
// observable_estimation.py:1145-1173, one bootstrap resample, tomography.py:440-451, one Kraus set).
// What ranks do exchange is small and happens outside the estimators:
//   * ncclBroadcast  of design-sized constants / a reference channel        (fbx_comm_broadcast_dev)
//   * ncclAllGather  of result slabs when the consumer wants all of them    (fbx_comm_allgather_dev)
//   * ncclAllReduce  (sum / max) of a summary vector of a few doubles       (fbx_comm_allreduce_f64)
// xGMI is point to point (7 links x ~153 GB/s per GPU): a 32 MiB slab per peer is ~0.2 ms, the
// summary vector is latency only -- ring size never matters for this path.
//
// librccl.so is opened on the first fbx_comm_* call (dlopen), not linked: it is a 570 MB image, and
// single-GPU users of libfbx.so never touch it.  Only the types of <rccl/rccl.h> are used at build time.
#include "fbx_common.hpp"
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <dlfcn.h>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <thread>
#include <rccl/rccl.h>

namespace fbx {

namespace {
struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommCuDevice)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
};
Rccl g_rccl;
std::mutex g_mu;             // guards the loader, the state below and the communicator pointer -- never held across a
                             // blocking RCCL call (ncclCommInitRank waits for every peer)
ncclComm_t g_comm = nullptr;
int g_rank = 0, g_world = 0, g_comm_device = -1;
// NONE -> INITIALISING (one fbx_comm_init in flight, g_mu released) -> READY | NONE; an initialisation that
// timed out leaves ABANDONED: its helper thread still sits inside ncclCommInitRank (there is no handle to abort
// before it returns), so no further communicator can be formed in this process, and every fbx_comm_* call
// says so at once instead of blocking.
enum CommState { COMM_NONE = 0, COMM_INITIALISING = 1, COMM_READY = 2, COMM_ABANDONED = 3 };
CommState g_state = COMM_NONE;

template <class F> bool sym(void* h, const char* name, F& out) {
    out = reinterpret_cast<F>(dlsym(h, name));
    return out != nullptr;
}

int load_rccl() {
    if (g_rccl.handle) return FBX_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (h) break; }
    if (!h) { set_error(std::string("fbx_comm: cannot open librccl.so: ") + dlerror()); return FBX_ERR_RCCL; }
    Rccl r; r.handle = h;
    const bool ok = sym(h, "ncclGetUniqueId", r.GetUniqueId) && sym(h, "ncclCommInitRank", r.CommInitRank) &&
                    sym(h, "ncclCommDestroy", r.CommDestroy) && sym(h, "ncclCommAbort", r.CommAbort) &&
                    sym(h, "ncclAllGather", r.AllGather) && sym(h, "ncclAllReduce", r.AllReduce) &&
                    sym(h, "ncclBroadcast", r.Broadcast) && sym(h, "ncclGetErrorString", r.GetErrorString) &&
                    sym(h, "ncclGetVersion", r.GetVersion) && sym(h, "ncclCommCount", r.CommCount) &&
                    sym(h, "ncclCommCuDevice", r.CommCuDevice) && sym(h, "ncclCommUserRank", r.CommUserRank);
    if (!ok) { dlclose(h); set_error("fbx_comm: librccl.so lacks an expected symbol"); return FBX_ERR_RCCL; }
    g_rccl = r;
    return FBX_OK;
}

int rccl_fail(ncclResult_t r, const char* what) {
    set_error(std::string("RCCL error in ") + what + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?"));
    return FBX_ERR_RCCL;
}
#define FBX_RCCL(call, what) do { ncclResult_t _r = (call); if (_r != ncclSuccess) return rccl_fail(_r, what); } while (0)

// A collective USES the communicator under a shared lock taken together with the state check; fbx_comm_destroy takes the
// exclusive one, so it cannot pull the communicator from under a call that is still enqueuing on it (round-3 review).
std::shared_mutex g_use;
struct CommUse { std::shared_lock<std::shared_mutex> lk; ncclComm_t comm = nullptr; int world = 0; };
int need_comm(const char* who, CommUse& use) {
    use.lk = std::shared_lock<std::shared_mutex>(g_use);
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_state == COMM_INITIALISING) { set_error(std::string(who) + ": fbx_comm_init is still in progress on another thread"); return FBX_ERR_BAD_ARG; }
    if (g_state == COMM_ABANDONED) { set_error(std::string(who) + ": an earlier fbx_comm_init timed out; this process cannot form a communicator any more"); return FBX_ERR_RCCL; }
    if (!g_comm) { set_error(std::string(who) + ": no communicator (call fbx_comm_init first)"); return FBX_ERR_BAD_ARG; }
    if (g_comm_device != current_device()) {
        set_error(std::string(who) + ": the communicator belongs to another device"); return FBX_ERR_BAD_ARG;
    }
    use.comm = g_comm; use.world = g_world;
    return FBX_OK;
}
}  // namespace

}  // namespace fbx

using namespace fbx;

extern "C" {

int fbx_comm_unique_id(uint8_t* id_out) {
    FBX_REQUIRE(id_out != nullptr, "fbx_comm_unique_id: NULL argument");
    std::lock_guard<std::mutex> lk(g_mu);
    int rc = load_rccl();
    if (rc) return rc;
    ncclUniqueId id;
    FBX_RCCL(g_rccl.GetUniqueId(&id), "ncclGetUniqueId");
    static_assert(sizeof(id) == FBX_COMM_ID_BYTES, "ncclUniqueId size");
    memcpy(id_out, &id, sizeof id);
    return FBX_OK;
}

// One initialisation: the collective ncclCommInitRank runs in a helper thread that owns no library context
// (it only selects the device), while the caller waits on a condition variable with a deadline.
namespace {
struct InitBox {
    std::mutex mu; std::condition_variable cv;
    bool done = false, abandoned = false; ncclResult_t result = ncclSuccess; hipError_t hip = hipSuccess; ncclComm_t comm = nullptr;
};
}

int fbx_comm_init_timeout(const uint8_t* id_in, int rank, int world, double timeout_seconds) {
    FBX_REQUIRE(id_in != nullptr, "fbx_comm_init: NULL id");
    FBX_REQUIRE(world >= 1 && rank >= 0 && rank < world, "fbx_comm_init: need 0 <= rank < world");
    FBX_REQUIRE(timeout_seconds > 0.0, "fbx_comm_init: timeout must be positive");
    int rc = ensure_device();
    if (rc) return rc;
    const int device = current_device();
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (g_state == COMM_ABANDONED) { set_error("fbx_comm_init: an earlier initialisation timed out; this process cannot form a communicator any more"); return FBX_ERR_RCCL; }
        FBX_REQUIRE(g_state != COMM_INITIALISING, "fbx_comm_init: another initialisation is in progress");
        FBX_REQUIRE(g_state == COMM_NONE && g_comm == nullptr, "fbx_comm_init: a communicator already exists (fbx_comm_destroy first)");
        rc = load_rccl();
        if (rc) return rc;
        g_state = COMM_INITIALISING;
    }
    ncclUniqueId id;
    memcpy(&id, id_in, sizeof id);
    auto box = std::make_shared<InitBox>();
    const auto init_rank = g_rccl.CommInitRank;
    const auto comm_abort = g_rccl.CommAbort;
    std::thread([box, init_rank, comm_abort, id, rank, world, device]() {
        ncclComm_t comm = nullptr;
        hipError_t he = hipSetDevice(device);
        ncclResult_t r = he == hipSuccess ? init_rank(&comm, world, id, rank) : ncclUnhandledCudaError;
        bool orphan;
        {
            std::lock_guard<std::mutex> lk(box->mu);
            box->hip = he; box->result = r; box->comm = comm; box->done = true;
            orphan = box->abandoned;
            box->cv.notify_all();
        }
        // the caller gave up waiting: nobody will ever own this communicator -- release it here instead of leaking it
        if (orphan && r == ncclSuccess && comm) (void)comm_abort(comm);
    }).detach();
    bool finished;
    {
        std::unique_lock<std::mutex> lk(box->mu);
        finished = box->cv.wait_for(lk, std::chrono::duration<double>(timeout_seconds), [&] { return box->done; });
        if (!finished) box->abandoned = true;
    }
    std::lock_guard<std::mutex> lk(g_mu);
    if (!finished) {
        g_state = COMM_ABANDONED;
        char msg[160];
        snprintf(msg, sizeof msg, "fbx_comm_init: ncclCommInitRank (rank %d of %d) did not return within %.0f s -- a peer is missing", rank, world, timeout_seconds);
        set_error(msg);
        return FBX_ERR_RCCL;
    }
    if (box->hip != hipSuccess) { g_state = COMM_NONE; return hip_fail(box->hip, "hipSetDevice (fbx_comm_init)", __FILE__, __LINE__); }
    if (box->result != ncclSuccess) { g_state = COMM_NONE; return rccl_fail(box->result, "ncclCommInitRank"); }
    g_comm = box->comm; g_rank = rank; g_world = world; g_comm_device = device; g_state = COMM_READY;
    return FBX_OK;
}

int fbx_comm_init(const uint8_t* id_in, int rank, int world) {
    double limit = 180.0;
    if (const char* e = getenv("FBX_RCCL_INIT_TIMEOUT")) { const double v = atof(e); if (v > 0.0) limit = v; }
    return fbx_comm_init_timeout(id_in, rank, world, limit);
}

int fbx_comm_info(int* rank, int* world, int* rccl_version) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (rank) *rank = g_comm ? g_rank : 0;
    if (world) *world = g_comm ? g_world : 0;
    if (rccl_version) {
        *rccl_version = 0;
        if (g_rccl.handle) { int v = 0; if (g_rccl.GetVersion(&v) == ncclSuccess) *rccl_version = v; }
    }
    return FBX_OK;
}

// What the COMMUNICATOR says about itself (ncclCommUserRank / ncclCommCount / ncclCommCuDevice), not what the
// launcher's environment claimed: bench.py prints these per rank.
int fbx_comm_query(int* rank, int* world, int* device) {
    CommUse use;
    int rc = need_comm("fbx_comm_query", use);
    if (rc) return rc;
    int r = -1, w = -1, d = -1;
    FBX_RCCL(g_rccl.CommUserRank(use.comm, &r), "ncclCommUserRank");
    FBX_RCCL(g_rccl.CommCount(use.comm, &w), "ncclCommCount");
    FBX_RCCL(g_rccl.CommCuDevice(use.comm, &d), "ncclCommCuDevice");
    if (rank) *rank = r;
    if (world) *world = w;
    if (device) *device = d;
    return FBX_OK;
}

int fbx_comm_destroy(void) {
    ncclComm_t c;
    std::unique_lock<std::shared_mutex> excl(g_use);        // waits for collectives that are still enqueuing on the communicator
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (g_state == COMM_INITIALISING) { set_error("fbx_comm_destroy: fbx_comm_init is still in progress on another thread"); return FBX_ERR_BAD_ARG; }
        if (!g_comm) return FBX_OK;              // nothing to destroy (also after an abandoned initialisation)
        c = g_comm;
        g_comm = nullptr; g_world = 0; g_rank = 0; g_comm_device = -1; g_state = COMM_NONE;
    }
    FBX_RCCL(g_rccl.CommDestroy(c), "ncclCommDestroy");
    return FBX_OK;
}

int fbx_comm_allgather_dev(const void* d_send, void* d_recv, size_t bytes_per_rank) {
    CommUse use;
    int rc = need_comm("fbx_comm_allgather_dev", use);
    if (rc) return rc;
    FBX_REQUIRE(bytes_per_rank == 0 || (d_send && d_recv), "fbx_comm_allgather_dev: NULL buffer");
    if (bytes_per_rank == 0) return FBX_OK;
    // slabs are complex128 / float64 / int32 arrays: moved as bytes (8-byte words when the size allows)
    if (bytes_per_rank % 8 == 0)
        FBX_RCCL(g_rccl.AllGather(d_send, d_recv, bytes_per_rank / 8, ncclUint64, use.comm, stream()), "ncclAllGather");
    else
        FBX_RCCL(g_rccl.AllGather(d_send, d_recv, bytes_per_rank, ncclUint8, use.comm, stream()), "ncclAllGather");
    return FBX_OK;
}

int fbx_comm_broadcast_dev(void* d_buf, size_t bytes, int root) {
    CommUse use;
    int rc = need_comm("fbx_comm_broadcast_dev", use);
    if (rc) return rc;
    FBX_REQUIRE(root >= 0 && root < use.world, "fbx_comm_broadcast_dev: root out of range");
    FBX_REQUIRE(bytes == 0 || d_buf, "fbx_comm_broadcast_dev: NULL buffer");
    if (bytes == 0) return FBX_OK;
    FBX_RCCL(g_rccl.Broadcast(d_buf, d_buf, bytes, ncclUint8, root, use.comm, stream()), "ncclBroadcast");
    return FBX_OK;
}

int fbx_comm_allreduce_f64_dev(const double* d_send, double* d_recv, size_t n, int op) {
    CommUse use;
    int rc = need_comm("fbx_comm_allreduce_f64_dev", use);
    if (rc) return rc;
    FBX_REQUIRE(op == FBX_COMM_SUM || op == FBX_COMM_MAX || op == FBX_COMM_MIN, "fbx_comm_allreduce_f64_dev: bad op");
    FBX_REQUIRE(n == 0 || (d_send && d_recv), "fbx_comm_allreduce_f64_dev: NULL buffer");
    if (n == 0) return FBX_OK;
    const ncclRedOp_t rop = op == FBX_COMM_SUM ? ncclSum : op == FBX_COMM_MAX ? ncclMax : ncclMin;
    FBX_RCCL(g_rccl.AllReduce(d_send, d_recv, n, ncclDouble, rop, use.comm, stream()), "ncclAllReduce");
    return FBX_OK;
}

int fbx_comm_allreduce_f64(double* host_inout, size_t n, int op) {
    int rc;
    { CommUse use; rc = need_comm("fbx_comm_allreduce_f64", use); }     // (checked here; each piece below takes its own shared lock)
    if (rc) return rc;
    FBX_REQUIRE(n == 0 || host_inout, "fbx_comm_allreduce_f64: NULL buffer");
    if (n == 0) return FBX_OK;
    // any length: pieces of at most CHUNK doubles through one small staging workspace
    constexpr size_t CHUNK = 4096;
    void* w = nullptr;
    rc = workspace(WS_COMM, sizeof(double) * CHUNK, &w);
    if (rc) return rc;
    double* d = (double*)w;
    for (size_t o = 0; o < n; o += CHUNK) {
        const size_t k = n - o < CHUNK ? n - o : CHUNK;
        FBX_HIP(hipMemcpyAsync(d, host_inout + o, sizeof(double) * k, hipMemcpyHostToDevice, stream()));
        rc = fbx_comm_allreduce_f64_dev(d, d, k, op);
        if (rc) return rc;
        FBX_HIP(hipMemcpyAsync(host_inout + o, d, sizeof(double) * k, hipMemcpyDeviceToHost, stream()));
        FBX_HIP(hipStreamSynchronize(stream()));
    }
    return FBX_OK;
}

int fbx_comm_barrier(void) {
    // every kernel this thread enqueued is complete on every rank when this returns
    double one = 1.0;
    return fbx_comm_allreduce_f64(&one, 1, FBX_COMM_SUM);
}

}  // extern "C"
