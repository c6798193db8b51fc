// fbx_runtime.hip -- library lifecycle, device memory helpers, HIP-event timing, designs.
#include "fbx_common.hpp"
#include <algorithm>
#include <cmath>
#include <complex>
#include <atomic>
#include <condition_variable>
#include <map>
#include <thread>

namespace fbx {

static thread_local std::string g_err;
static std::atomic<int> g_device{-1};     // process-wide: one process per GPU (the primary device of fbx_set_devices)
static std::atomic<int> g_epoch{0};
static thread_local int t_device = -1;    // device workers (fbx_set_devices): the device this thread is bound to

// Per-thread context (see fbx_common.hpp).  Contexts are heap objects that are never destroyed behind
// the runtime's back (a thread_local destructor could run after the HIP runtime has shut down): a thread
// gives its device memory back with fbx_release_workspace().
struct ThreadCtx {
    int epoch = -1;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timing = false;
    hipStream_t copy_in = nullptr, copy_out = nullptr, compute2 = nullptr;   // H2D / D2H / odd stages of the pipelined host-pointer entry points
    std::vector<hipEvent_t> events;                        // ordering events of that pipeline (no timing)
    struct Block { void* p; size_t bytes; bool busy; };
    std::vector<Block> pool;
    void* ws[WS_COUNT] = {};
    size_t ws_bytes[WS_COUNT] = {};

    void drop_memory() {
        for (auto& b : pool) if (b.p) (void)hipFree(b.p);
        pool.clear();
        for (int i = 0; i < WS_COUNT; ++i) { if (ws[i]) (void)hipFree(ws[i]); ws[i] = nullptr; ws_bytes[i] = 0; }
    }
    void drop_all() {
        drop_memory();
        if (stream) { (void)hipStreamDestroy(stream); stream = nullptr; }
        if (ev0) { (void)hipEventDestroy(ev0); ev0 = nullptr; }
        if (ev1) { (void)hipEventDestroy(ev1); ev1 = nullptr; }
        if (copy_in) { (void)hipStreamDestroy(copy_in); copy_in = nullptr; }
        if (copy_out) { (void)hipStreamDestroy(copy_out); copy_out = nullptr; }
        if (compute2) { (void)hipStreamDestroy(compute2); compute2 = nullptr; }
        for (auto e : events) (void)hipEventDestroy(e);
        events.clear();
        timing = false;
    }
};
static thread_local ThreadCtx* t_ctx = nullptr;

// the calling thread's context, bound to the currently selected device (nullptr + error set on failure)
static ThreadCtx* ctx() {
    if (!t_ctx) t_ctx = new ThreadCtx();
    ThreadCtx* c = t_ctx;
    const int ep = g_epoch.load();
    if (c->epoch != ep) {                 // first use, or the process moved to another device
        c->drop_all();
        if (hipSetDevice(current_device()) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
            (void)hipGetLastError(); c->stream = nullptr; return nullptr;
        }
        c->epoch = ep;
    }
    return c;
}

void set_error(const std::string& msg) { g_err = msg; }

int hip_fail(hipError_t e, const char* what, const char* file, int line) {
    char buf[512];
    snprintf(buf, sizeof buf, "HIP error %d (%s) in %s at %s:%d", (int)e, hipGetErrorString(e),
             what, file, line);
    g_err = buf;
    return e == hipErrorOutOfMemory ? FBX_ERR_NOMEM : FBX_ERR_HIP;
}

int ensure_device() {
    if (g_device.load() < 0) {
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        if (e != hipSuccess || n <= 0) {
            (void)hipGetLastError();
            set_error("libfbx: no HIP device visible -- the MI355X path has no CPU fallback");
            return FBX_ERR_NO_DEVICE;
        }
        int rc = fbx_set_device(0);
        if (rc) return rc;
    }
    if (!ctx()) { set_error("libfbx: could not create the calling thread's HIP stream"); return FBX_ERR_HIP; }
    return FBX_OK;
}

hipStream_t stream() { ThreadCtx* c = ctx(); return c ? c->stream : nullptr; }
int device_epoch() { return g_epoch.load(); }
int current_device() { return t_device >= 0 ? t_device : g_device.load(); }

// ---- fbx_set_devices: one long-lived worker thread per entry of the device list.  A worker is an ordinary client of the library
// bound to its device (t_device): its context -- stream, staging pool, workspaces -- persists between calls, so a multi-device
// call allocates nothing once warm.  Workers are never joined (the process may exit with the HIP runtime already gone): they
// are parked on their condition variable.
namespace {
struct Worker {
    int device = 0;
    std::mutex mu;
    std::condition_variable cv;
    std::function<int()> job;
    bool has_job = false, done = false;
    int rc = FBX_OK;
    std::string err;
    void loop() {
        t_device = device;
        for (;;) {
            std::function<int()> j;
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return has_job; }); j = job; }
            int r = FBX_ERR_HIP;
            g_err.clear();
            if (hipSetDevice(device) == hipSuccess) r = j(); else { (void)hipGetLastError(); g_err = "libfbx: hipSetDevice failed in a device worker"; }
            { std::lock_guard<std::mutex> lk(mu); rc = r; err = g_err; has_job = false; done = true; }
            cv.notify_all();
        }
    }
};
std::mutex g_workers_mu;                  // serialises multi-device calls and changes of the list
std::vector<Worker*> g_workers;           // one per list entry (a device may appear twice: two workers share it)
std::vector<Worker*> g_retired;           // workers of earlier lists, parked with their device memory released; reused by later lists
std::atomic<int> g_worker_count{0};       // g_workers.size(), readable without the lock (the pre-check of the batch entry points)

// hand `j` to worker `w` / wait for it (g_workers_mu held by the caller)
void worker_post(Worker* w, std::function<int()> j) {
    { std::lock_guard<std::mutex> lk(w->mu); w->job = std::move(j); w->done = false; w->has_job = true; }
    w->cv.notify_all();
}
int worker_wait(Worker* w) {
    std::unique_lock<std::mutex> lk(w->mu);
    w->cv.wait(lk, [&] { return w->done; });
    return w->rc;
}
}  // namespace

int device_list_size() { return g_worker_count.load(); }
bool in_device_worker() { return t_device >= 0; }

// job(g, G) for g = 0 .. G - 1 with G = the length of the device list, read ONCE under the lock that fbx_set_devices takes: the
// caller derives its split from the G it is handed, so a concurrent change of the list can neither leave a block uncomputed nor
// address a worker that has gone.  A list of fewer than two entries runs job(0, 1) on the calling thread.
int run_on_devices(const std::function<int(int, int)>& job) {
    std::unique_lock<std::mutex> call(g_workers_mu);
    const int G = (int)g_workers.size();
    if (G < 2) { call.unlock(); return job(0, 1); }
    for (int g = 0; g < G; ++g) worker_post(g_workers[g], [&job, g, G] { return job(g, G); });
    int rc = FBX_OK;
    for (int g = 0; g < G; ++g) {
        Worker* w = g_workers[g];
        const int r = worker_wait(w);
        if (r != FBX_OK && rc == FBX_OK) { rc = r; set_error("device " + std::to_string(w->device) + ": " + w->err); }
    }
    return rc;
}

const fbx_design* design_on_this_device(const fbx_design* des, int* rc) {
    *rc = FBX_OK;
    const int dev = current_device();
    if (des->device == dev) return des;
    std::lock_guard<std::mutex> lk(des->replica_mu);
    auto it = des->replicas.find(dev);
    if (it != des->replicas.end() && it->second->epoch == device_epoch()) return it->second;
    fbx_design* rep = nullptr;
    *rc = fbx_design_create(des->arg_n, des->arg_kind, des->arg_m, des->arg_in_labels.empty() ? nullptr : des->arg_in_labels.data(),
                            des->arg_paulis.data(), des->arg_coefs.empty() ? nullptr : des->arg_coefs.data(), &rep);
    if (*rc) return nullptr;
    if (it != des->replicas.end()) { fbx_design_destroy(it->second); it->second = rep; } else des->replicas[dev] = rep;
    return rep;
}

int copy_streams(hipStream_t* in, hipStream_t* out, hipStream_t* compute2) {
    ThreadCtx* c = ctx();
    if (!c) { set_error("libfbx: no device context"); return FBX_ERR_HIP; }
    if (!c->copy_in) FBX_HIP(hipStreamCreateWithFlags(&c->copy_in, hipStreamNonBlocking));
    if (!c->copy_out) FBX_HIP(hipStreamCreateWithFlags(&c->copy_out, hipStreamNonBlocking));
    if (!c->compute2) {
        // the second compute stream carries the BULK of a pipelined batch: highest priority, so that its workgroups are placed
        // before those of the small first / last stages on the thread's own stream, which then only fill what it leaves idle
        int lo = 0, hi = 0;
        FBX_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
        FBX_HIP(hipStreamCreateWithPriority(&c->compute2, hipStreamNonBlocking, hi));
    }
    *in = c->copy_in; *out = c->copy_out; *compute2 = c->compute2;
    return FBX_OK;
}

int ordering_events(int n, hipEvent_t** out) {
    ThreadCtx* c = ctx();
    if (!c) { set_error("libfbx: no device context"); return FBX_ERR_HIP; }
    while ((int)c->events.size() < n) {
        hipEvent_t e = nullptr;
        FBX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c->events.push_back(e);
    }
    *out = c->events.data();
    return FBX_OK;
}

bool host_pointer_is_pinned(const void* p, size_t bytes) {
    if (!p) return false;
    // first AND last byte of the range that will be handed to hipMemcpyAsync on the copy streams (a caller may pass a
    // buffer that only starts inside a page-locked allocation)
    for (const char* q : {(const char*)p, (const char*)p + (bytes ? bytes - 1 : 0)}) {
        hipPointerAttribute_t a;
        if (hipPointerGetAttributes(&a, q) != hipSuccess) { (void)hipGetLastError(); return false; }   // plain pageable memory
        if (a.type != hipMemoryTypeHost) return false;
    }
    return true;
}

int workspace(WorkspaceSlot slot, size_t bytes, void** out) {
    ThreadCtx* c = ctx();
    if (!c) { set_error("libfbx: no device context"); return FBX_ERR_HIP; }
    if (bytes > c->ws_bytes[slot]) {
        if (c->ws[slot]) {
            FBX_HIP(hipStreamSynchronize(c->stream));       // kernels of this thread may still read the old block
            if (c->compute2) FBX_HIP(hipStreamSynchronize(c->compute2));
            (void)hipFree(c->ws[slot]);
            c->ws[slot] = nullptr; c->ws_bytes[slot] = 0;
        }
        FBX_HIP(hipMalloc(&c->ws[slot], bytes));
        c->ws_bytes[slot] = bytes;
    }
    *out = c->ws[slot];
    return FBX_OK;
}

int pool_take(size_t bytes, void** out) {
    ThreadCtx* c = ctx();
    if (!c) { set_error("libfbx: no device context"); return FBX_ERR_HIP; }
    int best = -1;
    for (int i = 0; i < (int)c->pool.size(); ++i) {
        const auto& b = c->pool[i];
        if (!b.busy && b.bytes >= bytes && (best < 0 || b.bytes < c->pool[best].bytes)) best = i;
    }
    if (best >= 0 && c->pool[best].bytes <= 4 * bytes + (1u << 20)) {
        c->pool[best].busy = true; *out = c->pool[best].p; return FBX_OK;
    }
    if (c->pool.size() >= 24) {            // bound the cache: drop what is idle before adding a block
        FBX_HIP(hipStreamSynchronize(c->stream));
        std::vector<ThreadCtx::Block> keep;
        for (auto& b : c->pool) { if (b.busy) keep.push_back(b); else (void)hipFree(b.p); }
        c->pool.swap(keep);
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) return hip_fail(e, "hipMalloc(staging)", __FILE__, __LINE__);
    c->pool.push_back({p, bytes, true});
    *out = p;
    return FBX_OK;
}

void pool_give(void* p) {
    ThreadCtx* c = t_ctx;
    if (!c) return;
    for (auto& b : c->pool) if (b.p == p) { b.busy = false; return; }
    (void)hipFree(p);                      // block of an earlier device epoch
}

int check_design(const fbx_design* des, const char* who) {
    if (!des) { set_error(std::string(who) + ": NULL design"); return FBX_ERR_BAD_ARG; }
    if (des->device != current_device() || des->epoch != g_epoch.load()) {
        set_error(std::string(who) + ": the design was created on another device (or before fbx_set_device "
                  "changed the device); create it again");
        return FBX_ERR_BAD_ARG;
    }
    return FBX_OK;
}

}  // namespace fbx

using namespace fbx;

// ---- tunables (process-wide, read at launch time)
namespace fbx {
namespace {
std::atomic<double> g_eig_rel_tol2{FBX_JTOL_REL};     // 1- and 2-qubit PGDB (fbx_pgdb.hip)
std::atomic<double> g_eig_rel_tol3{FBX3_JTOL_REL};     // 3-qubit PGDB (fbx_pgdb3.hip)
std::atomic<int> g_eigh_coop{1};                      // large eigendecompositions may use a cooperative launch
std::atomic<long long> g_host_chunk{4096};            // items of the first / last stage of the pipelined host-pointer PGDB entry point
}
long long option_pgdb_host_chunk() { return g_host_chunk.load(); }
double option_pgdb_eig_rel_tol(int n_qubits) { return n_qubits >= 3 ? g_eig_rel_tol3.load() : g_eig_rel_tol2.load(); }
bool option_eigh_cooperative() { return g_eigh_coop.load() != 0; }
std::atomic<int> g_packed_1q{1};                      // single-qubit PGDB: 64 reconstructions per wavefront (fbx_pgdb1.hip)
int option_pgdb_packed_1q() { return g_packed_1q.load(); }
std::atomic<int> g_pieces{8};                         // 2-qubit two-waves kernel: pieces per reconstruction (fbx_pgdb_lean.hip)
std::atomic<int> g_binned_1q{1};                      // single-qubit lane-per-item kernel: binned relaunch 0 never / 1 large batches / 2 always
int option_pgdb_pieces() { return g_pieces.load(); }
int option_pgdb1_binned() { return g_binned_1q.load(); }
}  // namespace fbx

extern "C" {

int fbx_version(void) { return 200; }

const char* fbx_last_error(void) { return g_err.c_str(); }

int fbx_device_count(int* count) {
    FBX_REQUIRE(count != nullptr, "fbx_device_count: NULL argument");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { (void)hipGetLastError(); n = 0; }
    *count = n;
    return FBX_OK;
}

int fbx_set_device(int device_id) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        set_error("libfbx: no HIP device visible -- the MI355X path has no CPU fallback");
        return FBX_ERR_NO_DEVICE;
    }
    FBX_REQUIRE(device_id >= 0 && device_id < n, "fbx_set_device: device id out of range");
    FBX_HIP(hipSetDevice(device_id));
    if (g_device.load() != device_id) {     // every thread context and every design of the old device goes stale
        g_device.store(device_id);
        g_epoch.fetch_add(1);
    }
    if (!ctx()) { set_error("libfbx: could not create the calling thread's HIP stream"); return FBX_ERR_HIP; }
    return FBX_OK;
}

int fbx_set_devices(const int* device_ids, int count) {
    FBX_REQUIRE(count >= 0 && (count == 0 || device_ids != nullptr), "fbx_set_devices: bad arguments");
    FBX_REQUIRE(!in_device_worker(), "fbx_set_devices: called from a device worker");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        set_error("libfbx: no HIP device visible -- the MI355X path has no CPU fallback");
        return FBX_ERR_NO_DEVICE;
    }
    for (int k = 0; k < count; ++k) FBX_REQUIRE(device_ids[k] >= 0 && device_ids[k] < n, "fbx_set_devices: device id out of range");
    if (count > 0) { const int rc = fbx_set_device(device_ids[0]); if (rc) return rc; }
    std::lock_guard<std::mutex> lk(g_workers_mu);
    // Workers are reused -- from the current list first, then from the pool of retired ones -- so that alternating lists
    // ([0] -> [0, 1] -> [0] -> ...) neither start a thread per change nor strand a parked thread with its context.  A worker
    // that leaves the list gives its device memory back (staging pool, workspaces up to the multi-GB basis stores) before it
    // is parked: nothing else could, fbx_release_workspace() only reaches the calling thread's context.
    std::vector<Worker*> next;
    for (int k = 0; k < (count > 1 ? count : 0); ++k) {
        Worker* w = nullptr;
        for (auto& old : g_workers) if (old && old->device == device_ids[k]) { w = old; old = nullptr; break; }
        if (!w) for (auto& old : g_retired) if (old && old->device == device_ids[k]) { w = old; old = nullptr; break; }
        if (!w) { w = new Worker(); w->device = device_ids[k]; std::thread(&Worker::loop, w).detach(); }
        next.push_back(w);
    }
    g_retired.erase(std::remove(g_retired.begin(), g_retired.end(), (Worker*)nullptr), g_retired.end());
    for (Worker* old : g_workers) if (old) { worker_post(old, [] { return fbx_release_workspace(); }); g_retired.push_back(old); }
    for (Worker* old : g_workers) if (old) (void)worker_wait(old);
    g_workers.swap(next);
    g_worker_count.store((int)g_workers.size());
    return FBX_OK;
}

int fbx_device_name(char* buf, size_t len, int* compute_units) {
    int rc = ensure_device();
    if (rc) return rc;
    hipDeviceProp_t prop;
    FBX_HIP(hipGetDeviceProperties(&prop, g_device.load()));
    if (buf && len) snprintf(buf, len, "%s (%s)", prop.name, prop.gcnArchName);
    if (compute_units) *compute_units = prop.multiProcessorCount;
    return FBX_OK;
}

// selected device: ordinal and PCI bus id ("0000:05:00.0") -- what tells two ranks' devices apart
int fbx_device_id(int* ordinal, char* pci_bus_id, size_t len) {
    int rc = ensure_device();
    if (rc) return rc;
    if (ordinal) *ordinal = g_device.load();
    if (pci_bus_id && len) FBX_HIP(hipDeviceGetPCIBusId(pci_bus_id, (int)len, g_device.load()));
    return FBX_OK;
}

int fbx_synchronize(void) {
    int rc = ensure_device();
    if (rc) return rc;
    FBX_HIP(hipStreamSynchronize(stream()));
    return FBX_OK;
}

// ---- tunables (process-wide, read at launch time; state above the extern "C" block)

int fbx_set_option(const char* name, double value) {
    FBX_REQUIRE(name != nullptr, "fbx_set_option: NULL name");
    const std::string n(name);
    if (n == "pgdb_eig_rel_tol" || n == "pgdb3_eig_rel_tol") {
        FBX_REQUIRE(value >= 0.0 && value <= 1e-3, "fbx_set_option: the relative eigensolver tolerance must be in [0, 1e-3]");
        (n == "pgdb_eig_rel_tol" ? fbx::g_eig_rel_tol2 : fbx::g_eig_rel_tol3).store(value);
        return FBX_OK;
    }
    if (n == "eigh_cooperative") { fbx::g_eigh_coop.store(value != 0.0 ? 1 : 0); return FBX_OK; }
    if (n == "pgdb_packed_1q") {
        FBX_REQUIRE(value == 0.0 || value == 1.0 || value == 2.0, "fbx_set_option: pgdb_packed_1q must be 0 (never), 1 (large batches) or 2 (always)");
        fbx::g_packed_1q.store((int)value); return FBX_OK;
    }
    if (n == "pgdb_host_chunk") {
        FBX_REQUIRE(value >= 256.0 && value <= 1048576.0, "fbx_set_option: pgdb_host_chunk must be in [256, 1048576]");
        fbx::g_host_chunk.store((long long)value); return FBX_OK;
    }
    if (n == "pgdb_pieces") {
        FBX_REQUIRE(value >= 1.0 && value <= 64.0 && value == (double)(int)value, "fbx_set_option: pgdb_pieces must be an integer in [1, 64]");
        fbx::g_pieces.store((int)value); return FBX_OK;
    }
    if (n == "pgdb1_binned") {
        FBX_REQUIRE(value == 0.0 || value == 1.0 || value == 2.0, "fbx_set_option: pgdb1_binned must be 0 (never), 1 (large batches) or 2 (always)");
        fbx::g_binned_1q.store((int)value); return FBX_OK;
    }
    set_error("fbx_set_option: unknown option '" + n + "'");
    return FBX_ERR_BAD_ARG;
}

int fbx_get_option(const char* name, double* value) {
    FBX_REQUIRE(name != nullptr && value != nullptr, "fbx_get_option: NULL argument");
    const std::string n(name);
    if (n == "pgdb_eig_rel_tol") { *value = fbx::g_eig_rel_tol2.load(); return FBX_OK; }
    if (n == "pgdb3_eig_rel_tol") { *value = fbx::g_eig_rel_tol3.load(); return FBX_OK; }
    if (n == "eigh_cooperative") { *value = fbx::g_eigh_coop.load(); return FBX_OK; }
    if (n == "pgdb_packed_1q") { *value = fbx::g_packed_1q.load(); return FBX_OK; }
    if (n == "pgdb_host_chunk") { *value = (double)fbx::g_host_chunk.load(); return FBX_OK; }
    if (n == "pgdb_pieces") { *value = fbx::g_pieces.load(); return FBX_OK; }
    if (n == "pgdb1_binned") { *value = fbx::g_binned_1q.load(); return FBX_OK; }
    set_error("fbx_get_option: unknown option '" + n + "'");
    return FBX_ERR_BAD_ARG;
}

int fbx_release_workspace(void) {
    if (g_device.load() < 0) return FBX_OK;
    int worker_rc = FBX_OK;
    if (!in_device_worker() && device_list_size() > 1) {      // the workers of the device list hold the memory of multi-device calls
        std::lock_guard<std::mutex> lk(g_workers_mu);
        for (Worker* w : g_workers) worker_post(w, [] { return fbx_release_workspace(); });
        for (Worker* w : g_workers) { const int rc = worker_wait(w); if (rc != FBX_OK && worker_rc == FBX_OK) worker_rc = rc; }
    }
    if (t_ctx) {
        ThreadCtx* c = t_ctx;
        if (c->stream && c->epoch == g_epoch.load()) FBX_HIP(hipStreamSynchronize(c->stream));
        c->drop_memory();
    }
    return worker_rc;                 // the first failure of a device worker's release, if any
}

int fbx_malloc(void** dev_ptr, size_t bytes) {
    FBX_REQUIRE(dev_ptr != nullptr, "fbx_malloc: NULL argument");
    int rc = ensure_device();
    if (rc) return rc;
    FBX_HIP(hipMalloc(dev_ptr, bytes ? bytes : 16));
    return FBX_OK;
}

int fbx_free(void* dev_ptr) {
    if (!dev_ptr) return FBX_OK;
    FBX_HIP(hipFree(dev_ptr));
    return FBX_OK;
}

// Page-locked host memory: buffers a caller allocates here move at the full PCIe rate and asynchronously, and the
// host-pointer PGDB entry point pipelines them (include/fbx.h).
int fbx_host_alloc(void** host_ptr, size_t bytes) {
    FBX_REQUIRE(host_ptr != nullptr, "fbx_host_alloc: NULL argument");
    int rc = ensure_device();
    if (rc) return rc;
    FBX_HIP(hipHostMalloc(host_ptr, bytes ? bytes : 16, hipHostMallocDefault));
    return FBX_OK;
}

int fbx_host_free(void* host_ptr) {
    if (!host_ptr) return FBX_OK;
    FBX_HIP(hipHostFree(host_ptr));
    return FBX_OK;
}

int fbx_memcpy_h2d(void* dev_dst, const void* host_src, size_t bytes) {
    int rc = ensure_device();
    if (rc) return rc;
    FBX_HIP(hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, stream()));
    FBX_HIP(hipStreamSynchronize(stream()));
    return FBX_OK;
}

int fbx_memcpy_d2h(void* host_dst, const void* dev_src, size_t bytes) {
    int rc = ensure_device();
    if (rc) return rc;
    FBX_HIP(hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, stream()));
    FBX_HIP(hipStreamSynchronize(stream()));
    return FBX_OK;
}

int fbx_timer_begin(void) {
    int rc = ensure_device();
    if (rc) return rc;
    ThreadCtx* c = ctx();
    if (!c->ev0) { FBX_HIP(hipEventCreate(&c->ev0)); FBX_HIP(hipEventCreate(&c->ev1)); }
    FBX_HIP(hipEventRecord(c->ev0, c->stream));
    c->timing = true;
    return FBX_OK;
}

int fbx_timer_end(double* elapsed_ms) {
    ThreadCtx* c = t_ctx;
    FBX_REQUIRE(elapsed_ms != nullptr && c != nullptr && c->timing, "fbx_timer_end without fbx_timer_begin");
    FBX_HIP(hipEventRecord(c->ev1, c->stream));
    FBX_HIP(hipEventSynchronize(c->ev1));
    float ms = 0.f;
    FBX_HIP(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    *elapsed_ms = ms;
    c->timing = false;
    return FBX_OK;
}

// ---------------------------------------------------------------------------- designs
// One-qubit state vectors of pyquil.simulation.matrices.STATES (pyquil==4.5.0), restated
// (SURVEY.md 8a-a3); Bloch components r = (1, tr(X rho), tr(Y rho), tr(Z rho)).
static void bloch_of_state(int code, double r[4]) {
    using cd = std::complex<double>;
    const double s2 = std::sqrt(2.0), s3 = std::sqrt(3.0);
    const double pi = 3.14159265358979323846;
    cd v0, v1;
    switch (code) {
        case 0: v0 = 1 / s2; v1 = 1 / s2; break;
        case 1: v0 = 1 / s2; v1 = -1 / s2; break;
        case 2: v0 = 1 / s2; v1 = cd(0, 1 / s2); break;
        case 3: v0 = 1 / s2; v1 = cd(0, -1 / s2); break;
        case 4: v0 = 1; v1 = 0; break;
        case 5: v0 = 0; v1 = 1; break;
        case 6: v0 = 1; v1 = 0; break;
        case 7: v0 = 1 / s3; v1 = s2 / s3; break;
        case 8: v0 = 1 / s3; v1 = std::exp(cd(0, -2 * pi / 3)) * s2 / s3; break;
        default: v0 = 1 / s3; v1 = std::exp(cd(0, 2 * pi / 3)) * s2 / s3; break;
    }
    cd r00 = v0 * std::conj(v0), r01 = v0 * std::conj(v1), r11 = v1 * std::conj(v1);
    r[0] = (r00 + r11).real();
    r[1] = 2 * r01.real();
    r[2] = -2 * r01.imag();
    r[3] = (r00 - r11).real();
}

// One-sided (Hestenes) Jacobi SVD of a real rows x n matrix `a` (row-major; destroyed): on return
// the columns of a V are mutually orthogonal, w[e] = squared norm of column e (= sigma_e^2) and v
// holds V (n x n, row-major).  Unlike an eigendecomposition of the Gram matrix this keeps zero
// singular values at (eps sigma_max)^2, so rank-deficient blocks are cut off like scipy's pinv does.
static void host_onesided_svd(std::vector<double>& a, int rows, int n, std::vector<double>& w, std::vector<double>& v) {
    v.assign((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) v[(size_t)i * n + i] = 1.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool rotated = false;
        for (int p = 0; p < n - 1; ++p) for (int q = p + 1; q < n; ++q) {
            double al = 0.0, be = 0.0, ga = 0.0;
            for (int k = 0; k < rows; ++k) {
                const double x = a[(size_t)k * n + p], y = a[(size_t)k * n + q];
                al += x * x; be += y * y; ga += x * y;
            }
            if (ga == 0.0 || ga * ga <= 1e-30 * al * be) continue;
            rotated = true;
            const double zeta = (be - al) / (2.0 * ga);
            const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(zeta * zeta + 1.0));
            const double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
            for (int k = 0; k < rows; ++k) {
                const double x = a[(size_t)k * n + p], y = a[(size_t)k * n + q];
                a[(size_t)k * n + p] = c * x - sn * y; a[(size_t)k * n + q] = sn * x + c * y;
            }
            for (int k = 0; k < n; ++k) {
                const double x = v[(size_t)k * n + p], y = v[(size_t)k * n + q];
                v[(size_t)k * n + p] = c * x - sn * y; v[(size_t)k * n + q] = sn * x + c * y;
            }
        }
        if (!rotated) break;
    }
    w.assign(n, 0.0);
    for (int e = 0; e < n; ++e)
        for (int k = 0; k < rows; ++k) w[e] += a[(size_t)k * n + e] * a[(size_t)k * n + e];
}

int fbx_design_create(int n_qubits, int kind, int m, const uint8_t* in_labels,
                      const uint8_t* paulis, const double* coefs, fbx_design** out) {
    FBX_REQUIRE(out != nullptr, "fbx_design_create: NULL out");
    *out = nullptr;
    FBX_REQUIRE(kind == FBX_KIND_STATE || kind == FBX_KIND_PROCESS, "fbx_design_create: bad kind");
    FBX_REQUIRE(n_qubits >= 1 && n_qubits <= (kind == FBX_KIND_STATE ? 5 : 3),
                "fbx_design_create: n_qubits must be 1..3 for process designs, 1..5 for state designs");
    FBX_REQUIRE(m >= 1 && m < 65536, "fbx_design_create: m must be in [1, 65535]");
    FBX_REQUIRE(paulis != nullptr, "fbx_design_create: NULL paulis");
    FBX_REQUIRE(kind == FBX_KIND_STATE || in_labels != nullptr,
                "fbx_design_create: process designs need in_labels");
    const int n = n_qubits, d = 1 << n, D = d * d;
    for (int k = 0; k < m * n; ++k) {
        FBX_REQUIRE(paulis[k] < 4, "fbx_design_create: Pauli code out of range");
        if (kind == FBX_KIND_PROCESS)
            FBX_REQUIRE(in_labels[k] < 10, "fbx_design_create: input-state code out of range");
    }
    int rc = ensure_device();
    if (rc) return rc;

    // distinct input states in first-appearance order
    std::vector<int> sidx(m, 0), pidx(m, 0);
    std::vector<std::vector<int>> states;
    std::map<int, int> key2s;
    for (int k = 0; k < m; ++k) {
        int pk = 0;
        for (int q = 0; q < n; ++q) pk = pk * 4 + paulis[k * n + q];
        pidx[k] = pk;
        int key = 0;
        if (kind == FBX_KIND_PROCESS)
            for (int q = 0; q < n; ++q) key = key * 10 + in_labels[k * n + q];
        auto it = key2s.find(key);
        if (it == key2s.end()) {
            int s = (int)states.size();
            key2s[key] = s;
            std::vector<int> codes(n, 4);
            if (kind == FBX_KIND_PROCESS)
                for (int q = 0; q < n; ++q) codes[q] = in_labels[k * n + q];
            states.push_back(codes);
            sidx[k] = s;
        } else {
            sidx[k] = it->second;
        }
    }
    const int S = (int)states.size();

    auto* des = new fbx_design();
    // Bloch coefficient matrix C[j][s] = prod_q r^{(q)}_{digit_q(j)}(s)
    des->C_host.assign((size_t)D * S, 0.0);
    for (int s = 0; s < S; ++s) {
        std::vector<double> r(4 * n);
        for (int q = 0; q < n; ++q) bloch_of_state(states[s][q], &r[4 * q]);
        for (int j = 0; j < D; ++j) {
            double c = 1.0;
            for (int q = 0; q < n; ++q) {
                int digit = (j >> (2 * (n - 1 - q))) & 3;
                c *= r[4 * q + digit];
            }
            des->C_host[(size_t)j * S + s] = c;
        }
    }
    // stable grouping of settings by input state
    std::vector<int> order(m);
    for (int k = 0; k < m; ++k) order[k] = k;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return sidx[a] < sidx[b]; });
    std::vector<int> sptr(S + 1, 0);
    for (int k = 0; k < m; ++k) sptr[sidx[k] + 1]++;
    for (int s = 0; s < S; ++s) sptr[s + 1] += sptr[s];
    des->order_host = order;
    des->sp_host.resize(m);
    des->coef_host.resize(m);
    int unit = 1;
    for (int g = 0; g < m; ++g) {
        int k = order[g];
        des->sp_host[g] = ((uint32_t)sidx[k] << 16) | (uint32_t)pidx[k];
        des->coef_host[g] = coefs ? coefs[k] : 1.0;
        if (des->coef_host[g] != 1.0) unit = 0;
    }

    // one slab: C | Ct | coef | order | sp | sptr
    size_t oC = 0, oCt = oC + sizeof(double) * D * S, oCoef = oCt + sizeof(double) * D * S, oOrder = oCoef + sizeof(double) * m;
    size_t oSp = oOrder + sizeof(int) * m, oPtr = oSp + sizeof(uint32_t) * m;
    size_t total = oPtr + sizeof(int) * (S + 1);
    std::vector<char> host(total);
    memcpy(&host[oC], des->C_host.data(), sizeof(double) * D * S);
    {
        double* ct = reinterpret_cast<double*>(&host[oCt]);
        for (int j = 0; j < D; ++j) for (int s = 0; s < S; ++s) ct[(size_t)s * D + j] = des->C_host[(size_t)j * S + s];
    }
    memcpy(&host[oCoef], des->coef_host.data(), sizeof(double) * m);
    memcpy(&host[oOrder], order.data(), sizeof(int) * m);
    memcpy(&host[oSp], des->sp_host.data(), sizeof(uint32_t) * m);
    memcpy(&host[oPtr], sptr.data(), sizeof(int) * (S + 1));
    hipError_t e = hipMalloc(&des->slab, total);
    if (e != hipSuccess) { delete des; return hip_fail(e, "hipMalloc(design)", __FILE__, __LINE__); }
    e = hipMemcpy(des->slab, host.data(), total, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(des->slab); delete des;
        return hip_fail(e, "hipMemcpy(design)", __FILE__, __LINE__);
    }
    char* base = (char*)des->slab;
    des->dev.n = n; des->dev.kind = kind; des->dev.m = m; des->dev.S = S; des->dev.d = d;
    des->dev.D = D; des->dev.unit_coefs = unit;
    des->dev.C = (const double*)(base + oC);
    des->dev.Ct = (const double*)(base + oCt);
    des->dev.coef = (const double*)(base + oCoef);
    des->dev.order = (const int*)(base + oOrder);
    des->dev.sp = (const uint32_t*)(base + oSp);
    des->dev.sptr = (const int*)(base + oPtr);
    des->dev.porder = nullptr; des->dev.pptr = nullptr; des->dev.pinvT = nullptr; des->dev.eig_rel_tol = 0.0;

    if (kind == FBX_KIND_PROCESS) {
        // ---- linear-inversion tables.  In the orthonormal operator basis {P_j^T (x) P_i / d} the
        // measurement matrix of tomography.py:482-486 is block diagonal over the observable index i
        // with blocks Abar_i[k][j] = coef_k * c_j(s_k); its pseudo-inverse is the block-wise one.
        std::vector<int> porder(m), pptr(D + 1, 0);
        for (int k = 0; k < m; ++k) porder[k] = k;
        std::stable_sort(porder.begin(), porder.end(), [&](int a, int b) { return pidx[a] < pidx[b]; });
        for (int k = 0; k < m; ++k) pptr[pidx[k] + 1]++;
        for (int i = 0; i < D; ++i) pptr[i + 1] += pptr[i];
        std::vector<double> pinvT((size_t)m * D, 0.0);
        struct Block { std::vector<double> w, v; };
        std::map<std::vector<double>, int> cache;          // signature -> first group with it
        std::vector<Block> blocks(D);
        std::vector<int> alias(D, -1);
        double smax2 = 0.0;
        for (int i = 0; i < D; ++i) {
            const int g0 = pptr[i], g1 = pptr[i + 1];
            if (g0 == g1) continue;
            std::vector<double> sig;
            for (int g = g0; g < g1; ++g) { sig.push_back(sidx[porder[g]]); sig.push_back(coefs ? coefs[porder[g]] : 1.0); }
            auto it = cache.find(sig);
            if (it != cache.end()) { alias[i] = it->second; continue; }
            cache[sig] = i; alias[i] = i;
            std::vector<double> abar((size_t)(g1 - g0) * D, 0.0);
            for (int g = g0; g < g1; ++g) {
                const int k = porder[g]; const double ck = coefs ? coefs[k] : 1.0;
                for (int a = 0; a < D; ++a) abar[(size_t)(g - g0) * D + a] = ck * des->C_host[(size_t)a * S + sidx[k]];
            }
            host_onesided_svd(abar, g1 - g0, D, blocks[i].w, blocks[i].v);
            for (double l : blocks[i].w) smax2 = std::max(smax2, l);
        }
        // scipy.linalg.pinv cut-off: singular values <= max(M, N) * eps * sigma_max are dropped
        const double rt = (double)std::max(m, D * D) * 2.220446049250313e-16;
        const double cut2 = rt * rt * smax2;
        for (int i = 0; i < D; ++i) {
            const int g0 = pptr[i], g1 = pptr[i + 1];
            if (g0 == g1) continue;
            const Block& bl = blocks[alias[i]];
            // w = sigma^2: (Abar^T Abar)^+ = V diag(1/sigma^2) V^T ; pinv(Abar) = that times Abar^T -> column of setting k
            std::vector<double> gp((size_t)D * D, 0.0);
            for (int e = 0; e < D; ++e) {
                if (!(bl.w[e] > cut2)) continue;
                const double il = 1.0 / bl.w[e];
                for (int a = 0; a < D; ++a) {
                    const double va = bl.v[(size_t)a * D + e] * il;
                    if (va == 0.0) continue;
                    for (int b = 0; b < D; ++b) gp[(size_t)a * D + b] += va * bl.v[(size_t)b * D + e];
                }
            }
            for (int g = g0; g < g1; ++g) {
                const int k = porder[g]; const double ck = coefs ? coefs[k] : 1.0;
                for (int a = 0; a < D; ++a) {
                    double acc = 0.0;
                    for (int b = 0; b < D; ++b) acc += gp[(size_t)a * D + b] * ck * des->C_host[(size_t)b * S + sidx[k]];
                    pinvT[(size_t)g * D + a] = acc;
                }
            }
        }
        size_t oPo = 0, oPp = oPo + sizeof(int) * m, oPi = (oPp + sizeof(int) * (D + 1) + 15) & ~(size_t)15;
        size_t total2 = oPi + sizeof(double) * (size_t)m * D;
        std::vector<char> host2(total2);
        memcpy(&host2[oPo], porder.data(), sizeof(int) * m);
        memcpy(&host2[oPp], pptr.data(), sizeof(int) * (D + 1));
        memcpy(&host2[oPi], pinvT.data(), sizeof(double) * (size_t)m * D);
        e = hipMalloc(&des->slab2, total2);
        if (e == hipSuccess) e = hipMemcpy(des->slab2, host2.data(), total2, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            if (des->slab2) (void)hipFree(des->slab2);
            (void)hipFree(des->slab); delete des;
            return hip_fail(e, "hipMalloc/hipMemcpy(design linear-inversion tables)", __FILE__, __LINE__);
        }
        char* b2 = (char*)des->slab2;
        des->dev.porder = (const int*)(b2 + oPo);
        des->dev.pptr = (const int*)(b2 + oPp);
        des->dev.pinvT = (const double*)(b2 + oPi);
    }
    des->device = current_device(); des->epoch = device_epoch();
    des->arg_n = n_qubits; des->arg_kind = kind; des->arg_m = m;
    if (in_labels) des->arg_in_labels.assign(in_labels, in_labels + (size_t)m * n);
    des->arg_paulis.assign(paulis, paulis + (size_t)m * n);
    if (coefs) des->arg_coefs.assign(coefs, coefs + m);
    *out = des;
    return FBX_OK;
}

int fbx_design_destroy(fbx_design* design) {
    if (!design) return FBX_OK;
    for (auto& kv : design->replicas) fbx_design_destroy(kv.second);
    design->replicas.clear();
    if (design->slab) (void)hipFree(design->slab);
    if (design->slab2) (void)hipFree(design->slab2);
    delete design;
    return FBX_OK;
}

int fbx_design_info(const fbx_design* design, int* n_qubits, int* kind, int* m,
                    int* n_input_states) {
    FBX_REQUIRE(design != nullptr, "fbx_design_info: NULL design");
    if (n_qubits) *n_qubits = design->dev.n;
    if (kind) *kind = design->dev.kind;
    if (m) *m = design->dev.m;
    if (n_input_states) *n_input_states = design->dev.S;
    return FBX_OK;
}

}  // extern "C"
