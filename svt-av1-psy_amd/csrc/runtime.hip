// runtime.hip -- device binding, per-thread host-call context, instruction self-test.
#include <mutex>
#include <vector>
#include <atomic>

#include "svt_hip_common.h"
#include "../../include/svtav1_hip.h"

namespace svthip { // (one per kernel file: SVT_HIP_DEFINE_WARM)
void warm_cdef(hipStream_t st);
void warm_cdef_pick(hipStream_t st);
void warm_deblock(hipStream_t st);
void warm_hme(hipStream_t st);
void warm_lr_search(hipStream_t st);
void warm_lr_stats(hipStream_t st);
void warm_me_results(hipStream_t st);
void warm_me_session(hipStream_t st);
void warm_misc(hipStream_t st);
void warm_picprep(hipStream_t st);
void warm_pme(hipStream_t st);
void warm_quant(hipStream_t st);
void warm_restoration(hipStream_t st);
void warm_sad(hipStream_t st);
void warm_tf(hipStream_t st);
void warm_tf_picture(hipStream_t st);
void warm_tf_subpel(hipStream_t st);
void warm_tpl(hipStream_t st);
void warm_tpl_full(hipStream_t st);
void warm_txfm(hipStream_t st);
void warm_txfm_fused(hipStream_t st);
}

namespace svthip {

static std::atomic<int>  g_device{-1};         // the default device (svt_hip_init)
static std::atomic<bool> g_initialised{false}; // written by svt_hip_init, read by every worker thread on first entry
static std::atomic<bool> g_multi{false};       // a thread or a session has asked for a device of its own: the binding is re-established on every entry
static char g_name[256]   = "uninitialised";

// Device selection is PER HOST THREAD (as HIP's own current device): svt_hip_set_thread_device() / a session's DeviceGuard set t_want, everything else runs on
// the default device.  One encoder process can therefore drive several GPUs -- a seam binds the calling worker thread to the device its picture is sharded to
// (frame-level sharding) and every host-call arena, stream and session below belongs to exactly one device.
static thread_local int t_want  = -1; // device this thread asked for (-1: the default)
static thread_local int t_bound = -1; // device this library last made current on this thread

int current_device() { return t_want >= 0 ? t_want : (int)g_device; }

// LOGICAL devices.  Every device number of this ABI (svt_hip_init, svt_hip_set_thread_device, a session's device, a partition's device list, SVT_HIP_DEVICES /
// SVT_HIP_STRIPS of the encoder binding) is a logical one: logical device d runs on physical GPU d % (number of GPUs), and everything this library keeps per device --
// host-call arenas and their streams, the lease pool, the resident-plane table, the ticket ring, sessions, partition peers with their streams / events / arenas -- is
// keyed by the LOGICAL number.  With SVT_HIP_VIRTUAL_DEVICES=V (or svt_hip_set_virtual_devices(V)) there are V logical devices per GPU: a node with ONE MI355X then
// runs the whole multi-device code -- per-device sessions, peer streams, the ready / done event protocol, hipMemcpyPeerAsync (legal with both ends on one GPU) -- truly
// concurrently on an asynchronous device.  Default V = 1: logical = physical.
static std::atomic<int> g_virtual{0}; // 0: not read from the environment yet
static int virtual_per_gpu() {
    int v = g_virtual.load(std::memory_order_acquire);
    if (v == 0) {
        const char* e = getenv("SVT_HIP_VIRTUAL_DEVICES");
        v = e ? atoi(e) : 1;
        if (v < 1) v = 1;
        if (v > MAX_DEVICES) v = MAX_DEVICES;
        g_virtual.store(v, std::memory_order_release);
    }
    return v;
}
static int physical_count() {
    static std::atomic<int> n{-1};
    int v = n.load();
    if (v < 0) {
        int c = 0;
        if (hipGetDeviceCount(&c) != hipSuccess) { (void)hipGetLastError(); c = 0; }
        n.store(v = c);
    }
    return v;
}
int physical_device(int logical) {
    const int n = physical_count();
    return n > 0 ? logical % n : logical;
}

// ---- error policy (svt_hip_common.h: HIP_CHECK) ------------------------------------------------------------------------------------------------------------
static std::atomic<bool> g_failed{false};
static std::mutex        g_fail_m;
static char              g_last_error[512] = "";
bool failed() { return g_failed.load(std::memory_order_acquire); }
extern "C" void svt_hip_rtcd_unhook(void); // rtcd_hook.hip
void device_fail(const int code, const char* what, const char* file, const int line) {
    {
        std::lock_guard<std::mutex> g(g_fail_m);
        if (!g_failed.load()) {
            snprintf(g_last_error, sizeof(g_last_error), "%s failed: %s (%s:%d)", what, code > 0 ? hipGetErrorString((hipError_t)code) : "library limit", file, line);
            fprintf(stderr, "libsvtav1_hip: %s -- the device path is off from here on: dispatch pointers restored, stage entry points decline\n", g_last_error);
            g_failed.store(true, std::memory_order_release);
            svt_hip_rtcd_unhook();
        }
    }
    (void)hipGetLastError();
    throw DeviceError{code};
}

void ensure_device() {
    if (failed()) throw DeviceError{-1}; // (every entry point opens with ensure_device: after the first error nothing touches the device any more)
    if (!g_initialised) {
        if (svt_hip_init(0) != 0) device_fail(-2, "svt_hip_init(0): no usable HIP device (gfx950 expected)", __FILE__, __LINE__);
    }
    // the current device is per host thread: the encoder's worker threads (pointers are called concurrently from ME / EncDec /
    // CDEF / REST threads) bind to their device the first time they enter the library, and again whenever it changes
    const int d = current_device();
    if (t_bound != d || g_multi) {
        HIP_CHECK(hipSetDevice(physical_device(d)));
        t_bound = d;
    }
}

DeviceGuard::DeviceGuard(int device) : prev(t_want) {
    t_want = device;
    g_multi = true;
    ensure_device();
}
DeviceGuard::~DeviceGuard() { t_want = prev; } // (the previous device is made current again at this thread's next entry into the library)

static thread_local HostCall t_calls[MAX_DEVICES]; // one arena + stream per (thread, device)

HostCall& host_call() {
    ensure_device();
    HostCall& c = t_calls[current_device()];
    if (!c.stream) HIP_CHECK(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
    return c;
}

static std::mutex              g_lease_m;
static std::vector<HostCall*> g_lease_pool[MAX_DEVICES];
HostCallLease::HostCallLease() {
    ensure_device();
    device = current_device();
    c = nullptr;
    {
        std::lock_guard<std::mutex> g(g_lease_m);
        if (!g_lease_pool[device].empty()) { c = g_lease_pool[device].back(); g_lease_pool[device].pop_back(); }
    }
    if (!c) {
        c = new HostCall();
        HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    }
}
HostCallLease::~HostCallLease() {
    std::lock_guard<std::mutex> g(g_lease_m);
    g_lease_pool[device].push_back(c); // (most recently used first: the arena that has already grown is the one that is taken again)
}

// Side streams + events for a stage that runs independent launch sequences side by side and joins them (the loop-restoration search: the self-guided search's groups
// of parameter sets on two streams, the Wiener refinement -- a chain of short dependent launches -- on a third one with the highest priority, so that its launches are
// not queued behind the long self-guided workgroups).  The sets are POOLED per device, not kept per thread: an encoder calls the stage from whichever worker thread
// holds the picture, and a set made per thread cost every new thread three stream creations -- 5 ms per call at 1080p (profiles/r05_lr_seam_calls.txt).  A set is
// taken for the duration of a call and handed back with its work possibly still in flight: streams are in-order and every event is recorded again before it is waited
// for, so the next user's work simply queues behind.
static std::mutex                   g_streams_m;
static std::vector<ThreadStreams*> g_streams_pool[MAX_DEVICES];
static void stream_set_free(ThreadStreams* set) { // the device of the set is current
    for (hipStream_t s : set->st)
        if (s) (void)hipStreamDestroy(s);
    for (hipEvent_t e : set->ev)
        if (e) (void)hipEventDestroy(e);
    if (set->pinned) (void)hipHostFree(set->pinned);
    delete set;
}
static void stream_pool_free_all() { // svt_hip_shutdown
    std::lock_guard<std::mutex> g(g_streams_m);
    for (int d = 0; d < MAX_DEVICES; d++) {
        if (g_streams_pool[d].empty()) continue;
        (void)hipSetDevice(physical_device(d));
        for (ThreadStreams* s : g_streams_pool[d]) stream_set_free(s);
        g_streams_pool[d].clear();
    }
}
StreamSetLease::StreamSetLease() {
    ensure_device();
    device = current_device();
    set = nullptr;
    {
        std::lock_guard<std::mutex> g(g_streams_m);
        if (!g_streams_pool[device].empty()) { set = g_streams_pool[device].back(); g_streams_pool[device].pop_back(); }
    }
    if (!set) {
        set = new ThreadStreams();
        try {
            int lo = 0, hi = 0;
            if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { (void)hipGetLastError(); lo = hi = 0; } // (hi = the numerically lowest value = the highest priority)
            HIP_CHECK(hipStreamCreateWithFlags(&set->st[0], hipStreamNonBlocking));
            HIP_CHECK(hipStreamCreateWithFlags(&set->st[1], hipStreamNonBlocking));
            if (hipStreamCreateWithPriority(&set->st[2], hipStreamNonBlocking, hi) != hipSuccess) { (void)hipGetLastError(); HIP_CHECK(hipStreamCreateWithFlags(&set->st[2], hipStreamNonBlocking)); }
            for (hipEvent_t& e : set->ev) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            HIP_CHECK(hipHostMalloc((void**)&set->pinned, 64 * sizeof(int32_t), hipHostMallocDefault));
        } catch (...) { // a half-built set is taken apart again (a constructor that throws runs no destructor)
            stream_set_free(set);
            set = nullptr;
            throw;
        }
    }
}
StreamSetLease::~StreamSetLease() {
    std::lock_guard<std::mutex> g(g_streams_m);
    g_streams_pool[device].push_back(set);
}

// Four zeroed device words for ONE launch sequence on `st` (tickets / counters of a kernel that orders its own workgroups): slots of a per-device ring, cleared in
// stream order.  A slot is taken again after RING later requests -- far more than launch sequences are ever in flight on one device.
uint32_t* stream_scratch_u32x4(hipStream_t st) {
    constexpr uint32_t RING = 4096;
    static std::mutex            m;
    static std::atomic<uint32_t*> ring[MAX_DEVICES]; // (read outside the mutex by every worker thread: atomic)
    static std::atomic<uint32_t>  next[MAX_DEVICES];
    ensure_device();
    const int d = current_device();
    uint32_t* base = ring[d].load(std::memory_order_acquire);
    if (!base) {
        std::lock_guard<std::mutex> g(m);
        base = ring[d].load(std::memory_order_relaxed);
        if (!base) {
            HIP_CHECK(hipMalloc((void**)&base, (size_t)RING * 16));
            ring[d].store(base, std::memory_order_release);
        }
    }
    uint32_t* p = base + 4 * (next[d].fetch_add(1) % RING);
    HIP_CHECK(hipMemsetAsync(p, 0, 16, st));
    return p;
}

// ---- device-resident copies of host picture planes, across host calls (the TPL stage: a picture of a TPL group is the source of one dispenser call and a
// reference of several others; its TPL reconstruction is produced on the device and read by later pictures) ------------------------------------------------------------
// Key = (host buffer, id): the caller names the CONTENT (e.g. picture number + 1), so a pool buffer that is reused for another picture never aliases.  An entry is
// LOADING while the call that creates it runs (other callers then take their own staged copy) and READY after that call's synchronisation; entries are pinned for the
// duration of a call and recycled least-recently-used.  One table per device.
namespace {
struct CachedPlane { uintptr_t ptr = 0; uint64_t id = 0; uint8_t* dev = nullptr; size_t cap = 0; uint64_t stamp = 0; int pins = 0, state = 0; }; // state: 0 empty, 1 loading, 2 ready
constexpr int PLANE_CACHE_ENTRIES = 32;
struct PlaneCache { std::mutex m; CachedPlane e[PLANE_CACHE_ENTRIES]; uint64_t clock = 0, hits = 0, misses = 0; std::vector<void*> slabs; /* every device allocation made for the table */ };
PlaneCache g_plane_cache[MAX_DEVICES];
} // namespace
// -> the entry's device buffer, pinned; *hit = its content is valid.  nullptr: not cacheable right now (someone else is loading it, or every entry is pinned).
uint8_t* plane_cache_acquire(const void* host_ptr, uint64_t id, size_t bytes, bool* hit, int* token) {
    *hit = false; *token = -1;
    if (!host_ptr || !id) return nullptr;
    ensure_device();
    const int   dev_ = current_device();
    PlaneCache& C = g_plane_cache[dev_];
    std::lock_guard<std::mutex> g(C.m);
    int victim = -1;
    for (int i = 0; i < PLANE_CACHE_ENTRIES; i++) {
        CachedPlane& e = C.e[i];
        if (e.state && e.ptr == (uintptr_t)host_ptr && e.id == id) {
            if (e.state == 1 || e.cap < bytes) return nullptr;
            e.pins++; e.stamp = ++C.clock; *hit = true; *token = dev_ * PLANE_CACHE_ENTRIES + i; C.hits++;
            return e.dev;
        }
        if (e.pins == 0 && e.state != 1 && (victim < 0 || e.stamp < C.e[victim].stamp)) victim = i;
    }
    if (victim < 0) return nullptr;
    CachedPlane& e = C.e[victim];
    if (e.cap < bytes) {
        // Rounded up to 4 MB (the source planes and the reconstruction planes of one encode differ by their borders only, and a recycled entry must fit either), and
        // taken eight entries at a time: hipMalloc / hipFree synchronise the device, which an encoder with several stages in flight pays for -- after the first
        // few pictures the table allocates nothing.  An outgrown slice is left to the table's slab list (freed at shutdown).
        const size_t cap = align_up(bytes, (size_t)4 << 20);
        int fresh[8], nf = 0;
        fresh[nf++] = victim;
        for (int i = 0; i < PLANE_CACHE_ENTRIES && nf < 8; i++)
            if (i != victim && !C.e[i].dev && C.e[i].state == 0 && C.e[i].pins == 0) fresh[nf++] = i;
        uint8_t* slab = nullptr;
        if (hipMalloc((void**)&slab, cap * nf) != hipSuccess) { (void)hipGetLastError(); return nullptr; } // (not cacheable right now: the caller stages the plane in its own arena)
        C.slabs.push_back(slab);
        for (int k = 0; k < nf; k++) { C.e[fresh[k]].dev = slab + (size_t)k * cap; C.e[fresh[k]].cap = cap; }
    }
    e.ptr = (uintptr_t)host_ptr; e.id = id; e.state = 1; e.pins = 1; e.stamp = ++C.clock; C.misses++;
    *token = dev_ * PLANE_CACHE_ENTRIES + victim;
    return e.dev;
}
void plane_cache_release(int token, bool now_ready) { // (the token names the table it came from: the releasing thread's current device does not matter)
    if (token < 0) return;
    PlaneCache& C = g_plane_cache[token / PLANE_CACHE_ENTRIES];
    std::lock_guard<std::mutex> g(C.m);
    CachedPlane& e = C.e[token % PLANE_CACHE_ENTRIES];
    if (e.state == 1) e.state = now_ready ? 2 : 0;
    if (e.pins > 0) e.pins--;
}
// the host rewrote (or is about to rewrite) the buffer by other means: whatever the device holds of it is stale
void plane_cache_drop(const void* host_ptr) { // (on every device: the host buffer is one, its mirrors may be several)
    for (int d = 0; d < MAX_DEVICES; d++) {
        PlaneCache& C = g_plane_cache[d];
        std::lock_guard<std::mutex> g(C.m);
        for (CachedPlane& e : C.e)
            if (e.state == 2 && e.pins == 0 && e.ptr == (uintptr_t)host_ptr) e.state = 0;
    }
}
void plane_cache_counts(uint64_t* hits, uint64_t* misses) {
    PlaneCache& C = g_plane_cache[current_device()];
    std::lock_guard<std::mutex> g(C.m);
    *hits = C.hits; *misses = C.misses;
}
static void plane_cache_free_all() {
    for (int d = 0; d < MAX_DEVICES; d++) {
        PlaneCache& C = g_plane_cache[d];
        std::lock_guard<std::mutex> g(C.m);
        for (void* p : C.slabs) { (void)hipSetDevice(physical_device(d)); (void)hipFree(p); }
        C.slabs.clear();
        for (CachedPlane& e : C.e) e = CachedPlane();
    }
}

static std::atomic<uint64_t> g_commit_violations{0};
void HostCall::touch() {
    if (committed) g_commit_violations.fetch_add(1, std::memory_order_relaxed);
}
void HostCall::begin() {
    dev_used  = 0;
    pin_used  = 0;
    n_pend    = 0;
    committed = false;
    zc        = false;
}
void HostCall::begin_small() {
    begin();
    static const bool off = [] { const char* e = getenv("SVT_HIP_NO_ZERO_COPY"); return e && *e && *e != '0'; }(); // (A/B measurements)
    zc = !off;
}
void HostCall::reserve(size_t dev_bytes, size_t pin_bytes) {
    touch();
    dev_bytes += 4096;
    pin_bytes += 4096;
    if (zc) { // everything lives in the pinned arena
        pin_bytes += dev_bytes + 4096;
        dev_bytes = 0;
    }
    if (dev_bytes > dev_cap) {
        if (dev) HIP_CHECK(hipFree(dev));
        dev_cap = align_up(dev_bytes * 2, 1 << 20);
        HIP_CHECK(hipMalloc((void**)&dev, dev_cap));
    }
    if (pin_bytes > pin_cap) {
        if (pin) HIP_CHECK(hipHostFree(pin));
        pin_cap = align_up(pin_bytes * 2, 1 << 20);
        HIP_CHECK(hipHostMalloc((void**)&pin, pin_cap, hipHostMallocDefault));
    }
    // SVT_HIP_POISON=<byte> (debugging aid): both arenas are filled with the byte at the start of every host call, so that a kernel or a download that reads what the
    // call never wrote gives a result that depends on the byte -- reproducibly, also on the CPU emulator -- instead of on whatever the previous call left behind
    static const int poison = [] { const char* e = getenv("SVT_HIP_POISON"); return e ? (int)strtol(e, nullptr, 0) & 0xff : -1; }();
    if (poison >= 0) {
        HIP_CHECK(hipMemsetAsync(dev, poison, dev_cap, stream));
        memset(pin, poison, pin_cap);
    }
}
void* HostCall::dalloc(size_t bytes) {
    if (zc) { // (same 256-byte alignment as the device arena)
        pin_used = align_up(pin_used, 256);
        return palloc(bytes);
    }
    size_t off = align_up(dev_used, 256);
    if (off + bytes > dev_cap) {
        device_fail(-3, "host-call device arena overflow", __FILE__, __LINE__);
    }
    dev_used = off + bytes;
    return dev + off;
}
void* HostCall::palloc(size_t bytes) {
    size_t off = align_up(pin_used, 64);
    if (off + bytes > pin_cap) {
        device_fail(-3, "host-call pinned arena overflow", __FILE__, __LINE__);
    }
    pin_used = off + bytes;
    return pin + off;
}
void HostCall::up2d(void* ddst, size_t dpitch, const void* hsrc, size_t spitch, size_t width_bytes, size_t rows) {
    touch();
    if (zc) { // the destination IS host memory the GPU reads in place
        for (size_t y = 0; y < rows; y++) memcpy((uint8_t*)ddst + y * dpitch, (const uint8_t*)hsrc + y * spitch, width_bytes);
        return;
    }
    // pack through the pinned buffer so the device copy is one contiguous DMA
    uint8_t* p = (uint8_t*)palloc(dpitch * rows);
    for (size_t y = 0; y < rows; y++) memcpy(p + y * dpitch, (const uint8_t*)hsrc + y * spitch, width_bytes);
    HIP_CHECK(hipMemcpyAsync(ddst, p, dpitch * rows, hipMemcpyHostToDevice, stream));
}
// Page-locked by the caller through svt_hip_host_register?  Then the copy engine reads the source directly: no staging pass through the arena.  The ranges
// are kept by the library (filled by svt_hip_host_register / _unregister below): the WHOLE of [p, p + bytes) has to lie inside ONE registered range -- a source
// that starts inside a registered buffer and runs past its end is staged like pageable memory -- and no runtime query is made per upload.
struct LockedRange { uintptr_t lo, hi; };
static std::mutex               g_locked_m;
static std::vector<LockedRange> g_locked;
static std::atomic<int>         g_locked_n{0}; // (read without the lock: most processes never register anything)
static bool host_range_is_locked(const void* p, size_t bytes) {
    if (g_locked_n.load(std::memory_order_acquire) == 0) return false;
    const uintptr_t lo = (uintptr_t)p, hi = lo + bytes;
    std::lock_guard<std::mutex> g(g_locked_m);
    for (const LockedRange& r : g_locked)
        if (lo >= r.lo && hi <= r.hi) return true;
    return false;
}
void HostCall::up(void* ddst, const void* hsrc, size_t bytes) {
    touch();
    if (zc) { memcpy(ddst, hsrc, bytes); return; }
    if (bytes >= (256u << 10) && host_range_is_locked(hsrc, bytes)) { // (every host form synchronises before it returns: the source outlives the copy)
        if (hipMemcpyAsync(ddst, hsrc, bytes, hipMemcpyHostToDevice, stream) == hipSuccess) return;
        (void)hipGetLastError(); // the runtime refused the direct copy: fall through to the staged one
    }
    uint8_t* p = (uint8_t*)palloc(bytes);
    memcpy(p, hsrc, bytes);
    HIP_CHECK(hipMemcpyAsync(ddst, p, bytes, hipMemcpyHostToDevice, stream));
}
void HostCall::down(void* hdst, const void* dsrc, size_t bytes) {
    touch();
    if (zc) { // the kernel wrote pinned host memory: visible once the stream has drained
        HIP_CHECK(hipStreamSynchronize(stream));
        memcpy(hdst, dsrc, bytes);
        committed = true;
        return;
    }
    uint8_t* p = (uint8_t*)palloc(bytes);
    HIP_CHECK(hipMemcpyAsync(p, dsrc, bytes, hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    memcpy(hdst, p, bytes);
    committed = true;
}
void HostCall::down2d(void* hdst, size_t hpitch, const void* dsrc, size_t dpitch, size_t width_bytes, size_t rows) {
    touch();
    if (zc) {
        HIP_CHECK(hipStreamSynchronize(stream));
        for (size_t y = 0; y < rows; y++) memcpy((uint8_t*)hdst + y * hpitch, (const uint8_t*)dsrc + y * dpitch, width_bytes);
        committed = true;
        return;
    }
    uint8_t* p = (uint8_t*)palloc(dpitch * rows);
    HIP_CHECK(hipMemcpyAsync(p, dsrc, dpitch * rows, hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    for (size_t y = 0; y < rows; y++) memcpy((uint8_t*)hdst + y * hpitch, p + y * dpitch, width_bytes);
    committed = true;
}
void HostCall::down2d_later(void* hdst, size_t hpitch, const void* dsrc, size_t dpitch, size_t width_bytes, size_t rows) {
    touch();
    if (n_pend == 16) finish();
    if (zc) { // nothing to enqueue: finish() copies from where the kernel wrote
        pend[n_pend++] = Pending{hdst, (const uint8_t*)dsrc, hpitch, dpitch, width_bytes, rows};
        return;
    }
    uint8_t* p = (uint8_t*)palloc(dpitch * rows);
    HIP_CHECK(hipMemcpyAsync(p, dsrc, dpitch * rows, hipMemcpyDeviceToHost, stream));
    pend[n_pend++] = Pending{hdst, p, hpitch, dpitch, width_bytes, rows};
}
void HostCall::down_later(void* hdst, const void* dsrc, size_t bytes) { down2d_later(hdst, bytes, dsrc, bytes, bytes, 1); }
void HostCall::finish() {
    touch();
    HIP_CHECK(hipStreamSynchronize(stream));
    for (int i = 0; i < n_pend; i++)
        for (size_t y = 0; y < pend[i].rows; y++) memcpy((uint8_t*)pend[i].h + y * pend[i].hpitch, pend[i].p + y * pend[i].dpitch, pend[i].width);
    committed = n_pend > 0 || committed;
    n_pend    = 0;
}
void HostCall::sync() {
    touch();
    HIP_CHECK(hipStreamSynchronize(stream));
}
extern "C" uint64_t svt_hip_debug_commit_violations(void) { return g_commit_violations.load(std::memory_order_relaxed); }

static std::atomic<int>  g_tune_lr_ur{32}, g_tune_cdef_gpw{0}, g_tune_cdef_minb{3}, g_tune_sad_form{0};
static std::atomic<bool> g_tune_loaded{false}; // stored last, with release ordering: a getter that sees it sees every knob
static std::mutex        g_tune_m;
static void tuning_read() {
    std::lock_guard<std::mutex> g(g_tune_m);
    const char* a = getenv("SVT_HIP_LR_UR");
    const char* b = getenv("SVT_HIP_CDEF_GPW");
    const int   ur = a ? atoi(a) : 32, gp = b ? atoi(b) : 0;
    g_tune_lr_ur    = (ur == 16 || ur == 64) ? ur : 32;
    g_tune_cdef_gpw = (gp == 1 || gp == 2 || gp == 4) ? gp : 0;
    const char* sf = getenv("SVT_HIP_SAD_FORM");
    g_tune_sad_form = (sf && atoi(sf) == 1) ? 1 : 0;
    const char* m = getenv("SVT_HIP_CDEF_MINB");
    g_tune_cdef_minb = (m && atoi(m) == 2) ? 2 : 3;
    g_tune_loaded.store(true, std::memory_order_release);
}
static inline void tuning_ensure() {
    if (!g_tune_loaded.load(std::memory_order_acquire)) tuning_read();
}
int tuning_lr_rows_per_workgroup() {
    tuning_ensure();
    return g_tune_lr_ur;
}
int tuning_sad_form() {
    tuning_ensure();
    return g_tune_sad_form;
}
int tuning_cdef_search_minb() {
    tuning_ensure();
    return g_tune_cdef_minb;
}
int tuning_cdef_groups_per_workgroup() {
    tuning_ensure();
    return g_tune_cdef_gpw;
}

} // namespace svthip

// ---------------------------------------------------------------------------------------------------------------
// Instruction self-test: one wave executes each cross-lane / packed-byte primitive the kernels rely on and dumps
// the results; tests/test_gpu_primitives.py compares them with the C models in tests/emu/hipemu.h, so that a wrong
// reading of the ISA shows up as ONE named mismatch instead of as a parity failure deep inside a kernel.
// ---------------------------------------------------------------------------------------------------------------
__global__ void svt_hip_selftest_kernel(uint32_t* out) {
    const int      l = threadIdx.x;
    const uint32_t a = 0x01020304u * (uint32_t)(l + 1) + 0x9e3779b9u * (uint32_t)l;
    const uint32_t b = 0x10305070u ^ (0x85ebca6bu * (uint32_t)(l + 3));
    const uint64_t w = ((uint64_t)b << 32) | a;
    out[0 * 64 + l]  = __builtin_amdgcn_sad_u8(a, b, 7u);
    uint64_t q       = __builtin_amdgcn_qsad_pk_u16_u8(w, a ^ 0x55aa00ffu, 0x0001000200030004ull);
    out[1 * 64 + l]  = (uint32_t)q;
    out[2 * 64 + l]  = (uint32_t)(q >> 32);
    out[3 * 64 + l]  = __builtin_amdgcn_alignbyte(b, a, (uint32_t)l);
    out[4 * 64 + l]  = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)a, 0xB1, 0xf, 0xf, false);  // quad_perm [1,0,3,2]
    out[5 * 64 + l]  = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)a, 0x4E, 0xf, 0xf, false);  // quad_perm [2,3,0,1]
    out[6 * 64 + l]  = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)a, 0x124, 0xf, 0xf, false); // row_ror:4
    out[7 * 64 + l]  = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)a, 0x128, 0xf, 0xf, false); // row_ror:8
    out[8 * 64 + l]  = (uint32_t)__shfl_xor((int)a, 16);
    out[9 * 64 + l]  = (uint32_t)__shfl_xor((int)a, 32);
    out[10 * 64 + l] = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)a, 0x111, 0xf, 0xf, false); // row_shr:1
    out[11 * 64 + l] = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)a, 0x101, 0xf, 0xf, false); // row_shl:1
    const auto s16 = __builtin_amdgcn_permlane16_swap(a, b, false, false); // gfx950: rows 1, 3 of vdst <-> rows 0, 2 of src0
    out[12 * 64 + l] = s16[0];
    out[13 * 64 + l] = s16[1];
    const auto s32 = __builtin_amdgcn_permlane32_swap(a, b, false, false); // lanes 32-63 of vdst <-> lanes 0-31 of src0
    out[14 * 64 + l] = s32[0];
    out[15 * 64 + l] = s32[1];
    out[16 * 64 + l] = __builtin_amdgcn_udot4(a, b, 7u, false);
}

// Instruction-rate probe (DESIGN.md "measured instruction rates"): every lane runs `iters` rounds of 8 independent
// chains of one VALU opcode, so time / (iters * 8 * lanes) is that opcode's issue cost.
template <int KIND> __global__ __launch_bounds__(256) void svt_hip_rate_kernel(uint32_t iters, uint32_t* sink) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    uint32_t       a[8];
    uint64_t       w[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        a[i] = t * 2654435761u + i * 40503u;
        w[i] = ((uint64_t)a[i] << 32) | (a[i] ^ 0x5bd1e995u);
    }
    const uint32_t k = t | 1u;
    for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (KIND == 0) a[i] = __builtin_amdgcn_sad_u8(a[i], k, a[i]);
            if (KIND == 1) w[i] = __builtin_amdgcn_qsad_pk_u16_u8(w[i], k, w[i]);
            if (KIND == 2) a[i] = a[i] + (a[i] ^ k);
            if (KIND == 3) a[i] = a[i] * k + 1u;
            if (KIND == 4) w[i] = (uint64_t)((int64_t)(int32_t)a[i] * (int64_t)(int32_t)k + (int64_t)w[i]);
            if (KIND == 5) a[i] = __builtin_amdgcn_alignbyte(a[i], k, a[i]);
            // KIND 6 / 7 / 8 ("hybrid SAD issue"): the same 16 absolute differences per chain and round as one v_qsad_pk_u16_u8 -- four byte
            // positions x four bytes -- with 2 / 4 / 8 of the eight chains computed by v_sad_u8 on funnel-shifted operands instead (3 v_alignbyte_b32 + 4 v_sad_u8;
            // the 32-bit sums are NOT repacked into the u16 lanes the search's reductions want, so the mix is priced optimistically).  If a mix is not faster than
            // KIND 1 here, no split of the search's SAD work between the two opcodes can be.
            if (KIND >= 6) {
                const int nsad = KIND == 6 ? 2 : (KIND == 7 ? 4 : 8);
                if (i >= 8 - nsad) {
                    uint32_t lo = (uint32_t)w[i], hi = (uint32_t)(w[i] >> 32);
                    SVT_HIP_OPAQUE_I32(lo); // (the operands of a search step come from the window ring: not loop invariant)
                    a[i] = __builtin_amdgcn_sad_u8(lo, k, a[i]);
                    a[i] = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(hi, lo, 1u), k, a[i]);
                    a[i] = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(hi, lo, 2u), k, a[i]);
                    a[i] = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(hi, lo, 3u), k, a[i]);
                } else {
                    w[i] = __builtin_amdgcn_qsad_pk_u16_u8(w[i], k, w[i]);
                }
            }
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r ^= a[i] ^ (uint32_t)w[i] ^ (uint32_t)(w[i] >> 32);
    if (r == 0x12345u) sink[0] = r;
}

// Memory-traffic probe (bench.py's calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE: MI355X_MICROARCH.md says the gfx950 factor of 2 holds for wide coalesced streaming
// reads only and that every other access shape has to be calibrated on a known byte count).  Lane t of the grid moves W bytes of segment (t / (seg / W)), at offset
// (t % (seg / W)) * W inside it; segment s starts at s * pitch -- so (W = 16, seg = pitch) is a contiguous 16 B/lane stream, (16, 64, 2056) is the 64-byte rows of a padded
// 1080p luma plane that the independent-pair SAD kernel walks, (4, ...) the dword accesses of the search kernels.  READ: every byte is loaded once and folded into a word
// that is (never) stored; WRITE: every byte is stored once.
struct alignas(8) ProbeW2 { uint32_t x, y; };
struct alignas(16) ProbeW4 { uint32_t x, y, z, w; };
template <int W, bool WRITE> __global__ __launch_bounds__(256) void svt_hip_mem_probe_kernel(uint8_t* __restrict__ base, const uint64_t lanes, const uint32_t seg, const uint32_t pitch,
                                                                                             uint32_t* __restrict__ sink) {
    const uint32_t per = seg / W;
    uint32_t       acc = 0;
    for (uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x; t < lanes; t += (uint64_t)gridDim.x * 256) {
        uint8_t* p = base + (t / per) * (uint64_t)pitch + (t % per) * W;
        if constexpr (W == 4) {
            if (WRITE) *(uint32_t*)p = (uint32_t)t;
            else acc ^= *(const uint32_t*)p;
        } else if constexpr (W == 8) {
            if (WRITE) *(ProbeW2*)p = ProbeW2{(uint32_t)t, 1u};
            else { const ProbeW2 v = *(const ProbeW2*)p; acc ^= v.x ^ v.y; }
        } else {
            if (WRITE) *(ProbeW4*)p = ProbeW4{(uint32_t)t, 1u, 2u, 3u};
            else { const ProbeW4 v = *(const ProbeW4*)p; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
        }
    }
    if (!WRITE && acc == 0x9e3779b9u) sink[0] = acc;
}

// The same for the access shape of the block kernels: the buffer is a picture of `pitch`-byte rows; 64x64-byte blocks tile it (`bpr` blocks per block row, starting
// `mis` bytes into the row -- a reference block sits at an arbitrary byte offset); 256 consecutive lanes read one block as 64 rows x 4 lanes x 16 bytes, the way
// sad_nxm_pipe_kernel's waves do.  Every byte of every block is read once: lanes * 16 bytes.
__global__ __launch_bounds__(256) void svt_hip_mem_probe_blocks_kernel(const uint8_t* __restrict__ base, const uint64_t lanes, const uint32_t pitch, const uint32_t bpr, const uint32_t mis,
                                                                       uint32_t* __restrict__ sink) {
    uint32_t acc = 0;
    for (uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x; t < lanes; t += (uint64_t)gridDim.x * 256) {
        const uint64_t b = t >> 8;
        const uint32_t r = (uint32_t)(t >> 2) & 63u, q = (uint32_t)t & 3u;
        const uint8_t* p = base + ((b / bpr) * 64 + r) * (uint64_t)pitch + (b % bpr) * 64 + q * 16 + mis;
        uint32_t       w[4];
        __builtin_memcpy(w, p, 16); // (unaligned 16-byte load, as u32x4_a1 in sad.hip)
        acc ^= w[0] ^ w[1] ^ w[2] ^ w[3];
    }
    if (acc == 0x9e3779b9u) sink[0] = acc;
}

// Delay kernel (test instrument): one lane waits `ticks` of the constant-rate wall clock, so that whatever is queued behind it on its stream starts late.
__global__ void svt_hip_spin_kernel(uint32_t ticks) {
#ifndef SVT_HIP_EMU
    const uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < (uint64_t)ticks) __builtin_amdgcn_s_sleep(16);
#else
    (void)ticks; // (the emulator runs every stream in program order: a delay has nothing to reorder)
#endif
}

extern "C" {

void svt_hip_debug_spin(void* stream, uint32_t microseconds) {
    svthip::ensure_device();
    static std::atomic<int> khz{0};
    int k = khz.load();
    if (k == 0) {
        int v = 0;
#ifndef SVT_HIP_EMU
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeWallClockRate, svthip::physical_device(svthip::current_device())) != hipSuccess || v <= 0) { (void)hipGetLastError(); v = 100000; }
#else
        v = 100000;
#endif
        khz.store(k = v);
    }
    if (microseconds > 100000) microseconds = 100000;
    const uint64_t ticks = (uint64_t)microseconds * (uint64_t)k / 1000;
    hipLaunchKernelGGL(svt_hip_spin_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (uint32_t)ticks);
    SVT_LAUNCH_CHECK();
}

// bytes moved = lanes * width; width 4 / 8 / 16, segment and pitch multiples of width, base aligned to it; the buffer holds ceil(lanes / (seg / width)) * pitch bytes
void svt_hip_mem_probe(int write, int width, void* base, uint64_t lanes, uint32_t seg, uint32_t pitch, uint32_t* sink, void* stream) {
    svthip::ensure_device();
    hipStream_t  st = (hipStream_t)stream;
    uint8_t*     b  = (uint8_t*)base;
    const dim3   grid(256 * 32), blk(256);
#define PROBE(W, WR) hipLaunchKernelGGL(HIP_KERNEL_NAME(svt_hip_mem_probe_kernel<W, WR>), grid, blk, 0, st, b, lanes, seg, pitch, sink)
    if (width == 4) { if (write) PROBE(4, true); else PROBE(4, false); }
    else if (width == 8) { if (write) PROBE(8, true); else PROBE(8, false); }
    else { if (write) PROBE(16, true); else PROBE(16, false); }
#undef PROBE
    SVT_LAUNCH_CHECK();
}

void svt_hip_mem_probe_blocks(const void* base, uint64_t lanes, uint32_t pitch, uint32_t blocks_per_row, uint32_t misalign, uint32_t* sink, void* stream) {
    svthip::ensure_device();
    hipLaunchKernelGGL(svt_hip_mem_probe_blocks_kernel, dim3(256 * 32), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)base, lanes, pitch, blocks_per_row, misalign, sink);
    SVT_LAUNCH_CHECK();
}

void svt_hip_rate_probe(int kind, uint32_t iters, uint32_t blocks, uint32_t* sink, void* stream) {
    svthip::ensure_device();
    hipStream_t st = (hipStream_t)stream;
    switch (kind) {
    case 0: hipLaunchKernelGGL(svt_hip_rate_kernel<0>, dim3(blocks), dim3(256), 0, st, iters, sink); break;
    case 1: hipLaunchKernelGGL(svt_hip_rate_kernel<1>, dim3(blocks), dim3(256), 0, st, iters, sink); break;
    case 2: hipLaunchKernelGGL(svt_hip_rate_kernel<2>, dim3(blocks), dim3(256), 0, st, iters, sink); break;
    case 3: hipLaunchKernelGGL(svt_hip_rate_kernel<3>, dim3(blocks), dim3(256), 0, st, iters, sink); break;
    case 4: hipLaunchKernelGGL(svt_hip_rate_kernel<4>, dim3(blocks), dim3(256), 0, st, iters, sink); break;
    case 5: hipLaunchKernelGGL(svt_hip_rate_kernel<5>, dim3(blocks), dim3(256), 0, st, iters, sink); break;
    case 6: hipLaunchKernelGGL(svt_hip_rate_kernel<6>, dim3(blocks), dim3(256), 0, st, iters, sink); break;
    case 7: hipLaunchKernelGGL(svt_hip_rate_kernel<7>, dim3(blocks), dim3(256), 0, st, iters, sink); break;
    default: hipLaunchKernelGGL(svt_hip_rate_kernel<8>, dim3(blocks), dim3(256), 0, st, iters, sink); break;
    }
    SVT_LAUNCH_CHECK();
}

int svt_hip_init(int device) {
    using namespace svthip;
    const int n = svt_hip_device_count(); // logical devices
    if (n <= 0) return -1;
    if (device < 0 || device >= n || device >= svthip::MAX_DEVICES) return -1;
    if (hipSetDevice(physical_device(device)) != hipSuccess) return -1;
    // SVT_HIP_SYNC=block: host threads SLEEP while they wait for the device (hipDeviceScheduleBlockingSync) instead of spinning on the completion signal, the
    // runtime's default.  A stage call of an encoder seam is one synchronous round trip; on a host whose cores are all busy with the encoder's own threads the
    // spinning costs CPU time another worker could use (environment table; profiles/r04_*: host CPU seconds per frame).
    const char* sync_env = getenv("SVT_HIP_SYNC");
    if (sync_env && sync_env[0] == 'b') {
        if (hipSetDeviceFlags(hipDeviceScheduleBlockingSync) != hipSuccess) (void)hipGetLastError(); // (refused once the context is active: keep the default)
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, physical_device(device)) != hipSuccess) return -1;
    snprintf(g_name, sizeof(g_name), "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    g_device      = device;
    g_initialised = true;
    return 0;
}

void svt_hip_shutdown(void) {
    using namespace svthip;
    for (int d = 0; d < MAX_DEVICES; d++) { // the calling thread's arenas
        HostCall& c = t_calls[d];
        if (!c.dev && !c.pin && !c.stream) continue;
        (void)hipSetDevice(physical_device(d));
        if (c.dev) (void)hipFree(c.dev);
        if (c.pin) (void)hipHostFree(c.pin);
        if (c.stream) (void)hipStreamDestroy(c.stream);
        c = HostCall();
    }
    {   // the leased stage arenas (the largest ones: 20-50 MB pinned + device each) of every device
        std::lock_guard<std::mutex> g(g_lease_m);
        for (int d = 0; d < MAX_DEVICES; d++) {
            if (g_lease_pool[d].empty()) continue;
            (void)hipSetDevice(physical_device(d));
            for (HostCall* c : g_lease_pool[d]) {
                if (c->dev) (void)hipFree(c->dev);
                if (c->pin) (void)hipHostFree(c->pin);
                if (c->stream) (void)hipStreamDestroy(c->stream);
                delete c;
            }
            g_lease_pool[d].clear();
        }
    }
    stream_pool_free_all(); // the pooled side-stream sets (3 streams + 6 events each)
    plane_cache_free_all();
    partition_pool_free();
    t_bound       = -1;
    g_initialised = false;
    g_failed      = false; // (a later svt_hip_init starts clean; the dispatch pointers stay restored until svt_hip_setup_rtcd is called again)
}

int svt_hip_device_count(void) { // LOGICAL devices: GPUs x SVT_HIP_VIRTUAL_DEVICES (see physical_device above), at most MAX_DEVICES
    const int n = svthip::physical_count() * svthip::virtual_per_gpu();
    return n > svthip::MAX_DEVICES ? svthip::MAX_DEVICES : n;
}
int svt_hip_physical_device_count(void) { return svthip::physical_count(); }
int svt_hip_physical_device(int device) { return device < 0 || device >= svt_hip_device_count() ? -1 : svthip::physical_device(device); }
int svt_hip_set_virtual_devices(int per_gpu) { // before the devices beyond the first are used; a later change only changes how many logical numbers are valid
    if (per_gpu < 1 || per_gpu > svthip::MAX_DEVICES) return -1;
    svthip::g_virtual.store(per_gpu, std::memory_order_release);
    return 0;
}
int svt_hip_set_thread_device(int device) {
    SVT_HIP_ENTRY_TRY
    using namespace svthip;
    if (device >= MAX_DEVICES || device >= svt_hip_device_count()) return -1;
    t_want = device < 0 ? -1 : device;
    if (device >= 0 && device != (int)g_device) g_multi = true;
    ensure_device();
    return 0;
    SVT_HIP_ENTRY_CATCH(SVT_HIP_E_DEVICE)
}
int svt_hip_get_thread_device(void) {
    SVT_HIP_ENTRY_TRY
    svthip::ensure_device();
    return svthip::current_device();
    SVT_HIP_ENTRY_CATCH(SVT_HIP_E_DEVICE)
}

const char* svt_hip_device_name(void) { return svthip::g_name; }
const char* svt_hip_last_error(void) { return svthip::failed() ? svthip::g_last_error : nullptr; }
int         svt_hip_failed(void) { return svthip::failed() ? 1 : 0; }
// test instrument: the next HIP_CHECK-level failure is simulated now (as if a hipMalloc had failed inside the calling entry point)
int svt_hip_debug_inject_failure(void) {
    try { svthip::device_fail((int)hipErrorOutOfMemory, "injected failure (svt_hip_debug_inject_failure)", __FILE__, __LINE__); } catch (const svthip::DeviceError&) {}
    return 0;
}
void svt_hip_tuning_reload(void) { svthip::tuning_read(); }

// ---- HIP graphs: every batched entry point only enqueues work on the stream it is given (no host synchronisation, no host-side state), so a
// whole per-picture sequence (padding + decimations, the transform chain, the in-loop filter chain) can be captured once and replayed.
void* svt_hip_stream_create(void) {
    SVT_HIP_ENTRY_TRY
    svthip::ensure_device();
    hipStream_t st;
    HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    return (void*)st;
    SVT_HIP_ENTRY_CATCH(nullptr)
}
void svt_hip_stream_destroy(void* stream) {
    SVT_HIP_ENTRY_TRY HIP_CHECK(hipStreamDestroy((hipStream_t)stream));     SVT_HIP_ENTRY_CATCH((void)0)
}
void svt_hip_stream_synchronize(void* stream) {
    SVT_HIP_ENTRY_TRY HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));     SVT_HIP_ENTRY_CATCH((void)0)
}
// (no exception leaves an extern "C" entry point: a failed graph call records the error -- svt_hip_last_error -- and returns; capture_end returns NULL)
void svt_hip_graph_capture_begin(void* stream) {
    SVT_HIP_ENTRY_TRY HIP_CHECK(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));     SVT_HIP_ENTRY_CATCH((void)0)
}
void* svt_hip_graph_capture_end(void* stream) {
    SVT_HIP_ENTRY_TRY
    hipGraph_t     graph = nullptr;
    hipGraphExec_t exec  = nullptr;
    HIP_CHECK(hipStreamEndCapture((hipStream_t)stream, &graph));
    HIP_CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    HIP_CHECK(hipGraphDestroy(graph));
    return (void*)exec;
    SVT_HIP_ENTRY_CATCH(nullptr)
}
void svt_hip_graph_launch(void* graph_exec, void* stream) {
    SVT_HIP_ENTRY_TRY HIP_CHECK(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream));     SVT_HIP_ENTRY_CATCH((void)0)
}
void svt_hip_graph_destroy(void* graph_exec) {
    SVT_HIP_ENTRY_TRY HIP_CHECK(hipGraphExecDestroy((hipGraphExec_t)graph_exec));     SVT_HIP_ENTRY_CATCH((void)0)
}

// Pays the one-time costs of the calling thread's device up front (an encoder calls it while it initialises): the HIP context, the loading of this library's code
// objects (the runtime loads them at the first launch), the first pinned and device allocations.  ~60 ms that would otherwise sit inside the first picture's stage.
static void warmup_impl(int stage_arenas, size_t first_arena_bytes);
void svt_hip_warmup(void) { warmup_impl(3, 192u << 20); }
// the same with the pool pre-reservation sized by the caller: `stage_arenas` leased arenas (0-3: as many stage-sized host forms as will be in flight at once; 0 for an
// encoder that only installs the dispatch pointers), the first one with `first_arena_mb` MB of device memory (the loop-restoration search of a 1080p plane wants ~100 MB,
// a 4K one ~400 MB; the other stages 96 MB)
void svt_hip_warmup_sized(int stage_arenas, uint32_t first_arena_mb) {
    warmup_impl(stage_arenas < 0 ? 0 : (stage_arenas > 3 ? 3 : stage_arenas), (size_t)(first_arena_mb < 16 ? 16 : first_arena_mb) << 20);
}
static void warmup_impl(int stage_arenas, size_t first_arena_bytes) {
    SVT_HIP_ENTRY_TRY
    svthip::ensure_device();
    svthip::HostCall& c = svthip::host_call();
    c.begin();
    c.reserve(64u << 20, 16u << 20);
    uint32_t* d = (uint32_t*)c.dalloc(17 * 64 * 4);
    hipLaunchKernelGGL(svt_hip_selftest_kernel, dim3(1), dim3(64), 0, c.stream, d);
    SVT_LAUNCH_CHECK();
    svthip::warm_cdef(c.stream);
    svthip::warm_cdef_pick(c.stream);
    svthip::warm_deblock(c.stream);
    svthip::warm_hme(c.stream);
    svthip::warm_lr_search(c.stream);
    svthip::warm_lr_stats(c.stream);
    svthip::warm_me_results(c.stream);
    svthip::warm_me_session(c.stream);
    svthip::warm_misc(c.stream);
    svthip::warm_picprep(c.stream);
    svthip::warm_pme(c.stream);
    svthip::warm_quant(c.stream);
    svthip::warm_restoration(c.stream);
    svthip::warm_sad(c.stream);
    svthip::warm_tf(c.stream);
    svthip::warm_tf_picture(c.stream);
    svthip::warm_tf_subpel(c.stream);
    svthip::warm_tpl(c.stream);
    svthip::warm_tpl_full(c.stream);
    svthip::warm_txfm(c.stream);
    svthip::warm_txfm_fused(c.stream);
    SVT_LAUNCH_CHECK();
    c.sync();
    {   // ... and what the stage-sized host forms take from pools on their first calls: three leased arenas (device + pinned: the pinned allocation is the slow part) and one
        // set of side streams, made now instead of inside the first pictures' stage calls
        // (all leases are held at once so that they are distinct arenas; they return to the pool at the end of the block)
        if (stage_arenas >= 1) {
            svthip::HostCallLease a;
            (*a).begin(); (*a).reserve(first_arena_bytes, 24u << 20);
            if (stage_arenas >= 2) {
                svthip::HostCallLease b;
                (*b).begin(); (*b).reserve(96u << 20, 24u << 20);
                if (stage_arenas >= 3) {
                    svthip::HostCallLease c3;
                    (*c3).begin(); (*c3).reserve(96u << 20, 24u << 20);
                }
            }
            svthip::StreamSetLease s1;
        }
    }
    SVT_HIP_ENTRY_CATCH((void)0)
}

int svt_hip_host_register(void* buffer, size_t bytes) {
    SVT_HIP_ENTRY_TRY
    svthip::ensure_device();
    if (!buffer || !bytes) return -1;
    const hipError_t e = hipHostRegister(buffer, bytes, hipHostRegisterPortable);
    if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; } // (already registered, or not lockable: the caller keeps using it as pageable memory)
    {
        std::lock_guard<std::mutex> g(svthip::g_locked_m);
        svthip::g_locked.push_back({(uintptr_t)buffer, (uintptr_t)buffer + bytes});
        svthip::g_locked_n.store((int)svthip::g_locked.size(), std::memory_order_release);
    }
    return 0;
    SVT_HIP_ENTRY_CATCH(SVT_HIP_E_DEVICE)
}
int svt_hip_host_unregister(void* buffer) {
    SVT_HIP_ENTRY_TRY
    svthip::ensure_device();
    {
        std::lock_guard<std::mutex> g(svthip::g_locked_m);
        for (size_t i = 0; i < svthip::g_locked.size(); i++)
            if (svthip::g_locked[i].lo == (uintptr_t)buffer) { svthip::g_locked.erase(svthip::g_locked.begin() + i); break; }
        svthip::g_locked_n.store((int)svthip::g_locked.size(), std::memory_order_release);
    }
    const hipError_t e = hipHostUnregister(buffer);
    if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }
    return 0;
    SVT_HIP_ENTRY_CATCH(SVT_HIP_E_DEVICE)
}

int svt_hip_selftest(uint32_t* results, void* stream) {
    SVT_HIP_ENTRY_TRY
    svthip::ensure_device();
    hipLaunchKernelGGL(svt_hip_selftest_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, results);
    SVT_LAUNCH_CHECK();
    return 0;
    SVT_HIP_ENTRY_CATCH(SVT_HIP_E_DEVICE)
}

} // extern "C"
