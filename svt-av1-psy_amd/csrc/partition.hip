// partition.hip -- ONE picture over several MI355X from a C host (the frame-partition case; BASELINE north_star: "RCCL over xGMI only for the
// frame-partition case").
//
// A partition is a list of devices; devices[0] is the HOME device: the caller's buffers live there and the caller's stream belongs to it.  Every entry point below
// is the batched primitive of the single-device ABI with the picture cut into contiguous strips (SB rows for ME, filter-block rows for CDEF, 64-row stripes for the
// loop restoration -- the units the primitives already take as ranges), strip k on devices[k]:
//
//     home stream --record--> [ready]                      (the inputs are final)
//     device k:   wait [ready]; inputs home -> k (hipMemcpyPeerAsync: over xGMI between the GPUs of one node); the strip's launch on k; the strip's OUTPUT ROWS
//                 k -> home, into the caller's arrays where the single-device call would have written them; --record--> [done k]
//     home:       its own strip on the caller's stream; wait [done k] for every k.  The call returns with everything ENQUEUED, like the primitives it wraps.
//
// "Broadcast in, all-gather out" of SURVEY 8(e) with point-to-point copies: the exchange is one-to-all and all-to-one with the home device as root (the reference
// encoder consumes the result on the host, behind the home device), so a ring collective has nothing to add -- each remote device moves the planes once in and its
// strip once out over its own xGMI link to the home device, all links in parallel.  No RCCL dependency inside the library: the peer copies are plain HIP, which is also
// what lets the CPU emulator run the path (tests/emu: a peer copy is legal when each pointer belongs to the device it is said to live on).
// The per-device mirrors of the inputs live in an arena per (partition, device) that only grows; a partition is used by one host thread at a time.
#include <vector>

#include "svt_hip_common.h"
#include "../../include/svtav1_hip.h"

namespace {

struct Peer {
    int         device = 0; // logical (svt_hip_common.h: physical_device())
    int         phys   = 0; // the GPU ordinal the peer copies name
    hipStream_t stream = nullptr;
    hipEvent_t  done   = nullptr;
    uint8_t*    arena  = nullptr;
    size_t      cap = 0, used = 0;
};
struct Partition {
    int               n = 0;
    std::vector<Peer> peer; // peer[0] = the home device (no stream / arena of its own: the caller's)
    hipEvent_t        ready = nullptr;
    uint64_t          bytes_in = 0, bytes_out = 0, calls = 0; // peer traffic so far (svt_hip_frame_partition_stats)
    uint32_t          jitter_us = 0, jitter_state = 0;        // test instrument (svt_hip_frame_partition_set_jitter): random delays between the steps of the protocol
};
// With jitter on, a delay kernel of pseudo-random length (0 .. jitter_us, a third of the draws none) goes into `st` here: the steps of the ready / done protocol on the
// home and peer streams then finish in a different interleaving on every call, so a missing wait shows up as a wrong strip instead of passing by luck.
void jitter(Partition& P, hipStream_t st) {
    if (!P.jitter_us) return;
    uint32_t x = P.jitter_state;
    x ^= x << 13; x ^= x >> 17; x ^= x << 5;
    P.jitter_state = x;
    if (x % 3 == 0) return;
    svt_hip_debug_spin(st, (x >> 8) % (P.jitter_us + 1));
}

// contiguous strips, the first (total % n) one unit longer (the rule of bench.py --mode strips and of the gloo tests)
inline size_t plane_bytes(size_t stride, size_t width, size_t rows, size_t px) { return rows ? ((rows - 1) * stride + width) * px : 0; } // (the last row ends at `width`)
inline void strip_of(int total, int k, int n, int& b, int& e) {
    const int q = total / n, r = total % n;
    b = k * q + (k < r ? k : r);
    e = b + q + (k < r ? 1 : 0);
}
void arena_begin(Peer& p, size_t need) {
    need += 4096;
    if (need > p.cap) { // (grows between calls only: the previous call's work on this device has been waited for by the home stream, and is drained here)
        svthip::DeviceGuard g(p.device);
        HIP_CHECK(hipStreamSynchronize(p.stream));
        if (p.arena) HIP_CHECK(hipFree(p.arena));
        p.cap = svthip::align_up(need + need / 4, 1 << 20);
        HIP_CHECK(hipMalloc((void**)&p.arena, p.cap));
    }
    p.used = 0;
}
void* arena_take(Peer& p, size_t bytes) {
    const size_t off = svthip::align_up(p.used, 256);
    if (off + bytes > p.cap) svthip::device_fail(-3, "frame-partition arena overflow", __FILE__, __LINE__);
    p.used = off + bytes;
    return p.arena + off;
}
// input `src` (home device) -> a mirror on peer p, in p's stream
void* mirror_in(Partition& P, Peer& p, const void* src, size_t bytes) {
    if (!src || !bytes) return nullptr;
    void* d = arena_take(p, bytes);
    HIP_CHECK(hipMemcpyPeerAsync(d, p.phys, src, P.peer[0].phys, bytes, p.stream));
    P.bytes_in += bytes;
    return d;
}
// rows [r0, r1) of a plane (row pitch `row_bytes`, the last row ending after `width_bytes`) -> a mirror of just those rows on peer p; returns the VIRTUAL plane base:
// the address row 0 would have, so that the strip kernels -- which address the full plane -- find row y of the band at base + y * row_bytes.  Only rows of the band
// may be touched through it (the strip's own rows + the halo the caller adds).
uint8_t* mirror_rows(Partition& P, Peer& p, const void* src, size_t row_bytes, size_t width_bytes, int r0, int r1) {
    if (!src || r1 <= r0) return nullptr;
    const size_t bytes = (size_t)(r1 - r0 - 1) * row_bytes + width_bytes;
    uint8_t* d = (uint8_t*)arena_take(p, bytes);
    HIP_CHECK(hipMemcpyPeerAsync(d, p.phys, (const uint8_t*)src + (size_t)r0 * row_bytes, P.peer[0].phys, bytes, p.stream));
    P.bytes_in += bytes;
    return d - (size_t)r0 * row_bytes;
}
void rows_out(Partition& P, Peer& p, void* dst_home, const void* src_peer, size_t bytes) {
    if (!bytes) return;
    HIP_CHECK(hipMemcpyPeerAsync(dst_home, P.peer[0].phys, src_peer, p.phys, bytes, p.stream));
    P.bytes_out += bytes;
}
void open_call(Partition& P, hipStream_t home) {
    jitter(P, home); // (delays `ready`: the inputs become final late)
    HIP_CHECK(hipEventRecord(P.ready, home));
    P.calls++;
}
void close_call(Partition& P, hipStream_t home) {
    for (int k = 1; k < P.n; k++) HIP_CHECK(hipStreamWaitEvent(home, P.peer[k].done, 0));
    // the peers' guards left the LAST peer current on this host thread: a caller that relies on HIP's current device (a torch process, an encoder thread between two
    // library calls) finds its home device current again
    if (P.n > 1) HIP_CHECK(hipSetDevice(P.peer[0].phys));
}

} // namespace

// ---- the host forms' switch (svt_hip_set_frame_partition): with a device list set, the picture-sized HOST forms of the in-loop filters (svt_hip_cdef_apply_host,
// svt_hip_cdef_search_host, svt_hip_lr_filter_frame_host -- what the encoder's CDEF / REST seams call) run their frame launches through a partition instead of on
// the calling thread's device alone.  A partition serves one frame launch at a time and comes from a pool (below); its home is the thread's current device, and a
// thread whose device is not the list's first entry keeps the single-device path.
#include <atomic>
#include <mutex>
namespace svthip {
static std::mutex       g_strips_m;
static int              g_strips_n = 0, g_strips_dev[MAX_DEVICES];
static std::atomic<int> g_strips_on{0};
static std::atomic<unsigned long long> g_strips_calls{0};
// Partitions of the host forms are POOLED per device list (generation), not kept per worker thread: an encoder calls the filter stages from whichever thread holds the
// picture, and a partition per thread multiplied the peers' streams and arenas by the number of seam threads and leaked them on a change of the list.  A
// partition is taken for one frame launch and handed back with its work possibly in flight: the peers' streams are in-order, an arena that has to grow drains its
// stream first, and `ready` / `done k` are recorded again before they are waited for.
static std::atomic<int>   g_strips_gen{0};
static std::vector<void*> g_part_pool; // (under g_strips_m) partitions of generation g_part_gen
static int                g_part_gen = -1;
struct PartitionLease {
    void* part = nullptr;
    int   gen  = -1;
    PartitionLease() {
        if (!g_strips_on.load(std::memory_order_acquire)) return;
        int devs[MAX_DEVICES], n;
        {
            std::lock_guard<std::mutex> g(g_strips_m);
            gen = g_strips_gen.load();
            if (g_part_gen != gen) { // the list changed: the old generation's partitions go
                for (void* p : g_part_pool) svt_hip_frame_partition_destroy(p);
                g_part_pool.clear();
                g_part_gen = gen;
            }
            n = g_strips_n;
            for (int i = 0; i < n; i++) devs[i] = g_strips_dev[i];
            if (n > 1 && devs[0] == current_device() && !g_part_pool.empty()) { part = g_part_pool.back(); g_part_pool.pop_back(); }
        }
        if (!part && n > 1 && devs[0] == current_device()) part = svt_hip_frame_partition_create(devs, n); // (a thread whose device is not the home keeps the single-device path)
        if (part) g_strips_calls.fetch_add(1);
    }
    ~PartitionLease() {
        if (!part) return;
        std::lock_guard<std::mutex> g(g_strips_m);
        if (g_part_gen == gen) g_part_pool.push_back(part);
        else svt_hip_frame_partition_destroy(part);
    }
};
void partition_pool_free() { // svt_hip_shutdown
    std::lock_guard<std::mutex> g(g_strips_m);
    for (void* p : g_part_pool) svt_hip_frame_partition_destroy(p);
    g_part_pool.clear();
}
void cdef_frame_dispatch(int mode, const SvtHipCdefParams* P, hipStream_t st) {
    PartitionLease L;
    if (L.part) svt_hip_frame_partition_cdef(L.part, mode, P, st);
    else svt_hip_cdef_frame(mode, P, st);
}
void lr_frame_dispatch(const SvtHipLrParams* P, hipStream_t st) {
    PartitionLease L;
    if (L.part) svt_hip_frame_partition_lr(L.part, P, st);
    else svt_hip_lr_filter_frame(P, st);
}
} // namespace svthip

extern "C" {

int svt_hip_set_frame_partition(const int* devices, int n) {
    SVT_HIP_ENTRY_TRY
    using namespace svthip;
    if (n < 0 || n > MAX_DEVICES || (n > 0 && !devices)) return -1;
    const int have = svt_hip_device_count();
    for (int k = 0; k < n; k++)
        if (devices[k] < 0 || devices[k] >= have) return -1;
    std::lock_guard<std::mutex> g(g_strips_m);
    g_strips_n = n;
    for (int k = 0; k < n; k++) g_strips_dev[k] = devices[k];
    g_strips_gen.fetch_add(1);
    g_strips_on.store(n > 1, std::memory_order_release);
    return 0;
    SVT_HIP_ENTRY_CATCH(SVT_HIP_E_DEVICE)
}
unsigned long long svt_hip_frame_partition_host_calls(void) { return svthip::g_strips_calls.load(); }

void* svt_hip_frame_partition_create(const int* devices, int n) {
    SVT_HIP_ENTRY_TRY
    if (!devices || n < 1 || n > svthip::MAX_DEVICES) return nullptr;
    const int have = svt_hip_device_count();
    for (int k = 0; k < n; k++) {
        if (devices[k] < 0 || devices[k] >= have) return nullptr;
        for (int j = 0; j < k; j++)
            if (devices[j] == devices[k]) return nullptr;
    }
    Partition* P = new Partition;
    P->n = n;
    P->peer.resize(n);
    for (int k = 0; k < n; k++) {
        P->peer[k].device = devices[k];
        P->peer[k].phys   = svthip::physical_device(devices[k]);
        svthip::DeviceGuard g(devices[k]);
        if (k == 0) {
            HIP_CHECK(hipEventCreateWithFlags(&P->ready, hipEventDisableTiming));
            continue;
        }
        HIP_CHECK(hipStreamCreateWithFlags(&P->peer[k].stream, hipStreamNonBlocking));
        HIP_CHECK(hipEventCreateWithFlags(&P->peer[k].done, hipEventDisableTiming));
        int can = 0; // direct xGMI access where the platform offers it (the copies work without it, staged by the runtime)
        const int pk = P->peer[k].phys, p0 = svthip::physical_device(devices[0]);
        if (pk == p0) continue; // (two logical devices of one GPU: SVT_HIP_VIRTUAL_DEVICES)
        if (hipDeviceCanAccessPeer(&can, pk, p0) == hipSuccess && can) {
            if (hipDeviceEnablePeerAccess(p0, 0) != hipSuccess) (void)hipGetLastError(); // (already enabled: fine)
        } else (void)hipGetLastError();
    }
    return P;
    SVT_HIP_ENTRY_CATCH(nullptr)
}
void svt_hip_frame_partition_destroy(void* part) {
    SVT_HIP_ENTRY_TRY
    Partition* P = (Partition*)part;
    if (!P) return;
    for (int k = 0; k < P->n; k++) {
        svthip::DeviceGuard g(P->peer[k].device);
        if (k == 0) { (void)hipEventDestroy(P->ready); continue; }
        (void)hipStreamSynchronize(P->peer[k].stream);
        if (P->peer[k].arena) (void)hipFree(P->peer[k].arena);
        (void)hipEventDestroy(P->peer[k].done);
        (void)hipStreamDestroy(P->peer[k].stream);
    }
    delete P;
    SVT_HIP_ENTRY_CATCH((void)0)
}
void svt_hip_frame_partition_set_jitter(void* part, uint32_t seed, uint32_t max_us) {
    Partition* P = (Partition*)part;
    if (!P) return;
    P->jitter_us    = max_us;
    P->jitter_state = seed ? seed : 0x9e3779b9u;
}
int svt_hip_frame_partition_size(const void* part) { return part ? ((const Partition*)part)->n : 0; }
void svt_hip_frame_partition_stats(const void* part, uint64_t* calls, uint64_t* bytes_in, uint64_t* bytes_out) {
    const Partition* P = (const Partition*)part;
    if (calls) *calls = P ? P->calls : 0;
    if (bytes_in) *bytes_in = P ? P->bytes_in : 0;
    if (bytes_out) *bytes_out = P ? P->bytes_out : 0;
}

// ---- open-loop ME: items (SB x reference descriptors) [0, n) in contiguous strips; the caller orders them SB row after SB row, so a strip is a band of SB rows --------
int svt_hip_frame_partition_me(void* part, const uint8_t* src_base, size_t src_bytes, const uint8_t* ref_base, size_t ref_bytes, const SvtHipMeSearchDesc* descs,
                               uint32_t n, uint32_t max_w, uint32_t max_h, int sub_sad, uint32_t* best_sad, uint32_t* best_mv, void* workspace, void* stream) {
    SVT_HIP_ENTRY_TRY
    Partition* P = (Partition*)part;
    if (!P) return -1;
    hipStream_t home = (hipStream_t)stream;
    svthip::DeviceGuard gh(P->peer[0].device);
    open_call(*P, home);
    const size_t row = (size_t)SVT_HIP_ME_NUM_BLOCKS * 4;
    for (int k = 1; k < P->n; k++) {
        int b, e;
        strip_of((int)n, k, P->n, b, e);
        Peer& p = P->peer[k];
        if (e <= b) { svthip::DeviceGuard g(p.device); HIP_CHECK(hipEventRecord(p.done, p.stream)); continue; }
        const uint32_t cnt = (uint32_t)(e - b);
        const size_t   ws  = svt_hip_me_fullpel_search_workspace(cnt, max_w, max_h);
        const bool     one = src_base == ref_base;
        arena_begin(p, src_bytes + (one ? 0 : ref_bytes) + (size_t)cnt * (sizeof(SvtHipMeSearchDesc) + 2 * row) + ws + 8 * 256);
        svthip::DeviceGuard g(p.device);
        HIP_CHECK(hipStreamWaitEvent(p.stream, P->ready, 0));
        jitter(*P, p.stream);
        const uint8_t* m_src = (const uint8_t*)mirror_in(*P, p, src_base, src_bytes);
        const uint8_t* m_ref = one ? m_src : (const uint8_t*)mirror_in(*P, p, ref_base, ref_bytes);
        const SvtHipMeSearchDesc* m_d = (const SvtHipMeSearchDesc*)mirror_in(*P, p, descs + b, (size_t)cnt * sizeof(SvtHipMeSearchDesc));
        uint32_t* o_sad = (uint32_t*)arena_take(p, cnt * row);
        uint32_t* o_mv  = (uint32_t*)arena_take(p, cnt * row);
        void*     m_ws  = ws ? arena_take(p, ws) : nullptr;
        jitter(*P, p.stream);
        svt_hip_me_fullpel_search_batch(m_src, m_ref, m_d, cnt, max_w, max_h, sub_sad, o_sad, o_mv, m_ws, p.stream);
        jitter(*P, p.stream);
        rows_out(*P, p, (uint8_t*)best_sad + (size_t)b * row, o_sad, cnt * row);
        rows_out(*P, p, (uint8_t*)best_mv + (size_t)b * row, o_mv, cnt * row);
        HIP_CHECK(hipEventRecord(p.done, p.stream));
    }
    int b0, e0;
    strip_of((int)n, 0, P->n, b0, e0);
    jitter(*P, home);
    if (e0 > b0)
        svt_hip_me_fullpel_search_batch(src_base, ref_base, descs + b0, (uint32_t)(e0 - b0), max_w, max_h, sub_sad, best_sad + (size_t)b0 * SVT_HIP_ME_NUM_BLOCKS,
                                        best_mv + (size_t)b0 * SVT_HIP_ME_NUM_BLOCKS, workspace, home);
    close_call(*P, home);
    return 0;
    SVT_HIP_ENTRY_CATCH(SVT_HIP_E_DEVICE)
}

// ---- CDEF (search, apply, apply with the search's directions): filter-block rows in strips ------------------------------------------------------------------------------
int svt_hip_frame_partition_cdef(void* part, int mode, const SvtHipCdefParams* params, void* stream) {
    SVT_HIP_ENTRY_TRY
    Partition* P = (Partition*)part;
    if (!P || !params) return -1;
    const SvtHipCdefParams& C = *params;
    if (mode == 1 && C.ncand == 0) return 0;
    hipStream_t home = (hipStream_t)stream;
    svthip::DeviceGuard gh(P->peer[0].device);
    open_call(*P, home);
    const int    bw = 64 >> C.xdec, bh = 64 >> C.ydec, px = C.is_16bit ? 2 : 1;
    const int    nhfb = ((int)C.width + bw - 1) / bw, nvfb = ((int)C.height + bh - 1) / bh, nfb = nhfb * nvfb;
    const size_t recon_b = plane_bytes(C.recon_stride, C.width, C.height, px), source_b = mode == 1 ? plane_bytes(C.source_stride, C.width, C.height, px) : 0;
    const size_t out_b = mode == 1 ? 0 : plane_bytes(C.out_stride, C.width, C.height, px), skip_b = (size_t)nvfb * 8 * nhfb * 8;
    const size_t str_b = (size_t)(mode == 1 ? C.ncand : (uint32_t)nfb) * 4, dir_b = (size_t)nfb * 64, var_b = (size_t)nfb * 64 * 4, mse_b = mode == 1 ? (size_t)nfb * C.ncand * 8 : 0;
    const bool   dir_in = C.pli != 0 || mode == 2; // chroma planes and the apply behind a search read the luma directions / variances
    for (int k = 1; k < P->n; k++) {
        int b, e;
        strip_of(nvfb, k, P->n, b, e);
        Peer& p = P->peer[k];
        svthip::DeviceGuard g(p.device);
        if (e <= b) { HIP_CHECK(hipEventRecord(p.done, p.stream)); continue; }
        // the strip's rows of every plane-sized input plus the halo the tiles read (VB = 3 rows of the neighbouring filter blocks; 8 taken), the strip's rows of the
        // 8x8 skip map and of the per-filter-block arrays -- not the whole picture per peer: peer traffic is ~ one picture per call, whatever n is
        const int    y0 = b * bh, y1 = e * bh < (int)C.height ? e * bh : (int)C.height;
        const int    h0 = y0 - 8 > 0 ? y0 - 8 : 0, h1 = y1 + 8 < (int)C.height ? y1 + 8 : (int)C.height;
        const size_t rrow = (size_t)C.recon_stride * px, srow = (size_t)C.source_stride * px, orow = (size_t)C.out_stride * px, wb = (size_t)C.width * px;
        const size_t f0 = (size_t)b * nhfb, fn = (size_t)(e - b) * nhfb;
        arena_begin(p, (size_t)(h1 - h0) * rrow + (mode == 1 ? (size_t)(y1 - y0) * srow : (size_t)(y1 - y0) * orow) + (size_t)(e - b) * 8 * nhfb * 8 + 2 * str_b + fn * 64 * 5 +
                           (mode == 1 ? fn * C.ncand * 8 : 0) + 16 * 256);
        HIP_CHECK(hipStreamWaitEvent(p.stream, P->ready, 0));
        jitter(*P, p.stream);
        SvtHipCdefParams M = C;
        M.recon  = mirror_rows(*P, p, C.recon, rrow, wb, h0, h1);
        M.source = mode == 1 ? mirror_rows(*P, p, C.source, srow, wb, y0, y1) : nullptr;
        M.skip   = mirror_rows(*P, p, C.skip, (size_t)nhfb * 8, (size_t)nhfb * 8, b * 8, e * 8);
        M.pri    = (const int32_t*)mirror_in(*P, p, C.pri, str_b);
        M.sec    = (const int32_t*)mirror_in(*P, p, C.sec, str_b);
        // (always mirrored, also where they are outputs: a filter block without a single unit to filter leaves its entries as the caller had them)
        M.dir    = mirror_rows(*P, p, C.dir, (size_t)nhfb * 64, (size_t)nhfb * 64, b, e);
        M.var    = (int32_t*)mirror_rows(*P, p, C.var, (size_t)nhfb * 64 * 4, (size_t)nhfb * 64 * 4, b, e);
        M.mse    = mode == 1 ? (uint64_t*)((uint8_t*)arena_take(p, fn * C.ncand * 8) - f0 * C.ncand * 8) : nullptr;
        // apply writes out of place onto a pre-copied plane (skipped units are not touched): the strip's rows of the caller's `out` are that pre-copy
        M.out = mode == 1 ? nullptr : (void*)mirror_rows(*P, p, C.out, orow, wb, y0, y1);
        const size_t strip_b = plane_bytes(C.out_stride, C.width, (size_t)(y1 - y0), px);
        jitter(*P, p.stream);
        svt_hip_cdef_frame_rows(mode, &M, b, e, p.stream);
        jitter(*P, p.stream);
        if (mode == 1) rows_out(*P, p, C.mse + f0 * C.ncand, M.mse + f0 * C.ncand, fn * C.ncand * 8);
        else rows_out(*P, p, (uint8_t*)C.out + y0 * orow, (const uint8_t*)M.out + y0 * orow, strip_b);
        if (!dir_in) { // luma: the strip's directions / variances are results too
            rows_out(*P, p, C.dir + f0 * 64, M.dir + f0 * 64, fn * 64);
            rows_out(*P, p, C.var + f0 * 64, M.var + f0 * 64, fn * 64 * 4);
        }
        HIP_CHECK(hipEventRecord(p.done, p.stream));
    }
    int b0, e0;
    strip_of(nvfb, 0, P->n, b0, e0);
    jitter(*P, home);
    if (e0 > b0) svt_hip_cdef_frame_rows(mode, params, b0, e0, home);
    close_call(*P, home);
    return 0;
    SVT_HIP_ENTRY_CATCH(SVT_HIP_E_DEVICE)
}

// ---- loop restoration filter: 64-row stripes (offset by 8 rows, restoration.c:1082-1098) in strips ------------------------------------------------------------------
int svt_hip_frame_partition_lr(void* part, const SvtHipLrParams* params, void* stream) {
    SVT_HIP_ENTRY_TRY
    Partition* P = (Partition*)part;
    if (!P || !params) return -1;
    const SvtHipLrParams& L = *params;
    hipStream_t home = (hipStream_t)stream;
    svthip::DeviceGuard gh(P->peer[0].device);
    open_call(*P, home);
    const int    px = L.highbd ? 2 : 1, sh = 64 >> L.ss_y, off = 8 >> L.ss_y;
    const int    nstripes = ((int)L.height + off + sh - 1) / sh;
    const int    us = (int)L.unit_size, nvu = ((int)L.height + us / 2) / us > 0 ? ((int)L.height + us / 2) / us : 1, nhu = ((int)L.width + us / 2) / us > 0 ? ((int)L.width + us / 2) / us : 1;
    const size_t data_b = plane_bytes(L.stride, L.width, L.height, px), bnd_b = plane_bytes(L.boundary_stride, L.width, (size_t)2 * nstripes, px);
    const size_t dst_b = plane_bytes(L.dst_stride, L.width, L.height, px);
    const size_t unit_b = (size_t)nvu * nhu * sizeof(SvtHipLrUnit);
    for (int k = 1; k < P->n; k++) {
        int b, e;
        strip_of(nstripes, k, P->n, b, e);
        Peer& p = P->peer[k];
        svthip::DeviceGuard g(p.device);
        if (e <= b) { HIP_CHECK(hipEventRecord(p.done, p.stream)); continue; }
        // stripe s covers picture rows [s * sh - off, (s + 1) * sh - off) clipped to the plane; the filters read 3 rows beyond a stripe (8 taken) and the stripe's
        // own two saved boundary lines above / below: the strip's rows + halo and boundary lines [2 b, 2 e) only, not the whole planes
        const int    y0 = b * sh - off > 0 ? b * sh - off : 0, y1 = e * sh - off < (int)L.height ? e * sh - off : (int)L.height;
        const int    h0 = y0 - 8 > 0 ? y0 - 8 : 0, h1 = y1 + 8 < (int)L.height ? y1 + 8 : (int)L.height;
        const size_t drow_in = (size_t)L.stride * px, brow = (size_t)L.boundary_stride * px, drow = (size_t)L.dst_stride * px, wb = (size_t)L.width * px;
        arena_begin(p, (size_t)(h1 - h0) * drow_in + 2 * (size_t)(2 * (e - b)) * brow + (size_t)(y1 - y0) * drow + unit_b + 8 * 256);
        HIP_CHECK(hipStreamWaitEvent(p.stream, P->ready, 0));
        jitter(*P, p.stream);
        SvtHipLrParams M = L;
        M.data           = mirror_rows(*P, p, L.data, drow_in, wb, h0, h1);
        M.boundary_above = mirror_rows(*P, p, L.boundary_above, brow, wb, 2 * b, 2 * e);
        M.boundary_below = mirror_rows(*P, p, L.boundary_below, brow, wb, 2 * b, 2 * e);
        M.units          = (const SvtHipLrUnit*)mirror_in(*P, p, L.units, unit_b);
        M.dst            = (uint8_t*)arena_take(p, plane_bytes(L.dst_stride, L.width, (size_t)(y1 - y0), px)) - (size_t)y0 * drow;
        jitter(*P, p.stream);
        svt_hip_lr_filter_frame_stripes(&M, b, e, p.stream);
        jitter(*P, p.stream);
        rows_out(*P, p, (uint8_t*)L.dst + y0 * drow, (const uint8_t*)M.dst + y0 * drow, plane_bytes(L.dst_stride, L.width, (size_t)(y1 - y0), px));
        HIP_CHECK(hipEventRecord(p.done, p.stream));
    }
    int b0, e0;
    strip_of(nstripes, 0, P->n, b0, e0);
    jitter(*P, home);
    if (e0 > b0) svt_hip_lr_filter_frame_stripes(params, b0, e0, home);
    close_call(*P, home);
    return 0;
    SVT_HIP_ENTRY_CATCH(SVT_HIP_E_DEVICE)
}

} // extern "C"
