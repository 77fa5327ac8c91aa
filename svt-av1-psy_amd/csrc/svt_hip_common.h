// svt_hip_common.h -- host-side plumbing shared by every kernel family of libsvtav1_hip.so.
//  * HIP_CHECK: a HIP error never propagates into the encoder and never kills it (SURVEY 8b "errors").  The first one is recorded (svt_hip_last_error), the dispatch
//    pointers svt_hip_setup_rtcd overwrote are put back -- the encoder continues on the SIMD variant the reference had selected --, every stage entry point returns
//    non-zero from then on (the seams decline and run the reference's own function), and the call in flight unwinds with a C++ exception to the entry point that
//    catches it (the per-pointer guard of rtcd_hook.hip finishes the call through the saved pointer).  There is no CPU path in this library.
//  * HostCall: per-thread stream + growable device arena + pinned staging, used by the `*_hip` RTCD-signature
//    functions that receive plain host pointers (Source/Lib/Codec/aom_dsp_rtcd.h contract: synchronous,
//    re-entrant, caller owns every buffer, arbitrary strides).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

// Redefines a 32-bit per-lane value through an empty asm: everything computed from it afterwards is no longer loop invariant for the compiler, which
// otherwise hoists dozens of per-lane LDS / global addresses out of a short outer loop and doubles the kernel's VGPR count.  (The CPU emulator used by the
// tests defines it as a no-op first.)
// A pointer rebuilt from integer arithmetic (row maps that select between planes), or taken from a struct passed by value, has no address space and
// the compiler emits FLAT loads, which also occupy the LDS queue; picture planes are always global memory, so the staging loads say so.  Native vector
// types only: a struct would be copied through its (generic-reference) copy constructor and the address space would be lost again.  (The CPU emulator
// defines SVT_HIP_GLOBAL_AS as nothing.)
#ifndef SVT_HIP_GLOBAL_AS
#define SVT_HIP_GLOBAL_AS __attribute__((address_space(1)))
#endif
typedef uint32_t svt_u32x4_a2 __attribute__((vector_size(16), aligned(2))); // 16 bytes at any even address
typedef uint32_t svt_u32x2_a1 __attribute__((vector_size(8), aligned(1)));  // 8 bytes at any address
__device__ __forceinline__ svt_u32x4_a2 svt_hip_global_load_x4(const void* p) { return *(const SVT_HIP_GLOBAL_AS svt_u32x4_a2*)p; }
__device__ __forceinline__ svt_u32x2_a1 svt_hip_global_load_x2(const void* p) { return *(const SVT_HIP_GLOBAL_AS svt_u32x2_a1*)p; }
// Write-through hand-off between workgroups of ONE launch (MI355X_MICROARCH.md, "cross-CU hand-off"): the producer's payload leaves the XCD's L2 with `sc0 sc1` stores,
// `s_waitcnt vmcnt(0)` drains them, a relaxed agent-scope store publishes the flag -- instead of an agent-scope release fence, which writes back the whole L2's dirty lines
// (1.7-6.5 us per fence, per workgroup).  The consumer still polls the flag with relaxed agent loads and then executes ONE agent-scope acquire before its plain loads.
// (s_nop: a store of more than 64 bits must not be followed directly by a write of its data registers -- the compiler pads that hazard for its own stores and cannot see
//  into this one; without it the next instruction overwrote the first data dword: four wrong samples per row, gpurun call 36.  The CPU emulator defines SVT_HIP_EMU.)
__device__ __forceinline__ void svt_hip_store_x4_wt(void* p, const uint32_t x, const uint32_t y, const uint32_t z, const uint32_t w) {
#ifdef SVT_HIP_EMU
    uint32_t v[4] = {x, y, z, w};
    memcpy(p, v, 16);
#else
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 d = {x, y, z, w};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(d) : "memory");
#endif
}
__device__ __forceinline__ void svt_hip_drain_stores() {
#ifndef SVT_HIP_EMU
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}
#ifndef SVT_HIP_WAVES_PER_EU // (waves per SIMD the register allocation of a kernel is cut for; the CPU emulator defines it as nothing)
#define SVT_HIP_WAVES_PER_EU(lo, hi) __attribute__((amdgpu_waves_per_eu(lo, hi)))
#endif
#ifndef SVT_HIP_OPAQUE_I32
#define SVT_HIP_OPAQUE_I32(x) asm volatile("" : "+v"(x))
#endif

// Workgroups are handed to the eight XCDs round-robin (workgroup b runs on XCD b % 8, each XCD with its own L2): give every XCD a CONTIGUOUS run of work items, so
// that items which share cache lines (neighbouring SBs / filter blocks / stripe columns of a picture) meet in one L2 instead of being fetched by two.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t n) {
    const uint32_t per = n >> 3;
    if (per == 0 || b >= per * 8) return b;
    return (b & 7) * per + (b >> 3);
}

namespace svthip {
struct DeviceError { int code; };                                                     // thrown by device_fail, caught at the guarded entry points
[[noreturn]] void device_fail(int code, const char* what, const char* file, int line); // runtime.hip
bool               failed();                                                          // a HIP error has switched the device path off
}
#define HIP_CHECK(expr)                                                                      \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) svthip::device_fail((int)e_, #expr, __FILE__, __LINE__);       \
    } while (0)
// the body of a stage / host-form entry point that returns int: non-zero (SVT_HIP_E_DEVICE) once the device path is off, or when this call hits the first error
#ifndef SVT_HIP_E_DEVICE
#define SVT_HIP_E_DEVICE (-100)
#endif
#define SVT_HIP_ENTRY_TRY try {
#define SVT_HIP_ENTRY_CATCH(ret) } catch (const svthip::DeviceError&) { return ret; }

#define SVT_LAUNCH_CHECK() HIP_CHECK(hipGetLastError())

struct SvtHipCdefParams;
struct SvtHipLrParams;
struct SvtHipTplSrcParams;
struct SvtHipTplReconParams;
struct SvtHipTplSrcStats;
struct SvtHipTplReconStats;

namespace svthip {

constexpr int MAX_DEVICES = 16; // width of the per-device tables (host-call arenas, the lease pool)

void ensure_device(); // binds the calling thread to its device (the default of svt_hip_init, or the one svt_hip_set_thread_device / a DeviceGuard chose);
                      // aborts with a clear message when svt_hip_init() found no GPU
int  current_device();
int  physical_device(int logical); // the GPU ordinal behind a logical device number (runtime.hip: SVT_HIP_VIRTUAL_DEVICES); what hipSetDevice / hipMemcpyPeerAsync take
// Makes `device` the calling thread's device for the guard's lifetime: every entry point of an object that lives on one device (an ME session) opens with it.
struct DeviceGuard {
    int prev;
    explicit DeviceGuard(int device);
    ~DeviceGuard();
};

// Bump allocator over one device buffer and one pinned host buffer; reset at the start of every host call.
struct HostCall {
    hipStream_t stream   = nullptr;
    uint8_t*    dev      = nullptr;
    size_t      dev_cap  = 0, dev_used = 0;
    uint8_t*    pin      = nullptr;
    size_t      pin_cap  = 0, pin_used = 0;

    void begin();
    // The same for a wrapper that moves a few kilobytes (one block per call): the call's "device" allocations come from the PINNED HOST arena, which the GPU reads and
    // writes directly over PCIe -- up() is a memcpy, down() a synchronisation + a memcpy, and the two or three DMA operations of a staged call (5-8 us each to submit, more
    // to complete) disappear.  Only for kernels that touch those buffers with plain loads and stores (no atomics, no hipMemset on them).
    void begin_small();
    bool zc = false;
    // device allocation of `bytes` (256-B aligned); may grow the arena (only legal before any kernel was queued
    // in this call, which is how every wrapper uses it: reserve(total) first, then carve).
    void  reserve(size_t dev_bytes, size_t pin_bytes);
    void* dalloc(size_t bytes);
    void* palloc(size_t bytes);
    // strided host rectangle -> packed device rectangle (row pitch = width bytes rounded up to `dpitch`)
    void up2d(void* ddst, size_t dpitch, const void* hsrc, size_t spitch, size_t width_bytes, size_t rows);
    void up(void* ddst, const void* hsrc, size_t bytes);
    void down(void* hdst, const void* dsrc, size_t bytes);
    void down2d(void* hdst, size_t hpitch, const void* dsrc, size_t dpitch, size_t width_bytes, size_t rows);
    // the same, DEFERRED: the device-to-pinned copy is enqueued now, the copy into the caller's memory happens in finish() after ONE synchronisation -- a host form
    // that returns several arrays pays one wait instead of one per array (up to 16 pending copies)
    void down_later(void* hdst, const void* dsrc, size_t bytes);
    void down2d_later(void* hdst, size_t hpitch, const void* dsrc, size_t dpitch, size_t width_bytes, size_t rows);
    void finish();
    struct Pending { void* h; const uint8_t* p; size_t hpitch, dpitch, width, rows; };
    Pending pend[16];
    int     n_pend = 0;
    void sync();
    // Commit discipline (what rtcd_hook.hip's Guard relies on when it finishes a failed `_hip` call through the saved dispatch pointer, and what every seam relies on when a
    // stage entry declines): a host form writes caller memory only after its LAST HIP operation has succeeded.  down() / down2d() / finish() mark the call committed;
    // an upload, allocation, download or synchronisation issued after that is counted (svt_hip_debug_commit_violations) -- the reference-fixture binary and
    // tests/test_rtcd_hook.py run every `_hip` wrapper and require the count to stay zero.
    bool committed = false;
    void touch();
};
HostCall& host_call();
struct ThreadStreams { hipStream_t st[3] = {nullptr, nullptr, nullptr}; hipEvent_t ev[6] = {}; int32_t* pinned = nullptr; }; // st[2]: highest priority; pinned: 64 page-locked host words (read-backs that must not stall the enqueueing thread)
struct StreamSetLease { // a pooled set of side streams + events on the calling thread's device, for the duration of one stage call (runtime.hip)
    ThreadStreams* set;
    int            device;
    StreamSetLease();
    ~StreamSetLease();
    StreamSetLease(const StreamSetLease&) = delete;
    StreamSetLease& operator=(const StreamSetLease&) = delete;
};
uint32_t* stream_scratch_u32x4(hipStream_t st);
// the TPL dispenser with the option set of tpl levels 0-3 (tpl_full.hip): every intra mode, SATD costs, sub-pel vectors, rate
bool tpl_full_wanted(const ::SvtHipTplSrcParams& P);
bool tpl_full_supported(const ::SvtHipTplSrcParams& P);
void tpl_full_src_launch(const ::SvtHipTplSrcParams& P, const uint8_t* src, const uint8_t* ref, const uint8_t* tot, const uint32_t* mv, const uint8_t* cand,
                         ::SvtHipTplSrcStats* stats, hipStream_t st);
void tpl_full_recon_launch(const ::SvtHipTplReconParams& R, const uint8_t* src, const uint8_t* ref, const ::SvtHipTplSrcStats* ss, uint8_t* rec,
                           ::SvtHipTplReconStats* out, uint32_t* sync, int cols16, int rows16, int wt, hipStream_t st);
// device-resident copies of host picture planes kept across host calls (runtime.hip): acquire pins an entry for (host buffer, content id) on the current device
uint8_t* plane_cache_acquire(const void* host_ptr, uint64_t id, size_t bytes, bool* hit, int* token);
void     plane_cache_release(int token, bool now_ready);
void     plane_cache_drop(const void* host_ptr);
void     plane_cache_counts(uint64_t* hits, uint64_t* misses); // four zeroed device words for one launch sequence on `st` (runtime.hip)
// The stage-sized host forms (a whole picture's planes per call: svt_hip_tf_picture_host, svt_hip_tpl_src_stage_host, the CDEF / LR / deblocking / sub-pel host forms)
// do not use the calling thread's own arena: an encoder calls them from dozens of worker threads, and every thread growing a private 20-50 MB pinned + device arena
// on its first call costs 10-20 ms each (pinned allocation).  They lease an arena from a per-device pool for the duration of the call instead -- as many arenas as
// calls are ever in flight at once.  (All of these forms synchronise before they return, so a returned arena is idle.)
struct HostCallLease {
    HostCall* c;
    int       device;
    HostCallLease();
    ~HostCallLease();
    HostCallLease(const HostCallLease&) = delete;
    HostCallLease& operator=(const HostCallLease&) = delete;
    HostCall& operator*() { return *c; }
};

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// XCD-aware placement of a launch whose workgroups walk a picture in raster order: the hardware hands workgroups to the 8 XCDs round-robin by id, so raster neighbours --
// whose windows overlap -- land on 8 different L2s and each L2 fetches its own copy of the overlap.  With xcd_band_per(n_wg) != 0 the launch takes 8 * per workgroups and
// workgroup b works at logical position (b % 8) * per + b / 8: every XCD walks one contiguous band.  Positions >= n_wg are padding (the kernel's own bounds check drops them).
// An affinity for speed only -- results never depend on it.  (hme_chain_kernel: 7.7x -> 1.9x of the planes' bytes moved, 62.8 -> 58.0 us; profiles/r06_*.)
static inline uint32_t xcd_band_per(uint32_t n_wg) { return n_wg >= 64 ? (n_wg + 7) / 8 : 0; }

// Measurement knobs (environment table): read from the environment ONCE (first use) instead of getenv() on every launch -- getenv is hot-path
// work and is not safe against a concurrent setenv in a multi-threaded encoder.  svt_hip_tuning_reload() re-reads them (tests that sweep a knob call it).
int tuning_lr_rows_per_workgroup(); // SVT_HIP_LR_UR: 16 / 32 / 64, default 32
int tuning_cdef_groups_per_workgroup(); // SVT_HIP_CDEF_GPW: 1 / 2 / 4, 0 = by frame size
int tuning_sad_form(); // SVT_HIP_SAD_FORM: 0 (default: pair-per-wave forms) / 1 (strip form, measured slower): which independent-pairs SAD kernel runs
int tuning_cdef_search_minb(); // SVT_HIP_CDEF_MINB: 3 (default) / 2: which register budget of the CDEF search kernel runs (A/B measurement)

// frame launches of the picture-sized host forms: through the calling thread's frame partition when svt_hip_set_frame_partition named several devices (partition.hip)
void cdef_frame_dispatch(int mode, const ::SvtHipCdefParams* P, hipStream_t st);
void lr_frame_dispatch(const ::SvtHipLrParams* P, hipStream_t st);
void partition_pool_free(); // the pooled partitions of the host forms (partition.hip), at svt_hip_shutdown

} // namespace svthip

// The runtime loads a translation unit's code object at the first launch of one of its kernels (tens of milliseconds for the larger ones: the first loop-restoration
// stage call of an encode took 51 ms, profiles/r05_lr_seam_calls.txt).  Every kernel file defines an empty kernel + a launcher with this macro; svt_hip_warmup()
// (runtime.hip), which an encoder calls while it initialises, launches them all, so that no stage pays the load inside its first picture.
#define SVT_HIP_DEFINE_WARM(tu)                                                                                              \
    __global__ void svt_hip_warm_kernel_##tu(int) {}                                                                         \
    namespace svthip { void warm_##tu(hipStream_t st) { hipLaunchKernelGGL(svt_hip_warm_kernel_##tu, dim3(1), dim3(1), 0, st, 0); } }
