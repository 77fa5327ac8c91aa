// hipemu.h -- TEST INFRASTRUCTURE ONLY (never part of the shipped library).
//
// A minimal lock-step SIMT interpreter that lets the *unmodified* HIP kernel
// sources under svt-av1-psy_amd/csrc be compiled with g++ and executed on the
// CPU of the development container (which has no GPU).  Every thread of a
// workgroup is a ucontext fiber; __syncthreads() and the wave-level exchange
// primitives (shuffles, DPP, readfirstlane, ballot) are rendezvous points at
// which fibers yield to a round-robin scheduler, so kernels observe the same
// barrier / cross-lane semantics they get on a 64-wide CDNA4 wavefront.
//
// Purpose: catch indexing / LDS-layout / reduction-order bugs before spending
// GPU minutes.  The product library (libsvtav1_hip.so) is built by hipcc only
// and has no way to select this path; `tests/` builds it into a separate
// libsvtav1_hipemu.so that only `pytest -m "not gpu"` loads.
#pragma once
#include <ucontext.h>
#include <time.h>
#include <sys/mman.h>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <functional>
#include <map>
#include <atomic>
#include <mutex>
#include <type_traits>
#include <vector>

#define SVT_HIP_EMU 1
#define __global__
#define __device__
#define __host__
#define __constant__ static const
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define HIP_DYNAMIC_SHARED(type, var) type* var = (type*)hipemu::g.dyn_smem;

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_emu {
    unsigned x, y, z;
};

typedef int   hipError_t;
typedef void* hipStream_t;
struct hipemuEvent {
    double t;
};
typedef hipemuEvent* hipEvent_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNoDevice = 100 };
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipStreamNonBlocking = 1, hipHostMallocDefault = 0, hipHostRegisterDefault = 0 };
struct hipDeviceProp_t {
    char name[256];
    char gcnArchName[256];
    int  multiProcessorCount;
    size_t totalGlobalMem;
};

// ThreadSanitizer build of the emulator (profiles/r05_tsan_seams.txt: g++ -fsanitize=thread over the same sources, loaded by a TSan build of the reference encoder): every
// lane is a ucontext fiber, and TSan has to be told about each switch or it sees one thread jumping between 1 024 stacks.
#if defined(__SANITIZE_THREAD__)
extern "C" {
void* __tsan_get_current_fiber(void);
void* __tsan_create_fiber(unsigned flags);
void  __tsan_destroy_fiber(void* fiber);
void  __tsan_switch_to_fiber(void* fiber, unsigned flags);
}
#define HIPEMU_TSAN_SWITCH(f) __tsan_switch_to_fiber((f), 0)
#else
#define HIPEMU_TSAN_SWITCH(f) ((void)0)
#endif
namespace hipemu {
constexpr int WAVE        = 64;
constexpr int MAX_THREADS = 1024;
constexpr size_t STACK_SZ = 256 * 1024;

struct Fiber {
    ucontext_t ctx;
    bool       done;
};
struct State {
    ucontext_t            main_ctx;
    void*                 tsan_main = nullptr;        // (ThreadSanitizer builds: the launching thread's own context ...
    void*                 tsan_fib[MAX_THREADS] = {}; //  ... and one per lane of the running workgroup)
    Fiber                 fib[MAX_THREADS];
    char*                 stacks = nullptr;
    int                   cur = 0, nthreads = 0, alive = 0;
    int                   bar_count = 0, bar_gen = 0;
    int                   wbar_count[MAX_THREADS / WAVE], wbar_gen[MAX_THREADS / WAVE], wave_alive[MAX_THREADS / WAVE];
    uint64_t              xchg[MAX_THREADS];
    std::function<void()> body;
    char*                 dyn_smem = nullptr;
    size_t                dyn_smem_sz = 0;
    dim3                  block_dim, grid_dim;
};
inline State g;
inline uint3_emu tIdx, bIdx;
inline dim3      bDim, gDim;

inline void yield() {
    HIPEMU_TSAN_SWITCH(g.tsan_main);
    swapcontext(&g.fib[g.cur].ctx, &g.main_ctx);
}
inline void set_tid(int t) {
    tIdx.x = t % g.block_dim.x;
    tIdx.y = (t / g.block_dim.x) % g.block_dim.y;
    tIdx.z = t / (g.block_dim.x * g.block_dim.y);
}
inline int  flat_tid() { return g.cur; }
inline int  lane_id() { return g.cur % WAVE; }
inline int  wave_id() { return g.cur / WAVE; }
inline void block_barrier() {
    int gen = g.bar_gen;
    if (++g.bar_count >= g.alive) {
        g.bar_count = 0;
        g.bar_gen++;
    } else {
        while (g.bar_gen == gen) yield();
    }
}
inline void wave_barrier() {
    int w   = wave_id();
    int gen = g.wbar_gen[w];
    if (++g.wbar_count[w] >= g.wave_alive[w]) {
        g.wbar_count[w] = 0;
        g.wbar_gen[w]++;
    } else {
        while (g.wbar_gen[w] == gen) yield();
    }
}
// all lanes of the wave publish `v`; each returns the value published by `src` (lane index in wave)
inline uint64_t wave_xchg(uint64_t v, int src, bool* src_valid = nullptr) {
    int w           = wave_id();
    g.xchg[g.cur]   = v;
    wave_barrier();
    int      n      = std::min(WAVE, g.nthreads - w * WAVE);
    bool     ok     = src >= 0 && src < n;
    uint64_t r      = ok ? g.xchg[w * WAVE + src] : v;
    if (src_valid) *src_valid = ok;
    wave_barrier();
    return r;
}
inline void fiber_entry() {
    g.body();
    g.fib[g.cur].done = true;
    g.alive--;
    g.wave_alive[g.cur / WAVE]--;
    // a thread that exits must release barriers other threads are waiting on
    if (g.bar_count >= g.alive && g.alive > 0 && g.bar_count > 0) {
        g.bar_count = 0;
        g.bar_gen++;
    }
    int w = g.cur / WAVE;
    if (g.wave_alive[w] > 0 && g.wbar_count[w] >= g.wave_alive[w] && g.wbar_count[w] > 0) {
        g.wbar_count[w] = 0;
        g.wbar_gen[w]++;
    }
    HIPEMU_TSAN_SWITCH(g.tsan_main);
    swapcontext(&g.fib[g.cur].ctx, &g.main_ctx);
}
inline void run_block(int nthreads) {
    if (!g.stacks) {
        g.stacks = (char*)mmap(nullptr, STACK_SZ * MAX_THREADS, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (g.stacks == (char*)MAP_FAILED) { perror("hipemu mmap"); abort(); }
    }
    g.nthreads  = nthreads;
    g.alive     = nthreads;
    g.bar_count = 0;
    for (int w = 0; w < (nthreads + WAVE - 1) / WAVE; w++) {
        g.wbar_count[w] = 0;
        g.wave_alive[w] = std::min(WAVE, nthreads - w * WAVE);
    }
    for (int t = 0; t < nthreads; t++) {
        getcontext(&g.fib[t].ctx);
        g.fib[t].ctx.uc_stack.ss_sp   = g.stacks + STACK_SZ * t;
        g.fib[t].ctx.uc_stack.ss_size = STACK_SZ;
        g.fib[t].ctx.uc_link          = &g.main_ctx;
        g.fib[t].done                 = false;
        makecontext(&g.fib[t].ctx, (void (*)())fiber_entry, 0);
    }
#if defined(__SANITIZE_THREAD__)
    g.tsan_main = __tsan_get_current_fiber();
    for (int t = 0; t < nthreads; t++) g.tsan_fib[t] = __tsan_create_fiber(0);
#endif
    int remaining = nthreads;
    while (remaining > 0) {
        remaining = 0;
        for (int t = 0; t < nthreads; t++) {
            if (g.fib[t].done) continue;
            g.cur = t;
            set_tid(t);
            HIPEMU_TSAN_SWITCH(g.tsan_fib[t]);
            swapcontext(&g.main_ctx, &g.fib[t].ctx);
            if (!g.fib[t].done) remaining++;
        }
    }
#if defined(__SANITIZE_THREAD__)
    for (int t = 0; t < nthreads; t++) __tsan_destroy_fiber(g.tsan_fib[t]);
#endif
}
// The fiber state (and every kernel's `__shared__` storage) is global, so launches from different host threads -- the encoder's
// worker threads in tests/test_encoder_identity.py -- are serialised.
inline std::mutex launch_mutex;
template <typename F> inline void launch(F&& f, dim3 grid, dim3 block, size_t shmem) {
    std::lock_guard<std::mutex> lock(launch_mutex);
    g.block_dim = block;
    g.grid_dim  = grid;
    bDim        = block;
    gDim        = grid;
    if (shmem > g.dyn_smem_sz) {
        free(g.dyn_smem);
        g.dyn_smem    = (char*)aligned_alloc(256, (shmem + 255) & ~size_t(255));
        g.dyn_smem_sz = shmem;
    }
    int nthreads = block.x * block.y * block.z;
    assert(nthreads <= MAX_THREADS);
    g.body = f;
    // SVT_HIPEMU_LDS_POISON=<byte>: the dynamic shared memory of every workgroup starts filled with a byte derived from this one and the workgroup's index -- on the
    // GPU, LDS holds whatever the previous workgroup on that CU left there, so a kernel that reads a cell it never wrote gives results that change from run to run
    // THERE and never here; with the poison the dependence shows up here as well (two runs with different bytes must agree)
    static const int lds_poison = [] { const char* e = getenv("SVT_HIPEMU_LDS_POISON"); return e ? (int)strtol(e, nullptr, 0) & 0xff : -1; }();
    for (unsigned bz = 0; bz < grid.z; bz++)
        for (unsigned by = 0; by < grid.y; by++)
            for (unsigned bx = 0; bx < grid.x; bx++) {
                bIdx.x = bx;
                bIdx.y = by;
                bIdx.z = bz;
                if (lds_poison >= 0 && shmem) memset(g.dyn_smem, (lds_poison + 37 * (int)(bx + by * 7 + bz * 13)) & 0xff, shmem);
                run_block(nthreads);
            }
}
template <typename T> inline uint64_t to_u64(T v) {
    static_assert(sizeof(T) <= 8, "shuffle type too wide");
    uint64_t u = 0;
    memcpy(&u, &v, sizeof(T));
    return u;
}
template <typename T> inline T from_u64(uint64_t u) {
    T v;
    memcpy(&v, &u, sizeof(T));
    return v;
}
} // namespace hipemu

#define threadIdx hipemu::tIdx
#define blockIdx hipemu::bIdx
#define blockDim hipemu::bDim
#define gridDim hipemu::gDim
#define warpSize 64

// ---- several emulated devices (SVT_HIPEMU_DEVICES=N) ------------------------------------------------------------------------------------
// The current device is per host thread (as in HIP); every device allocation and every created stream remembers the device that was current when
// it was made, and a copy / memset / kernel launch that names memory or a stream of ANOTHER device than the current one aborts with a message --
// on real hardware that is an invalid-device-pointer fault (no peer access is enabled anywhere in the library).  This is how the multi-device
// paths (sessions bound to a device, the seams' picture -> device sharding) are exercised on a machine without a GPU.
namespace hipemu {
struct Stream { int device; };
struct Alloc { size_t n; int device; };
inline int device_count() {
    static const int n = [] { const char* e = getenv("SVT_HIPEMU_DEVICES"); const int v = e ? atoi(e) : 1; return v < 1 ? 1 : (v > 16 ? 16 : v); }();
    return n;
}
inline int& cur_device() { static thread_local int d = 0; return d; }
inline std::map<uintptr_t, Alloc>& allocs() { static std::map<uintptr_t, Alloc> m; return m; }
inline std::mutex& alloc_lock() { static std::mutex m; return m; }
inline void check_ptr(const void* p, const char* what) {
    if (device_count() == 1 || !p) return;
    std::lock_guard<std::mutex> g(alloc_lock());
    auto it = allocs().upper_bound((uintptr_t)p);
    if (it == allocs().begin()) return; // not a device allocation (host memory)
    --it;
    if ((uintptr_t)p < it->first + it->second.n && it->second.device != cur_device()) {
        fprintf(stderr, "hipemu: %s touches memory of device %d while device %d is current\n", what, it->second.device, cur_device());
        abort();
    }
}
inline void check_stream(void* s, const char* what) {
    if (s && ((Stream*)s)->device != cur_device()) {
        fprintf(stderr, "hipemu: %s on a stream of device %d while device %d is current\n", what, ((Stream*)s)->device, cur_device());
        abort();
    }
}
template <typename... P, typename... A> inline void launch_k(void (*k)(P...), dim3 grid, dim3 block, size_t shmem, A... a) {
    launch([=]() { k(a...); }, grid, block, shmem);
}
} // namespace hipemu
#define HIP_KERNEL_NAME(...) __VA_ARGS__
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    (hipemu::check_stream((hipStream_t)(stream), #kernel), hipemu::launch_k(kernel, dim3(grid), dim3(block), (size_t)(shmem), __VA_ARGS__))

inline void __syncthreads() { hipemu::block_barrier(); }
inline void __builtin_amdgcn_s_barrier() { hipemu::block_barrier(); }
#define SVT_HIP_GLOBAL_AS /* no address spaces on the CPU */
#define SVT_HIP_WAVES_PER_EU(lo, hi)
#define SVT_HIP_OPAQUE_I32(x) ((void)0) /* device build: an empty asm that redefines x so loop-invariant code stays inside the loop */
inline void __builtin_amdgcn_wave_barrier() { hipemu::wave_barrier(); } // lanes run as fibers here: same-wave LDS hand-offs need the rendezvous
inline void __builtin_amdgcn_sched_barrier(int) {}
inline void __builtin_amdgcn_sched_group_barrier(int, int, int) {}
inline void __builtin_amdgcn_s_setprio(int) {}
inline void __threadfence() {}
inline void __builtin_amdgcn_fence(int, const char*) {}
#define __HIP_MEMORY_SCOPE_AGENT 3
#define __hip_atomic_load(p, order, scope) (*(volatile __typeof__(*(p))*)(p))
#define __hip_atomic_store(p, v, order, scope) (*(p) = (v))
// A sleeping lane gives the other fibers of its workgroup a turn: a wave that polls for a flag ANOTHER WAVE OF THE SAME WORKGROUP sets (the TPL reconstruction kernels put
// several independent single-wave blocks into one workgroup) would otherwise spin for ever -- on the device the waves of a workgroup run side by side.  (Workgroups still run
// one after the other in index order, so a wait for a LATER workgroup cannot end: the kernels only ever wait for earlier tickets.)
inline void __builtin_amdgcn_s_sleep(int) { hipemu::yield(); }
inline void __threadfence_block() {}

// ---- wave-level data movement -------------------------------------------------------------
template <typename T> inline T __shfl(T v, int src, int width = 64) {
    int lane = hipemu::lane_id();
    int base = lane & ~(width - 1);
    return hipemu::from_u64<T>(hipemu::wave_xchg(hipemu::to_u64(v), base + (src & (width - 1))));
}
template <typename T> inline T __shfl_xor(T v, int mask, int width = 64) {
    int lane = hipemu::lane_id();
    int src  = lane ^ mask;
    if ((src & ~(width - 1)) != (lane & ~(width - 1))) src = lane;
    return hipemu::from_u64<T>(hipemu::wave_xchg(hipemu::to_u64(v), src));
}
template <typename T> inline T __shfl_down(T v, unsigned d, int width = 64) {
    int lane = hipemu::lane_id();
    int src  = lane + (int)d;
    if ((src & ~(width - 1)) != (lane & ~(width - 1))) src = lane;
    return hipemu::from_u64<T>(hipemu::wave_xchg(hipemu::to_u64(v), src));
}
template <typename T> inline T __shfl_up(T v, unsigned d, int width = 64) {
    int lane = hipemu::lane_id();
    int src  = lane - (int)d;
    if (src < 0 || (src & ~(width - 1)) != (lane & ~(width - 1))) src = lane;
    return hipemu::from_u64<T>(hipemu::wave_xchg(hipemu::to_u64(v), src));
}
inline unsigned long long __ballot(int pred) {
    int                lane = hipemu::lane_id();
    unsigned long long m    = 0;
    // gather predicate bits: every lane publishes, every lane reads all
    int w                   = hipemu::wave_id();
    hipemu::g.xchg[hipemu::g.cur] = pred ? 1 : 0;
    hipemu::wave_barrier();
    int n = std::min(hipemu::WAVE, hipemu::g.nthreads - w * hipemu::WAVE);
    for (int i = 0; i < n; i++)
        if (hipemu::g.xchg[w * hipemu::WAVE + i]) m |= 1ull << i;
    hipemu::wave_barrier();
    (void)lane;
    return m;
}
inline int      __builtin_amdgcn_readfirstlane(int v) { return (int)hipemu::wave_xchg((uint32_t)v, 0); }
inline int      __builtin_amdgcn_readlane(int v, int lane) { return (int)hipemu::wave_xchg((uint32_t)v, lane); }
inline int      __builtin_amdgcn_ds_bpermute(int addr, int v) { return (int)hipemu::wave_xchg((uint32_t)v, (addr >> 2) & 63); }
inline unsigned __builtin_amdgcn_mbcnt_lo(unsigned, unsigned) { return 0; }
inline unsigned __lane_id() { return hipemu::lane_id(); }

// DPP (gfx9 encodings).  Lanes whose source is invalid keep `old` (bound_ctrl=0 semantics with row/bank mask 0xf)
inline int hipemu_dpp_src(int lane, int ctrl, bool* valid) {
    *valid  = true;
    int row = lane & ~15, l = lane & 15;
    if (ctrl <= 0xFF) { // quad_perm
        int q = lane & 3;
        return (lane & ~3) | ((ctrl >> (2 * q)) & 3);
    }
    if (ctrl >= 0x101 && ctrl <= 0x10F) { // row_shl: lane l reads l+n
        int s = l + (ctrl & 15);
        *valid = s < 16;
        return row | (s & 15);
    }
    if (ctrl >= 0x111 && ctrl <= 0x11F) { // row_shr: lane l reads l-n
        int s = l - (ctrl & 15);
        *valid = s >= 0;
        return row | (s & 15);
    }
    if (ctrl >= 0x121 && ctrl <= 0x12F) { // row_ror: lane l reads (l-n) mod 16
        return row | ((l - (ctrl & 15)) & 15);
    }
    if (ctrl == 0x140) return row | (15 - l); // row_mirror
    if (ctrl == 0x141) return row | (l < 8 ? 7 - l : 23 - l); // row_half_mirror
    if (ctrl == 0x142) { // row_bcast15: lane 15 of each row -> all lanes of next row
        *valid = row >= 16;
        return row - 1;
    }
    if (ctrl == 0x143) { // row_bcast31
        *valid = lane >= 32;
        return 31;
    }
    fprintf(stderr, "hipemu: unsupported dpp ctrl 0x%x\n", ctrl);
    abort();
}
inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    int  lane = hipemu::lane_id();
    bool valid;
    int  s   = hipemu_dpp_src(lane, ctrl, &valid);
    bool sv;
    int  got = (int)hipemu::wave_xchg((uint32_t)src, valid ? s : lane, &sv);
    bool en  = ((row_mask >> (lane >> 4)) & 1) && ((bank_mask >> ((lane >> 2) & 3)) & 1);
    if (!en) return old;
    if (!valid || !sv) return bound_ctrl ? 0 : old;
    return got;
}
inline int __builtin_amdgcn_mov_dpp(int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    return __builtin_amdgcn_update_dpp(0, src, ctrl, row_mask, bank_mask, bound_ctrl);
}

// ---- packed-byte VALU instructions (C models of the gfx9 ISA definitions) -----------------
inline unsigned __builtin_amdgcn_sad_u8(unsigned a, unsigned b, unsigned c) {
    for (int i = 0; i < 4; i++) {
        int x = (a >> (8 * i)) & 255, y = (b >> (8 * i)) & 255;
        c += (unsigned)std::abs(x - y);
    }
    return c;
}
inline unsigned __builtin_amdgcn_sad_u16(unsigned a, unsigned b, unsigned c) {
    for (int i = 0; i < 2; i++) {
        int x = (a >> (16 * i)) & 65535, y = (b >> (16 * i)) & 65535;
        c += (unsigned)std::abs(x - y);
    }
    return c;
}
inline unsigned long long __builtin_amdgcn_qsad_pk_u16_u8(unsigned long long s0, unsigned s1, unsigned long long s2) {
    unsigned long long d = 0;
    for (int k = 0; k < 4; k++) {
        unsigned ref = (unsigned)(s0 >> (8 * k));
        unsigned acc = (unsigned)((s2 >> (16 * k)) & 0xffff);
        unsigned r   = __builtin_amdgcn_sad_u8(ref, s1, acc) & 0xffff;
        d |= (unsigned long long)r << (16 * k);
    }
    return d;
}
inline unsigned __builtin_amdgcn_alignbyte(unsigned hi, unsigned lo, unsigned sh) {
    unsigned long long v = ((unsigned long long)hi << 32) | lo;
    return (unsigned)(v >> (8 * (sh & 3)));
}
inline unsigned __builtin_amdgcn_alignbit(unsigned hi, unsigned lo, unsigned sh) {
    unsigned long long v = ((unsigned long long)hi << 32) | lo;
    return (unsigned)(v >> (sh & 31));
}
typedef short hipemu_s2 __attribute__((vector_size(4)));
inline int __builtin_amdgcn_sdot2(hipemu_s2 a, hipemu_s2 b, int c, bool) { return c + (int)a[0] * (int)b[0] + (int)a[1] * (int)b[1]; }
// v_mfma_i32_16x16x64_i8: D[i][j] = C[i][j] + sum over the 64 k of A[i][k] * B[k][j], signed int8.  Lane l holds A[l & 15][16 (l >> 4) .. +15] and
// B[16 (l >> 4) .. +15][l & 15] (16 bytes each) and C/D[4 (l >> 4) + r][l & 15], r = 0..3.  Implemented with the wave exchange primitive.
typedef int hipemu_i32x4 __attribute__((vector_size(16)));
inline hipemu_i32x4 __builtin_amdgcn_mfma_i32_16x16x64_i8(hipemu_i32x4 a, hipemu_i32x4 b, hipemu_i32x4 c, int, int, int) {
    const int lane = hipemu::lane_id(), col = lane & 15, rg = lane >> 4;
    hipemu_i32x4 d = c;
    for (int kg = 0; kg < 4; kg++) {
        unsigned bw[4];
        for (int w = 0; w < 4; w++) bw[w] = (unsigned)hipemu::wave_xchg((unsigned)b[w], col + 16 * kg);
        for (int r = 0; r < 4; r++) {
            const int row = 4 * rg + r;
            for (int w = 0; w < 4; w++) {
                const unsigned aw = (unsigned)hipemu::wave_xchg((unsigned)a[w], row + 16 * kg);
                for (int e = 0; e < 4; e++) d[r] += (int)(signed char)(aw >> (8 * e)) * (int)(signed char)(bw[w] >> (8 * e));
            }
        }
    }
    return d;
}
// clang element-wise saturating subtraction on unsigned vectors (v_pk_sub_u16 clamp)
template <class V> inline V __builtin_elementwise_sub_sat(V a, V b) { return (a > b) ? (a - b) : (a - a); }
// clang's ext_vector_type is only used for the <2 x u16> operands of v_dot2_u32_u16; g++ spells the same 4-byte vector vector_size(4)
#define ext_vector_type(N) vector_size((N) * 2)
typedef unsigned short hipemu_us2 __attribute__((vector_size(4)));
inline unsigned __builtin_amdgcn_udot2(hipemu_us2 a, hipemu_us2 b, unsigned c, bool) {
    return c + (unsigned)a[0] * (unsigned)b[0] + (unsigned)a[1] * (unsigned)b[1];
}
// gfx950 v_permlane16_swap_b32 / v_permlane32_swap_b32: exchange rows (halves) between the two operands; both results are returned
struct hipemu_u2 { unsigned v[2]; unsigned operator[](int i) const { return v[i]; } };
inline hipemu_u2 __builtin_amdgcn_permlane16_swap(unsigned vdst, unsigned src0, bool, bool) {
    const int      l  = hipemu::lane_id();
    const unsigned ps = (unsigned)hipemu::wave_xchg(src0, l ^ 16), pd = (unsigned)hipemu::wave_xchg(vdst, l ^ 16);
    const bool     odd = (l >> 4) & 1;
    return hipemu_u2{{odd ? ps : vdst, odd ? src0 : pd}};
}
inline hipemu_u2 __builtin_amdgcn_permlane32_swap(unsigned vdst, unsigned src0, bool, bool) {
    const int      l  = hipemu::lane_id();
    const unsigned ps = (unsigned)hipemu::wave_xchg(src0, l ^ 32), pd = (unsigned)hipemu::wave_xchg(vdst, l ^ 32);
    const bool     hi = l >= 32;
    return hipemu_u2{{hi ? ps : vdst, hi ? src0 : pd}};
}
inline unsigned __builtin_amdgcn_udot4(unsigned a, unsigned b, unsigned c, bool) { // v_dot4_u32_u8
    for (int i = 0; i < 4; i++) c += ((a >> (8 * i)) & 0xff) * ((b >> (8 * i)) & 0xff);
    return c;
}
inline int __builtin_amdgcn_sdot4(int a, int b, int c, bool) { // v_dot4_i32_i8
    for (int i = 0; i < 4; i++) c += (int)(int8_t)((unsigned)a >> (8 * i)) * (int)(int8_t)((unsigned)b >> (8 * i));
    return c;
}
inline unsigned __builtin_amdgcn_perm(unsigned a, unsigned b, unsigned sel) {
    unsigned long long v = ((unsigned long long)a << 32) | b;
    unsigned           r = 0;
    for (int i = 0; i < 4; i++) {
        unsigned s = (sel >> (8 * i)) & 255, byte;
        if (s <= 7) byte = (unsigned)(v >> (8 * s)) & 255;
        else if (s == 12) byte = 0;
        else if (s >= 13) byte = 255;
        else byte = 0; // sign-replication selectors 8..11 unused here
        r |= byte << (8 * i);
    }
    return r;
}
inline int      __builtin_amdgcn_sbfe(int v, unsigned off, unsigned w) { return w == 0 ? 0 : (int)((int64_t)((uint64_t)(uint32_t)v << (64 - off - w)) >> (64 - w)) ; }
inline unsigned __builtin_amdgcn_ubfe(unsigned v, unsigned off, unsigned w) { return w >= 32 ? v >> off : (v >> off) & ((1u << w) - 1); }
inline int      __clz(unsigned v) { return v ? __builtin_clz(v) : 32; }
inline int      __popc(unsigned v) { return __builtin_popcount(v); }
inline int      __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int      __ffs(unsigned v) { return __builtin_ffs((int)v); }
inline int      __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
inline int      __mulhi(int a, int b) { return (int)(((long long)a * b) >> 32); }

// min/max overload set as in HIP device code
inline int                min(int a, int b) { return a < b ? a : b; }
inline unsigned           min(unsigned a, unsigned b) { return a < b ? a : b; }
inline long long          min(long long a, long long b) { return a < b ? a : b; }
inline unsigned long long min(unsigned long long a, unsigned long long b) { return a < b ? a : b; }
inline int                max(int a, int b) { return a > b ? a : b; }
inline unsigned           max(unsigned a, unsigned b) { return a > b ? a : b; }
inline long long          max(long long a, long long b) { return a > b ? a : b; }
inline unsigned long long max(unsigned long long a, unsigned long long b) { return a > b ? a : b; }

// atomics (single host thread => plain RMW)
template <typename T> inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <typename T> inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <typename T> inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <typename T> inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }
template <typename T> inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }

// ---- host runtime subset -----------------------------------------------------------------
inline hipError_t  hipGetLastError() { return hipSuccess; }
inline hipError_t  hipPeekAtLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t e) { return e == hipErrorOutOfMemory ? "hipemu: injected failure (SVT_HIPEMU_FAIL_AFTER)" : "hipemu"; }
// SVT_HIPEMU_FAIL_AFTER=<n>: the n-th device allocation / copy / synchronisation of the process and every one after it FAILS (hipErrorOutOfMemory) -- the library's
// error policy (svt_hip_common.h: HIP_CHECK) can then be exercised on the CPU: the encoder must finish, on the reference's own kernels, with an identical bitstream.
namespace hipemu {
inline bool inject_failure() {
    static const long limit = [] { const char* e = getenv("SVT_HIPEMU_FAIL_AFTER"); return e ? atol(e) : -1L; }();
    if (limit < 0) return false;
    static std::atomic<long> calls{0};
    return calls.fetch_add(1) >= limit;
}
} // namespace hipemu
inline hipError_t  hipGetDeviceCount(int* n) { *n = hipemu::device_count(); return hipSuccess; }
inline hipError_t  hipSetDevice(int d) { if (d < 0 || d >= hipemu::device_count()) return hipErrorInvalidValue; hipemu::cur_device() = d; return hipSuccess; }
inline hipError_t  hipGetDevice(int* d) { *d = hipemu::cur_device(); return hipSuccess; }
inline hipError_t  hipGetDeviceProperties(hipDeviceProp_t* p, int) {
    memset(p, 0, sizeof(*p));
    strcpy(p->name, "hipemu lock-step interpreter");
    strcpy(p->gcnArchName, "emu");
    p->multiProcessorCount = 1;
    return hipSuccess;
}
inline hipError_t hipMalloc(void** p, size_t n) {
    if (hipemu::inject_failure()) return hipErrorOutOfMemory;
    *p = aligned_alloc(256, (n + 255) & ~size_t(255));
    if (*p && hipemu::device_count() > 1) { std::lock_guard<std::mutex> g(hipemu::alloc_lock()); hipemu::allocs()[(uintptr_t)*p] = hipemu::Alloc{n, hipemu::cur_device()}; }
    return *p ? hipSuccess : 2;
}
inline hipError_t hipFree(void* p) {
    if (p && hipemu::device_count() > 1) { std::lock_guard<std::mutex> g(hipemu::alloc_lock()); hipemu::allocs().erase((uintptr_t)p); }
    free(p);
    return hipSuccess;
}
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { *p = aligned_alloc(256, (n + 255) & ~size_t(255)); return *p ? hipSuccess : 2; }
inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
enum { hipHostRegisterPortable = 1, hipMemoryTypeHost = 1, hipDeviceScheduleBlockingSync = 4 };
inline hipError_t hipSetDeviceFlags(unsigned) { return hipSuccess; }
struct hipPointerAttribute_t { int type; };
inline hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void*) { a->type = 0; return 1; } // (the emulator has no page-locked memory: always the staged path)
inline hipError_t hipHostRegister(void*, size_t, unsigned) { return hipSuccess; }
inline hipError_t hipHostUnregister(void*) { return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { hipemu::check_ptr(d, "hipMemcpy"); hipemu::check_ptr(s, "hipMemcpy"); memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t st = nullptr) {
    if (hipemu::inject_failure()) return hipErrorOutOfMemory;
    hipemu::check_stream(st, "hipMemcpyAsync"); hipemu::check_ptr(d, "hipMemcpyAsync"); hipemu::check_ptr(s, "hipMemcpyAsync");
    memcpy(d, s, n);
    return hipSuccess;
}
// a peer copy is LEGAL between devices: each pointer must belong to the device it is said to live on (a home-device pointer passed off as a peer's, or the other
// way round, is the bug this catches); the stream may be either device's
namespace hipemu {
inline int device_of(const void* p) {
    if (device_count() == 1 || !p) return cur_device();
    std::lock_guard<std::mutex> g(alloc_lock());
    auto it = allocs().upper_bound((uintptr_t)p);
    if (it == allocs().begin()) return -1;
    --it;
    return (uintptr_t)p < it->first + it->second.n ? it->second.device : -1;
}
} // namespace hipemu
inline hipError_t hipMemcpyPeerAsync(void* d, int dd, const void* s, int sd, size_t n, hipStream_t st = nullptr) {
    // (memory the emulator did not allocate -- the tests' numpy arrays standing in for the home device's buffers -- has no owner and passes)
    if (hipemu::device_count() > 1 && ((hipemu::device_of(d) >= 0 && hipemu::device_of(d) != dd) || (hipemu::device_of(s) >= 0 && hipemu::device_of(s) != sd))) {
        fprintf(stderr, "hipemu: hipMemcpyPeerAsync: destination lives on device %d (said %d), source on device %d (said %d)\n", hipemu::device_of(d), dd, hipemu::device_of(s), sd);
        abort();
    }
    if (st && ((hipemu::Stream*)st)->device != dd && ((hipemu::Stream*)st)->device != sd) { fprintf(stderr, "hipemu: hipMemcpyPeerAsync on a stream of a third device\n"); abort(); }
    memcpy(d, s, n);
    return hipSuccess;
}
inline hipError_t hipDeviceCanAccessPeer(int* can, int, int) { *can = 1; return hipSuccess; }
inline hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
inline hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t st = nullptr) {
    hipemu::check_stream(st, "hipMemcpy2DAsync"); hipemu::check_ptr(d, "hipMemcpy2DAsync"); hipemu::check_ptr(s, "hipMemcpy2DAsync");
    for (size_t y = 0; y < h; y++) memcpy((char*)d + y * dp, (const char*)s + y * sp, w);
    return hipSuccess;
}
inline hipError_t hipMemset(void* d, int v, size_t n) { hipemu::check_ptr(d, "hipMemset"); memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st = nullptr) {
    hipemu::check_stream(st, "hipMemsetAsync"); hipemu::check_ptr(d, "hipMemsetAsync");
    memset(d, v, n);
    return hipSuccess;
}
inline hipError_t hipStreamCreate(hipStream_t* s) { *s = new hipemu::Stream{hipemu::cur_device()}; return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = new hipemu::Stream{hipemu::cur_device()}; return hipSuccess; }
inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned f, int) { return hipStreamCreateWithFlags(s, f); }
inline hipError_t hipStreamDestroy(hipStream_t s) { delete (hipemu::Stream*)s; return hipSuccess; }
inline hipError_t hipStreamGetDevice(hipStream_t s, int* d) { *d = s ? ((hipemu::Stream*)s)->device : hipemu::cur_device(); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipemu::inject_failure() ? (hipError_t)hipErrorOutOfMemory : (hipError_t)hipSuccess; }
// HIP graphs: the interpreter executes a launch when it is issued, so "capture" runs the work once and a graph launch cannot replay it;
// the product's graph entry points link, CPU tests do not use them.
typedef void* hipGraph_t;
typedef void* hipGraphExec_t;
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal = 0, hipStreamCaptureModeThreadLocal = 1, hipStreamCaptureModeRelaxed = 2 };
#define hipStreamNonBlocking 1u
inline hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return hipSuccess; }
inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = nullptr; return hipSuccess; }
inline hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t, void*, void*, size_t) { *e = nullptr; return hipSuccess; }
inline hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
#define hipHostMallocDefault 0u
#define hipEventDisableTiming 2u
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = 0; return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new hipemuEvent{0}; return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new hipemuEvent{0}; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) {
    timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); e->t = ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; return hipSuccess;
}
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t - a->t); return hipSuccess; }
