/* enc_handle_binding.c -- REFERENCE-SIDE BINDING (what a maintainer of the reference adds; built into the reference encoder by oracle/Makefile for the identity / fps runs): the reference encoder with the binding of INTEGRATION.md §1.
 *
 * This translation unit IS Source/Lib/Globals/enc_handle.c of the reference (included below where it lies; nothing is copied)
 * plus the few lines a maintainer adds after enc_handle.c:1444-1445, where svt_av1_enc_init() assigns the run-time dispatch
 * table single-threaded, before init_fn_ptr() copies SAD pointers into svt_aom_mefn_ptr[] and before any worker thread exists:
 *
 *     svt_aom_setup_common_rtcd_internal(flags);
 *     svt_aom_setup_rtcd_internal(flags);
 *   + if (getenv("SVT_HIP")) { svt_hip_init(atoi(getenv("SVT_HIP"))); svt_hip_setup_rtcd(0); }
 *
 * The insertion is made by giving the second call a macro name for the duration of the #include.  The HIP library is
 * dlopen()ed (RTLD_GLOBAL, so its weak references to the RTCD pointer globals bind to libSvtAv1Enc's), which keeps this
 * encoder build free of any link-time dependency on ROCm: with SVT_HIP unset it is the plain C-only reference encoder
 * (the `--asm c` leg of the bitstream-identity check, .gitlab/workflows/linux/.gitlab-ci.yml:351-367).
 *
 * Environment: SVT_HIP=<device index> enables the hook; SVT_HIP_LIB=<path> overrides the library (the CPU test-suite points
 * it at the lock-step emulator build); SVT_HIP_ONLY=<comma list of pointer-name prefixes> and SVT_HIP_COUNT=<file> are read
 * by svt_hip_setup_rtcd itself (csrc/rtcd_hook.hip).
 */
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>

#include "aom_dsp_rtcd.h" /* declares svt_aom_setup_rtcd_internal before the macro below exists */

/* ---- several GPUs for one encode (frame-level sharding, no exchange step) -------------------------------------------------------------------
 * SVT_HIP_DEVICES=<d0,d1,...>: every picture-level stage seam runs the stages of picture n on device d[n % count].  A seam calls svt_hip_seam_bind(picture_number)
 * on the worker thread that is about to issue the picture's device calls: the thread is bound to that device (svt_hip_set_thread_device: host-call arenas and
 * streams are per thread and device) and the index into the list comes back -- the ME seam keeps one resident session per index.  Unset: one device, index 0. */
static struct {
    int n, ids[16];
    int (*set_thread_device)(int);
    unsigned long long per_device[16];
} SVT_HIP_SHARD;
int svt_hip_seam_device_count(void) { return SVT_HIP_SHARD.n > 0 ? SVT_HIP_SHARD.n : 1; }
int svt_hip_seam_device_id(int index) { return SVT_HIP_SHARD.n > 0 ? SVT_HIP_SHARD.ids[index] : -1; }
int svt_hip_seam_bind(unsigned long long picture_number) {
    if (SVT_HIP_SHARD.n <= 0) return 0;
    const int k = (int)(picture_number % (unsigned long long)SVT_HIP_SHARD.n);
    if (SVT_HIP_SHARD.set_thread_device(SVT_HIP_SHARD.ids[k])) { fprintf(stderr, "SVT_HIP_DEVICES: device %d is not available\n", SVT_HIP_SHARD.ids[k]); abort(); }
    __atomic_fetch_add(&SVT_HIP_SHARD.per_device[k], 1, __ATOMIC_RELAXED);
    return k;
}
static void svt_hip_shard_stats(void) {
    const char *f = getenv("SVT_HIP_DEVICES_STATS");
    FILE       *o = f ? fopen(f, "w") : NULL;
    if (!o) return;
    for (int k = 0; k < SVT_HIP_SHARD.n; k++) fprintf(o, "stage_calls_bound_to_device_%d %llu\n", SVT_HIP_SHARD.ids[k], SVT_HIP_SHARD.per_device[k]);
    fclose(o);
}

/* ---- host CPU time per stage (integration/seam_cpu.h) ---- */
#include "../integration/seam_cpu.h"
static struct { int on; unsigned long long ns[SEAM_CPU_STAGES], calls[SEAM_CPU_STAGES]; } SVT_HIP_CPU = {-1, {0}, {0}};
static void svt_hip_seam_cpu_stats(void) {
    static const char *names[SEAM_CPU_STAGES] = {"me", "tf", "tpl", "dlf", "cdef", "lr"};
    const char *f = getenv("SVT_HIP_SEAM_CPU_STATS");
    FILE       *o = f ? fopen(f, "w") : NULL;
    if (!o) return;
    for (int k = 0; k < SEAM_CPU_STAGES; k++) fprintf(o, "%s_cpu_ms %llu\n%s_calls %llu\n", names[k], SVT_HIP_CPU.ns[k] / 1000000ull, names[k], SVT_HIP_CPU.calls[k]);
    fclose(o);
}
int svt_hip_seam_cpu_on(void) {
    if (SVT_HIP_CPU.on < 0) { /* (racing first calls both arrive at the same answer; atexit may then be registered twice: the second write repeats the first) */
        const int on = getenv("SVT_HIP_SEAM_CPU_STATS") != NULL;
        if (on) atexit(svt_hip_seam_cpu_stats);
        SVT_HIP_CPU.on = on;
    }
    return SVT_HIP_CPU.on;
}
void svt_hip_seam_cpu_add(int stage, unsigned long long ns) {
    __atomic_fetch_add(&SVT_HIP_CPU.ns[stage], ns, __ATOMIC_RELAXED);
    __atomic_fetch_add(&SVT_HIP_CPU.calls[stage], 1, __ATOMIC_RELAXED);
}

#include <time.h>
static double svt_hip_ms_now(void) { struct timespec t_; clock_gettime(CLOCK_MONOTONIC, &t_); return 1e3 * (double)t_.tv_sec + 1e-6 * (double)t_.tv_nsec; }
static unsigned long long (*SVT_HIP_STRIPS_CALLS)(void);
static void svt_hip_strips_stats(void) {
    const char *f = getenv("SVT_HIP_STRIPS_STATS");
    FILE       *o = f ? fopen(f, "w") : NULL;
    if (!o) return;
    fprintf(o, "frame_launches_through_a_partition %llu\n", SVT_HIP_STRIPS_CALLS ? SVT_HIP_STRIPS_CALLS() : 0ull);
    fclose(o);
}
static void svt_aom_setup_rtcd_then_hip(EbCpuFlags flags) {
    svt_aom_setup_rtcd_internal(flags);
    const char *dev = getenv("SVT_HIP");
    if (!dev)
        return;
    const int    timing = getenv("SVT_HIP_INIT_TIMING") != NULL; /* one-time costs of the binding, phase by phase, to stderr */
    const double t_a = svt_hip_ms_now();
    const char *path = getenv("SVT_HIP_LIB");
    void       *h    = dlopen(path ? path : "libsvtav1_hip.so", RTLD_NOW | RTLD_GLOBAL);
    const double t_b = svt_hip_ms_now();
    if (!h) { /* (not silent: tools/enc_identity.py demands the "dispatch pointers now select" line below before it accepts a run as a HIP run) */
        fprintf(stderr, "SVT_HIP: cannot load the HIP variant (%s): the encoder continues on the reference's own kernels\n", dlerror());
        unsetenv("SVT_HIP"); /* the seams look at it: all of them stay off */
        return;
    }
    int (*init)(int)               = (int (*)(int))dlsym(h, "svt_hip_init");
    int (*setup)(unsigned long long) = (int (*)(unsigned long long))dlsym(h, "svt_hip_setup_rtcd");
    if (!init || !setup || init(atoi(dev)) != 0) {
        fprintf(stderr, "SVT_HIP: svt_hip_init(%s) failed: the encoder continues on the reference's own kernels\n", dev);
        unsetenv("SVT_HIP");
        return;
    }
    /* one-time costs now, while the encoder is still initialising, not inside the first picture's stage call: the device context, the library's code objects */
    const double t_c = svt_hip_ms_now();
    {   /* the pooled stage arenas are sized by what this encode will use: one per stage seam that is on (at most 3 are ever in flight), none for dispatch pointers only;
         * the loop-restoration search is the stage with the large device arena (SVT_HIP_WARM_ARENA_MB overrides its size, e.g. 448 for 4K) */
        static const char *const seams[] = {"SVT_HIP_ME_SEAM", "SVT_HIP_TF_ME_SEAM", "SVT_HIP_LR_SEAM", "SVT_HIP_CDEF_SEAM", "SVT_HIP_DLF_SEAM", "SVT_HIP_TPL_SEAM"};
        int on = 0;
        for (unsigned k = 0; k < sizeof(seams) / sizeof(seams[0]); k++) on += getenv(seams[k]) != NULL;
        const char *mb = getenv("SVT_HIP_WARM_ARENA_MB");
        void (*warm_sized)(int, unsigned) = (void (*)(int, unsigned))dlsym(h, "svt_hip_warmup_sized");
        void (*warmup)(void)              = (void (*)(void))dlsym(h, "svt_hip_warmup");
        if (warm_sized) warm_sized(on > 3 ? 3 : on, mb ? (unsigned)atoi(mb) : (getenv("SVT_HIP_LR_SEAM") ? 192u : 96u));
        else if (warmup) warmup();
    }
    if (timing) fprintf(stderr, "SVT_HIP_INIT_TIMING: dlopen %.1f ms, svt_hip_init (device context) %.1f ms, warm-up (code objects, first arenas) %.1f ms\n", t_b - t_a, t_c - t_b, svt_hip_ms_now() - t_c);
    const char *list = getenv("SVT_HIP_DEVICES");
    if (list && *list) {
        int (*count)(void) = (int (*)(void))dlsym(h, "svt_hip_device_count");
        *(void **)&SVT_HIP_SHARD.set_thread_device = dlsym(h, "svt_hip_set_thread_device");
        const int have = count ? count() : 0;
        for (const char *p = list; *p && SVT_HIP_SHARD.n < 16;) {
            const int d = atoi(p);
            if (d < 0 || d >= have || !SVT_HIP_SHARD.set_thread_device) { fprintf(stderr, "SVT_HIP_DEVICES: device %d of \"%s\" is not available (%d devices)\n", d, list, have); abort(); }
            SVT_HIP_SHARD.ids[SVT_HIP_SHARD.n++] = d;
            while (*p && *p != ',') p++;
            if (*p == ',') p++;
        }
        fprintf(stderr, "SVT_HIP_DEVICES: pictures are sharded over %d device(s) by picture number\n", SVT_HIP_SHARD.n);
        atexit(svt_hip_shard_stats);
    }
    /* SVT_HIP_STRIPS=<d0,d1,...>: ONE picture over several GPUs (the frame-partition case): the picture-sized host forms of the in-loop filters -- what
     * the CDEF and REST seams call -- cut their frame launches into strips over the listed devices (svt_hip_set_frame_partition; d0 = the device of SVT_HIP). */
    const char *strips = getenv("SVT_HIP_STRIPS");
    if (strips && *strips) {
        int ids[16], ns = 0;
        for (const char *p = strips; *p && ns < 16;) {
            ids[ns++] = atoi(p);
            while (*p && *p != ',') p++;
            if (*p == ',') p++;
        }
        int (*set_part)(const int *, int) = (int (*)(const int *, int))dlsym(h, "svt_hip_set_frame_partition");
        if (!set_part || set_part(ids, ns)) { fprintf(stderr, "SVT_HIP_STRIPS: \"%s\" is not a list of available devices\n", strips); abort(); }
        *(void **)&SVT_HIP_STRIPS_CALLS = dlsym(h, "svt_hip_frame_partition_host_calls");
        fprintf(stderr, "SVT_HIP_STRIPS: the in-loop filter stages cut every picture into strips over %d device(s)\n", ns);
        atexit(svt_hip_strips_stats);
    }
    int n = setup((unsigned long long)flags);
    fprintf(stderr, "SVT_HIP: %d dispatch pointers now select the HIP variant\n", n);
}

/* The second insertion: the LAST statement of svt_av1_enc_init (enc_handle.c:2318, `svt_print_memory_usage();`, a no-op macro outside DEBUG_MEMORY_USAGE builds) is given
 * this body, so that with the ME seam on the device sessions are created while the encoder initialises -- from the geometry of a real object of the PA-reference pool
 * the function has just built -- instead of inside the first picture's stage call. */
#include "enc_handle.h"
#include "svt_malloc.h"
#include "sys_resource_manager.h"
#include "reference_object.h" /* EbTplReferenceObject */
void svt_hip_seam_me_prepare(const void *pa_reference_object); /* integration/me_process_seam.c */
void svt_hip_seam_me_register_buffer(void *buffer, size_t bytes);
static void svt_hip_after_enc_init(EbEncHandle *h) {
    if (!getenv("SVT_HIP") || !getenv("SVT_HIP_ME_SEAM") || !h || !h->pa_reference_picture_pool_ptr_array) return;
    EbSystemResource *pool = h->pa_reference_picture_pool_ptr_array[0];
    if (!pool || !pool->object_total_count || !pool->wrapper_ptr_pool || !pool->wrapper_ptr_pool[0]) return;
    const int    timing = getenv("SVT_HIP_INIT_TIMING") != NULL;
    const double t_a = svt_hip_ms_now();
    svt_hip_seam_me_prepare(pool->wrapper_ptr_pool[0]->object_ptr);
    const double t_b = svt_hip_ms_now();
    /* the 8-bit luma planes every ME stage call uploads (the y8b pool of :1781-1796; pa_ref->input_padded_pic->buffer_y points into it): page-locked once, here */
    EbSystemResource *y8b = h->input_y8b_buffer_resource_ptr;
    for (uint32_t i = 0; y8b && y8b->wrapper_ptr_pool && i < y8b->object_total_count; i++) {
        const EbBufferHeaderType *hdr = y8b->wrapper_ptr_pool[i] ? (const EbBufferHeaderType *)y8b->wrapper_ptr_pool[i]->object_ptr : NULL;
        const EbPictureBufferDesc *pic = hdr ? (const EbPictureBufferDesc *)hdr->p_buffer : NULL;
        if (pic && pic->buffer_y) svt_hip_seam_me_register_buffer(pic->buffer_y, pic->luma_size);
    }
    /* the TPL reconstruction pictures the reconstruction seam uploads / downloads whole (mc_flow_rec_picture_buffer[] = pictures of this pool, src_ops_process.c:1835) */
    EbSystemResource *tpl = getenv("SVT_HIP_TPL_RECON_SEAM") && h->tpl_reference_picture_pool_ptr_array ? h->tpl_reference_picture_pool_ptr_array[0] : NULL;
    for (uint32_t i = 0; tpl && tpl->wrapper_ptr_pool && i < tpl->object_total_count; i++) {
        const EbTplReferenceObject *o = tpl->wrapper_ptr_pool[i] ? (const EbTplReferenceObject *)tpl->wrapper_ptr_pool[i]->object_ptr : NULL;
        if (o && o->ref_picture_ptr && o->ref_picture_ptr->buffer_y) svt_hip_seam_me_register_buffer(o->ref_picture_ptr->buffer_y, o->ref_picture_ptr->luma_size);
    }
    if (timing) fprintf(stderr, "SVT_HIP_INIT_TIMING: ME session(s) %.1f ms, page-locking the luma / TPL pools %.1f ms\n", t_b - t_a, svt_hip_ms_now() - t_b);
}
/* ... and the page locks are released at the start of svt_av1_enc_deinit (enc_handle.c:2364: its first svt_shutdown_process call; queues are drained by then), before
 * svt_av1_enc_deinit_handle destroys the pool that owns the buffers. */
void svt_hip_seam_me_unregister_buffer(void *buffer);
static void svt_hip_before_enc_deinit(EbEncHandle *h) {
    static int done;
    if (done || !getenv("SVT_HIP") || !getenv("SVT_HIP_ME_SEAM") || !h) return;
    done = 1;
    EbSystemResource *y8b = h->input_y8b_buffer_resource_ptr;
    for (uint32_t i = 0; y8b && y8b->wrapper_ptr_pool && i < y8b->object_total_count; i++) {
        const EbBufferHeaderType *hdr = y8b->wrapper_ptr_pool[i] ? (const EbBufferHeaderType *)y8b->wrapper_ptr_pool[i]->object_ptr : NULL;
        const EbPictureBufferDesc *pic = hdr ? (const EbPictureBufferDesc *)hdr->p_buffer : NULL;
        if (pic && pic->buffer_y) svt_hip_seam_me_unregister_buffer(pic->buffer_y);
    }
    EbSystemResource *tpl = getenv("SVT_HIP_TPL_RECON_SEAM") && h->tpl_reference_picture_pool_ptr_array ? h->tpl_reference_picture_pool_ptr_array[0] : NULL;
    for (uint32_t i = 0; tpl && tpl->wrapper_ptr_pool && i < tpl->object_total_count; i++) {
        const EbTplReferenceObject *o = tpl->wrapper_ptr_pool[i] ? (const EbTplReferenceObject *)tpl->wrapper_ptr_pool[i]->object_ptr : NULL;
        if (o && o->ref_picture_ptr && o->ref_picture_ptr->buffer_y) svt_hip_seam_me_unregister_buffer(o->ref_picture_ptr->buffer_y);
    }
}
#define svt_shutdown_process(r) (svt_hip_before_enc_deinit(handle), svt_shutdown_process(r)) /* (every use is inside svt_av1_enc_deinit, where `handle` is the encoder) */
#undef svt_print_memory_usage
#define svt_print_memory_usage() svt_hip_after_enc_init(enc_handle_ptr)

#define svt_aom_setup_rtcd_internal(flags) svt_aom_setup_rtcd_then_hip(flags)
#include "enc_handle.c" /* resolves through -I$(REF)/Source/Lib/Globals */
