/* dlf_process_seam.c -- REFERENCE-SIDE BINDING (what a maintainer of the reference adds; built into the reference encoder by oracle/Makefile for the identity / fps runs): the reference's deblocking process with the frame-filter seam of INTEGRATION.md §3.
 *
 * This translation unit IS Source/Lib/Codec/dlf_process.c of the reference (included below where it lies; nothing is copied).  The one change: the call
 *
 *     svt_av1_loop_filter_frame(recon_buffer, pcs, 0, 3);                                                                       (dlf_process.c:122)
 *
 * is given a macro name for the duration of the #include and lands in seam_loop_filter_frame() below.  With SVT_HIP_DLF_SEAM unset (or the HIP library not
 * loaded) that function IS the reference call.  With SVT_HIP_DLF_SEAM=1 the reference's own driver (svt_av1_loop_filter_frame -> svt_aom_loop_filter_sb ->
 * set_lpf_parameters: filter levels, transform-size edges, skip -- the serial, mode-info dependent part) still runs, but the sixteen leaf pointers
 * svt_aom_[highbd_]lpf_{horizontal,vertical}_{4,6,8,14} are recording functions while this thread is inside the seam: they append the 4-sample segment
 * (position, length, blimit / limit / thresh) to a list instead of filtering.  The driver's decisions do not depend on sample values, so the list is exactly
 * what it would have filtered; every plane is then filtered by ONE svt_hip_lpf_plane_host() call (all vertical-edge segments, then all horizontal ones).
 * Outside the seam (other threads, svt_av1_pick_filter_level's trial filterings) the recording functions forward to the pointers they replaced.
 * SVT_HIP_DLF_SEAM_STATS=<file> receives the counters at exit.
 *
 * Round 3 -- the path presets >= 7 actually take (dlf_ctrls.sb_based_dlf, enc_mode_config.c:1466-1487): there the picture is deblocked SB by SB INSIDE the coding loop
 * (coding_loop.c:2278-2298: svt_aom_loop_filter_sb right after an SB is reconstructed) and this process only passes it on.  integration/coding_loop_seam.c sends that
 * one call to svt_hip_seam_loop_filter_sb() below: the reference's svt_aom_loop_filter_sb still runs (filter levels, transform edges, skip: its own decisions) with
 * the RECORDING leaf functions, the segments of every SB are appended to the picture's lists, and the picture is filtered by one device call per plane when it
 * reaches this process -- before anything reads the deblocked reconstruction (the restoration boundary lines, CDEF: dlf_process.c:133-160; the call that marks the
 * spot is the svt_aom_get_recon_pic of the pre-CDEF preparation, :136, told apart from the file's other calls with __COUNTER__).  Nothing between an SB's
 * reconstruction and this point reads deblocked samples: intra prediction of later SBs uses the pre-filter neighbour arrays, and the picture becomes a reference
 * only after the in-loop filters.  The set of segments is the one the frame-level driver produces (it calls the same svt_aom_loop_filter_sb for every SB,
 * deblocking_filter.c:642-653), and the device applies all vertical edges, then all horizontal ones -- the order of the standard, which the reference's staggered
 * in-place order must (and does) reproduce.
 */
#define _GNU_SOURCE /* RTLD_DEFAULT */
#include "../integration/seam_cpu.h"
#include <dlfcn.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pcs.h"
#include "sequence_control_set.h"
#include "aom_dsp_rtcd.h"
#include "deblocking_filter.h" /* svt_aom_loop_filter_sb */
#include "svtav1_hip.h" /* include/svtav1_hip.h of this repository: the C ABI */

void svt_av1_loop_filter_frame(EbPictureBufferDesc *frame_buffer, PictureControlSet *pcs, int32_t plane_start, int32_t plane_end);

typedef void (*LpfFn)(uint8_t *s, int32_t pitch, const uint8_t *blimit, const uint8_t *limit, const uint8_t *thresh);
typedef void (*LpfHbdFn)(uint16_t *s, int32_t pitch, const uint8_t *blimit, const uint8_t *limit, const uint8_t *thresh, int32_t bd);
#include <time.h>
static double seam_ms_now(void) { struct timespec t_; clock_gettime(CLOCK_MONOTONIC, &t_); return 1e3 * (double)t_.tv_sec + 1e-6 * (double)t_.tv_nsec; }
static struct {
    pthread_mutex_t lock;
    int             mode;
    int (*plane_host)(void *, uint32_t, uint32_t, uint32_t, int, int, const SvtHipLpfEdge *, uint32_t, const SvtHipLpfEdge *, uint32_t); /* non-zero: the device path is off */
    unsigned long long us_stage; /* microseconds inside svt_hip_lpf_plane_host */
    LpfFn    orig[2][4];     /* [vertical][length index 4, 6, 8, 14] */
    LpfHbdFn orig_hbd[2][4];
    uint64_t n_pictures, n_segments, n_sb_pictures, n_sb_calls, n_replayed;
} F = {PTHREAD_MUTEX_INITIALIZER};

typedef struct { SvtHipLpfEdge *e; uint32_t n, cap; } EdgeList;
static __thread struct {
    int            on;
    const uint8_t *base[3];   /* byte address of sample (0, 0) of each plane */
    size_t         stride[3]; /* samples */
    uint32_t       rows[3];
    int            px;
    EdgeList       list[3][2]; /* [plane][vertical] */
} T;

static void record(const void *s, int vertical, int len, const uint8_t *blimit, const uint8_t *limit, const uint8_t *thresh) {
    for (int pl = 0; pl < 3; pl++) {
        const ptrdiff_t off = (const uint8_t *)s - T.base[pl];
        if (off < 0 || (size_t)off >= T.stride[pl] * T.rows[pl] * T.px) continue;
        EdgeList *L = &T.list[pl][vertical];
        if (L->n == L->cap) { L->cap = L->cap ? 2 * L->cap : 4096; L->e = realloc(L->e, (size_t)L->cap * sizeof(*L->e)); }
        SvtHipLpfEdge *e = &L->e[L->n++];
        memset(e, 0, sizeof(*e));
        e->y = (uint32_t)((size_t)off / T.px / T.stride[pl]); e->x = (uint32_t)((size_t)off / T.px % T.stride[pl]);
        e->vertical = (uint8_t)vertical; e->length = (uint8_t)len; e->blimit = *blimit; e->limit = *limit; e->thresh = *thresh;
        return;
    }
    fprintf(stderr, "SVT_HIP_DLF_SEAM: a segment outside the picture's planes\n");
    abort();
}
#define REC(V, VN, LI, LEN)                                                                                                                              \
    static void rec_##VN##_##LEN(uint8_t *s, int32_t pitch, const uint8_t *b, const uint8_t *l, const uint8_t *t) {                                      \
        if (T.on) record(s, V, LEN, b, l, t); else F.orig[V][LI](s, pitch, b, l, t);                                                                     \
    }                                                                                                                                                    \
    static void rec_hbd_##VN##_##LEN(uint16_t *s, int32_t pitch, const uint8_t *b, const uint8_t *l, const uint8_t *t, int32_t bd) {                     \
        if (T.on) record(s, V, LEN, b, l, t); else F.orig_hbd[V][LI](s, pitch, b, l, t, bd);                                                             \
    }
REC(0, horizontal, 0, 4) REC(0, horizontal, 1, 6) REC(0, horizontal, 2, 8) REC(0, horizontal, 3, 14)
REC(1, vertical, 0, 4) REC(1, vertical, 1, 6) REC(1, vertical, 2, 8) REC(1, vertical, 3, 14)

static void dlf_seam_stats(void) {
    const char *f = getenv("SVT_HIP_DLF_SEAM_STATS");
    FILE       *o = f ? fopen(f, "w") : NULL;
    if (!o) return;
    fprintf(o, "ms_in_stage_calls %llu\n", (unsigned long long)(F.us_stage / 1000));
    fprintf(o, "pictures_filtered %llu\nsegments %llu\npictures_filtered_from_sb_records %llu\nsb_calls_recorded %llu\nplanes_declined %llu\n", (unsigned long long)F.n_pictures,
            (unsigned long long)F.n_segments, (unsigned long long)F.n_sb_pictures, (unsigned long long)F.n_sb_calls, (unsigned long long)F.n_replayed);
    fclose(o);
}
static void dlf_seam_init(void) {
    const char *e = getenv("SVT_HIP_DLF_SEAM");
    if (!e || !atoi(e) || !getenv("SVT_HIP")) return;
    *(void **)&F.plane_host = dlsym(RTLD_DEFAULT, "svt_hip_lpf_plane_host");
    if (!F.plane_host) { fprintf(stderr, "SVT_HIP_DLF_SEAM: libsvtav1_hip is not loaded\n"); abort(); }
#define SWAP(V, VN, LI, LEN)                                                                               \
    F.orig[V][LI] = svt_aom_lpf_##VN##_##LEN; svt_aom_lpf_##VN##_##LEN = rec_##VN##_##LEN;                  \
    F.orig_hbd[V][LI] = svt_aom_highbd_lpf_##VN##_##LEN; svt_aom_highbd_lpf_##VN##_##LEN = rec_hbd_##VN##_##LEN;
    SWAP(0, horizontal, 0, 4) SWAP(0, horizontal, 1, 6) SWAP(0, horizontal, 2, 8) SWAP(0, horizontal, 3, 14)
    SWAP(1, vertical, 0, 4) SWAP(1, vertical, 1, 6) SWAP(1, vertical, 2, 8) SWAP(1, vertical, 3, 14)
    atexit(dlf_seam_stats);
    fprintf(stderr, "SVT_HIP_DLF_SEAM: the deblocking filter of a picture runs as one device call per plane\n");
    F.mode = 1;
}
static int dlf_seam_on(void) {
    static pthread_once_t once = PTHREAD_ONCE_INIT;
    pthread_once(&once, dlf_seam_init);
    return F.mode;
}

/* geometry of the picture's three planes for the recorders (both seams) */
static void set_planes(EbPictureBufferDesc *fb, const PictureControlSet *pcs, uint32_t *w, uint32_t *h) {
    const bool is_16bit = pcs->scs->is_16bit_pipeline;
    T.px = is_16bit ? 2 : 1;
    const uint32_t strides[3] = {fb->stride_y, fb->stride_cb, fb->stride_cr};
    uint8_t       *bufs[3]    = {fb->buffer_y, fb->buffer_cb, fb->buffer_cr};
    for (int pl = 0; pl < 3; pl++) {
        const int ss = pl > 0;
        const uint32_t ox = fb->org_x >> ss, oy = fb->org_y >> ss;
        w[pl] = (fb->width + ss) >> ss; h[pl] = (fb->height + ss) >> ss;
        T.stride[pl] = strides[pl]; T.rows[pl] = ((h[pl] + 7) & ~7u) + 8; /* (segments may start up to the 8-aligned height) */
        T.base[pl]   = bufs[pl] + ((size_t)oy * strides[pl] + ox) * T.px;
        T.list[pl][0].n = T.list[pl][1].n = 0;
    }
}
/* one device call per plane over the given lists */
int svt_hip_seam_bind(unsigned long long picture_number); /* integration/enc_handle_binding.c: SVT_HIP_DEVICES sharding */
void svt_aom_get_recon_pic(PictureControlSet *pcs, EbPictureBufferDesc **recon_ptr, bool is_highbd);
static uint64_t filter_planes(const PictureControlSet *pcs, const uint32_t *w, EdgeList (*list)[2]) {
    svt_hip_seam_bind(pcs->picture_number);
    const bool is_16bit = pcs->scs->is_16bit_pipeline;
    const int  bd = is_16bit ? (int)pcs->scs->static_config.encoder_bit_depth : 8;
    uint64_t   segs = 0;
    for (int pl = 0; pl < 3; pl++) {
        const uint32_t nv = list[pl][1].n, nh = list[pl][0].n;
        if (!(nv + nh)) continue;
        uint32_t rows = 0; /* upload only the rows the segments reach */
        for (uint32_t i = 0; i < nv; i++) { const uint32_t r = list[pl][1].e[i].y + 4; rows = r > rows ? r : rows; }
        for (uint32_t i = 0; i < nh; i++) { const uint32_t r = list[pl][0].e[i].y + 8; rows = r > rows ? r : rows; }
        const double t0_ = seam_ms_now();
        const int rc = F.plane_host((void *)T.base[pl], (uint32_t)T.stride[pl], ((w[pl] + 7) & ~7u), rows, is_16bit, bd, list[pl][1].e, nv, list[pl][0].e, nh);
        __atomic_fetch_add(&F.us_stage, (unsigned long long)((seam_ms_now() - t0_) * 1e3), __ATOMIC_RELAXED);
        if (rc) { /* the device path is off (the host form writes the plane last: nothing has changed): the recorded segments through the reference's OWN edge filters --
                   * the pointers the recorders replaced --, every vertical one in the recorded order, then every horizontal one: the order of the device stage */
            static const int li_of[15] = {0, 0, 0, 0, 0, 0, 1, 0, 2, 0, 0, 0, 0, 0, 3};
            /* The reference's SIMD edge filters LOAD 8 or 16 bytes from each threshold pointer and expect the value replicated (_mm_loadl_epi64 / _mm_loadu_si128 on
             * _blimit, ASM_SSE2/dlf_intrin_sse2.c:273,605,634 -- LoopFilterThresh keeps mblim / lim / hev_thr as 16-byte arrays, deblocking_filter.h): the single bytes
             * packed inside an SvtHipLpfEdge are not that.  Each replayed call gets three aligned 16-byte arrays filled with the edge's values. */
            uint8_t bl[16] __attribute__((aligned(16))), li[16] __attribute__((aligned(16))), th[16] __attribute__((aligned(16)));
            for (int v = 1; v >= 0; v--)
                for (uint32_t i = 0; i < list[pl][v].n; i++) {
                    const SvtHipLpfEdge *e = &list[pl][v].e[i];
                    uint8_t *sp = (uint8_t *)T.base[pl] + ((size_t)e->y * T.stride[pl] + e->x) * T.px;
                    memset(bl, e->blimit, 16); memset(li, e->limit, 16); memset(th, e->thresh, 16);
                    if (is_16bit) F.orig_hbd[v][li_of[e->length]]((uint16_t *)sp, (int32_t)T.stride[pl], bl, li, th, bd);
                    else F.orig[v][li_of[e->length]](sp, (int32_t)T.stride[pl], bl, li, th);
                }
            __atomic_fetch_add(&F.n_replayed, 1, __ATOMIC_RELAXED);
        }
        segs += nv + nh;
    }
    return segs;
}

/* ---- SB-based deblocking (coding_loop.c:2297 through integration/coding_loop_seam.c): per-picture segment lists filled SB by SB from the EncDec threads ---- */
enum { SB_RECS = 64 };
typedef struct SbPicture {
    PictureControlSet *pcs;
    uint64_t           picture_number;
    int                live;
    uint32_t           calls;
    EdgeList           list[3][2];
} SbPicture;
static SbPicture sb_rec[SB_RECS];

static void svt_hip_seam_loop_filter_sb_body(EbPictureBufferDesc *fb, PictureControlSet *pcs, int32_t mi_row, int32_t mi_col, int32_t plane_start, int32_t plane_end, uint8_t last_col) {
    if (!dlf_seam_on()) { svt_aom_loop_filter_sb(fb, pcs, mi_row, mi_col, plane_start, plane_end, last_col); return; }
    uint32_t w[3], h[3];
    set_planes(fb, pcs, w, h);
    T.on = 1;
    svt_aom_loop_filter_sb(fb, pcs, mi_row, mi_col, plane_start, plane_end, last_col); /* the reference's per-SB driver, recording */
    T.on = 0;
    pthread_mutex_lock(&F.lock);
    SbPicture *R = NULL, *spare = NULL;
    for (int i = 0; i < SB_RECS; i++) {
        if (sb_rec[i].live && sb_rec[i].pcs == pcs && sb_rec[i].picture_number == pcs->picture_number) { R = &sb_rec[i]; break; }
        if (!sb_rec[i].live && !spare) spare = &sb_rec[i];
    }
    if (!R) {
        if (!spare) { fprintf(stderr, "SVT_HIP_DLF_SEAM: more than %d pictures between the coding loop and the deblocking process\n", SB_RECS); abort(); }
        R = spare; R->live = 1; R->pcs = pcs; R->picture_number = pcs->picture_number; R->calls = 0;
        for (int pl = 0; pl < 3; pl++) R->list[pl][0].n = R->list[pl][1].n = 0;
    }
    for (int pl = 0; pl < 3; pl++)
        for (int v = 0; v < 2; v++) {
            const EdgeList *src = &T.list[pl][v];
            EdgeList       *dst = &R->list[pl][v];
            if (!src->n) continue;
            if (dst->n + src->n > dst->cap) { dst->cap = 2 * (dst->n + src->n) + 4096; dst->e = realloc(dst->e, (size_t)dst->cap * sizeof(*dst->e)); }
            memcpy(dst->e + dst->n, src->e, (size_t)src->n * sizeof(*src->e));
            dst->n += src->n;
        }
    R->calls++;
    F.n_sb_calls++;
    pthread_mutex_unlock(&F.lock);
}
void svt_hip_seam_loop_filter_sb(EbPictureBufferDesc *fb, PictureControlSet *pcs, int32_t mi_row, int32_t mi_col, int32_t plane_start, int32_t plane_end, uint8_t last_col) {
    SEAM_CPU_BEGIN();
    svt_hip_seam_loop_filter_sb_body(fb, pcs, mi_row, mi_col, plane_start, plane_end, last_col);
    SEAM_CPU_END(SEAM_CPU_DLF);
}

/* the picture has reached the deblocking process: apply what its SBs recorded (nothing when the picture was not deblocked SB by SB) */
static void flush_sb_picture(PictureControlSet *pcs) {
    if (!dlf_seam_on()) return;
    pthread_mutex_lock(&F.lock);
    SbPicture *R = NULL;
    for (int i = 0; i < SB_RECS; i++)
        if (sb_rec[i].live && sb_rec[i].pcs == pcs && sb_rec[i].picture_number == pcs->picture_number) { R = &sb_rec[i]; break; }
    pthread_mutex_unlock(&F.lock);
    if (!R) return; /* (every SB of the picture has passed the coding loop: nobody appends to the record any more) */
    EbPictureBufferDesc *fb;
    svt_aom_get_recon_pic(pcs, &fb, pcs->scs->is_16bit_pipeline);
    uint32_t w[3], h[3];
    set_planes(fb, pcs, w, h);
    const uint64_t segs = filter_planes(pcs, w, R->list);
    pthread_mutex_lock(&F.lock);
    R->live = 0;
    F.n_pictures++; F.n_sb_pictures++; F.n_segments += segs;
    pthread_mutex_unlock(&F.lock);
}

static void seam_loop_filter_frame_body(EbPictureBufferDesc *fb, PictureControlSet *pcs, int32_t plane_start, int32_t plane_end) {
    if (!dlf_seam_on()) { svt_av1_loop_filter_frame(fb, pcs, plane_start, plane_end); return; }
    uint32_t w[3], h[3];
    set_planes(fb, pcs, w, h);
    T.on = 1;
    svt_av1_loop_filter_frame(fb, pcs, plane_start, plane_end); /* the reference's driver, recording */
    T.on = 0;
    const uint64_t segs = filter_planes(pcs, w, T.list);
    pthread_mutex_lock(&F.lock);
    F.n_pictures++; F.n_segments += segs;
    pthread_mutex_unlock(&F.lock);
}
static void seam_loop_filter_frame(EbPictureBufferDesc *fb, PictureControlSet *pcs, int32_t plane_start, int32_t plane_end) {
    SEAM_CPU_BEGIN();
    seam_loop_filter_frame_body(fb, pcs, plane_start, plane_end);
    SEAM_CPU_END(SEAM_CPU_DLF);
}


/* svt_aom_get_recon_pic's uses in dlf_process.c, in file order: the two prototypes (:23, :28), the 8-bit -> 16-bit conversion (:90, :91), the frame filter (:108) and
 * the pre-CDEF preparation (:136) -- the point every picture passes after its deblocking and before anything reads the result */
static void get_recon_use2(PictureControlSet *pcs, EbPictureBufferDesc **r, bool hbd) { svt_aom_get_recon_pic(pcs, r, hbd); }
static void get_recon_use3(PictureControlSet *pcs, EbPictureBufferDesc **r, bool hbd) { svt_aom_get_recon_pic(pcs, r, hbd); }
static void get_recon_use4(PictureControlSet *pcs, EbPictureBufferDesc **r, bool hbd) { svt_aom_get_recon_pic(pcs, r, hbd); }
static void get_recon_use5(PictureControlSet *pcs, EbPictureBufferDesc **r, bool hbd) {
    flush_sb_picture(pcs);
    svt_aom_get_recon_pic(pcs, r, hbd);
}
#define SEAM_CAT_(a, b) a##b
#define SEAM_CAT(a, b) SEAM_CAT_(a, b)
#define svt_aom_get_recon_pic(...) SEAM_CAT(get_recon_use, __COUNTER__)(__VA_ARGS__)
#define svt_av1_loop_filter_frame(a, b, c, d) seam_loop_filter_frame(a, b, c, d)
#include "dlf_process.c" /* resolves through -I$(REF)/Source/Lib/Codec */
