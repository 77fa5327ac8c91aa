/* rest_process_seam.c -- REFERENCE-SIDE BINDING (what a maintainer of the reference adds; built into the reference encoder by oracle/Makefile for the identity / fps runs): the reference's restoration process with the per-picture batching seam of INTEGRATION.md §3.
 *
 * This translation unit IS Source/Lib/Codec/rest_process.c of the reference (included below where it lies; nothing is copied).  The one change: the call
 *
 *     restoration_seg_search(context_ptr->rst_tmpbuf, &org_fts, &cpi_source, &trial_frame_rst, pcs, cdef_results->segment_index);      (rest_process.c:612)
 *
 * is given a macro name for the duration of the #include and lands in seam_restoration_seg_search() below.  With SVT_HIP_LR_SEAM unset (or the HIP library not
 * loaded) that function IS the reference call.  With SVT_HIP_LR_SEAM=1 the first segment of a picture to arrive runs the per-unit half of the search for ALL
 * units of every searched plane on the device -- one svt_hip_lr_search_plane_host() per plane, parameters taken field by field from cm->wn_filter_ctrls /
 * cm->sg_filter_ctrls as restoration_seg_search and the functions below it read them (restoration_pick.c:1205-1527) -- and writes what search_norestore_seg /
 * search_wiener_seg / search_sgrproj_seg leave behind: pcs->rusi_picture[plane][unit].sse / .wiener / .sgrproj, cm->sg_frame_ep_cnt[], the extended borders of
 * the searched planes and pcs->rest_extend_flag[].  The other segments of the picture find the work done.  rest_finish_search (the serial rate decisions) and the
 * frame filtering that follow are untouched reference code.  SVT_HIP_LR_SEAM_STATS=<file> receives the counters at exit.
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
#include "restoration.h"
#include "svtav1_hip.h" /* include/svtav1_hip.h of this repository: the C ABI */

void restoration_seg_search(int32_t *rst_tmpbuf, Yv12BufferConfig *org_fts, const Yv12BufferConfig *src, Yv12BufferConfig *trial_frame_rst,
                            PictureControlSet *pcs, uint32_t segment_index);
void svt_av1_loop_restoration_filter_frame(int32_t *rst_tmpbuf, Yv12BufferConfig *frame, Av1Common *cm, int32_t optimized_lr);

int svt_hip_seam_bind(unsigned long long picture_number); /* integration/enc_handle_binding.c: SVT_HIP_DEVICES sharding */
#include <time.h>
static double seam_ms_now(void) { struct timespec t_; clock_gettime(CLOCK_MONOTONIC, &t_); return 1e3 * (double)t_.tv_sec + 1e-6 * (double)t_.tv_nsec; }
static struct {
    pthread_mutex_t lock;
    int             mode; /* 0 off, 1 on */
    int (*search_host)(const SvtHipLrSearchParams *, const SvtHipLrPrevUnit *, SvtHipLrSearchUnit *);
    int (*filter_host)(const SvtHipLrParams *); /* (non-zero from either: the device path is off -- svtav1_hip.h, error policy) */
    unsigned long long us_stage; /* microseconds inside the stage calls */
    unsigned long long us_first, us_max, n_calls; /* the first stage call of the encode, the longest one, how many */
    PictureControlSet *done_pcs[64]; /* pictures whose search has been done by the seam (keyed by pcs + picture number) ... */
    uint64_t           done_num[64];
    uint32_t           seen[64];     /* ... and how many of their segments have passed: the record is dropped with the last one */
    uint64_t           n_pictures, n_planes, n_units, n_declined, n_filtered_planes;
} L = {PTHREAD_MUTEX_INITIALIZER};

static void lr_seam_stats(void) {
    const char *f = getenv("SVT_HIP_LR_SEAM_STATS");
    FILE       *o = f ? fopen(f, "w") : NULL;
    if (!o) return;
    fprintf(o, "ms_in_stage_calls %llu\nus_first_search_call %llu\nus_longest_search_call %llu\nsearch_calls %llu\n", (unsigned long long)(L.us_stage / 1000), L.us_first, L.us_max, L.n_calls);
    fprintf(o, "pictures_offloaded %llu\nplanes_searched %llu\nunits_searched %llu\npictures_declined %llu\nplanes_filtered %llu\n", (unsigned long long)L.n_pictures,
            (unsigned long long)L.n_planes, (unsigned long long)L.n_units, (unsigned long long)L.n_declined, (unsigned long long)L.n_filtered_planes);
    fclose(o);
}
static void lr_seam_init(void) {
    const char *e = getenv("SVT_HIP_LR_SEAM");
    if (!e || !atoi(e) || !getenv("SVT_HIP")) return;
    *(void **)&L.search_host = dlsym(RTLD_DEFAULT, "svt_hip_lr_search_plane_host");
    *(void **)&L.filter_host = dlsym(RTLD_DEFAULT, "svt_hip_lr_filter_frame_host");
    if (!L.search_host || !L.filter_host) { fprintf(stderr, "SVT_HIP_LR_SEAM: libsvtav1_hip is not loaded\n"); abort(); }
    atexit(lr_seam_stats);
    fprintf(stderr, "SVT_HIP_LR_SEAM: the loop-restoration unit search runs as one device stage per plane\n");
    L.mode = 1;
}
static int lr_seam_on(void) {
    static pthread_once_t once = PTHREAD_ONCE_INIT;
    pthread_once(&once, lr_seam_init);
    return L.mode;
}

/* one plane: restoration_seg_search's body for `plane` (restoration_pick.c:1470-1525) with the three unit loops on the device */
/* -> 0, or the stage's return code: nothing of the picture's search state has been touched then */
static int search_plane(const Yv12BufferConfig *org_fts, const Yv12BufferConfig *src, PictureControlSet *pcs, int plane, SvtHipLrSearchUnit **out_p, SvtHipLrSearchParams *P_out) {
    Av1Common *const cm     = pcs->ppcs->av1_cm;
    const int        is_uv  = plane > 0, highbd = cm->use_highbitdepth;
    const int        w = src->crop_widths[is_uv], h = src->crop_heights[is_uv];
    uint8_t         *dgd8 = org_fts->buffers[plane];
    if (!pcs->rest_extend_flag[plane]) { /* :1480-1496, same arguments */
        const int32_t align16_pad = (w % 16) ? 16 - (w % 16) : 0;
        svt_extend_frame(dgd8, w, h, org_fts->strides[is_uv], RESTORATION_BORDER + 1 + align16_pad, RESTORATION_BORDER, highbd);
        pcs->rest_extend_flag[plane] = true;
    }
    const WnFilterCtrls *wn = &cm->wn_filter_ctrls;
    const SgFilterCtrls *sg = &cm->sg_filter_ctrls;
    SvtHipLrSearchParams P;
    memset(&P, 0, sizeof(P));
    P.dgd = highbd ? (const void *)CONVERT_TO_SHORTPTR(dgd8) : (const void *)dgd8;
    P.src = highbd ? (const void *)CONVERT_TO_SHORTPTR(src->buffers[plane]) : (const void *)src->buffers[plane];
    P.dgd_stride = (uint32_t)org_fts->strides[is_uv]; P.src_stride = (uint32_t)src->strides[is_uv];
    P.width = (uint32_t)w; P.height = (uint32_t)h;
    P.unit_size = (uint32_t)pcs->rst_info[plane].restoration_unit_size;
    P.ss_y = (uint8_t)(is_uv && cm->subsampling_y); P.highbd = (uint8_t)highbd; P.bit_depth = (uint8_t)cm->bit_depth;
    P.wn_enabled = wn->enabled && (!plane || wn->use_chroma);
    const int wn_luma = wn->filter_tap_lvl == 1 ? WIENER_WIN : (wn->filter_tap_lvl == 2 ? WIENER_WIN_CHROMA : WIENER_WIN_3TAP); /* :1286-1290 */
    P.wiener_win = (uint8_t)(plane == AOM_PLANE_Y ? wn_luma : (wn_luma < WIENER_WIN_CHROMA ? wn_luma : WIENER_WIN_CHROMA));
    P.wn_use_refinement = wn->use_refinement; P.wn_max_one_refinement_step = wn->max_one_refinement_step;
    P.sg_enabled = sg->enabled && (!plane || sg->use_chroma);
    if (!P.sg_enabled) { /* (the controls of a disabled tool are not initialised by the reference: whatever the structure held) */
    } else if (sg->step_range < 16) { /* the reference-frame based range of search_selfguided_restoration (:560-572) */
        const int8_t *e = cm->sg_ref_frame_ep, step = sg->step_range;
        const int     none = e[0] < 0 && e[1] < 0, mid = none ? 0 : (e[1] < 0 ? e[0] : (e[0] < 0 ? e[1] : (e[0] + e[1]) / 2));
        P.sg_start_ep = (uint8_t)(none ? 0 : AOMMAX(0, mid - step)); P.sg_end_ep = (uint8_t)(none ? SGRPROJ_PARAMS : AOMMIN(SGRPROJ_PARAMS, mid + step));
        P.sg_ep_inc = 1; P.sg_refine = 1;
    } else {
        P.sg_start_ep = (uint8_t)sg->start_ep[is_uv]; P.sg_end_ep = (uint8_t)sg->end_ep[is_uv]; P.sg_ep_inc = (uint8_t)sg->ep_inc[is_uv]; P.sg_refine = (uint8_t)sg->refine[is_uv];
    }
    const int           n    = pcs->rst_info[plane].units_per_tile;
    SvtHipLrSearchUnit *out  = calloc((size_t)n, sizeof(*out));
    SvtHipLrPrevUnit   *prev = NULL;
    const FrameType     ft   = pcs->ppcs->frm_hdr.frame_type;
    if (P.wn_enabled && wn->use_prev_frame_coeffs && ft != KEY_FRAME && ft != INTRA_ONLY_FRAME) { /* :1297-1302 */
        prev = calloc((size_t)n, sizeof(*prev));
        for (int u = 0; u < n; u++)
            if (pcs->rst_info[plane].unit_info[u].restoration_type == RESTORE_WIENER) {
                prev[u].use = 1;
                memcpy(prev[u].vfilter, pcs->rst_info[plane].unit_info[u].wiener_info.vfilter, 16);
                memcpy(prev[u].hfilter, pcs->rst_info[plane].unit_info[u].wiener_info.hfilter, 16);
            }
    }
    svt_hip_seam_bind(pcs->picture_number);
    const double ts_ = seam_ms_now();
    const int rc = L.search_host(&P, prev, out);
    { const unsigned long long us_ = (unsigned long long)((seam_ms_now() - ts_) * 1e3);
      __atomic_fetch_add(&L.us_stage, us_, __ATOMIC_RELAXED);
      if (__atomic_fetch_add(&L.n_calls, 1, __ATOMIC_RELAXED) == 0) L.us_first = us_;
      if (us_ > L.us_max) L.us_max = us_; }
    free(prev);
    if (rc) { /* the device path is off (or the stage refused): the caller runs the reference's search for the whole picture */
        fprintf(stderr, "SVT_HIP_LR_SEAM: svt_hip_lr_search_plane_host returned %d (picture %llu plane %d %ux%u unit %u win %u bd %u wn %d sg %d ep %u..%u/%u): the reference's search takes the picture\n", rc,
                (unsigned long long)pcs->picture_number, plane, P.width, P.height, P.unit_size, P.wiener_win, P.bit_depth, P.wn_enabled, P.sg_enabled, P.sg_start_ep, P.sg_end_ep,
                P.sg_ep_inc);
        free(out);
        return rc;
    }
    *out_p = out; *P_out = P;
    return 0;
}
/* the results of one plane into the picture's search state (restoration_seg_search's stores) */
static void commit_plane(PictureControlSet *pcs, int plane, const SvtHipLrSearchParams *Pp, SvtHipLrSearchUnit *out) {
    Av1Common *const cm = pcs->ppcs->av1_cm;
    const SvtHipLrSearchParams P = *Pp;
    const int n = pcs->rst_info[plane].units_per_tile;
    RestUnitSearchInfo *rusi = pcs->rusi_picture[plane];
    for (int u = 0; u < n; u++) {
        rusi[u].sse[RESTORE_NONE] = out[u].sse[0];
        if (P.wn_enabled) {
            rusi[u].sse[RESTORE_WIENER] = out[u].sse[1];
            if (out[u].sse[1] != INT64_MAX) { memcpy(rusi[u].wiener.vfilter, out[u].vfilter, 16); memcpy(rusi[u].wiener.hfilter, out[u].hfilter, 16); }
        }
        if (P.sg_enabled) {
            rusi[u].sse[RESTORE_SGRPROJ] = out[u].sse[2];
            rusi[u].sgrproj.ep = out[u].ep; rusi[u].sgrproj.xqd[0] = out[u].xqd[0]; rusi[u].sgrproj.xqd[1] = out[u].xqd[1];
            cm->sg_frame_ep_cnt[out[u].ep]++; /* :1239-1241 (this thread is the only one searching the picture) */
        }
    }
    L.n_planes++; L.n_units += (uint64_t)n;
    free(out);
}

static void seam_restoration_seg_search_body(int32_t *rst_tmpbuf, Yv12BufferConfig *org_fts, const Yv12BufferConfig *src, Yv12BufferConfig *trial_frame_rst,
                                        PictureControlSet *pcs, uint32_t segment_index) {
    if (!lr_seam_on()) { restoration_seg_search(rst_tmpbuf, org_fts, src, trial_frame_rst, pcs, segment_index); return; }
    pthread_mutex_lock(&L.lock); /* (one picture at a time; the other segments of this picture wait here and then find it done) */
    int slot = -1, free_slot = -1;
    for (int i = 0; i < 64; i++) {
        if (L.done_pcs[i] == pcs && L.done_num[i] == pcs->picture_number) slot = i;
        if (!L.done_pcs[i] && free_slot < 0) free_slot = i;
    }
    if (slot < 0) {
        Av1Common *const cm        = pcs->ppcs->av1_cm;
        const int32_t    plane_end = ((cm->wn_filter_ctrls.enabled && cm->wn_filter_ctrls.use_chroma) || (cm->sg_filter_ctrls.enabled && cm->sg_filter_ctrls.use_chroma))
               ? AOM_PLANE_V : AOM_PLANE_Y; /* :1462-1466 */
        SvtHipLrSearchUnit  *outs[3] = {NULL, NULL, NULL};
        SvtHipLrSearchParams Ps[3];
        int                  rc = 0;
        for (int32_t plane = AOM_PLANE_Y; plane <= plane_end && !rc; ++plane) rc = search_plane(org_fts, src, pcs, plane, &outs[plane], &Ps[plane]);
        if (!rc) {
            for (int32_t plane = AOM_PLANE_Y; plane <= plane_end; ++plane) commit_plane(pcs, plane, &Ps[plane], outs[plane]);
            L.n_pictures++;
        } else { /* nothing committed: every segment of the picture through the reference's own function, here and now (the other segments find it done) */
            for (int32_t plane = AOM_PLANE_Y; plane <= plane_end; ++plane) free(outs[plane]);
            for (uint32_t sg = 0; sg < pcs->rest_segments_total_count; sg++) restoration_seg_search(rst_tmpbuf, org_fts, src, trial_frame_rst, pcs, sg);
            L.n_declined++;
        }
        if (free_slot < 0) { fprintf(stderr, "SVT_HIP_LR_SEAM: more than 64 pictures in the restoration stage\n"); abort(); }
        slot = free_slot;
        L.done_pcs[slot] = pcs; L.done_num[slot] = pcs->picture_number; L.seen[slot] = 0;
    }
    if (++L.seen[slot] == pcs->rest_segments_total_count) L.done_pcs[slot] = NULL; /* every segment has passed */
    pthread_mutex_unlock(&L.lock);
}
static void seam_restoration_seg_search(int32_t *rst_tmpbuf, Yv12BufferConfig *org_fts, const Yv12BufferConfig *src, Yv12BufferConfig *trial_frame_rst, PictureControlSet *pcs, uint32_t segment_index) {
    seam_test_delay();
    SEAM_CPU_BEGIN();
    seam_restoration_seg_search_body(rst_tmpbuf, org_fts, src, trial_frame_rst, pcs, segment_index);
    SEAM_CPU_END(SEAM_CPU_LR);
}


/* svt_av1_loop_restoration_filter_frame (restoration.c:1179-1247) with the unit loop of every restored plane as one device launch: the same border
 * extension, the units of rst_info[plane].unit_info, the saved deblocked boundary lines of rsi->boundaries; the filtered plane replaces the frame's plane
 * (the reference filters into cm->rst_frame and copies it back, :1241). */
static void seam_loop_restoration_filter_frame_body(int32_t *rst_tmpbuf, Yv12BufferConfig *frame, Av1Common *cm, int32_t optimized_lr) {
    if (!lr_seam_on() || optimized_lr) { svt_av1_loop_restoration_filter_frame(rst_tmpbuf, frame, cm, optimized_lr); return; }
    const int32_t highbd = cm->use_highbitdepth;
    for (int32_t plane = 0; plane < 3; ++plane) {
        RestorationInfo *rsi = &cm->child_pcs->rst_info[plane];
        rsi->optimized_lr    = optimized_lr;
        if (rsi->frame_restoration_type == RESTORE_NONE) continue;
        const int32_t is_uv = plane > 0, w = frame->crop_widths[is_uv], h = frame->crop_heights[is_uv];
        svt_extend_frame(frame->buffers[plane], w, h, frame->strides[is_uv], RESTORATION_BORDER, RESTORATION_BORDER, highbd);
        const int     n = rsi->units_per_tile;
        SvtHipLrUnit *units = calloc((size_t)n, sizeof(*units));
        for (int u = 0; u < n; u++) {
            const RestorationUnitInfo *ri = &rsi->unit_info[u];
            units[u].rtype = (int32_t)ri->restoration_type;
            memcpy(units[u].vfilter, ri->wiener_info.vfilter, 16); memcpy(units[u].hfilter, ri->wiener_info.hfilter, 16);
            units[u].ep = ri->sgrproj_info.ep; units[u].xqd[0] = ri->sgrproj_info.xqd[0]; units[u].xqd[1] = ri->sgrproj_info.xqd[1];
        }
        SvtHipLrParams P;
        memset(&P, 0, sizeof(P));
        void *data = highbd ? (void *)CONVERT_TO_SHORTPTR(frame->buffers[plane]) : (void *)frame->buffers[plane];
        P.data = data; P.dst = data;
        /* buffer column c of the saved lines holds frame column c - RESTORATION_EXTRA_HORZ (restoration.c:296-300) */
        P.boundary_above = rsi->boundaries.stripe_boundary_above + (RESTORATION_EXTRA_HORZ << highbd);
        P.boundary_below = rsi->boundaries.stripe_boundary_below + (RESTORATION_EXTRA_HORZ << highbd);
        P.stride = P.dst_stride = (uint32_t)frame->strides[is_uv]; P.boundary_stride = (uint32_t)rsi->boundaries.stripe_boundary_stride;
        P.width = (uint32_t)w; P.height = (uint32_t)h; P.unit_size = (uint32_t)rsi->restoration_unit_size;
        P.ss_x = (uint8_t)(is_uv && cm->subsampling_x); P.ss_y = (uint8_t)(is_uv && cm->subsampling_y); P.highbd = (uint8_t)highbd; P.bit_depth = (uint8_t)cm->bit_depth;
        P.units = units;
        svt_hip_seam_bind(cm->child_pcs->picture_number);
        const double tf_ = seam_ms_now();
        const int frc = L.filter_host(&P); /* (in place: the plane is written after the call's last device operation -- a failed call has changed nothing) */
        __atomic_fetch_add(&L.us_stage, (unsigned long long)((seam_ms_now() - tf_) * 1e3), __ATOMIC_RELAXED);
        free(units);
        if (frc) { /* the device path is off: the reference's frame function for this plane and the ones after it -- it skips planes of type RESTORE_NONE, which is what
                    * the planes already filtered above are told to be for the duration of the call */
            RestorationType keep[3];
            for (int32_t q = 0; q < 3; q++) { keep[q] = cm->child_pcs->rst_info[q].frame_restoration_type; if (q < plane) cm->child_pcs->rst_info[q].frame_restoration_type = RESTORE_NONE; }
            svt_av1_loop_restoration_filter_frame(rst_tmpbuf, frame, cm, optimized_lr);
            for (int32_t q = 0; q < 3; q++) cm->child_pcs->rst_info[q].frame_restoration_type = keep[q];
            pthread_mutex_lock(&L.lock); L.n_declined++; pthread_mutex_unlock(&L.lock);
            return;
        }
        pthread_mutex_lock(&L.lock);
        L.n_filtered_planes++;
        pthread_mutex_unlock(&L.lock);
    }
}
static void seam_loop_restoration_filter_frame(int32_t *rst_tmpbuf, Yv12BufferConfig *frame, Av1Common *cm, int32_t optimized_lr) {
    seam_test_delay();
    SEAM_CPU_BEGIN();
    seam_loop_restoration_filter_frame_body(rst_tmpbuf, frame, cm, optimized_lr);
    SEAM_CPU_END(SEAM_CPU_LR);
}


#define restoration_seg_search(a, b, c, d, e, f) seam_restoration_seg_search(a, b, c, d, e, f)
#define svt_av1_loop_restoration_filter_frame(a, b, c, d) seam_loop_restoration_filter_frame(a, b, c, d)
#include "rest_process.c" /* resolves through -I$(REF)/Source/Lib/Codec */
