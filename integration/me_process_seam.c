/* me_process_seam.c -- REFERENCE-SIDE BINDING (what a maintainer of the reference adds; built into the reference encoder by oracle/Makefile for the identity / fps runs): the reference's open-loop ME process with the per-picture batching seam of
 * INTEGRATION.md §3.
 *
 * This translation unit IS Source/Lib/Codec/me_process.c of the reference (included below where it lies; nothing is copied).  The one
 * change: inside svt_aom_motion_estimation_kernel's 64x64 loop (me_process.c:174-266) the call
 *
 *     svt_aom_motion_estimation_b64(pcs, b64_index, b64_origin_x, b64_origin_y, me_ctx, input_pic);
 *
 * is given a macro name for the duration of the #include and lands in seam_motion_estimation_b64() below.  With SVT_HIP_ME_SEAM unset (or
 * the HIP library not loaded) that function IS the reference call.  With SVT_HIP_ME_SEAM=1 the first SB of a picture to arrive hands the
 * WHOLE picture to the device stage -- one svt_hip_me_session_submit_stage() per picture, parameters filled field by field from the
 * MeContext that the reference's own svt_aom_sig_deriv_me() (enc_mode_config.c:681) has just derived for this picture -- and every SB
 * (this one and the ones the other segments / threads bring) copies its slice of the result into pcs->pa_me_data->me_results[b64_index]
 * and the per-SB statistics arrays, which is everything svt_aom_motion_estimation_b64 leaves behind for a PAME task
 * (motion_estimation.c:3076-3152).  Segments, the processed-SB counter, global motion, open-loop intra search: untouched reference code.
 *
 * A picture whose settings the device stage does not cover (super-resolution / resize re-ME) is DECLINED as a whole and runs the reference's C code
 * (the low-delay level-0 resizing, enable_me_sr_adjustment == 2 and level-0-only HME are covered since round 3); declines are counted and the
 * identity tests require zero of them for the configurations they claim.  SVT_HIP_ME_SEAM_STATS=<file> receives the counters at exit.
 */
#define _GNU_SOURCE /* RTLD_DEFAULT */
#include "../integration/seam_cpu.h"
#include <dlfcn.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "motion_estimation.h" /* declares svt_aom_motion_estimation_b64 before the macro below exists */
#include "me_context.h"
#include "pcs.h"
#include "sequence_control_set.h"
#include "reference_object.h"
#include "svtav1_hip.h" /* include/svtav1_hip.h of this repository: the C ABI */

static EbErrorType seam_motion_estimation_b64(PictureParentControlSet *pcs, uint32_t b64_index, uint32_t b64_origin_x, uint32_t b64_origin_y,
                                              MeContext *me_ctx, EbPictureBufferDesc *input_ptr);

/* ---- the C ABI, resolved from the library enc_handle_binding.c has dlopen()ed with RTLD_GLOBAL ---- */
static struct {
    void *(*create)(uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t);
    void *(*create_on)(int, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t);
    int (*enable_stage)(void *, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t);
    int (*submit_stage)(void *, int64_t, const uint8_t *, const int64_t *, uint32_t, const SvtHipMeStageParams *, const SvtHipMeResultsHost *);
    int (*wait)(void *, int); /* non-zero: the device path is off (svtav1_hip.h, error policy) -- the picture is declined and runs the reference's ME */
    void (*invalidate)(void *, int64_t);
    int (*resident)(void *, int64_t);
    void *(*host_alloc)(size_t);
} abi;

enum { SEAM_RING = 24, SEAM_MAX_REFS = 7, SEAM_RECS = 64, SEAM_DEVS = 16, SEAM_SLOTS = 4 }; /* SEAM_SLOTS: submissions in flight per device session (eight changed neither the time under the per-device lock nor the fps: profiles/r04_call12_*) */
/* integration/enc_handle_binding.c: SVT_HIP_DEVICES=<d0,d1,...> shards pictures over GPUs by picture number; one resident session (ring, checksums, lock) per index */
int svt_hip_seam_bind(unsigned long long picture_number);
int svt_hip_seam_device_count(void);
int svt_hip_seam_device_id(int index);
typedef struct SeamPicture { /* results of one picture, consumed SB by SB */
    PictureParentControlSet *pcs;
    uint64_t                 picture_number;
    int                      state; /* 0 free, 1 being computed, 2 ready, 3 declined (run the reference) */
    uint32_t                 n_sb, n_pus, max_refs, max_cand, consumed;
    uint8_t                 *total, *cand;
    uint32_t                *mv;
    SvtHipMeSbStats         *stats;
    size_t                   cap_total, cap_cand, cap_mv, cap_stats;
} SeamPicture;
static struct {
    pthread_mutex_t lock;  /* the record tables (rec[], tf_rec[]) and the counters: held for table look-ups only, never across device work */
    pthread_cond_t  ready;
    pthread_mutex_t dev[SEAM_DEVS]; /* the device session (picture ring, checksum table): held while residency is decided and a stage is ENQUEUED, not while it runs --
                            * several pictures are in flight on the device at once (session slots), their owners wait outside both locks */
    int             mode; /* -1 unknown, 0 off, 1 on */
    void           *session[SEAM_DEVS];
    int             session_failed[SEAM_DEVS]; /* the session could not be made (the device path is off): pictures of this device are declined */
    uint32_t        width, height, stride, org_x, org_y, rows;
    SeamPicture     rec[SEAM_RECS];
    uint64_t        sum[SEAM_DEVS][SEAM_RING * 2][2]; /* per device: (picture id, plane checksum) of what is resident */
    uint64_t        n_per_dev[SEAM_DEVS];
    uint64_t        n_pictures, n_declined, n_sb, n_uploads, n_reuploads, n_registered;
    double          t_stage, t_first; /* seconds in run_picture / run_tf_pair (all threads), in the first stage call (session creation, kernel code loading): under `lock` */
    uint64_t        ns_hash, ns_dev_lock; /* hashing planes, holding a device lock -- added to from under DIFFERENT device locks (or none): relaxed atomics, like n_uploads /
                                           * n_reuploads (ThreadSanitizer, profiles/r05_tsan_seams.txt) */
    char            why[128];
} G = {PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, {PTHREAD_MUTEX_INITIALIZER, PTHREAD_MUTEX_INITIALIZER, PTHREAD_MUTEX_INITIALIZER, PTHREAD_MUTEX_INITIALIZER,
      PTHREAD_MUTEX_INITIALIZER, PTHREAD_MUTEX_INITIALIZER, PTHREAD_MUTEX_INITIALIZER, PTHREAD_MUTEX_INITIALIZER, PTHREAD_MUTEX_INITIALIZER, PTHREAD_MUTEX_INITIALIZER,
      PTHREAD_MUTEX_INITIALIZER, PTHREAD_MUTEX_INITIALIZER, PTHREAD_MUTEX_INITIALIZER, PTHREAD_MUTEX_INITIALIZER, PTHREAD_MUTEX_INITIALIZER, PTHREAD_MUTEX_INITIALIZER}, -1};

static int      seam_hash_on(void) { SEAM_ENV_ONCE(on, (getenv("SVT_HIP_ME_SEAM_HASH") && atoi(getenv("SVT_HIP_ME_SEAM_HASH")))); return on; } /* SVT_HIP_ME_SEAM_HASH=1 */
static uint64_t n_invalidated;
static uint64_t tf_pairs, tf_sb, tf_declined; /* the temporal filter's (picture, reference) pairs through the stage, see the end of this file */
static void seam_stats(void) {
    const char *f = getenv("SVT_HIP_ME_SEAM_STATS");
    FILE       *o = f ? fopen(f, "w") : NULL;
    if (!o) return;
    fprintf(o, "picture_buffers_page_locked %llu\n", (unsigned long long)G.n_registered);
    fprintf(o, "pictures_offloaded %llu\npictures_declined %llu\nsb_results %llu\nplane_uploads %llu\nplane_reuploads_by_checksum %llu\nlast_decline %s\n",
            (unsigned long long)G.n_pictures, (unsigned long long)G.n_declined, (unsigned long long)G.n_sb, (unsigned long long)G.n_uploads,
            (unsigned long long)G.n_reuploads, G.why[0] ? G.why : "-");
    fprintf(o, "tf_pairs_offloaded %llu\ntf_sb_results %llu\ntf_pairs_declined %llu\n", (unsigned long long)tf_pairs, (unsigned long long)tf_sb, (unsigned long long)tf_declined);
    for (int k = 0; k < svt_hip_seam_device_count() && svt_hip_seam_device_count() > 1; k++)
        fprintf(o, "stage_calls_on_device_%d %llu\n", svt_hip_seam_device_id(k), (unsigned long long)G.n_per_dev[k]);
    fprintf(o, "planes_invalidated_after_temporal_filtering %llu\n", (unsigned long long)n_invalidated);
    fprintf(o, "ms_in_stage_calls %llu\nms_first_stage_call %llu\nms_holding_device_lock %llu\n", (unsigned long long)(G.t_stage * 1e3),
            (unsigned long long)(G.t_first * 1e3), (unsigned long long)(G.ns_dev_lock / 1000000));
    if (seam_hash_on()) fprintf(o, "ms_hashing_planes %llu\n", (unsigned long long)(G.ns_hash / 1000000));
    fclose(o);
}
static void seam_init(void) { /* once (pthread_once): ME threads arriving during the initialisation wait instead of seeing "off" */
    __atomic_store_n(&G.mode, 0, __ATOMIC_RELEASE); /* (also read by svt_hip_seam_me_invalidate, which does not pass through the once) */
    const char *e = getenv("SVT_HIP_ME_SEAM");
    if (!e || !atoi(e) || !getenv("SVT_HIP")) return;
#define SYM(field, name) *(void **)&abi.field = dlsym(RTLD_DEFAULT, name)
    SYM(create, "svt_hip_me_session_create"); SYM(create_on, "svt_hip_me_session_create_on"); SYM(enable_stage, "svt_hip_me_session_enable_stage"); SYM(submit_stage, "svt_hip_me_session_submit_stage");
    SYM(wait, "svt_hip_me_session_wait"); SYM(invalidate, "svt_hip_me_session_invalidate"); SYM(resident, "svt_hip_me_session_resident");
    SYM(host_alloc, "svt_hip_host_alloc");
#undef SYM
    if (!abi.create || !abi.enable_stage || !abi.submit_stage || !abi.wait || !abi.invalidate || !abi.resident) {
        fprintf(stderr, "SVT_HIP_ME_SEAM: libsvtav1_hip is not loaded\n");
        abort();
    }
    atexit(seam_stats);
    fprintf(stderr, "SVT_HIP_ME_SEAM: open-loop ME runs as one device stage per picture\n");
    __atomic_store_n(&G.mode, 1, __ATOMIC_RELEASE);
}
static int seam_on(void) {
    static pthread_once_t once = PTHREAD_ONCE_INIT;
    pthread_once(&once, seam_init);
    return G.mode;
}

static double seam_now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }
static uint64_t plane_sum_(const EbPictureBufferDesc *p);
/* Residency is decided by EXPLICIT invalidation: the one writer that changes a picture after it may have been uploaded is its own temporal filtering, and the
 * temporal-filter seam reports it (svt_hip_seam_me_invalidate below, called after every produce_temporally_filtered_pic[_ld] call, offloaded or not).  The sampled
 * checksum of round 3 (every 8th row: right only with high probability, 7 ms of hashing per 60 frames) remains as a debugging aid: SVT_HIP_ME_SEAM_HASH=1 compares
 * on top and counts what the explicit rule missed (`plane_reuploads_by_checksum`, expected 0). */
static uint64_t plane_sum(const EbPictureBufferDesc *p) { /* called OUTSIDE the locks */
    if (!seam_hash_on()) return 2; /* (constant: the comparison with the stored value never asks for an upload) */
    const double   t0 = seam_now();
    const uint64_t h  = plane_sum_(p);
    const double   dt = seam_now() - t0;
    __atomic_fetch_add(&G.ns_hash, (uint64_t)(dt * 1e9), __ATOMIC_RELAXED);
    return h;
}
/* content check of a picture's luma: every 8th row of the visible samples (the padding is a function of them).  The only thing that rewrites a picture after it may
 * have been uploaded is its own temporal filtering, which changes the whole picture, so a sparse sample detects it; hashing all rows cost 0.6 ms per plane at 1080p
 * -- a third of the seam's host time per picture (profiles/r03_reg1_bench_default.json: ms_hashing_planes). */
static uint64_t plane_sum_(const EbPictureBufferDesc *p) {
    uint64_t h = 1469598103934665603ull;
    for (uint32_t y = 0; y < p->height; y += 8) {
        const uint8_t *r = p->buffer_y + (size_t)(p->org_y + y) * p->stride_y + p->org_x;
        uint64_t       a = 0;
        for (uint32_t x = 0; x + 8 <= p->width; x += 8) { uint64_t v; memcpy(&v, r + x, 8); a = a * 1099511628211ull + v; }
        h = (h ^ a) * 1099511628211ull;
    }
    return h;
}
/* the device-side identity of a picture's luma planes: an overlay picture carries the picture number of the alt-ref it overlays (and names that alt-ref as its own
 * reference, pd_process.c:4163) with different samples -- the unfiltered source -- so it gets an id of its own */
#define SEAM_ID(pcs) ((uint64_t)(pcs)->picture_number * 2 + ((pcs)->is_overlay ? 1 : 0))
/* the picture's host planes were rewritten (temporal filtering): whatever copy a device holds is stale */
void svt_hip_seam_me_invalidate(unsigned long long picture_number) {
    if (__atomic_load_n(&G.mode, __ATOMIC_ACQUIRE) != 1) return; /* (not seam_on(): a temporal filter that runs before any ME call has nothing resident to invalidate) */
    for (int di = 0; di < SEAM_DEVS; di++) {
        if (!__atomic_load_n(&G.session[di], __ATOMIC_ACQUIRE)) continue;
        pthread_mutex_lock(&G.dev[di]);
        if (G.session[di] && abi.resident(G.session[di], (int64_t)(picture_number * 2))) {
            abi.invalidate(G.session[di], (int64_t)(picture_number * 2));
            n_invalidated++;
        }
        pthread_mutex_unlock(&G.dev[di]);
    }
}
static int sum_slot(int di, uint64_t id, int make) {
    int free_i = -1;
    for (int i = 0; i < SEAM_RING * 2; i++) {
        if (G.sum[di][i][1] && G.sum[di][i][0] == id) return i;
        if (!G.sum[di][i][1] && free_i < 0) free_i = i;
    }
    if (!make) return -1;
    if (free_i < 0) { /* the table (2 x the ring) is full: drop the entries of pictures that have left the device's ring -- never one of a RESIDENT picture, whose stored
                       * checksum is what SVT_HIP_ME_SEAM_HASH compares against (clearing the whole table, as before, made every resident picture read as
                       * "content changed" once per wrap: 5 false re-uploads in a 60-frame encode, none in the shorter identity clips) */
        for (int i = 0; i < SEAM_RING * 2; i++)
            if (!G.session[di] || !abi.resident(G.session[di], (int64_t)G.sum[di][i][0])) { G.sum[di][i][1] = 0; if (free_i < 0) free_i = i; }
        if (free_i < 0) { memset(G.sum[di], 0, sizeof(G.sum[di])); free_i = 0; } /* (cannot happen: at most SEAM_RING pictures are resident) */
    }
    G.sum[di][free_i][0] = id;
    return free_i;
}
/* make `pic` (picture id `id`) resident with its current content: upload it when it is absent or its host content changed since the upload */
/* (device lock held; `now` = the plane's checksum, computed by the caller outside the lock.)  The copy of a REFERENCE must be complete before the caller returns to
 * the encoder: another encoder thread may rewrite its host plane later (a picture is temporally filtered in place after it served as a neighbour's reference).  It is
 * not waited for HERE any more (that kept the per-device lock for 0.1-0.3 ms per upload, ~1.3 uploads per picture, while other pictures queued behind it): the upload's
 * slot goes to `pend`, and the caller waits for it after it has released the lock -- the stage it submits is ordered behind the upload by the session's events. */
static int ensure_resident(int di, uint64_t id, const EbPictureBufferDesc *pic, const SvtHipMeStageParams *S, uint64_t now, int *pend, int *n_pend) {
    const int k = sum_slot(di, id, 1);
    if (abi.resident(G.session[di], (int64_t)id)) {
        if (!seam_hash_on() || G.sum[di][k][1] == now) return 0; /* (without the checksum aid a resident picture is current: invalidation is explicit) */
        abi.invalidate(G.session[di], (int64_t)id); /* e.g. temporally filtered in place after it was uploaded */
        __atomic_fetch_add(&G.n_reuploads, 1, __ATOMIC_RELAXED);
    }
    const int slot = abi.submit_stage(G.session[di], (int64_t)id, pic->buffer_y, NULL, 0, S, NULL);
    if (slot < 0) return slot;
    if (*n_pend < 16) pend[(*n_pend)++] = slot;
    else abi.wait(G.session[di], slot);
    G.sum[di][k][1] = now;
    __atomic_fetch_add(&G.n_uploads, 1, __ATOMIC_RELAXED);
    return 0;
}

static int decline(const char *why) {
    snprintf(G.why, sizeof(G.why), "%s", why);
    return -1;
}
/* SvtHipMeStageParams from the MeContext svt_aom_sig_deriv_me filled + the per-picture set-up of me_process.c:236-262 (same field names) */
static int fill_stage(PictureParentControlSet *pcs, MeContext *c, SvtHipMeStageParams *S, int64_t *ref_ids, const EbPictureBufferDesc **ref_pics,
                      uint32_t *n_refs_out, int tf /* the temporal filter's call: me_type == ME_MCTF, one reference taken from the context */) {
    SequenceControlSet *scs = pcs->scs;
    memset(S, 0, sizeof(*S));
    if (pcs->frame_superres_enabled || pcs->frame_resize_enabled) return decline("super-resolution / resize");
    if (!c->enable_hme_flag || !c->enable_hme_level0_flag || (!c->enable_hme_level1_flag && c->enable_hme_level2_flag)) return decline("HME without level 0, or level 2 without level 1");
    if (c->num_hme_sa_w * c->num_hme_sa_h > 4) return decline("more than 2 x 2 HME regions");
    const uint32_t n0 = c->num_of_ref_pic_to_search[0], n1 = c->num_of_list_to_search > 1 ? c->num_of_ref_pic_to_search[1] : 0, n = n0 + n1;
    if (n == 0 || n > SEAM_MAX_REFS || n0 > 4 || n1 > 4) return decline("reference count");
    S->num_hme_sa_w = (uint8_t)c->num_hme_sa_w; S->num_hme_sa_h = (uint8_t)c->num_hme_sa_h;
    S->hme_sub_sampled = c->hme_search_method != FULL_SAD_SEARCH;
    S->me_sub_sad      = c->me_search_method == SUB_SAD_SEARCH;
    S->hme_levels      = c->enable_hme_level2_flag ? 3 : (c->enable_hme_level1_flag ? 2 : 1); /* (1: the temporal filter at tf_ctrls.hme_me_level 3 / 4, enc_mode_config.c:1655-1661) */
    S->hme_sa_width[1] = (int16_t)c->hme_l1_sa.width; S->hme_sa_height[1] = (int16_t)c->hme_l1_sa.height;
    S->hme_sa_width[2] = (int16_t)c->hme_l2_sa.width; S->hme_sa_height[2] = (int16_t)c->hme_l2_sa.height;
    S->me_sa_min_width = (int16_t)c->me_sa.sa_min.width; S->me_sa_min_height = (int16_t)c->me_sa.sa_min.height;
    S->me_sa_max_width = (int16_t)c->me_sa.sa_max.width; S->me_sa_max_height = (int16_t)c->me_sa.sa_max.height;
    S->mv_adj_enabled = c->mv_based_sa_adj.enabled; S->mv_adj_nearest_ref_only = c->mv_based_sa_adj.nearest_ref_only;
    S->mv_adj_mv_size_th = c->mv_based_sa_adj.mv_size_th; S->mv_adj_sa_multiplier = c->mv_based_sa_adj.sa_multiplier;
    const MeHmeRefPruneCtrls *pr = &c->me_hme_prune_ctrls;
    const MeSrCtrls          *sr = &c->me_sr_adjustment_ctrls;
    /* hme_prune_ref_and_adjust_sr runs when prune_ref = enable_hme_flag && me_type != ME_MCTF (motion_estimation.c:3103, :3115-3117) */
    S->hme_prune_enabled = pr->enable_me_hme_ref_pruning && pr->prune_ref_if_hme_sad_dev_bigger_than_th != (uint16_t)~0;
    S->prune_ref_if_hme_sad_dev_bigger_than_th = pr->prune_ref_if_hme_sad_dev_bigger_than_th;
    S->sr_adjustment = sr->enable_me_sr_adjustment;
    S->reduce_me_sr_based_on_mv_length_th = sr->reduce_me_sr_based_on_mv_length_th; S->stationary_hme_sad_abs_th = sr->stationary_hme_sad_abs_th;
    S->stationary_me_sr_divisor = sr->stationary_me_sr_divisor; S->reduce_me_sr_based_on_hme_sad_abs_th = sr->reduce_me_sr_based_on_hme_sad_abs_th;
    S->me_sr_divisor_for_low_hme_sad = sr->me_sr_divisor_for_low_hme_sad;
    S->me_early_exit_th = c->me_early_exit_th;
    S->is_ref = c->is_ref;
    S->me_8x8_var_enabled = c->me_8x8_var_ctrls.enabled; S->me_sr_div4_th = c->me_8x8_var_ctrls.me_sr_div4_th;
    S->me_sr_div2_th = c->me_8x8_var_ctrls.me_sr_div2_th; S->me_sr_mult2_th = c->me_8x8_var_ctrls.me_sr_mult2_th;
    S->temporal_layer_gt0 = c->temporal_layer_index > 0;
    S->prehme_enabled = c->prehme_ctrl.enable; S->prehme_skip_search_line = c->prehme_ctrl.skip_search_line; S->prehme_l1_early_exit = c->prehme_ctrl.l1_early_exit;
    for (int k = 0; k < 2; k++) {
        S->prehme_sa_min_width[k] = c->prehme_ctrl.prehme_sa_cfg[k].sa_min.width; S->prehme_sa_min_height[k] = c->prehme_ctrl.prehme_sa_cfg[k].sa_min.height;
        S->prehme_sa_max_width[k] = c->prehme_ctrl.prehme_sa_cfg[k].sa_max.width; S->prehme_sa_max_height[k] = c->prehme_ctrl.prehme_sa_cfg[k].sa_max.height;
    }
    S->zz_sad_th = pr->zz_sad_th; S->zz_sad_pct = (uint16_t)pr->zz_sad_pct; S->phme_sad_th = pr->phme_sad_th; S->phme_sad_pct = (uint16_t)pr->phme_sad_pct;
    S->prev_me_stage_based_exit_th = c->prev_me_stage_based_exit_th;
    /* init_zz_sad's picture-level conditions (motion_estimation.c:2419-2421) */
    S->me_safe_limit_zz_th = (c->me_safe_limit_zz_th && pcs->hierarchical_levels > 0 && c->num_of_list_to_search == 2 &&
                              pcs->temporal_layer_index >= pcs->hierarchical_levels && pcs->similar_brightness_refs) ? c->me_safe_limit_zz_th : 0;
    /* per reference slot, list 0 first: picture distance through svt_aom_get_scaled_picture_distance (:1239-1243, :1300-1302) and the level-0 search area of
     * get_hme_l0_search_area (:1806-1866; distance-based resizing in its non-RTC form divides the base area by 1 + ref index, hme_level0_b64 restores it) */
    S->hme_l0_per_ref = 1;
    uint32_t k = 0;
    for (uint32_t li = 0; li < c->num_of_list_to_search; li++)
        for (uint32_t ri = 0; ri < c->num_of_ref_pic_to_search[li]; ri++, k++) {
            /* open-loop ME: the PA reference objects of the picture; temporal filter: the one reference temporal_filtering.c:3159-3168 put into the context */
            const uint64_t             ref_number = tf ? c->me_ds_ref_array[li][ri].picture_number
                                                       : ((EbPaReferenceObject *)pcs->ref_pa_pic_ptr_array[li][ri]->object_ptr)->picture_number;
            const EbPictureBufferDesc *ref_padded = tf ? c->me_ds_ref_array[li][ri].picture_ptr
                                                       : ((EbPaReferenceObject *)pcs->ref_pa_pic_ptr_array[li][ri]->object_ptr)->input_padded_pic;
            ref_ids[k] = (int64_t)(ref_number * 2); ref_pics[k] = ref_padded; /* (references are never overlay pictures: SEAM_ID below) */
            const int64_t  d64  = (int64_t)pcs->picture_number - (int64_t)ref_number;
            const uint16_t dist = (uint16_t)(int16_t)(d64 < 0 ? -d64 : d64), f = (uint16_t)((dist * 5) / 8 + ((dist % 8) ? 1 : 0));
            S->dist[k] = tf ? dist : f; S->ref_pic_index[k] = (uint8_t)ri; /* (ME_MCTF: the integer search takes the distance unscaled, :1300-1302) */
            SearchAreaMinMax a = c->hme_l0_sa;
            if (sr->enable_me_sr_adjustment && sr->distance_based_hme_resizing) {
                a.sa_min.width /= 1 + ri; a.sa_min.height /= 1 + ri; a.sa_max.width /= 1 + ri; a.sa_max.height /= 1 + ri;
            }
            int w = a.sa_min.width / c->num_hme_sa_w, h = a.sa_min.height / c->num_hme_sa_h;
            const int wmax = ((a.sa_max.width / c->num_hme_sa_w) + 15) & ~15, hmax = a.sa_max.height / c->num_hme_sa_h;
            w = ((w * f) + 15) & ~15; h = h * f;
            S->hme_l0_sa_width_ref[k] = (int16_t)(w < wmax ? w : wmax); S->hme_l0_sa_height_ref[k] = (int16_t)(h < hmax ? h : hmax);
            if (sr->enable_me_sr_adjustment && sr->distance_based_hme_resizing && c->reduce_hme_l0_sr_th_min && c->reduce_hme_l0_sr_th_max) {
                /* the low-delay settings (enc_mode_config.c:702-714): per SB the divisor is (1 + index) or (2 + index), decided on the device from list 0 / reference 0's
                 * level-0 motion (get_hme_l0_search_area :1809-1850); here the areas of the second form */
                SearchAreaMinMax a2 = c->hme_l0_sa;
                a2.sa_min.width /= 2 + ri; a2.sa_min.height /= 2 + ri; a2.sa_max.width /= 2 + ri; a2.sa_max.height /= 2 + ri;
                int w2 = a2.sa_min.width / c->num_hme_sa_w, h2 = a2.sa_min.height / c->num_hme_sa_h;
                const int w2max = ((a2.sa_max.width / c->num_hme_sa_w) + 15) & ~15, h2max = a2.sa_max.height / c->num_hme_sa_h;
                w2 = ((w2 * f) + 15) & ~15; h2 = h2 * f;
                S->hme_l0_sa_width_ref2[k] = (int16_t)(w2 < w2max ? w2 : w2max); S->hme_l0_sa_height_ref2[k] = (int16_t)(h2 < h2max ? h2 : h2max);
                S->reduce_hme_l0_sr_th_min = c->reduce_hme_l0_sr_th_min; S->reduce_hme_l0_sr_th_max = c->reduce_hme_l0_sr_th_max;
                if (sr->enable_me_sr_adjustment == 2) { /* the screen-content levels: (4 + index) where list 0's motion is small on both axes (:1836-1841) */
                    SearchAreaMinMax a4 = c->hme_l0_sa;
                    a4.sa_min.width /= 4 + ri; a4.sa_min.height /= 4 + ri; a4.sa_max.width /= 4 + ri; a4.sa_max.height /= 4 + ri;
                    int w4 = a4.sa_min.width / c->num_hme_sa_w, h4 = a4.sa_min.height / c->num_hme_sa_h;
                    const int w4max = ((a4.sa_max.width / c->num_hme_sa_w) + 15) & ~15, h4max = a4.sa_max.height / c->num_hme_sa_h;
                    w4 = ((w4 * f) + 15) & ~15; h4 = h4 * f;
                    S->hme_l0_sa_width_ref4[k] = (int16_t)(w4 < w4max ? w4 : w4max); S->hme_l0_sa_height_ref4[k] = (int16_t)(h4 < h4max ? h4 : h4max);
                }
            }
            S->results.ref_picture_number[li][ri] = ref_number;
        }
    SvtHipMeResultsParams *R = &S->results;
    R->num_of_list_to_search = c->num_of_list_to_search; R->num_of_ref_pic_to_search[0] = (uint8_t)n0; R->num_of_ref_pic_to_search[1] = (uint8_t)n1;
    if (tf) { /* raw tables only: no candidate formatting */
        S->me_type_mctf = 1; S->tf_me_exit_th = c->tf_me_exit_th;
        R->max_cand = 1; R->max_refs = 1; R->max_l0 = 1;
    } else {
        R->max_cand = pcs->pa_me_data->max_cand; R->max_refs = pcs->pa_me_data->max_refs; R->max_l0 = pcs->pa_me_data->max_l0;
    }
    R->enable_me_16x16 = pcs->enable_me_16x16; R->enable_me_8x8 = pcs->enable_me_8x8;
    R->only_l_bwd = scs->mrp_ctrls.only_l_bwd;
    R->use_best_unipred_cand_only = c->use_best_unipred_cand_only;
    R->prune_ref = pr->enable_me_hme_ref_pruning && pr->prune_ref_if_me_sad_dev_bigger_than_th != (uint16_t)~0; /* me_prune_ref's second half (:1545-1563) */
    R->prune_ref_if_me_sad_dev_bigger_than_th = pr->prune_ref_if_me_sad_dev_bigger_than_th;
    R->low_resolution = scs->input_resolution <= INPUT_SIZE_480p_RANGE;
    R->gm_enabled = pcs->gm_ctrls.enabled; R->gm_use_distance_based_active_th = pcs->gm_ctrls.use_distance_based_active_th;
    R->prune_me_candidates_th = c->prune_me_candidates_th;
    R->picture_number = pcs->picture_number;
    *n_refs_out = n;
    return 0;
}

static void reserve(void **p, size_t *cap, size_t bytes) {
    if (bytes <= *cap) return;
    *p = abi.host_alloc ? abi.host_alloc(bytes) : malloc(bytes); /* (the previous, smaller buffer is left to the process: sizes settle after the first picture) */
    *cap = bytes;
}

/* the session (picture ring + stage buffers) is sized once for the encode, from the first picture that reaches a seam */
static void create_session(int di, const EbPaReferenceObject *pa) {
    const EbPictureBufferDesc *src = pa->input_padded_pic;
    G.width = src->width; G.height = src->height; G.stride = src->stride_y; G.org_x = src->org_x; G.org_y = src->org_y;
    G.rows = src->luma_size / src->stride_y;
    /* largest ME area any preset derives is 256 x 256 (x 2 by the MV-based adjustment, x 3 / 2 by the variance probe) */
    const int dev_id = svt_hip_seam_device_id(di); /* -1: no sharding, the default device */
    G.session[di] = dev_id >= 0 && abi.create_on ? abi.create_on(dev_id, G.width, G.height, G.stride, G.org_x, G.org_y, G.rows, SEAM_RING, SEAM_MAX_REFS, 768, 768, SEAM_SLOTS)
                                                 : abi.create(G.width, G.height, G.stride, G.org_x, G.org_y, G.rows, SEAM_RING, SEAM_MAX_REFS, 768, 768, SEAM_SLOTS); /* four pictures in flight */
    if (!G.session[di] || abi.enable_stage(G.session[di], pa->quarter_downsampled_picture_ptr->org_x, pa->sixteenth_downsampled_picture_ptr->org_x, 4, 768, 768)) {
        fprintf(stderr, "SVT_HIP_ME_SEAM: cannot create the ME session (svt_hip_last_error): every picture of this device takes the reference's ME\n");
        G.session[di] = NULL; /* (a half-made session is left to the process; ensure_session declines from here on) */
        G.session_failed[di] = 1;
    }
}
/* Called by the binding at the end of svt_av1_enc_init (integration/enc_handle_binding.c) with an object of the encoder's PA-reference pool: the sessions of all
 * devices -- ring, stage buffers, pinned slots: ~40 ms -- are created while the encoder initialises instead of inside the first picture's stage call. */
void svt_hip_seam_me_prepare(const void *pa_reference_object) {
    if (!seam_on() || !pa_reference_object) return;
    const EbPaReferenceObject *pa = (const EbPaReferenceObject *)pa_reference_object;
    if (!pa->input_padded_pic || !pa->quarter_downsampled_picture_ptr || !pa->sixteenth_downsampled_picture_ptr) return;
    for (int di = 0; di < svt_hip_seam_device_count() && di < SEAM_DEVS; di++) {
        pthread_mutex_lock(&G.dev[di]);
        if (!G.session[di] && !G.session_failed[di]) {
            if (svt_hip_seam_device_count() > 1) svt_hip_seam_bind((unsigned long long)di); /* (binds this thread to device di: picture numbers di, di + N, ... map to it) */
            create_session(di, pa);
        }
        pthread_mutex_unlock(&G.dev[di]);
    }
}
/* Also at init, for every picture of the encoder's 8-bit luma pool (input_y8b_buffer_resource_ptr: the planes pa_ref->input_padded_pic->buffer_y points at,
 * resource_coordination_process.c:1130 -- what every ME stage call uploads): the buffer is page-locked, so that the upload is one DMA instead of a staged copy
 * from pageable memory (svt_hip_host_register; a refusal leaves the buffer as it is). */
void svt_hip_seam_me_register_buffer(void *buffer, size_t bytes) {
    if (!seam_on() || !buffer || !bytes) return;
    static int (*reg)(void *, size_t);
    if (!reg) *(void **)&reg = dlsym(RTLD_DEFAULT, "svt_hip_host_register");
    if (reg && reg(buffer, bytes) == 0) __atomic_fetch_add(&G.n_registered, 1, __ATOMIC_RELAXED);
}
void svt_hip_seam_me_unregister_buffer(void *buffer) { /* before the pool that owns the buffer is destroyed */
    if (!seam_on() || !buffer || !G.n_registered) return;
    static int (*unreg)(void *);
    if (!unreg) *(void **)&unreg = dlsym(RTLD_DEFAULT, "svt_hip_host_unregister");
    if (unreg) unreg(buffer);
}
static int ensure_session(int di, PictureParentControlSet *pcs, const EbPictureBufferDesc *src) {
    if (!G.session[di] && !G.session_failed[di]) create_session(di, (const EbPaReferenceObject *)pcs->pa_ref_pic_wrapper->object_ptr);
    if (!G.session[di]) return decline("no ME session (the device path is off)");
    if (src->width != G.width || src->height != G.height || src->stride_y != G.stride || src->org_x != G.org_x || src->org_y != G.org_y)
        return decline("picture geometry changed");
    return 0;
}

/* the whole picture on the device; called by the thread that brought the picture's first SB, WITHOUT G.lock (the record is in state 1: it belongs to this thread) */
static int run_picture(SeamPicture *P, PictureParentControlSet *pcs, MeContext *c, const EbPictureBufferDesc *src) {
    SvtHipMeStageParams        S;
    int64_t                    ref_ids[8];
    const EbPictureBufferDesc *ref_pics[8];
    uint64_t                   ref_sum[8];
    uint32_t                   n_refs = 0;
    if (fill_stage(pcs, c, &S, ref_ids, ref_pics, &n_refs, 0)) return -1;
    for (uint32_t k = 0; k < n_refs; k++) ref_sum[k] = plane_sum(ref_pics[k]) | 1; /* content checks outside the locks */
    const uint64_t now = plane_sum(src) | 1;
    P->n_sb = pcs->b64_total_count;
    P->n_pus = pcs->enable_me_16x16 ? (pcs->enable_me_8x8 ? 85 : 21) : 5;
    P->max_refs = S.results.max_refs; P->max_cand = S.results.max_cand;
    reserve((void **)&P->total, &P->cap_total, (size_t)P->n_sb * P->n_pus);
    reserve((void **)&P->cand, &P->cap_cand, (size_t)P->n_sb * P->n_pus * P->max_cand);
    reserve((void **)&P->mv, &P->cap_mv, (size_t)P->n_sb * P->n_pus * P->max_refs * 4);
    reserve((void **)&P->stats, &P->cap_stats, (size_t)P->n_sb * sizeof(SvtHipMeSbStats));
    SvtHipMeResultsHost H;
    memset(&H, 0, sizeof(H));
    H.total_me_candidate_index = P->total; H.me_mv_array = P->mv; H.me_candidate_array = P->cand; H.sb_stats = P->stats;
    const int di = svt_hip_seam_bind(pcs->picture_number); /* the device this picture is sharded to (0 without SVT_HIP_DEVICES) */
    pthread_mutex_lock(&G.dev[di]);
    const double td0 = seam_now();
    int rc = ensure_session(di, pcs, src), slot = -1, pend[16], n_pend = 0;
    void *ses = G.session[di];
    for (int pass = 0; !rc; pass++) { /* an upload may evict a reference another upload just brought in (round-robin ring): repeat until all are there */
        int missing = 0;
        for (uint32_t k = 0; k < n_refs && !rc; k++) {
            if (ref_pics[k]->width != G.width || ref_pics[k]->stride_y != G.stride) rc = decline("reference geometry");
            else if (ensure_resident(di, (uint64_t)ref_ids[k], ref_pics[k], &S, ref_sum[k], pend, &n_pend)) rc = decline("reference upload");
        }
        for (uint32_t k = 0; k < n_refs && !rc; k++) missing += !abi.resident(ses, ref_ids[k]);
        if (rc || !missing) break;
        if (pass == 3) rc = decline("ring too small for the reference set");
    }
    if (!rc) {
        /* the source: (re)uploaded when its content differs from what is resident (the same picture may have served as a reference before its own ME) */
        const int ks = sum_slot(di, SEAM_ID(pcs), 1);
        if (seam_hash_on() && abi.resident(ses, (int64_t)SEAM_ID(pcs)) && G.sum[di][ks][1] != now) { abi.invalidate(ses, (int64_t)SEAM_ID(pcs)); __atomic_fetch_add(&G.n_reuploads, 1, __ATOMIC_RELAXED); }
        if (!abi.resident(ses, (int64_t)SEAM_ID(pcs))) __atomic_fetch_add(&G.n_uploads, 1, __ATOMIC_RELAXED);
        G.sum[di][ks][1] = now;
        slot = abi.submit_stage(ses, (int64_t)SEAM_ID(pcs), src->buffer_y, ref_ids, n_refs, &S, &H);
        if (slot < 0) {
            char why[64];
            snprintf(why, sizeof(why), "svt_hip_me_session_submit_stage returned %d", slot);
            rc = decline(why);
        }
    }
    G.n_per_dev[di]++;
    __atomic_fetch_add(&G.ns_dev_lock, (uint64_t)((seam_now() - td0) * 1e9), __ATOMIC_RELAXED);
    pthread_mutex_unlock(&G.dev[di]);
    for (int k = 0; k < n_pend; k++) abi.wait(ses, pend[k]); /* the reference uploads (outside the locks; complete before the stage below is) */
    if (rc) return -1;
    if (abi.wait(ses, slot)) { decline("the device path is off (svt_hip_last_error)"); return -1; } /* outside the locks: other pictures enqueue their stages meanwhile */
    return 0;
}

static EbErrorType seam_motion_estimation_b64_body(PictureParentControlSet *pcs, uint32_t b64_index, uint32_t b64_origin_x, uint32_t b64_origin_y,
                                              MeContext *me_ctx, EbPictureBufferDesc *input_ptr) {
    if (!seam_on() || me_ctx->me_type != ME_OPEN_LOOP)
        return svt_aom_motion_estimation_b64(pcs, b64_index, b64_origin_x, b64_origin_y, me_ctx, input_ptr);
    SEAM_ENV_ONCE(verify, (getenv("SVT_HIP_ME_SEAM_VERIFY") != NULL)); /* SVT_HIP_ME_SEAM_VERIFY=1: run the reference's function for the SB as well and report the first difference (diagnostic) */
    /* The common case -- the picture's stage has run, this SB only fetches its slice -- takes NO lock: 510 SBs x every picture x dozens of ME threads on one mutex
     * cost more host CPU per SB (14 us) than the reference's AVX2 search of the SB (10 us; profiles/r04_call2_*).  A record in state 2 / 3 is immutable until its
     * last SB has fetched (consumed == n_sb), and every SB of the picture comes here exactly once, so a reader that found it cannot lose it. */
    SeamPicture *P = NULL, *spare = NULL;
    for (int i = 0; i < SEAM_RECS && !verify; i++) {
        /* (pcs / picture_number are written by a claimer under G.lock while this path reads them without it: relaxed atomics on both sides, published by the release
         * store of `state` -- no plain-field data race for a compiler or TSan to object to) */
        if (__atomic_load_n(&G.rec[i].state, __ATOMIC_ACQUIRE) >= 2 && __atomic_load_n(&G.rec[i].pcs, __ATOMIC_RELAXED) == pcs &&
            __atomic_load_n(&G.rec[i].picture_number, __ATOMIC_RELAXED) == pcs->picture_number && __atomic_load_n(&G.rec[i].state, __ATOMIC_ACQUIRE) >= 2) { /* (still ready after the comparison: the slot was not freed and claimed anew meanwhile) */
            P = &G.rec[i];
            break;
        }
    }
    if (P) goto fetch;
    pthread_mutex_lock(&G.lock);
    for (int i = 0; i < SEAM_RECS; i++) {
        const int st_ = __atomic_load_n(&G.rec[i].state, __ATOMIC_ACQUIRE); /* (records are freed without the lock by a store of 0) */
        if (st_ && G.rec[i].pcs == pcs && G.rec[i].picture_number == pcs->picture_number) { P = &G.rec[i]; break; }
        if (!st_ && !spare) spare = &G.rec[i];
    }
    if (!P) { /* first SB of this picture: compute everything now */
        if (!spare) { fprintf(stderr, "SVT_HIP_ME_SEAM: more than %d pictures in flight\n", SEAM_RECS); abort(); }
        P = spare;
        __atomic_store_n(&P->pcs, pcs, __ATOMIC_RELAXED); __atomic_store_n(&P->picture_number, pcs->picture_number, __ATOMIC_RELAXED); P->consumed = 0;
        __atomic_store_n(&P->state, 1, __ATOMIC_RELEASE);
        EbPaReferenceObject *pa = (EbPaReferenceObject *)pcs->pa_ref_pic_wrapper->object_ptr;
        pthread_mutex_unlock(&G.lock); /* the record is ours (state 1); the other SBs of this picture wait on the condition, other pictures proceed */
        const double t0 = seam_now();
        const int rc = run_picture(P, pcs, me_ctx, pa->input_padded_pic);
        const double dt = seam_now() - t0;
        pthread_mutex_lock(&G.lock);
        G.t_stage += dt;
        if (G.n_pictures + G.n_declined + tf_pairs + tf_declined == 0) G.t_first = dt;
        if (rc) { P->n_sb = pcs->b64_total_count; G.n_declined++; __atomic_store_n(&P->state, 3, __ATOMIC_RELEASE); }
        else    { G.n_pictures++; __atomic_store_n(&P->state, 2, __ATOMIC_RELEASE); }
        pthread_cond_broadcast(&G.ready);
    }
    while (P->state == 1) pthread_cond_wait(&G.ready, &G.lock);
    if (!verify) pthread_mutex_unlock(&G.lock);
fetch:;
    const int declined = __atomic_load_n(&P->state, __ATOMIC_ACQUIRE) == 3;
    if (!declined && verify) { /* (diagnostic mode: the lock is still held) */
        pthread_mutex_unlock(&G.lock);
        svt_aom_motion_estimation_b64(pcs, b64_index, b64_origin_x, b64_origin_y, me_ctx, input_ptr);
        pthread_mutex_lock(&G.lock);
        const MeSbResults *r = pcs->pa_me_data->me_results[b64_index];
        const uint8_t     *t = P->total + (size_t)b64_index * P->n_pus, *c = P->cand + (size_t)b64_index * P->n_pus * P->max_cand;
        const uint32_t    *m = P->mv + (size_t)b64_index * P->n_pus * P->max_refs;
        const SvtHipMeSbStats *vs = &P->stats[b64_index];
        if (pcs->me_64x64_distortion[b64_index] != vs->me_64x64_distortion || pcs->me_32x32_distortion[b64_index] != vs->me_32x32_distortion ||
            pcs->me_16x16_distortion[b64_index] != vs->me_16x16_distortion || pcs->me_8x8_distortion[b64_index] != vs->me_8x8_distortion ||
            pcs->me_8x8_cost_variance[b64_index] != vs->me_8x8_cost_variance || pcs->rc_me_distortion[b64_index] != vs->rc_me_distortion ||
            pcs->stationary_block_present_sb[b64_index] != vs->stationary_block_present_sb || pcs->rc_me_allow_gm[b64_index] != vs->rc_me_allow_gm)
            fprintf(stderr, "SVT_HIP_ME_SEAM_VERIFY: picture %llu SB %u statistics (reference / device): 64 %u/%u 32 %u/%u 16 %u/%u 8 %u/%u var %u/%u rc %u/%u stat %u/%u gm %u/%u\n",
                    (unsigned long long)pcs->picture_number, b64_index, pcs->me_64x64_distortion[b64_index], vs->me_64x64_distortion, pcs->me_32x32_distortion[b64_index],
                    vs->me_32x32_distortion, pcs->me_16x16_distortion[b64_index], vs->me_16x16_distortion, pcs->me_8x8_distortion[b64_index], vs->me_8x8_distortion,
                    pcs->me_8x8_cost_variance[b64_index], vs->me_8x8_cost_variance, (unsigned)pcs->rc_me_distortion[b64_index], (unsigned)vs->rc_me_distortion,
                    pcs->stationary_block_present_sb[b64_index], vs->stationary_block_present_sb, pcs->rc_me_allow_gm[b64_index], vs->rc_me_allow_gm);
        for (uint32_t pu = 0; pu < P->n_pus; pu++) {
            int bad = r->total_me_candidate_index[pu] != t[pu];
            for (uint32_t k = 0; !bad && k < t[pu]; k++) bad = ((const uint8_t *)r->me_candidate_array)[pu * P->max_cand + k] != c[pu * P->max_cand + k];
            for (uint32_t k = 0; !bad && k < P->max_refs; k++) bad = ((const uint32_t *)r->me_mv_array)[pu * P->max_refs + k] != m[pu * P->max_refs + k] ? 2 : 0;
            if (bad) {
                fprintf(stderr, "SVT_HIP_ME_SEAM_VERIFY: picture %llu SB %u (%u,%u) pu %u: total %u / %u (reference / device)%s;", (unsigned long long)pcs->picture_number,
                        b64_index, b64_origin_x, b64_origin_y, pu, r->total_me_candidate_index[pu], t[pu], bad == 2 ? " MV" : "");
                for (uint32_t k = 0; k < P->max_refs; k++)
                    fprintf(stderr, " %08x/%08x", ((const uint32_t *)r->me_mv_array)[pu * P->max_refs + k], m[pu * P->max_refs + k]);
                fprintf(stderr, "\n");
                break;
            }
        }
    }
    if (!declined) {
        MeSbResults *r = pcs->pa_me_data->me_results[b64_index];
        memcpy(r->total_me_candidate_index, P->total + (size_t)b64_index * P->n_pus, P->n_pus);
        memcpy(r->me_candidate_array, P->cand + (size_t)b64_index * P->n_pus * P->max_cand, (size_t)P->n_pus * P->max_cand);
        memcpy(r->me_mv_array, P->mv + (size_t)b64_index * P->n_pus * P->max_refs, (size_t)P->n_pus * P->max_refs * 4);
        const SvtHipMeSbStats *st = &P->stats[b64_index];
        pcs->me_64x64_distortion[b64_index] = st->me_64x64_distortion; pcs->me_32x32_distortion[b64_index] = st->me_32x32_distortion;
        pcs->me_16x16_distortion[b64_index] = st->me_16x16_distortion; pcs->me_8x8_distortion[b64_index] = st->me_8x8_distortion;
        pcs->me_8x8_cost_variance[b64_index] = st->me_8x8_cost_variance; pcs->rc_me_distortion[b64_index] = st->rc_me_distortion;
        pcs->stationary_block_present_sb[b64_index] = st->stationary_block_present_sb; pcs->rc_me_allow_gm[b64_index] = st->rc_me_allow_gm;
        __atomic_fetch_add(&G.n_sb, 1, __ATOMIC_RELAXED);
    }
    const uint32_t n_sb_rec = P->n_sb; /* (read before the count: the record may be claimed anew the moment the last SB has fetched) */
    if (__atomic_add_fetch(&P->consumed, 1, __ATOMIC_ACQ_REL) == n_sb_rec) __atomic_store_n(&P->state, 0, __ATOMIC_RELEASE); /* every SB has fetched its slice: free again */
    if (verify) pthread_mutex_unlock(&G.lock);
    if (declined) return svt_aom_motion_estimation_b64(pcs, b64_index, b64_origin_x, b64_origin_y, me_ctx, input_ptr);
    return EB_ErrorNone;
}
static EbErrorType seam_motion_estimation_b64(PictureParentControlSet *pcs, uint32_t b64_index, uint32_t b64_origin_x, uint32_t b64_origin_y, MeContext *me_ctx, EbPictureBufferDesc *input_ptr) {
    SEAM_CPU_BEGIN();
    const EbErrorType r_ = seam_motion_estimation_b64_body(pcs, b64_index, b64_origin_x, b64_origin_y, me_ctx, input_ptr);
    SEAM_CPU_END(SEAM_CPU_ME);
    return r_;
}



/* ---- the temporal filter's ME (temporal_filtering.c:3180: svt_aom_motion_estimation_b64 with me_type == ME_MCTF, one reference per call) ---------------------------
 * temporal_filtering.c enters the encoder library through integration/temporal_filtering_seam.c, which renames that one call to the function below.  With
 * SVT_HIP_TF_ME_SEAM=1 the first 64x64 block of a (central picture, reference picture) pair to arrive runs the stage for ALL blocks of the pair in its ME_MCTF
 * form (unscaled distance, tf_me_exit_th, no pruning, raw tables) and every block then receives what the reference's call leaves in the MeContext for the
 * temporal filter: search_results[0][0].hme_sc_x / hme_sc_y / hme_sad, the tf_tot_horz_blks / tf_tot_vert_blks vote (motion_estimation.c:2469-2474), the early
 * exit (tf_use_pred_64x64_only_th = ~0, :3109-3113) or p_sb_best_sad / p_sb_best_mv[0][0][85] with the p_best_* pointers on them (:1372-1395). */
enum { TF_RECS = 32 };
typedef struct SeamTfPair {
    PictureParentControlSet *pcs;
    uint64_t                 picture_number, ref_number;
    int                      state; /* 0 free / complete (the tables stay readable until the slot is reused), 1 being computed, 2 ready, 3 declined */
    uint32_t                 n_sb, consumed;
    uint64_t                 stamp; /* order of creation: the oldest complete record is reused first */
    int                      tables_valid; /* the stage ran for this pair (not declined) */
    uint32_t                *best_sad, *best_mv;
    int16_t                 *hme_sc;
    uint64_t                *hme_sad;
    size_t                   cap_sad, cap_mv, cap_sc, cap_hs;
} SeamTfPair;
static SeamTfPair tf_rec[TF_RECS];
static int        tf_mode = -1;

static int run_tf_pair(SeamTfPair *T, PictureParentControlSet *pcs, MeContext *c) {
    SvtHipMeStageParams        S;
    int64_t                    ref_ids[8];
    const EbPictureBufferDesc *ref_pics[8];
    uint32_t                   n_refs = 0;
    if (c->num_of_list_to_search != 1 || c->num_of_ref_pic_to_search[0] != 1) return decline("temporal filter: more than one reference per call");
    if (fill_stage(pcs, c, &S, ref_ids, ref_pics, &n_refs, 1)) return -1;
    EbPaReferenceObject       *pa  = (EbPaReferenceObject *)pcs->pa_ref_pic_wrapper->object_ptr;
    const EbPictureBufferDesc *src = pa->input_padded_pic; /* same luma as input_picture_ptr_central at this point, in the session's geometry */
    const uint64_t ref_now = plane_sum(ref_pics[0]) | 1, now = plane_sum(src) | 1; /* outside the locks */
    T->n_sb = pcs->b64_total_count;
    reserve((void **)&T->best_sad, &T->cap_sad, (size_t)T->n_sb * 85 * 4);
    reserve((void **)&T->best_mv, &T->cap_mv, (size_t)T->n_sb * 85 * 4);
    reserve((void **)&T->hme_sc, &T->cap_sc, (size_t)T->n_sb * 2 * sizeof(int16_t));
    reserve((void **)&T->hme_sad, &T->cap_hs, (size_t)T->n_sb * sizeof(uint64_t));
    SvtHipMeResultsHost H;
    memset(&H, 0, sizeof(H));
    H.best_sad = T->best_sad; H.best_mv = T->best_mv; H.hme_sc = T->hme_sc; H.hme_sad = T->hme_sad;
    const int di = svt_hip_seam_bind(pcs->picture_number);
    pthread_mutex_lock(&G.dev[di]);
    const double td0 = seam_now();
    int rc = ensure_session(di, pcs, src), slot = -1, pend[16], n_pend = 0;
    void *ses = G.session[di];
    if (!rc && (ref_pics[0]->width != G.width || ref_pics[0]->stride_y != G.stride)) rc = decline("reference geometry");
    for (int pass = 0; !rc; pass++) {
        if (ensure_resident(di, (uint64_t)ref_ids[0], ref_pics[0], &S, ref_now, pend, &n_pend)) rc = decline("reference upload");
        else if (abi.resident(ses, ref_ids[0])) break;
        else if (pass == 3) rc = decline("ring too small for the reference set");
    }
    if (!rc) {
        const int ks = sum_slot(di, SEAM_ID(pcs), 1);
        if (seam_hash_on() && abi.resident(ses, (int64_t)SEAM_ID(pcs)) && G.sum[di][ks][1] != now) { abi.invalidate(ses, (int64_t)SEAM_ID(pcs)); __atomic_fetch_add(&G.n_reuploads, 1, __ATOMIC_RELAXED); }
        if (!abi.resident(ses, (int64_t)SEAM_ID(pcs))) __atomic_fetch_add(&G.n_uploads, 1, __ATOMIC_RELAXED);
        G.sum[di][ks][1] = now;
        slot = abi.submit_stage(ses, (int64_t)SEAM_ID(pcs), src->buffer_y, ref_ids, 1, &S, &H);
        if (slot < 0) rc = decline("svt_hip_me_session_submit_stage (ME_MCTF form) refused the parameters");
    }
    G.n_per_dev[di]++;
    __atomic_fetch_add(&G.ns_dev_lock, (uint64_t)((seam_now() - td0) * 1e9), __ATOMIC_RELAXED);
    pthread_mutex_unlock(&G.dev[di]);
    for (int k = 0; k < n_pend; k++) abi.wait(ses, pend[k]);
    if (rc) return -1;
    if (abi.wait(ses, slot)) { decline("the device path is off (svt_hip_last_error)"); return -1; }
    return 0;
}

EbErrorType svt_hip_seam_tf_motion_estimation_b64(PictureParentControlSet *pcs, uint32_t b64_index, uint32_t b64_origin_x, uint32_t b64_origin_y, MeContext *c,
                                                  EbPictureBufferDesc *input_ptr) {
    if (tf_mode < 0) { const char *e = getenv("SVT_HIP_TF_ME_SEAM"); tf_mode = e && atoi(e); }
    if (!tf_mode || !seam_on() || c->me_type != ME_MCTF) return svt_aom_motion_estimation_b64(pcs, b64_index, b64_origin_x, b64_origin_y, c, input_ptr);
    const uint64_t ref_number = c->me_ds_ref_array[0][0].picture_number;
    pthread_mutex_lock(&G.lock);
    static uint64_t stamp;
    SeamTfPair *T = NULL, *spare = NULL;
    for (int i = 0; i < TF_RECS; i++) {
        if (tf_rec[i].state && tf_rec[i].pcs == pcs && tf_rec[i].picture_number == pcs->picture_number && tf_rec[i].ref_number == ref_number) { T = &tf_rec[i]; break; }
        if (!tf_rec[i].state && (!spare || tf_rec[i].stamp < spare->stamp)) spare = &tf_rec[i];
    }
    if (!T) {
        if (!spare) { fprintf(stderr, "SVT_HIP_TF_ME_SEAM: more than %d (picture, reference) pairs in flight\n", TF_RECS); abort(); }
        T = spare;
        T->pcs = pcs; T->picture_number = pcs->picture_number; T->ref_number = ref_number; T->consumed = 0; T->stamp = ++stamp;
        T->state = 1; T->tables_valid = 0; /* being computed: the record is this thread's, the pair's other blocks wait on the condition */
        pthread_mutex_unlock(&G.lock);
        const double t0 = seam_now();
        const int rc = run_tf_pair(T, pcs, c);
        const double dt = seam_now() - t0;
        pthread_mutex_lock(&G.lock);
        G.t_stage += dt;
        if (G.n_pictures + G.n_declined + tf_pairs + tf_declined == 0) G.t_first = dt;
        if (rc) { T->state = 3; T->tables_valid = 0; T->n_sb = pcs->b64_total_count; tf_declined++; }
        else    { T->state = 2; T->tables_valid = 1; tf_pairs++; }
        pthread_cond_broadcast(&G.ready);
    }
    while (T->state == 1) pthread_cond_wait(&G.ready, &G.lock);
    const int declined = T->state == 3;
    if (!declined) {
        /* what svt_aom_motion_estimation_b64 leaves behind for the temporal filter (see the header of this section) */
        const uint16_t aw = (uint16_t)ALIGN_POWER_OF_TWO(input_ptr->width, 3), ah = (uint16_t)ALIGN_POWER_OF_TWO(input_ptr->height, 3);
        c->b64_width  = (aw - b64_origin_x) < BLOCK_SIZE_64 ? aw - b64_origin_x : BLOCK_SIZE_64;
        c->b64_height = (ah - b64_origin_y) < BLOCK_SIZE_64 ? ah - b64_origin_y : BLOCK_SIZE_64;
        c->search_results[0][0].hme_sc_x = T->hme_sc[2 * b64_index]; c->search_results[0][0].hme_sc_y = T->hme_sc[2 * b64_index + 1];
        c->search_results[0][0].hme_sad  = T->hme_sad[b64_index];
        if (ABS(c->search_results[0][0].hme_sc_x) > ABS(c->search_results[0][0].hme_sc_y)) c->tf_tot_horz_blks++;
        else c->tf_tot_vert_blks++;
        if (c->search_results[0][0].hme_sad < c->tf_me_exit_th) c->tf_use_pred_64x64_only_th = (uint8_t)~0;
        else {
            memcpy(c->p_sb_best_sad[0][0], T->best_sad + (size_t)b64_index * 85, 85 * sizeof(uint32_t));
            memcpy(c->p_sb_best_mv[0][0], T->best_mv + (size_t)b64_index * 85, 85 * sizeof(uint32_t));
            c->p_best_sad_64x64 = &c->p_sb_best_sad[0][0][ME_TIER_ZERO_PU_64x64]; c->p_best_sad_32x32 = &c->p_sb_best_sad[0][0][ME_TIER_ZERO_PU_32x32_0];
            c->p_best_sad_16x16 = &c->p_sb_best_sad[0][0][ME_TIER_ZERO_PU_16x16_0]; c->p_best_sad_8x8 = &c->p_sb_best_sad[0][0][ME_TIER_ZERO_PU_8x8_0];
            c->p_best_mv64x64 = &c->p_sb_best_mv[0][0][ME_TIER_ZERO_PU_64x64]; c->p_best_mv32x32 = &c->p_sb_best_mv[0][0][ME_TIER_ZERO_PU_32x32_0];
            c->p_best_mv16x16 = &c->p_sb_best_mv[0][0][ME_TIER_ZERO_PU_16x16_0]; c->p_best_mv8x8 = &c->p_sb_best_mv[0][0][ME_TIER_ZERO_PU_8x8_0];
        }
        tf_sb++;
    }
    if (++T->consumed == T->n_sb) T->state = 0;
    pthread_mutex_unlock(&G.lock);
    if (declined) return svt_aom_motion_estimation_b64(pcs, b64_index, b64_origin_x, b64_origin_y, c, input_ptr);
    return EB_ErrorNone;
}


/* The whole (central, reference) pair for the temporal filter's driver seam (integration/temporal_filtering_seam.c, SVT_HIP_TF_SEAM): runs the pair's ME stage
 * like the first block's svt_hip_seam_tf_motion_estimation_b64 would (c = the MeContext set up as produce_temporally_filtered_pic does before its ME call,
 * temporal_filtering.c:3140-3177) and hands out all four tables.  1 = tables filled; 0 = the seam is off or the pair is outside the stage (the caller then
 * leaves the picture to the reference). */
int svt_hip_seam_tf_pair_run(PictureParentControlSet *pcs, MeContext *c, uint32_t n_sb, uint32_t *best_sad, uint32_t *best_mv, int16_t *hme_sc, uint64_t *hme_sad) {
    if (tf_mode < 0) { const char *e = getenv("SVT_HIP_TF_ME_SEAM"); tf_mode = e && atoi(e); }
    if (!tf_mode || !seam_on() || c->me_type != ME_MCTF || n_sb != pcs->b64_total_count) return 0;
    static SeamTfPair      mine[TF_RECS]; /* scratch records of this entry point: one per thread that may be inside it */
    static pthread_mutex_t mlock = PTHREAD_MUTEX_INITIALIZER;
    SeamTfPair *T = NULL;
    pthread_mutex_lock(&mlock);
    for (int i = 0; i < TF_RECS && !T; i++)
        if (!mine[i].state) { T = &mine[i]; T->state = 1; }
    pthread_mutex_unlock(&mlock);
    if (!T) return 0;
    const double t0 = seam_now();
    const int rc = run_tf_pair(T, pcs, c);
    const double dt = seam_now() - t0;
    pthread_mutex_lock(&G.lock);
    G.t_stage += dt;
    if (G.n_pictures + G.n_declined + tf_pairs + tf_declined == 0) G.t_first = dt;
    if (rc) tf_declined++; else { tf_pairs++; tf_sb += n_sb; }
    pthread_mutex_unlock(&G.lock);
    if (!rc) {
        memcpy(best_sad, T->best_sad, (size_t)n_sb * 85 * 4); memcpy(best_mv, T->best_mv, (size_t)n_sb * 85 * 4);
        memcpy(hme_sc, T->hme_sc, (size_t)n_sb * 2 * sizeof(int16_t)); memcpy(hme_sad, T->hme_sad, (size_t)n_sb * 8);
    }
    pthread_mutex_lock(&mlock);
    T->state = 0;
    pthread_mutex_unlock(&mlock);
    return !rc;
}

/* the pair's whole-picture tables for the sub-pel seam (integration/temporal_filtering_seam.c); copied out under the lock.  best_mv: [n_sb][85], hme_sc: [n_sb][2],
 * hme_sad: [n_sb].  0 when the pair did not go through the stage. */
int svt_hip_seam_tf_pair_tables(PictureParentControlSet *pcs, uint64_t ref_number, uint32_t n_sb, uint32_t *best_mv, int16_t *hme_sc, uint64_t *hme_sad) {
    int ok = 0;
    pthread_mutex_lock(&G.lock);
    for (int i = 0; i < TF_RECS; i++) {
        const SeamTfPair *T = &tf_rec[i];
        if (T->pcs == pcs && T->picture_number == pcs->picture_number && T->ref_number == ref_number && T->tables_valid && T->n_sb == n_sb) {
            memcpy(best_mv, T->best_mv, (size_t)n_sb * 85 * 4); memcpy(hme_sc, T->hme_sc, (size_t)n_sb * 2 * sizeof(int16_t)); memcpy(hme_sad, T->hme_sad, (size_t)n_sb * 8);
            ok = 1;
            break;
        }
    }
    pthread_mutex_unlock(&G.lock);
    return ok;
}

#define svt_aom_motion_estimation_b64(pcs, i, x, y, ctx, pic) seam_motion_estimation_b64(pcs, i, x, y, ctx, pic)
#include "me_process.c" /* resolves through -I$(REF)/Source/Lib/Codec */
