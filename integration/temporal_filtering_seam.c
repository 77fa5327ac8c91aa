/* temporal_filtering_seam.c -- REFERENCE-SIDE BINDING (what a maintainer of the reference adds; built into the reference encoder by oracle/Makefile for the identity / fps runs): the reference's temporal filter with two seams.
 *
 * This translation unit IS Source/Lib/Codec/temporal_filtering.c of the reference (included below where it lies; nothing is copied).  Two calls are renamed for
 * the duration of the #include:
 *
 * 1. svt_aom_motion_estimation_b64(centre_pcs, ...) (:3180, me_type == ME_MCTF) lands in svt_hip_seam_tf_motion_estimation_b64() (integration/me_process_seam.c,
 *    which owns the device session and the picture ring).  With SVT_HIP_TF_ME_SEAM unset that function IS the reference call.
 *
 * 2. tf_subpel_search(...) -- `static`, defined at :1670 and called from tf_64x64_ / tf_32x32_ / tf_16x16_ / tf_8x8_sub_pel_search (:1886, :1998, :2118, :2237).
 *    The macro appends __COUNTER__ (unused anywhere else in this translation unit): the DEFINITION becomes tf_subpel_search_use0 -- the reference's body,
 *    untouched -- and the four call sites tf_subpel_search_use1 .. _use4, all of which are seam_tf_subpel_search() below.  With SVT_HIP_TF_SUBPEL_SEAM=1 (on top
 *    of the ME seams) the first block of a (central picture, reference picture) pair to arrive has the refinement of EVERY block the reference may ask for --
 *    64x64, 32x32, 16x16 (and 8x8 with tf_ctrls.enable_8x8_pred) of every 64x64 block, starting from the vectors the pair's ME tables hold -- computed by ONE
 *    svt_hip_tf_subpel_search_host() call; a block's search is then a table lookup.  The refinement of a block depends only on the pictures, the block and its
 *    starting vector, not on the 64 / 32 / 16 decisions the reference takes between the searches, which stay the reference's own code.  Every lookup checks
 *    the caller's starting vector and interpolation filter against what the batch assumed and runs the reference's function when they differ (or when the
 *    pair is outside what the batch covers: the high-bit-depth path, a pair that did not go through the ME stage).
 *
 * 3. produce_temporally_filtered_pic(...) -- `static`, defined at :2782, called once from svt_av1_init_temporal_filtering (:4244), per segment of the central
 *    picture.  Same __COUNTER__ renaming (definition = _use5, the call = _use8 = seam_produce_temporally_filtered_pic() below).  The low-delay form
 *    produce_temporally_filtered_pic_ld (:3415, called at :4236 when pred_structure is low delay: no ME, every block predicted from the co-located block of
 *    every frame of the window) takes the same route with its own name (definition = _ld_use6, call = _ld_use7) and the stage's zero_motion form.  With SVT_HIP_TF_SEAM=1 (on top of
 *    the ME seams) the first segment of a central picture to arrive runs the WHOLE picture as one device stage -- svt_hip_tf_picture_host: sub-pel refinement, the
 *    64x64 / 32x32 / 16x16 / 8x8 decisions, final motion compensation, filter -- and the picture's other segments return once it is done.  What stays the
 *    reference's own code: the picture-level decisions (which frames are skipped, :3105-3131), the decay factors (the function's preamble, :2870-3035, executed by
 *    calling the reference's function over an EMPTY block range: SEGMENT_END_IDX is redefined below to collapse the range while a thread-local flag is set), the
 *    set-up of the ME context (:3140-3177).  Outside what the stage covers (a pair the ME stage declined)
 *    every segment runs the reference's function as before.
 */
#define _GNU_SOURCE /* RTLD_DEFAULT */
#include "../integration/seam_cpu.h"
#include <dlfcn.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "motion_estimation.h" /* declares svt_aom_motion_estimation_b64 before the macro below exists */
#include "me_context.h"
#include "pcs.h"
#include "temporal_filtering.h"
#include "svtav1_hip.h" /* include/svtav1_hip.h of this repository: the C ABI */

/* seam 3: the block range of a segment collapses while this thread asks the reference's function for its preamble only */
static __thread int seam_tf_preamble_only;
#undef SEGMENT_END_IDX
#define SEGMENT_END_IDX(index, pic_size_in_sb, num_of_seg) \
    (seam_tf_preamble_only ? SEGMENT_START_IDX(index, pic_size_in_sb, num_of_seg) : ((((index) + 1) * (pic_size_in_sb)) / (num_of_seg))) /* av1_common.h:30 */

EbErrorType svt_hip_seam_tf_motion_estimation_b64(PictureParentControlSet *pcs, uint32_t b64_index, uint32_t b64_origin_x, uint32_t b64_origin_y, MeContext *me_ctx,
                                                  EbPictureBufferDesc *input_ptr);
int         svt_hip_seam_tf_pair_tables(PictureParentControlSet *pcs, uint64_t ref_number, uint32_t n_sb, uint32_t *best_mv, int16_t *hme_sc, uint64_t *hme_sad);

#define TF_SUBPEL_ARGS                                                                                                                                          \
    TF_SUBPEL_SEARCH_PARAMS *tf_sp_param, PictureParentControlSet *pcs, MeContext *me_ctx, BlkStruct *blk_ptr, EbPictureBufferDesc *pic_ptr_ref,               \
        EbPictureBufferDesc *prediction_ptr, EbByte *pred, uint16_t **pred_16bit, uint32_t *stride_pred, EbByte *src, uint16_t **src_16bit, uint32_t *stride_src, \
        uint64_t *best_dist, int16_t *best_mv_x, int16_t *best_mv_y
#define TF_SUBPEL_PASS \
    tf_sp_param, pcs, me_ctx, blk_ptr, pic_ptr_ref, prediction_ptr, pred, pred_16bit, stride_pred, src, src_16bit, stride_src, best_dist, best_mv_x, best_mv_y
static void tf_subpel_search_use0(TF_SUBPEL_ARGS); /* the reference's function (defined by the #include) */
static void seam_tf_subpel_search(TF_SUBPEL_ARGS);
static void tf_subpel_search_use1(TF_SUBPEL_ARGS) { seam_tf_subpel_search(TF_SUBPEL_PASS); }
static void tf_subpel_search_use2(TF_SUBPEL_ARGS) { seam_tf_subpel_search(TF_SUBPEL_PASS); }
static void tf_subpel_search_use3(TF_SUBPEL_ARGS) { seam_tf_subpel_search(TF_SUBPEL_PASS); }
static void tf_subpel_search_use4(TF_SUBPEL_ARGS) { seam_tf_subpel_search(TF_SUBPEL_PASS); }

#define SEAM_CAT_(a, b) a##b
#define SEAM_CAT(a, b) SEAM_CAT_(a, b)
#define tf_subpel_search(...) SEAM_CAT(tf_subpel_search_use, __COUNTER__)(__VA_ARGS__)
#define TF_PIC_ARGS                                                                                                                                  \
    PictureParentControlSet **pcs_list, EbPictureBufferDesc **list_input_picture_ptr, uint8_t index_center, MotionEstimationContext_t *me_context_ptr, \
        const int32_t *noise_levels_log1p_fp16, int32_t segment_index, bool is_highbd
#define TF_PIC_PASS pcs_list, list_input_picture_ptr, index_center, me_context_ptr, noise_levels_log1p_fp16, segment_index, is_highbd
static EbErrorType produce_temporally_filtered_pic_use5(TF_PIC_ARGS); /* the reference's function (defined by the #include) */
static EbErrorType seam_produce_temporally_filtered_pic(TF_PIC_ARGS);
static EbErrorType seam_produce_temporally_filtered_pic_ld(TF_PIC_ARGS);
/* __COUNTER__ in file order: the definition of produce_temporally_filtered_pic (:2782) takes 5, the definition of the low-delay form (:3415) 6, its call (:4236) 7, the
 * call of the first (:4244) 8 */
static EbErrorType produce_temporally_filtered_pic_ld_use6(TF_PIC_ARGS); /* the reference's low-delay function (defined by the #include) */
static EbErrorType produce_temporally_filtered_pic_ld_use7(TF_PIC_ARGS) { return seam_produce_temporally_filtered_pic_ld(TF_PIC_PASS); }
static EbErrorType produce_temporally_filtered_pic_use8(TF_PIC_ARGS) { return seam_produce_temporally_filtered_pic(TF_PIC_PASS); }
#define produce_temporally_filtered_pic(...) SEAM_CAT(produce_temporally_filtered_pic_use, __COUNTER__)(__VA_ARGS__)
#define produce_temporally_filtered_pic_ld(...) SEAM_CAT(produce_temporally_filtered_pic_ld_use, __COUNTER__)(__VA_ARGS__)
#define svt_aom_motion_estimation_b64(pcs, i, x, y, ctx, pic) svt_hip_seam_tf_motion_estimation_b64(pcs, i, x, y, ctx, pic)
#include "temporal_filtering.c" /* resolves through -I$(REF)/Source/Lib/Codec */
#undef tf_subpel_search
#undef produce_temporally_filtered_pic
#undef produce_temporally_filtered_pic_ld
#undef svt_aom_motion_estimation_b64

/* ---- the sub-pel seam (after the #include: the block-numbering tables of temporal_filtering.c:44-90 are in scope) ---- */
enum { SP_RECS = 24 }; /* > the (picture, reference) pairs that can be in the filter at once */
typedef struct SubpelBatch {
    PictureParentControlSet *pcs;
    uint64_t                 picture_number;
    const EbPictureBufferDesc *ref;
    int                      live;
    uint32_t                 n_sb, per_sb, seen64;
    uint64_t                 stamp;
    SvtHipTfSubpelDesc      *descs;
    SvtHipTfSubpelResult    *res;
    size_t                   cap;
} SubpelBatch;
int svt_hip_seam_bind(unsigned long long picture_number); /* integration/enc_handle_binding.c: SVT_HIP_DEVICES sharding */
static struct {
    pthread_mutex_t lock;
    int             mode; /* -1 unknown */
    int (*search_host)(const SvtHipTfSubpelParams *, const void *, size_t, const void *, size_t, const SvtHipTfSubpelDesc *, uint32_t, SvtHipTfSubpelResult *); /* non-zero: the device path is off */
    SubpelBatch rec[SP_RECS];
    uint64_t    n_batches, n_blocks, n_served, n_fallback, stamp;
} SPS = {PTHREAD_MUTEX_INITIALIZER, -1};

static void sp_stats(void) {
    const char *f = getenv("SVT_HIP_TF_SUBPEL_SEAM_STATS");
    FILE       *o = f ? fopen(f, "w") : NULL;
    if (!o) return;
    fprintf(o, "pairs_batched %llu\nblocks_computed %llu\nsearches_served %llu\nsearches_by_reference %llu\n", (unsigned long long)SPS.n_batches,
            (unsigned long long)SPS.n_blocks, (unsigned long long)SPS.n_served, (unsigned long long)SPS.n_fallback);
    fclose(o);
}
static int sp_on(void) {
    if (__atomic_load_n(&SPS.mode, __ATOMIC_ACQUIRE) < 0) { /* (double-checked: the fast path reads outside the lock) */
        pthread_mutex_lock(&SPS.lock);
        if (SPS.mode < 0) {
            const char *e = getenv("SVT_HIP_TF_SUBPEL_SEAM");
            int         m = e && atoi(e) && getenv("SVT_HIP") && getenv("SVT_HIP_TF_ME_SEAM");
            if (m) {
                *(void **)&SPS.search_host = dlsym(RTLD_DEFAULT, "svt_hip_tf_subpel_search_host");
                if (!SPS.search_host) { fprintf(stderr, "SVT_HIP_TF_SUBPEL_SEAM: libsvtav1_hip is not loaded\n"); abort(); }
                atexit(sp_stats);
                fprintf(stderr, "SVT_HIP_TF_SUBPEL_SEAM: the temporal filter's sub-pel refinement runs as one device call per (picture, reference) pair\n");
            }
            __atomic_store_n(&SPS.mode, m, __ATOMIC_RELEASE);
        }
        pthread_mutex_unlock(&SPS.lock);
    }
    return __atomic_load_n(&SPS.mode, __ATOMIC_ACQUIRE);
}
static int is_bilinear(uint32_t interp_filters) { return interp_filters == (uint32_t)av1_make_interp_filters(BILINEAR, BILINEAR); }

/* every block of the pair the reference may search, with the starting vector its caller would pass (:1866-1870, :1980-1982, :2113-2115, :2232-2234) */
static int build_batch(SubpelBatch *B, PictureParentControlSet *pcs, MeContext *me_ctx, EbPictureBufferDesc *ref, EbByte src_sb, uint32_t src_stride,
                       const TF_SUBPEL_SEARCH_PARAMS *sp) {
    const uint32_t n_sb = pcs->b64_total_count, with8 = pcs->tf_ctrls.enable_8x8_pred ? 64 : 0, per_sb = 1 + 4 + 16 + with8, n = n_sb * per_sb;
    uint32_t *best_mv = malloc((size_t)n_sb * 85 * 4);
    int16_t  *hme_sc  = malloc((size_t)n_sb * 2 * sizeof(int16_t));
    uint64_t *hme_sad = malloc((size_t)n_sb * 8);
    if (!svt_hip_seam_tf_pair_tables(pcs, me_ctx->me_ds_ref_array[0][0].picture_number, n_sb, best_mv, hme_sc, hme_sad)) { free(best_mv); free(hme_sc); free(hme_sad); return 0; }
    if (n > B->cap) { B->descs = realloc(B->descs, (size_t)n * sizeof(*B->descs)); B->res = realloc(B->res, (size_t)n * sizeof(*B->res)); B->cap = n; }
    memset(B->descs, 0, (size_t)n * sizeof(*B->descs));
    /* the central picture's luma: src_sb is the top-left sample of the caller's 64x64 block (sb origin = pu origin - local origin) */
    const uint32_t sb_x0 = sp->pu_origin_x - sp->local_origin_x, sb_y0 = sp->pu_origin_y - sp->local_origin_y;
    EbPictureBufferDesc *cen = pcs->enhanced_pic;
    const uint8_t *src_buf = cen->buffer_y;
    const size_t   src_pic0 = (size_t)(src_sb - src_buf) - ((size_t)sb_y0 * src_stride + sb_x0); /* offset of picture sample (0, 0) in the buffer */
    if (src_stride != cen->stride_y || src_pic0 != (size_t)cen->org_y * cen->stride_y + cen->org_x) { free(best_mv); free(hme_sc); free(hme_sad); return 0; }
    const uint32_t pic_w_sb = (pcs->aligned_width + 63) / 64;
    const int      only64 = pcs->tf_ctrls.use_pred_64x64_only_th == (uint8_t)~0;
    for (uint32_t sb = 0; sb < n_sb; sb++) {
        const uint32_t  x0 = (sb % pic_w_sb) * 64, y0 = (sb / pic_w_sb) * 64;
        const uint32_t *mv = best_mv + (size_t)sb * 85;
        SvtHipTfSubpelDesc *d = B->descs + (size_t)sb * per_sb;
        const int exited = hme_sad[sb] < me_ctx->tf_me_exit_th; /* the ME call's early exit: no tables, 64x64 only, from the HME centre (:1866-1870) */
#define SP_SET(D, PX, PY, BS, BIL, MVW, FROM_SC)                                                                                                  \
    do {                                                                                                                                          \
        (D)->src_off = src_pic0 + (size_t)(PY) * src_stride + (PX); (D)->src_stride = src_stride; (D)->pu_x = (uint16_t)(PX); (D)->pu_y = (uint16_t)(PY); \
        (D)->bsize = (uint8_t)(BS); (D)->bilinear = (uint8_t)(BIL);                                                                               \
        (D)->mv_x = (int16_t)((FROM_SC) ? hme_sc[2 * sb] << 3 : (_MVXT(MVW)) << 3); (D)->mv_y = (int16_t)((FROM_SC) ? hme_sc[2 * sb + 1] << 3 : (_MVYT(MVW)) << 3); \
    } while (0)
        const int two_tap = pcs->tf_ctrls.use_2tap; /* (me_ctx->tf_ctrls mirrors it; the 64 / 32 searches take BILINEAR with it, :1801-1804, :1911-1914) */
        SP_SET(&d[0], x0, y0, 64, two_tap, mv[0], exited || only64);
        for (uint32_t i = 0; i < 4; i++) SP_SET(&d[1 + i], x0 + (i & 1) * 32, y0 + (i >> 1) * 32, 32, two_tap, mv[1 + i], 0);
        for (uint32_t i32 = 0; i32 < 4; i32++)
            for (uint32_t i16 = 0; i16 < 4; i16++) {
                const uint32_t pu = idx_32x32_to_idx_16x16[i32][i16], iy = subblock_xy_16x16[pu][0], ix = subblock_xy_16x16[pu][1];
                SP_SET(&d[5 + iy * 4 + ix], x0 + ix * 16, y0 + iy * 16, 16, 0, mv[5 + tab16x16[pu]], 0);
            }
        if (with8)
            for (uint32_t i32 = 0; i32 < 4; i32++)
                for (uint32_t i16 = 0; i16 < 4; i16++)
                    for (uint32_t i8 = 0; i8 < 4; i8++) {
                        const uint32_t pu = idx_32x32_to_idx_8x8[i32][i16][i8], iy = subblock_xy_8x8[pu][0], ix = subblock_xy_8x8[pu][1];
                        SP_SET(&d[21 + iy * 8 + ix], x0 + ix * 8, y0 + iy * 8, 8, 0, mv[21 + tab8x8[pu]], 0);
                    }
#undef SP_SET
        (void)exited;
    }
    SvtHipTfSubpelParams P;
    memset(&P, 0, sizeof(P));
    P.half_pel_mode = pcs->tf_ctrls.half_pel_mode; P.quarter_pel_mode = pcs->tf_ctrls.quarter_pel_mode; P.eight_pel_mode = pcs->tf_ctrls.eight_pel_mode;
    P.subsampling_shift = pcs->tf_ctrls.sub_sampling_shift; P.bit_depth = 8; P.early_exit_th = me_ctx->tf_subpel_early_exit_th;
    P.mi_rows = (uint32_t)pcs->av1_cm->mi_rows; P.mi_cols = (uint32_t)pcs->av1_cm->mi_cols;
    P.ref_org_x = ref->org_x; P.ref_org_y = ref->org_y; P.ref_stride = ref->stride_y;
    svt_hip_seam_bind(pcs->picture_number);
    if (SPS.search_host(&P, src_buf, cen->luma_size, ref->buffer_y, ref->luma_size, B->descs, n, B->res)) { /* no batch: every search of the pair takes the reference's function */
        free(best_mv); free(hme_sc); free(hme_sad);
        return 0;
    }
    B->n_sb = n_sb; B->per_sb = per_sb;
    SPS.n_batches++; SPS.n_blocks += n;
    free(best_mv); free(hme_sc); free(hme_sad);
    return 1;
}

static void seam_tf_subpel_search(TF_SUBPEL_ARGS) {
    if (!sp_on() || tf_sp_param->is_highbd) { tf_subpel_search_use0(TF_SUBPEL_PASS); return; }
    pthread_mutex_lock(&SPS.lock);
    SubpelBatch *B = NULL, *spare = NULL;
    for (int i = 0; i < SP_RECS; i++) {
        SubpelBatch *r = &SPS.rec[i];
        if (r->live && r->pcs == pcs && r->picture_number == pcs->picture_number && r->ref == pic_ptr_ref) { B = r; break; }
        if (!spare || r->stamp < spare->stamp) spare = r; /* the oldest batch (or a never-used slot: stamp 0) */
    }
    if (!B && spare) {
        spare->pcs = pcs; spare->picture_number = pcs->picture_number; spare->ref = pic_ptr_ref; spare->seen64 = 0; spare->stamp = ++SPS.stamp;
        /* src[C_Y] = the caller's 64x64 source block (produce_temporally_filtered_pic hands the block's plane pointers on) */
        if (build_batch(spare, pcs, me_ctx, pic_ptr_ref, src[C_Y], stride_src[C_Y], tf_sp_param)) { spare->live = 1; B = spare; }
    }
    int served = 0;
    if (B) {
        const uint32_t pic_w_sb = (pcs->aligned_width + 63) / 64;
        const uint32_t sb_x0 = tf_sp_param->pu_origin_x - tf_sp_param->local_origin_x, sb_y0 = tf_sp_param->pu_origin_y - tf_sp_param->local_origin_y;
        const uint32_t sb = (sb_y0 / 64) * pic_w_sb + sb_x0 / 64, lx = tf_sp_param->local_origin_x, ly = tf_sp_param->local_origin_y, bs = tf_sp_param->bsize;
        int slot = -1;
        if (bs == 64) slot = 0;
        else if (bs == 32) slot = 1 + (ly / 32) * 2 + lx / 32;
        else if (bs == 16) slot = 5 + (ly / 16) * 4 + lx / 16;
        else if (bs == 8 && B->per_sb > 21) slot = 21 + (ly / 8) * 8 + lx / 8;
        if (slot >= 0 && sb < B->n_sb) {
            const SvtHipTfSubpelDesc   *d = &B->descs[(size_t)sb * B->per_sb + slot];
            const SvtHipTfSubpelResult *r = &B->res[(size_t)sb * B->per_sb + slot];
            if (d->bsize == bs && d->pu_x == tf_sp_param->pu_origin_x && d->pu_y == tf_sp_param->pu_origin_y && d->mv_x == *best_mv_x && d->mv_y == *best_mv_y &&
                d->bilinear == is_bilinear(tf_sp_param->interp_filters) && *best_dist == (uint64_t)INT_MAX &&
                tf_sp_param->subsampling_shift == pcs->tf_ctrls.sub_sampling_shift) {
                *best_dist = r->dist; *best_mv_x = r->mv_x; *best_mv_y = r->mv_y;
                served = 1;
            }
        }
        /* (a batch stays until its slot is reused for a newer pair -- the oldest first; a pair's key never recurs.  Ending it with the last 64x64 search would
         * take it away from the 32x32 / 16x16 searches that follow in that block and in the blocks other threads are still working on.) */
    }
    if (served) SPS.n_served++; else SPS.n_fallback++;
    pthread_mutex_unlock(&SPS.lock);
    if (!served) tf_subpel_search_use0(TF_SUBPEL_PASS);
}


#include <time.h>
static double seam_ms_now(void) { struct timespec t_; clock_gettime(CLOCK_MONOTONIC, &t_); return 1e3 * (double)t_.tv_sec + 1e-6 * (double)t_.tv_nsec; }
/* ---- seam 3: one central picture = one device stage ------------------------------------------------------------------------------------------------------------ */
int svt_hip_seam_tf_pair_run(PictureParentControlSet *pcs, MeContext *c, uint32_t n_sb, uint32_t *best_sad, uint32_t *best_mv, int16_t *hme_sc, uint64_t *hme_sad);
enum { TFD_RECS = 16 };
typedef struct TfPicRec {
    PictureParentControlSet *pcs;
    uint64_t                 picture_number;
    int                      state; /* 0 free, 1 running, 2 done on the device, 3 left to the reference */
    uint32_t                 seen;  /* segments that passed through */
} TfPicRec;
static struct {
    pthread_mutex_t lock;
    pthread_cond_t  ready;
    int             mode; /* -1 unknown */
    int (*picture_host)(const SvtHipTfPictureParams *, const SvtHipTfHostPicture *, const SvtHipTfHostPicture *, const SvtHipTfMeTables *, uint32_t, void *, void *, void *,
                        SvtHipTfPictureStats *);
    TfPicRec    rec[TFD_RECS];
    uint64_t    n_pictures, n_declined, n_refs, n64, n32, n16, n8, n_exit;
    unsigned long long us_stage, us_pairs; /* microseconds inside svt_hip_tf_picture_host / inside the pairs' ME stage calls */
    const char *last_decline;
} TFD = {PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, -1};

static void tfd_stats(void) {
    const char *f = getenv("SVT_HIP_TF_SEAM_STATS");
    FILE       *o = f ? fopen(f, "w") : NULL;
    if (!o) return;
    fprintf(o, "ms_in_stage_calls %llu\nms_in_me_pairs %llu\n", (unsigned long long)(TFD.us_stage / 1000), (unsigned long long)(TFD.us_pairs / 1000));
    fprintf(o, "pictures_filtered %llu\npictures_declined %llu\nreference_frames %llu\npred_64x64 %llu\npred_32x32 %llu\npred_16x16 %llu\npred_8x8 %llu\nearly_exit_blocks %llu\nlast_decline %s\n",
            (unsigned long long)TFD.n_pictures, (unsigned long long)TFD.n_declined, (unsigned long long)TFD.n_refs, (unsigned long long)TFD.n64, (unsigned long long)TFD.n32,
            (unsigned long long)TFD.n16, (unsigned long long)TFD.n8, (unsigned long long)TFD.n_exit, TFD.last_decline ? TFD.last_decline : "-");
    fclose(o);
}
static int tfd_on(void) {
    if (__atomic_load_n(&TFD.mode, __ATOMIC_ACQUIRE) < 0) { /* (double-checked: the fast path reads outside the lock) */
        pthread_mutex_lock(&TFD.lock);
        if (TFD.mode < 0) {
            const char *e = getenv("SVT_HIP_TF_SEAM");
            int         m = e && atoi(e) && getenv("SVT_HIP") && getenv("SVT_HIP_TF_ME_SEAM");
            if (m) {
                *(void **)&TFD.picture_host = dlsym(RTLD_DEFAULT, "svt_hip_tf_picture_host");
                if (!TFD.picture_host) { fprintf(stderr, "SVT_HIP_TF_SEAM: libsvtav1_hip is not loaded\n"); abort(); }
                atexit(tfd_stats);
                fprintf(stderr, "SVT_HIP_TF_SEAM: the temporal filter of a central picture runs as one device stage\n");
            }
            __atomic_store_n(&TFD.mode, m, __ATOMIC_RELEASE);
        }
        pthread_mutex_unlock(&TFD.lock);
    }
    return __atomic_load_n(&TFD.mode, __ATOMIC_ACQUIRE);
}
static int tfd_decline(const char *why) { TFD.last_decline = why; return -1; }

/* the whole picture on the device; 0 = the central picture's buffers hold the filtered blocks */
static int tfd_run_picture(TF_PIC_ARGS, int low_delay) {
    PictureParentControlSet *centre_pcs = pcs_list[index_center];
    SequenceControlSet      *scs        = centre_pcs->scs;
    EbPictureBufferDesc     *cen        = list_input_picture_ptr[index_center];
    MeContext               *ctx        = me_context_ptr->me_ctx;
    const uint32_t ss_x = scs->subsampling_x, ss_y = scs->subsampling_y;
    /* high bit depth: the reference filters the packed 16-bit copies of the pictures (altref_buffer_highbd, made by svt_av1_init_temporal_filtering :4171-4190);
     * with tf_ctrls.use_8bit_subpel its sub-pel searches read the 8-bit luma of the same pictures (:3203, :3242) */
    const int enc_bd = scs->static_config.encoder_bit_depth, sp8 = is_highbd && ctx->tf_ctrls.use_8bit_subpel;
    if (is_highbd && enc_bd != 10) return tfd_decline("bit depth");
    if (is_highbd && (cen->stride_bit_inc_cb != cen->stride_cb || cen->stride_bit_inc_cr != cen->stride_cr)) return tfd_decline("chroma stride of the packed planes");
    if (ss_x != 1 || ss_y != 1) return tfd_decline("not 4:2:0");
    const uint32_t blk_cols = (uint32_t)(cen->width + BW - 1) / BW, blk_rows = (uint32_t)(cen->height + BH - 1) / BH, n_sb = blk_cols * blk_rows; /* :2823-2826 */
    if (n_sb != centre_pcs->b64_total_count) return tfd_decline("block grid");
    /* the frames the reference would use, in its order (:3086-3131) */
    int       used[ALTREF_MAX_NFRAMES], n_used = 0;
    const int start_frame_index[3] = {0, centre_pcs->past_altref_nframes, centre_pcs->past_altref_nframes + 1};
    const int end_frame_index[3]   = {centre_pcs->past_altref_nframes - 1, centre_pcs->past_altref_nframes, centre_pcs->past_altref_nframes + centre_pcs->future_altref_nframes};
    /* the low-delay form filters with every frame of the window (:3665-3668) and takes no picture-level decision */
    for (int frame_index = 0; low_delay && frame_index < centre_pcs->past_altref_nframes + centre_pcs->future_altref_nframes + 1; frame_index++)
        if (frame_index != index_center) used[n_used++] = frame_index;
    for (int segment_idx = 0; segment_idx < 3 && !low_delay; segment_idx++)
        for (int frame_index = start_frame_index[segment_idx]; frame_index <= end_frame_index[segment_idx]; frame_index = frame_index + ctx->tf_ctrls.ref_frame_factor) {
            if (frame_index == index_center) continue;
            const uint32_t low_ahd_err = centre_pcs->aligned_width * centre_pcs->aligned_height;
            const uint8_t  th          = (centre_pcs->slice_type == I_SLICE) ? 20 : 40;
            if (pcs_list[frame_index]->tf_ahd_error_to_central > low_ahd_err &&
                ((int)(((int)pcs_list[frame_index]->tf_ahd_error_to_central - (int)centre_pcs->tf_avg_ahd_error) * 100)) > (th * (int)centre_pcs->tf_avg_ahd_error))
                continue;
            uint32_t bright_change_region_cnt = 0;
            for (uint32_t rw = 0; rw < scs->picture_analysis_number_of_regions_per_width; rw++)
                for (uint32_t rh = 0; rh < scs->picture_analysis_number_of_regions_per_height; rh++)
                    if (ABS((int)pcs_list[frame_index]->average_intensity_per_region[rw][rh] - (int)centre_pcs->average_intensity_per_region[rw][rh]) > 2 &&
                        pcs_list[frame_index]->avg_luma != centre_pcs->tf_avg_luma)
                        bright_change_region_cnt++;
            if (bright_change_region_cnt >= ((14 * scs->picture_analysis_number_of_regions_per_width * scs->picture_analysis_number_of_regions_per_height) / 16)) continue;
            used[n_used++] = frame_index;
        }
    if (n_used == 0) return 0; /* every frame skipped: accumulators hold 1000 x the central picture, the count is 1000 -- (1000 c + 500) / 1000 = c, the picture stays as it is (:2608-2672) */
    if (n_used > SVT_HIP_TF_MAX_FRAMES) return tfd_decline("more frames than SVT_HIP_TF_MAX_FRAMES");
    for (int i = 0; i < n_used; i++) {
        const EbPictureBufferDesc *r = list_input_picture_ptr[used[i]];
        if (r->stride_y != cen->stride_y || r->stride_cb != cen->stride_cb || r->org_x != cen->org_x || r->org_y != cen->org_y || r->luma_size != cen->luma_size ||
            r->chroma_size != cen->chroma_size)
            return tfd_decline("frame geometry");
    }
    /* the preamble of the reference's function -- decay factors into ctx (:2870-3035) -- over an empty block range */
    seam_tf_preamble_only = 1;
    if (low_delay) produce_temporally_filtered_pic_ld_use6(TF_PIC_PASS); /* (:3516-3607) */
    else produce_temporally_filtered_pic_use5(TF_PIC_PASS);
    seam_tf_preamble_only = 0;
    /* the ME stage of every pair, context as the reference sets it up (:3140-3177) */
    uint32_t *best_sad = malloc((size_t)n_used * n_sb * 85 * 4), *best_mv = malloc((size_t)n_used * n_sb * 85 * 4);
    int16_t  *hme_sc   = malloc((size_t)n_used * n_sb * 2 * sizeof(int16_t));
    uint64_t *hme_sad  = malloc((size_t)n_used * n_sb * 8);
    int       rc       = 0;
    SvtHipTfMeTables    me[SVT_HIP_TF_MAX_FRAMES];
    SvtHipTfHostPicture refs[SVT_HIP_TF_MAX_FRAMES];
    for (int i = 0; i < n_used && !rc; i++) {
        const int frame_index = used[i];
        ctx->tf_frame_index = frame_index; ctx->tf_index_center = index_center;
        if (!low_delay) {
        create_me_context_and_picture_control(me_context_ptr, pcs_list[frame_index], centre_pcs, cen, 0, 0, ss_x, ss_y);
        ctx->num_of_list_to_search = 1; ctx->num_of_ref_pic_to_search[0] = 1; ctx->num_of_ref_pic_to_search[1] = 0;
        ctx->temporal_layer_index = centre_pcs->temporal_layer_index; ctx->is_ref = centre_pcs->is_ref;
        EbPaReferenceObject *ref_object = (EbPaReferenceObject *)ctx->alt_ref_reference_ptr;
        ctx->me_ds_ref_array[0][0].picture_ptr = ref_object->input_padded_pic;
        ctx->me_ds_ref_array[0][0].sixteenth_picture_ptr = ref_object->sixteenth_downsampled_picture_ptr;
        ctx->me_ds_ref_array[0][0].quarter_picture_ptr = ref_object->quarter_downsampled_picture_ptr;
        ctx->me_ds_ref_array[0][0].picture_number = ref_object->picture_number;
        ctx->tf_me_exit_th = centre_pcs->tf_ctrls.me_exit_th; ctx->tf_use_pred_64x64_only_th = centre_pcs->tf_ctrls.use_pred_64x64_only_th;
        ctx->tf_subpel_early_exit_th = centre_pcs->tf_ctrls.subpel_early_exit_th;
        set_hme_search_params_mctf(ctx, 0);
        }
        me[i].best_sad = best_sad + (size_t)i * n_sb * 85; me[i].best_mv = best_mv + (size_t)i * n_sb * 85; me[i].hme_sc = hme_sc + (size_t)i * n_sb * 2; me[i].hme_sad = hme_sad + (size_t)i * n_sb;
        const double tp_ = seam_ms_now();
        if (low_delay) { /* no search: the prediction is the co-located block (:3711-3726) */
            me[i].best_sad = me[i].best_mv = NULL; me[i].hme_sc = NULL; me[i].hme_sad = NULL;
        } else if (!svt_hip_seam_tf_pair_run(centre_pcs, ctx, n_sb, (uint32_t *)me[i].best_sad, (uint32_t *)me[i].best_mv, (int16_t *)me[i].hme_sc, (uint64_t *)me[i].hme_sad))
            rc = tfd_decline("a pair outside the ME stage");
        __atomic_fetch_add(&TFD.us_pairs, (unsigned long long)((seam_ms_now() - tp_) * 1e3), __ATOMIC_RELAXED);
        const EbPictureBufferDesc *r = list_input_picture_ptr[frame_index];
        if (!is_highbd) { refs[i].y = r->buffer_y; refs[i].u = r->buffer_cb; refs[i].v = r->buffer_cr; refs[i].y8 = NULL; }
        else {
            uint16_t **hb = pcs_list[frame_index]->altref_buffer_highbd;
            refs[i].y = hb[C_Y]; refs[i].u = hb[C_U]; refs[i].v = hb[C_V]; refs[i].y8 = sp8 ? r->buffer_y : NULL;
            if (!hb[C_Y] || (ctx->tf_chroma && (!hb[C_U] || !hb[C_V]))) rc = tfd_decline("a packed 16-bit frame is missing");
        }
        refs[i].y_samples = r->luma_size; refs[i].uv_samples = r->chroma_size;
    }
    if (!rc) {
        SvtHipTfPictureParams P;
        memset(&P, 0, sizeof(P));
        P.sp.half_pel_mode = centre_pcs->tf_ctrls.half_pel_mode; P.sp.quarter_pel_mode = centre_pcs->tf_ctrls.quarter_pel_mode; P.sp.eight_pel_mode = centre_pcs->tf_ctrls.eight_pel_mode;
        P.sp.subsampling_shift = centre_pcs->tf_ctrls.sub_sampling_shift; P.sp.bit_depth = (uint8_t)(is_highbd ? enc_bd : 8); P.sp.early_exit_th = centre_pcs->tf_ctrls.subpel_early_exit_th;
        P.subpel_8bit = (uint8_t)sp8; P.zero_motion = (uint8_t)low_delay;
        P.sp.mi_rows = (uint32_t)centre_pcs->av1_cm->mi_rows; P.sp.mi_cols = (uint32_t)centre_pcs->av1_cm->mi_cols;
        P.sp.ref_org_x = cen->org_x; P.sp.ref_org_y = cen->org_y; P.sp.ref_stride = cen->stride_y;
        for (int k = 0; k < 3; k++) P.tf.tf_decay_factor_fp16[k] = ctx->tf_decay_factor_fp16[k];
        P.tf.tf_mv_dist_th = ctx->tf_mv_dist_th; P.tf.tf_chroma = ctx->tf_chroma; P.tf.use_zz_based_filter = ctx->tf_ctrls.use_zz_based_filter;
        P.tf.encoder_bit_depth = (uint8_t)(is_highbd ? enc_bd : 8); P.tf.ss_x = (uint8_t)ss_x; P.tf.ss_y = (uint8_t)ss_y;
        P.pic_w_sb = blk_cols; P.pic_h_sb = blk_rows; P.uv_stride = cen->stride_cb;
        P.me_exit_th = centre_pcs->tf_ctrls.me_exit_th; P.pred_error_32x32_th = centre_pcs->tf_ctrls.pred_error_32x32_th;
        P.use_2tap = centre_pcs->tf_ctrls.use_2tap; P.enable_8x8_pred = centre_pcs->tf_ctrls.enable_8x8_pred; P.use_pred_64x64_only_th = centre_pcs->tf_ctrls.use_pred_64x64_only_th;
        uint16_t **chb = centre_pcs->altref_buffer_highbd;
        SvtHipTfHostPicture C = {cen->buffer_y, cen->buffer_cb, cen->buffer_cr, cen->luma_size, cen->chroma_size, NULL};
        if (is_highbd) { C.y = chb[C_Y]; C.u = chb[C_U]; C.v = chb[C_V]; C.y8 = sp8 ? cen->buffer_y : NULL; }
        SvtHipTfPictureStats st;
        memset(&st, 0, sizeof(st));
        svt_hip_seam_bind(centre_pcs->picture_number);
        if (is_highbd && (!chb[C_Y] || (ctx->tf_chroma && (!chb[C_U] || !chb[C_V])))) rc = tfd_decline("the central picture's packed 16-bit copy is missing");
        const double ts_ = seam_ms_now();
        if (!rc && TFD.picture_host(&P, &C, refs, me, (uint32_t)n_used, (void *)C.y, (void *)C.u, (void *)C.v, &st)) rc = tfd_decline("svt_hip_tf_picture_host refused the parameters");
        __atomic_fetch_add(&TFD.us_stage, (unsigned long long)((seam_ms_now() - ts_) * 1e3), __ATOMIC_RELAXED);
        if (!rc) {
            /* the horizontal / vertical vote of the ME calls (motion_estimation.c:2469-2474): one per (block, frame), summed into the picture by the caller (:4255-4258) */
            for (size_t k = 0; k < (size_t)n_used * n_sb && !low_delay; k++) {
                if (ABS(hme_sc[2 * k]) > ABS(hme_sc[2 * k + 1])) ctx->tf_tot_horz_blks++;
                else ctx->tf_tot_vert_blks++;
            }
            pthread_mutex_lock(&TFD.lock);
            TFD.n_refs += (uint64_t)n_used; TFD.n64 += st.blocks_64x64; TFD.n32 += st.blocks_32x32; TFD.n16 += st.blocks_16x16; TFD.n8 += st.blocks_8x8; TFD.n_exit += st.early_exit_blocks;
            pthread_mutex_unlock(&TFD.lock);
        }
    }
    free(best_sad); free(best_mv); free(hme_sc); free(hme_sad);
    return rc;
}

static EbErrorType seam_tf_picture(TF_PIC_ARGS, int low_delay) {
#define TF_REFERENCE_FORM() (low_delay ? produce_temporally_filtered_pic_ld_use6(TF_PIC_PASS) : produce_temporally_filtered_pic_use5(TF_PIC_PASS))
    if (!tfd_on()) return TF_REFERENCE_FORM();
    PictureParentControlSet *centre_pcs = pcs_list[index_center];
    pthread_mutex_lock(&TFD.lock);
    TfPicRec *R = NULL, *spare = NULL;
    for (int i = 0; i < TFD_RECS; i++) {
        if (TFD.rec[i].state && TFD.rec[i].pcs == centre_pcs && TFD.rec[i].picture_number == centre_pcs->picture_number) { R = &TFD.rec[i]; break; }
        if (!TFD.rec[i].state && !spare) spare = &TFD.rec[i];
    }
    if (!R) {
        if (!spare) { fprintf(stderr, "SVT_HIP_TF_SEAM: more than %d central pictures in flight\n", TFD_RECS); abort(); }
        R = spare;
        R->pcs = centre_pcs; R->picture_number = centre_pcs->picture_number; R->seen = 0; R->state = 1;
        pthread_mutex_unlock(&TFD.lock);
        const int rc = tfd_run_picture(TF_PIC_PASS, low_delay); /* this segment's thread runs the picture; the others wait below */
        pthread_mutex_lock(&TFD.lock);
        R->state = rc ? 3 : 2;
        if (rc) TFD.n_declined++; else TFD.n_pictures++;
        pthread_cond_broadcast(&TFD.ready);
    }
    while (R->state == 1) pthread_cond_wait(&TFD.ready, &TFD.lock);
    const int declined = R->state == 3;
    if (++R->seen == centre_pcs->tf_segments_total_count) R->state = 0;
    pthread_mutex_unlock(&TFD.lock);
    return declined ? TF_REFERENCE_FORM() : EB_ErrorNone;
#undef TF_REFERENCE_FORM
}
/* Every call rewrites (its segment of) the central picture in place, on the device or by the reference's own code: the copy the ME stage may hold of it is stale
 * from here on (integration/me_process_seam.c decides residency by this notice, not by content) */
void svt_hip_seam_me_invalidate(unsigned long long picture_number);
static EbErrorType seam_produce_temporally_filtered_pic(TF_PIC_ARGS) {
    SEAM_CPU_BEGIN();
    const EbErrorType e = seam_tf_picture(TF_PIC_PASS, 0);
    svt_hip_seam_me_invalidate(pcs_list[index_center]->picture_number);
    SEAM_CPU_END(SEAM_CPU_TF);
    return e;
}
static EbErrorType seam_produce_temporally_filtered_pic_ld(TF_PIC_ARGS) {
    SEAM_CPU_BEGIN();
    const EbErrorType e = seam_tf_picture(TF_PIC_PASS, 1);
    svt_hip_seam_me_invalidate(pcs_list[index_center]->picture_number);
    SEAM_CPU_END(SEAM_CPU_TF);
    return e;
}
