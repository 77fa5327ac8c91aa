/* src_ops_process_seam.c -- REFERENCE-SIDE BINDING (what a maintainer of the reference adds; built into the reference encoder by oracle/Makefile for the identity / fps runs): the reference's source-based-operations process with the TPL dispenser seam of INTEGRATION.md §3.
 *
 * This translation unit IS Source/Lib/Codec/src_ops_process.c of the reference (included below where it lies; nothing is copied).  Two `static` functions are renamed
 * for the duration of the #include, with __COUNTER__ (unused anywhere else in the file) appended in file order: tpl_mc_flow_dispenser_sb_generic(...) -- defined at
 * :519 (_use0: the reference's body, untouched), called per SB by the dispenser threads at :2060 / :2077 (_use3 / _use4 = the per-SB seam below) -- and
 * tpl_mc_flow_dispenser(...) -- defined at :1347 (_use1), called once per (TPL group, picture) from tpl_mc_flow at :1848 (_use2 = the picture seam below).
 *
 * With SVT_HIP_TPL_RECON_SEAM=1 on top of SVT_HIP_TPL_SEAM the picture seam ALSO runs the reconstruction half of every block on the device
 * (svt_hip_tpl_recon_stage_host: prediction into the picture's TPL reconstruction, transform, quantisation, inverse transform, the block's statistics), hands the
 * statistics to the reference's own result_model_store() block by block, and lets the reference's dispenser run with the per-SB function reduced to nothing for that
 * picture: its segment threads still walk the SBs, count them and release the picture, and tpl_mc_flow_dispenser pads the reconstruction (:1400-1406).
 *
 * With SVT_HIP_TPL_SEAM=1 and a picture whose source-based statistics are still to be made (pcs->tpl_src_data_ready == 0) the seam computes them for EVERY block
 * of the picture with ONE svt_hip_tpl_src_stage_host() call -- DC intra cost from source neighbours, SAD of each uni-directional ME candidate, winner, forward
 * transform + quantisation error of the NEWMV residual (src_ops_process.c:606-957) --, writes them where the reference itself would store them
 * (pa_me_data->tpl_src_stats_buffer, :958-967) and runs the reference's dispenser with pcs->tpl_src_data_ready = 1, i.e. through its own "statistics already
 * there" branch (:969-977): the reconstruction half (prediction from the TPL recon pictures, transform, inverse transform, result_model_store) stays the
 * reference's C, segment threads and all.  The source-based half of a block reads only source pictures, the ME results and the quantizer row, so doing it for the
 * whole picture before the first segment is an exact reordering.  Pictures outside the covered option set (tpl levels 4 / 5: use_sad_in_src_search, DC only,
 * full-pel, no rate, no QPS) run the reference unchanged.  SVT_HIP_TPL_SEAM_STATS=<file> receives the counters at exit.
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
#include "encode_context.h"
#include "svtav1_hip.h" /* include/svtav1_hip.h of this repository: the C ABI */

struct SourceBasedOperationsContext;
#define TPL_DISP_ARGS                                                                                                                                         \
    EncodeContext *enc_ctx, SequenceControlSet *scs, int32_t *base_rdmult, PictureParentControlSet *pcs, int32_t frame_idx, struct SourceBasedOperationsContext *context_ptr
#define TPL_DISP_PASS enc_ctx, scs, base_rdmult, pcs, frame_idx, context_ptr
static void tpl_mc_flow_dispenser_use1(TPL_DISP_ARGS); /* the reference's function (defined by the #include) */
static void tpl_mc_flow_dispenser_use2(TPL_DISP_ARGS); /* the picture seam */
#define TPL_SB_ARGS EncodeContext *enc_ctx, SequenceControlSet *scs, PictureParentControlSet *pcs, int32_t frame_idx, uint32_t sb_index, int32_t qIndex, uint8_t dispenser_search_level
#define TPL_SB_PASS enc_ctx, scs, pcs, frame_idx, sb_index, qIndex, dispenser_search_level
static void tpl_mc_flow_dispenser_sb_generic_use0(TPL_SB_ARGS); /* the reference's per-SB function (defined by the #include) */
static void seam_tpl_sb(TPL_SB_ARGS);
/* (the per-SB calls run on the TPL dispenser threads: they are where the reference spends the stage's CPU time, so they carry the accounting of seam_cpu.h) */
static void tpl_mc_flow_dispenser_sb_generic_use3(TPL_SB_ARGS) { SEAM_CPU_BEGIN(); seam_tpl_sb(TPL_SB_PASS); SEAM_CPU_END(SEAM_CPU_TPL); }
static void tpl_mc_flow_dispenser_sb_generic_use4(TPL_SB_ARGS) { SEAM_CPU_BEGIN(); seam_tpl_sb(TPL_SB_PASS); SEAM_CPU_END(SEAM_CPU_TPL); }

#define SEAM_CAT_(a, b) a##b
#define SEAM_CAT(a, b) SEAM_CAT_(a, b)
#define tpl_mc_flow_dispenser(...) SEAM_CAT(tpl_mc_flow_dispenser_use, __COUNTER__)(__VA_ARGS__)
#define tpl_mc_flow_dispenser_sb_generic(...) SEAM_CAT(tpl_mc_flow_dispenser_sb_generic_use, __COUNTER__)(__VA_ARGS__)
#include "src_ops_process.c" /* resolves through -I$(REF)/Source/Lib/Codec */
#undef tpl_mc_flow_dispenser
#undef tpl_mc_flow_dispenser_sb_generic

int svt_hip_seam_bind(unsigned long long picture_number); /* integration/enc_handle_binding.c: SVT_HIP_DEVICES sharding */
static struct {
    pthread_mutex_t lock;
    int             mode;
    int (*stage_host)(const SvtHipTplSrcParams *, const SvtHipTplHostPlanes *, const uint8_t *, const uint32_t *, const uint8_t *, SvtHipTplSrcStats *);
    uint64_t n_pictures, n_blocks, n_newmv, n_declined, n_reused;
    double   ms_stage;
    int      recon; /* SVT_HIP_TPL_RECON_SEAM */
    int (*recon_host)(const SvtHipTplReconParams *, const SvtHipTplHostPlanes *, const SvtHipTplSrcStats *, uint8_t *, uint32_t, SvtHipTplReconStats *);
    int (*fused_host)(const SvtHipTplReconParams *, const SvtHipTplHostPlanes *, const SvtHipTplHostPlanes *, const SvtHipTplPlaneIds *, const uint8_t *, const uint32_t *,
                      const uint8_t *, SvtHipTplSrcStats *, uint8_t *, uint32_t, SvtHipTplReconStats *); /* both halves (or the second alone) in one call, planes kept resident
                                                                                                          * on the device across calls (svt_hip_tpl_stage_host_resident) */
    void (*plane_drop)(const void *);
    void (*plane_counts)(uint64_t *, uint64_t *);
    uint64_t n_recon_pictures, n_recon_blocks, n_recon_coded, n_sb_calls_skipped, n_fused;
    double   ms_recon;
    PictureParentControlSet *done[16]; /* pictures whose blocks were reconstructed on the device: the per-SB function has nothing left to do */
} TS = {PTHREAD_MUTEX_INITIALIZER};

static void tpl_seam_stats(void) {
    const char *f = getenv("SVT_HIP_TPL_SEAM_STATS");
    FILE       *o = f ? fopen(f, "w") : NULL;
    if (!o) return;
    fprintf(o, "pictures_offloaded %llu\nblocks %llu\nblocks_newmv %llu\npictures_declined %llu\npictures_with_stored_statistics %llu\nms_in_stage_calls %.0f\n",
            (unsigned long long)TS.n_pictures, (unsigned long long)TS.n_blocks, (unsigned long long)TS.n_newmv, (unsigned long long)TS.n_declined,
            (unsigned long long)TS.n_reused, TS.ms_stage);
    if (TS.recon && TS.plane_counts) {
        uint64_t h_ = 0, m_ = 0;
        TS.plane_counts(&h_, &m_);
        fprintf(o, "planes_found_resident %llu\nplanes_uploaded %llu\n", (unsigned long long)h_, (unsigned long long)m_);
    }
    if (TS.recon)
        fprintf(o, "recon_pictures %llu\nrecon_blocks %llu\nrecon_blocks_coded %llu\nsb_calls_skipped %llu\nms_in_recon_stage_calls %.0f\npictures_both_halves_in_one_call %llu\n", (unsigned long long)TS.n_recon_pictures,
                (unsigned long long)TS.n_recon_blocks, (unsigned long long)TS.n_recon_coded, (unsigned long long)TS.n_sb_calls_skipped, TS.ms_recon, (unsigned long long)TS.n_fused);
    fclose(o);
}
static void tpl_seam_init(void) {
    const char *e = getenv("SVT_HIP_TPL_SEAM");
    if (!e || !atoi(e) || !getenv("SVT_HIP")) return;
    *(void **)&TS.stage_host = dlsym(RTLD_DEFAULT, "svt_hip_tpl_src_stage_host");
    if (!TS.stage_host) { fprintf(stderr, "SVT_HIP_TPL_SEAM: libsvtav1_hip is not loaded\n"); abort(); }
    atexit(tpl_seam_stats);
    fprintf(stderr, "SVT_HIP_TPL_SEAM: the source-based half of the TPL dispenser runs as one device stage per picture\n");
    TS.mode = 1;
    const char *r = getenv("SVT_HIP_TPL_RECON_SEAM");
    if (r && atoi(r)) {
        *(void **)&TS.recon_host = dlsym(RTLD_DEFAULT, "svt_hip_tpl_recon_stage_host");
        *(void **)&TS.fused_host = dlsym(RTLD_DEFAULT, "svt_hip_tpl_stage_host_resident");
        *(void **)&TS.plane_drop = dlsym(RTLD_DEFAULT, "svt_hip_tpl_plane_drop");
        *(void **)&TS.plane_counts = dlsym(RTLD_DEFAULT, "svt_hip_tpl_plane_counts");
        if (!TS.recon_host) { fprintf(stderr, "SVT_HIP_TPL_RECON_SEAM: libsvtav1_hip is not loaded\n"); abort(); }
        fprintf(stderr, "SVT_HIP_TPL_RECON_SEAM: the reconstruction half of the TPL dispenser runs as a device stage per picture too\n");
        TS.recon = 1;
    }
}
static int tpl_seam_on(void) {
    static pthread_once_t once = PTHREAD_ONCE_INIT;
    pthread_once(&once, tpl_seam_init);
    return TS.mode;
}
static double now_ms(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e3 + t.tv_nsec / 1e6;
}

/* the option set svt_hip_tpl_src_stage covers (tpl levels 4 / 5, initial_rc_process.c:343-378) on an 8-bit single-tile-grid picture */
/* search_flags of SvtHipTplSrcParams for this picture's controls; 0 with intra_mode_end == DC_PRED = the option set of tpl levels 4 / 5 (the fast kernels) */
static uint8_t tpl_search_flags(const TplControls *tc) {
    const int rounds = FULL_PEL - tc->subpel_depth; /* (subpel_diag_refinement is read by the sub-pel search only: levels 4 / 5 set it to 4 with FULL_PEL) */
    return (uint8_t)((tc->use_sad_in_src_search ? 0 : 1) | (tc->compute_rate ? 2 : 0) | (rounds << 2) | (rounds && tc->subpel_diag_refinement >= 4 ? 16 : 0));
}
static int tpl_seam_covers(const SequenceControlSet *scs, const PictureParentControlSet *pcs) {
    const TplControls *tc = &pcs->tpl_ctrls;
    if (!scs->in_loop_ois || scs->b64_size != 64) return 0;
    if (tc->intra_mode_end == DC_PRED && !tpl_search_flags(tc)) /* tpl levels 4 / 5 */
        return (tc->dispenser_search_level == 0 && tc->subsample_tx == 0) || (tc->dispenser_search_level == 1 && tc->subsample_tx == 2);
    /* tpl levels 0-3 (set_tpl_params, initial_rc_process.c:301-342): 16x16 blocks, every intra mode, SATD costs, sub-pel vectors, rate */
    if (tc->dispenser_search_level != 0 || tc->subsample_tx != 0 || tc->intra_mode_end > PAETH_PRED) return 0;
    if (tc->subpel_depth < QUARTER_PEL || tc->subpel_depth > FULL_PEL) return 0;
    if (tc->subpel_depth != FULL_PEL && tc->subpel_diag_refinement != 0 && tc->subpel_diag_refinement < 4) return 0; /* (1-3 scale org_error: no level selects them) */
    const EbPictureBufferDesc *inp = pcs->enhanced_pic; /* the block-edge geometry the device derives (init_xd_tpl, :403-416) must be the picture's */
    if (pcs->av1_cm->mi_rows != (int32_t)(((inp->height + 7) & ~7u) >> 2) || pcs->av1_cm->mi_cols != (int32_t)(pcs->aligned_width >> 2)) return 0;
    return 1;
}
/* the dispenser's qIndex: tpl_mc_flow_dispenser :1352-1372 (with enable_tpl_qps the picture's quantizer moves with its temporal layer) */
static int32_t tpl_q_index(const SequenceControlSet *scs, const PictureParentControlSet *pcs) {
    int32_t q = quantizer_to_qindex[(uint8_t)scs->static_config.qp];
    if (pcs->tpl_ctrls.enable_tpl_qps) {
        static const double rate[6][6] = {{1, 1, 1, 1, 1, 1}, {0.6, 1, 1, 1, 1, 1}, {0.6, 0.8, 1, 1, 1, 1}, {0.6, 0.8, 0.9, 1, 1, 1}, {0.35, 0.6, 0.8, 0.9, 1, 1},
                                           {0.35, 0.6, 0.8, 0.9, 0.95, 1}};
        const double q_val = svt_av1_convert_qindex_to_q(q, 8);
        q += pcs->tpl_data.tpl_slice_type == I_SLICE ? svt_av1_compute_qdelta(q_val, q_val * 0.25, 8)
                                                     : svt_av1_compute_qdelta(q_val, q_val * rate[pcs->hierarchical_levels][pcs->tpl_data.tpl_temporal_layer_index], 8);
    }
    return q;
}

/* the per-SB seam: nothing for a picture the device reconstructed, else the reference's function */
static void seam_tpl_sb(TPL_SB_ARGS) {
    if (TS.recon) { /* (every SB of every picture passes here from the dispenser threads: no lock) */
        int done = 0;
        for (int i = 0; i < 16; i++) done |= __atomic_load_n(&TS.done[i], __ATOMIC_ACQUIRE) == pcs;
        if (done) { __atomic_fetch_add(&TS.n_sb_calls_skipped, 1, __ATOMIC_RELAXED); return; }
    }
    tpl_mc_flow_dispenser_sb_generic_use0(TPL_SB_PASS);
}

/* The CONTENT id of a TPL reconstruction buffer: a picture's reconstruction is made again in a later TPL group (other quantizer, other references) into a buffer that may
 * be the same one, so (picture number) alone does not name what a buffer holds -- a device that mirrored the first version would serve it for the second (seen as a
 * bitstream difference with two devices: the rewriting device refreshes its own copy, the other one kept the old upload).  Every reconstruction the seam produces gets a
 * fresh serial; references look their buffer's current serial up; a buffer the seam did not write (a reference outside the sliding window, a declined picture) has none
 * and is uploaded per call. */
static struct { const void *buf; uint64_t id; } tpl_rec_ids[64];
static uint64_t tpl_rec_serial;
static uint64_t tpl_rec_id(const void *buf, int fresh) { /* (TS.lock is not held by the callers) */
    uint64_t id = 0;
    pthread_mutex_lock(&TS.lock);
    int slot = -1, spare = -1;
    for (int i = 0; i < 64; i++) {
        if (tpl_rec_ids[i].buf == buf) slot = i;
        if (!tpl_rec_ids[i].buf && spare < 0) spare = i;
    }
    if (fresh) {
        if (slot < 0) slot = spare >= 0 ? spare : (int)(tpl_rec_serial % 64);
        tpl_rec_ids[slot].buf = buf;
        tpl_rec_ids[slot].id  = id = (++tpl_rec_serial << 20) | 0x80000; /* (never equal to a picture number + 1 used for source planes of the same buffer address) */
    } else if (slot >= 0) id = tpl_rec_ids[slot].id;
    pthread_mutex_unlock(&TS.lock);
    return id;
}
static void tpl_rec_forget(const void *buf) {
    pthread_mutex_lock(&TS.lock);
    for (int i = 0; i < 64; i++)
        if (tpl_rec_ids[i].buf == buf) { tpl_rec_ids[i].buf = NULL; tpl_rec_ids[i].id = 0; }
    pthread_mutex_unlock(&TS.lock);
}
/* the reconstruction half of a whole picture on the device (src_ops_process.c:979-1198); st = the source-based statistics of every cell.  0 = done */
/* fused != NULL: the source-based half runs in the same call (svt_hip_tpl_stage_host) -- fused = the source half's host planes, tot / mvs / cand its ME tables, and
 * st receives the statistics instead of supplying them */
static int tpl_recon_picture(EncodeContext *enc_ctx, SequenceControlSet *scs, PictureParentControlSet *pcs, int32_t frame_idx, const SvtHipTplSrcParams *P,
                             SvtHipTplSrcStats *st, uint32_t cells, const SvtHipTplHostPlanes *fused, const uint8_t *tot, const uint32_t *mvs, const uint8_t *cand) {
    const EbPictureBufferDesc *inp = pcs->enhanced_pic;
    EbPictureBufferDesc       *rec = enc_ctx->mc_flow_rec_picture_buffer[frame_idx];
    SvtHipTplReconParams R;
    SvtHipTplHostPlanes  H;
    memset(&R, 0, sizeof(R));
    memset(&H, 0, sizeof(H));
    R.src = *P;
    R.recon_off = (uint64_t)rec->org_y * rec->stride_y + rec->org_x; R.recon_stride = rec->stride_y; R.is_ref = pcs->tpl_data.is_ref;
    H.src_buf = inp->buffer_y; H.src_rows = inp->luma_size / inp->stride_y;
    for (int l = 0; l < 2; l++)
        for (int r = 0; r < 4; r++) {
            if (!P->refs[l * 4 + r].valid) continue;
            const EbPictureBufferDesc *rp; /* (:1016-1023) */
            if (pcs->tpl_data.ref_in_slide_window[l][r]) {
                uint32_t k = 0;
                while (k < MAX_TPL_LA_SW && enc_ctx->poc_map_idx[k] != pcs->tpl_data.tpl_ref_ds_ptr_array[l][r].picture_number) k++;
                if (k == MAX_TPL_LA_SW) return -1;
                rp = enc_ctx->mc_flow_rec_picture_buffer[k];
            } else rp = (const EbPictureBufferDesc *)pcs->tpl_data.tpl_ref_ds_ptr_array[l][r].picture_ptr;
            if (!rp) return -1;
            SvtHipTplRef *Q = &R.rec_refs[l * 4 + r];
            Q->valid = 1; Q->plane_off = 0; Q->stride = rp->stride_y; Q->org_x = rp->org_x; Q->org_y = rp->org_y; Q->max_width = rp->max_width; Q->max_height = rp->max_height;
            H.ref_buf[l * 4 + r] = rp->buffer_y; H.ref_rows[l * 4 + r] = rp->luma_size / rp->stride_y;
        }
    SvtHipTplReconStats *out = malloc((size_t)cells * sizeof(*out));
    const double t0 = now_ms();
    /* the content of every buffer, by picture number: a picture of a TPL group is the source of one call and a reference of several others, and its TPL reconstruction
     * is produced on the device -- named, they stay there (one upload per call instead of up to nine) */
    SvtHipTplPlaneIds ids;
    SvtHipTplHostPlanes S;
    memset(&ids, 0, sizeof(ids));
    memset(&S, 0, sizeof(S));
    if (fused) S = *fused;
    else { S.src_buf = inp->buffer_y; S.src_rows = inp->luma_size / inp->stride_y; } /* the reconstruction half alone (stored statistics): the source is still read, its references are not */
    ids.src = pcs->picture_number + 1;
    for (int i = 0; i < 8; i++)
        if (P->refs[i].valid) { ids.src_ref[i] = fused ? P->refs[i].picture_number + 1 : 0; ids.rec_ref[i] = tpl_rec_id(H.ref_buf[i], 0); }
    ids.recon = tpl_rec_id(rec->buffer_y, 1);
    ids.recon_width = rec->width; ids.recon_height = rec->height; ids.recon_org_x = rec->org_x; ids.recon_org_y = rec->org_y;
    SEAM_ENV_ONCE(resident, (!(getenv("SVT_HIP_TPL_RESIDENT") && !atoi(getenv("SVT_HIP_TPL_RESIDENT"))))); /* SVT_HIP_TPL_RESIDENT=0: every call uploads its planes again (A/B and bisecting aid) */
    const int    rc = TS.fused_host ? TS.fused_host(&R, &S, &H, resident ? &ids : NULL, fused ? tot : NULL, fused ? mvs : NULL, fused ? cand : NULL, st, rec->buffer_y,
                                                    rec->luma_size / rec->stride_y, out)
                                    : TS.recon_host(&R, &H, st, rec->buffer_y, rec->luma_size / rec->stride_y, out);
    const double t1 = now_ms();
    if (rc) { free(out); return rc; }
    const uint32_t cols16 = (pcs->aligned_width + 15) >> 4, aligned_h = (inp->height + 7) & ~7u;
    uint64_t nb = 0, nc = 0;
    for (uint32_t i = 0; i < cells; i++) {
        if (!out[i].written) continue;
        const uint32_t x0 = (i % cols16) << 4, y0 = (i / cols16) << 4;
        const int complete = (pcs->aligned_width - (x0 & ~63u) >= 64) && (aligned_h - (y0 & ~63u) >= 64);
        TplStats ts;
        memset(&ts, 0, sizeof(ts));
        ts.srcrf_dist = out[i].srcrf_dist; ts.recrf_dist = out[i].recrf_dist; ts.srcrf_rate = out[i].srcrf_rate; ts.recrf_rate = out[i].recrf_rate;
        if (pcs->tpl_data.tpl_slice_type != I_SLICE && st[i].best_rf_idx != -1) { /* (:1182-1185) */
            ts.mv.row = st[i].mv_row; ts.mv.col = st[i].mv_col; ts.ref_frame_poc = st[i].ref_frame_poc;
        }
        result_model_store(pcs, &ts, x0, y0, (complete && P->dispenser_search_level) ? 32 : 16); /* the reference's own function (:266) */
        nb++; nc += out[i].coded;
    }
    free(out);
    pthread_mutex_lock(&TS.lock);
    TS.n_recon_pictures++; TS.n_recon_blocks += nb; TS.n_recon_coded += nc; TS.ms_recon += t1 - t0; TS.n_fused += fused != NULL;
    pthread_mutex_unlock(&TS.lock);
    return 0;
}
/* mark / unmark a picture for the per-SB seam */
static void tpl_recon_mark(PictureParentControlSet *pcs, int on) {
    pthread_mutex_lock(&TS.lock);
    int i = 0;
    while (i < 16 && TS.done[i] != (on ? NULL : pcs)) i++;
    if (i == 16 && on) { fprintf(stderr, "SVT_HIP_TPL_RECON_SEAM: more than 16 pictures inside the TPL dispenser at once\n"); abort(); }
    if (i < 16) __atomic_store_n(&TS.done[i], on ? pcs : NULL, __ATOMIC_RELEASE);
    pthread_mutex_unlock(&TS.lock);
}
/* the cells of the statistics grid that hold a block (the rule of :575-582 on the grid of :2048-2051), for statistics read back from the reference's own buffer */
static void tpl_written_cells(const PictureParentControlSet *pcs, uint8_t level, uint8_t *written) {
    const EbPictureBufferDesc *inp = pcs->enhanced_pic;
    const uint32_t cols16 = (pcs->aligned_width + 15) >> 4, aligned_h = (inp->height + 7) & ~7u, sbs_x = (pcs->aligned_width + 63) >> 6;
    for (uint32_t sb = 0; sb < pcs->b64_total_count; sb++) {
        const uint32_t sx = (sb % sbs_x) * 64, sy = (sb / sbs_x) * 64;
        const int      complete = (pcs->aligned_width - sx >= 64) && (aligned_h - sy >= 64);
        const uint32_t size = (complete && level) ? 32 : 16;
        for (uint32_t y0 = sy; y0 < sy + 64; y0 += size)
            for (uint32_t x0 = sx; x0 < sx + 64; x0 += size)
                if (x0 < pcs->aligned_width && y0 < aligned_h && !(x0 + (size >> 1) > inp->width || y0 + (size >> 1) > inp->height)) written[(y0 >> 4) * cols16 + (x0 >> 4)] = 1;
    }
}

static void tpl_mc_flow_dispenser_use2_body(TPL_DISP_ARGS) {
    if (!tpl_seam_on() || (pcs->tpl_src_data_ready && !TS.recon) || !tpl_seam_covers(scs, pcs)) {
        /* the reference's dispenser rewrites this picture's TPL reconstruction on the host: whatever a device holds of that buffer is stale from here on */
        if (TS.recon && enc_ctx->mc_flow_rec_picture_buffer[frame_idx]) {
            tpl_rec_forget(enc_ctx->mc_flow_rec_picture_buffer[frame_idx]->buffer_y);
            if (TS.plane_drop) TS.plane_drop(enc_ctx->mc_flow_rec_picture_buffer[frame_idx]->buffer_y);
        }
        if (TS.mode) {
            pthread_mutex_lock(&TS.lock);
            if (pcs->tpl_src_data_ready) TS.n_reused++; else TS.n_declined++;
            pthread_mutex_unlock(&TS.lock);
        }
        tpl_mc_flow_dispenser_use1(TPL_DISP_PASS);
        return;
    }
    const double t0 = now_ms();
    int          tpl_decline_rc = 0;
    const EbPictureBufferDesc *inp = pcs->enhanced_pic;
    MotionEstimationData      *med = pcs->pa_me_data;
    const int32_t q_index = tpl_q_index(scs, pcs);
    SvtHipTplSrcParams P;
    SvtHipTplHostPlanes H;
    memset(&P, 0, sizeof(P));
    memset(&H, 0, sizeof(H));
    P.width = inp->width; P.height = inp->height; P.aligned_width = pcs->aligned_width;
    P.sbs_x = (pcs->aligned_width + 63) >> 6; P.n_sb = pcs->b64_total_count;
    P.src_stride = inp->stride_y; P.src_off = (uint64_t)inp->org_y * inp->stride_y + inp->org_x;
    P.dispenser_search_level = pcs->tpl_ctrls.dispenser_search_level; P.subsample_tx = pcs->tpl_ctrls.subsample_tx; P.pf_shape = (uint8_t)pcs->tpl_ctrls.pf_shape;
    P.disable_intra_pred = pcs->tpl_ctrls.disable_intra_pred_nref && (pcs->temporal_layer_index == pcs->hierarchical_levels); /* :557 */
    P.i_slice = pcs->slice_type == I_SLICE;
    P.intra_mode_end = pcs->tpl_ctrls.intra_mode_end; P.search_flags = tpl_search_flags(&pcs->tpl_ctrls);
    P.enable_me_16x16 = pcs->enable_me_16x16; P.enable_me_8x8 = pcs->enable_me_8x8;
    P.max_cand = med->max_cand; P.max_refs = med->max_refs; P.max_l0 = med->max_l0;
    for (int i = 0; i < 2; i++) {
        P.quant_fp[i] = scs->enc_ctx->quants_8bit.y_quant_fp[q_index][i];
        P.round_fp[i] = scs->enc_ctx->quants_8bit.y_round_fp[q_index][i];
        P.dequant[i]  = scs->enc_ctx->deq_8bit.y_dequant_qtx[q_index][i];
    }
    H.src_buf = inp->buffer_y; H.src_rows = inp->luma_size / inp->stride_y;
    if (!P.i_slice)
        for (int l = 0; l < 2; l++) {
            const int cnt = l == 0 ? pcs->tpl_data.tpl_ref0_count : pcs->tpl_data.tpl_ref1_count;
            for (int r = 0; r < cnt && r < 4; r++) {
                const EbPictureBufferDesc *rp = (const EbPictureBufferDesc *)pcs->tpl_data.tpl_ref_ds_ptr_array[l][r].picture_ptr;
                SvtHipTplRef *R = &P.refs[l * 4 + r];
                if (!rp) continue;
                const int32_t grp = pcs->tpl_data.ref_tpl_group_idx[l][r];
                R->valid = !(grp > 0 && pcs->tpl_data.base_pcs->tpl_valid_pic[grp] == 0); /* :779-781 */
                R->plane_off = 0; R->picture_number = pcs->tpl_data.tpl_ref_ds_ptr_array[l][r].picture_number;
                R->stride = rp->stride_y; R->org_x = rp->org_x; R->org_y = rp->org_y; R->max_width = rp->max_width; R->max_height = rp->max_height;
                H.ref_buf[l * 4 + r] = rp->buffer_y; H.ref_rows[l * 4 + r] = rp->luma_size / rp->stride_y;
            }
        }
    const int stored = pcs->tpl_src_data_ready != 0; /* (reached with the reconstruction seam only) the source-based statistics of an earlier TPL group are kept: :969-977 */
    /* the picture's MeSbResults, SB after SB, in the flat layout of the C ABI */
    const uint32_t n_pus = pcs->enable_me_8x8 ? 85 : (pcs->enable_me_16x16 ? 21 : 5);
    uint8_t  *tot  = malloc((size_t)P.n_sb * n_pus);
    uint32_t *mvs  = malloc((size_t)P.n_sb * n_pus * (P.max_refs ? P.max_refs : 1) * 4);
    uint8_t  *cand = malloc((size_t)P.n_sb * n_pus * (P.max_cand ? P.max_cand : 1));
    for (uint32_t sb = 0; sb < P.n_sb; sb++) {
        const MeSbResults *m = med->me_results[sb];
        memcpy(tot + (size_t)sb * n_pus, m->total_me_candidate_index, n_pus);
        memcpy(mvs + (size_t)sb * n_pus * P.max_refs, m->me_mv_array, (size_t)n_pus * P.max_refs * 4);
        memcpy(cand + (size_t)sb * n_pus * P.max_cand, m->me_candidate_array, (size_t)n_pus * P.max_cand);
    }
    const uint32_t cols16 = (pcs->aligned_width + 15) >> 4, rows16 = (((inp->height + 7) & ~7u) + 15) >> 4, cells = cols16 * rows16;
    SvtHipTplSrcStats *st = malloc((size_t)cells * sizeof(*st));
    svt_hip_seam_bind(pcs->picture_number);
    int fused_done = 0;
    if (stored) {
        uint8_t *wr = calloc(cells, 1);
        tpl_written_cells(pcs, P.dispenser_search_level, wr);
        memset(st, 0, (size_t)cells * sizeof(*st));
        for (uint32_t i = 0; i < cells; i++) {
            if (!wr[i]) continue;
            const TplSrcStats *d = &med->tpl_src_stats_buffer[i];
            st[i].written = 1; st[i].srcrf_dist = d->srcrf_dist; st[i].srcrf_rate = d->srcrf_rate; st[i].ref_frame_poc = d->ref_frame_poc;
            st[i].mv_row = d->mv.row; st[i].mv_col = d->mv.col; st[i].best_mode = d->best_mode; st[i].best_rf_idx = d->best_rf_idx; st[i].best_intra_mode = (uint8_t)d->best_intra_mode;
        }
        free(wr);
    } else if (TS.recon && TS.fused_host) { /* both halves in one device call: one upload of every picture buffer, one synchronisation */
        const int rc = tpl_recon_picture(enc_ctx, scs, pcs, frame_idx, &P, st, cells, &H, tot, mvs, cand);
        if (rc) { tpl_decline_rc = rc; goto decline; }
        fused_done = 1;
    } else if ((tpl_decline_rc = TS.stage_host(&P, &H, tot, mvs, cand, st)) != 0) goto decline;
    /* into the buffer the reference's own "already computed" branch reads (:969-977); a sequence without stored statistics (tpl_lad_mg == 0) has none: lend one */
    TplSrcStats *own = med->tpl_src_stats_buffer, *buf = own;
    const uint32_t ref_cells = ((pcs->aligned_width + 15) >> 4) * ((inp->height + 15) >> 4 > rows16 ? (inp->height + 15) >> 4 : rows16);
    if (!buf) buf = calloc(ref_cells, sizeof(*buf));
    uint64_t nb = 0, nn = 0;
    for (uint32_t i = 0; i < cells && !stored; i++) {
        if (!st[i].written) continue;
        TplSrcStats *d = &buf[i];
        d->srcrf_dist = st[i].srcrf_dist; d->srcrf_rate = st[i].srcrf_rate; d->ref_frame_poc = st[i].ref_frame_poc;
        d->mv.row = st[i].mv_row; d->mv.col = st[i].mv_col; d->best_mode = st[i].best_mode; d->best_rf_idx = st[i].best_rf_idx;
        d->best_intra_mode = (PredictionMode)st[i].best_intra_mode;
        nb++; nn += st[i].best_mode == NEWMV;
    }
    const double t1 = now_ms();
    int on_device = 0;
    if (TS.recon) { /* the reconstruction half too: the per-SB function of this picture becomes a no-op */
        const int rc = fused_done ? 0 : tpl_recon_picture(enc_ctx, scs, pcs, frame_idx, &P, st, cells, NULL, NULL, NULL, NULL);
        if (rc) { /* (the statistics already copied into `buf` are the values the reference's own first half writes there again) */
            if (!own) free(buf);
            tpl_decline_rc = rc;
            goto decline;
        }
        tpl_recon_mark(pcs, 1);
        on_device = 1;
    }
    med->tpl_src_stats_buffer = buf;
    pcs->tpl_src_data_ready   = 1;
    tpl_mc_flow_dispenser_use1(TPL_DISP_PASS); /* the reference's dispenser: segments, (reconstruction half, result_model_store,) padding of the reconstruction */
    pcs->tpl_src_data_ready   = (uint8_t)stored; /* tpl_mc_flow sets it itself when the statistics are kept (:1856-1858) */
    if (on_device) tpl_recon_mark(pcs, 0);
    med->tpl_src_stats_buffer = own;
    if (!own) free(buf);
    free(tot); free(mvs); free(cand); free(st);
    pthread_mutex_lock(&TS.lock);
    if (stored) TS.n_reused++;
    else { TS.n_pictures++; TS.n_blocks += nb; TS.n_newmv += nn; TS.ms_stage += t1 - t0; }
    pthread_mutex_unlock(&TS.lock);
    return;
decline:
    /* A stage call returned non-zero -- the device path is off (svtav1_hip.h, error policy), a block gave up waiting for a neighbour (-4), or the parameters were
     * refused: this picture takes the reference's dispenser, both halves, as if the seam were off.  Nothing of the picture's state has been changed; what a device
     * holds of the picture's TPL reconstruction buffer is stale from here on (the reference rewrites it on the host). */
    fprintf(stderr, "SVT_HIP_TPL_SEAM: a stage call returned %d for picture %llu: the reference's dispenser takes it\n", tpl_decline_rc, (unsigned long long)pcs->picture_number);
    free(tot); free(mvs); free(cand); free(st);
    if (TS.recon && enc_ctx->mc_flow_rec_picture_buffer[frame_idx]) {
        tpl_rec_forget(enc_ctx->mc_flow_rec_picture_buffer[frame_idx]->buffer_y);
        if (TS.plane_drop) TS.plane_drop(enc_ctx->mc_flow_rec_picture_buffer[frame_idx]->buffer_y);
    }
    pthread_mutex_lock(&TS.lock);
    TS.n_declined++;
    pthread_mutex_unlock(&TS.lock);
    tpl_mc_flow_dispenser_use1(TPL_DISP_PASS);
}
static void tpl_mc_flow_dispenser_use2(TPL_DISP_ARGS) {
    SEAM_CPU_BEGIN();
    tpl_mc_flow_dispenser_use2_body(TPL_DISP_PASS);
    SEAM_CPU_END(SEAM_CPU_TPL);
}

