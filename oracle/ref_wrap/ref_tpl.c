/*
 * ref_tpl.c -- TEST INFRASTRUCTURE, built only into oracle/_ref/libsvtref_me.so (make -C oracle ref).
 *
 * The TPL dispenser's per-SB function tpl_mc_flow_dispenser_sb_generic (Codec/src_ops_process.c:519-1198) is `static`.  This translation unit compiles that
 * reference source file WHERE IT LIES (the #include below resolves through -I$(REF)/Source/Lib/Codec; nothing is copied) and adds one plain-C entry point that
 * builds the handful of structures the function reads -- SequenceControlSet (b64_geom, enc_ctx with the 8-bit quantizer tables of svt_av1_build_quantizer),
 * PictureParentControlSet (enhanced_pic, tpl_ctrls, tpl_data with the reference pictures, pa_me_data with the MeSbResults) -- and runs it for every 64x64
 * SB of a picture in raster order, exactly as svt_aom_tpl_disp_kernel's no-segment path does (:2060-2075: incomplete SBs run at dispenser level 0).
 * Outputs: the source-based statistics the function stores per 16x16 cell (TplSrcStats, :958-967: what tests/test_tpl.py pins the oracle and the device
 * stage on) and, for the whole-dispenser pin, the final TplStats grid and the reconstructed TPL picture.
 */
#include "src_ops_process.c"
#include "md_config_process.h"
void init_fn_ptr(void);
void svt_av1_init_me_luts(void);

typedef struct RefTplRef { /* = SvtHipTplRef */
    uint64_t plane_off, picture_number;
    uint32_t stride, org_x, org_y;
    uint16_t max_width, max_height;
    uint8_t  valid, pad[3];
} RefTplRef;
typedef struct RefTplParams { /* = SvtHipTplSrcParams (include/svtav1_hip.h) */
    uint32_t width, height, aligned_width, sbs_x, n_sb;
    uint32_t src_stride;
    uint64_t src_off; /* of picture sample (0, 0) */
    uint8_t  dispenser_search_level, subsample_tx, pf_shape, disable_intra_pred, i_slice, enable_me_16x16, enable_me_8x8, max_cand, max_refs, max_l0;
    uint8_t  intra_mode_end; /* tpl_ctrls.intra_mode_end: DC_PRED (0) .. PAETH_PRED (12) */
    uint8_t  search_flags;   /* bit 0: !use_sad_in_src_search (transform + SATD costs), bit 1: compute_rate, bits 2-3: sub-pel rounds (0 FULL_PEL, 1 HALF_PEL,
                                2 QUARTER_PEL), bit 4: subpel_diag_refinement = 4 */
    int16_t  quant_fp[2], round_fp[2], dequant[2];
    RefTplRef refs[8];
} RefTplParams;
typedef struct RefTplSrcStats { /* = SvtHipTplSrcStats */
    int64_t  srcrf_dist, srcrf_rate;
    uint64_t ref_frame_poc;
    int16_t  mv_row, mv_col;
    int32_t  best_rf_idx;
    uint8_t  best_mode, best_intra_mode, written, pad[5];
} RefTplSrcStats;

_Static_assert(sizeof(RefTplSrcStats) == 40 && sizeof(RefTplParams) == 376, "layout");
/* q_index: the dispenser's qIndex (the quantizer tables are built here for every index, as initial_rc_process.c:754 does, and returned in P_out->quant_fp / round_fp /
 * dequant for the caller to hand to the oracle / the device stage).  src_org_x / y: enhanced_pic->org_x / org_y (P->src_off must equal org_y * stride + org_x).
 * recon: a (width + 2 * 32... ) scratch plane is allocated inside.  tpl_stats_out: [rows16][cols16] of 8 int64 (TplStats without padding assumptions), or NULL. */
void ref_tpl_dispenser_picture(RefTplParams *P, int q_index, const uint8_t *src_buf, uint32_t src_org_x, uint32_t src_org_y, const uint8_t *ref_base,
                               const uint8_t *total_me_candidate_index, const uint32_t *me_mv_array, const uint8_t *me_candidate_array, uint32_t n_pus,
                               RefTplSrcStats *src_stats_out, int64_t *tpl_stats_out, uint8_t *recon_out /* [height][width] or NULL */) {
    static int rtcd_done;
    if (!rtcd_done) {
        svt_aom_setup_common_rtcd_internal(0);
        svt_aom_setup_rtcd_internal(0);
        svt_aom_asm_set_convolve_asm_table(); /* the initialisations svt_av1_enc_init runs after the RTCD setup (enc_handle.c:1447-1453) */
        svt_aom_init_intra_dc_predictors_c_internal();
        svt_aom_asm_set_convolve_hbd_asm_table();
        svt_aom_init_intra_predictors_internal();
        init_fn_ptr();          /* svt_aom_mefn_ptr: the variance functions of the sub-pel search (enc_handle.c:1460) */
        svt_av1_init_me_luts(); /* sad_per_bit tables read by svt_tpl_init_mv_cost_params (enc_handle.c:1459) */
        rtcd_done = 1;
    }
    SequenceControlSet      *scs  = calloc(1, sizeof(*scs));
    EncodeContext           *enc  = calloc(1, sizeof(*enc));
    PictureParentControlSet *pcs  = calloc(1, sizeof(*pcs));
    PictureParentControlSet *base = calloc(1, sizeof(*base));
    MotionEstimationData    *med  = calloc(1, sizeof(*med));
    Av1Common               *cm   = calloc(1, sizeof(*cm));
    EbPictureBufferDesc     *inp = calloc(1, sizeof(*inp)), *rec = calloc(1, sizeof(*rec)), *refp[8];
    scs->enc_ctx = enc;
    pcs->scs     = scs;
    pcs->av1_cm  = cm;
    pcs->pa_me_data = med;
    /* quantizer tables, 8 bit (initial_rc_process.c:754) */
    svt_av1_build_quantizer(pcs, EB_EIGHT_BIT, 0, 0, 0, 0, 0, &enc->quants_8bit, &enc->deq_8bit);
    for (int i = 0; i < 2; i++) {
        P->quant_fp[i] = enc->quants_8bit.y_quant_fp[q_index][i];
        P->round_fp[i] = enc->quants_8bit.y_round_fp[q_index][i];
        P->dequant[i]  = enc->deq_8bit.y_dequant_qtx[q_index][i];
    }
    scs->in_loop_ois = 1;
    scs->tpl_lad_mg  = 1; /* > 0: the source-based statistics are stored (:958) */
    scs->max_input_luma_width  = (uint16_t)P->width;
    scs->max_input_luma_height = (uint16_t)P->height;
    scs->b64_size = 64;
    /* b64 geometry as svt_aom_b64_geom_init (pcs.c) derives it from the aligned picture */
    const uint32_t aligned_h = (P->height + 7) & ~7u;
    const uint32_t sbs_y     = P->n_sb / P->sbs_x;
    scs->b64_geom = calloc(P->n_sb, sizeof(B64Geom));
    for (uint32_t i = 0; i < P->n_sb; i++) {
        B64Geom *g = &scs->b64_geom[i];
        g->horizontal_index = (uint8_t)(i % P->sbs_x);
        g->vertical_index   = (uint8_t)(i / P->sbs_x);
        g->org_x  = (uint16_t)(g->horizontal_index * 64);
        g->org_y  = (uint16_t)(g->vertical_index * 64);
        g->width  = (uint8_t)((P->aligned_width - g->org_x) < 64 ? (P->aligned_width - g->org_x) : 64);
        g->height = (uint8_t)((aligned_h - g->org_y) < 64 ? (aligned_h - g->org_y) : 64);
    }
    (void)sbs_y;
    /* pictures */
    inp->buffer_y = (uint8_t *)src_buf; inp->stride_y = (uint16_t)P->src_stride; inp->org_x = (uint16_t)src_org_x; inp->org_y = (uint16_t)src_org_y;
    inp->width = (uint16_t)P->width; inp->height = (uint16_t)P->height; inp->max_width = (uint16_t)P->width; inp->max_height = (uint16_t)P->height;
    pcs->enhanced_pic  = inp;
    pcs->aligned_width = (uint16_t)P->aligned_width;
    pcs->aligned_height = (uint16_t)aligned_h;
    const uint32_t rpad = 32 + 64, rstride = P->width + 2 * rpad + 64;
    uint8_t *rec_buf = calloc((size_t)rstride * (P->height + 2 * rpad + 64), 1);
    rec->buffer_y = rec_buf; rec->stride_y = (uint16_t)rstride; rec->org_x = (uint16_t)rpad; rec->org_y = (uint16_t)rpad; rec->width = (uint16_t)P->width;
    rec->height = (uint16_t)P->height; rec->max_width = (uint16_t)P->width; rec->max_height = (uint16_t)P->height;
    enc->mc_flow_rec_picture_buffer[0] = rec;
    enc->poc_map_idx[0] = 1000;
    cm->mi_rows = (int32_t)(aligned_h >> 2);
    cm->mi_cols = (int32_t)(P->aligned_width >> 2);
    /* controls: the set the device stage covers (tpl levels 4 and 5 of set_tpl_params, initial_rc_process.c:331-378) */
    TplControls *tc = &pcs->tpl_ctrls;
    tc->enable = 1; tc->enable_tpl_qps = 0;
    /* ... and, with intra_mode_end / search_flags set, the rest of set_tpl_params' levels (0-3: :301-342) */
    tc->intra_mode_end = P->intra_mode_end; tc->use_sad_in_src_search = !(P->search_flags & 1); tc->compute_rate = (P->search_flags >> 1) & 1;
    tc->subpel_depth = (SUBPEL_FORCE_STOP)(FULL_PEL - ((P->search_flags >> 2) & 3)); tc->subpel_diag_refinement = (P->search_flags & 16) ? 4 : 0;
    scs->static_config.qp = 35; pcs->update_type = SVT_AV1_ARF_UPDATE; /* read by tpl_subpel_search's MV-cost set-up only (MV_COST_NONE: no effect on the result) */
    svt_av1_setup_scale_factors_for_frame(&scs->sf_identity, P->width, P->height, P->width, P->height);
    tc->disable_intra_pred_nref = P->disable_intra_pred; /* with temporal_layer_index == hierarchical_levels below */
    tc->pf_shape = (EB_TRANS_COEFF_SHAPE)P->pf_shape;
    tc->dispenser_search_level = P->dispenser_search_level;
    tc->subsample_tx = P->subsample_tx;
    tc->synth_blk_size = 16;
    pcs->temporal_layer_index = 3; pcs->hierarchical_levels = 3;
    pcs->slice_type = P->i_slice ? I_SLICE : B_SLICE;
    pcs->tpl_data.tpl_slice_type = pcs->slice_type;
    pcs->tpl_data.is_ref = 1;
    pcs->tpl_data.base_pcs = base;
    pcs->enable_me_16x16 = P->enable_me_16x16;
    pcs->tpl_src_data_ready = 0;
    pcs->b64_total_count = (uint16_t)P->n_sb;
    for (int l = 0; l < 2; l++)
        for (int r = 0; r < 4; r++) {
            const RefTplRef *R = &P->refs[l * 4 + r];
            EbPictureBufferDesc *d = refp[l * 4 + r] = calloc(1, sizeof(*d));
            d->buffer_y = (uint8_t *)ref_base + R->plane_off; d->stride_y = (uint16_t)R->stride; d->org_x = (uint16_t)R->org_x; d->org_y = (uint16_t)R->org_y;
            d->width = d->max_width = R->max_width; d->height = d->max_height = R->max_height;
            pcs->tpl_data.tpl_ref_ds_ptr_array[l][r].picture_ptr    = d;
            pcs->tpl_data.tpl_ref_ds_ptr_array[l][r].picture_number = R->picture_number;
            /* an excluded reference (:779-781): inside the sliding window at group index 1 + slot, marked invalid there */
            pcs->tpl_data.ref_tpl_group_idx[l][r]   = R->valid ? -1 : 1 + l * 4 + r;
            pcs->tpl_data.ref_in_slide_window[l][r] = false;
            base->tpl_valid_pic[1 + l * 4 + r]      = 0;
        }
    /* ME results: MeSbResults per SB pointing into the caller's flat arrays */
    MeSbResults *res = calloc(P->n_sb, sizeof(MeSbResults));
    med->me_results  = calloc(P->n_sb, sizeof(MeSbResults *));
    for (uint32_t i = 0; i < P->n_sb; i++) {
        res[i].total_me_candidate_index = (uint8_t *)total_me_candidate_index + (size_t)i * n_pus;
        res[i].me_mv_array              = (MvCandidate *)me_mv_array + (size_t)i * n_pus * P->max_refs;
        res[i].me_candidate_array       = (MeCandidate *)me_candidate_array + (size_t)i * n_pus * P->max_cand;
        med->me_results[i]              = &res[i];
    }
    med->max_cand = P->max_cand; med->max_refs = P->max_refs; med->max_l0 = P->max_l0;
    const uint32_t cols16 = (P->aligned_width + 15) >> 4, rows16 = (aligned_h + 15) >> 4, cells = cols16 * (rows16 + 1);
    med->tpl_src_stats_buffer = calloc(cells, sizeof(TplSrcStats));
    memset(med->tpl_src_stats_buffer, 0xA5, (size_t)cells * sizeof(TplSrcStats)); /* cells the function skips stay recognisable */
    TplStats *stats = calloc(cells, sizeof(TplStats));
    med->tpl_stats  = calloc(cells, sizeof(TplStats *));
    for (uint32_t i = 0; i < cells; i++) med->tpl_stats[i] = &stats[i];

    for (uint32_t sb = 0; sb < P->n_sb; sb++) {
        const B64Geom *g = &scs->b64_geom[sb];
        tpl_mc_flow_dispenser_sb_generic(enc, scs, pcs, 0, sb, q_index, (g->width == 64 && g->height == 64) ? tc->dispenser_search_level : 0);
    }
    for (uint32_t i = 0; i < cols16 * rows16; i++) {
        const TplSrcStats *s = &med->tpl_src_stats_buffer[i];
        RefTplSrcStats    *o = &src_stats_out[i];
        memset(o, 0, sizeof(*o));
        const uint8_t *raw = (const uint8_t *)s;
        int untouched = 1;
        for (size_t b = 0; b < 8; b++) untouched &= raw[b] == 0xA5; /* srcrf_dist never equals 0xA5A5... when written (it is < 2^40) */
        if (untouched) continue;
        o->written = 1;
        o->srcrf_dist = s->srcrf_dist; o->srcrf_rate = s->srcrf_rate; o->ref_frame_poc = s->ref_frame_poc; o->mv_row = s->mv.row; o->mv_col = s->mv.col;
        o->best_rf_idx = s->best_rf_idx; o->best_mode = s->best_mode; o->best_intra_mode = s->best_intra_mode;
    }
    if (tpl_stats_out)
        for (uint32_t i = 0; i < cols16 * rows16; i++) {
            const TplStats *t = &stats[i];
            int64_t *o = tpl_stats_out + (size_t)i * 8;
            o[0] = t->srcrf_dist; o[1] = t->recrf_dist; o[2] = t->srcrf_rate; o[3] = t->recrf_rate; o[4] = t->mc_dep_rate; o[5] = t->mc_dep_dist;
            o[6] = ((int64_t)t->mv.row << 16) | (uint16_t)t->mv.col; o[7] = (int64_t)t->ref_frame_poc;
        }
    if (recon_out)
        for (uint32_t y = 0; y < P->height; y++) memcpy(recon_out + (size_t)y * P->width, rec_buf + (size_t)(y + rpad) * rstride + rpad, P->width);
    for (int i = 0; i < 8; i++) free(refp[i]);
    free(res); free(med->me_results); free(med->tpl_src_stats_buffer); free(stats); free(med->tpl_stats); free(rec_buf); free(scs->b64_geom);
    free(scs); free(enc); free(pcs); free(base); free(med); free(cm); free(inp); free(rec);
}
