/*
 * ref_static_me.c -- TEST INFRASTRUCTURE, built only into oracle/_ref/libsvtref_me.so (make -C oracle ref).
 *
 * The reference's ME result formatting (me_prune_ref, construct_me_candidate_array{,_mrp_off,_single_ref}, compute_distortion,
 * perform_gm_detection; Codec/motion_estimation.c:1522-1566, 2532-3008) is `static`, so it cannot be called by symbol from
 * libsvtref.so.  This translation unit compiles that reference source file WHERE IT LIES (the #include below resolves through
 * -I$(REF)/Source/Lib/Codec; nothing is copied) and adds one plain-C entry point that fills the handful of context fields those
 * functions read and runs them in the order of svt_aom_motion_estimation_b64 (motion_estimation.c:3121-3152).
 */
#include "motion_estimation.c"

typedef struct RefMeResultsParams { /* same layout as SvtHipMeResultsParams (include/svtav1_hip.h) */
    uint32_t n_sb;
    uint8_t  num_of_list_to_search, num_of_ref_pic_to_search[2];
    uint8_t  max_cand, max_refs, max_l0;
    uint8_t  enable_me_16x16, enable_me_8x8, only_l_bwd, use_best_unipred_cand_only;
    uint8_t  prune_ref, low_resolution, gm_enabled, gm_use_distance_based_active_th;
    uint16_t prune_ref_if_me_sad_dev_bigger_than_th;
    int32_t  prune_me_candidates_th;
    uint64_t picture_number;
    uint64_t ref_picture_number[2][4];
} RefMeResultsParams;

typedef struct RefMeSbStats { /* = SvtHipMeSbStats */
    uint32_t me_64x64_distortion, me_32x32_distortion, me_16x16_distortion, me_8x8_distortion, me_8x8_cost_variance, rc_me_distortion;
    uint8_t  stationary_block_present_sb, rc_me_allow_gm, pad[2];
} RefMeSbStats;

void ref_me_results_sb(const RefMeResultsParams *P, const uint32_t *best_sad /*[2][4][85]*/, const uint32_t *best_mv, uint8_t *do_ref /*[2][4] in/out*/,
                       uint8_t sb_width, uint8_t sb_height, uint8_t *total_me_candidate_index, uint32_t *me_mv_array, uint8_t *me_candidate_array,
                       RefMeSbStats *st) {
    static PictureParentControlSet *pcs;
    static SequenceControlSet      *scs;
    static MeContext               *ctx;
    static MotionEstimationData    *med;
    static MeSbResults             *res, *res_tab[1];
    static B64Geom                  geom;
    static uint32_t                 u32s[6];
    static uint8_t                  u8s[2];
    if (!pcs) {
        pcs = calloc(1, sizeof(*pcs));
        scs = calloc(1, sizeof(*scs));
        ctx = calloc(1, sizeof(*ctx));
        med = calloc(1, sizeof(*med));
        res = calloc(1, sizeof(*res));
    }
    pcs->scs = scs;
    pcs->pa_me_data = med;
    res_tab[0] = res;
    med->me_results = res_tab;
    med->max_cand = P->max_cand; med->max_refs = P->max_refs; med->max_l0 = P->max_l0;
    res->total_me_candidate_index = total_me_candidate_index;
    res->me_mv_array              = (MvCandidate *)me_mv_array;
    res->me_candidate_array       = (MeCandidate *)me_candidate_array;
    pcs->enable_me_16x16 = P->enable_me_16x16; pcs->enable_me_8x8 = P->enable_me_8x8;
    pcs->max_number_of_pus_per_sb = SQUARE_PU_COUNT; /* resource_coordination_process.c:425 */
    scs->mrp_ctrls.only_l_bwd = P->only_l_bwd;
    scs->input_resolution = P->low_resolution ? INPUT_SIZE_480p_RANGE : INPUT_SIZE_1080p_RANGE;
    geom.width = sb_width; geom.height = sb_height;
    pcs->b64_geom = &geom;
    pcs->me_64x64_distortion = &u32s[0]; pcs->me_32x32_distortion = &u32s[1]; pcs->me_16x16_distortion = &u32s[2];
    pcs->me_8x8_distortion = &u32s[3]; pcs->me_8x8_cost_variance = &u32s[4]; pcs->rc_me_distortion = &u32s[5];
    pcs->stationary_block_present_sb = &u8s[0]; pcs->rc_me_allow_gm = &u8s[1];
    pcs->gm_ctrls.enabled = P->gm_enabled; pcs->gm_ctrls.use_distance_based_active_th = P->gm_use_distance_based_active_th;
    pcs->picture_number = P->picture_number;

    ctx->num_of_list_to_search = P->num_of_list_to_search;
    ctx->num_of_ref_pic_to_search[0] = P->num_of_ref_pic_to_search[0]; ctx->num_of_ref_pic_to_search[1] = P->num_of_ref_pic_to_search[1];
    ctx->me_hme_prune_ctrls.enable_me_hme_ref_pruning = P->prune_ref;
    ctx->me_hme_prune_ctrls.prune_ref_if_me_sad_dev_bigger_than_th = P->prune_ref_if_me_sad_dev_bigger_than_th;
    ctx->prune_me_candidates_th = P->prune_me_candidates_th;
    ctx->use_best_unipred_cand_only = P->use_best_unipred_cand_only;
    memcpy(ctx->p_sb_best_sad, best_sad, sizeof(ctx->p_sb_best_sad));
    memcpy(ctx->p_sb_best_mv, best_mv, sizeof(ctx->p_sb_best_mv));
    for (int l = 0; l < 2; l++)
        for (int r = 0; r < 4; r++) {
            ctx->search_results[l][r].do_ref  = do_ref[l * 4 + r];
            ctx->search_results[l][r].hme_sad = MAX_U32; /* init_me_hme_data, motion_estimation.c:3061 (refs HME never touched) */
            ctx->me_ds_ref_array[l][r].picture_number = P->ref_picture_number[l][r];
        }
    /* order of svt_aom_motion_estimation_b64, motion_estimation.c:3121-3152 */
    if (P->prune_ref) me_prune_ref(ctx);
    if (ctx->num_of_ref_pic_to_search[0] == 1 && ctx->num_of_ref_pic_to_search[1] == 0)
        construct_me_candidate_array_single_ref(pcs, ctx, P->num_of_list_to_search, 0);
    else if (ctx->num_of_ref_pic_to_search[0] == 1 && ctx->num_of_ref_pic_to_search[1] == 1)
        construct_me_candidate_array_mrp_off(pcs, ctx, P->num_of_list_to_search, 0);
    else
        construct_me_candidate_array(pcs, ctx, P->num_of_list_to_search, 0);
    compute_distortion(pcs, 0, ctx);
    pcs->stationary_block_present_sb[0] = 0;
    pcs->rc_me_allow_gm[0]              = 0;
    if (pcs->gm_ctrls.enabled) perform_gm_detection(pcs, 0, ctx);

    for (int l = 0; l < 2; l++)
        for (int r = 0; r < 4; r++) do_ref[l * 4 + r] = ctx->search_results[l][r].do_ref;
    st->me_64x64_distortion = u32s[0]; st->me_32x32_distortion = u32s[1]; st->me_16x16_distortion = u32s[2];
    st->me_8x8_distortion = u32s[3]; st->me_8x8_cost_variance = u32s[4]; st->rc_me_distortion = u32s[5];
    st->stationary_block_present_sb = u8s[0]; st->rc_me_allow_gm = u8s[1];
    st->pad[0] = st->pad[1] = 0;
}

/* One HME level through the reference's own leaf drivers hme_level_0 / hme_level_1 (static) / hme_level_2, which call svt_sad_loop_kernel through the
 * RTCD pointer (set it up with svt_aom_setup_rtcd_internal first).  Same plain arguments as oracle_hme_level. */
void ref_hme_level(int level, int sub_sampled, int num_hme_sa_w, int num_hme_sa_h, int sr_w, int sr_h, uint8_t *src, uint32_t src_stride,
                   uint8_t *ref_plane, uint32_t ref_stride, int ref_org_x, int ref_org_y, int ref_width, int ref_height, int16_t org_x, int16_t org_y,
                   uint32_t block_width, uint32_t block_height, int16_t sa_width, int16_t sa_height, int16_t prev_sc_x, int16_t prev_sc_y,
                   uint64_t *best_sad, int16_t *sc_x, int16_t *sc_y) {
    static MeContext   *ctx;
    EbPictureBufferDesc pic;
    if (!ctx) ctx = calloc(1, sizeof(*ctx));
    memset(&pic, 0, sizeof(pic));
    pic.buffer_y = ref_plane; pic.stride_y = (uint16_t)ref_stride; pic.org_x = (uint16_t)ref_org_x; pic.org_y = (uint16_t)ref_org_y;
    pic.width = (uint16_t)ref_width; pic.height = (uint16_t)ref_height;
    ctx->num_hme_sa_w = (uint16_t)num_hme_sa_w; ctx->num_hme_sa_h = (uint16_t)num_hme_sa_h;
    ctx->hme_search_method = sub_sampled ? SUB_SAD_SEARCH : FULL_SAD_SEARCH;
    ctx->sixteenth_b64_buffer = src; ctx->sixteenth_b64_buffer_stride = src_stride;
    ctx->quarter_b64_buffer = src; ctx->quarter_b64_buffer_stride = src_stride;
    ctx->b64_src_ptr = src; ctx->b64_src_stride = src_stride;
    if (level == 0) hme_level_0(ctx, org_x, org_y, block_width, block_height, sa_width, sa_height, &pic, (uint32_t)sr_w, (uint32_t)sr_h, best_sad, sc_x, sc_y);
    else if (level == 1) hme_level_1(ctx, org_x, org_y, block_width, block_height, &pic, sa_width, sa_height, prev_sc_x, prev_sc_y, best_sad, sc_x, sc_y);
    else hme_level_2(ctx, org_x, org_y, block_width, block_height, &pic, sa_width, sa_height, prev_sc_x, prev_sc_y, best_sad, sc_x, sc_y);
}

/* set_final_seach_centre_sb + integer_search_b64 (static) for ONE reference slot of one SB, with the option set the restatement covers
 * (no early exit, is_ref = 0, no search-range adjustment pass, no 8x8-variance probe).  HME results enter as the level-2 arrays. */
typedef struct RefIntSearch { /* = OracleIntSearch */
    int16_t  sa_min_w, sa_min_h, sa_max_w, sa_max_h;
    uint16_t dist;
    uint8_t  mv_adj_enabled, mv_adj_nearest_ref_only, ref_pic_index, sub_sad;
    uint16_t mv_size_th, sa_multiplier;
    uint32_t divisor;
} RefIntSearch;
void ref_me_integer_search(const RefIntSearch *P, int num_sa_w, int num_sa_h, const uint64_t *hme_sad, const int16_t *hme_sc, uint8_t *src_plane,
                           uint32_t src_stride, int src_org_x, int src_org_y, uint8_t *ref_plane, uint32_t ref_stride, int ref_org_x, int ref_org_y,
                           int pic_w, int pic_h, int b64_origin_x, int b64_origin_y, int picture_width, int picture_height, int16_t *sc_out,
                           uint64_t *sad_out, uint32_t *best_sad, uint32_t *best_mv) {
    static MeContext               *ctx;
    static PictureParentControlSet *pcs;
    static EbPictureBufferDesc      refp, inp;
    if (!ctx) { ctx = calloc(1, sizeof(*ctx)); pcs = calloc(1, sizeof(*pcs)); }
    const int l = 0, r = P->ref_pic_index;
    memset(ctx->search_results, 0, sizeof(ctx->search_results));
    ctx->num_of_list_to_search = 1;
    ctx->num_of_ref_pic_to_search[0] = (uint8_t)(r + 1); ctx->num_of_ref_pic_to_search[1] = 0;
    ctx->temporal_layer_index = 1;
    ctx->enable_hme_flag = 1; ctx->enable_hme_level0_flag = 1; ctx->enable_hme_level1_flag = 1; ctx->enable_hme_level2_flag = 1;
    ctx->num_hme_sa_w = (uint16_t)num_sa_w; ctx->num_hme_sa_h = (uint16_t)num_sa_h;
    for (int rr = 0; rr <= r; rr++) { /* earlier slots: not searched (do_ref = 0) */
        ctx->search_results[l][rr].do_ref = rr == r;
        ctx->reduce_me_sr_divisor[l][rr]  = P->divisor;
        for (int h = 0; h < num_sa_h; h++)
            for (int w = 0; w < num_sa_w; w++) {
                ctx->hme_level2_sad[l][rr][w][h]             = hme_sad[h * num_sa_w + w];
                ctx->x_hme_level2_search_center[l][rr][w][h] = hme_sc[2 * (h * num_sa_w + w)];
                ctx->y_hme_level2_search_center[l][rr][w][h] = hme_sc[2 * (h * num_sa_w + w) + 1];
            }
        ctx->me_ds_ref_array[l][rr].picture_ptr    = &refp;
        ctx->me_ds_ref_array[l][rr].picture_number = 100;
    }
    set_final_seach_centre_sb(pcs, ctx);
    sc_out[0] = ctx->search_results[l][r].hme_sc_x; sc_out[1] = ctx->search_results[l][r].hme_sc_y; *sad_out = ctx->search_results[l][r].hme_sad;

    memset(&refp, 0, sizeof(refp)); memset(&inp, 0, sizeof(inp));
    refp.buffer_y = ref_plane; refp.stride_y = (uint16_t)ref_stride; refp.org_x = (uint16_t)ref_org_x; refp.org_y = (uint16_t)ref_org_y;
    refp.width = (uint16_t)pic_w; refp.height = (uint16_t)pic_h;
    inp.width = (uint16_t)pic_w; inp.height = (uint16_t)pic_h;
    pcs->aligned_width = (uint16_t)picture_width; pcs->aligned_height = (uint16_t)picture_height;
    pcs->picture_number = 100 + P->dist; /* ME_MCTF: the distance is used as it is (:1300-1302), so the test passes the final value */
    ctx->me_type = ME_MCTF;
    ctx->me_sa.sa_min.width = (uint16_t)P->sa_min_w; ctx->me_sa.sa_min.height = (uint16_t)P->sa_min_h;
    ctx->me_sa.sa_max.width = (uint16_t)P->sa_max_w; ctx->me_sa.sa_max.height = (uint16_t)P->sa_max_h;
    ctx->mv_based_sa_adj.enabled = P->mv_adj_enabled; ctx->mv_based_sa_adj.nearest_ref_only = P->mv_adj_nearest_ref_only;
    ctx->mv_based_sa_adj.mv_size_th = P->mv_size_th; ctx->mv_based_sa_adj.sa_multiplier = P->sa_multiplier;
    ctx->me_early_exit_th = 0; ctx->is_ref = false; ctx->me_sr_adjustment_ctrls.enable_me_sr_adjustment = 0; ctx->me_8x8_var_ctrls.enabled = 0;
    ctx->me_search_method = P->sub_sad ? SUB_SAD_SEARCH : FULL_SAD_SEARCH;
    ctx->b64_width  = (uint32_t)((picture_width - b64_origin_x) < 64 ? picture_width - b64_origin_x : 64);
    ctx->b64_height = (uint32_t)((picture_height - b64_origin_y) < 64 ? picture_height - b64_origin_y : 64);
    ctx->b64_src_ptr = src_plane + (size_t)(src_org_y + b64_origin_y) * src_stride + src_org_x + b64_origin_x;
    ctx->b64_src_stride = src_stride;
    integer_search_b64(pcs, ctx, (uint32_t)b64_origin_x, (uint32_t)b64_origin_y, &inp);
    memcpy(best_sad, ctx->p_sb_best_sad[l][r], 85 * 4);
    memcpy(best_mv, ctx->p_sb_best_mv[l][r], 85 * 4);
}

/* The reference's TOP-LEVEL open-loop ME of one SB: svt_aom_motion_estimation_b64 (motion_estimation.c:3076-3152) = hme_b64 -> reference pruning ->
 * integer_search_b64 -> me_prune_ref -> construct_me_candidate_array* -> compute_distortion -> GM detection, with the per-SB set-up of
 * me_process.c:183-266.  Everything enters through plain structs; used to pin the device stage end to end. */
typedef struct RefPlane { uint8_t *buf; uint32_t stride, org_x, org_y, width, height; } RefPlane; /* buffer_y[0] of a padded plane */
typedef struct RefPicture { RefPlane lvl[3]; /* [0] sixteenth, [1] quarter, [2] full */ uint64_t picture_number; } RefPicture;
typedef struct RefMeStageOptions {
    uint8_t  num_hme_sa_w, num_hme_sa_h, hme_sub_sampled, me_sub_sad;
    uint16_t hme_l0_min_w, hme_l0_min_h, hme_l0_max_w, hme_l0_max_h; /* TOTAL level-0 area (hme_l0_sa) */
    uint16_t hme_l1_w, hme_l1_h, hme_l2_w, hme_l2_h;
    uint16_t me_min_w, me_min_h, me_max_w, me_max_h;
    uint8_t  mv_adj_enabled, mv_adj_nearest_ref_only; uint16_t mv_adj_mv_size_th, mv_adj_sa_multiplier;
    uint8_t  temporal_layer_index, is_ref;
    uint32_t me_early_exit_th;
    uint8_t  sr_adjustment, me_8x8_var_enabled; uint32_t me_sr_div4_th, me_sr_div2_th, me_sr_mult2_th;
    uint8_t  hme_prune_enabled; uint16_t prune_ref_if_hme_sad_dev_bigger_than_th;
    uint16_t reduce_me_sr_based_on_mv_length_th, stationary_hme_sad_abs_th, stationary_me_sr_divisor, reduce_me_sr_based_on_hme_sad_abs_th,
             me_sr_divisor_for_low_hme_sad;
    uint8_t  distance_based_hme_resizing;
    uint8_t  prehme_enabled, prehme_skip_search_line, prehme_l1_early_exit;
    uint16_t prehme_sa_min_width[2], prehme_sa_min_height[2], prehme_sa_max_width[2], prehme_sa_max_height[2];
    uint32_t zz_sad_th, phme_sad_th; uint16_t zz_sad_pct, phme_sad_pct;
    uint32_t prev_me_stage_based_exit_th, me_safe_limit_zz_th;
    uint32_t me_type_mctf, tf_me_exit_th; /* the temporal filter's form of the call */
    uint8_t  hme_level2_off;              /* enable_hme_level2_flag = 0 (presets M7 and above, enc_mode_config.c:1636-1640) */
    uint8_t  pad;
    uint16_t reduce_hme_l0_sr_th_min, reduce_hme_l0_sr_th_max; /* the low-delay settings' level-0 resizing from list 0 / reference 0's motion (enc_mode_config.c:702-714) */
} RefMeStageOptions;
static MeContext *g_last_ctx; /* the context of the last ref_motion_estimation_b64 call, for ref_me_last_hme */
void ref_motion_estimation_b64(const RefMeStageOptions *O, const RefMeResultsParams *P, const RefPicture *src, const RefPicture *refs /*[2][4]*/,
                               int pic_width, int pic_height, int b64_origin_x, int b64_origin_y, uint8_t *total_me_candidate_index,
                               uint32_t *me_mv_array, uint8_t *me_candidate_array, RefMeSbStats *st, uint32_t *best_sad /*[2][4][85]*/,
                               uint32_t *best_mv, uint8_t *do_ref_out /*[2][4]*/) {
    static PictureParentControlSet *pcs;
    static SequenceControlSet      *scs;
    static MeContext               *ctx;
    static MotionEstimationData    *med;
    static MeSbResults             *res, *res_tab[1];
    static B64Geom                  geom;
    static uint32_t                 u32s[6];
    static uint8_t                  u8s[2];
    static EbPictureBufferDesc      pics[1 + 8][3];
    if (!pcs) {
        pcs = calloc(1, sizeof(*pcs)); scs = calloc(1, sizeof(*scs)); ctx = calloc(1, sizeof(*ctx)); med = calloc(1, sizeof(*med)); res = calloc(1, sizeof(*res));
    }
    memset(ctx, 0, sizeof(*ctx));
    pcs->scs = scs; pcs->pa_me_data = med; res_tab[0] = res; med->me_results = res_tab;
    med->max_cand = P->max_cand; med->max_refs = P->max_refs; med->max_l0 = P->max_l0;
    res->total_me_candidate_index = total_me_candidate_index; res->me_mv_array = (MvCandidate *)me_mv_array; res->me_candidate_array = (MeCandidate *)me_candidate_array;
    pcs->enable_me_16x16 = P->enable_me_16x16; pcs->enable_me_8x8 = P->enable_me_8x8; pcs->max_number_of_pus_per_sb = SQUARE_PU_COUNT;
    scs->mrp_ctrls.only_l_bwd = P->only_l_bwd;
    scs->input_resolution = P->low_resolution ? INPUT_SIZE_480p_RANGE : INPUT_SIZE_1080p_RANGE;
    pcs->aligned_width = (uint16_t)((pic_width + 7) & ~7); pcs->aligned_height = (uint16_t)((pic_height + 7) & ~7);
    geom.width  = (uint8_t)((pcs->aligned_width - b64_origin_x) < 64 ? pcs->aligned_width - b64_origin_x : 64);
    geom.height = (uint8_t)((pcs->aligned_height - b64_origin_y) < 64 ? pcs->aligned_height - b64_origin_y : 64);
    pcs->b64_geom = &geom;
    pcs->me_64x64_distortion = &u32s[0]; pcs->me_32x32_distortion = &u32s[1]; pcs->me_16x16_distortion = &u32s[2];
    pcs->me_8x8_distortion = &u32s[3]; pcs->me_8x8_cost_variance = &u32s[4]; pcs->rc_me_distortion = &u32s[5];
    pcs->stationary_block_present_sb = &u8s[0]; pcs->rc_me_allow_gm = &u8s[1];
    pcs->gm_ctrls.enabled = P->gm_enabled; pcs->gm_ctrls.use_distance_based_active_th = P->gm_use_distance_based_active_th;
    pcs->picture_number = src->picture_number;
#define FILL(dst, pl) do { memset(&(dst), 0, sizeof(dst)); (dst).buffer_y = (pl).buf; (dst).stride_y = (uint16_t)(pl).stride; (dst).org_x = (uint16_t)(pl).org_x; \
                           (dst).org_y = (uint16_t)(pl).org_y; (dst).width = (uint16_t)(pl).width; (dst).height = (uint16_t)(pl).height; } while (0)
    for (int k = 0; k < 3; k++) FILL(pics[0][k], src->lvl[k]);
    ctx->me_type = O->me_type_mctf ? ME_MCTF : ME_OPEN_LOOP; ctx->tf_me_exit_th = O->tf_me_exit_th; ctx->tf_use_pred_64x64_only_th = 0;
    ctx->tf_tot_horz_blks = ctx->tf_tot_vert_blks = 0; g_last_ctx = ctx;
    ctx->num_of_list_to_search = P->num_of_list_to_search;
    ctx->num_of_ref_pic_to_search[0] = P->num_of_ref_pic_to_search[0]; ctx->num_of_ref_pic_to_search[1] = P->num_of_ref_pic_to_search[1];
    ctx->temporal_layer_index = O->temporal_layer_index; ctx->is_ref = O->is_ref;
    for (int l = 0; l < 2; l++)
        for (int r = 0; r < 4; r++) {
            const RefPicture *rp = &refs[l * 4 + r];
            if (!rp->lvl[2].buf) continue;
            for (int k = 0; k < 3; k++) FILL(pics[1 + l * 4 + r][k], rp->lvl[k]);
            ctx->me_ds_ref_array[l][r].sixteenth_picture_ptr = &pics[1 + l * 4 + r][0];
            ctx->me_ds_ref_array[l][r].quarter_picture_ptr   = &pics[1 + l * 4 + r][1];
            ctx->me_ds_ref_array[l][r].picture_ptr           = &pics[1 + l * 4 + r][2];
            ctx->me_ds_ref_array[l][r].picture_number        = rp->picture_number;
        }
    /* per-SB set-up of me_process.c:196-215 */
    ctx->b64_src_ptr = pics[0][2].buffer_y + (pics[0][2].org_y + b64_origin_y) * pics[0][2].stride_y + pics[0][2].org_x + b64_origin_x;
    ctx->b64_src_stride = pics[0][2].stride_y;
    ctx->quarter_b64_buffer = pics[0][1].buffer_y + (pics[0][1].org_y + (b64_origin_y >> 1)) * pics[0][1].stride_y + pics[0][1].org_x + (b64_origin_x >> 1);
    ctx->quarter_b64_buffer_stride = pics[0][1].stride_y;
    ctx->sixteenth_b64_buffer = pics[0][0].buffer_y + (pics[0][0].org_y + (b64_origin_y >> 2)) * pics[0][0].stride_y + pics[0][0].org_x + (b64_origin_x >> 2);
    ctx->sixteenth_b64_buffer_stride = pics[0][0].stride_y;
    ctx->enable_hme_flag = 1; ctx->enable_hme_level0_flag = 1; ctx->enable_hme_level1_flag = O->hme_level2_off < 2; ctx->enable_hme_level2_flag = !O->hme_level2_off; /* 0: three levels, 1: levels 0 and 1, 2: level 0 only */
    ctx->num_hme_sa_w = O->num_hme_sa_w; ctx->num_hme_sa_h = O->num_hme_sa_h;
    ctx->hme_search_method = O->hme_sub_sampled ? SUB_SAD_SEARCH : FULL_SAD_SEARCH;
    ctx->me_search_method  = O->me_sub_sad ? SUB_SAD_SEARCH : FULL_SAD_SEARCH;
    ctx->hme_l0_sa.sa_min.width = O->hme_l0_min_w; ctx->hme_l0_sa.sa_min.height = O->hme_l0_min_h;
    ctx->hme_l0_sa.sa_max.width = O->hme_l0_max_w; ctx->hme_l0_sa.sa_max.height = O->hme_l0_max_h;
    ctx->hme_l1_sa.width = O->hme_l1_w; ctx->hme_l1_sa.height = O->hme_l1_h; ctx->hme_l2_sa.width = O->hme_l2_w; ctx->hme_l2_sa.height = O->hme_l2_h;
    ctx->me_sa.sa_min.width = O->me_min_w; ctx->me_sa.sa_min.height = O->me_min_h; ctx->me_sa.sa_max.width = O->me_max_w; ctx->me_sa.sa_max.height = O->me_max_h;
    ctx->mv_based_sa_adj.enabled = O->mv_adj_enabled; ctx->mv_based_sa_adj.nearest_ref_only = O->mv_adj_nearest_ref_only;
    ctx->mv_based_sa_adj.mv_size_th = O->mv_adj_mv_size_th; ctx->mv_based_sa_adj.sa_multiplier = O->mv_adj_sa_multiplier;
    ctx->me_early_exit_th = O->me_early_exit_th;
    ctx->me_sr_adjustment_ctrls.enable_me_sr_adjustment = O->sr_adjustment;
    ctx->me_sr_adjustment_ctrls.distance_based_hme_resizing = O->distance_based_hme_resizing;
    ctx->reduce_hme_l0_sr_th_min = O->reduce_hme_l0_sr_th_min; ctx->reduce_hme_l0_sr_th_max = O->reduce_hme_l0_sr_th_max;
    ctx->me_sr_adjustment_ctrls.reduce_me_sr_based_on_mv_length_th = O->reduce_me_sr_based_on_mv_length_th;
    ctx->me_sr_adjustment_ctrls.stationary_hme_sad_abs_th = O->stationary_hme_sad_abs_th;
    ctx->me_sr_adjustment_ctrls.stationary_me_sr_divisor = O->stationary_me_sr_divisor;
    ctx->me_sr_adjustment_ctrls.reduce_me_sr_based_on_hme_sad_abs_th = O->reduce_me_sr_based_on_hme_sad_abs_th;
    ctx->me_sr_adjustment_ctrls.me_sr_divisor_for_low_hme_sad = O->me_sr_divisor_for_low_hme_sad;
    ctx->me_8x8_var_ctrls.enabled = O->me_8x8_var_enabled; ctx->me_8x8_var_ctrls.me_sr_div4_th = O->me_sr_div4_th;
    ctx->me_8x8_var_ctrls.me_sr_div2_th = O->me_sr_div2_th; ctx->me_8x8_var_ctrls.me_sr_mult2_th = O->me_sr_mult2_th;
    ctx->me_hme_prune_ctrls.enable_me_hme_ref_pruning = O->hme_prune_enabled || P->prune_ref;
    ctx->me_hme_prune_ctrls.prune_ref_if_hme_sad_dev_bigger_than_th = O->hme_prune_enabled ? O->prune_ref_if_hme_sad_dev_bigger_than_th : (uint16_t)~0;
    ctx->me_hme_prune_ctrls.prune_ref_if_me_sad_dev_bigger_than_th  = P->prune_ref ? P->prune_ref_if_me_sad_dev_bigger_than_th : (uint16_t)~0;
    ctx->me_hme_prune_ctrls.zz_sad_th = O->zz_sad_th; ctx->me_hme_prune_ctrls.zz_sad_pct = O->zz_sad_pct;
    ctx->me_hme_prune_ctrls.phme_sad_th = O->phme_sad_th; ctx->me_hme_prune_ctrls.phme_sad_pct = O->phme_sad_pct;
    ctx->prev_me_stage_based_exit_th = O->prev_me_stage_based_exit_th;
    ctx->me_safe_limit_zz_th = O->me_safe_limit_zz_th; // the picture-level conditions of init_zz_sad (:2419-2421) are made true: top layer of a 1-level hierarchy
    pcs->hierarchical_levels = 1; pcs->temporal_layer_index = O->temporal_layer_index; pcs->similar_brightness_refs = 1;
    ctx->prehme_ctrl.enable = O->prehme_enabled; ctx->prehme_ctrl.skip_search_line = O->prehme_skip_search_line; ctx->prehme_ctrl.l1_early_exit = O->prehme_l1_early_exit;
    for (int k = 0; k < 2; k++) {
        ctx->prehme_ctrl.prehme_sa_cfg[k].sa_min.width = O->prehme_sa_min_width[k]; ctx->prehme_ctrl.prehme_sa_cfg[k].sa_min.height = O->prehme_sa_min_height[k];
        ctx->prehme_ctrl.prehme_sa_cfg[k].sa_max.width = O->prehme_sa_max_width[k]; ctx->prehme_ctrl.prehme_sa_cfg[k].sa_max.height = O->prehme_sa_max_height[k];
    }
    ctx->prune_me_candidates_th = P->prune_me_candidates_th;
    ctx->use_best_unipred_cand_only = P->use_best_unipred_cand_only;
    svt_aom_motion_estimation_b64(pcs, 0, (uint32_t)b64_origin_x, (uint32_t)b64_origin_y, ctx, &pics[0][2]);
    st->me_64x64_distortion = u32s[0]; st->me_32x32_distortion = u32s[1]; st->me_16x16_distortion = u32s[2];
    st->me_8x8_distortion = u32s[3]; st->me_8x8_cost_variance = u32s[4]; st->rc_me_distortion = u32s[5];
    st->stationary_block_present_sb = u8s[0]; st->rc_me_allow_gm = u8s[1]; st->pad[0] = st->pad[1] = 0;
    memcpy(best_sad, ctx->p_sb_best_sad, sizeof(ctx->p_sb_best_sad));
    memcpy(best_mv, ctx->p_sb_best_mv, sizeof(ctx->p_sb_best_mv));
    for (int l = 0; l < 2; l++)
        for (int r = 0; r < 4; r++) do_ref_out[l * 4 + r] = ctx->search_results[l][r].do_ref;
#undef FILL
}
/* search_results[list][ref].hme_sc_x / hme_sc_y / hme_sad of the last call, and what the ME_MCTF form leaves for the temporal filter */
void ref_me_last_hme(int16_t *sc /*[2][4][2]*/, uint32_t *sad /*[2][4]*/, uint32_t *tf /*[3]: tf_use_pred_64x64_only_th, horz, vert*/) {
    for (int l = 0; l < 2; l++)
        for (int r = 0; r < 4; r++) {
            sc[(l * 4 + r) * 2] = g_last_ctx->search_results[l][r].hme_sc_x; sc[(l * 4 + r) * 2 + 1] = g_last_ctx->search_results[l][r].hme_sc_y;
            sad[l * 4 + r] = g_last_ctx->search_results[l][r].hme_sad;
        }
    tf[0] = g_last_ctx->tf_use_pred_64x64_only_th; tf[1] = g_last_ctx->tf_tot_horz_blks; tf[2] = g_last_ctx->tf_tot_vert_blks;
}


/* The reference's OWN derivation of the ME settings for a picture: svt_aom_sig_deriv_me (enc_mode_config.c:681-815) run on a zeroed SequenceControlSet /
 * PictureParentControlSet carrying the handful of inputs it reads, with the picture-level HME flags the way svt_aom_sig_deriv_multi_processes sets them
 * (enc_mode_config.c:1630-1642: level 2 only up to M6 or for screen content).  The context fields come back in the RefMeStageOptions layout, so a test can
 * pin "what preset 8 means at 1080p" on the reference instead of on a reading of it.  extra[0..5] = prune_ref_if_me_sad_dev_bigger_than_th,
 * prune_me_candidates_th, enable_me_hme_ref_pruning, hme_level2 flag, hme_level1 flag, me_safe_limit_zz_th. */
void svt_aom_sig_deriv_me(SequenceControlSet *scs, PictureParentControlSet *pcs, MeContext *me_ctx);
void ref_sig_deriv_me(int enc_mode, int input_resolution, int qp, int sc_class1, int temporal_layer_index, int hierarchical_levels, int low_delay,
                      RefMeStageOptions *O, int32_t *extra) {
    SequenceControlSet      *scs = calloc(1, sizeof(*scs));
    PictureParentControlSet *pcs = calloc(1, sizeof(*pcs));
    MeContext               *c   = calloc(1, sizeof(*c));
    pcs->scs = scs;
    pcs->enc_mode = (EncMode)enc_mode; pcs->sc_class1 = (uint8_t)sc_class1; pcs->temporal_layer_index = (uint8_t)temporal_layer_index;
    pcs->hierarchical_levels = (uint8_t)hierarchical_levels; pcs->input_resolution = (uint8_t)input_resolution;
    scs->input_resolution = (EbInputResolution)input_resolution;
    scs->static_config.qp = (uint32_t)qp;
    scs->static_config.pred_structure = low_delay ? SVT_AV1_PRED_LOW_DELAY_B : SVT_AV1_PRED_RANDOM_ACCESS;
    scs->frame_rate = 30; /* < 1 << 16: not the low-frame-rate case (enc_mode_config.c:339-343 tests frame_rate >> 16) */
    pcs->enable_hme_flag = 1; pcs->enable_hme_level0_flag = 1; pcs->enable_hme_level1_flag = 1;
    pcs->enable_hme_level2_flag = (sc_class1 || enc_mode <= ENC_M6) ? 1 : 0; /* enc_mode_config.c:1632-1642 */
    svt_aom_sig_deriv_me(scs, pcs, c);
    memset(O, 0, sizeof(*O));
    O->num_hme_sa_w = (uint8_t)c->num_hme_sa_w; O->num_hme_sa_h = (uint8_t)c->num_hme_sa_h;
    O->hme_sub_sampled = c->hme_search_method != FULL_SAD_SEARCH; O->me_sub_sad = c->me_search_method == SUB_SAD_SEARCH;
    O->hme_l0_min_w = c->hme_l0_sa.sa_min.width; O->hme_l0_min_h = c->hme_l0_sa.sa_min.height; O->hme_l0_max_w = c->hme_l0_sa.sa_max.width; O->hme_l0_max_h = c->hme_l0_sa.sa_max.height;
    O->hme_l1_w = c->hme_l1_sa.width; O->hme_l1_h = c->hme_l1_sa.height; O->hme_l2_w = c->hme_l2_sa.width; O->hme_l2_h = c->hme_l2_sa.height;
    O->me_min_w = c->me_sa.sa_min.width; O->me_min_h = c->me_sa.sa_min.height; O->me_max_w = c->me_sa.sa_max.width; O->me_max_h = c->me_sa.sa_max.height;
    O->mv_adj_enabled = c->mv_based_sa_adj.enabled; O->mv_adj_nearest_ref_only = c->mv_based_sa_adj.nearest_ref_only;
    O->mv_adj_mv_size_th = c->mv_based_sa_adj.mv_size_th; O->mv_adj_sa_multiplier = c->mv_based_sa_adj.sa_multiplier;
    O->temporal_layer_index = (uint8_t)temporal_layer_index;
    O->me_early_exit_th = c->me_early_exit_th;
    O->sr_adjustment = c->me_sr_adjustment_ctrls.enable_me_sr_adjustment; O->distance_based_hme_resizing = c->me_sr_adjustment_ctrls.distance_based_hme_resizing;
    O->reduce_hme_l0_sr_th_min = c->reduce_hme_l0_sr_th_min; O->reduce_hme_l0_sr_th_max = c->reduce_hme_l0_sr_th_max;
    O->reduce_me_sr_based_on_mv_length_th = c->me_sr_adjustment_ctrls.reduce_me_sr_based_on_mv_length_th;
    O->stationary_hme_sad_abs_th = c->me_sr_adjustment_ctrls.stationary_hme_sad_abs_th; O->stationary_me_sr_divisor = c->me_sr_adjustment_ctrls.stationary_me_sr_divisor;
    O->reduce_me_sr_based_on_hme_sad_abs_th = c->me_sr_adjustment_ctrls.reduce_me_sr_based_on_hme_sad_abs_th;
    O->me_sr_divisor_for_low_hme_sad = c->me_sr_adjustment_ctrls.me_sr_divisor_for_low_hme_sad;
    O->me_8x8_var_enabled = c->me_8x8_var_ctrls.enabled; O->me_sr_div4_th = c->me_8x8_var_ctrls.me_sr_div4_th; O->me_sr_div2_th = c->me_8x8_var_ctrls.me_sr_div2_th;
    O->me_sr_mult2_th = c->me_8x8_var_ctrls.me_sr_mult2_th;
    O->hme_prune_enabled = c->me_hme_prune_ctrls.enable_me_hme_ref_pruning && c->me_hme_prune_ctrls.prune_ref_if_hme_sad_dev_bigger_than_th != (uint16_t)~0;
    O->prune_ref_if_hme_sad_dev_bigger_than_th = c->me_hme_prune_ctrls.prune_ref_if_hme_sad_dev_bigger_than_th;
    O->prehme_enabled = c->prehme_ctrl.enable; O->prehme_skip_search_line = c->prehme_ctrl.skip_search_line; O->prehme_l1_early_exit = c->prehme_ctrl.l1_early_exit;
    for (int k = 0; k < 2; k++) {
        O->prehme_sa_min_width[k] = c->prehme_ctrl.prehme_sa_cfg[k].sa_min.width; O->prehme_sa_min_height[k] = c->prehme_ctrl.prehme_sa_cfg[k].sa_min.height;
        O->prehme_sa_max_width[k] = c->prehme_ctrl.prehme_sa_cfg[k].sa_max.width; O->prehme_sa_max_height[k] = c->prehme_ctrl.prehme_sa_cfg[k].sa_max.height;
    }
    O->zz_sad_th = c->me_hme_prune_ctrls.zz_sad_th; O->zz_sad_pct = (uint16_t)c->me_hme_prune_ctrls.zz_sad_pct;
    O->phme_sad_th = c->me_hme_prune_ctrls.phme_sad_th; O->phme_sad_pct = (uint16_t)c->me_hme_prune_ctrls.phme_sad_pct;
    O->prev_me_stage_based_exit_th = c->prev_me_stage_based_exit_th;
    O->me_safe_limit_zz_th = c->me_safe_limit_zz_th;
    O->hme_level2_off = !c->enable_hme_level2_flag + !c->enable_hme_level1_flag;
    extra[0] = c->me_hme_prune_ctrls.prune_ref_if_me_sad_dev_bigger_than_th; extra[1] = c->prune_me_candidates_th;
    extra[2] = c->me_hme_prune_ctrls.enable_me_hme_ref_pruning; extra[3] = c->enable_hme_level2_flag; extra[4] = c->enable_hme_level1_flag;
    extra[5] = (int32_t)c->me_safe_limit_zz_th;
    free(c); free(pcs); free(scs);
}
