/*
 * ref_tf_subpel.c -- TEST INFRASTRUCTURE, built only into oracle/_ref/libsvtref_me.so (make -C oracle ref).
 *
 * The temporal filter's sub-pel refinement (tf_subpel_search / svt_check_position, Codec/temporal_filtering.c:1560-1790) is `static`.  This translation unit
 * compiles that reference source file WHERE IT LIES (the #include below resolves through -I$(REF)/Source/Lib/Codec; nothing is copied) and adds a plain-C entry
 * point that fills the few pcs / context fields the search reads and runs it for ONE block, exactly as tf_{64x64,32x32,16x16,8x8}_sub_pel_search set it up
 * (:1793-2250): the 64x64 prediction scratch with pitch BW, the block's MacroBlockD edges, the starting MV.  The luma prediction goes through the
 * reference's own svt_aom_simple_luma_unipred and convolve tables, the distortion through svt_aom_mefn_ptr[].vf / vf_hbd_10.
 */
#include "temporal_filtering.c"

typedef struct RefTfSubpelParams { /* = SvtHipTfSubpelParams / OracleTfSubpelParams */
    uint8_t  half_pel_mode, quarter_pel_mode, eight_pel_mode, subsampling_shift, bit_depth, pad[3];
    uint32_t early_exit_th, mi_rows, mi_cols, ref_org_x, ref_org_y, ref_stride;
} RefTfSubpelParams;

void svt_aom_asm_set_convolve_asm_table(void);
void svt_aom_asm_set_convolve_hbd_asm_table(void);
void init_fn_ptr(void);

/* src: top-left sample of the 64x64 source block this block belongs to (local origin = (idx_x, idx_y) * bsize inside it); ref_buffer_y: buffer_y of the padded
 * reference picture (u8 or u16 samples); pic_w / pic_h: its visible size.  *mv_x / *mv_y: starting MV (1/8 pel) in, best MV out; *dist: INT_MAX in, best out. */
void ref_tf_subpel_search(const RefTfSubpelParams *P, void *src_sb, int src_stride, void *ref_buffer_y, int pic_w, int pic_h, int sb_origin_x, int sb_origin_y,
                          int bsize, int idx_x, int idx_y, int bilinear, int16_t *mv_x, int16_t *mv_y, uint64_t *dist) {
    static PictureParentControlSet *pcs;
    static SequenceControlSet      *scs;
    static MeContext               *ctx;
    static Av1Common               *cm;
    static void                    *pred_buf;
    if (!pcs) {
        pcs = calloc(1, sizeof(*pcs)); scs = calloc(1, sizeof(*scs)); ctx = calloc(1, sizeof(*ctx)); cm = calloc(1, sizeof(*cm));
        pred_buf = calloc(BW * BH, 2);
        svt_aom_asm_set_convolve_asm_table();
        svt_aom_asm_set_convolve_hbd_asm_table();
        init_fn_ptr();
    }
    pcs->scs = scs; pcs->av1_cm = cm;
    cm->mi_rows = (int32_t)P->mi_rows; cm->mi_cols = (int32_t)P->mi_cols;
    svt_av1_setup_scale_factors_for_frame(&scs->sf_identity, pic_w, pic_h, pic_w, pic_h);
    pcs->tf_ctrls.half_pel_mode = P->half_pel_mode; pcs->tf_ctrls.quarter_pel_mode = P->quarter_pel_mode; pcs->tf_ctrls.eight_pel_mode = P->eight_pel_mode;
    pcs->tf_ctrls.sub_sampling_shift = P->subsampling_shift;
    ctx->tf_subpel_early_exit_th = P->early_exit_th;
    const bool is_highbd = P->bit_depth > 8;
    EbPictureBufferDesc ref_pic, prediction_ptr;
    memset(&ref_pic, 0, sizeof(ref_pic)); memset(&prediction_ptr, 0, sizeof(prediction_ptr));
    ref_pic.buffer_y = ref_buffer_y; ref_pic.org_x = (uint16_t)P->ref_org_x; ref_pic.org_y = (uint16_t)P->ref_org_y; ref_pic.stride_y = (uint16_t)P->ref_stride;
    ref_pic.width = (uint16_t)pic_w; ref_pic.height = (uint16_t)pic_h;
    prediction_ptr.stride_y = BW; prediction_ptr.buffer_y = pred_buf;
    EbByte    pred[3]       = {pred_buf, NULL, NULL}, src[3] = {src_sb, NULL, NULL};
    uint16_t *pred_16bit[3] = {pred_buf, NULL, NULL}, *src_16bit[3] = {src_sb, NULL, NULL};
    uint32_t  stride_pred[3] = {BW, BW >> 1, BW >> 1}, stride_src[3] = {(uint32_t)src_stride, 0, 0};
    BlkStruct   blk_struct;
    MacroBlockD av1xd;
    blk_struct.av1xd = &av1xd;
    const uint16_t local_origin_x = (uint16_t)(idx_x * bsize), local_origin_y = (uint16_t)(idx_y * bsize);
    const uint16_t pu_origin_x = (uint16_t)(sb_origin_x + local_origin_x), pu_origin_y = (uint16_t)(sb_origin_y + local_origin_y);
    const int32_t  mirow = pu_origin_y >> MI_SIZE_LOG2, micol = pu_origin_x >> MI_SIZE_LOG2, bmi = bsize >> 2;
    av1xd.mb_to_top_edge    = -(int32_t)((mirow * MI_SIZE) * 8);
    av1xd.mb_to_bottom_edge = ((cm->mi_rows - bmi - mirow) * MI_SIZE) * 8;
    av1xd.mb_to_left_edge   = -(int32_t)((micol * MI_SIZE) * 8);
    av1xd.mb_to_right_edge  = ((cm->mi_cols - bmi - micol) * MI_SIZE) * 8;
    TF_SUBPEL_SEARCH_PARAMS sp;
    memset(&sp, 0, sizeof(sp));
    sp.subsampling_shift = P->subsampling_shift;
    sp.interp_filters = (uint32_t)(bilinear ? av1_make_interp_filters(BILINEAR, BILINEAR) : av1_make_interp_filters(EIGHTTAP_REGULAR, EIGHTTAP_REGULAR));
    sp.pu_origin_x = pu_origin_x; sp.pu_origin_y = pu_origin_y; sp.local_origin_x = local_origin_x; sp.local_origin_y = local_origin_y;
    sp.bsize = (uint32_t)bsize; sp.is_highbd = is_highbd; sp.encoder_bit_depth = P->bit_depth; sp.idx_x = (uint32_t)idx_x; sp.idx_y = (uint32_t)idx_y;
    tf_subpel_search(&sp, pcs, ctx, &blk_struct, &ref_pic, &prediction_ptr, pred, pred_16bit, stride_pred, src, src_16bit, stride_src, dist, mv_x, mv_y);
}

/* the luma prediction alone (svt_aom_simple_luma_unipred), w = h = bsize, into dst (pitch 64 samples) */
void ref_tf_luma_pred(const RefTfSubpelParams *P, void *ref_buffer_y, int pic_w, int pic_h, int pu_x, int pu_y, int bsize, int mv_x, int mv_y, int bilinear,
                      int subsampling_shift, void *dst64) {
    uint64_t d = 0;
    (void)d;
    static SequenceControlSet *scs;
    if (!scs) { scs = calloc(1, sizeof(*scs)); svt_aom_asm_set_convolve_asm_table(); svt_aom_asm_set_convolve_hbd_asm_table(); }
    svt_av1_setup_scale_factors_for_frame(&scs->sf_identity, pic_w, pic_h, pic_w, pic_h);
    EbPictureBufferDesc ref_pic, prediction_ptr;
    memset(&ref_pic, 0, sizeof(ref_pic)); memset(&prediction_ptr, 0, sizeof(prediction_ptr));
    ref_pic.buffer_y = ref_buffer_y; ref_pic.org_x = (uint16_t)P->ref_org_x; ref_pic.org_y = (uint16_t)P->ref_org_y; ref_pic.stride_y = (uint16_t)P->ref_stride;
    ref_pic.width = (uint16_t)pic_w; ref_pic.height = (uint16_t)pic_h;
    prediction_ptr.stride_y = BW; prediction_ptr.buffer_y = dst64;
    BlkStruct   blk_struct;
    MacroBlockD av1xd;
    blk_struct.av1xd = &av1xd;
    const int32_t mirow = pu_y >> MI_SIZE_LOG2, micol = pu_x >> MI_SIZE_LOG2, bmi = bsize >> 2;
    av1xd.mb_to_top_edge    = -(int32_t)((mirow * MI_SIZE) * 8);
    av1xd.mb_to_bottom_edge = (((int32_t)P->mi_rows - bmi - mirow) * MI_SIZE) * 8;
    av1xd.mb_to_left_edge   = -(int32_t)((micol * MI_SIZE) * 8);
    av1xd.mb_to_right_edge  = (((int32_t)P->mi_cols - bmi - micol) * MI_SIZE) * 8;
    MvUnit mv_unit;
    mv_unit.pred_direction = UNI_PRED_LIST_0;
    mv_unit.mv->x = (int16_t)mv_x; mv_unit.mv->y = (int16_t)mv_y;
    svt_aom_simple_luma_unipred(scs, scs->sf_identity, (uint32_t)(bilinear ? av1_make_interp_filters(BILINEAR, BILINEAR) : av1_make_interp_filters(EIGHTTAP_REGULAR, EIGHTTAP_REGULAR)),
                                &blk_struct, 0, &mv_unit, (uint16_t)pu_x, (uint16_t)pu_y, (uint8_t)bsize, (uint8_t)bsize, &ref_pic, &prediction_ptr, 0, 0, P->bit_depth,
                                (uint8_t)subsampling_shift);
}

/* The final motion compensation of one block exactly as tf_{64x64,32x32,16x16,8x8}_inter_prediction issue it (:2256-2620): svt_aom_inter_prediction, one
 * direction, SIMPLE_TRANSLATION, MULTITAP_SHARP, luma (+ chroma with tf_chroma) into a 64-pitch prediction buffer at the block's local origin.
 * planes[3] / strides[3]: the reference picture's padded buffers (u8, or u16 for the high-bit-depth path = pcs_ref->altref_buffer_highbd). */
void svt_aom_build_blk_geom(GeomIndex geom);
void ref_tf_inter_pred(const RefTfSubpelParams *P, void *const *planes, const uint32_t *strides, int pic_w, int pic_h, int sb_origin_x, int sb_origin_y, int bsize,
                       int idx_x, int idx_y, int mv_x, int mv_y, int chroma, void *const *pred64 /* y: 64 x 64, u / v: 32 x 32, pitch 64 / 32 */) {
    static SequenceControlSet *scs;
    if (!scs) {
        scs = calloc(1, sizeof(*scs));
        svt_aom_asm_set_convolve_asm_table();
        svt_aom_asm_set_convolve_hbd_asm_table();
        svt_aom_build_blk_geom(GEOM_0);
    }
    svt_av1_setup_scale_factors_for_frame(&scs->sf_identity, pic_w, pic_h, pic_w, pic_h);
    const bool is_highbd = P->bit_depth > 8;
    EbPictureBufferDesc ref_pic, prediction_ptr;
    memset(&ref_pic, 0, sizeof(ref_pic)); memset(&prediction_ptr, 0, sizeof(prediction_ptr));
    ref_pic.buffer_y = planes[0]; ref_pic.buffer_cb = planes[1]; ref_pic.buffer_cr = planes[2];
    ref_pic.org_x = (uint16_t)P->ref_org_x; ref_pic.org_y = (uint16_t)P->ref_org_y;
    ref_pic.stride_y = (uint16_t)strides[0]; ref_pic.stride_cb = (uint16_t)strides[1]; ref_pic.stride_cr = (uint16_t)strides[2];
    ref_pic.width = (uint16_t)pic_w; ref_pic.height = (uint16_t)pic_h;
    prediction_ptr.stride_y = BW; prediction_ptr.stride_cb = BW >> 1; prediction_ptr.stride_cr = BW >> 1;
    prediction_ptr.buffer_y = pred64[0]; prediction_ptr.buffer_cb = pred64[1]; prediction_ptr.buffer_cr = pred64[2];
    BlkStruct   blk_ptr;
    MacroBlockD av1xd;
    MvUnit      mv_unit;
    memset(&blk_ptr, 0, sizeof(blk_ptr)); memset(&av1xd, 0, sizeof(av1xd)); memset(&mv_unit, 0, sizeof(mv_unit));
    blk_ptr.av1xd = &av1xd;
    mv_unit.pred_direction = UNI_PRED_LIST_0;
    const uint16_t local_origin_x = (uint16_t)(idx_x * bsize), local_origin_y = (uint16_t)(idx_y * bsize);
    const uint16_t pu_origin_x = (uint16_t)(sb_origin_x + local_origin_x), pu_origin_y = (uint16_t)(sb_origin_y + local_origin_y);
    const int32_t  mirow = pu_origin_y >> MI_SIZE_LOG2, micol = pu_origin_x >> MI_SIZE_LOG2, bmi = bsize >> 2;
    blk_ptr.mds_idx         = get_mds_idx(local_origin_x, local_origin_y, (uint32_t)bsize, 0);
    av1xd.mb_to_top_edge    = -(int32_t)((mirow * MI_SIZE) * 8);
    av1xd.mb_to_bottom_edge = (((int32_t)P->mi_rows - bmi - mirow) * MI_SIZE) * 8;
    av1xd.mb_to_left_edge   = -(int32_t)((micol * MI_SIZE) * 8);
    av1xd.mb_to_right_edge  = (((int32_t)P->mi_cols - bmi - micol) * MI_SIZE) * 8;
    mv_unit.mv->x = (int16_t)mv_x; mv_unit.mv->y = (int16_t)mv_y;
    svt_aom_inter_prediction(scs, NULL, (uint32_t)av1_make_interp_filters(MULTITAP_SHARP, MULTITAP_SHARP), &blk_ptr, 0, &mv_unit, 0, SIMPLE_TRANSLATION, 0, 0, 1, NULL,
                             NULL, NULL, NULL, 0, 0, 0, 0, pu_origin_x, pu_origin_y, (uint8_t)bsize, (uint8_t)bsize, &ref_pic, NULL, &prediction_ptr, local_origin_x,
                             local_origin_y, chroma ? PICTURE_BUFFER_DESC_FULL_MASK : PICTURE_BUFFER_DESC_LUMA_MASK, P->bit_depth, is_highbd);
}
