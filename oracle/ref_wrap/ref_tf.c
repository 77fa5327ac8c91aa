/*
 * ref_tf.c -- TEST INFRASTRUCTURE, built only into oracle/_ref/libsvtref_me.so (make -C oracle ref).
 *
 * The temporal filter's pixel kernels take a `struct MeContext *` (aom_dsp_rtcd.h:797-826, 835-838).  This file is compiled against the
 * reference's own headers (me_context.h, where they lie) so that the tests can call the reference's `_c` functions with a MeContext
 * filled from the plain parameter structs of include/svtav1_hip.h -- the same adapter a maintainer writes in INTEGRATION.md, in reverse.
 */
#include "me_context.h"
#include "temporal_filtering.h"
#include "aom_dsp_rtcd.h"

typedef struct RefTfParams { /* = SvtHipTfParams */
    uint32_t tf_decay_factor_fp16[3];
    uint16_t tf_mv_dist_th;
    uint8_t  tf_chroma, use_zz_based_filter, encoder_bit_depth, ss_x, ss_y, pad;
} RefTfParams;
typedef struct RefTfBlock { /* = SvtHipTfBlock */
    uint64_t block_error[4];
    int16_t  mv_x[4], mv_y[4];
    uint8_t  split, pad[7];
} RefTfBlock;

static MeContext *ctx_from(const RefTfParams *P, const RefTfBlock *B, int block_row, int block_col) {
    static MeContext *ctx;
    if (!ctx) ctx = calloc(1, sizeof(*ctx));
    for (int c = 0; c < 3; c++) ctx->tf_decay_factor_fp16[c] = P->tf_decay_factor_fp16[c];
    ctx->tf_mv_dist_th = P->tf_mv_dist_th;
    ctx->tf_chroma     = P->tf_chroma;
    ctx->tf_ctrls.use_zz_based_filter = P->use_zz_based_filter;
    ctx->tf_block_row = block_row;
    ctx->tf_block_col = block_col;
    if (B) {
        const int idx = block_col + 2 * block_row;
        ctx->tf_32x32_block_split_flag[idx] = B->split;
        if (B->split)
            for (int i = 0; i < 4; i++) {
                ctx->tf_16x16_block_error[4 * idx + i] = B->block_error[i];
                ctx->tf_16x16_mv_x[4 * idx + i] = B->mv_x[i];
                ctx->tf_16x16_mv_y[4 * idx + i] = B->mv_y[i];
            }
        else {
            ctx->tf_32x32_block_error[idx] = B->block_error[0];
            ctx->tf_32x32_mv_x[idx] = B->mv_x[0];
            ctx->tf_32x32_mv_y[idx] = B->mv_y[0];
        }
    }
    return ctx;
}

void ref_tf_planewise(const RefTfParams *P, const RefTfBlock *B, int block_row, int block_col, const void *y_src, int y_src_stride, const void *y_pre,
                      int y_pre_stride, const void *u_src, const void *v_src, int uv_src_stride, const void *u_pre, const void *v_pre, int uv_pre_stride,
                      unsigned bw, unsigned bh, int ss_x, int ss_y, uint32_t *y_accum, uint16_t *y_count, uint32_t *u_accum, uint16_t *u_count,
                      uint32_t *v_accum, uint16_t *v_count, int zz, int hbd) {
    MeContext *ctx = ctx_from(P, B, block_row, block_col);
    if (!hbd && !zz)
        svt_av1_apply_temporal_filter_planewise_medium_c(ctx, y_src, y_src_stride, y_pre, y_pre_stride, u_src, v_src, uv_src_stride, u_pre, v_pre, uv_pre_stride,
                                                         bw, bh, ss_x, ss_y, y_accum, y_count, u_accum, u_count, v_accum, v_count);
    else if (!hbd)
        svt_av1_apply_zz_based_temporal_filter_planewise_medium_c(ctx, y_pre, y_pre_stride, u_pre, v_pre, uv_pre_stride, bw, bh, ss_x, ss_y, y_accum, y_count,
                                                                  u_accum, u_count, v_accum, v_count);
    else if (!zz)
        svt_av1_apply_temporal_filter_planewise_medium_hbd_c(ctx, y_src, y_src_stride, y_pre, y_pre_stride, u_src, v_src, uv_src_stride, u_pre, v_pre,
                                                             uv_pre_stride, bw, bh, ss_x, ss_y, y_accum, y_count, u_accum, u_count, v_accum, v_count,
                                                             P->encoder_bit_depth);
    else
        svt_av1_apply_zz_based_temporal_filter_planewise_medium_hbd_c(ctx, y_pre, y_pre_stride, u_pre, v_pre, uv_pre_stride, bw, bh, ss_x, ss_y, y_accum,
                                                                      y_count, u_accum, u_count, v_accum, v_count, P->encoder_bit_depth);
}

/* accum/count of one 64x64 block initialised from the central picture (apply_filtering_central{,_highbd}) */
void ref_tf_central(const RefTfParams *P, void *src[3], int stride_y, uint32_t *accum[3], uint16_t *count[3], int hbd) {
    MeContext          *ctx = ctx_from(P, NULL, 0, 0);
    EbPictureBufferDesc pic;
    memset(&pic, 0, sizeof(pic));
    pic.stride_y = (uint16_t)stride_y;
    if (hbd) svt_aom_apply_filtering_central_highbd_c(ctx, &pic, (uint16_t **)src, accum, count, BW, BH, P->ss_x, P->ss_y);
    else svt_aom_apply_filtering_central_c(ctx, &pic, (EbByte *)src, accum, count, BW, BH, P->ss_x, P->ss_y);
}
/* normalisation of one 64x64 block into the picture (svt_aom_get_final_filtered_pixels_c) */
void ref_tf_final(const RefTfParams *P, void *dst[3], uint32_t *accum[3], uint16_t *count[3], const uint32_t stride[3], int y_off, int ch_off, int hbd) {
    MeContext *ctx = ctx_from(P, NULL, 0, 0);
    svt_aom_get_final_filtered_pixels_c(ctx, (EbByte *)dst, (uint16_t **)dst, accum, count, stride, y_off, ch_off, (uint16_t)(BW >> P->ss_x),
                                        (uint16_t)(BH >> P->ss_y), hbd != 0);
}
