/*
 * oracle_tf.c -- TEST INFRASTRUCTURE (checker only, never linked into the product).
 *
 * CPU restatement of the temporal filter's pixel kernels (the DSP part of Codec/temporal_filtering.c): the plane-wise
 * non-local-means accumulation with and without motion (svt_av1_apply_temporal_filter_planewise_medium{,_hbd}_c :1029-1400,
 * svt_av1_apply_zz_based_temporal_filter_planewise_medium{,_hbd}_c :819-1013), the central-picture initialisation
 * (svt_aom_apply_filtering_central{,_highbd}_c :350-425), the normalisation (svt_aom_get_final_filtered_pixels_c :2608-2672) and the
 * noise estimate (svt_estimate_noise{,_highbd}_fp16_c :3847-3920).  All fixed point.  Pinned against those functions in tests/test_tf.py.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "oracle_tf_tables_gen.h"

typedef struct OracleTfParams { /* = SvtHipTfParams: the per-picture MeContext fields the kernels read */
    uint32_t tf_decay_factor_fp16[3];
    uint16_t tf_mv_dist_th;
    uint8_t  tf_chroma, use_zz_based_filter, encoder_bit_depth, ss_x, ss_y, pad;
} OracleTfParams;
typedef struct OracleTfBlock { /* = SvtHipTfBlock: the per-32x32-block MeContext fields (idx_32x32 = tf_block_col + 2 * tf_block_row) */
    uint64_t block_error[4]; /* split: tf_16x16_block_error[4 idx + i]; else [0] = tf_32x32_block_error[idx] */
    int16_t  mv_x[4], mv_y[4];
    uint8_t  split, pad[7];
} OracleTfBlock;

#define TF_WEIGHT_SCALE 1000                 /* temporal_filtering.h:45 (= TF_PLANEWISE_FILTER_WEIGHT_SCALE :40) */
#define TF_WINDOW_BLOCK_BALANCE_WEIGHT 10    /* :49 */

static uint32_t ilog2(uint32_t x) { return 31 - (uint32_t)__builtin_clz(x); } /* svt_aom_log2f_32, utility.c:160-172 */
static uint32_t sqrt_fast(uint32_t x) { /* temporal_filtering.c:741-750 */
    if (x > 15) {
        const int log2_half = (int)(ilog2(x) >> 1);
        return kTfSqrtQ16[x >> (2 * log2_half - 2)] >> (17 - log2_half);
    }
    return kTfSqrtQ16[x] >> 16;
}
static uint32_t px(const void *p, int hbd, long i) { return hbd ? ((const uint16_t *)p)[i] : ((const uint8_t *)p)[i]; }

/* one plane of one block: window errors of the four quadrants (luma: stored for the chroma planes), weight per quadrant, accumulate */
static void planewise_partial(const OracleTfParams *P, const OracleTfBlock *B, const void *src, int src_stride, const void *pre, int pre_stride,
                              unsigned w, unsigned h, uint32_t *accum, uint16_t *count, uint32_t decay, uint32_t luma_win[4], int is_chroma, int hbd) {
    const int      shift = hbd ? (P->encoder_bit_depth - 8) * 2 : 0;
    const uint32_t dist_th = ((uint32_t)P->tf_mv_dist_th << 16) / 10 > (1u << 16) ? ((uint32_t)P->tf_mv_dist_th << 16) / 10 : (1u << 16);
    uint32_t d_factor[4], blk_err[4], win[4];
    for (int i = 0; i < 4; i++) {
        const int k = B->split ? i : 0;
        const int32_t col = B->mv_x[k], row = B->mv_y[k];
        const uint32_t dist = sqrt_fast(((uint32_t)(col * col + row * row)) << 8);
        const uint32_t d = (dist << 12) / (dist_th >> 8);
        d_factor[i] = d > (1u << 8) ? d : (1u << 8);
        blk_err[i] = B->split ? (uint32_t)(B->block_error[i] >> (hbd ? 4 : 0)) : (uint32_t)(B->block_error[0] >> (hbd ? 6 : 2));
    }
    if (!B->split) decay <<= 1;
    const unsigned wh = w >> 1, hh = h >> 1;
    for (int q = 0; q < 4; q++) {
        const long so = (long)(q >> 1) * hh * src_stride + (q & 1) * wh, po = (long)(q >> 1) * hh * pre_stride + (q & 1) * wh;
        uint32_t sum = 0;
        for (unsigned i = 0; i < hh; i++)
            for (unsigned j = 0; j < wh; j++) {
                const int d = (int)px(src, hbd, so + (long)i * src_stride + j) - (int)px(pre, hbd, po + (long)i * pre_stride + j);
                sum += (uint32_t)(d * d);
            }
        sum >>= shift;
        win[q] = (((sum << 4) / wh) << 4) / hh;
        if (is_chroma) win[q] = (win[q] * 5 + luma_win[q]) / 6;
        else luma_win[q] = win[q];
    }
    for (int q = 0; q < 4; q++) {
        const uint32_t combined = (win[q] * TF_WINDOW_BLOCK_BALANCE_WEIGHT + blk_err[q]) / (TF_WINDOW_BLOCK_BALANCE_WEIGHT + 1);
        const uint64_t avg = (uint64_t)((combined >> 3) * (d_factor[q] >> 3)); /* 32-bit product, as the reference */
        const uint32_t den = (decay >> 10) > 1 ? (decay >> 10) : 1;
        const uint64_t sd  = avg / den;
        const uint32_t weight = (uint32_t)(kTfExpQ16[sd < 7 * 16 ? sd : 7 * 16] * TF_WEIGHT_SCALE) >> 16;
        const int x0 = (q & 1) * (int)w / 2, y0 = (q >> 1) * (int)h / 2;
        for (unsigned i = 0; i < h / 2; i++)
            for (unsigned j = 0; j < w / 2; j++) {
                const long k = (long)(i + y0) * pre_stride + j + x0;
                count[k] = (uint16_t)(count[k] + weight);
                accum[k] += weight * px(pre, hbd, k);
            }
    }
}
static void zz_partial(const OracleTfBlock *B, const void *pre, int pre_stride, unsigned w, unsigned h, uint32_t *accum, uint16_t *count, uint32_t decay,
                       int hbd) {
    for (int q = 0; q < 4; q++) {
        const uint32_t err = B->split ? (uint32_t)(B->block_error[q] >> (hbd ? 4 : 0)) : (uint32_t)(B->block_error[0] >> (hbd ? 6 : 2));
        const uint32_t avg = err << 2;
        const uint32_t den = (decay >> 10) > 1 ? (decay >> 10) : 1;
        const uint32_t sd  = avg / den < 7 * 16 ? avg / den : 7 * 16;
        const uint32_t weight = (uint32_t)(kTfExpQ16[sd] * TF_WEIGHT_SCALE) >> 17;
        const int x0 = (q & 1) * (int)w / 2, y0 = (q >> 1) * (int)h / 2;
        for (unsigned i = 0; i < h / 2; i++)
            for (unsigned j = 0; j < w / 2; j++) {
                const long k = (long)(i + y0) * pre_stride + j + x0;
                count[k] = (uint16_t)(count[k] + weight);
                accum[k] += weight * px(pre, hbd, k);
            }
    }
}

/* the four RTCD functions in one: zz = use_zz_based_filter, hbd = 16-bit samples */
void oracle_tf_planewise(const OracleTfParams *P, const OracleTfBlock *B, const void *y_src, int y_src_stride, const void *y_pre, int y_pre_stride,
                         const void *u_src, const void *v_src, int uv_src_stride, const void *u_pre, const void *v_pre, int uv_pre_stride, unsigned bw,
                         unsigned bh, int ss_x, int ss_y, uint32_t *y_accum, uint16_t *y_count, uint32_t *u_accum, uint16_t *u_count, uint32_t *v_accum,
                         uint16_t *v_count, int zz, int hbd) {
    uint32_t luma_win[4];
    if (zz) {
        zz_partial(B, y_pre, y_pre_stride, bw, bh, y_accum, y_count, P->tf_decay_factor_fp16[0], hbd);
        if (P->tf_chroma) {
            zz_partial(B, u_pre, uv_pre_stride, bw >> ss_x, bh >> ss_y, u_accum, u_count, P->tf_decay_factor_fp16[1], hbd);
            zz_partial(B, v_pre, uv_pre_stride, bw >> ss_x, bh >> ss_y, v_accum, v_count, P->tf_decay_factor_fp16[2], hbd);
        }
        return;
    }
    planewise_partial(P, B, y_src, y_src_stride, y_pre, y_pre_stride, bw, bh, y_accum, y_count, P->tf_decay_factor_fp16[0], luma_win, 0, hbd);
    if (P->tf_chroma) {
        planewise_partial(P, B, u_src, uv_src_stride, u_pre, uv_pre_stride, bw >> ss_x, bh >> ss_y, u_accum, u_count, P->tf_decay_factor_fp16[1], luma_win, 1, hbd);
        planewise_partial(P, B, v_src, uv_src_stride, v_pre, uv_pre_stride, bw >> ss_x, bh >> ss_y, v_accum, v_count, P->tf_decay_factor_fp16[2], luma_win, 1, hbd);
    }
}

/* Whole-frame driver = what produce_temporally_filtered_pic leaves in the central picture (:3040-3400) given the motion-compensated predictions:
 * per 32x32 luma block, accum/count start from the central picture with weight 1000 (apply_filtering_central), every reference adds its
 * plane-wise term, and the result is (accum + count / 2) / count (get_final_filtered_pixels).  blocks[ref][by][bx]. */
void oracle_tf_filter_frame(const OracleTfParams *P, const void *const central[3], const int central_stride[2], const void *const *preds /*[n_refs][3]*/,
                            const int *pred_strides /*[n_refs][2]*/, const OracleTfBlock *blocks, int n_refs, int nbx, int nby, void *const out[3],
                            const int out_stride[2]) {
    const int hbd = P->encoder_bit_depth > 8, cw = 32 >> P->ss_x, ch = 32 >> P->ss_y;
    uint32_t *acc[3];
    uint16_t *cnt[3];
    for (int c = 0; c < 3; c++) { acc[c] = malloc(32 * 32 * 4); cnt[c] = malloc(32 * 32 * 2); }
    for (int by = 0; by < nby; by++)
        for (int bx = 0; bx < nbx; bx++) {
            long off[3], poff;
            for (int c = 0; c < 3; c++) {
                const int w = c ? cw : 32, h = c ? ch : 32, st = central_stride[c > 0];
                off[c] = (long)by * h * st + (long)bx * w;
                if (c && !P->tf_chroma) continue;
                for (int i = 0; i < h; i++)
                    for (int j = 0; j < w; j++) {
                        acc[c][i * w + j] = TF_WEIGHT_SCALE * px(central[c], hbd, off[c] + (long)i * st + j);
                        cnt[c][i * w + j] = TF_WEIGHT_SCALE;
                    }
            }
            for (int r = 0; r < n_refs; r++) {
                /* the accumulators are addressed with the prediction's stride (temporal_filtering.c:1136): stage the prediction block packed */
                uint16_t pb16[3][32 * 32];
                uint8_t  pb8[3][32 * 32];
                for (int c = 0; c < 3; c++) {
                    const int w = c ? cw : 32, h = c ? ch : 32, st = pred_strides[2 * r + (c > 0)];
                    poff = (long)by * h * st + (long)bx * w;
                    if (c && !P->tf_chroma) continue;
                    for (int i = 0; i < h; i++)
                        for (int j = 0; j < w; j++) {
                            const uint32_t v = px(preds[3 * r + c], hbd, poff + (long)i * st + j);
                            pb16[c][i * w + j] = (uint16_t)v; pb8[c][i * w + j] = (uint8_t)v;
                        }
                }
                const void *pp[3] = {hbd ? (void *)pb16[0] : (void *)pb8[0], hbd ? (void *)pb16[1] : (void *)pb8[1], hbd ? (void *)pb16[2] : (void *)pb8[2]};
                const void *sp[3];
                for (int c = 0; c < 3; c++) sp[c] = (const uint8_t *)central[c] + (off[c] << hbd);
                oracle_tf_planewise(P, &blocks[((long)r * nby + by) * nbx + bx], sp[0], central_stride[0], pp[0], 32, sp[1], sp[2], central_stride[1], pp[1],
                                    pp[2], cw, 32, 32, P->ss_x, P->ss_y, acc[0], cnt[0], acc[1], cnt[1], acc[2], cnt[2], P->use_zz_based_filter, hbd);
            }
            for (int c = 0; c < 3; c++) {
                const int w = c ? cw : 32, h = c ? ch : 32, st = out_stride[c > 0];
                if (c && !P->tf_chroma) continue;
                const long oo = (long)by * h * st + (long)bx * w;
                for (int i = 0; i < h; i++)
                    for (int j = 0; j < w; j++) {
                        const uint32_t v = (acc[c][i * w + j] + (cnt[c][i * w + j] >> 1)) / cnt[c][i * w + j];
                        if (hbd) ((uint16_t *)out[c])[oo + (long)i * st + j] = (uint16_t)v;
                        else ((uint8_t *)out[c])[oo + (long)i * st + j] = (uint8_t)v;
                    }
            }
        }
    for (int c = 0; c < 3; c++) { free(acc[c]); free(cnt[c]); }
}

/* svt_estimate_noise_fp16_c / svt_estimate_noise_highbd_fp16_c (:3847-3920): mean |Laplacian| over the non-edge interior pixels, x sqrt(pi/2)/6, Q16 */
int32_t oracle_estimate_noise_fp16(const void *src, int width, int height, int stride, int bd) {
    const int hbd = bd > 8;
    int64_t   sum = 0, num = 0;
    for (int i = 1; i < height - 1; i++)
        for (int j = 1; j < width - 1; j++) {
            const long k = (long)i * stride + j;
#define S(dy, dx) ((int)px(src, hbd, k + (long)(dy) * stride + (dx)))
            const int gx = (S(-1, -1) - S(-1, 1)) + (S(1, -1) - S(1, 1)) + 2 * (S(0, -1) - S(0, 1));
            const int gy = (S(-1, -1) - S(1, -1)) + (S(-1, 1) - S(1, 1)) + 2 * (S(-1, 0) - S(1, 0));
            int       ga = abs(gx) + abs(gy);
            if (hbd) ga = (ga + ((1 << (bd - 8)) >> 1)) >> (bd - 8);
            if (ga < 50) { /* EDGE_THRESHOLD */
                int v = abs(4 * S(0, 0) - 2 * (S(0, -1) + S(0, 1) + S(-1, 0) + S(1, 0)) + (S(-1, -1) + S(-1, 1) + S(1, -1) + S(1, 1)));
                if (hbd) v = (v + ((1 << (bd - 8)) >> 1)) >> (bd - 8);
                sum += v;
                num++;
            }
#undef S
        }
    if (num < 16) return -65536; /* SMOOTH_THRESHOLD: too few smooth pixels, -1.0 */
    return (int32_t)((sum * 82137) / (6 * num)); /* SQRT_PI_BY_2_FP16 */
}
