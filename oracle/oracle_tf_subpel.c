/* oracle_tf_subpel.c -- TEST INFRASTRUCTURE (checker only).  The temporal filter's sub-pel motion refinement of ONE block:
 * tf_subpel_search + svt_check_position (Source/Lib/Codec/temporal_filtering.c:1560-1790) = for a handful of candidate MVs around the best one so far
 * (half-, quarter-, eighth-pel rings), motion-compensate the luma block from the reference picture (svt_aom_simple_luma_unipred ->
 * tf_inter_predictor -> svt_av1_[highbd_]convolve_{2d,x,y,2d_copy}_sr_c, enc_inter_prediction.c:3158-3265, 3392-3447; inter_prediction.c:311-420,
 * 670-790) and keep the candidate whose variance against the source block (svt_aom_varianceWxH_c, C_DEFAULT/variance.c:300-306;
 * svt_aom_highbd_10_varianceWxH_c, svt_psnr.c:160-177) is smallest.  Pinned against the reference's own (static) tf_subpel_search through
 * oracle/ref_wrap/ref_tf_subpel.c (tests/test_tf_subpel.py). */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct OracleTfSubpelParams { /* = SvtHipTfSubpelParams (include/svtav1_hip.h) */
    uint8_t  half_pel_mode, quarter_pel_mode, eight_pel_mode; /* pcs->tf_ctrls: 0 = off, 1 = all 8 neighbours, >= 2 = no diagonals */
    uint8_t  subsampling_shift;                               /* pcs->tf_ctrls.sub_sampling_shift */
    uint8_t  bit_depth;                                       /* 8 or 10 (samples u8 / u16) */
    uint8_t  pad[3];
    uint32_t early_exit_th;                                   /* me_ctx->tf_subpel_early_exit_th (0 = off) */
    uint32_t mi_rows, mi_cols;                                /* pcs->av1_cm */
    uint32_t ref_org_x, ref_org_y, ref_stride;                /* reference picture: padding origin and stride (samples) */
} OracleTfSubpelParams;

/* av1 sub_pel_filters_8 (EIGHTTAP_REGULAR) and bilinear_filters (filter.h), 16 phases x 8 taps */
static const int16_t REGULAR[16][8] = {{0, 0, 0, 128, 0, 0, 0, 0},      {0, 2, -6, 126, 8, -2, 0, 0},    {0, 2, -10, 122, 18, -4, 0, 0},
                                       {0, 2, -12, 116, 28, -8, 2, 0},  {0, 2, -14, 110, 38, -10, 2, 0}, {0, 2, -14, 102, 48, -12, 2, 0},
                                       {0, 2, -16, 94, 58, -12, 2, 0},  {0, 2, -14, 84, 66, -14, 2, 0},  {0, 2, -14, 76, 76, -14, 2, 0},
                                       {0, 2, -14, 66, 84, -14, 2, 0},  {0, 2, -12, 58, 94, -16, 2, 0},  {0, 2, -12, 48, 102, -14, 2, 0},
                                       {0, 2, -10, 38, 110, -14, 2, 0}, {0, 2, -8, 28, 116, -12, 2, 0},  {0, 0, -4, 18, 122, -10, 2, 0},
                                       {0, 0, -2, 8, 126, -6, 2, 0}};
/* sub_pel_filters_4 (inter_prediction.c:239-254): av1_get_convolve_filter_params picks the 4-tap kernel for a block dimension <= 4
 * (inter_prediction.h:147-153) -- reachable here only through the sub-sampled centre prediction of an 8x8 block (8 x 4 rows) */
static const int16_t REGULAR4[16][8] = {{0, 0, 0, 128, 0, 0, 0, 0},     {0, 0, -4, 126, 8, -2, 0, 0},    {0, 0, -8, 122, 18, -4, 0, 0},
                                        {0, 0, -10, 116, 28, -6, 0, 0}, {0, 0, -12, 110, 38, -8, 0, 0},  {0, 0, -12, 102, 48, -10, 0, 0},
                                        {0, 0, -14, 94, 58, -10, 0, 0}, {0, 0, -12, 84, 66, -10, 0, 0},  {0, 0, -12, 76, 76, -12, 0, 0},
                                        {0, 0, -10, 66, 84, -12, 0, 0}, {0, 0, -10, 58, 94, -14, 0, 0},  {0, 0, -10, 48, 102, -12, 0, 0},
                                        {0, 0, -8, 38, 110, -12, 0, 0}, {0, 0, -6, 28, 116, -10, 0, 0},  {0, 0, -4, 18, 122, -8, 0, 0},
                                        {0, 0, -2, 8, 126, -4, 0, 0}};
static void filter_of(int bilinear, int phase, int dim, int16_t f[8]) {
    if (!bilinear) { memcpy(f, dim <= 4 ? REGULAR4[phase] : REGULAR[phase], 16); return; }
    memset(f, 0, 16);
    f[3] = (int16_t)(128 - 8 * phase);
    f[4] = (int16_t)(8 * phase);
}
static int rpot(int v, int n) { return n ? (v + (1 << (n - 1))) >> n : v; }
static int clip_bd(int v, int bd) { const int mx = (1 << bd) - 1; return v < 0 ? 0 : (v > mx ? mx : v); }
static int px_at(const void *p, int hbd, long off) { return hbd ? ((const uint16_t *)p)[off] : ((const uint8_t *)p)[off]; }

/* luma prediction of a w x h block whose top-left reference sample is ref[0] (already displaced by the integer MV), row stride rs */
static void convolve_sr(const void *ref, long rs, int hbd, int bd, int w, int h, int sx, int sy, int bilinear, uint16_t *dst /* w x h */) {
    int16_t fx[8], fy[8];
    filter_of(bilinear, sx, w, fx);
    filter_of(bilinear, sy, h, fy);
    int r0 = 3, r1 = 11; /* get_conv_params_no_round (convolve.h:40-64), not compound */
    if (bd + 7 - r0 + 2 > 16) { r1 -= bd + 7 - r0 + 2 - 16; r0 += bd + 7 - r0 + 2 - 16; }
    if (!sx && !sy) {
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) dst[y * w + x] = (uint16_t)px_at(ref, hbd, y * rs + x);
    } else if (sx && !sy) {
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                int res = 0;
                for (int k = 0; k < 8; k++) res += fx[k] * px_at(ref, hbd, y * rs + x - 3 + k);
                res = rpot(res, r0);
                dst[y * w + x] = (uint16_t)clip_bd(rpot(res, 7 - r0), bd);
            }
    } else if (!sx && sy) {
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                int res = 0;
                for (int k = 0; k < 8; k++) res += fy[k] * px_at(ref, hbd, (y - 3 + k) * rs + x);
                dst[y * w + x] = (uint16_t)clip_bd(rpot(res, 7), bd);
            }
    } else {
        int16_t  *im = (int16_t *)malloc(sizeof(int16_t) * (size_t)(h + 7) * w);
        const int bits = 14 - r0 - r1, offset_bits = bd + 14 - r0;
        for (int y = 0; y < h + 7; y++)
            for (int x = 0; x < w; x++) {
                int sum = 1 << (bd + 6);
                for (int k = 0; k < 8; k++) sum += fx[k] * px_at(ref, hbd, (y - 3) * rs + x - 3 + k);
                im[y * w + x] = (int16_t)rpot(sum, r0);
            }
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                int sum = 1 << offset_bits;
                for (int k = 0; k < 8; k++) sum += fy[k] * im[(y + k) * w + x];
                int res = rpot(sum, r1) - ((1 << (offset_bits - r1)) + (1 << (offset_bits - r1 - 1)));
                if (!hbd) res = (int16_t)res; /* the 8-bit kernel narrows to ConvBufType first (inter_prediction.c:345) */
                dst[y * w + x] = (uint16_t)clip_bd(rpot(res, bits), bd);
            }
        free(im);
    }
}

/* svt_aom_simple_luma_unipred for an unscaled reference: MV in 1/8 pel, clamped like clamp_mv_to_umv_border_sb (enc_inter_prediction.c:30-50) */
void oracle_tf_luma_pred(const OracleTfSubpelParams *P, const void *ref_buffer_y, int pu_x, int pu_y, int bsize, int mv_x, int mv_y, int bilinear,
                         int subsampling_shift, uint16_t *dst /* bsize x (bsize >> shift) */) {
    const int hbd = P->bit_depth > 8, bmi = bsize >> 2;
    const int mirow = pu_y >> 2, micol = pu_x >> 2;
    const int to_top = -((mirow * 4) * 8), to_bottom = (((int)P->mi_rows - bmi - mirow) * 4) * 8;
    const int to_left = -((micol * 4) * 8), to_right = (((int)P->mi_cols - bmi - micol) * 4) * 8;
    const int spel_left = (4 + bsize) << 4, spel_right = spel_left - 16, spel_top = spel_left, spel_bottom = spel_top - 16;
    int row = (int16_t)(mv_y * 2), col = (int16_t)(mv_x * 2);
    const int min_col = to_left * 2 - spel_left, max_col = to_right * 2 + spel_right, min_row = to_top * 2 - spel_top, max_row = to_bottom * 2 + spel_bottom;
    col = col < min_col ? min_col : (col > max_col ? max_col : col);
    row = row < min_row ? min_row : (row > max_row ? max_row : row);
    col = (int16_t)col; row = (int16_t)row;
    const int  sx = col & 15, sy = row & 15;
    const long pos_x = pu_x + (col >> 4), pos_y = pu_y + (row >> 4);
    const long rs = (long)P->ref_stride;
    const void *src = hbd ? (const void *)((const uint16_t *)ref_buffer_y + P->ref_org_x + (long)P->ref_org_y * rs + pos_x + pos_y * rs)
                          : (const void *)((const uint8_t *)ref_buffer_y + P->ref_org_x + (long)P->ref_org_y * rs + pos_x + pos_y * rs);
    convolve_sr(src, rs << subsampling_shift, hbd, P->bit_depth, bsize, bsize >> subsampling_shift, sx, sy, bilinear, dst);
}

static uint64_t block_variance(const uint16_t *pred, int pstride, const void *src, long sstride, int hbd, int w, int h) {
    if (!hbd) {
        int      sum = 0;
        uint32_t sse = 0;
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                const int d = (int)pred[y * pstride + x] - px_at(src, 0, y * sstride + x);
                sum += d;
                sse += (uint32_t)(d * d);
            }
        return (uint32_t)(sse - (uint32_t)(((int64_t)sum * sum) / (w * h)));
    }
    int64_t  sum_long = 0;
    uint64_t sse_long = 0;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const int d = (int)pred[y * pstride + x] - px_at(src, 1, y * sstride + x);
            sum_long += d;
            sse_long += (uint32_t)(d * d);
        }
    const uint32_t sse = (uint32_t)((sse_long + 8) >> 4);
    const int      sum = (int)((sum_long + 2) >> 2); /* ROUND_POWER_OF_TWO on an int64: arithmetic shift */
    const int64_t  var = (int64_t)sse - (((int64_t)sum * sum) / (w * h));
    return var >= 0 ? (uint32_t)var : 0;
}

/* tf_subpel_search for one block: src = the block's top-left source sample (stride src_stride samples); *mv_x / *mv_y = starting MV in 1/8 pel */
void oracle_tf_subpel_search(const OracleTfSubpelParams *P, const void *src, int src_stride, const void *ref_buffer_y, int pu_x, int pu_y, int bsize,
                             int bilinear, int16_t *mv_x, int16_t *mv_y, uint64_t *best_dist_io) {
    const int hbd = P->bit_depth > 8, ss = P->subsampling_shift;
    uint16_t *pred = (uint16_t *)malloc(sizeof(uint16_t) * 64 * 64);
    uint64_t  best = *best_dist_io;
    int16_t   bx = *mv_x, by = *mv_y;
    const int modes[4] = {P->half_pel_mode, P->half_pel_mode, P->quarter_pel_mode, P->eight_pel_mode};
    const int steps[4] = {0, 4, 2, 1};
    for (int ring = 0; ring < 4; ring++) {
        if (ring && !modes[ring]) continue;
        const int16_t base_x = bx, base_y = by;
        const int     st = steps[ring];
        for (int i = -st; i <= st; i += (st ? st : 1))       /* xd: outer loop */
            for (int j = -st; j <= st; j += (st ? st : 1)) { /* yd */
                if (ring && i == 0 && j == 0) continue;      /* point already searched */
                /* svt_check_position */
                if (modes[ring] >= 2 && i != 0 && j != 0) continue;
                if (best == 0) continue;
                if (P->early_exit_th && best < (((uint64_t)(bsize * bsize) * P->early_exit_th) << hbd)) continue;
                const int16_t cx = (int16_t)(base_x + i), cy = (int16_t)(base_y + j);
                const int     pss = (i == 0 && j == 0) ? ss : 0; /* only the centre is predicted on the sub-sampled rows */
                oracle_tf_luma_pred(P, ref_buffer_y, pu_x, pu_y, bsize, cx, cy, bilinear, pss, pred);
                /* the prediction lands in a 64-pitch buffer (rows 0, 1, ... or 0, 2, ... when pss); the variance reads rows 0, 1 << ss, ... */
                const uint64_t d = block_variance(pred, bsize << (pss ? 0 : ss), src, (long)src_stride << ss, hbd, bsize, bsize >> ss) << ss;
                if (d < best) { best = d; bx = cx; by = cy; }
            }
    }
    *best_dist_io = best;
    *mv_x = bx; *mv_y = by;
    free(pred);
}

/* ---- the temporal filter's final motion compensation (tf_{64x64,32x32,16x16,8x8}_inter_prediction, temporal_filtering.c:2256-2620): one square block,
 * luma and -- with tf_chroma -- both 4:2:0 chroma blocks, through svt_aom_inter_prediction's uni-directional SIMPLE_TRANSLATION path
 * (enc_inter_prediction.c:4102-4400 -> svt_aom_enc_make_inter_predictor): MULTITAP_SHARP kernels (the 4-tap regular kernel for a chroma dimension <= 4,
 * inter_prediction.h:133-142), MV clamped per plane (clamp_mv_to_umv_border_sb with the plane's sub-sampling), chroma block origin ((pu >> 3) << 3) / 2. ---- */
static const int16_t SHARP[16][8] = {{0, 0, 0, 128, 0, 0, 0, 0},         {-2, 2, -6, 126, 8, -2, 2, 0},     {-2, 6, -12, 124, 16, -6, 4, -2},  {-2, 8, -18, 120, 26, -10, 6, -2},
                                     {-4, 10, -22, 116, 38, -14, 6, -2}, {-4, 10, -22, 108, 48, -18, 8, -2}, {-4, 10, -24, 100, 60, -20, 8, -2}, {-4, 10, -24, 90, 70, -22, 10, -2},
                                     {-4, 12, -24, 80, 80, -24, 12, -4}, {-2, 10, -22, 70, 90, -24, 10, -4}, {-2, 8, -20, 60, 100, -24, 10, -4}, {-2, 8, -18, 48, 108, -22, 10, -4},
                                     {-2, 6, -14, 38, 116, -22, 10, -4}, {-2, 6, -10, 26, 120, -18, 8, -2},  {-2, 4, -6, 16, 124, -12, 6, -2},   {0, 2, -2, 8, 126, -6, 2, -2}};
static void convolve_sharp(const void *ref, long rs, int hbd, int bd, int w, int h, int sx, int sy, uint16_t *dst, int dpitch) {
    const int16_t *fx = w <= 4 ? REGULAR4[sx] : SHARP[sx], *fy = h <= 4 ? REGULAR4[sy] : SHARP[sy];
    int r0 = 3, r1 = 11;
    if (bd + 7 - r0 + 2 > 16) { r1 -= bd + 7 - r0 + 2 - 16; r0 += bd + 7 - r0 + 2 - 16; }
    if (!sx && !sy) {
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) dst[y * dpitch + x] = (uint16_t)px_at(ref, hbd, y * rs + x);
    } else if (sx && !sy) {
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                int res = 0;
                for (int k = 0; k < 8; k++) res += fx[k] * px_at(ref, hbd, y * rs + x - 3 + k);
                dst[y * dpitch + x] = (uint16_t)clip_bd(rpot(rpot(res, r0), 7 - r0), bd);
            }
    } else if (!sx && sy) {
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                int res = 0;
                for (int k = 0; k < 8; k++) res += fy[k] * px_at(ref, hbd, (y - 3 + k) * rs + x);
                dst[y * dpitch + x] = (uint16_t)clip_bd(rpot(res, 7), bd);
            }
    } else {
        int16_t  *im = (int16_t *)malloc(sizeof(int16_t) * (size_t)(h + 7) * w);
        const int bits = 14 - r0 - r1, offset_bits = bd + 14 - r0;
        for (int y = 0; y < h + 7; y++)
            for (int x = 0; x < w; x++) {
                int sum = 1 << (bd + 6);
                for (int k = 0; k < 8; k++) sum += fx[k] * px_at(ref, hbd, (y - 3) * rs + x - 3 + k);
                im[y * w + x] = (int16_t)rpot(sum, r0);
            }
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                int sum = 1 << offset_bits;
                for (int k = 0; k < 8; k++) sum += fy[k] * im[(y + k) * w + x];
                int res = rpot(sum, r1) - ((1 << (offset_bits - r1)) + (1 << (offset_bits - r1 - 1)));
                if (!hbd) res = (int16_t)res;
                dst[y * dpitch + x] = (uint16_t)clip_bd(rpot(res, bits), bd);
            }
        free(im);
    }
}
/* planes[3] = the reference picture's padded buffers (buffer_y / buffer_cb / buffer_cr), strides[3]; P->ref_org_x / ref_org_y = the LUMA padding origin (the
 * chroma origin is half of it); out[3] with pitches opitch[3] receive the bsize x bsize luma block and the two (bsize / 2)^2 chroma blocks */
void oracle_tf_inter_pred(const OracleTfSubpelParams *P, const void *const *planes, const uint32_t *strides, int pu_x, int pu_y, int bsize, int mv_x, int mv_y,
                          int chroma, uint16_t *const *out, const int *opitch) {
    const int hbd = P->bit_depth > 8;
    for (int pl = 0; pl < (chroma ? 3 : 1); pl++) {
        const int ss = pl > 0, bw = bsize >> ss, bmi = bsize >> 2; /* (the MacroBlockD edges are the luma block's, :2318-2324) */
        const int mirow = pu_y >> 2, micol = pu_x >> 2;
        const int to_top = -((mirow * 4) * 8), to_bottom = (((int)P->mi_rows - bmi - mirow) * 4) * 8;
        const int to_left = -((micol * 4) * 8), to_right = (((int)P->mi_cols - bmi - micol) * 4) * 8;
        /* clamp_mv_to_umv_border_sb (enc_inter_prediction.c:30-50) with the plane's sub-sampling */
        const int spel_left = (4 + bw) << 4, spel_right = spel_left - 16, spel_top = (4 + bw) << 4, spel_bottom = spel_top - 16;
        int row = (int16_t)(mv_y * (1 << (1 - ss))), col = (int16_t)(mv_x * (1 << (1 - ss)));
        const int min_col = to_left * (1 << (1 - ss)) - spel_left, max_col = to_right * (1 << (1 - ss)) + spel_right;
        const int min_row = to_top * (1 << (1 - ss)) - spel_top, max_row = to_bottom * (1 << (1 - ss)) + spel_bottom;
        col = col < min_col ? min_col : (col > max_col ? max_col : col);
        row = row < min_row ? min_row : (row > max_row ? max_row : row);
        col = (int16_t)col; row = (int16_t)row;
        const int  sx = col & 15, sy = row & 15;
        const int  ox = ss ? ((pu_x >> 3) << 3) / 2 : pu_x, oy = ss ? ((pu_y >> 3) << 3) / 2 : pu_y;
        const long rs = (long)strides[pl], org = (long)(P->ref_org_x >> ss) + (long)(P->ref_org_y >> ss) * rs;
        const long pos = org + ox + (col >> 4) + (long)(oy + (row >> 4)) * rs;
        const void *src = hbd ? (const void *)((const uint16_t *)planes[pl] + pos) : (const void *)((const uint8_t *)planes[pl] + pos);
        convolve_sharp(src, rs, hbd, P->bit_depth, bw, bw, sx, sy, out[pl], opitch[pl]);
    }
}
