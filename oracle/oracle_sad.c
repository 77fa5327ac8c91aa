/*
 * oracle_sad.c -- TEST INFRASTRUCTURE: plain-C restatement of the reference's SAD family, used only as the checker
 * in tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing in the product links or loads this.
 *
 * Each function restates the algorithm of the reference function cited above it (paths relative to
 * /root/reference/Source/Lib).  Parity of this restatement is pinned in tests/test_oracle_pin.py against
 *   (a) oracle/_ref/libsvtref.so = the reference's own `*_c` functions compiled from the mounted sources, and
 *   (b) golden vectors under tests/golden/ produced by (a) with tools/gen_golden.py.
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>

#define MAX_SAD_VALUE (128 * 128 * 255) /* Codec/motion_estimation.h:85 */

static inline uint32_t absd(int a, int b) { return (uint32_t)(a > b ? a - b : b - a); }

/* C_DEFAULT/compute_sad_c.c:20-37 (svt_fast_loop_nxm_sad_kernel) == :209 svt_nxm_sad_kernel_helper_c
 * and the sadMxN macro family (:104-120). */
uint32_t oracle_sad_nxm(const uint8_t *src, uint32_t src_stride, const uint8_t *ref, uint32_t ref_stride, uint32_t height,
                        uint32_t width) {
    uint32_t sad = 0;
    for (uint32_t y = 0; y < height; y++)
        for (uint32_t x = 0; x < width; x++) sad += absd(src[y * src_stride + x], ref[y * ref_stride + x]);
    return sad;
}

/* C_DEFAULT/compute_sad_c.c:39-56 (svt_aom_sad_16b_kernel_c) */
uint32_t oracle_sad_16b(const uint16_t *src, uint32_t src_stride, const uint16_t *ref, uint32_t ref_stride, uint32_t height,
                        uint32_t width) {
    uint32_t sad = 0;
    for (uint32_t y = 0; y < height; y++)
        for (uint32_t x = 0; x < width; x++) sad += absd(src[y * src_stride + x], ref[y * ref_stride + x]);
    return sad;
}

/* C_DEFAULT/compute_sad_c.c:58-100 (svt_sad_loop_kernel_c): exhaustive search, raster order, strict '<',
 * best initialised to 0xffffff, even search lines skipped when (width == 16 && height <= 16 && skip_search_line). */
void oracle_sad_loop(const uint8_t *src, uint32_t src_stride, const uint8_t *ref, uint32_t ref_stride, uint32_t block_height,
                     uint32_t block_width, uint64_t *best_sad, int16_t *x_search_center, int16_t *y_search_center,
                     uint32_t src_stride_raw, uint8_t skip_search_line, int16_t search_area_width, int16_t search_area_height) {
    *best_sad = 0xffffff;
    for (int yy = 0; yy < search_area_height; yy++) {
        const uint8_t *line = ref + (size_t)yy * src_stride_raw;
        if (block_width == 16 && block_height <= 16 && skip_search_line && (yy & 1) == 0) continue;
        for (int xx = 0; xx < search_area_width; xx++) {
            uint32_t sad = 0;
            for (uint32_t y = 0; y < block_height; y++)
                for (uint32_t x = 0; x < block_width; x++) sad += absd(src[y * src_stride + x], line[xx + y * ref_stride + x]);
            if (sad < *best_sad) {
                *best_sad        = sad;
                *x_search_center = (int16_t)xx;
                *y_search_center = (int16_t)yy;
            }
        }
    }
}

/* 8x8 SAD, or 8x4 on the even rows doubled when sub_sad (Codec/motion_estimation.c:36-95, :105-126) */
static uint32_t sad8x8(const uint8_t *s, uint32_t ss, const uint8_t *r, uint32_t rs, int sub_sad) {
    uint32_t sad = 0;
    for (int y = 0; y < 8; y += sub_sad ? 2 : 1)
        for (int x = 0; x < 8; x++) sad += absd(s[y * ss + x], r[y * rs + x]);
    return sub_sad ? sad << 1 : sad;
}

static inline uint32_t pack_mv(int x, int y) { return ((uint32_t)(uint16_t)(int16_t)y << 16) | (uint16_t)(int16_t)x; }

/* The reference numbers the 64 8x8 blocks of a 64x64 SB in Z order: block index = 4 * (16x16 index) + (dy * 2 + dx),
 * and the 16 16x16 blocks themselves in Z order inside the 4 32x32 (offsets table, motion_estimation.c:341). */
static inline int z8(int bx, int by) { /* bx,by in 0..7 -> p_best_sad_8x8 index */
    return (bx & 1) | ((by & 1) << 1) | (((bx >> 1) & 1) << 2) | (((by >> 1) & 1) << 3) | ((bx >> 2) << 4) | ((by >> 2) << 5);
}

/* Codec/motion_estimation.c:781-816 (open_loop_me_fullpel_search_sblock) driving :429-474 (eight positions:
 * svt_ext_all_sad_calculation_8x8_16x16_c :335-362 + svt_ext_eight_sad_calculation_32x32_64x64_c :369-425) and
 * :476-779 (single position).  Both paths update the same per-block bests with strict '<' while positions are
 * visited in raster order, so the result is the first raster-order minimum per block.
 * Output layout: [0] 64x64, [1..4] 32x32, [5..20] 16x16, [21..84] 8x8 (me_context.h ME_TIER_ZERO_PU_*). */
void oracle_me_fullpel_search(const uint8_t *src, uint32_t src_stride, const uint8_t *ref, uint32_t ref_stride, int x_origin,
                              int y_origin, int width, int height, int sub_sad, uint32_t *best_sad /*85*/,
                              uint32_t *best_mv /*85*/) {
    for (int i = 0; i < 85; i++) { /* svt_initialize_buffer_32bits(p_sb_best_sad, 21, 1, MAX_SAD_VALUE), :1366 */
        best_sad[i] = MAX_SAD_VALUE;
        best_mv[i]  = 0;
    }
    for (int yy = 0; yy < height; yy++)
        for (int xx = 0; xx < width; xx++) {
            const uint8_t *r  = ref + (size_t)yy * ref_stride + xx;
            const uint32_t mv = pack_mv(xx + x_origin, yy + y_origin);
            uint32_t       s16[16], s32[4], s64 = 0;
            memset(s16, 0, sizeof(s16));
            memset(s32, 0, sizeof(s32));
            for (int by = 0; by < 8; by++)
                for (int bx = 0; bx < 8; bx++) {
                    const int      i8 = z8(bx, by);
                    const uint32_t v  = sad8x8(src + by * 8 * src_stride + bx * 8, src_stride, r + by * 8 * ref_stride + bx * 8,
                                               ref_stride, sub_sad);
                    s16[i8 >> 2] += v;
                    if (v < best_sad[21 + i8]) { best_sad[21 + i8] = v; best_mv[21 + i8] = mv; }
                }
            for (int i = 0; i < 16; i++) {
                s32[i >> 2] += s16[i];
                if (s16[i] < best_sad[5 + i]) { best_sad[5 + i] = s16[i]; best_mv[5 + i] = mv; }
            }
            for (int i = 0; i < 4; i++) {
                s64 += s32[i];
                if (s32[i] < best_sad[1 + i]) { best_sad[1 + i] = s32[i]; best_mv[1 + i] = mv; }
            }
            if (s64 < best_sad[0]) { best_sad[0] = s64; best_mv[0] = mv; }
        }
}

/* Codec/motion_estimation.c:335-362 + :210-333 (svt_ext_all_sad_calculation_8x8_16x16_c): 8 consecutive x positions */
void oracle_ext_all_sad_calculation_8x8_16x16(const uint8_t *src, uint32_t src_stride, const uint8_t *ref, uint32_t ref_stride,
                                              uint32_t mv, uint32_t *p_best_sad_8x8, uint32_t *p_best_sad_16x16,
                                              uint32_t *p_best_mv8x8, uint32_t *p_best_mv16x16, uint32_t p_eight_sad16x16[16][8],
                                              int sub_sad) {
    const int16_t xm = (int16_t)(mv & 0xffff), ym = (int16_t)(mv >> 16);
    for (int by = 0; by < 4; by++)
        for (int bx = 0; bx < 4; bx++) {
            const int i16 = z8(bx * 2, by * 2) >> 2;
            for (int k = 0; k < 8; k++) {
                uint32_t sum = 0;
                for (int sub = 0; sub < 4; sub++) {
                    const int      ox = bx * 16 + (sub & 1) * 8, oy = by * 16 + (sub >> 1) * 8;
                    const uint32_t v  = sad8x8(src + oy * src_stride + ox, src_stride, ref + oy * ref_stride + ox + k, ref_stride, sub_sad);
                    sum += v;
                    if (v < p_best_sad_8x8[4 * i16 + sub]) {
                        p_best_sad_8x8[4 * i16 + sub] = v;
                        p_best_mv8x8[4 * i16 + sub]   = pack_mv(xm + k, ym);
                    }
                }
                p_eight_sad16x16[i16][k] = sum;
                if (sum < p_best_sad_16x16[i16]) {
                    p_best_sad_16x16[i16] = sum;
                    p_best_mv16x16[i16]   = pack_mv(xm + k, ym);
                }
            }
        }
}

/* Codec/motion_estimation.c:369-425 (svt_ext_eight_sad_calculation_32x32_64x64_c) */
void oracle_ext_eight_sad_calculation_32x32_64x64(uint32_t p_sad16x16[16][8], uint32_t *p_best_sad_32x32, uint32_t *p_best_sad_64x64,
                                                  uint32_t *p_best_mv32x32, uint32_t *p_best_mv64x64, uint32_t mv,
                                                  uint32_t p_sad32x32[4][8]) {
    const int16_t xm = (int16_t)(mv & 0xffff), ym = (int16_t)(mv >> 16);
    for (int k = 0; k < 8; k++) {
        uint32_t s64 = 0;
        for (int i = 0; i < 4; i++) {
            const uint32_t v = p_sad16x16[4 * i][k] + p_sad16x16[4 * i + 1][k] + p_sad16x16[4 * i + 2][k] + p_sad16x16[4 * i + 3][k];
            p_sad32x32[i][k] = v;
            s64 += v;
            if (v < p_best_sad_32x32[i]) { p_best_sad_32x32[i] = v; p_best_mv32x32[i] = pack_mv(xm + k, ym); }
        }
        if (s64 < p_best_sad_64x64[0]) { p_best_sad_64x64[0] = s64; p_best_mv64x64[0] = pack_mv(xm + k, ym); }
    }
}

/* Codec/motion_estimation.c:98-164 (svt_ext_sad_calculation_8x8_16x16_c): one 16x16 block, one position */
void oracle_ext_sad_calculation_8x8_16x16(const uint8_t *src, uint32_t src_stride, const uint8_t *ref, uint32_t ref_stride,
                                          uint32_t *p_best_sad_8x8, uint32_t *p_best_sad_16x16, uint32_t *p_best_mv8x8,
                                          uint32_t *p_best_mv16x16, uint32_t mv, uint32_t *p_sad16x16, uint32_t *p_sad8x8, int sub_sad) {
    uint32_t sum = 0;
    for (int sub = 0; sub < 4; sub++) {
        const int ox = (sub & 1) * 8, oy = (sub >> 1) * 8;
        p_sad8x8[sub] = sad8x8(src + oy * src_stride + ox, src_stride, ref + oy * ref_stride + ox, ref_stride, sub_sad);
        sum += p_sad8x8[sub];
        if (p_sad8x8[sub] < p_best_sad_8x8[sub]) { p_best_sad_8x8[sub] = p_sad8x8[sub]; p_best_mv8x8[sub] = mv; }
    }
    if (sum < p_best_sad_16x16[0]) { p_best_sad_16x16[0] = sum; p_best_mv16x16[0] = mv; }
    *p_sad16x16 = sum;
}

/* Codec/motion_estimation.c:171-205 (svt_ext_sad_calculation_32x32_64x64_c) */
void oracle_ext_sad_calculation_32x32_64x64(const uint32_t *p_sad16x16, uint32_t *p_best_sad_32x32, uint32_t *p_best_sad_64x64,
                                            uint32_t *p_best_mv32x32, uint32_t *p_best_mv64x64, uint32_t mv, uint32_t *p_sad32x32) {
    uint32_t s64 = 0;
    for (int i = 0; i < 4; i++) {
        const uint32_t v = p_sad16x16[4 * i] + p_sad16x16[4 * i + 1] + p_sad16x16[4 * i + 2] + p_sad16x16[4 * i + 3];
        p_sad32x32[i] = v;
        s64 += v;
        if (v < p_best_sad_32x32[i]) { p_best_sad_32x32[i] = v; p_best_mv32x32[i] = mv; }
    }
    if (s64 < p_best_sad_64x64[0]) { p_best_sad_64x64[0] = s64; p_best_mv64x64[0] = mv; }
}

/* Codec/me_sad_calculation.c:14-17 (svt_initialize_buffer_32bits_c) */
void oracle_initialize_buffer_32bits(uint32_t *pointer, uint32_t count128, uint32_t count32, uint32_t value) {
    const uint32_t n = count128 * 4 + count32;
    for (uint32_t i = 0; i < n; i++) pointer[i] = value;
}

/* ---- a7: svt_pme_sad_loop_kernel_c (product_coding_loop.c:1900-1951) with svt_mv_err_cost (mcomp.c:44-68),
 * svt_mv_cost (mcomp.h:135-138) and svt_av1_get_mv_joint (rd_cost.c:55-60).  Parameters are passed flat (no reference structs). */
static int oracle_mv_rate(int16_t mv_row, int16_t mv_col, int16_t ref_row, int16_t ref_col, int cost_type, const int *mvjcost, const int *mvcost0,
                          const int *mvcost1, int error_per_bit) {
    const int16_t drow = (int16_t)(mv_row - ref_row), dcol = (int16_t)(mv_col - ref_col);
    const int16_t arow = (int16_t)abs(drow), acol = (int16_t)abs(dcol);
    switch (cost_type) {
    case 0:
        if (mvcost0 && mvcost1 && mvjcost) {
            const int joint = drow == 0 ? (dcol == 0 ? 0 : 1) : (dcol == 0 ? 2 : 3);
            const int cr = drow < -(1 << 14) ? -(1 << 14) : (drow > (1 << 14) ? (1 << 14) : drow);
            const int cc = dcol < -(1 << 14) ? -(1 << 14) : (dcol > (1 << 14) ? (1 << 14) : dcol);
            const int64_t v = (int64_t)(mvjcost[joint] + mvcost0[cr] + mvcost1[cc]) * error_per_bit;
            return (int)((v + ((int64_t)1 << 13)) >> 14); /* RDDIV_BITS 7 + AV1_PROB_COST_SHIFT 9 - RD_EPB_SHIFT 6 + PIXEL_TRANSFORM_ERROR_SCALE 4 */
        }
        return 0;
    case 1: return (2 * (arow + acol)) >> 3;
    case 2: return (0 * (arow + acol)) >> 3;
    case 3: return (1 * (arow + acol)) >> 3;
    case 4: {
        const int64_t v = (int64_t)((arow + acol) << 8) * error_per_bit;
        return (int)((v + ((int64_t)1 << 13)) >> 14);
    }
    default: return 0;
    }
}
void oracle_pme_sad_loop(const uint8_t *src, uint32_t src_stride, const uint8_t *ref, uint32_t ref_stride, uint32_t block_height,
                         uint32_t block_width, uint32_t *best_cost, int16_t *best_mvx, int16_t *best_mvy, int16_t start_x, int16_t start_y,
                         int16_t search_area_width, int16_t search_area_height, int16_t search_step, int16_t mvx, int16_t mvy, int16_t ref_row,
                         int16_t ref_col, int cost_type, const int *mvjcost, const int *mvcost0, const int *mvcost1, int error_per_bit) {
    int16_t col_num = 0, search_step_x = 1;
    for (int16_t ys = 0; ys < search_area_height; ys += search_step) {
        for (int16_t xs = 0; xs < search_area_width; xs += search_step_x) {
            if (((search_area_width - xs) < 8) && (col_num == 0)) continue;
            if (col_num == 7) { col_num = 0; search_step_x = search_step; }
            else { col_num++; search_step_x = 1; }
            uint32_t cost = 0;
            for (uint32_t y = 0; y < block_height; y++)
                for (uint32_t x = 0; x < block_width; x++) {
                    const int a = src[y * src_stride + x], b = ref[xs + y * ref_stride + x];
                    cost += (uint32_t)(a > b ? a - b : b - a);
                }
            const uint32_t px = (uint32_t)(start_x + xs), py = (uint32_t)(start_y + ys);
            const int16_t  mc = (int16_t)(mvx + (px * 8)), mr = (int16_t)(mvy + (py * 8));
            cost += (uint32_t)oracle_mv_rate(mr, mc, ref_row, ref_col, cost_type, mvjcost, mvcost0, mvcost1, error_per_bit);
            if (cost < *best_cost) { *best_mvx = mc; *best_mvy = mr; *best_cost = cost; }
        }
        ref += search_step * ref_stride;
    }
}

/* ---- one HME level for one (64x64 SB, reference, search region): hme_level_0 / hme_level_1 / hme_level_2 (Codec/motion_estimation.c:820-921,
 * :923-1018, :1020-1116).  Search-area placement (level 0: the region's cell of the num_hme_sa_w x num_hme_sa_h grid centred on the co-located
 * block; levels 1-2: centred on the previous level's result), clipping to the reference picture, svt_sad_loop_kernel over the clipped area (on
 * every other row, SAD doubled, unless FULL_SAD_SEARCH), result scaled to the next level's resolution.  All arithmetic in int16_t like the
 * reference.  ref_plane = buffer_y[0] of the (padded) reference at this level's resolution; src = the block's first sample. */
void oracle_hme_level(int level, int sub_sampled, int num_hme_sa_w, int num_hme_sa_h, int sr_w, int sr_h, const uint8_t *src, uint32_t src_stride,
                      const uint8_t *ref_plane, uint32_t ref_stride, int ref_org_x, int ref_org_y, int ref_width, int ref_height, int16_t org_x,
                      int16_t org_y, uint32_t block_width, uint32_t block_height, int16_t sa_width, int16_t sa_height, int16_t prev_sc_x,
                      int16_t prev_sc_y, uint64_t *best_sad, int16_t *sc_x, int16_t *sc_y) {
    sa_width = (int16_t)((sa_width + 7) & ~0x07);
    const int16_t pad_width = (int16_t)(level == 2 ? 63 : ref_org_x - 1), pad_height = (int16_t)(level == 2 ? 63 : ref_org_y - 1);
    int16_t sa_origin_x, sa_origin_y;
    if (level == 0) {
        sa_origin_x = (int16_t)(-(int16_t)((sa_width * num_hme_sa_w) >> 1) + (int16_t)(sa_width * sr_w));
        sa_origin_y = (int16_t)(-(int16_t)((sa_height * num_hme_sa_h) >> 1) + (int16_t)(sa_height * sr_h));
    } else {
        sa_origin_x = (int16_t)(-(sa_width >> 1) + prev_sc_x);
        sa_origin_y = (int16_t)(-(sa_height >> 1) + prev_sc_y);
    }
    if ((org_x + sa_origin_x) < -pad_width) {
        sa_origin_x = (int16_t)(-pad_width - org_x);
        sa_width    = (int16_t)(sa_width - (-pad_width - (org_x + sa_origin_x)));
    }
    if ((org_x + sa_origin_x) > (int16_t)ref_width - 1) sa_origin_x = (int16_t)(sa_origin_x - ((org_x + sa_origin_x) - ((int16_t)ref_width - 1)));
    if ((org_x + sa_origin_x + sa_width) > (int16_t)ref_width) {
        const int w = sa_width - ((org_x + sa_origin_x + sa_width) - (int16_t)ref_width);
        sa_width    = (int16_t)(w > 1 ? w : 1);
    }
    sa_width = (int16_t)(sa_width < 8 ? sa_width : sa_width & ~0x07);
    if ((org_y + sa_origin_y) < -pad_height) {
        sa_origin_y = (int16_t)(-pad_height - org_y);
        sa_height   = (int16_t)(sa_height - (-pad_height - (org_y + sa_origin_y)));
    }
    if ((org_y + sa_origin_y) > (int16_t)ref_height - 1) sa_origin_y = (int16_t)(sa_origin_y - ((org_y + sa_origin_y) - ((int16_t)ref_height - 1)));
    if ((org_y + sa_origin_y + sa_height) > (int16_t)ref_height) {
        const int h = sa_height - ((org_y + sa_origin_y + sa_height) - (int16_t)ref_height);
        sa_height   = (int16_t)(h > 1 ? h : 1);
    }
    const int16_t x_tl = (int16_t)(((int16_t)ref_org_x + org_x) + sa_origin_x), y_tl = (int16_t)(((int16_t)ref_org_y + org_y) + sa_origin_y);
    const uint32_t index = (uint32_t)(x_tl + y_tl * (int)ref_stride);
    const int full = !sub_sampled;
    oracle_sad_loop(src, full ? src_stride : src_stride * 2, ref_plane + index, full ? ref_stride : ref_stride * 2, full ? block_height : block_height >> 1,
                    block_width, best_sad, sc_x, sc_y, ref_stride, 0, sa_width, sa_height);
    if (!full) *best_sad *= 2;
    const int scale = level == 0 ? 4 : (level == 1 ? 2 : 1);
    *sc_x = (int16_t)((int16_t)(*sc_x + sa_origin_x) * scale);
    *sc_y = (int16_t)((int16_t)(*sc_y + sa_origin_y) * scale);
}

/* ---- integer ME of one (SB, reference) from its HME results = set_final_seach_centre_sb (motion_estimation.c:2182-2368: the first strictly
 * smallest SAD over the search regions, regions walked sr_h outer / sr_w inner) followed by integer_search_b64's geometry (:1294-1325, :1458-1508:
 * area = min(sa_min * dist, sa_max), optional enlargement for long search-centre components, division by the per-reference divisor, width
 * rounded up to 8, area centred on the HME result and clipped to the picture + 63-sample border) and open_loop_me_fullpel_search_sblock.
 * Restated for the option set without content-dependent probes: me_early_exit_th = 0, is_ref = 0, me_sr_adjustment < 2, me_8x8_var off.
 * plane pointers = buffer_y[0]; outputs p_sb_best_sad / p_sb_best_mv[85] plus the selected centre. */
typedef struct OracleIntSearch {
    int16_t  sa_min_w, sa_min_h, sa_max_w, sa_max_h;
    uint16_t dist;       /* already through svt_aom_get_scaled_picture_distance unless ME_MCTF */
    uint8_t  mv_adj_enabled, mv_adj_nearest_ref_only, ref_pic_index, sub_sad;
    uint16_t mv_size_th, sa_multiplier;
    uint32_t divisor;    /* reduce_me_sr_divisor[list][ref] */
} OracleIntSearch;
void oracle_me_integer_search(const OracleIntSearch *P, int n_regions, const uint64_t *hme_sad, const int16_t *hme_sc /*[n_regions][2]*/,
                              const uint8_t *src_plane, uint32_t src_stride, int src_org_x, int src_org_y, const uint8_t *ref_plane,
                              uint32_t ref_stride, int ref_org_x, int ref_org_y, int b64_origin_x, int b64_origin_y, int picture_width,
                              int picture_height, int16_t *sc_out /*[2]*/, uint64_t *sad_out, int16_t *area_out /*[4]: origin x, y, width, height*/,
                              uint32_t *best_sad, uint32_t *best_mv) {
    int16_t  x_search_center = hme_sc[0], y_search_center = hme_sc[1];
    uint64_t best = hme_sad[0];
    for (int i = 1; i < n_regions; i++)
        if (hme_sad[i] < best) { best = hme_sad[i]; x_search_center = hme_sc[2 * i]; y_search_center = hme_sc[2 * i + 1]; }
    sc_out[0] = x_search_center; sc_out[1] = y_search_center; *sad_out = best;

    const int16_t pad_width = 63, pad_height = 63, org_x = (int16_t)b64_origin_x, org_y = (int16_t)b64_origin_y;
    int16_t search_area_width = P->sa_min_w, search_area_height = P->sa_min_h;
    { const int w = search_area_width * P->dist, h = search_area_height * P->dist;
      search_area_width = (int16_t)(w < (uint16_t)P->sa_max_w ? w : (uint16_t)P->sa_max_w);
      search_area_height = (int16_t)(h < (uint16_t)P->sa_max_h ? h : (uint16_t)P->sa_max_h); }
    if (P->mv_adj_enabled && (!P->mv_adj_nearest_ref_only || P->ref_pic_index == 0)) {
        if ((x_search_center < 0 ? -x_search_center : x_search_center) > P->mv_size_th) search_area_width = (int16_t)(search_area_width * P->sa_multiplier);
        if ((y_search_center < 0 ? -y_search_center : y_search_center) > P->mv_size_th) search_area_height = (int16_t)(search_area_height * P->sa_multiplier);
    }
    { const uint32_t w = (uint32_t)search_area_width / P->divisor, h = (uint32_t)search_area_height / P->divisor; /* unsigned division, as the reference */
      search_area_width = (int16_t)(((w > 1 ? w : 1) + 7) & ~0x07u);
      search_area_height = (int16_t)(h > 3 ? h : 3); }
    int16_t x_search_area_origin = (int16_t)(x_search_center - (search_area_width >> 1));
    int16_t y_search_area_origin = (int16_t)(y_search_center - (search_area_height >> 1));
    /* the reference updates origin and size in two separate conditional expressions, the second evaluated with the corrected origin (:1462-1467) */
    x_search_area_origin = (int16_t)(((org_x + x_search_area_origin) < -pad_width) ? -pad_width - org_x : x_search_area_origin);
    search_area_width = (int16_t)(((org_x + x_search_area_origin) < -pad_width) ? search_area_width - (-pad_width - (org_x + x_search_area_origin)) : search_area_width);
    x_search_area_origin = (int16_t)(((org_x + x_search_area_origin) > picture_width - 1) ? x_search_area_origin - ((org_x + x_search_area_origin) - (picture_width - 1))
                                                                                            : x_search_area_origin);
    if ((org_x + x_search_area_origin + search_area_width) > picture_width) {
        const int w = search_area_width - ((org_x + x_search_area_origin + search_area_width) - picture_width);
        search_area_width = (int16_t)(w > 1 ? w : 1);
    }
    search_area_width = (int16_t)(search_area_width < 8 ? search_area_width : search_area_width & ~0x07);
    y_search_area_origin = (int16_t)(((org_y + y_search_area_origin) < -pad_height) ? -pad_height - org_y : y_search_area_origin);
    search_area_height = (int16_t)(((org_y + y_search_area_origin) < -pad_height) ? search_area_height - (-pad_height - (org_y + y_search_area_origin)) : search_area_height);
    y_search_area_origin = (int16_t)(((org_y + y_search_area_origin) > picture_height - 1) ? y_search_area_origin - ((org_y + y_search_area_origin) - (picture_height - 1))
                                                                                             : y_search_area_origin);
    if ((org_y + y_search_area_origin + search_area_height) > picture_height) {
        const int h = search_area_height - ((org_y + y_search_area_origin + search_area_height) - picture_height);
        search_area_height = (int16_t)(h > 1 ? h : 1);
    }
    area_out[0] = x_search_area_origin; area_out[1] = y_search_area_origin; area_out[2] = search_area_width; area_out[3] = search_area_height;
    const uint8_t *src = src_plane + (size_t)(src_org_y + b64_origin_y) * src_stride + src_org_x + b64_origin_x;
    const uint8_t *ref = ref_plane + (ptrdiff_t)(ref_org_y + b64_origin_y + y_search_area_origin) * ref_stride + ref_org_x + b64_origin_x + x_search_area_origin;
    oracle_me_fullpel_search(src, src_stride, ref, ref_stride, x_search_area_origin, y_search_area_origin, search_area_width, search_area_height, P->sub_sad,
                             best_sad, best_mv);
}
