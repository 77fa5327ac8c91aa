/* oracle_tf_picture.c -- TEST INFRASTRUCTURE (checker only).  The temporal filter of one central picture = produce_temporally_filtered_pic's block loop
 * (Source/Lib/Codec/temporal_filtering.c:3037-3400) restated on the oracle's own pieces (oracle_tf_subpel_search, oracle_tf_inter_pred, oracle_tf_filter_frame),
 * in the REFERENCE's order and laziness: a block size is searched only where the reference searches it (the product computes every size up front and decides
 * afterwards; the two must agree).  Picture-level reference skips (:3105-3131) are the caller's.  Pinned through the encoder: with the driver seam
 * (integration/temporal_filtering_seam.c, SVT_HIP_TF_SEAM=1) the reference encoder's bitstream is unchanged (tests/test_encoder_identity.py). */
#include <limits.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct OracleTfSubpelParams { /* = SvtHipTfSubpelParams */
    uint8_t  half_pel_mode, quarter_pel_mode, eight_pel_mode, subsampling_shift, bit_depth, pad[3];
    uint32_t early_exit_th, mi_rows, mi_cols, ref_org_x, ref_org_y, ref_stride;
} OracleTfSubpelParams;
typedef struct OracleTfParams { /* = SvtHipTfParams */
    uint32_t tf_decay_factor_fp16[3];
    uint16_t tf_mv_dist_th;
    uint8_t  tf_chroma, use_zz_based_filter, encoder_bit_depth, ss_x, ss_y, pad;
} OracleTfParams;
typedef struct OracleTfBlock { /* = SvtHipTfBlock */
    uint64_t block_error[4];
    int16_t  mv_x[4], mv_y[4];
    uint8_t  split, pad[7];
} OracleTfBlock;
typedef struct OracleTfPictureParams { /* = SvtHipTfPictureParams */
    OracleTfSubpelParams sp;
    OracleTfParams       tf;
    uint32_t pic_w_sb, pic_h_sb, uv_stride, me_exit_th;
    uint64_t pred_error_32x32_th;
    uint8_t  use_2tap, enable_8x8_pred, use_pred_64x64_only_th, subpel_8bit, zero_motion, pad[3];
} OracleTfPictureParams;

void oracle_tf_subpel_search(const OracleTfSubpelParams *P, const void *src, int src_stride, const void *ref_buffer_y, int pu_x, int pu_y, int bsize, int bilinear,
                             int16_t *mv_x, int16_t *mv_y, uint64_t *best_dist_io);
void oracle_tf_inter_pred(const OracleTfSubpelParams *P, const void *const *planes, const uint32_t *strides, int pu_x, int pu_y, int bsize, int mv_x, int mv_y, int chroma,
                          uint16_t *const *out, const int *opitch);
void oracle_tf_filter_frame(const OracleTfParams *P, const void *const central[3], const int central_stride[2], const void *const *preds, const int *pred_strides,
                            const OracleTfBlock *blocks, int n_refs, int nbx, int nby, void *const out[3], const int out_stride[2]);

/* z-order geometry of the ME tables' entries (tab16x16 / tab8x8, motion_estimation.h:101-116): 0 = 64x64; 1 + i32; 5 + 4 i32 + i16; 21 + 16 i32 + 4 i16 + i8 */
static void slot_geometry(int slot, int *bs, int *lx, int *ly) {
    if (slot == 0) { *bs = 64; *lx = *ly = 0; }
    else if (slot < 5) { const int i = slot - 1; *bs = 32; *lx = (i & 1) * 32; *ly = (i >> 1) * 32; }
    else if (slot < 21) { const int z = slot - 5; *bs = 16; *lx = ((z >> 2) & 1) * 32 + (z & 1) * 16; *ly = ((z >> 3) & 1) * 32 + ((z >> 1) & 1) * 16; }
    else { const int z = slot - 21; *bs = 8; *lx = ((z >> 4) & 1) * 32 + ((z >> 2) & 1) * 16 + (z & 1) * 8; *ly = ((z >> 5) & 1) * 32 + ((z >> 3) & 1) * 16 + ((z >> 1) & 1) * 8; }
}

typedef struct Ctx {
    const OracleTfPictureParams *P;
    const OracleTfSubpelParams  *sp; /* the searches' parameters: P->sp, or its 8-bit form with subpel_8bit (tf_ctrls.use_8bit_subpel, :3203) */
    const void *cy, *ry;     /* central / reference luma buffers */
    long        pic0;
    int         hbd, x0, y0; /* block origin */
    const uint32_t *mv;      /* the block's 85 ME vectors */
} Ctx;
static void search(const Ctx *c, int slot, int from_sc_x, int from_sc_y, int from_sc, int16_t *mvx, int16_t *mvy, uint64_t *err) {
    int bs, lx, ly;
    slot_geometry(slot, &bs, &lx, &ly);
    const long so = c->pic0 + (long)(c->y0 + ly) * c->P->sp.ref_stride + c->x0 + lx;
    const void *src = c->hbd ? (const void *)((const uint16_t *)c->cy + so) : (const void *)((const uint8_t *)c->cy + so);
    *mvx = (int16_t)((from_sc ? from_sc_x : (int16_t)(c->mv[slot] & 0xffffu)) << 3);
    *mvy = (int16_t)((from_sc ? from_sc_y : (int16_t)(c->mv[slot] >> 16)) << 3);
    *err = INT_MAX; /* (:1866, :1980, :2117, :2236) */
    oracle_tf_subpel_search(c->sp, src, (int)c->P->sp.ref_stride, c->ry, c->x0 + lx, c->y0 + ly, bs, bs >= 32 ? c->P->use_2tap : 0, mvx, mvy, err);
}

static uint64_t var32(const void *pred, int pstride, const void *src, long sstride, int hbd, int ss) { /* :2718-2756 */
    const int rows = 32 >> ss;
    int64_t   sum = 0;
    uint64_t  sse = 0;
    for (int r = 0; r < rows; r++)
        for (int x = 0; x < 32; x++) {
            const int a = hbd ? ((const uint16_t *)pred)[(long)(r << ss) * pstride + x] : ((const uint8_t *)pred)[(long)(r << ss) * pstride + x];
            const int b = hbd ? ((const uint16_t *)src)[(long)(r << ss) * sstride + x] : ((const uint8_t *)src)[(long)(r << ss) * sstride + x];
            sum += a - b; sse += (uint64_t)((a - b) * (a - b));
        }
    uint64_t var;
    if (!hbd) var = (uint32_t)((uint32_t)sse - (uint32_t)(((int64_t)(int)sum * (int)sum) / (32 * rows)));
    else {
        const uint32_t s32 = (uint32_t)((sse + 8) >> 4);
        const int      su  = (int)((sum + 2) >> 2);
        const int64_t  v   = (int64_t)s32 - (((int64_t)su * su) / (32 * rows));
        var = v >= 0 ? (uint32_t)v : 0;
    }
    return var << ss;
}

/* central[3] / refs[n_refs][3]: whole padded buffers (one geometry); tables per reference: [n_sb][85], [n_sb][2], [n_sb]; out[3] may be central.
 * stats[5]: predictions per size 64 / 32 / 16 / 8, early-exit blocks */
int oracle_tf_picture(const OracleTfPictureParams *P, const void *const central[3], const void *const *refs, const uint32_t *const *best_sad,
                      const uint32_t *const *best_mv, const int16_t *const *hme_sc, const uint64_t *const *hme_sad, int n_refs, void *const out[3], uint32_t stats[5],
                      const void *central_y8, const void *const *refs_y8 /* subpel_8bit: the pictures' 8-bit luma buffers; else NULL */) {
    const int  hbd = P->sp.bit_depth > 8, px = hbd ? 2 : 1, chroma = P->tf.tf_chroma, ss = P->sp.subsampling_shift;
    const int  nsbx = (int)P->pic_w_sb, nsby = (int)P->pic_h_sb, n_sb = nsbx * nsby, pw = 64 * nsbx, ph = 64 * nsby, nbx = 2 * nsbx, nby = 2 * nsby;
    const long pic0 = (long)P->sp.ref_org_y * P->sp.ref_stride + P->sp.ref_org_x, cpic0 = (long)(P->sp.ref_org_y >> 1) * P->uv_stride + (P->sp.ref_org_x >> 1);
    OracleTfBlock *blocks = calloc((size_t)n_refs * nbx * nby, sizeof(*blocks));
    uint8_t **pred = calloc((size_t)n_refs * 3, sizeof(*pred));
    uint16_t *tmp[3] = {malloc(64 * 64 * 2), malloc(32 * 32 * 2), malloc(32 * 32 * 2)};
    const int tp[3] = {64, 32, 32};
    if (stats) memset(stats, 0, 5 * sizeof(uint32_t));
    const int            sp8 = hbd && P->subpel_8bit;
    OracleTfSubpelParams SP8 = P->sp;
    SP8.bit_depth = 8;
    for (int r = 0; r < n_refs; r++) {
        pred[3 * r] = calloc((size_t)pw * ph, px);
        pred[3 * r + 1] = calloc((size_t)(pw / 2) * (ph / 2), px);
        pred[3 * r + 2] = calloc((size_t)(pw / 2) * (ph / 2), px);
        const void *const planes[3] = {refs[3 * r], refs[3 * r + 1], refs[3 * r + 2]};
        const uint32_t    strides[3] = {P->sp.ref_stride, P->uv_stride, P->uv_stride};
        for (int sb = 0; sb < n_sb; sb++) {
            const int x0 = (sb % nsbx) * 64, y0 = (sb / nsbx) * 64;
#define MC(SLOT, MVX, MVY)                                                                                                                                      \
    do {                                                                                                                                                       \
        int bs_, lx_, ly_;                                                                                                                                      \
        slot_geometry(SLOT, &bs_, &lx_, &ly_);                                                                                                                  \
        oracle_tf_inter_pred(&P->sp, planes, strides, x0 + lx_, y0 + ly_, bs_, MVX, MVY, chroma, tmp, tp);                                                      \
        for (int pl_ = 0; pl_ < (chroma ? 3 : 1); pl_++) {                                                                                                      \
            const int s_ = pl_ > 0, w_ = bs_ >> s_, st_ = pw >> s_;                                                                                             \
            const int ox_ = s_ ? (((x0 + lx_) >> 3) << 3) / 2 : x0 + lx_, oy_ = s_ ? (((y0 + ly_) >> 3) << 3) / 2 : y0 + ly_;                                    \
            for (int yy = 0; yy < w_; yy++)                                                                                                                     \
                for (int xx = 0; xx < w_; xx++) {                                                                                                               \
                    const uint16_t v_ = tmp[pl_][yy * tp[pl_] + xx];                                                                                            \
                    if (hbd) ((uint16_t *)pred[3 * r + pl_])[(long)(oy_ + yy) * st_ + ox_ + xx] = v_;                                                           \
                    else pred[3 * r + pl_][(long)(oy_ + yy) * st_ + ox_ + xx] = (uint8_t)v_;                                                                    \
                }                                                                                                                                               \
        }                                                                                                                                                       \
    } while (0)
            if (P->zero_motion) { /* produce_temporally_filtered_pic_ld (:3415-3846): no ME, no refinement -- one 64x64 prediction at (0, 0), its 32x32 errors, the filter */
                MC(0, 0, 0);
                if (stats) stats[0]++;
                for (int i = 0; i < 4; i++) {
                    OracleTfBlock *B = &blocks[((size_t)r * nby + 2 * (sb / nsbx) + (i >> 1)) * nbx + 2 * (sb % nsbx) + (i & 1)];
                    const long po = (long)(y0 + (i >> 1) * 32) * pw + x0 + (i & 1) * 32, so = pic0 + (long)(y0 + (i >> 1) * 32) * P->sp.ref_stride + x0 + (i & 1) * 32;
                    B->block_error[0] = var32(pred[3 * r] + po * px, pw, (const uint8_t *)central[0] + so * px, P->sp.ref_stride, hbd, ss);
                }
                continue;
            }
            Ctx c = {P, sp8 ? &SP8 : &P->sp, sp8 ? central_y8 : central[0], sp8 ? refs_y8[r] : refs[3 * r], pic0, sp8 ? 0 : hbd, x0, y0, best_mv[r] + (size_t)sb * 85};
            const uint32_t *sd = best_sad[r] + (size_t)sb * 85;
            const int     exited = hme_sad[r][sb] < P->me_exit_th; /* motion_estimation.c:3110-3111 */
            const uint8_t th = exited ? (uint8_t)0xff : P->use_pred_64x64_only_th;
            int16_t  mv64x, mv64y, mv32x[4], mv32y[4], mv16x[16], mv16y[16], mv8x[64], mv8y[64];
            uint64_t e64, e32[4], e16[16], e8[64];
            int      split32[4] = {0, 0, 0, 0}, split16[16] = {0};
            int      p64 = 0;
            if (stats && exited) stats[4]++;
            /* the 64x64 search comes first on both branches (:3191, :3229) */
            search(&c, 0, hme_sc[r][2 * sb], hme_sc[r][2 * sb + 1], th == 0xff, &mv64x, &mv64y, &e64);
            if (th && (th == 0xff || ({ /* tf_use_64x64_pred (:2676-2690) */
                           uint32_t d32 = 0;
                           for (int i = 0; i < 4; i++) d32 += sd[1 + i];
                           const int64_t a = sd[0] > 1 ? sd[0] : 1, b = d32 > 1 ? d32 : 1;
                           (a - b) * 100 / b < (int64_t)th;
                       })))
                p64 = 1;
            else {
                for (int i = 0; i < 4; i++) search(&c, 1 + i, 0, 0, 0, &mv32x[i], &mv32y[i], &e32[i]);
                const uint64_t s32 = e32[0] + e32[1] + e32[2] + e32[3];
                if (e64 * 14 < s32 * 16 && e64 < (1u << 18)) p64 = 1; /* (:3263-3265) */
                else
                    for (int i = 0; i < 4; i++) {
                        if (e32[i] < P->pred_error_32x32_th) continue; /* no split (:3292-3296) */
                        for (int j = 0; j < 4; j++) search(&c, 5 + 4 * i + j, 0, 0, 0, &mv16x[4 * i + j], &mv16y[4 * i + j], &e16[4 * i + j]);
                        if (P->enable_8x8_pred)
                            for (int j = 0; j < 16; j++) search(&c, 21 + 16 * i + j, 0, 0, 0, &mv8x[16 * i + j], &mv8y[16 * i + j], &e8[16 * i + j]);
                        /* derive_tf_32x32_block_split_flag (:237-286) */
                        int sum = 0;
                        for (int j = 0; j < 4; j++) {
                            int sub = (int)e16[4 * i + j];
                            if (P->enable_8x8_pred) {
                                int s8 = 0;
                                for (int k = 0; k < 4; k++) s8 += (int)e8[16 * i + 4 * j + k];
                                if (!(sub * 8 < s8 * 16)) { split16[4 * i + j] = 1; e16[4 * i + j] = (uint64_t)(int64_t)s8; sub = s8; }
                            }
                            sum += sub;
                        }
                        split32[i] = !((int)e32[i] * 14 < sum * 16);
                    }
            }
            /* motion compensation (tf_64x64_ / tf_32x32_inter_prediction, :2256-2605) into the reference's prediction planes */
            if (p64) { MC(0, mv64x, mv64y); if (stats) stats[0]++; }
            for (int i = 0; i < 4; i++) {
                OracleTfBlock *B = &blocks[((size_t)r * nby + 2 * (sb / nsbx) + (i >> 1)) * nbx + 2 * (sb % nsbx) + (i & 1)];
                if (p64) { /* convert_64x64_info_to_32x32_info (:2691-2758) */
                    const long po = (long)(y0 + (i >> 1) * 32) * pw + x0 + (i & 1) * 32, so = pic0 + (long)(y0 + (i >> 1) * 32) * P->sp.ref_stride + x0 + (i & 1) * 32;
                    B->mv_x[0] = mv64x; B->mv_y[0] = mv64y;
                    B->block_error[0] = var32(pred[3 * r] + po * px, pw, (const uint8_t *)central[0] + so * px, P->sp.ref_stride, hbd, ss);
                } else if (!split32[i]) {
                    B->mv_x[0] = mv32x[i]; B->mv_y[0] = mv32y[i]; B->block_error[0] = e32[i];
                    MC(1 + i, mv32x[i], mv32y[i]); if (stats) stats[1]++;
                } else {
                    B->split = 1;
                    for (int j = 0; j < 4; j++) {
                        B->mv_x[j] = mv16x[4 * i + j]; B->mv_y[j] = mv16y[4 * i + j]; B->block_error[j] = e16[4 * i + j];
                        if (split16[4 * i + j]) {
                            for (int k = 0; k < 4; k++) MC(21 + 16 * i + 4 * j + k, mv8x[16 * i + 4 * j + k], mv8y[16 * i + 4 * j + k]);
                            if (stats) stats[3] += 4;
                        } else { MC(5 + 4 * i + j, mv16x[4 * i + j], mv16y[4 * i + j]); if (stats) stats[2]++; }
                    }
                }
            }
#undef MC
        }
    }
    /* steps 2-3 for the whole picture */
    const void *cen[3] = {(const uint8_t *)central[0] + pic0 * px, (const uint8_t *)central[1] + cpic0 * px, (const uint8_t *)central[2] + cpic0 * px};
    void       *o[3]   = {(uint8_t *)out[0] + pic0 * px, (uint8_t *)out[1] + cpic0 * px, (uint8_t *)out[2] + cpic0 * px};
    const int   cst[2] = {(int)P->sp.ref_stride, (int)P->uv_stride};
    int        *pst    = malloc(sizeof(int) * 2 * n_refs);
    for (int r = 0; r < n_refs; r++) { pst[2 * r] = pw; pst[2 * r + 1] = pw / 2; }
    OracleTfParams T = P->tf;
    T.encoder_bit_depth = P->sp.bit_depth;
    oracle_tf_filter_frame(&T, cen, cst, (const void *const *)pred, pst, blocks, n_refs, nbx, nby, o, cst);
    for (int r = 0; r < 3 * n_refs; r++) free(pred[r]);
    for (int i = 0; i < 3; i++) free(tmp[i]);
    free(pred); free(blocks); free(pst);
    return 0;
}
