/*
 * oracle_tpl.c -- TEST INFRASTRUCTURE: plain-C restatement of the SOURCE-BASED half of the TPL dispenser.
 *
 * Reference: tpl_mc_flow_dispenser_sb_generic, Codec/src_ops_process.c:519-969 -- per 16x16 (dispenser level 0) or 32x32 (level 1, complete SBs only,
 * :2048-2051) block of a picture, when pcs->tpl_src_data_ready == 0:
 *   1. intra: DC prediction from the SOURCE picture's neighbours (intra_dc_sad_path, :620-657: get_neighbor_samples_dc :362 inside the picture,
 *      svt_aom_update_neighbor_samples_array_open_loop_mb, enc_intra_prediction.c:1127-1207, at its borders; svt_aom_dc_pred[x > 0][y > 0],
 *      intra_prediction.c:1023-1073), cost = SAD against the source block; skipped when disable_intra_pred (:557);
 *   2. inter: every uni-directional ME candidate of the block's PU (:761-890): full-pel vector clamped to the reference picture + TPL_PADX/Y (:791-801),
 *      cost = SAD against the reference's SOURCE picture (tpl_ref_ds_ptr_array: input_padded_pic, :141), first strict minimum wins;
 *   3. NEWMV when the best inter cost is below the intra cost (:892): residual (rows subsampled by 1 << subsample_tx, :930-937) -> svt_av1_wht_fwd_txfm =
 *      forward DCT_DCT with the N2 / N4 partial-frequency shape (transforms.c:3640-3655) -> get_quantize_error (:224-247: svt_av1_quantize_fp, log_scale 0,
 *      svt_av1_block_error >> 2 (0 for TX_32X32), at least 1) -> srcrf_dist = recon_error << TPL_DEP_COST_SCALE_LOG2 << subsample_tx (:955; rate 0: compute_rate = 0).
 * Covered option set = tpl levels 4 and 5 (initial_rc_process.c:343-378, every preset from M3 up): use_sad_in_src_search = 1, intra_mode_end = DC_PRED,
 * subpel_depth = FULL_PEL, compute_rate = 0, scs->in_loop_ois = 1.  Pinned against the reference's own function in tests/test_tpl.py through
 * oracle/ref_wrap/ref_tpl.c.  Output = the TplSrcStats the reference stores per 16x16 cell (:958-967).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

void oracle_fwd_txfm2d(const int16_t *input, int32_t *output, uint32_t stride, int tx_type, int tx_size, int bd, int pf);

typedef struct OracleTplRef { /* = SvtHipTplRef */
    uint64_t plane_off, picture_number;
    uint32_t stride, org_x, org_y;
    uint16_t max_width, max_height;
    uint8_t  valid, pad[3];
} OracleTplRef;
typedef struct OracleTplParams { /* = SvtHipTplSrcParams */
    uint32_t width, height, aligned_width, sbs_x, n_sb;
    uint32_t src_stride;
    uint64_t src_off;
    uint8_t  dispenser_search_level, subsample_tx, pf_shape, disable_intra_pred, i_slice, enable_me_16x16, enable_me_8x8, max_cand, max_refs, max_l0, intra_mode_end, search_flags; /* the last two must be 0 here: this restatement covers tpl levels 4 / 5 */
    int16_t  quant_fp[2], round_fp[2], dequant[2];
    OracleTplRef refs[8];
} OracleTplParams;
typedef struct OracleTplSrcStats { /* = SvtHipTplSrcStats */
    int64_t  srcrf_dist, srcrf_rate;
    uint64_t ref_frame_poc;
    int16_t  mv_row, mv_col;
    int32_t  best_rf_idx;
    uint8_t  best_mode, best_intra_mode, written, pad[5];
} OracleTplSrcStats;

_Static_assert(sizeof(OracleTplSrcStats) == 40 && sizeof(OracleTplParams) == 376 && sizeof(OracleTplRef) == 40, "layout");
enum { TPL_PAD = 32 /* TPL_PADX / TPL_PADY, encode_context.h:43 */, DC_PRED_ = 0, NEWMV_ = 16 /* definitions.h: PredictionMode */, TPL_DEP_COST_SCALE_LOG2_ = 4 };

static int tx_id(int w, int h) { /* TxSize numbering, definitions.h */
    static const int W[19] = {4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64}, H[19] = {4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16};
    for (int i = 0; i < 19; i++)
        if (W[i] == w && H[i] == h) return i;
    return -1;
}
static uint32_t sad_blk(const uint8_t *a, uint32_t as, const uint8_t *b, uint32_t bs, int size) {
    uint32_t s = 0;
    for (int y = 0; y < size; y++)
        for (int x = 0; x < size; x++) s += (uint32_t)abs((int)a[(size_t)y * as + x] - (int)b[(size_t)y * bs + x]);
    return s;
}

/* one block; src = picture sample (0, 0) of the source plane */
static void tpl_src_block(const OracleTplParams *P, const uint8_t *src, const uint8_t *ref_base, int x0, int y0, int size, int me_pu, const uint8_t *tot,
                          const uint32_t *mvs, const uint8_t *cands, OracleTplSrcStats *o) {
    const uint32_t ss = P->src_stride;
    const uint8_t *blk = src + (size_t)y0 * ss + x0;
    const int      W = (int)P->width, H = (int)P->height;
    int64_t        best_intra = INT64_MAX, best_inter = INT64_MAX;
    int            best_rf = -1;
    uint64_t       best_poc = 0;
    int16_t        mvr = 0, mvc = 0;
    if (!P->disable_intra_pred) {
        /* neighbours: inside the picture they are the picture's samples; a left sample below the picture keeps the array's fill value 129, an above sample right
         * of the picture keeps 127 (memset + clipped copies of svt_aom_update_neighbor_samples_array_open_loop_mb); DC form by availability */
        int dc;
        int sa = 0, sl = 0;
        for (int i = 0; i < size; i++) {
            sa += (y0 > 0) ? ((x0 + i < W) ? src[(size_t)(y0 - 1) * ss + x0 + i] : 127) : 0;
            sl += (x0 > 0) ? ((y0 + i < H) ? src[(size_t)(y0 + i) * ss + x0 - 1] : 129) : 0;
        }
        if (x0 > 0 && y0 > 0) dc = (sa + sl + size) / (2 * size);
        else if (x0 > 0) dc = (sl + (size >> 1)) / size;
        else if (y0 > 0) dc = (sa + (size >> 1)) / size;
        else dc = 128;
        uint32_t s = 0;
        for (int y = 0; y < size; y++)
            for (int x = 0; x < size; x++) s += (uint32_t)abs((int)blk[(size_t)y * ss + x] - dc);
        best_intra = s;
    }
    const int n_cand = P->i_slice ? 0 : tot[me_pu];
    for (int i = 0; i < n_cand; i++) {
        const uint8_t c = cands[me_pu * P->max_cand + i];
        const int dir = c & 3, r0 = (c >> 2) & 3, r1 = (c >> 4) & 3;
        if (dir > 1) continue;
        const int list = dir, ref = list == 0 ? r0 : r1, rf = list * 4 + ref;
        const OracleTplRef *R = &P->refs[rf];
        if (!R->valid) continue;
        const uint32_t m = mvs[me_pu * P->max_refs + (list ? P->max_l0 : 0) + ref];
        int16_t xm = (int16_t)((int16_t)(m & 0xffff) << 3), ym = (int16_t)((int16_t)(m >> 16) << 3);
        if (x0 + (xm >> 3) < -TPL_PAD) xm = (int16_t)((-TPL_PAD - x0) << 3);
        if (x0 + size + (xm >> 3) > TPL_PAD + (int)R->max_width - 1) xm = (int16_t)(((TPL_PAD + (int)R->max_width - 1) - (x0 + size)) << 3);
        if (y0 + (ym >> 3) < -TPL_PAD) ym = (int16_t)((-TPL_PAD - y0) << 3);
        if (y0 + size + (ym >> 3) > TPL_PAD + (int)R->max_height - 1) ym = (int16_t)(((TPL_PAD + (int)R->max_height - 1) - (y0 + size)) << 3);
        const uint8_t *rp = ref_base + R->plane_off + (size_t)((int)R->org_y + y0 + ym / 8) * R->stride + (int)R->org_x + x0 + xm / 8;
        const int64_t cost = sad_blk(blk, ss, rp, R->stride, size);
        if (cost < best_inter) { best_inter = cost; best_rf = rf; best_poc = R->picture_number; mvr = ym; mvc = xm; }
    }
    memset(o, 0, sizeof(*o));
    o->written = 1;
    o->best_mode = best_inter < best_intra ? NEWMV_ : DC_PRED_;
    o->best_intra_mode = DC_PRED_;
    o->best_rf_idx = best_rf; o->ref_frame_poc = best_poc; o->mv_row = mvr; o->mv_col = mvc;
    if (o->best_mode == NEWMV_) {
        const OracleTplRef *R = &P->refs[best_rf];
        const int st = P->subsample_tx, th = size >> st, n = size * th;
        const uint8_t *rp = ref_base + R->plane_off + (size_t)((int)R->org_y + y0 + (mvr >> 3)) * R->stride + (int)R->org_x + x0 + (mvc >> 3);
        int16_t diff[32 * 32];
        int32_t coeff[32 * 32];
        for (int y = 0; y < th; y++)
            for (int x = 0; x < size; x++) diff[y * size + x] = (int16_t)((int)blk[(size_t)(y << st) * ss + x] - (int)rp[(size_t)(y << st) * R->stride + x]);
        /* (the reference passes the residual with stride size << st and rows size >> st: the same samples) */
        oracle_fwd_txfm2d(diff, coeff, (uint32_t)size, 0 /* DCT_DCT */, tx_id(size, th), 8, P->pf_shape);
        int64_t err = 0;
        for (int i = 0; i < n; i++) { /* svt_av1_quantize_fp (quantize_fp_helper_c, full_loop.c:282-342, log_scale 0, no matrices) + svt_av1_block_error */
            const int     k = i != 0;
            const int32_t c = coeff[i], sign = c < 0 ? -1 : 0, a = (c ^ sign) - sign;
            int32_t       dq = 0;
            if (((int64_t)a << 1) >= (int32_t)P->dequant[k]) {
                int64_t t = (int64_t)a + P->round_fp[k];
                t = t < -32768 ? -32768 : (t > 32767 ? 32767 : t);
                const int32_t q = (int32_t)((t * P->quant_fp[k]) >> 16);
                if (q) dq = (((int32_t)((uint32_t)q * (uint32_t)(int32_t)P->dequant[k])) ^ sign) - sign;
            }
            err += (int64_t)(c - dq) * (c - dq);
        }
        err >>= (size == 32 && th == 32) ? 0 : 2;
        if (err < 1) err = 1;
        o->srcrf_dist = (err << TPL_DEP_COST_SCALE_LOG2_) << st;
        o->srcrf_rate = 0;
    }
}

/* every block of the picture: out[(y0 >> 4) * cols16 + (x0 >> 4)], cols16 = (aligned_width + 15) >> 4; cells no block writes keep written = 0 */
void oracle_tpl_src_picture(const OracleTplParams *P, const uint8_t *src_base, const uint8_t *ref_base, const uint8_t *total_me_candidate_index,
                            const uint32_t *me_mv_array, const uint8_t *me_candidate_array, OracleTplSrcStats *out) {
    const uint8_t *src = src_base + P->src_off;
    const int      n_pus = P->enable_me_8x8 ? 85 : (P->enable_me_16x16 ? 21 : 5), cols16 = (int)((P->aligned_width + 15) >> 4);
    const int      aligned_h = (int)((P->height + 7) & ~7u);
    for (uint32_t sb = 0; sb < P->n_sb; sb++) {
        const int sx = (int)(sb % P->sbs_x) * 64, sy = (int)(sb / P->sbs_x) * 64;
        const int bw = (int)P->aligned_width - sx < 64 ? (int)P->aligned_width - sx : 64, bh = aligned_h - sy < 64 ? aligned_h - sy : 64;
        const int level = (bw == 64 && bh == 64) ? P->dispenser_search_level : 0, size = level ? 32 : 16, per = 64 / size;
        const uint8_t  *tot = total_me_candidate_index + (size_t)sb * n_pus, *cands = me_candidate_array + (size_t)sb * n_pus * P->max_cand;
        const uint32_t *mvs = me_mv_array + (size_t)sb * n_pus * P->max_refs;
        for (int by = 0; by < per; by++)
            for (int bx = 0; bx < per; bx++) {
                const int x0 = sx + bx * size, y0 = sy + by * size;
                if (x0 + (size >> 1) > (int)P->width || y0 + (size >> 1) > (int)P->height) continue; /* at least half of the block inside (:580) */
                int pu = level ? 1 + by * 2 + bx : 5 + by * 4 + bx; /* tpl_blk_idx_tab[1] (:355): z-order block -> raster PU of the ME tables */
                if (!P->enable_me_16x16) pu = (pu - 1) / 4;       /* :762-763 */
                tpl_src_block(P, src, ref_base, x0, y0, size, pu, tot, mvs, cands, &out[(size_t)(y0 >> 4) * cols16 + (x0 >> 4)]);
            }
    }
}

/* ---- the RECONSTRUCTION half (tpl_mc_flow_dispenser_sb_generic, src_ops_process.c:979-1198), same option set ----------------------------------------------------
 * Per block, in the reference's order (SBs in raster order, the blocks of an SB in z-order, tpl_blk_idx_tab :353):
 *   1. prediction into the TPL reconstruction picture (mc_flow_rec_picture_buffer): NEWMV = the block at the full-pel vector of the reference's TPL picture
 *      (:1016-1019: the reconstruction of a frame inside the sliding window, else its source picture; vectors are full-pel in this option set: subpel_depth =
 *      FULL_PEL), else DC from the RECONSTRUCTED neighbours (:1038-1070: get_neighbor_samples_dc inside the picture,
 *      svt_aom_update_neighbor_samples_array_open_loop_mb_recon, enc_intra_prediction.c:1214-1300, at its borders: the same 127 / 129 fill values as the source form);
 *   2. residual against the source on every (1 << subsample_tx)-th row -> forward DCT_DCT (partial-frequency shape) -> svt_av1_quantize_fp -> recon_error
 *      (get_quantize_error :224-247);
 *   3. when the picture is a reference or intra prediction is on (:1135) and any coefficient survived: svt_aom_inv_transform_recon8bit (DCT_DCT, bd 8) onto the
 *      prediction rows the transform saw, the rows between them become copies of the row above (:1149-1167);
 *   4. recrf_dist = recon_error << 4 << subsample_tx; a block that is not NEWMV takes it as srcrf_dist too; recrf = max(srcrf, recrf) (:1170-1180).
 * Output per block (at its top-left 16x16 cell): the four statistics BEFORE result_model_store (:266), which the caller applies. */
void oracle_inv_txfm2d_add(const int32_t *input, const uint16_t *out_r, int stride_r, uint16_t *out_w, int stride_w, int tx_type, int tx_size, int bd);

typedef struct OracleTplReconStats { /* = SvtHipTplReconStats */
    int64_t srcrf_dist, recrf_dist, srcrf_rate, recrf_rate;
    uint8_t  written, coded, pad[2]; /* coded: a coefficient survived the quantizer (eob != 0) */
    uint32_t reserved;
} OracleTplReconStats;
_Static_assert(sizeof(OracleTplReconStats) == 40, "layout");

static void tpl_recon_block(const OracleTplParams *P, const OracleTplRef *rec_refs, int is_ref, const uint8_t *src, const uint8_t *rec_ref_base,
                            const OracleTplSrcStats *s, uint8_t *rec, uint32_t rs, int x0, int y0, int size, OracleTplReconStats *o) {
    const uint32_t ss = P->src_stride;
    const int      W = (int)P->width, H = (int)P->height, st = P->subsample_tx, th = size >> st, n = size * th;
    uint8_t       *dst = rec + (size_t)y0 * rs + x0;
    if (s->best_mode == NEWMV_) {
        const OracleTplRef *R = &rec_refs[s->best_rf_idx];
        const uint8_t *rp = rec_ref_base + R->plane_off + (size_t)((int)R->org_y + y0 + (s->mv_row >> 3)) * R->stride + (int)R->org_x + x0 + (s->mv_col >> 3);
        for (int y = 0; y < size; y++) memcpy(dst + (size_t)y * rs, rp + (size_t)y * R->stride, (size_t)size);
    } else {
        int sa = 0, sl = 0, dc;
        for (int i = 0; i < size; i++) {
            sa += (y0 > 0) ? ((x0 + i < W) ? rec[(size_t)(y0 - 1) * rs + x0 + i] : 127) : 0;
            sl += (x0 > 0) ? ((y0 + i < H) ? rec[(size_t)(y0 + i) * rs + x0 - 1] : 129) : 0;
        }
        if (x0 > 0 && y0 > 0) dc = (sa + sl + size) / (2 * size);
        else if (x0 > 0) dc = (sl + (size >> 1)) / size;
        else if (y0 > 0) dc = (sa + (size >> 1)) / size;
        else dc = 128;
        for (int y = 0; y < size; y++) memset(dst + (size_t)y * rs, dc, (size_t)size);
    }
    int16_t diff[32 * 32];
    int32_t coeff[32 * 32], dq[32 * 32];
    const uint8_t *blk = src + (size_t)y0 * ss + x0;
    for (int y = 0; y < th; y++)
        for (int x = 0; x < size; x++) diff[y * size + x] = (int16_t)((int)blk[(size_t)(y << st) * ss + x] - (int)dst[(size_t)(y << st) * rs + x]);
    oracle_fwd_txfm2d(diff, coeff, (uint32_t)size, 0 /* DCT_DCT */, tx_id(size, th), 8, P->pf_shape);
    int64_t err = 0;
    int     coded = 0;
    for (int i = 0; i < n; i++) {
        const int     k = i != 0;
        const int32_t c = coeff[i], sign = c < 0 ? -1 : 0, a = (c ^ sign) - sign;
        int32_t       d = 0;
        if (((int64_t)a << 1) >= (int32_t)P->dequant[k]) {
            int64_t t = (int64_t)a + P->round_fp[k];
            t = t < -32768 ? -32768 : (t > 32767 ? 32767 : t);
            const int32_t q = (int32_t)((t * P->quant_fp[k]) >> 16);
            if (q) { d = (((int32_t)((uint32_t)q * (uint32_t)(int32_t)P->dequant[k])) ^ sign) - sign; coded = 1; }
        }
        dq[i] = d;
        err += (int64_t)(c - d) * (c - d);
    }
    err >>= (size == 32 && th == 32) ? 0 : 2;
    if (err < 1) err = 1;
    if ((!P->disable_intra_pred || is_ref) && coded) {
        uint16_t tmp[32 * 32]; /* svt_av1_inv_txfm_add_c (inv_transforms.c:3177-3193): widen, inverse at bd 8, narrow */
        for (int y = 0; y < th; y++)
            for (int x = 0; x < size; x++) tmp[y * size + x] = dst[(size_t)(y << st) * rs + x];
        oracle_inv_txfm2d_add(dq, tmp, size, tmp, size, 0, tx_id(size, th), 8);
        for (int y = 0; y < th; y++)
            for (int x = 0; x < size; x++) dst[(size_t)(y << st) * rs + x] = (uint8_t)tmp[y * size + x];
        for (int y = 0; y < size; y++)
            if (y & ((1 << st) - 1)) memcpy(dst + (size_t)y * rs, dst + (size_t)(y & ~((1 << st) - 1)) * rs, (size_t)size);
    }
    memset(o, 0, sizeof(*o));
    o->written = 1; o->coded = (uint8_t)coded;
    o->recrf_dist = (err << TPL_DEP_COST_SCALE_LOG2_) << st;
    o->recrf_rate = 0;
    o->srcrf_dist = s->best_mode == NEWMV_ ? s->srcrf_dist : o->recrf_dist;
    o->srcrf_rate = s->best_mode == NEWMV_ ? s->srcrf_rate : 0;
    if (o->srcrf_dist > o->recrf_dist) o->recrf_dist = o->srcrf_dist;
    if (o->srcrf_rate > o->recrf_rate) o->recrf_rate = o->srcrf_rate;
}

/* recon: sample (0, 0) of the picture's TPL reconstruction plane (read for the DC neighbours, written block by block); rec_refs[list * 4 + ref]: the plane NEWMV blocks copy from */
void oracle_tpl_recon_picture(const OracleTplParams *P, const OracleTplRef *rec_refs, int is_ref, const uint8_t *src_base, const uint8_t *rec_ref_base,
                              const OracleTplSrcStats *src_stats, uint8_t *recon, uint32_t recon_stride, OracleTplReconStats *out) {
    const uint8_t *src = src_base + P->src_off;
    const int      cols16 = (int)((P->aligned_width + 15) >> 4), aligned_h = (int)((P->height + 7) & ~7u);
    for (uint32_t sb = 0; sb < P->n_sb; sb++) {
        const int sx = (int)(sb % P->sbs_x) * 64, sy = (int)(sb / P->sbs_x) * 64;
        const int bw = (int)P->aligned_width - sx < 64 ? (int)P->aligned_width - sx : 64, bh = aligned_h - sy < 64 ? aligned_h - sy : 64;
        const int level = (bw == 64 && bh == 64) ? P->dispenser_search_level : 0, size = level ? 32 : 16, per = 64 / size;
        for (int k = 0; k < per * per; k++) { /* z-order */
            const int bx = (k & 1) | ((k >> 1) & 2), by = ((k >> 1) & 1) | ((k >> 2) & 2);
            const int x0 = sx + bx * size, y0 = sy + by * size;
            if (x0 + (size >> 1) > (int)P->width || y0 + (size >> 1) > (int)P->height) continue;
            const size_t cell = (size_t)(y0 >> 4) * cols16 + (x0 >> 4);
            tpl_recon_block(P, rec_refs, is_ref, src, rec_ref_base, &src_stats[cell], recon, recon_stride, x0, y0, size, &out[cell]);
        }
    }
}
