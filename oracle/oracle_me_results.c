/*
 * oracle_me_results.c -- TEST INFRASTRUCTURE (checker only, never linked into the product).
 *
 * CPU restatement of the reference's ME result formatting for one 64x64 SB: reference pruning on the integer-ME SADs,
 * the per-PU candidate lists handed to mode decision (MeSbResults), the per-SB distortion statistics and the global-motion detection
 * flags.  Pinned against the reference's own static functions through oracle/_ref/libsvtref_me.so (tests/test_me_results.py).
 *
 * Index spaces: n_idx = position in p_sb_best_sad/mv[85] (64x64 @0, 32x32 @1-4, 16x16 @5-20, 8x8 @21-84, z-order inside a level);
 * pu_index = raster position inside the level, which is how MeSbResults is addressed (z_to_raster, motion_estimation.c:2520-2531).
 */
#include <stdint.h>
#include <string.h>

typedef struct OracleMeResultsParams { /* same layout as SvtHipMeResultsParams (include/svtav1_hip.h) */
    uint32_t n_sb;
    uint8_t  num_of_list_to_search, num_of_ref_pic_to_search[2];
    uint8_t  max_cand, max_refs, max_l0;
    uint8_t  enable_me_16x16, enable_me_8x8, only_l_bwd, use_best_unipred_cand_only;
    uint8_t  prune_ref, low_resolution, gm_enabled, gm_use_distance_based_active_th;
    uint16_t prune_ref_if_me_sad_dev_bigger_than_th;
    int32_t  prune_me_candidates_th;
    uint64_t picture_number;
    uint64_t ref_picture_number[2][4];
} OracleMeResultsParams;

typedef struct OracleMeSbStats { /* = SvtHipMeSbStats */
    uint32_t me_64x64_distortion, me_32x32_distortion, me_16x16_distortion, me_8x8_distortion, me_8x8_cost_variance, rc_me_distortion;
    uint8_t  stationary_block_present_sb, rc_me_allow_gm, pad[2];
} OracleMeSbStats;

#define NPU 85
#define MAX_SAD_VALUE (128 * 128 * 255) /* motion_estimation.h:85 */

/* z_to_raster (motion_estimation.c:2520-2531) as arithmetic: de-interleave the z-order bits into (row, col) */
static int z_to_raster(int n) {
    if (n < 5) return n;
    if (n < 21) {
        const int k = n - 5;
        return 5 + ((((k >> 3) & 1) * 2 + ((k >> 1) & 1)) * 4) + ((k >> 2) & 1) * 2 + (k & 1);
    }
    const int k = n - 21;
    const int row = ((k >> 5) & 1) * 4 + ((k >> 3) & 1) * 2 + ((k >> 1) & 1);
    const int col = ((k >> 4) & 1) * 4 + ((k >> 2) & 1) * 2 + (k & 1);
    return 21 + row * 8 + col;
}
/* me_idx_85_8x8_to_16x16_conversion / me_idx_16x16_to_parent_32x32_conversion (definitions.h:2613-2632): raster child -> raster parent */
static int parent16(int n) { const int k = n - 21; return 5 + (k >> 4) * 4 + ((k & 7) >> 1); }
static int parent32(int n) { const int k = n - 5;  return 1 + (k >> 3) * 2 + ((k & 3) >> 1); }

/* MeCandidate bit-field (me_sb_results.h:28-34, LSB first): direction:2, ref_idx_l0:2, ref_idx_l1:2, ref0_list:1, ref1_list:1.
 * The reference stores `24` into the 1-bit list fields of the "other" list of a uni-pred candidate, which truncates to 0. */
static uint8_t cand(int dir, int l0, int l1, int r0, int r1) { return (uint8_t)((dir & 3) | (l0 & 3) << 2 | (l1 & 3) << 4 | (r0 & 1) << 6 | (r1 & 1) << 7); }

#define SAD(l, r, n) best_sad[((l) * 4 + (r)) * NPU + (n)]
#define MV(l, r, n)  best_mv[((l) * 4 + (r)) * NPU + (n)]

/* me_prune_ref, motion_estimation.c:1522-1566.  hme_sad of the slots the search never touched keeps its initial MAX_U32 (:3061). */
static void prune_refs(const OracleMeResultsParams *P, const uint32_t *best_sad, uint8_t *do_ref) {
    uint64_t sum[2][4];
    for (int l = 0; l < 2; l++)
        for (int r = 0; r < 4; r++) {
            sum[l][r] = 0xffffffffu;
            if (l >= P->num_of_list_to_search || r >= P->num_of_ref_pic_to_search[l]) continue;
            if (!do_ref[l * 4 + r]) { sum[l][r] = (uint64_t)MAX_SAD_VALUE * 64; continue; }
            sum[l][r] = 0;
            for (int i = 0; i < 64; i++) sum[l][r] += SAD(l, r, 21 + i);
        }
    const uint16_t th = P->prune_ref_if_me_sad_dev_bigger_than_th;
    if (th == 0xffff) return;
    uint64_t best = ~(uint64_t)0;
    for (int l = 0; l < 2; l++)
        for (int r = 0; r < 4; r++)
            if (sum[l][r] < best) best = sum[l][r];
    for (int l = 0; l < 2; l++)
        for (int r = 1; r < 4; r++)
            if ((sum[l][r] - best) * 100 > th * best) do_ref[l * 4 + r] = 0;
}

static int use_pu(const OracleMeResultsParams *P, int n) { return P->enable_me_16x16 ? (P->enable_me_8x8 || n < 21) : n < 5; }
static int num_pus(const OracleMeResultsParams *P) { return P->enable_me_16x16 ? (P->enable_me_8x8 ? 85 : 21) : 5; }

void oracle_me_results_sb(const OracleMeResultsParams *P, const uint32_t *best_sad /*[2][4][85]*/, const uint32_t *best_mv, uint8_t *do_ref /*[2][4] in/out*/,
                          uint8_t sb_width, uint8_t sb_height, uint8_t *total, uint32_t *mv_array, uint8_t *cand_array, OracleMeSbStats *st) {
    uint32_t dist[NPU];
    if (P->prune_ref) prune_refs(P, best_sad, do_ref);
    const int      nr0 = P->num_of_ref_pic_to_search[0], nr1 = P->num_of_ref_pic_to_search[1];
    uint32_t       nlist = P->num_of_list_to_search;
    const uint32_t cand_th = (uint32_t)P->prune_me_candidates_th;

    if (nr0 == 1 && nr1 == 0) { /* construct_me_candidate_array_single_ref, :2646-2697 */
        memset(total, 1, num_pus(P));
        for (int n = 0; n < NPU; n++) {
            const int pu = z_to_raster(n);
            dist[pu] = SAD(0, 0, n);
            if (!do_ref[0] || !use_pu(P, n)) continue;
            cand_array[pu * P->max_cand] = cand(0, 0, 0, 0, 0);
            mv_array[pu * P->max_refs]   = MV(0, 0, n);
        }
    } else if (nr0 == 1 && nr1 == 1) { /* construct_me_candidate_array_mrp_off, :2532-2645 */
        const uint8_t org0 = do_ref[0], org1 = nlist == 1 ? 0 : do_ref[4];
        if (nlist < 2 || !do_ref[4]) nlist = 1;
        const uint32_t th = (org0 && org1) ? cand_th : 0;
        memset(total, 1, num_pus(P));
        for (int n = 0; n < NPU; n++) {
            const int pu = z_to_raster(n), use = use_pu(P, n);
            uint8_t   blk[2] = {org0, org1};
            const uint32_t s0 = SAD(0, 0, n), s1 = SAD(1, 0, n);
            const uint32_t best = (org0 && org1) ? (s0 < s1 ? s0 : s1) : org0 ? s0 : s1;
            dist[pu] = best;
            int min_list = -1, off = 0;
            if (P->use_best_unipred_cand_only && blk[0] && blk[1]) min_list = s0 < s1 ? 0 : 1;
            for (uint32_t l = 0; l < nlist && (use || off == 0); l++) {
                if (!blk[l]) continue;
                if (th > 0 && (uint32_t)((SAD(l, 0, n) - best) * 100) > (uint32_t)(best * th)) { blk[l] = 0; continue; }
                if (min_list != -1 && min_list != (int)l) { /* keeps the MV for the bi-pred candidate */
                    if (use) mv_array[pu * P->max_refs + (l ? P->max_l0 : 0)] = MV(l, 0, n);
                    continue;
                }
                if (use) {
                    cand_array[pu * P->max_cand + off]              = cand((int)l, 0, 0, 0, l == 1);
                    mv_array[pu * P->max_refs + (l ? P->max_l0 : 0)] = MV(l, 0, n);
                }
                off++;
            }
            if (blk[0] && blk[1] && use) {
                cand_array[pu * P->max_cand + off] = cand(2, 0, 0, 0, 1);
                total[pu] = (uint8_t)(off + 1);
            }
        }
    } else { /* construct_me_candidate_array, :2698-2828 */
        for (int n = 0; n < NPU; n++) {
            const int pu = n > 4 ? z_to_raster(n) : n, use = use_pu(P, n);
            uint8_t   blk[2][4];
            uint32_t  best = ~0u;
            int       off = 0;
            for (uint32_t l = 0; l < nlist; l++)
                for (int r = 0; r < P->num_of_ref_pic_to_search[l]; r++) {
                    blk[l][r] = do_ref[l * 4 + r];
                    if (blk[l][r] && SAD(l, r, n) < best) best = SAD(l, r, n);
                }
            dist[pu] = best;
            for (uint32_t l = 0; l < nlist && (use || off == 0); l++)
                for (int r = 0; r < P->num_of_ref_pic_to_search[l] && (use || off == 0); r++) {
                    if (!blk[l][r]) continue;
                    if (cand_th > 0 && (uint32_t)((SAD(l, r, n) - best) * 100) > (uint32_t)(best * cand_th)) { blk[l][r] = 0; continue; }
                    if (use) {
                        cand_array[pu * P->max_cand + off]                  = cand((int)l, r, r, 0, l == 1);
                        mv_array[pu * P->max_refs + (l ? P->max_l0 : 0) + r] = MV(l, r, n);
                    }
                    off++;
                }
            if (nlist == 2 && use) {
                for (int a = 0; a < nr0; a++) /* (L0[a], L1[b]) */
                    for (int b = 0; b < nr1; b++) {
                        if (P->only_l_bwd && (a > 0 || b > 0)) continue;
                        if (blk[0][a] && blk[1][b]) cand_array[pu * P->max_cand + off++] = cand(2, a, b, 0, 1);
                    }
                if (!P->only_l_bwd) {
                    for (int a = 1; a < nr0; a++) /* (LAST, L0[a]) */
                        if (blk[0][0] && blk[0][a]) cand_array[pu * P->max_cand + off++] = cand(2, 0, a, 0, 0);
                    if (nr1 == 3 && blk[1][0] && blk[1][2]) cand_array[pu * P->max_cand + off++] = cand(2, 0, 2, 1, 1); /* (BWD, ALT) */
                }
            }
            if (use) total[pu] = (uint8_t)off;
        }
    }

    /* compute_distortion, :2964-3008 */
    uint32_t d64 = dist[0], d32 = 0, d16 = 0, d8 = 0;
    for (int i = 0; i < 4; i++) d32 += dist[1 + i];
    for (int i = 0; i < 16; i++) d16 += dist[5 + i];
    for (int i = 0; i < 64; i++) d8 += dist[21 + i];
    const uint64_t mean = d8 / 64;
    uint64_t       ssq = 0;
    for (int i = 0; i < 64; i++) {
        const int64_t diff = (int64_t)dist[21 + i] - (int64_t)mean;
        ssq += (uint64_t)(diff * diff);
    }
    const uint32_t pix = (uint32_t)sb_width * sb_height;
    st->me_8x8_cost_variance = (uint32_t)(ssq / 64);
    st->rc_me_distortion     = P->low_resolution ? d8 : d16;
    st->me_64x64_distortion  = (uint32_t)(d64 * 4096u) / pix; /* 32-bit product, as the reference */
    st->me_32x32_distortion  = (uint32_t)(d32 * 4096u) / pix;
    st->me_16x16_distortion  = (uint32_t)(d16 * 4096u) / pix;
    st->me_8x8_distortion    = (uint32_t)(d8 * 4096u) / pix;
    st->stationary_block_present_sb = st->rc_me_allow_gm = 0;
    st->pad[0] = st->pad[1] = 0;

    /* perform_gm_detection, :2833-2961 (reads the first candidate of each PU as just written -- or as the caller left it) */
    if (!P->gm_enabled) return;
    uint64_t cnt[2][4][2][2];
    memset(cnt, 0, sizeof(cnt));
    const int nblk = P->low_resolution ? 64 : 16;
    uint64_t  stationary = 0;
    for (int i = 0; i < nblk; i++) {
        int n = (P->low_resolution ? 21 : 5) + i;
        if (P->low_resolution) {
            if (!P->enable_me_8x8) {
                if (n >= 21) n = parent16(n);
                if (!P->enable_me_16x16 && n >= 5) n = parent32(n);
            }
        } else if (!P->enable_me_16x16 && n >= 5)
            n = parent32(n);
        const uint8_t c = cand_array[n * P->max_cand];
        const int dir = c & 3, fwd = dir == 0 || dir == 2;
        const int l = fwd ? (c >> 6) & 1 : (c >> 7) & 1, r = fwd ? (c >> 2) & 3 : (c >> 4) & 3;
        const uint64_t a = P->picture_number, b = P->ref_picture_number[l][r];
        int d, active_th;
        if (P->low_resolution) {
            d = (int16_t)((a > b ? a : b) - (a > b ? b : a));
            d = d < 0 ? -d : d;
            d = (uint16_t)d;
            active_th = P->gm_use_distance_based_active_th ? ((d >> 1) > 4 ? (d >> 1) : 4) : 4;
        } else {
            d = (int16_t)(a - b);
            d = d < 0 ? -d : d;
            d = (uint16_t)d;
            active_th = P->gm_use_distance_based_active_th ? (d * 16 > 32 ? d * 16 : 32) : 32;
        }
        const uint32_t mv = MV(l, r, n);
        const int mx = (int16_t)(mv & 0xffff) << 2, my = (int16_t)(mv >> 16) << 2;
        if (mx < -active_th) cnt[l][r][0][0]++; else if (mx > active_th) cnt[l][r][0][1]++;
        if (my < -active_th) cnt[l][r][1][0]++; else if (my > active_th) cnt[l][r][1][1]++;
        const int sth = P->low_resolution ? 0 : 4;
        if ((mx < 0 ? -mx : mx) <= sth && (my < 0 ? -my : my) <= sth) stationary++;
    }
    const uint64_t tot = (uint64_t)nblk;
    if (stationary > (tot * 5) / 100) st->stationary_block_present_sb = 1;
    for (int i = 0; i < 32; i++)
        if ((&cnt[0][0][0][0])[i] > tot / 2) st->rc_me_allow_gm = 1;
}
