/* oracle_lr_search.c -- TEST INFRASTRUCTURE (checker only).  The per-unit ("seg") half of the loop-restoration search of one plane
 * (restoration_seg_search, Source/Lib/Codec/restoration_pick.c:1448-1527): for every restoration unit
 *   - search_norestore_seg (:1409-1418): SSE of the unrestored unit;
 *   - search_wiener_seg (:1281-1359): compute_stats -> wiener_decompose_sep_sym (:754-923, integer alternating least squares) -> finalize_sym_filter
 *     (:962-991) -> compute_score (:925-960) -> finer_tile_search_wiener_seg (:1027-1131: coordinate descent on the taps, every trial = filter the
 *     unit + SSE against the source);
 *   - search_sgrproj_seg (:1205-1249): search_selfguided_restoration (:542-640: for every parameter set, self-guided filter -> projection
 *     (svt_get_proj_subspace) -> encode_xq -> finer_search_pixel_proj_error :320-411) -> SSE of the unit restored with the winner.
 * The picture-level decisions (search_*_finish, rate costs against the previous unit's coefficients: serial) stay with the encoder.
 * scs->use_boundaries_in_rest_search is 0 (enc_handle.c:4129), so every trial filters the plain, edge-extended plane (restoration.c:1115-1131).
 * Pinned against the reference's own static functions through oracle/ref_wrap/ref_lr_search.c (tests/test_lr_search.py). */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

void    oracle_wiener_convolve_add_src(const void *src, int src_stride, void *dst, int dst_stride, const int16_t *fx, const int16_t *fy, int w, int h, int bd,
                                       int highbd);
void    oracle_selfguided_restoration(const void *dgd, int width, int height, int stride, int32_t *flt0, int32_t *flt1, int flt_stride, int idx, int bd, int highbd);
void    oracle_apply_selfguided_restoration(const void *dat, int width, int height, int stride, int eps, const int32_t *xqd, void *dst, int dst_stride, int bd,
                                            int highbd);
int     oracle_sgr_r(int idx, int k);
void    oracle_compute_stats(int win, const void *dgd, const void *src, int h_start, int h_end, int v_start, int v_end, int dgd_stride, int src_stride, int64_t *M,
                             int64_t *H, int bit_depth);
int64_t oracle_pixel_proj_error(const void *src, int width, int height, int src_stride, const void *dat, int dat_stride, const int32_t *flt0, int flt0_stride,
                                const int32_t *flt1, int flt1_stride, const int32_t *xq, int r0, int r1, int is16);
void    oracle_get_proj_subspace(const void *src, int width, int height, int src_stride, const void *dat, int dat_stride, int is16, const int32_t *flt0,
                                 int flt0_stride, const int32_t *flt1, int flt1_stride, int32_t *xq, int r0, int r1);

typedef struct OracleLrSearchParams { /* = SvtHipLrSearchParams (include/svtav1_hip.h) with host pointers */
    const void *dgd;                  /* plane to restore (CDEF output), origin at sample (0, 0), edges extended by >= 3 samples (+ 1 more on the right) */
    const void *src;                  /* source plane */
    uint32_t    dgd_stride, src_stride, width, height, unit_size;
    uint8_t     ss_y, highbd, bit_depth;
    uint8_t     wn_enabled, wiener_win, wn_use_refinement, wn_max_one_refinement_step; /* cm->wn_filter_ctrls (wiener_win resolved: 7, 5 or 3) */
    uint8_t     sg_enabled, sg_start_ep, sg_end_ep, sg_ep_inc, sg_refine; /* the ep loop of search_selfguided_restoration, resolved (:560-579) */
    uint8_t     pad[3];
} OracleLrSearchParams;
typedef struct OracleLrSearchUnit { /* what the seg functions leave in RestUnitSearchInfo */
    int64_t sse[3];                 /* [RESTORE_NONE], [RESTORE_WIENER] (INT64_MAX: filter rejected), [RESTORE_SGRPROJ] */
    int16_t vfilter[8], hfilter[8]; /* WienerInfo */
    int32_t ep, xqd[2];             /* SgrprojInfo */
    int32_t pad;
} OracleLrSearchUnit;
typedef struct OracleLrPrevUnit { int32_t use; int16_t vfilter[8], hfilter[8]; } OracleLrPrevUnit; /* use_prev_frame_coeffs (:1297-1302) */

#define WIENER_TAP_SCALE_FACTOR ((int64_t)1 << 16)
#define WIENER_FILT_STEP 128
static const int TAP_MIN[3] = {-5, -23, -17}, TAP_MAX[3] = {10, 8, 46}; /* WIENER_FILT_TAPn_MINV / MAXV (restoration.h) */

static int px(const void *p, int hbd, long off) { return hbd ? ((const uint16_t *)p)[off] : ((const uint8_t *)p)[off]; }
static int64_t unit_sse(const void *a, long astride, const void *b, long bstride, int hbd, int h0, int h1, int v0, int v1) {
    int64_t s = 0; /* svt_aom_get_y_sse_part / highbd (sse_restoration_unit :56-60) */
    for (int y = v0; y < v1; y++)
        for (int x = h0; x < h1; x++) {
            const int d = px(a, hbd, y * astride + x) - px(b, hbd, y * bstride + x);
            s += (int64_t)d * d;
        }
    return s;
}

/* ---- Wiener: integer alternating least squares (restoration_pick.c:747-923) ---- */
static int wrap_index(int i, int win) { const int h1 = (win >> 1) + 1; return i >= h1 ? win - 1 - i : i; }
static int linsolve_wiener(int n, int64_t *A, int stride, int64_t *b, int32_t *x) {
    for (int k = 0; k < n - 1; k++) {
        for (int i = n - 1; i > k; i--)
            if (llabs(A[(i - 1) * stride + k]) < llabs(A[i * stride + k])) {
                for (int j = 0; j < n; j++) { const int64_t c = A[i * stride + j]; A[i * stride + j] = A[(i - 1) * stride + j]; A[(i - 1) * stride + j] = c; }
                const int64_t c = b[i]; b[i] = b[i - 1]; b[i - 1] = c;
            }
        for (int i = k; i < n - 1; i++) {
            if (A[k * stride + k] == 0) return 0;
            const int64_t c = A[(i + 1) * stride + k], cd = A[k * stride + k];
            for (int j = 0; j < n; j++) A[(i + 1) * stride + j] -= c / 256 * A[k * stride + j] / cd * 256;
            b[i + 1] -= c * b[k] / cd;
        }
    }
    for (int i = n - 1; i >= 0; i--) {
        if (A[i * stride + i] == 0) return 0;
        int64_t c = 0;
        for (int j = i + 1; j <= n - 1; j++) c += A[i * stride + j] * x[j] / WIENER_TAP_SCALE_FACTOR;
        x[i] = (int32_t)(WIENER_TAP_SCALE_FACTOR * (b[i] - c) / A[i * stride + i]);
    }
    return 1;
}
/* update_a_sep_sym (fix_b = 1: solve for a with b fixed) / update_b_sep_sym; M[i * win + j], H[(i * win + j) * win2 + k * win + l] */
static void update_sep_sym(int win, const int64_t *M, const int64_t *H, int32_t *a, int32_t *b, int fix_b) {
    const int win2 = win * win, h1 = (win >> 1) + 1;
    int32_t   S[7];
    int64_t   A[4], B[16];
    memset(A, 0, sizeof(A));
    memset(B, 0, sizeof(B));
    for (int i = 0; i < win; i++)
        for (int j = 0; j < win; j++) {
            if (fix_b) A[wrap_index(j, win)] += M[i * win + j] * b[i] / WIENER_TAP_SCALE_FACTOR;
            else A[wrap_index(i, win)] += M[i * win + j] * a[j] / WIENER_TAP_SCALE_FACTOR;
        }
    for (int i = 0; i < win; i++)
        for (int j = 0; j < win; j++)
            for (int k = 0; k < win; k++)
                for (int l = 0; l < win; l++) {
                    if (fix_b) /* hc[j * win + i][k * win2 + l] = H[j * win * win2 + i * win + k * win2 + l] */
                        B[wrap_index(l, win) * h1 + wrap_index(k, win)] +=
                            H[(long)j * win * win2 + i * win + k * win2 + l] * b[i] / WIENER_TAP_SCALE_FACTOR * b[j] / WIENER_TAP_SCALE_FACTOR;
                    else
                        B[wrap_index(j, win) * h1 + wrap_index(i, win)] +=
                            H[(long)i * win * win2 + j * win + k * win2 + l] * a[k] / WIENER_TAP_SCALE_FACTOR * a[l] / WIENER_TAP_SCALE_FACTOR;
                }
    const int64_t a_last = A[h1 - 1];
    for (int i = 0; i < h1 - 1; i++) A[i] -= a_last * 2 + B[i * h1 + h1 - 1] - 2 * B[(h1 - 1) * h1 + (h1 - 1)];
    for (int i = 0; i < h1 - 1; i++)
        for (int j = 0; j < h1 - 1; j++) B[i * h1 + j] -= 2 * (B[i * h1 + (h1 - 1)] + B[(h1 - 1) * h1 + j] - 2 * B[(h1 - 1) * h1 + (h1 - 1)]);
    if (linsolve_wiener(h1 - 1, B, h1, A, S)) {
        S[h1 - 1] = WIENER_TAP_SCALE_FACTOR;
        for (int i = h1; i < win; i++) { S[i] = S[win - 1 - i]; S[h1 - 1] -= 2 * S[i]; }
        memcpy(fix_b ? a : b, S, win * sizeof(int32_t));
    }
}
void oracle_wiener_decompose_sep_sym(int win, const int64_t *M, const int64_t *H, int32_t *a, int32_t *b) {
    static const int init_filt[7] = {3, -7, 15, 106, 15, -7, 3}; /* WIENER_FILT_TAPn_MIDV (tap 3 = 128 - 2 * (3 - 7 + 15)) */
    const int        plane_off    = (7 - win) >> 1;
    for (int i = 0; i < win; i++) a[i] = b[i] = WIENER_TAP_SCALE_FACTOR / WIENER_FILT_STEP * init_filt[i + plane_off];
    for (int iter = 1; iter < 5 /* NUM_WIENER_ITERS */; iter++) {
        update_sep_sym(win, M, H, a, b, 1);
        update_sep_sym(win, M, H, a, b, 0);
    }
}
void oracle_wiener_finalize_sym_filter(int win, const int32_t *f, int16_t *fi /* [8] */) {
    const int hw = win >> 1;
    for (int i = 0; i < hw; i++) {
        const int64_t dividend = (int64_t)f[i] * WIENER_FILT_STEP, divisor = WIENER_TAP_SCALE_FACTOR;
        fi[i] = (int16_t)(dividend < 0 ? (dividend - divisor / 2) / divisor : (dividend + divisor / 2) / divisor);
    }
#define CLIP3(v, lo, hi) ((v) < (lo) ? (lo) : ((v) > (hi) ? (hi) : (v)))
    if (win == 7) {
        fi[0] = (int16_t)CLIP3(fi[0], TAP_MIN[0], TAP_MAX[0]); fi[1] = (int16_t)CLIP3(fi[1], TAP_MIN[1], TAP_MAX[1]); fi[2] = (int16_t)CLIP3(fi[2], TAP_MIN[2], TAP_MAX[2]);
    } else { /* (the reference's chroma specialisation reads fi[1] / fi[0] whatever the window: 5- and 3-tap windows both land here) */
        fi[2] = (int16_t)CLIP3(fi[1], TAP_MIN[2], TAP_MAX[2]); fi[1] = (int16_t)CLIP3(fi[0], TAP_MIN[1], TAP_MAX[1]); fi[0] = 0;
    }
    fi[6] = fi[0]; fi[5] = fi[1]; fi[4] = fi[2];
    fi[3] = (int16_t)(-2 * (fi[0] + fi[1] + fi[2]));
    fi[7] = 0;
}
int64_t oracle_wiener_compute_score(int win, const int64_t *M, const int64_t *H, const int16_t *vfilt, const int16_t *hfilt) {
    int32_t   ab[49];
    int16_t   a[7], b[7];
    int64_t   P = 0, Q = 0;
    const int plane_off = (7 - win) >> 1, win2 = win * win;
    a[3] = b[3] = WIENER_FILT_STEP;
    for (int i = 0; i < 3; i++) {
        a[i] = a[6 - i] = vfilt[i]; b[i] = b[6 - i] = hfilt[i];
        a[3] = (int16_t)(a[3] - 2 * a[i]); b[3] = (int16_t)(b[3] - 2 * b[i]);
    }
    memset(ab, 0, sizeof(ab));
    for (int k = 0; k < win; k++)
        for (int l = 0; l < win; l++) ab[k * win + l] = a[l + plane_off] * b[k + plane_off];
    for (int k = 0; k < win2; k++) {
        P += ab[k] * M[k] / WIENER_FILT_STEP / WIENER_FILT_STEP;
        for (int l = 0; l < win2; l++) Q += ab[k] * H[k * win2 + l] * ab[l] / WIENER_FILT_STEP / WIENER_FILT_STEP / WIENER_FILT_STEP / WIENER_FILT_STEP;
    }
    return (Q - 2 * P) - (H[(win2 >> 1) * win2 + (win2 >> 1)] - 2 * M[win2 >> 1]);
}

/* try_restoration_unit_seg (:129-165) without stripe boundaries: restore the unit into `tmp` (unit-sized, pitch w) and return its SSE */
typedef struct { int h0, h1, v0, v1; } Rect;
static int64_t try_wiener(const OracleLrSearchParams *P, const Rect *r, const int16_t *vf, const int16_t *hf, void *tmp) {
    const int w = r->h1 - r->h0, h = r->v1 - r->v0, hbd = P->highbd, bs = hbd ? 2 : 1;
    oracle_wiener_convolve_add_src((const uint8_t *)P->dgd + ((long)r->v0 * P->dgd_stride + r->h0) * bs, (int)P->dgd_stride, tmp, w, hf, vf, w, h, P->bit_depth, hbd);
    return unit_sse(tmp, w, (const uint8_t *)P->src + ((long)r->v0 * P->src_stride + r->h0) * bs, P->src_stride, hbd, 0, w, 0, h);
}
static int64_t try_sgr(const OracleLrSearchParams *P, const Rect *r, int ep, const int32_t *xqd, void *tmp) {
    const int w = r->h1 - r->h0, h = r->v1 - r->v0, hbd = P->highbd, bs = hbd ? 2 : 1;
    oracle_apply_selfguided_restoration((const uint8_t *)P->dgd + ((long)r->v0 * P->dgd_stride + r->h0) * bs, w, h, (int)P->dgd_stride, ep, xqd, tmp, w, P->bit_depth, hbd);
    return unit_sse(tmp, w, (const uint8_t *)P->src + ((long)r->v0 * P->src_stride + r->h0) * bs, P->src_stride, hbd, 0, w, 0, h);
}

/* finer_tile_search_wiener_seg (:1027-1131); returns the SSE, refines vf / hf in place; *trials counts the filterings (diagnostic) */
static int64_t finer_search_wiener(const OracleLrSearchParams *P, const Rect *r, int16_t *vf, int16_t *hf, void *tmp, int *trials) {
    const int plane_off = (7 - P->wiener_win) >> 1;
    int64_t   err       = try_wiener(P, r, vf, hf, tmp);
    *trials = 1;
    if (!P->wn_use_refinement) return err;
    const int start_step = 4, end_step = P->wn_max_one_refinement_step ? 4 : 1;
    for (int s = start_step; s >= end_step; s >>= 1)
        for (int dir = 0; dir < 2; dir++) { /* the horizontal taps first, then the vertical ones */
            int16_t *f = dir ? vf : hf;
            for (int p = plane_off; p < 3; p++) {
                int skip = 0;
                for (int sign = -1; sign <= 1; sign += 2) {
                    for (;;) {
                        if (sign < 0 ? f[p] - s >= TAP_MIN[p] : f[p] + s <= TAP_MAX[p]) {
                            f[p] = (int16_t)(f[p] + sign * s); f[6 - p] = (int16_t)(f[6 - p] + sign * s); f[3] = (int16_t)(f[3] - 2 * sign * s);
                            const int64_t err2 = try_wiener(P, r, vf, hf, tmp);
                            (*trials)++;
                            if (err2 > err) {
                                f[p] = (int16_t)(f[p] - sign * s); f[6 - p] = (int16_t)(f[6 - p] - sign * s); f[3] = (int16_t)(f[3] + 2 * sign * s);
                            } else {
                                err = err2;
                                if (sign < 0) skip = 1;
                                if (s == start_step && !P->wn_max_one_refinement_step) continue; /* at the largest step keep moving in the same direction */
                            }
                        }
                        break;
                    }
                    if (skip) break;
                }
                if (skip) break; /* (:1062-1063: a successful downward move ends the loop over the TAPS of this direction, not just this tap) */
            }
        }
    return err;
}

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static void decode_xq(const int32_t *xqd, int32_t *xq, int r0, int r1) { /* svt_decode_xq (restoration.c:634-645) */
    if (r0 == 0) { xq[0] = 0; xq[1] = 128 - xqd[1]; }
    else if (r1 == 0) { xq[0] = xqd[0]; xq[1] = 0; }
    else { xq[0] = xqd[0]; xq[1] = 128 - xq[0] - xqd[1]; }
}
/* search_selfguided_restoration (:542-640) */
static void search_sgr(const OracleLrSearchParams *P, const Rect *r, int32_t *flt0, int32_t *flt1, int32_t *best_ep, int32_t *best_xqd) {
    const int  w = r->h1 - r->h0, h = r->v1 - r->v0, hbd = P->highbd, bs = hbd ? 2 : 1;
    const void *dat = (const uint8_t *)P->dgd + ((long)r->v0 * P->dgd_stride + r->h0) * bs, *src = (const uint8_t *)P->src + ((long)r->v0 * P->src_stride + r->h0) * bs;
    static const int tap_min[2] = {-96, -32}, tap_max[2] = {31, 95}; /* SGRPROJ_PRJ_MIN0 / MAX0, MIN1 / MAX1 */
    int64_t besterr = -1;
    *best_ep = 0; best_xqd[0] = best_xqd[1] = 0;
    for (int ep = P->sg_start_ep; ep < P->sg_end_ep; ep += P->sg_ep_inc) {
        const int r0 = oracle_sgr_r(ep, 0), r1 = oracle_sgr_r(ep, 1);
        int32_t   exq[2], xqd[2], xq[2];
        oracle_selfguided_restoration(dat, w, h, (int)P->dgd_stride, flt0, flt1, w, ep, P->bit_depth, hbd); /* apply_sgr: per processing unit, same samples */
        oracle_get_proj_subspace(src, w, h, (int)P->src_stride, dat, (int)P->dgd_stride, hbd, flt0, w, flt1, w, exq, r0, r1);
        if (r0 == 0) { xqd[0] = 0; xqd[1] = clampi(128 - exq[1], tap_min[1], tap_max[1]); } /* encode_xq (:500-511) */
        else if (r1 == 0) { xqd[0] = clampi(exq[0], tap_min[0], tap_max[0]); xqd[1] = clampi(128 - xqd[0], tap_min[1], tap_max[1]); }
        else { xqd[0] = clampi(exq[0], tap_min[0], tap_max[0]); xqd[1] = clampi(128 - xqd[0] - exq[1], tap_min[1], tap_max[1]); }
#define PROJ_ERR() (decode_xq(xqd, xq, r0, r1), oracle_pixel_proj_error(src, w, h, (int)P->src_stride, dat, (int)P->dgd_stride, flt0, w, flt1, w, xq, r0, r1, hbd))
        int64_t err = PROJ_ERR(); /* finer_search_pixel_proj_error (:320-411), start_step 2 */
        if (P->sg_refine)
            for (int s = 2; s >= 1; s >>= 1)
                for (int p = 0; p < 2; p++) {
                    if ((r0 == 0 && p == 0) || (r1 == 0 && p == 1)) continue;
                    int skip = 0;
                    for (int sign = -1; sign <= 1; sign += 2) {
                        for (;;) {
                            if (sign < 0 ? xqd[p] - s >= tap_min[p] : xqd[p] + s <= tap_max[p]) {
                                xqd[p] += sign * s;
                                const int64_t err2 = PROJ_ERR();
                                if (err2 > err) xqd[p] -= sign * s;
                                else {
                                    err = err2;
                                    if (sign < 0) skip = 1;
                                    if (s == 2) continue;
                                }
                            }
                            break;
                        }
                        if (skip) break;
                    }
                    if (skip) break; /* (:372-373: ends the loop over p) */
                }
        if (besterr == -1 || err < besterr) { *best_ep = ep; besterr = err; best_xqd[0] = xqd[0]; best_xqd[1] = xqd[1]; }
    }
}

/* unit grid of svt_aom_foreach_rest_unit_in_frame (restoration.c:1240-1330): units of unit_size, the last absorbing a remainder below 3/2 unit,
 * rows shifted up by 8 >> ss_y */
int oracle_lr_unit_rect(const OracleLrSearchParams *P, int idx, int32_t *rect /* h0 h1 v0 v1 */) {
    const int us = (int)P->unit_size, w = (int)P->width, h = (int)P->height, off = 8 >> P->ss_y;
    const int nvu = (h + (us >> 1)) / us > 0 ? (h + (us >> 1)) / us : 1, nhu = (w + (us >> 1)) / us > 0 ? (w + (us >> 1)) / us : 1;
    if (idx < 0) return nvu * nhu;
    const int ur = idx / nhu, uc = idx % nhu;
    rect[0] = uc * us; rect[1] = uc == nhu - 1 ? w : (uc + 1) * us;
    rect[2] = ur == 0 ? 0 : ur * us - off; rect[3] = ur == nvu - 1 ? h : (ur + 1) * us - off;
    return nvu * nhu;
}

/* restoration_seg_search for every unit of the plane; trials (optional) receives the number of Wiener filterings per unit */
void oracle_lr_search_plane(const OracleLrSearchParams *P, const OracleLrPrevUnit *prev, OracleLrSearchUnit *out, int32_t *trials) {
    const int n = oracle_lr_unit_rect(P, -1, NULL), hbd = P->highbd;
    void     *tmp  = malloc((size_t)P->width * P->height * 2 + 64);
    int32_t  *flt0 = (int32_t *)malloc((size_t)P->width * P->height * 4 + 64), *flt1 = (int32_t *)malloc((size_t)P->width * P->height * 4 + 64);
    int64_t  *M = (int64_t *)malloc(49 * 8), *H = (int64_t *)malloc(49 * 49 * 8);
    for (int u = 0; u < n; u++) {
        Rect    r;
        int32_t rc[4];
        oracle_lr_unit_rect(P, u, rc);
        r.h0 = rc[0]; r.h1 = rc[1]; r.v0 = rc[2]; r.v1 = rc[3];
        OracleLrSearchUnit *o = &out[u];
        memset(o, 0, sizeof(*o));
        o->sse[1] = o->sse[2] = INT64_MAX; /* a tool that is off never reads as a perfect restoration (the reference leaves rusi->sse[] of a disabled tool untouched) */
        o->sse[0] = unit_sse(P->dgd, P->dgd_stride, P->src, P->src_stride, hbd, r.h0, r.h1, r.v0, r.v1);
        if (trials) trials[u] = 0;
        if (P->wn_enabled) {
            int ok = 1;
            if (prev && prev[u].use) {
                memcpy(o->vfilter, prev[u].vfilter, 16); memcpy(o->hfilter, prev[u].hfilter, 16);
            } else {
                int32_t vd[7], hd[7];
                oracle_compute_stats(P->wiener_win, P->dgd, P->src, r.h0, r.h1, r.v0, r.v1, (int)P->dgd_stride, (int)P->src_stride, M, H, hbd ? P->bit_depth : 8);
                oracle_wiener_decompose_sep_sym(P->wiener_win, M, H, vd, hd);
                oracle_wiener_finalize_sym_filter(P->wiener_win, vd, o->vfilter);
                oracle_wiener_finalize_sym_filter(P->wiener_win, hd, o->hfilter);
                if (oracle_wiener_compute_score(P->wiener_win, M, H, o->vfilter, o->hfilter) > 0) ok = 0;
            }
            if (!ok) {
                o->sse[1] = INT64_MAX; /* (rusi->wiener is not written in this case, :1344-1347; zeroed here) */
                memset(o->vfilter, 0, 16); memset(o->hfilter, 0, 16);
            } else {
                int t = 0;
                o->sse[1] = finer_search_wiener(P, &r, o->vfilter, o->hfilter, tmp, &t);
                if (trials) trials[u] = t;
            }
        }
        if (P->sg_enabled) {
            search_sgr(P, &r, flt0, flt1, &o->ep, o->xqd);
            o->sse[2] = try_sgr(P, &r, o->ep, o->xqd, tmp);
        }
    }
    free(tmp); free(flt0); free(flt1); free(M); free(H);
}
