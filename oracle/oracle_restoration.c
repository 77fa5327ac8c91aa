/*
 * oracle_restoration.c -- TEST INFRASTRUCTURE: plain-C restatement of the reference's loop-restoration filters
 * (Wiener, self-guided) and of the stripe / restoration-unit driver.  Pinned against oracle/_ref in
 * tests/test_oracle_pin_restoration.py.  Paths relative to /root/reference/Source/Lib/Codec.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define FILTER_BITS 7
#define RPOT(v, n) (((v) + ((1 << (n)) >> 1)) >> (n)) /* ROUND_POWER_OF_TWO */

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* get_conv_params_wiener, convolve.h:70-88 */
static void wiener_rounds(int bd, int *r0, int *r1) {
    *r0 = 3;
    *r1 = 2 * FILTER_BITS - 3;
    const int range = bd + FILTER_BITS - *r0 + 2;
    if (range > 16) { *r0 += range - 16; *r1 -= range - 16; }
}

/* svt_av1_wiener_convolve_add_src_c (convolve.c:100-147) and svt_av1_highbd_wiener_convolve_add_src_c (:194-250):
 * separable 7-tap filter (8-entry kernels, last tap 0) with "add src"; the horizontal pass output is clamped to
 * WIENER_CLAMP_LIMIT.  `pix(y, x)` abstracts the source so the frame driver can substitute boundary rows. */
typedef int (*PixFn)(void *ctx, int y, int x);
static int wiener_px(PixFn pix, void *ctx, int y, int x, const int16_t *fx, const int16_t *fy, int bd) {
    int r0, r1;
    wiener_rounds(bd, &r0, &r1);
    const int lim = (1 << (bd + 1 + FILTER_BITS - r0)) - 1;
    int       col[7];
    for (int k = 0; k < 7; k++) { /* rows y-3 .. y+3 */
        int sum = (pix(ctx, y - 3 + k, x) << FILTER_BITS) + (1 << (bd + FILTER_BITS - 1));
        for (int t = 0; t < 8; t++) sum += pix(ctx, y - 3 + k, x - 3 + t) * fx[t];
        col[k] = clampi(RPOT(sum, r0), 0, lim);
    }
    int sum = (col[3] << FILTER_BITS) - (1 << (bd + r1 - 1));
    for (int t = 0; t < 7; t++) sum += col[t] * fy[t];
    /* tap 7 multiplies the row y+4 of the intermediate buffer; the reference's kernels always carry fy[7] == 0 and the
       buffer row it would read is zeroed (convolve.c:118), so it contributes nothing */
    return clampi(RPOT(sum, r1), 0, (1 << bd) - 1);
}
/* note: the horizontal kernel also has 8 taps; tap 7 reads x+4.  The reference multiplies it by fx[7] (== 0). */

typedef struct { const uint8_t *p8; const uint16_t *p16; int stride; } PlaneCtx;
static int plane_pix(void *c, int y, int x) {
    const PlaneCtx *p = (const PlaneCtx *)c;
    return p->p16 ? p->p16[y * p->stride + x] : p->p8[y * p->stride + x];
}

void oracle_wiener_convolve_add_src(const void *src, int src_stride, void *dst, int dst_stride, const int16_t *fx, const int16_t *fy, int w,
                                    int h, int bd, int highbd) {
    PlaneCtx c = {highbd ? NULL : (const uint8_t *)src, highbd ? (const uint16_t *)src : NULL, src_stride};
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const int v = wiener_px(plane_pix, &c, y, x, fx, fy, bd);
            if (highbd) ((uint16_t *)dst)[y * dst_stride + x] = (uint16_t)v;
            else ((uint8_t *)dst)[y * dst_stride + x] = (uint8_t)v;
        }
}

/* ---- self-guided filter (restoration.c:468-992) ------------------------------------------------------------------ */
static const int16_t SGR_R[16][2] = {{2, 1}, {2, 1}, {2, 1}, {2, 1}, {2, 1}, {2, 1}, {2, 1}, {2, 1}, {2, 1}, {2, 1}, {0, 1}, {0, 1}, {0, 1}, {0, 1}, {2, 0}, {2, 0}};
static const int16_t SGR_S[16][2] = {{140, 3236}, {112, 2158}, {93, 1618}, {80, 1438}, {70, 1295}, {58, 1177}, {47, 1079}, {37, 996},
                                     {30, 925},   {25, 863},   {-1, 2589}, {-1, 1618}, {-1, 1177}, {-1, 925},  {56, -1},   {22, -1}};
int oracle_sgr_r(int idx, int k) { return SGR_R[idx][k]; }
int oracle_sgr_s(int idx, int k) { return SGR_S[idx][k]; }
/* svt_aom_eb_x_by_xplus1 (restoration.c:647-662): round(256 z / (z+1)), with [0] = 1 and [255] = 256 */
int oracle_x_by_xplus1(int z) { return z == 0 ? 1 : (z >= 255 ? 256 : (256 * z + (z + 1) / 2) / (z + 1)); }
/* svt_aom_eb_one_by_x (restoration.c:664-667): round(4096 / n) */
int oracle_one_by_x(int n) { return (4096 + n / 2) / n; }

/* A/B of calculate_intermediate_result at position (i, j) relative to the unit (restoration.c:705-764) */
static void sgr_ab(PixFn pix, void *ctx, int i, int j, int r, int s, int bd, int32_t *A, int32_t *B) {
    uint32_t sum = 0, sq = 0;
    for (int dy = -r; dy <= r; dy++)
        for (int dx = -r; dx <= r; dx++) {
            const uint32_t v = (uint32_t)pix(ctx, i + dy, j + dx);
            sum += v;
            sq += v * v;
        }
    const uint32_t n = (uint32_t)((2 * r + 1) * (2 * r + 1));
    const uint32_t a = RPOT(sq, 2 * (bd - 8)), b = RPOT(sum, bd - 8);
    const uint32_t p = (a * n < b * b) ? 0 : a * n - b * b;
    const uint32_t z = RPOT(p * (uint32_t)s, 20);
    const int32_t  av = oracle_x_by_xplus1(z > 255 ? 255 : (int)z);
    *A = av;
    *B = (int32_t)RPOT((uint32_t)(256 - av) * sum * (uint32_t)oracle_one_by_x((int)n), 12);
}
/* one output sample of flt0 (r == 2, "fast": A/B exist on odd rows only) or flt1 (r == 1) */
static int32_t sgr_flt(PixFn pix, void *ctx, int i, int j, int pass, int idx, int bd) {
    const int r = SGR_R[idx][pass], s = SGR_S[idx][pass];
    int32_t   a = 0, b = 0, A, B, nb;
    if (pass == 0) {
        if (!(i & 1)) {
            nb = 5;
            for (int dy = -1; dy <= 1; dy += 2)
                for (int dx = -1; dx <= 1; dx++) {
                    sgr_ab(pix, ctx, i + dy, j + dx, r, s, bd, &A, &B);
                    a += A * (dx ? 5 : 6);
                    b += B * (dx ? 5 : 6);
                }
        } else {
            nb = 4;
            for (int dx = -1; dx <= 1; dx++) {
                sgr_ab(pix, ctx, i, j + dx, r, s, bd, &A, &B);
                a += A * (dx ? 5 : 6);
                b += B * (dx ? 5 : 6);
            }
        }
    } else {
        nb = 5;
        for (int dy = -1; dy <= 1; dy++)
            for (int dx = -1; dx <= 1; dx++) {
                sgr_ab(pix, ctx, i + dy, j + dx, r, s, bd, &A, &B);
                a += A * ((dx && dy) ? 3 : 4);
                b += B * ((dx && dy) ? 3 : 4);
            }
    }
    const int32_t v = a * pix(ctx, i, j) + b;
    return RPOT(v, 8 + nb - 4);
}
/* svt_av1_selfguided_restoration_c (restoration.c:923-955) */
void oracle_selfguided_restoration(const void *dgd, int width, int height, int stride, int32_t *flt0, int32_t *flt1, int flt_stride, int idx,
                                   int bd, int highbd) {
    PlaneCtx c = {highbd ? NULL : (const uint8_t *)dgd, highbd ? (const uint16_t *)dgd : NULL, stride};
    for (int i = 0; i < height; i++)
        for (int j = 0; j < width; j++) {
            if (SGR_R[idx][0] > 0) flt0[i * flt_stride + j] = sgr_flt(plane_pix, &c, i, j, 0, idx, bd);
            if (SGR_R[idx][1] > 0) flt1[i * flt_stride + j] = sgr_flt(plane_pix, &c, i, j, 1, idx, bd);
        }
}
static int sgr_apply_px(PixFn pix, void *ctx, int i, int j, int idx, const int32_t *xqd, int bd) {
    int xq0, xq1; /* svt_decode_xq, restoration.c:634-645 */
    if (SGR_R[idx][0] == 0) { xq0 = 0; xq1 = 128 - xqd[1]; }
    else if (SGR_R[idx][1] == 0) { xq0 = xqd[0]; xq1 = 0; }
    else { xq0 = xqd[0]; xq1 = 128 - xq0 - xqd[1]; }
    const int32_t u = pix(ctx, i, j) << 4;
    int32_t       v = u << 7;
    if (SGR_R[idx][0] > 0) v += xq0 * (sgr_flt(pix, ctx, i, j, 0, idx, bd) - u);
    if (SGR_R[idx][1] > 0) v += xq1 * (sgr_flt(pix, ctx, i, j, 1, idx, bd) - u);
    const int16_t w = (int16_t)RPOT(v, 11);
    return clampi(w, 0, (1 << bd) - 1);
}
/* svt_apply_selfguided_restoration_c (restoration.c:957-992) */
void oracle_apply_selfguided_restoration(const void *dat, int width, int height, int stride, int eps, const int32_t *xqd, void *dst,
                                         int dst_stride, int bd, int highbd) {
    PlaneCtx c = {highbd ? NULL : (const uint8_t *)dat, highbd ? (const uint16_t *)dat : NULL, stride};
    for (int i = 0; i < height; i++)
        for (int j = 0; j < width; j++) {
            const int v = sgr_apply_px(plane_pix, &c, i, j, eps, xqd, bd);
            if (highbd) ((uint16_t *)dst)[i * dst_stride + j] = (uint16_t)v;
            else ((uint8_t *)dst)[i * dst_stride + j] = (uint8_t)v;
        }
}

/* ---- frame driver: svt_av1_loop_restoration_filter_frame for one plane (restoration.c:1067-1230) ------------------
 * Restoration units of `unit_size` (last one absorbs a remainder < 3/2 unit, and units are shifted up by 8 >> ss_y rows),
 * processed in stripes of 64 >> ss_y rows offset by 8 >> ss_y; the 3 rows above / below a stripe come from the saved
 * deblocked boundary lines (rows max(i+2,0) resp. min(i,1) of the stripe's two saved lines, :303-332) except at the frame
 * top / bottom where the frame is edge-extended (svt_extend_frame); columns are edge-extended. */
typedef struct {
    const void *data, *above, *below; /* plane and saved lines, both `stride` pixels per row; above/below: 2 rows per stripe */
    int         stride, bstride, w, h, highbd, stripe_top, stripe_bot, stripe_idx;
} StripeCtx;
static int rd(const void *p, int highbd, int off) { return highbd ? ((const uint16_t *)p)[off] : ((const uint8_t *)p)[off]; }
static int stripe_pix(void *c, int y, int x) {
    const StripeCtx *s = (const StripeCtx *)c;
    x = clampi(x, 0, s->w - 1);
    if (y < s->stripe_top) {
        if (s->stripe_top == 0) return rd(s->data, s->highbd, clampi(y, 0, s->h - 1) * s->stride + x);
        const int i = y - s->stripe_top; /* -3..-1 */
        return rd(s->above, s->highbd, (2 * s->stripe_idx + (i + 2 > 0 ? i + 2 : 0)) * s->bstride + x);
    }
    if (y >= s->stripe_bot) {
        if (s->stripe_bot >= s->h) return rd(s->data, s->highbd, clampi(y, 0, s->h - 1) * s->stride + x);
        const int i = y - s->stripe_bot; /* 0..2 */
        return rd(s->below, s->highbd, (2 * s->stripe_idx + (i < 1 ? i : 1)) * s->bstride + x);
    }
    return rd(s->data, s->highbd, y * s->stride + x);
}
typedef struct {
    int32_t rtype; /* 0 none, 1 wiener, 2 sgrproj (RestorationType) */
    int16_t vfilter[8], hfilter[8];
    int32_t ep, xqd[2];
} OracleRestUnit;
void oracle_lr_filter_frame(const void *data, int stride, const void *above, const void *below, int bstride, void *dst, int dst_stride, int w,
                            int h, int ss_y, int unit_size, const OracleRestUnit *units, int bd, int highbd) {
    const int off = 8 >> ss_y, sh = 64 >> ss_y;
    const int nvu = (h + (unit_size >> 1)) / unit_size > 0 ? (h + (unit_size >> 1)) / unit_size : 1;
    const int nhu = (w + (unit_size >> 1)) / unit_size > 0 ? (w + (unit_size >> 1)) / unit_size : 1;
    for (int y = 0; y < h; y++) {
        StripeCtx c = {data, above, below, stride, bstride, w, h, highbd, 0, 0, 0};
        c.stripe_idx = (y + off) / sh;
        c.stripe_top = c.stripe_idx * sh - off < 0 ? 0 : c.stripe_idx * sh - off;
        c.stripe_bot = (c.stripe_idx + 1) * sh - off > h ? h : (c.stripe_idx + 1) * sh - off;
        int ur = (y + off) / unit_size;
        if (ur >= nvu) ur = nvu - 1;
        for (int x = 0; x < w; x++) {
            int uc = x / unit_size;
            if (uc >= nhu) uc = nhu - 1;
            const OracleRestUnit *u = &units[ur * nhu + uc];
            int v;
            if (u->rtype == 1) v = wiener_px(stripe_pix, &c, y, x, u->hfilter, u->vfilter, bd);
            else if (u->rtype == 2) v = sgr_apply_px(stripe_pix, &c, y, x, u->ep, u->xqd, bd);
            else v = rd(data, highbd, y * stride + x);
            if (highbd) ((uint16_t *)dst)[y * dst_stride + x] = (uint16_t)v;
            else ((uint8_t *)dst)[y * dst_stride + x] = (uint8_t)v;
        }
    }
}
