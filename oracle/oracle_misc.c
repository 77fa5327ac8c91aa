/*
 * oracle_misc.c -- TEST INFRASTRUCTURE: plain-C restatement of SATD / Hadamard / residual and of the
 * loop-restoration search statistics (a24).  Pinned against oracle/_ref in tests/test_oracle_pin_misc.py.
 * Paths relative to /root/reference/Source/Lib.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* svt_aom_satd_c, Codec/common_dsp_rtcd.c:70-78 */
int oracle_satd(const int32_t *coeff, int length) {
    int s = 0;
    for (int i = 0; i < length; i++) s += abs(coeff[i]);
    return s;
}
/* svt_residual_kernel8bit_c / 16bit_c, Codec/pic_operators.c:125-160 */
void oracle_residual(const void *input, uint32_t in_stride, const void *pred, uint32_t pred_stride, int16_t *residual, uint32_t res_stride, uint32_t w,
                     uint32_t h, int is16) {
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++) {
            const int a = is16 ? ((const uint16_t *)input)[y * in_stride + x] : ((const uint8_t *)input)[y * in_stride + x];
            const int b = is16 ? ((const uint16_t *)pred)[y * pred_stride + x] : ((const uint8_t *)pred)[y * pred_stride + x];
            residual[y * res_stride + x] = (int16_t)(a - b);
        }
}
/* 1-D Walsh-Hadamard butterflies with the reference's int16 intermediates and output permutation
 * (C_DEFAULT/picture_operators_c.c:175-186 for 4 points, :214-241 for 8 points) */
static void had4(const int16_t *s, long stride, int16_t *c) {
    const int16_t b0 = (int16_t)((s[0] + s[stride]) >> 1), b1 = (int16_t)((s[0] - s[stride]) >> 1);
    const int16_t b2 = (int16_t)((s[2 * stride] + s[3 * stride]) >> 1), b3 = (int16_t)((s[2 * stride] - s[3 * stride]) >> 1);
    c[0] = (int16_t)(b0 + b2); c[1] = (int16_t)(b1 + b3); c[2] = (int16_t)(b0 - b2); c[3] = (int16_t)(b1 - b3);
}
static void had8(const int16_t *s, long stride, int16_t *c) {
    int16_t b[8], d[8];
    for (int i = 0; i < 4; i++) { b[2 * i] = (int16_t)(s[2 * i * stride] + s[(2 * i + 1) * stride]); b[2 * i + 1] = (int16_t)(s[2 * i * stride] - s[(2 * i + 1) * stride]); }
    d[0] = (int16_t)(b[0] + b[2]); d[1] = (int16_t)(b[1] + b[3]); d[2] = (int16_t)(b[0] - b[2]); d[3] = (int16_t)(b[1] - b[3]);
    d[4] = (int16_t)(b[4] + b[6]); d[5] = (int16_t)(b[5] + b[7]); d[6] = (int16_t)(b[4] - b[6]); d[7] = (int16_t)(b[5] - b[7]);
    c[0] = (int16_t)(d[0] + d[4]); c[7] = (int16_t)(d[1] + d[5]); c[3] = (int16_t)(d[2] + d[6]); c[4] = (int16_t)(d[3] + d[7]);
    c[2] = (int16_t)(d[0] - d[4]); c[6] = (int16_t)(d[1] - d[5]); c[1] = (int16_t)(d[2] - d[6]); c[5] = (int16_t)(d[3] - d[7]);
}
/* svt_aom_hadamard_{4x4,8x8,16x16,32x32}_c, picture_operators_c.c:188-326 */
void oracle_hadamard(const int16_t *src, long stride, int32_t *coeff, int n) {
    if (n == 4 || n == 8) {
        int16_t t[64], u[64];
        for (int i = 0; i < n; i++) (n == 4 ? had4 : had8)(src + i, stride, t + n * i);
        for (int i = 0; i < n; i++) (n == 4 ? had4 : had8)(t + i, n, u + n * i);
        for (int i = 0; i < n * n; i++) coeff[i] = u[i];
        return;
    }
    const int hn = n / 2, q = hn * hn, sh = n == 16 ? 1 : 2;
    for (int i = 0; i < 4; i++) oracle_hadamard(src + (i >> 1) * hn * stride + (i & 1) * hn, stride, coeff + i * q, hn);
    for (int i = 0; i < q; i++) {
        const int32_t a0 = coeff[i], a1 = coeff[q + i], a2 = coeff[2 * q + i], a3 = coeff[3 * q + i];
        const int32_t b0 = (a0 + a1) >> sh, b1 = (a0 - a1) >> sh, b2 = (a2 + a3) >> sh, b3 = (a2 - a3) >> sh;
        coeff[i] = b0 + b2; coeff[q + i] = b1 + b3; coeff[2 * q + i] = b0 - b2; coeff[3 * q + i] = b1 - b3;
    }
}
/* one transform block of hadamard_path_c (Codec/enc_mode_config.c:2147-2215): residual -> hadamard -> satd */
uint32_t oracle_hadamard_satd(const uint8_t *input, uint32_t in_stride, const uint8_t *pred, uint32_t pred_stride, int n) {
    int16_t res[32 * 32];
    int32_t co[32 * 32];
    oracle_residual(input, in_stride, pred, pred_stride, res, (uint32_t)n, (uint32_t)n, (uint32_t)n, 0);
    oracle_hadamard(res, n, co, n);
    return (uint32_t)oracle_satd(co, n * n);
}

/* ---- loop-restoration search statistics (Codec/restoration_pick.c) ------------------------------------------------ */
/* svt_av1_compute_stats_c (:659-700) / svt_av1_compute_stats_highbd_c (:701-745): M = sum y*x, H = sum y*y^T over the window
 * of wiener_win^2 taps (column-major tap order: k over columns outer, l over rows inner), samples centred on the unit's
 * average of the degraded picture; the highbd form divides by 4 / 16 for 10 / 12 bit. */
void oracle_compute_stats(int win, const void *dgd, const void *src, int h_start, int h_end, int v_start, int v_end, int dgd_stride, int src_stride,
                          int64_t *M, int64_t *H, int bit_depth) {
    const int is16 = bit_depth > 8, w2 = win * win, hw = win >> 1;
    uint64_t  sum = 0;
    for (int i = v_start; i < v_end; i++)
        for (int j = h_start; j < h_end; j++) sum += is16 ? ((const uint16_t *)dgd)[i * dgd_stride + j] : ((const uint8_t *)dgd)[i * dgd_stride + j];
    const int32_t avg = (int32_t)(sum / (uint64_t)((v_end - v_start) * (h_end - h_start)));
    memset(M, 0, sizeof(int64_t) * w2);
    memset(H, 0, sizeof(int64_t) * w2 * w2);
    int32_t y[49];
    for (int i = v_start; i < v_end; i++)
        for (int j = h_start; j < h_end; j++) {
            const int32_t x = (is16 ? ((const uint16_t *)src)[i * src_stride + j] : ((const uint8_t *)src)[i * src_stride + j]) - avg;
            int idx = 0;
            for (int k = -hw; k <= hw; k++)
                for (int l = -hw; l <= hw; l++)
                    y[idx++] = (is16 ? ((const uint16_t *)dgd)[(i + l) * dgd_stride + j + k] : ((const uint8_t *)dgd)[(i + l) * dgd_stride + j + k]) - avg;
            for (int k = 0; k < w2; k++) {
                M[k] += (int64_t)y[k] * x;
                for (int l = k; l < w2; l++) H[k * w2 + l] += (int64_t)y[k] * y[l];
            }
        }
    const int div = bit_depth == 12 ? 16 : (bit_depth == 10 ? 4 : 1);
    for (int k = 0; k < w2; k++) {
        if (is16) M[k] /= div;
        for (int l = k; l < w2; l++) {
            if (is16) H[k * w2 + l] /= div;
            H[l * w2 + k] = H[k * w2 + l];
        }
    }
}
/* svt_av1_lowbd_pixel_proj_error_c (:167-232) / svt_av1_highbd_pixel_proj_error_c (:234-303) */
int64_t oracle_pixel_proj_error(const void *src, int width, int height, int src_stride, const void *dat, int dat_stride, const int32_t *flt0,
                                int flt0_stride, const int32_t *flt1, int flt1_stride, const int32_t *xq, int r0, int r1, int is16) {
    int64_t err = 0;
    for (int i = 0; i < height; i++)
        for (int j = 0; j < width; j++) {
            const int32_t d = is16 ? ((const uint16_t *)dat)[i * dat_stride + j] : ((const uint8_t *)dat)[i * dat_stride + j];
            const int32_t s = is16 ? ((const uint16_t *)src)[i * src_stride + j] : ((const uint8_t *)src)[i * src_stride + j];
            int32_t       e;
            if (r0 > 0 || r1 > 0) {
                const int32_t u = d << 4;
                int32_t       v = u << 7;
                if (r0 > 0) v += xq[0] * (flt0[i * flt0_stride + j] - u);
                if (r1 > 0) v += xq[1] * (flt1[i * flt1_stride + j] - u);
                e = ((v + (1 << 10)) >> 11) - s;
            } else {
                e = d - s;
            }
            err += (int64_t)e * e; /* the lowbd C code multiplies in int32; |e| < 2^13 so nothing wraps */
        }
    return err;
}
/* svt_get_proj_subspace_c (:413-498): 2x2 normal equations of the self-guided projection, double precision.
 * Every summand is an integer below 2^53 in magnitude, so the double sums are exact whatever the order. */
void oracle_get_proj_subspace(const void *src, int width, int height, int src_stride, const void *dat, int dat_stride, int is16, const int32_t *flt0,
                              int flt0_stride, const int32_t *flt1, int flt1_stride, int32_t *xq, int r0, int r1) {
    double H[2][2] = {{0, 0}, {0, 0}}, Cc[2] = {0, 0};
    const int size = width * height;
    xq[0] = xq[1] = 0;
    for (int i = 0; i < height; i++)
        for (int j = 0; j < width; j++) {
            const double u  = (double)((is16 ? ((const uint16_t *)dat)[i * dat_stride + j] : ((const uint8_t *)dat)[i * dat_stride + j]) << 4);
            const double s  = (double)((is16 ? ((const uint16_t *)src)[i * src_stride + j] : ((const uint8_t *)src)[i * src_stride + j]) << 4) - u;
            const double f1 = r0 > 0 ? (double)flt0[i * flt0_stride + j] - u : 0;
            const double f2 = r1 > 0 ? (double)flt1[i * flt1_stride + j] - u : 0;
            H[0][0] += f1 * f1; H[1][1] += f2 * f2; H[0][1] += f1 * f2; Cc[0] += f1 * s; Cc[1] += f2 * s;
        }
    H[0][0] /= size; H[0][1] /= size; H[1][1] /= size; H[1][0] = H[0][1]; Cc[0] /= size; Cc[1] /= size;
    if (r0 == 0) {
        if (H[1][1] < 1e-8) return;
        xq[1] = (int32_t)rint(Cc[1] / H[1][1] * 128);
    } else if (r1 == 0) {
        if (H[0][0] < 1e-8) return;
        xq[0] = (int32_t)rint(Cc[0] / H[0][0] * 128);
    } else {
        const double det = H[0][0] * H[1][1] - H[0][1] * H[1][0];
        if (det < 1e-8) return;
        xq[0] = (int32_t)rint((H[1][1] * Cc[0] - H[0][1] * Cc[1]) / det * 128);
        xq[1] = (int32_t)rint((H[0][0] * Cc[1] - H[1][0] * Cc[0]) / det * 128);
    }
}
