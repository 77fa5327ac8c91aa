/*
 * ref_drivers2.c -- TEST INFRASTRUCTURE (bench.py's cpu_baseline legs only).  Time-bounded loops over the REAL reference kernels -- the AVX2 / AVX-512 variants in
 * oracle/_ref/libsvtref.so, passed in as function pointers by the caller -- one loop per leg of the bench that had no all-core CPU figure:
 * quantize_b / quantize_fp, the Wiener and self-guided restoration kernels over a plane, compute_stats, the HME SAD loop, the CDEF strength search, Hadamard.
 * Thread `idx0` of `step` works on its share of the items until `seconds` of wall time have passed; the return value is the number of items done.
 */
#define _POSIX_C_SOURCE 200809L
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }

/* svt_aom_quantize_b / svt_aom_highbd_quantize_b (aom_dsp_rtcd.h:246-251): blocks of n_coeffs coefficients, private output per thread */
typedef void (*QuantBFn)(const int32_t *, intptr_t, const int16_t *, const int16_t *, const int16_t *, const int16_t *, int32_t *, int32_t *, const int16_t *, uint16_t *,
                         const int16_t *, const int16_t *, const void *, const void *, int32_t);
uint64_t oracle_time_quantize_b(QuantBFn fn, const int32_t *coeff, uint32_t n_blocks, int n_coeffs, const int16_t *zbin, const int16_t *round, const int16_t *quant,
                                const int16_t *shift, const int16_t *dequant, const int16_t *scan, const int16_t *iscan, int log_scale, int32_t *out /* 2 * n_coeffs */,
                                uint32_t idx0, uint32_t step, double seconds) {
    uint64_t     done = 0;
    uint16_t     eob;
    const double t_end = now_s() + seconds;
    for (;;)
        for (uint32_t i = idx0; i < n_blocks; i += step) {
            fn(coeff + (size_t)i * n_coeffs, n_coeffs, zbin, round, quant, shift, out, out + n_coeffs, dequant, &eob, scan, iscan, NULL, NULL, log_scale);
            if ((++done & 63) == 0 && now_s() >= t_end) return done;
        }
}
/* svt_av1_highbd_quantize_fp (aom_dsp_rtcd.h:256) */
typedef void (*QuantFpFn)(const int32_t *, intptr_t, const int16_t *, const int16_t *, const int16_t *, const int16_t *, int32_t *, int32_t *, const int16_t *, uint16_t *,
                          const int16_t *, const int16_t *, int16_t);
uint64_t oracle_time_quantize_fp(QuantFpFn fn, const int32_t *coeff, uint32_t n_blocks, int n_coeffs, const int16_t *zbin, const int16_t *round, const int16_t *quant,
                                 const int16_t *shift, const int16_t *dequant, const int16_t *scan, const int16_t *iscan, int log_scale, int32_t *out, uint32_t idx0,
                                 uint32_t step, double seconds) {
    uint64_t     done = 0;
    uint16_t     eob;
    const double t_end = now_s() + seconds;
    for (;;)
        for (uint32_t i = idx0; i < n_blocks; i += step) {
            fn(coeff + (size_t)i * n_coeffs, n_coeffs, zbin, round, quant, shift, out, out + n_coeffs, dequant, &eob, scan, iscan, (int16_t)log_scale);
            if ((++done & 63) == 0 && now_s() >= t_end) return done;
        }
}

/* Loop restoration of a 16-bit plane in the processing units the reference filters (restoration.c: 64 columns x one 64-row stripe): kind 0 =
 * svt_av1_highbd_wiener_convolve_add_src (common_dsp_rtcd.h:1903), kind 1 = svt_apply_selfguided_restoration (:1906).  Pointers are CONVERT_TO_BYTEPTR'd as the
 * reference passes them.  The plane carries a border of `border` samples all round (the filters read 3 beyond a unit).  Returns 64 x 64 units filtered. */
typedef void (*WienerHbdFn)(const uint8_t *, ptrdiff_t, uint8_t *, ptrdiff_t, const int16_t *, const int16_t *, int32_t, int32_t, const void *, int32_t);
typedef void (*SgrApplyFn)(const uint8_t *, int32_t, int32_t, int32_t, int32_t, const int32_t *, uint8_t *, int32_t, int32_t *, int32_t, int32_t);
typedef struct { int32_t ref, do_average, dst_a, dst_b, dst_stride, round_0, round_1, plane, is_compound, use_jnt, fwd, bck, use_dist; } OracleConvParams; /* >= sizeof(ConvolveParams) */
uint64_t oracle_time_lr_plane(void *fn, int kind, const uint16_t *plane /* sample (0, 0) */, int stride, int w, int h, uint16_t *out, int out_stride, int bd,
                              const int16_t *fx, const int16_t *fy, int eps, const int32_t *xqd, uint32_t idx0, uint32_t step, double seconds) {
    const int    nx = (w + 63) / 64, ny = (h + 63) / 64, n = nx * ny;
    int32_t     *tmp = NULL;
    if (posix_memalign((void **)&tmp, 64, sizeof(int32_t) * 4 * 512 * 512)) return 0; /* RESTORATION_TMPBUF_SIZE is smaller than this */
    /* get_conv_params_wiener (convolve.h:70-88): round_0 = 3 (5 at 12 bit), round_1 = 11 (9 at 12 bit) */
    struct { int32_t ref, do_average; void *dst; int32_t dst_stride, round_0, round_1, plane, is_compound, use_jnt_comp_avg, fwd_offset, bck_offset, use_dist_wtd_comp_avg; } cp;
    memset(&cp, 0, sizeof(cp));
    cp.round_0 = bd == 12 ? 5 : 3; cp.round_1 = bd == 12 ? 9 : 11;
    uint64_t     done = 0;
    const double t_end = now_s() + seconds;
    for (;;)
        for (int u = (int)idx0; u < n; u += (int)step) {
            const int x0 = (u % nx) * 64, y0 = (u / nx) * 64, uw = w - x0 < 64 ? w - x0 : 64, uh = h - y0 < 64 ? h - y0 : 64;
            const uint16_t *s = plane + (size_t)y0 * stride + x0;
            uint16_t       *d = out + (size_t)y0 * out_stride + x0;
            if (kind == 0) ((WienerHbdFn)fn)((const uint8_t *)((uintptr_t)s >> 1), stride, (uint8_t *)((uintptr_t)d >> 1), out_stride, fx, fy, uw, uh, &cp, bd);
            else ((SgrApplyFn)fn)((const uint8_t *)((uintptr_t)s >> 1), uw, uh, stride, eps, xqd, (uint8_t *)((uintptr_t)d >> 1), out_stride, tmp, bd, 1);
            done++;
            if ((done & 7) == 0 && now_s() >= t_end) { free(tmp); return done; }
        }
}

/* svt_av1_compute_stats_highbd (aom_dsp_rtcd.h:1378): the Wiener statistics of restoration units of `us` x `us` samples.  Returns units done. */
typedef void (*StatsHbdFn)(int32_t, const uint8_t *, const uint8_t *, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, int64_t *, int64_t *, int32_t);
uint64_t oracle_time_compute_stats(StatsHbdFn fn, int win, const uint16_t *dgd, int dgd_stride, const uint16_t *src, int src_stride, int w, int h, int us, int bd,
                                   uint32_t idx0, uint32_t step, double seconds) {
    const int nx = (w + us / 2) / us > 0 ? (w + us / 2) / us : 1, ny = (h + us / 2) / us > 0 ? (h + us / 2) / us : 1, n = nx * ny;
    int64_t  *M = NULL;
    if (posix_memalign((void **)&M, 64, sizeof(int64_t) * (49 + 49 * 49 + 8))) return 0;
    int64_t *H = M + 56;
    uint64_t     done = 0;
    const double t_end = now_s() + seconds;
    for (;;)
        for (int u = (int)idx0; u < n; u += (int)step) {
            const int ux = u % nx, uy = u / nx;
            const int h0 = ux * us, h1 = ux == nx - 1 ? w : (ux + 1) * us, v0 = uy * us, v1 = uy == ny - 1 ? h : (uy + 1) * us;
            fn(win, (const uint8_t *)((uintptr_t)dgd >> 1), (const uint8_t *)((uintptr_t)src >> 1), h0, h1, v0, v1, dgd_stride, src_stride, M, H, bd);
            done++;
            if (now_s() >= t_end) { free(M); return done; }
        }
}

/* svt_sad_loop_kernel (aom_dsp_rtcd.h:2012): one HME-style search of a (bw x bh) block over an area of (aw x ah) positions per item */
typedef void (*SadLoopFn)(uint8_t *, uint32_t, uint8_t *, uint32_t, uint32_t, uint32_t, uint64_t *, int16_t *, int16_t *, uint32_t, uint8_t, int16_t, int16_t);
typedef struct { uint64_t src_off, ref_off; } OracleSadLoopItem;
uint64_t oracle_time_sad_loop(SadLoopFn fn, uint8_t *src_base, uint32_t src_stride, uint8_t *ref_base, uint32_t ref_stride, const OracleSadLoopItem *it, uint32_t n, int bw, int bh,
                              int aw, int ah, uint32_t idx0, uint32_t step, double seconds, uint64_t *checksum) {
    uint64_t     done = 0, sum = 0;
    const double t_end = now_s() + seconds;
    for (;;)
        for (uint32_t i = idx0; i < n; i += step) {
            uint64_t best = 0;
            int16_t  xc = 0, yc = 0;
            fn(src_base + it[i].src_off, src_stride, ref_base + it[i].ref_off, ref_stride, (uint32_t)bh, (uint32_t)bw, &best, &xc, &yc, src_stride, 0, (int16_t)aw, (int16_t)ah);
            sum += best + (uint64_t)(uint16_t)xc + (uint64_t)(uint16_t)yc;
            if ((++done & 15) == 0 && now_s() >= t_end) { *checksum = sum; return done; }
        }
}

/* The CDEF strength search of a 16-bit luma plane as cdef_seg_search does it (cdef_process.c:208-300): per 64 x 64 filter block the tile is staged once, then every
 * candidate (pri, sec) strength is filtered (svt_cdef_filter_fb, dispatching through the RTCD pointers the caller pointed at the AVX2 kernels) and its distortion
 * against the source taken (svt_aom_compute_cdef_dist_16bit_avx2).  Returns filter blocks x strengths evaluated. */
typedef void (*CdefFilterFbFn)(uint8_t *, uint16_t *, int32_t, uint16_t *, int32_t, int32_t, uint8_t (*)[16], int32_t *, int32_t (*)[16], int32_t, void *, int32_t,
                               int32_t, int32_t, int32_t, int32_t, int32_t, uint8_t);
typedef uint64_t (*CdefDistFn)(const uint16_t *, int32_t, const uint16_t *, const void *, int32_t, int /* BlockSize */, int32_t, int32_t, uint8_t);
uint64_t oracle_time_cdef_search(CdefFilterFbFn fb, CdefDistFn dist, const uint16_t *recon, const uint16_t *source, int stride, int w, int h, const int32_t *pri, const int32_t *sec,
                                 int ncand, int damping, int coeff_shift, uint32_t idx0, uint32_t step, double seconds, uint64_t *checksum) {
    const int    nhfb = (w + 63) / 64, nvfb = (h + 63) / 64, nfb = nhfb * nvfb;
    uint16_t    *tile = NULL, *tmp = NULL;
    if (posix_memalign((void **)&tile, 64, sizeof(uint16_t) * 70 * 144 + 64) || posix_memalign((void **)&tmp, 64, sizeof(uint16_t) * 64 * 64)) return 0;
    uint64_t     done = 0, sum = 0;
    const double t_end = now_s() + seconds;
    for (;;)
        for (int f = (int)idx0; f < nfb; f += (int)step) {
            const int fbr = f / nhfb, fbc = f % nhfb;
            for (int i = 0; i < 70 * 144; i++) tile[i] = 0x7f7f;
            const int x0 = fbc * 64, y0 = fbr * 64;
            const int xs = x0 - (fbc ? 8 : 0), ys = y0 - (fbr ? 3 : 0);
            const int xe = (x0 + 64 < w ? x0 + 64 : w) + (fbc + 1 < nhfb ? 8 : 0), ye = (y0 + 64 < h ? y0 + 64 : h) + (fbr + 1 < nvfb ? 3 : 0);
            uint16_t *in = tile + 3 * 144 + 8;
            for (int y = ys; y < ye; y++) memcpy(in + (y - y0) * 144 + (xs - x0), recon + (size_t)y * stride + xs, sizeof(uint16_t) * (xe - xs));
            uint8_t dl[128], dir[16][16];
            int32_t var[16][16], dirinit = 0, cnt = 0;
            for (int by = 0; by < 8 && y0 + by * 8 < h; by++)
                for (int bx = 0; bx < 8 && x0 + bx * 8 < w; bx++) { dl[2 * cnt] = (uint8_t)by; dl[2 * cnt + 1] = (uint8_t)bx; cnt++; }
            for (int c = 0; c < ncand; c++) {
                fb(NULL, tmp, 0, in, 0, 0, dir, &dirinit, var, 0, dl, cnt, pri[c], sec[c], damping, damping, coeff_shift, 1); /* (dst8 NULL, dstride 0: packed 8x8 blocks, cdef.c) */
                sum += dist(source + (size_t)y0 * stride + x0, stride, tmp, dl, cnt, 3 /* BLOCK_8X8 */, coeff_shift, 0, 1);
                done++;
            }
            if (now_s() >= t_end) { free(tile); free(tmp); *checksum = sum; return done; }
        }
}

/* svt_aom_hadamard_WxW (common_dsp_rtcd.h:1072-1078) over blocks of a 16-bit residual; returns blocks done */
typedef void (*HadamardFn)(const int16_t *, ptrdiff_t, int32_t *);
uint64_t oracle_time_hadamard(HadamardFn fn, const int16_t *res, uint32_t n, int w, int32_t *out /* w * w */, uint32_t idx0, uint32_t step, double seconds) {
    uint64_t     done = 0;
    const double t_end = now_s() + seconds;
    for (;;)
        for (uint32_t i = idx0; i < n; i += step) {
            fn(res + (size_t)i * w * w, w, out);
            if ((++done & 63) == 0 && now_s() >= t_end) return done;
        }
}
