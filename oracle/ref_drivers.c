/*
 * ref_drivers.c -- TEST INFRASTRUCTURE.  Drives the REAL reference kernels (function pointers resolved from
 * oracle/_ref/libsvtref.so by the caller) in the same order the reference's own static driver functions do, so that
 * the oracle restatement can be pinned against "the reference run here" for paths whose driver is `static` in the
 * reference and therefore not callable by symbol.
 */
#define _POSIX_C_SOURCE 200809L
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef void (*ExtAllFn)(uint8_t *, uint32_t, uint8_t *, uint32_t, uint32_t, uint32_t *, uint32_t *, uint32_t *, uint32_t *, uint32_t[16][8],
                         uint32_t[64][8], bool);
typedef void (*ExtEightFn)(uint32_t[16][8], uint32_t *, uint32_t *, uint32_t *, uint32_t *, uint32_t, uint32_t[4][8]);
typedef void (*ExtOne816Fn)(uint8_t *, uint32_t, uint8_t *, uint32_t, uint32_t *, uint32_t *, uint32_t *, uint32_t *, uint32_t, uint32_t *,
                            uint32_t *, bool);
typedef void (*ExtOne3264Fn)(uint32_t *, uint32_t *, uint32_t *, uint32_t *, uint32_t *, uint32_t, uint32_t *);

/* Sequence of Codec/motion_estimation.c:781-816 (open_loop_me_fullpel_search_sblock): groups of 8 x-positions through
 * :429-474, remainder columns one at a time through :476-779 (16 calls in the 16x16 order 0,1,4,5,2,3,6,7,8,9,12,13,10,
 * 11,14,15 then the 32x32/64x64 call).  Result layout as p_sb_best_sad/mv[85] (me_context.h). */
void oracle_drive_ref_me_search(ExtAllFn all, ExtEightFn eight, ExtOne816Fn one816, ExtOne3264Fn one3264, uint8_t *src,
                                uint32_t src_stride, uint8_t *ref, uint32_t ref_stride, int x_origin, int y_origin, int width, int height,
                                int sub_sad, uint32_t *best_sad, uint32_t *best_mv) {
    uint32_t        eight16[16][8], eight8[64][8], eight32[4][8]; /* stack: the driver is called from several threads */
    uint32_t        sad16[16], sad8[64], sad32[4];
    for (int i = 0; i < 85; i++) { best_sad[i] = 128 * 128 * 255; best_mv[i] = 0; }
    uint32_t *b64 = best_sad, *b32 = best_sad + 1, *b16 = best_sad + 5, *b8 = best_sad + 21;
    uint32_t *m64 = best_mv, *m32 = best_mv + 1, *m16 = best_mv + 5, *m8 = best_mv + 21;
    static const int order[16] = {0, 1, 4, 5, 2, 3, 6, 7, 8, 9, 12, 13, 10, 11, 14, 15};
    const int w8 = width & ~7;
    for (int y = 0; y < height; y++) {
        for (int x = 0; x < w8; x += 8) {
            const uint32_t mv = ((uint32_t)(y + y_origin) << 16) | (uint16_t)(x + x_origin);
            all(src, src_stride, ref + y * ref_stride + x, ref_stride, mv, b8, b16, m8, m16, eight16, eight8, sub_sad != 0);
            eight(eight16, b32, b64, m32, m64, mv, eight32);
        }
        for (int x = w8; x < width; x++) {
            const uint32_t mv = ((uint32_t)(y + y_origin) << 16) | (uint16_t)(x + x_origin);
            for (int by = 0; by < 4; by++)
                for (int bx = 0; bx < 4; bx++) {
                    const int i = order[by * 4 + bx];
                    one816(src + by * 16 * src_stride + bx * 16, src_stride, ref + (y + by * 16) * ref_stride + x + bx * 16, ref_stride,
                           &b8[4 * i], &b16[i], &m8[4 * i], &m16[i], mv, &sad16[i], &sad8[4 * i], sub_sad != 0);
                }
            one3264(sad16, b32, b64, m32, m64, mv, sad32);
        }
    }
}

/* Many (SB, ref) items in one call (bench.py cpu_baseline: no Python in the timed loop).  Items idx0, idx0+step, ...
 * of the n descriptors are searched `repeat` times; returns the number of SB-ref searches done. */
typedef struct {
    uint64_t src_off, ref_off;
    uint32_t src_stride, ref_stride;
    int16_t  x_origin, y_origin;
    uint16_t width, height;
} OracleMeDesc; /* same layout as SvtHipMeSearchDesc */
uint64_t oracle_drive_ref_me_search_many(ExtAllFn all, ExtEightFn eight, ExtOne816Fn one816, ExtOne3264Fn one3264, uint8_t *src_base,
                                         uint8_t *ref_base, const OracleMeDesc *d, uint32_t n, uint32_t idx0, uint32_t step, uint32_t repeat,
                                         int sub_sad, uint32_t *best_sad, uint32_t *best_mv) {
    uint64_t done = 0;
    for (uint32_t r = 0; r < repeat; r++)
        for (uint32_t i = idx0; i < n; i += step) {
            oracle_drive_ref_me_search(all, eight, one816, one3264, src_base + d[i].src_off, d[i].src_stride, ref_base + d[i].ref_off,
                                       d[i].ref_stride, d[i].x_origin, d[i].y_origin, d[i].width, d[i].height, sub_sad,
                                       best_sad + (size_t)i * 85, best_mv + (size_t)i * 85);
            done++;
        }
    return done;
}

/* Time-bounded form: thread `idx0` of `step` keeps searching its share of the list until `seconds` of wall time have
 * passed (checked every item), so the cpu_baseline leg of bench.py has a hard duration whatever the host gives us. */
#include <time.h>
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }
uint64_t oracle_drive_ref_me_search_timed(ExtAllFn all, ExtEightFn eight, ExtOne816Fn one816, ExtOne3264Fn one3264, uint8_t *src_base,
                                          uint8_t *ref_base, const OracleMeDesc *d, uint32_t n, uint32_t idx0, uint32_t step, double seconds,
                                          int sub_sad) {
    uint32_t     bs[85], bm[85];
    uint64_t     done = 0;
    const double t_end = now_s() + seconds;
    for (;;)
        for (uint32_t i = idx0; i < n; i += step) {
            oracle_drive_ref_me_search(all, eight, one816, one3264, src_base + d[i].src_off, d[i].src_stride, ref_base + d[i].ref_off,
                                       d[i].ref_stride, d[i].x_origin, d[i].y_origin, d[i].width, d[i].height, sub_sad, bs, bm);
            done++;
            if (now_s() >= t_end) return done;
        }
}

/* ---- timed loops over other reference kernels (bench.py cpu_baseline legs) ------------------------------------------ */
typedef void (*FwdTxfmFn)(int16_t *, int32_t *, uint32_t, uint8_t, uint8_t);
/* blocks idx0, idx0+step, ... of `n` contiguous w*h residual blocks, until `seconds` elapse; returns blocks transformed */
uint64_t oracle_time_fwd_txfm(FwdTxfmFn fn, int16_t *in, uint32_t n, int w, int h, int32_t *out_scratch /* step * w*h */, uint8_t tx_type, uint8_t bd,
                              uint32_t idx0, uint32_t step, double seconds) {
    uint64_t     done = 0;
    const double t_end = now_s() + seconds;
    int32_t     *out = out_scratch + (size_t)idx0 * w * h;
    for (;;)
        for (uint32_t i = idx0; i < n; i += step) {
            fn(in + (size_t)i * w * h, out, (uint32_t)w, tx_type, bd);
            if ((++done & 63) == 0 && now_s() >= t_end) return done;
        }
}
typedef void (*CdefFilterFbFn)(uint8_t *, uint16_t *, int32_t, uint16_t *, int32_t, int32_t, uint8_t (*)[16], int32_t *, int32_t (*)[16], int32_t, void *, int32_t,
                               int32_t, int32_t, int32_t, int32_t, int32_t, uint8_t);
/* CDEF apply of a 16-bit luma plane, filter blocks idx0, idx0+step, ...: tile construction as cdef_seg_search
 * (cdef_process.c:208-228) + the reference's svt_cdef_filter_fb (which dispatches through the RTCD pointers the caller has
 * pointed at the AVX2 kernels).  Returns the number of 64x64 filter blocks processed. */
uint64_t oracle_time_cdef_apply(CdefFilterFbFn fb, const uint16_t *plane, int stride, int w, int h, uint16_t *out, int level, int sec, int damping,
                                int coeff_shift, uint32_t idx0, uint32_t step, double seconds) {
    const int    nhfb = (w + 63) / 64, nvfb = (h + 63) / 64, nfb = nhfb * nvfb;
    uint16_t    *tile = NULL; /* the AVX2 kernels use aligned loads */
    if (posix_memalign((void **)&tile, 64, sizeof(uint16_t) * 70 * 144 + 64)) return 0;
    uint64_t     done = 0;
    const double t_end = now_s() + seconds;
    for (;;)
        for (int f = (int)idx0; f < nfb; f += (int)step) {
            const int fbr = f / nhfb, fbc = f % nhfb;
            for (int i = 0; i < 70 * 144; i++) tile[i] = 0x7f7f;
            const int x0 = fbc * 64, y0 = fbr * 64;
            const int xs = x0 - (fbc ? 8 : 0), ys = y0 - (fbr ? 3 : 0);
            const int xe = (x0 + 64 < w ? x0 + 64 : w) + (fbc + 1 < nhfb ? 8 : 0), ye = (y0 + 64 < h ? y0 + 64 : h) + (fbr + 1 < nvfb ? 3 : 0);
            uint16_t *in = tile + 3 * 144 + 8;
            for (int y = ys; y < ye; y++) memcpy(in + (y - y0) * 144 + (xs - x0), plane + (size_t)y * stride + xs, sizeof(uint16_t) * (xe - xs));
            uint8_t dl[128], dir[16][16];
            int32_t var[16][16], dirinit = 0, cnt = 0;
            for (int by = 0; by < 8 && y0 + by * 8 < h; by++)
                for (int bx = 0; bx < 8 && x0 + bx * 8 < w; bx++) { dl[2 * cnt] = (uint8_t)by; dl[2 * cnt + 1] = (uint8_t)bx; cnt++; }
            fb(NULL, out + (size_t)y0 * stride + x0, stride, in, 0, 0, dir, &dirinit, var, 0, dl, cnt, level, sec, damping, damping, coeff_shift, 1);
            done++;
            if (now_s() >= t_end) { free(tile); return done; }
        }
}

/* 64x64 SAD of (src, ref) plane pairs through a reference kernel, for bench.py's `sad64x64_pairs` cpu_baseline leg: pairs idx0, idx0 + step, ... of the
 * descriptor list (byte offsets + strides, = SvtHipSadPair) until `seconds` elapse.  kind 0: fn(src, stride, ref, stride) -- svt_aom_sad64x64_avx2
 * (aom_dsp_rtcd.h:335); kind 1: fn(src, stride, ref, stride, h, w) -- svt_nxm_sad_kernel_helper_avx2 (the svt_nxm_sad_kernel variant). */
typedef struct { uint64_t src_off, ref_off; uint32_t src_stride, ref_stride; } OracleSadPair;
typedef uint32_t (*Sad4Fn)(const uint8_t *, int, const uint8_t *, int);
typedef uint32_t (*Sad6Fn)(const uint8_t *, uint32_t, const uint8_t *, uint32_t, uint32_t, uint32_t);
uint64_t oracle_time_sad_pairs(void *fn, int kind, const uint8_t *src_base, const uint8_t *ref_base, const OracleSadPair *d, uint32_t n, uint32_t idx0, uint32_t step,
                               double seconds, uint64_t *checksum) {
    uint64_t     done = 0, sum = 0;
    const double t_end = now_s() + seconds;
    for (;;)
        for (uint32_t i = idx0; i < n; i += step) {
            sum += kind ? ((Sad6Fn)fn)(src_base + d[i].src_off, d[i].src_stride, ref_base + d[i].ref_off, d[i].ref_stride, 64, 64)
                        : ((Sad4Fn)fn)(src_base + d[i].src_off, (int)d[i].src_stride, ref_base + d[i].ref_off, (int)d[i].ref_stride);
            if ((++done & 255) == 0 && now_s() >= t_end) { *checksum = sum; return done; }
        }
}

/* inverse transform + reconstruction through a reference kernel (svt_av1_inv_txfm2d_add_WxH_*: common_dsp_rtcd.h), for bench.py's cpu_baseline legs: blocks idx0,
 * idx0 + step, ... of `n` contiguous w*h coefficient blocks, reconstructed onto one private w*h tile (read = write, as the encoder calls it) */
typedef void (*InvTxfmFn)(const int32_t *, uint16_t *, int32_t, uint16_t *, int32_t, uint8_t, int32_t);
uint64_t oracle_time_inv_txfm(InvTxfmFn fn, const int32_t *coeff, uint32_t n, int w, int h, uint16_t *tile, uint8_t tx_type, uint8_t bd, uint32_t idx0, uint32_t step,
                              double seconds) {
    uint64_t     done = 0;
    const double t_end = now_s() + seconds;
    (void)idx0;
    for (;;)
        for (uint32_t i = 0; i < n; i += step) {
            if ((done & 15) == 0) memset(tile, 0, sizeof(uint16_t) * w * h); /* (keeps the accumulating reconstruction away from the clamp) */
            fn(coeff + (size_t)i * w * h, tile, w, tile, w, tx_type, bd);
            if ((++done & 63) == 0 && now_s() >= t_end) return done;
        }
}
