// misc.hip -- SATD / Hadamard / residual and loop-restoration search statistics (a24) for gfx950.
//
//  * hadamard_kernel: one 64-lane workgroup per transform block (<= 32x32).  Each of the first N^2/64 lanes transforms one
//    8x8 sub-block in registers with the reference's int16 intermediates (picture_operators_c.c:214-241), the 16x16 / 32x32
//    combination stages (>>1, >>2) run out of LDS; SATD is a wave reduction of |coeff|.
//  * stats_kernel: Wiener normal equations M = sum y x, H = sum y y^T (restoration_pick.c:659-745) for one restoration
//    unit per workgroup; the unit is walked in 32x32 tiles staged in LDS, every lane owns a fixed set of (k, l) tap pairs and
//    accumulates in int64 (v_mad_i64_i32).
//  * proj_kernel: self-guided projection error / 2x2 normal equations (restoration_pick.c:167-303, :413-498); sums are
//    integer-exact, the final 2x2 solve is evaluated in IEEE double as the reference does.
#include "svt_hip_common.h"
#include "../../include/svtav1_hip.h"

namespace {

__device__ __forceinline__ long long wave_sum_i64(long long v) {
    for (int m = 32; m >= 1; m >>= 1) {
        const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)v, m), hi = (unsigned)__shfl_xor((int)(unsigned)((unsigned long long)v >> 32), m);
        v += (long long)(((unsigned long long)hi << 32) | lo);
    }
    return v;
}

// ---- Hadamard -------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void had8_col(const int16_t* s, const int stride, int16_t* c) { // hadamard_col8, picture_operators_c.c:214-241
    int16_t b[8], d[8];
#pragma unroll
    for (int i = 0; i < 4; i++) { b[2 * i] = (int16_t)(s[2 * i * stride] + s[(2 * i + 1) * stride]); b[2 * i + 1] = (int16_t)(s[2 * i * stride] - s[(2 * i + 1) * stride]); }
    d[0] = (int16_t)(b[0] + b[2]); d[1] = (int16_t)(b[1] + b[3]); d[2] = (int16_t)(b[0] - b[2]); d[3] = (int16_t)(b[1] - b[3]);
    d[4] = (int16_t)(b[4] + b[6]); d[5] = (int16_t)(b[5] + b[7]); d[6] = (int16_t)(b[4] - b[6]); d[7] = (int16_t)(b[5] - b[7]);
    c[0] = (int16_t)(d[0] + d[4]); c[7] = (int16_t)(d[1] + d[5]); c[3] = (int16_t)(d[2] + d[6]); c[4] = (int16_t)(d[3] + d[7]);
    c[2] = (int16_t)(d[0] - d[4]); c[6] = (int16_t)(d[1] - d[5]); c[1] = (int16_t)(d[2] - d[6]); c[5] = (int16_t)(d[3] - d[7]);
}
__device__ __forceinline__ void had4_col(const int16_t* s, const int stride, int16_t* c) { // hadamard_col4, :175-186
    const int16_t b0 = (int16_t)((s[0] + s[stride]) >> 1), b1 = (int16_t)((s[0] - s[stride]) >> 1);
    const int16_t b2 = (int16_t)((s[2 * stride] + s[3 * stride]) >> 1), b3 = (int16_t)((s[2 * stride] - s[3 * stride]) >> 1);
    c[0] = (int16_t)(b0 + b2); c[1] = (int16_t)(b1 + b3); c[2] = (int16_t)(b0 - b2); c[3] = (int16_t)(b1 - b3);
}
// mode 0: residual given (int16); mode 1: residual = input - pred (8-bit), the per-block body of hadamard_path_c
__global__ __launch_bounds__(64) void hadamard_kernel(const int16_t* res_base, const uint8_t* in_base, const uint8_t* pred_base, const SvtHipSatdDesc* descs,
                                                      const int n, int32_t* coeff_out, uint32_t* satd_out) {
    __shared__ int16_t res[32 * 32];
    __shared__ int32_t co[32 * 32];
    const int tid = threadIdx.x;
    const SvtHipSatdDesc d = descs[blockIdx.x];
    for (int i = tid; i < n * n; i += 64) {
        const int r = i / n, c = i - r * n;
        res[i] = res_base ? res_base[d.in_off + (size_t)r * d.in_stride + c]
                          : (int16_t)((int)in_base[d.in_off + (size_t)r * d.in_stride + c] - (int)pred_base[d.pred_off + (size_t)r * d.pred_stride + c]);
    }
    __syncthreads();
    if (n == 4) {
        if (tid == 0) {
            int16_t t[16], u[16];
            for (int i = 0; i < 4; i++) had4_col(res + i, 4, t + 4 * i);
            for (int i = 0; i < 4; i++) had4_col(t + i, 4, u + 4 * i);
            for (int i = 0; i < 16; i++) co[i] = u[i];
        }
    } else {
        const int nb = n >> 3; // 8x8 sub-blocks per side; sub-block order follows the recursive quadrant layout of :270-326
        if (tid < nb * nb) {
            int sy, sx, base;
            if (n == 8) { sy = sx = 0; base = 0; }
            else if (n == 16) { sy = tid >> 1; sx = tid & 1; base = tid * 64; }
            else { const int q = tid >> 2, s = tid & 3; sy = (q >> 1) * 2 + (s >> 1); sx = (q & 1) * 2 + (s & 1); base = q * 256 + s * 64; }
            int16_t t[64], u[64];
            const int16_t* p = res + sy * 8 * n + sx * 8;
#pragma unroll
            for (int i = 0; i < 8; i++) had8_col(p + i, n, t + 8 * i);
#pragma unroll
            for (int i = 0; i < 8; i++) had8_col(t + i, 8, u + 8 * i);
#pragma unroll
            for (int i = 0; i < 64; i++) co[base + i] = u[i];
        }
    }
    __syncthreads();
    if (n >= 16) { // 16x16 combination inside every 256-coefficient group (:270-297)
        for (int i = tid; i < (n * n) / 4; i += 64) {
            const int     g = i >> 6, k = i & 63, o = g * 256 + k;
            const int32_t a0 = co[o], a1 = co[o + 64], a2 = co[o + 128], a3 = co[o + 192];
            const int32_t b0 = (a0 + a1) >> 1, b1 = (a0 - a1) >> 1, b2 = (a2 + a3) >> 1, b3 = (a2 - a3) >> 1;
            co[o] = b0 + b2; co[o + 64] = b1 + b3; co[o + 128] = b0 - b2; co[o + 192] = b1 - b3;
        }
        __syncthreads();
    }
    if (n == 32) { // :299-326
        for (int k = tid; k < 256; k += 64) {
            const int32_t a0 = co[k], a1 = co[k + 256], a2 = co[k + 512], a3 = co[k + 768];
            const int32_t b0 = (a0 + a1) >> 2, b1 = (a0 - a1) >> 2, b2 = (a2 + a3) >> 2, b3 = (a2 - a3) >> 2;
            co[k] = b0 + b2; co[k + 256] = b1 + b3; co[k + 512] = b0 - b2; co[k + 768] = b1 - b3;
        }
        __syncthreads();
    }
    long long s = 0;
    for (int i = tid; i < n * n; i += 64) {
        const int32_t v = co[i];
        s += v < 0 ? -v : v;
        if (coeff_out) coeff_out[(size_t)blockIdx.x * n * n + i] = v;
    }
    s = wave_sum_i64(s);
    if (tid == 0 && satd_out) satd_out[blockIdx.x] = (uint32_t)s;
}
__global__ __launch_bounds__(64) void satd_kernel(const int32_t* coeff, int length, int* out) { // svt_aom_satd_c
    long long s = 0;
    for (int i = threadIdx.x; i < length; i += 64) { const int32_t v = coeff[i]; s += v < 0 ? -(long long)v : v; }
    s = wave_sum_i64(s);
    if (threadIdx.x == 0) out[0] = (int)s;
}
__global__ void residual_kernel(const void* in, uint32_t is, const void* pr, uint32_t ps, int16_t* res, uint32_t rs, uint32_t w, uint32_t h, int is16) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= w * h) return;
    const uint32_t y = i / w, x = i - y * w;
    const int a = is16 ? ((const uint16_t*)in)[y * is + x] : ((const uint8_t*)in)[y * is + x];
    const int b = is16 ? ((const uint16_t*)pr)[y * ps + x] : ((const uint8_t*)pr)[y * ps + x];
    res[y * rs + x] = (int16_t)(a - b);
}

// ---- Wiener statistics: csrc/lr_stats.hip (matrix cores) ----

// ---- self-guided projection ------------------------------------------------------------------------------------------
// mode 0: pixel_proj_error -> out[0]; mode 1: get_proj_subspace -> xq[0..1]
__global__ __launch_bounds__(256) void proj_kernel(const void* src, int width, int height, int src_stride, const void* dat, int dat_stride, const int32_t* flt0,
                                                   int f0s, const int32_t* flt1, int f1s, int xq0, int xq1, int r0, int r1, int is16, int mode, long long* out,
                                                   int32_t* xq_out) {
    __shared__ long long part[5][4];
    const int tid = threadIdx.x;
    long long a[5] = {0, 0, 0, 0, 0};
    for (int i = tid; i < width * height; i += 256) {
        const int y = i / width, x = i - y * width;
        const int d = is16 ? ((const uint16_t*)dat)[(size_t)y * dat_stride + x] : ((const uint8_t*)dat)[(size_t)y * dat_stride + x];
        const int s = is16 ? ((const uint16_t*)src)[(size_t)y * src_stride + x] : ((const uint8_t*)src)[(size_t)y * src_stride + x];
        const int u = d << 4;
        if (mode == 0) {
            int e;
            if (r0 > 0 || r1 > 0) {
                int v = u << 7;
                if (r0 > 0) v += xq0 * (flt0[(size_t)y * f0s + x] - u);
                if (r1 > 0) v += xq1 * (flt1[(size_t)y * f1s + x] - u);
                e = ((v + (1 << 10)) >> 11) - s;
            } else {
                e = d - s;
            }
            a[0] += (long long)e * e;
        } else {
            const long long sd = (long long)(s << 4) - u;
            const long long f1 = r0 > 0 ? (long long)flt0[(size_t)y * f0s + x] - u : 0, f2 = r1 > 0 ? (long long)flt1[(size_t)y * f1s + x] - u : 0;
            a[0] += f1 * f1; a[1] += f2 * f2; a[2] += f1 * f2; a[3] += f1 * sd; a[4] += f2 * sd;
        }
    }
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const long long v = wave_sum_i64(a[k]);
        if ((tid & 63) == 0) part[k][tid >> 6] = v;
    }
    __syncthreads();
    if (tid == 0) {
        long long t[5];
        for (int k = 0; k < 5; k++) t[k] = part[k][0] + part[k][1] + part[k][2] + part[k][3];
        if (mode == 0) { out[0] = t[0]; return; }
        // svt_get_proj_subspace_c :470-498 -- the integer sums equal the reference's double sums exactly (all < 2^53)
        const double size = (double)(width * height);
        const double H00 = (double)t[0] / size, H11 = (double)t[1] / size, H01 = (double)t[2] / size, C0 = (double)t[3] / size, C1 = (double)t[4] / size;
        int x0 = 0, x1 = 0;
        if (r0 == 0) {
            if (!(H11 < 1e-8)) x1 = (int)rint(C1 / H11 * 128);
        } else if (r1 == 0) {
            if (!(H00 < 1e-8)) x0 = (int)rint(C0 / H00 * 128);
        } else {
            const double det = H00 * H11 - H01 * H01;
            if (!(det < 1e-8)) { x0 = (int)rint((H11 * C0 - H01 * C1) / det * 128); x1 = (int)rint((H00 * C1 - H01 * C0) / det * 128); }
        }
        xq_out[0] = x0; xq_out[1] = x1;
    }
}

} // namespace

extern "C" {

void svt_hip_hadamard_satd_batch(const uint8_t* input_base, const uint8_t* pred_base, const SvtHipSatdDesc* descs, uint32_t n, int tx_n, uint32_t* satd_out,
                                 int32_t* coeff_out, void* stream) {
    svthip::ensure_device();
    if (n == 0) return;
    hipLaunchKernelGGL(hadamard_kernel, dim3(n), dim3(64), 0, (hipStream_t)stream, (const int16_t*)nullptr, input_base, pred_base, descs, tx_n, coeff_out, satd_out);
    SVT_LAUNCH_CHECK();
}
// ---- RTCD-signature single-call forms ---------------------------------------------------------------------------------
int svt_aom_satd_hip(const int32_t* coeff, int length) {
    svthip::HostCall& c = svthip::host_call();
    c.begin_small();
    c.reserve((size_t)length * 4 + 1024, (size_t)length * 4 + 1024);
    int32_t* d = (int32_t*)c.dalloc((size_t)length * 4);
    int*     o = (int*)c.dalloc(4);
    c.up(d, coeff, (size_t)length * 4);
    hipLaunchKernelGGL(satd_kernel, dim3(1), dim3(64), 0, c.stream, (const int32_t*)d, length, o);
    SVT_LAUNCH_CHECK();
    int r;
    c.down(&r, o, 4);
    return r;
}
void svt_aom_hadamard_nxn_hip(const int16_t* src_diff, ptrdiff_t src_stride, int32_t* coeff, int n) {
    svthip::HostCall& c = svthip::host_call();
    c.begin_small();
    c.reserve((size_t)n * n * 8 + 4096, (size_t)n * n * 8 + 4096);
    int16_t*        d  = (int16_t*)c.dalloc((size_t)n * n * 2);
    int32_t*        o  = (int32_t*)c.dalloc((size_t)n * n * 4);
    SvtHipSatdDesc* dd = (SvtHipSatdDesc*)c.dalloc(sizeof(SvtHipSatdDesc));
    c.up2d(d, (size_t)n * 2, src_diff, (size_t)src_stride * 2, (size_t)n * 2, n);
    SvtHipSatdDesc ds = {0, 0, (uint32_t)n, 0};
    c.up(dd, &ds, sizeof(ds));
    hipLaunchKernelGGL(hadamard_kernel, dim3(1), dim3(64), 0, c.stream, (const int16_t*)d, (const uint8_t*)nullptr, (const uint8_t*)nullptr, (const SvtHipSatdDesc*)dd, n, o,
                       (uint32_t*)nullptr);
    SVT_LAUNCH_CHECK();
    c.down(coeff, o, (size_t)n * n * 4);
}
void svt_aom_hadamard_4x4_hip(const int16_t* s, ptrdiff_t st, int32_t* c) { svt_aom_hadamard_nxn_hip(s, st, c, 4); }
void svt_aom_hadamard_8x8_hip(const int16_t* s, ptrdiff_t st, int32_t* c) { svt_aom_hadamard_nxn_hip(s, st, c, 8); }
void svt_aom_hadamard_16x16_hip(const int16_t* s, ptrdiff_t st, int32_t* c) { svt_aom_hadamard_nxn_hip(s, st, c, 16); }
void svt_aom_hadamard_32x32_hip(const int16_t* s, ptrdiff_t st, int32_t* c) { svt_aom_hadamard_nxn_hip(s, st, c, 32); }
// hadamard_path (enc_mode_config.c:2147-2215) for a square block of `block_size` 8-bit pixels: tiles of min(32, block_size)
uint32_t svt_hadamard_path_hip(const uint8_t* input, uint32_t in_stride, const uint8_t* pred, uint32_t pred_stride, int block_size) {
    const int tx = block_size > 32 ? 32 : block_size, nt = block_size / tx;
    svthip::HostCall& c = svthip::host_call();
    c.begin_small();
    const size_t pitch = svthip::align_up((size_t)block_size, 16);
    c.reserve(2 * pitch * block_size + 4096, 2 * pitch * block_size + 4096);
    uint8_t*        di = (uint8_t*)c.dalloc(pitch * block_size);
    uint8_t*        dp = (uint8_t*)c.dalloc(pitch * block_size);
    SvtHipSatdDesc* dd = (SvtHipSatdDesc*)c.dalloc(sizeof(SvtHipSatdDesc) * nt * nt);
    uint32_t*       ds = (uint32_t*)c.dalloc(4 * nt * nt);
    c.up2d(di, pitch, input, in_stride, block_size, block_size);
    c.up2d(dp, pitch, pred, pred_stride, block_size, block_size);
    SvtHipSatdDesc h[16];
    for (int r = 0; r < nt; r++)
        for (int q = 0; q < nt; q++) h[r * nt + q] = {(uint64_t)(r * tx * pitch + q * tx), (uint64_t)(r * tx * pitch + q * tx), (uint32_t)pitch, (uint32_t)pitch};
    c.up(dd, h, sizeof(SvtHipSatdDesc) * nt * nt);
    svt_hip_hadamard_satd_batch(di, dp, dd, nt * nt, tx, ds, nullptr, c.stream);
    uint32_t s[16], tot = 0;
    c.down(s, ds, 4 * nt * nt);
    for (int i = 0; i < nt * nt; i++) tot += s[i];
    return tot;
}
static void residual_host(const void* input, uint32_t is, const void* pred, uint32_t ps, int16_t* residual, uint32_t rs, uint32_t w, uint32_t h, int is16) {
    svthip::HostCall& c = svthip::host_call();
    c.begin_small();
    const size_t px = is16 ? 2 : 1, n = (size_t)w * h;
    c.reserve(n * (2 * px + 2) + 4096, n * (2 * px + 4) + 4096);
    void*    di = c.dalloc(n * px);
    void*    dp = c.dalloc(n * px);
    int16_t* dr = (int16_t*)c.dalloc(n * 2);
    c.up2d(di, w * px, input, is * px, w * px, h);
    c.up2d(dp, w * px, pred, ps * px, w * px, h);
    hipLaunchKernelGGL(residual_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c.stream, (const void*)di, w, (const void*)dp, w, dr, w, w, h, is16);
    SVT_LAUNCH_CHECK();
    c.down2d(residual, (size_t)rs * 2, dr, (size_t)w * 2, (size_t)w * 2, h);
}
void svt_residual_kernel8bit_hip(uint8_t* input, uint32_t is, uint8_t* pred, uint32_t ps, int16_t* residual, uint32_t rs, uint32_t w, uint32_t h) {
    residual_host(input, is, pred, ps, residual, rs, w, h, 0);
}
void svt_residual_kernel16bit_hip(uint16_t* input, uint32_t is, uint16_t* pred, uint32_t ps, int16_t* residual, uint32_t rs, uint32_t w, uint32_t h) {
    residual_host(input, is, pred, ps, residual, rs, w, h, 1);
}

// px = bytes per sample of the caller's pictures: svt_av1_compute_stats_highbd takes 16-bit pictures at EVERY bit depth, 8 included (the 16-bit pipeline of an 8-bit encode;
// the reference's own av1_compute_stats_test_hbd runs it at EB_EIGHT_BIT, test/RestorationPickTest.cc:576-583)
static void stats_host(int win, const void* dgd, const void* src, int h_start, int h_end, int v_start, int v_end, int dgd_stride, int src_stride, int64_t* M,
                       int64_t* H, int bit_depth, int px) {
    const int hw = win >> 1, W = h_end - h_start, Hh = v_end - v_start;
    svthip::HostCall& c = svthip::host_call();
    c.begin();
    const size_t dp = svthip::align_up((size_t)(W + 2 * hw) * px, 16), sp = svthip::align_up((size_t)W * px, 16);
    c.reserve(dp * (Hh + 2 * hw) + sp * Hh + 49 * 50 * 8 + 4096, dp * (Hh + 2 * hw) + sp * Hh + 49 * 50 * 8 + 4096);
    uint8_t*    dd = (uint8_t*)c.dalloc(dp * (Hh + 2 * hw));
    uint8_t*    ds = (uint8_t*)c.dalloc(sp * Hh);
    SvtHipRect* dr = (SvtHipRect*)c.dalloc(sizeof(SvtHipRect));
    int64_t*    dM = (int64_t*)c.dalloc(49 * 8);
    int64_t*    dH = (int64_t*)c.dalloc(49 * 49 * 8);
    c.up2d(dd, dp, (const uint8_t*)dgd + ((size_t)(v_start - hw) * dgd_stride + h_start - hw) * px, (size_t)dgd_stride * px, (size_t)(W + 2 * hw) * px, Hh + 2 * hw);
    c.up2d(ds, sp, (const uint8_t*)src + ((size_t)v_start * src_stride + h_start) * px, (size_t)src_stride * px, (size_t)W * px, Hh);
    SvtHipRect R = {0, W, 0, Hh};
    c.up(dr, &R, sizeof(R));
    // origin of the uploaded dgd rectangle is (hw, hw)
    svt_hip_lr_compute_stats_batch_samples(dd + (hw * dp + hw * px), ds, dr, 1, W, Hh, (int)(dp / px), (int)(sp / px), win, bit_depth, px, dM, dH, c.stream);
    const int w2 = win * win;
    int64_t   hM[49], hH[49 * 49];
    c.down_later(hM, dM, 49 * 8);
    c.down_later(hH, dH, 49 * 49 * 8);
    c.finish();
    memcpy(M, hM, sizeof(int64_t) * w2);
    memcpy(H, hH, sizeof(int64_t) * w2 * w2);
}
void svt_av1_compute_stats_hip(int32_t wiener_win, const uint8_t* dgd, const uint8_t* src, int32_t h_start, int32_t h_end, int32_t v_start, int32_t v_end,
                               int32_t dgd_stride, int32_t src_stride, int64_t* M, int64_t* H) {
    stats_host(wiener_win, dgd, src, h_start, h_end, v_start, v_end, dgd_stride, src_stride, M, H, 8, 1);
}
void svt_av1_compute_stats_highbd_hip(int32_t wiener_win, const uint8_t* dgd8, const uint8_t* src8, int32_t h_start, int32_t h_end, int32_t v_start, int32_t v_end,
                                      int32_t dgd_stride, int32_t src_stride, int64_t* M, int64_t* H, unsigned int bit_depth) {
    stats_host(wiener_win, (const void*)((uintptr_t)dgd8 << 1), (const void*)((uintptr_t)src8 << 1), h_start, h_end, v_start, v_end, dgd_stride, src_stride, M, H,
               (int)bit_depth, 2);
}

static void proj_host(int mode, const void* src, int width, int height, int src_stride, const void* dat, int dat_stride, const int32_t* flt0, int f0s,
                      const int32_t* flt1, int f1s, int32_t* xq, int r0, int r1, int is16, int64_t* err) {
    svthip::HostCall& c = svthip::host_call();
    c.begin();
    const size_t px = is16 ? 2 : 1, n = (size_t)width * height;
    c.reserve(n * (2 * px + 8) + 4096, n * (2 * px + 8) + 4096);
    void*     ds = c.dalloc(n * px);
    void*     dd = c.dalloc(n * px);
    int32_t*  d0 = (int32_t*)c.dalloc(n * 4);
    int32_t*  d1 = (int32_t*)c.dalloc(n * 4);
    long long* de = (long long*)c.dalloc(8);
    int32_t*  dx = (int32_t*)c.dalloc(8);
    c.up2d(ds, width * px, src, (size_t)src_stride * px, width * px, height);
    c.up2d(dd, width * px, dat, (size_t)dat_stride * px, width * px, height);
    if (r0 > 0) c.up2d(d0, (size_t)width * 4, flt0, (size_t)f0s * 4, (size_t)width * 4, height);
    if (r1 > 0) c.up2d(d1, (size_t)width * 4, flt1, (size_t)f1s * 4, (size_t)width * 4, height);
    hipLaunchKernelGGL(proj_kernel, dim3(1), dim3(256), 0, c.stream, (const void*)ds, width, height, width, (const void*)dd, width, (const int32_t*)d0, width,
                       (const int32_t*)d1, width, mode == 0 ? xq[0] : 0, mode == 0 ? xq[1] : 0, r0, r1, is16, mode, de, dx);
    SVT_LAUNCH_CHECK();
    if (mode == 0) c.down(err, de, 8);
    else c.down(xq, dx, 8);
}
int64_t svt_av1_lowbd_pixel_proj_error_hip(const uint8_t* src8, int32_t width, int32_t height, int32_t src_stride, const uint8_t* dat8, int32_t dat_stride,
                                           int32_t* flt0, int32_t flt0_stride, int32_t* flt1, int32_t flt1_stride, int32_t xq[2], const SvtHipSgrParams* p) {
    int64_t e = 0;
    proj_host(0, src8, width, height, src_stride, dat8, dat_stride, flt0, flt0_stride, flt1, flt1_stride, xq, p->r[0], p->r[1], 0, &e);
    return e;
}
int64_t svt_av1_highbd_pixel_proj_error_hip(const uint8_t* src8, int32_t width, int32_t height, int32_t src_stride, const uint8_t* dat8, int32_t dat_stride,
                                            int32_t* flt0, int32_t flt0_stride, int32_t* flt1, int32_t flt1_stride, int32_t xq[2], const SvtHipSgrParams* p) {
    int64_t e = 0;
    proj_host(0, (const void*)((uintptr_t)src8 << 1), width, height, src_stride, (const void*)((uintptr_t)dat8 << 1), dat_stride, flt0, flt0_stride, flt1, flt1_stride, xq,
              p->r[0], p->r[1], 1, &e);
    return e;
}
void svt_get_proj_subspace_hip(const uint8_t* src8, int32_t width, int32_t height, int32_t src_stride, const uint8_t* dat8, int32_t dat_stride,
                               int32_t use_highbitdepth, int32_t* flt0, int32_t flt0_stride, int32_t* flt1, int32_t flt1_stride, int32_t* xq, const SvtHipSgrParams* p) {
    const void* s = use_highbitdepth ? (const void*)((uintptr_t)src8 << 1) : (const void*)src8;
    const void* d = use_highbitdepth ? (const void*)((uintptr_t)dat8 << 1) : (const void*)dat8;
    proj_host(1, s, width, height, src_stride, d, dat_stride, flt0, flt0_stride, flt1, flt1_stride, xq, p->r[0], p->r[1], use_highbitdepth, nullptr);
}

} // extern "C"

SVT_HIP_DEFINE_WARM(misc) // (svt_hip_warmup loads this translation unit's code object at encoder initialisation: svt_hip_common.h)
