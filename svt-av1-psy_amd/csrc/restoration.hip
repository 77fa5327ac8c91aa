// restoration.hip -- loop restoration for gfx950: separable 7-tap Wiener with "add src",
// self-guided (box r = 2 "fast" + r = 1) filter, and the stripe / restoration-unit frame driver.
//
// One workgroup filters one processing unit (64 >> ss_x columns of one 64 >> ss_y row stripe).  The unit plus its
// 3-pixel halo is staged into LDS once, with the stripe-boundary substitution of restoration.c:288-332 (rows above /
// below a stripe come from the saved deblocked lines, frame edges are edge-extended) folded into the staging, so the
// filters themselves never touch HBM again: Wiener keeps its clamped horizontal pass in LDS; the self-guided filter
// builds A/B (box sums -> x_by_xplus1 / one_by_x tables, restoration.c:705-764) in LDS and applies the 3x3 weighting
// from there.  Out of place, 2 B/px in + 2 B/px out at 10 bit = the 4 B/px of SURVEY 8(d).
#include "lr_core.h"

namespace {

constexpr int kSgrRH[16][2] = {{2, 1}, {2, 1}, {2, 1}, {2, 1}, {2, 1}, {2, 1}, {2, 1}, {2, 1}, {2, 1}, {2, 1}, {0, 1}, {0, 1}, {0, 1}, {0, 1}, {2, 0}, {2, 0}}; // host copy
// LDS: tile | union { Wiener mid (u16 [TH][64]) ; self-guided A (u16 [66][66], padded to a dword multiple) + B (i32 [66][66]) } = 36.2 KB -> 4 workgroups / CU
constexpr size_t LR_A_BYTES = (size_t)66 * 66 * 2 + 8;
constexpr size_t LR_SMEM    = (size_t)TH * TW * 2 + LR_A_BYTES + (size_t)66 * 66 * 4 + 512; // + x_by_xplus1 table

// ---- frame kernel ---------------------------------------------------------------------------------------------------------
// One workgroup = 64 >> ss_x columns x at most LR_UR rows of one stripe (a 64-row luma stripe is cut in two: twice the workgroups, half the
// LDS -> 8 resident workgroups per CU to hide the staging round trip; the cut is invisible to the filters because the rows on the far
// side of it are ordinary rows of the same stripe).
constexpr size_t lr_frame_a(const int ur) { return (size_t)(ur + 2) * 66 * 2 + 8; }
constexpr size_t lr_frame_mid(const int ur) { return (size_t)(ur + 6) * 64 * 2; }
constexpr size_t lr_frame_ab(const int ur) { return lr_frame_a(ur) + (size_t)(ur + 2) * 66 * 4; }
constexpr size_t lr_frame_union(const int ur) { return lr_frame_ab(ur) > lr_frame_mid(ur) ? lr_frame_ab(ur) : lr_frame_mid(ur); }
constexpr size_t lr_frame_smem(const int ur) { return (size_t)(ur + 6) * TW * 2 + lr_frame_union(ur) + 512; }
// grid (unit columns, halves of a stripe, stripes); nhu x nvu restoration units, ushift = log2(unit_size) or -1 (the host does the divisions once)
template <int LR_UR> // rows of a stripe per workgroup: 32 (two workgroups per 64-row luma stripe) or 64
__global__ __launch_bounds__(256) void lr_frame_kernel(const SvtHipLrParams P, const int nhu, const int nvu, const int ushift, const int stripe0 /* first stripe of the launch */) {
    HIP_DYNAMIC_SHARED(uint16_t, smem)
    uint16_t* tile = smem;
    uint16_t* mid  = tile + (LR_UR + 6) * TW;                  // Wiener only
    uint16_t* A16  = mid;                                      // self-guided only (aliases mid)
    int32_t*  B32  = (int32_t*)((uint8_t*)mid + lr_frame_a(LR_UR));
    uint16_t* xlut = (uint16_t*)((uint8_t*)mid + lr_frame_union(LR_UR));
    const int tid = threadIdx.x;
    const int pw = (int)P.width, ph = (int)P.height, off = 8 >> P.ss_y, sh = 64 >> P.ss_y, cw = 64 >> P.ss_x;
    TileSrc s;
    s.data = P.data; s.above = P.boundary_above; s.below = P.boundary_below;
    s.stride = (int)P.stride; s.bstride = (int)P.boundary_stride; s.w = pw; s.h = ph; s.highbd = P.highbd;
    // XCD-aware order over the linear workgroup index (x fastest): neighbouring stripe columns / halves share their 3-sample halos' cache lines
    const uint32_t lin = xcd_remap(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), gridDim.x * gridDim.y * gridDim.z);
    const int      bx = (int)(lin % gridDim.x), byz = (int)(lin / gridDim.x), bhalf = byz % (int)gridDim.y, bz = byz / (int)gridDim.y;
    s.stripe_idx = stripe0 + bz;
    s.stripe_top = s.stripe_idx * sh - off < 0 ? 0 : s.stripe_idx * sh - off;
    s.stripe_bot = (s.stripe_idx + 1) * sh - off > ph ? ph : (s.stripe_idx + 1) * sh - off;
    s.x0 = bx * cw; s.y0 = s.stripe_top + bhalf * LR_UR;
    s.uw = pw - s.x0 < cw ? pw - s.x0 : cw;
    s.uh = s.stripe_bot - s.y0 < LR_UR ? s.stripe_bot - s.y0 : LR_UR;
    if (s.uh <= 0 || s.uw <= 0) return;
    const int us = (int)P.unit_size;
    int       ur = ushift >= 0 ? (s.y0 + off) >> ushift : (s.y0 + off) / us, uc = ushift >= 0 ? s.x0 >> ushift : s.x0 / us;
    ur = ur >= nvu ? nvu - 1 : ur; uc = uc >= nhu ? nhu - 1 : uc;
    const SvtHipLrUnit u = P.units[ur * nhu + uc];
    const int highbd = P.highbd, bd = P.bit_depth;
    void* dst = P.dst;
    const size_t dstride = P.dst_stride;
    const int x0 = s.x0, y0 = s.y0;
    // two horizontally adjacent samples per store instruction (one dword / one halfword when the address allows it): single-sample stores
    // touched every 64-byte line twice and the write traffic was 5.5 x the plane (profiles/r02_call3_pmc_traffic.json).  The unit's first output
    // sample is a wave-uniform address and everything after it a 32-bit byte offset (r < 64, c < 64: fits for any stride the API accepts as int32 / 64),
    // and whether pair stores are aligned is decided once per workgroup (base and row pitch both multiples of the pair size).
    const int      pxb = highbd ? 2 : 1;
    uint8_t* const dst0 = (uint8_t*)dst + ((size_t)y0 * dstride + (size_t)x0) * pxb;
    const uint32_t dpitch = (uint32_t)dstride * pxb;
    const bool     pair_ok = !(((uintptr_t)dst0 | dpitch) & (uintptr_t)(2 * pxb - 1));
    auto store = [&](int r, int c, int v0, int v1, bool has1) {
        uint8_t* q = dst0 + ((uint32_t)r * dpitch + (uint32_t)(c * pxb));
        if (highbd) {
            if (has1 && pair_ok) *(uint32_t*)q = (uint32_t)v0 | ((uint32_t)v1 << 16);
            else { ((uint16_t*)q)[0] = (uint16_t)v0; if (has1) ((uint16_t*)q)[1] = (uint16_t)v1; }
        } else {
            if (has1 && pair_ok) *(uint16_t*)q = (uint16_t)(v0 | (v1 << 8));
            else { q[0] = (uint8_t)v0; if (has1) q[1] = (uint8_t)v1; }
        }
    };
    stage_tile<(LR_UR + 6 + 15) / 16>(tile, s, tid);
    __syncthreads();
    if (u.rtype == 1) {
        WienerTaps t;
#pragma unroll
        for (int k = 0; k < 8; k++) { t.fx[k] = u.hfilter[k]; t.fy[k] = u.vfilter[k]; }
        wiener_tile(tile, mid, t, s.uw, s.uh, bd, tid, store);
    } else if (u.rtype == 2) {
        const int idx = u.ep & 15;
        sgr_tile(tile, A16, B32, xlut, idx, s.uw, s.uh, bd, tid, [](int, int, int32_t) {},
                 [&](int r, int c, int32_t f0a, int32_t f1a, int32_t f0b, int32_t f1b, bool has1) {
                     store(r, c, sgr_combine(tile[(r + 3) * TW + c + 3], f0a, f1a, idx, u.xqd[0], u.xqd[1], bd),
                           sgr_combine(tile[(r + 3) * TW + c + 4], f0b, f1b, idx, u.xqd[0], u.xqd[1], bd), has1);
                 });
    } else {
        for (int i = tid; i < s.uh * 32; i += 256) {
            const int r = i >> 5, c = (i & 31) * 2;
            if (c < s.uw) store(r, c, tile[(r + 3) * TW + c + 3], tile[(r + 3) * TW + c + 4], c + 1 < s.uw);
        }
    }
}

// ---- single-call kernels: a w x h area whose source carries its own 3-pixel border (the RTCD contracts) ------------------
// kind 0: wiener -> dst pixels; 1: self-guided apply -> dst pixels; 2: self-guided flt0 / flt1 (int32)
__global__ __launch_bounds__(256) void lr_block_kernel(const void* src /* origin at pixel (0,0); border readable */, int sstride, void* dst, int dstride, int w, int h,
                                                       int highbd, int bd, int kind, WienerTaps taps, int idx, int xqd0, int xqd1, int32_t* f0out, int32_t* f1out,
                                                       int fstride) {
    HIP_DYNAMIC_SHARED(uint16_t, smem)
    uint16_t* tile = smem;
    uint16_t* mid  = tile + TH * TW;                           // Wiener only
    uint16_t* A16  = mid;                                      // self-guided only (aliases mid)
    int32_t*  B32  = (int32_t*)((uint8_t*)mid + LR_A_BYTES);
    uint16_t* xlut = (uint16_t*)(B32 + 66 * 66);
    const int tid = threadIdx.x;
    TileSrc s;
    s.data = src; s.above = s.below = nullptr; s.stride = sstride; s.bstride = 0; s.w = 0; s.h = 0; s.highbd = highbd;
    s.stripe_idx = 0; s.stripe_top = 0; s.stripe_bot = 0;
    s.x0 = blockIdx.x * 64; s.y0 = blockIdx.y * 64;
    s.uw = w - s.x0 < 64 ? w - s.x0 : 64; s.uh = h - s.y0 < 64 ? h - s.y0 : 64;
    const int x0 = s.x0, y0 = s.y0;
    auto store = [&](int r, int c, int v0, int v1, bool has1) {
        const size_t o = (size_t)(y0 + r) * dstride + x0 + c;
        if (highbd) { ((uint16_t*)dst)[o] = (uint16_t)v0; if (has1) ((uint16_t*)dst)[o + 1] = (uint16_t)v1; }
        else { ((uint8_t*)dst)[o] = (uint8_t)v0; if (has1) ((uint8_t*)dst)[o + 1] = (uint8_t)v1; }
    };
    stage_tile<(TH + 15) / 16>(tile, s, tid);
    __syncthreads();
    if (kind == 0) {
        wiener_tile(tile, mid, taps, s.uw, s.uh, bd, tid, store);
    } else if (kind == 1) {
        sgr_tile(tile, A16, B32, xlut, idx, s.uw, s.uh, bd, tid, [](int, int, int32_t) {},
                 [&](int r, int c, int32_t a0, int32_t b0, int32_t a1, int32_t b1, bool has1) {
                     store(r, c, sgr_combine(tile[(r + 3) * TW + c + 3], a0, b0, idx, xqd0, xqd1, bd), sgr_combine(tile[(r + 3) * TW + c + 4], a1, b1, idx, xqd0, xqd1, bd), has1);
                 });
    } else {
        const bool p0 = kSgrR[idx][0] > 0, p1 = kSgrR[idx][1] > 0;
        sgr_tile(tile, A16, B32, xlut, idx, s.uw, s.uh, bd, tid, [&](int r, int c, int32_t v) { if (p0) f0out[(size_t)(y0 + r) * fstride + x0 + c] = v; },
                 [&](int r, int c, int32_t, int32_t b0, int32_t, int32_t b1, bool has1) {
                     if (p1) { f1out[(size_t)(y0 + r) * fstride + x0 + c] = b0; if (has1) f1out[(size_t)(y0 + r) * fstride + x0 + c + 1] = b1; }
                 });
    }
}

// host helper for the single-call forms: upload (h + 7) x (w + 8) pixels around the block, run, download
void lr_block_host(const void* src, int sstride, void* dst, int dstride, int w, int h, int highbd, int bd, int kind, const int16_t* fx, const int16_t* fy,
                   int idx, const int32_t* xqd, int32_t* flt0, int32_t* flt1, int fstride) {
    svthip::HostCall& c = svthip::host_call();
    c.begin_small();
    const size_t px = highbd ? 2 : 1;
    const size_t pw = svthip::align_up((size_t)(w + 8) * px, 16), rows = (size_t)h + 6;
    const size_t dpitch = svthip::align_up((size_t)w * px, 16);
    const size_t fbytes = kind == 2 ? (size_t)w * h * 4 : 0;
    c.reserve(pw * rows + dpitch * h + 2 * fbytes + 8192, pw * rows + 2 * dpitch * h + 2 * fbytes + 8192);
    uint8_t* ds = (uint8_t*)c.dalloc(pw * rows);
    uint8_t* dd = (uint8_t*)c.dalloc(dpitch * h + 16);
    int32_t* d0 = kind == 2 ? (int32_t*)c.dalloc(fbytes) : nullptr;
    int32_t* d1 = kind == 2 ? (int32_t*)c.dalloc(fbytes) : nullptr;
    // the reference reads rows -3..h+2 and columns -3..w+3 (tap 7 = 0 still dereferences x+4 only inside the multiply; we
    // upload -3..w+2 and zero-fill the rest)
    c.up2d(ds, pw, (const uint8_t*)src - ((size_t)3 * sstride + 3) * px, (size_t)sstride * px, (size_t)(w + 6) * px, rows);
    WienerTaps t;
    memset(&t, 0, sizeof(t));
    if (fx) { memcpy(t.fx, fx, 16); memcpy(t.fy, fy, 16); }
    const void* origin = ds + (3 * pw + 3 * px);
    hipLaunchKernelGGL(lr_block_kernel, dim3((w + 63) / 64, (h + 63) / 64), dim3(256), LR_SMEM, c.stream, origin, (int)(pw / px), (void*)dd, (int)(dpitch / px), w, h,
                       highbd, bd, kind, t, idx, xqd ? xqd[0] : 0, xqd ? xqd[1] : 0, d0, d1, w);
    SVT_LAUNCH_CHECK();
    if (kind == 2) {
        if (kSgrRH[idx][0]) c.down2d_later(flt0, (size_t)fstride * 4, d0, (size_t)w * 4, (size_t)w * 4, h);
        if (kSgrRH[idx][1]) c.down2d_later(flt1, (size_t)fstride * 4, d1, (size_t)w * 4, (size_t)w * 4, h);
        c.finish(); // ONE commit point
    } else {
        c.down2d(dst, (size_t)dstride * px, dd, dpitch, (size_t)w * px, h);
    }
}

} // namespace

extern "C" {

void svt_hip_lr_filter_frame(const SvtHipLrParams* params, void* stream) { svt_hip_lr_filter_frame_stripes(params, 0, -1, stream); }
// stripes [stripe_begin, stripe_end) of the plane only (a strip of rows when a picture is split over several GPUs: the staging still reads the 3-sample halos and
// the saved boundary lines from the full inputs); stripe_end < 0 = to the last stripe
void svt_hip_lr_filter_frame_stripes(const SvtHipLrParams* params, int stripe_begin, int stripe_end, void* stream) {
    svthip::ensure_device();
    const SvtHipLrParams& P = *params;
    const int sh = 64 >> P.ss_y, off = 8 >> P.ss_y, cw = 64 >> P.ss_x;
    const int all_stripes = ((int)P.height + off + sh - 1) / sh;
    if (stripe_end < 0 || stripe_end > all_stripes) stripe_end = all_stripes;
    if (stripe_begin < 0) stripe_begin = 0;
    if (stripe_begin >= stripe_end) return;
    const int n_stripes = stripe_end - stripe_begin;
    const int n_cols    = ((int)P.width + cw - 1) / cw;
    const int ur_env = svthip::tuning_lr_rows_per_workgroup(); // SVT_HIP_LR_UR, read once: 32 (default, the measured optimum), 16 or 64 (profiles/r02_lr_walk_experiment.txt)
    const int us        = (int)P.unit_size;
    int       nvu = ((int)P.height + (us >> 1)) / us, nhu = ((int)P.width + (us >> 1)) / us, ushift = -1;
    nvu = nvu > 0 ? nvu : 1; nhu = nhu > 0 ? nhu : 1;
    if (us > 0 && !(us & (us - 1))) ushift = __builtin_ctz((unsigned)us);
    if (ur_env == 64) {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(lr_frame_kernel<64>), dim3(n_cols, 1, n_stripes), dim3(256), lr_frame_smem(64), (hipStream_t)stream, P, nhu, nvu, ushift, stripe_begin);
    } else if (ur_env == 16) {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(lr_frame_kernel<16>), dim3(n_cols, (sh + 15) / 16, n_stripes), dim3(256), lr_frame_smem(16), (hipStream_t)stream, P, nhu, nvu, ushift, stripe_begin);
    } else {
        const int nsplit = (sh + 31) / 32;
        hipLaunchKernelGGL(HIP_KERNEL_NAME(lr_frame_kernel<32>), dim3(n_cols, nsplit, n_stripes), dim3(256), lr_frame_smem(32), (hipStream_t)stream, P, nhu, nvu, ushift, stripe_begin);
    }
    SVT_LAUNCH_CHECK();
}

// Host-pointer form of the frame filter (what a seam around svt_av1_loop_restoration_filter_frame, rest_process.c:632, calls per restored plane): data, the two
// boundary-line buffers (pointing at frame column 0, i.e. past the reference's RESTORATION_EXTRA_HORZ margin), units and dst are host pointers; dst may be data.
int svt_hip_lr_filter_frame_host(const SvtHipLrParams* params) {
    SVT_HIP_ENTRY_TRY
    svthip::ensure_device();
    SvtHipLrParams P = *params;
    const size_t px = P.highbd ? 2 : 1, w = P.width, h = P.height;
    const int    sh = 64 >> P.ss_y, off = 8 >> P.ss_y, us = (int)P.unit_size;
    const size_t n_stripes = ((size_t)h + off + sh - 1) / sh, pitch = svthip::align_up(w * px, 16);
    int nvu = ((int)h + (us >> 1)) / us, nhu = ((int)w + (us >> 1)) / us;
    nvu = nvu > 0 ? nvu : 1; nhu = nhu > 0 ? nhu : 1;
    const size_t ub = (size_t)nvu * nhu * sizeof(SvtHipLrUnit);
    svthip::HostCallLease lease; // (a pooled arena: see svt_hip_common.h)
    svthip::HostCall& c = *lease;
    c.begin();
    c.reserve(pitch * (2 * h + 4 * n_stripes) + ub + 8192, pitch * (2 * h + 4 * n_stripes) + ub + 8192);
    uint8_t* d_data  = (uint8_t*)c.dalloc(pitch * h);
    uint8_t* d_dst   = (uint8_t*)c.dalloc(pitch * h);
    uint8_t* d_above = (uint8_t*)c.dalloc(pitch * 2 * n_stripes);
    uint8_t* d_below = (uint8_t*)c.dalloc(pitch * 2 * n_stripes);
    SvtHipLrUnit* d_units = (SvtHipLrUnit*)c.dalloc(ub);
    c.up2d(d_data, pitch, params->data, (size_t)params->stride * px, w * px, h);
    c.up2d(d_above, pitch, params->boundary_above, (size_t)params->boundary_stride * px, w * px, 2 * n_stripes);
    c.up2d(d_below, pitch, params->boundary_below, (size_t)params->boundary_stride * px, w * px, 2 * n_stripes);
    c.up(d_units, params->units, ub);
    P.data = d_data; P.dst = d_dst; P.boundary_above = d_above; P.boundary_below = d_below; P.units = d_units;
    P.stride = P.dst_stride = P.boundary_stride = (uint32_t)(pitch / px);
    svthip::lr_frame_dispatch(&P, c.stream);
    c.down2d(params->dst, (size_t)params->dst_stride * px, d_dst, pitch, w * px, h);
    return 0;
    SVT_HIP_ENTRY_CATCH(SVT_HIP_E_DEVICE)
}

// svt_av1_wiener_convolve_add_src -> _c (convolve.c:100-147); conv_params is rebuilt from the bit depth exactly as
// get_conv_params_wiener does (convolve.h:70-88), which is what every caller passes (restoration.c:443, :1021)
void svt_av1_wiener_convolve_add_src_hip(const uint8_t* src, ptrdiff_t src_stride, uint8_t* dst, ptrdiff_t dst_stride, const int16_t* filter_x,
                                         const int16_t* filter_y, int32_t w, int32_t h, const SvtHipConvolveParams* conv_params) {
    (void)conv_params;
    lr_block_host(src, (int)src_stride, dst, (int)dst_stride, w, h, 0, 8, 0, filter_x, filter_y, 0, nullptr, nullptr, nullptr, 0);
}
// highbd forms take CONVERT_TO_BYTEPTR()-style pointers: real address = (uintptr_t)p << 1 (definitions.h)
void svt_av1_highbd_wiener_convolve_add_src_hip(const uint8_t* src8, ptrdiff_t src_stride, uint8_t* dst8, ptrdiff_t dst_stride, const int16_t* filter_x,
                                                const int16_t* filter_y, int32_t w, int32_t h, const SvtHipConvolveParams* conv_params, int32_t bd) {
    (void)conv_params;
    lr_block_host((const void*)((uintptr_t)src8 << 1), (int)src_stride, (void*)((uintptr_t)dst8 << 1), (int)dst_stride, w, h, 1, bd, 0, filter_x, filter_y, 0,
                  nullptr, nullptr, nullptr, 0);
}
void svt_av1_selfguided_restoration_hip(const uint8_t* dgd8, int32_t width, int32_t height, int32_t dgd_stride, int32_t* flt0, int32_t* flt1,
                                        int32_t flt_stride, int32_t sgr_params_idx, int32_t bit_depth, int32_t highbd) {
    const void* src = highbd ? (const void*)((uintptr_t)dgd8 << 1) : (const void*)dgd8;
    lr_block_host(src, dgd_stride, nullptr, 0, width, height, highbd, bit_depth, 2, nullptr, nullptr, sgr_params_idx, nullptr, flt0, flt1, flt_stride);
}
void svt_apply_selfguided_restoration_hip(const uint8_t* dat8, int32_t width, int32_t height, int32_t stride, int32_t eps, const int32_t* xqd,
                                          uint8_t* dst8, int32_t dst_stride, int32_t* tmpbuf, int32_t bit_depth, int32_t highbd) {
    (void)tmpbuf;
    const void* src = highbd ? (const void*)((uintptr_t)dat8 << 1) : (const void*)dat8;
    void*       dst = highbd ? (void*)((uintptr_t)dst8 << 1) : (void*)dst8;
    lr_block_host(src, stride, dst, dst_stride, width, height, highbd, bit_depth, 1, nullptr, nullptr, eps, xqd, nullptr, nullptr, 0);
}

} // extern "C"

SVT_HIP_DEFINE_WARM(restoration) // (svt_hip_warmup loads this translation unit's code object at encoder initialisation: svt_hip_common.h)
