// tf_subpel.hip -- the temporal filter's sub-pel motion refinement, batched (the caller of the ME_MCTF search).
//
// Reference: tf_subpel_search + svt_check_position (Source/Lib/Codec/temporal_filtering.c:1560-1790), called by tf_{64x64,32x32,16x16,8x8}_sub_pel_search
// (:1793-2250) for every block of every (central picture, reference picture) pair.  A block's refinement is independent of every other block: starting from
// the integer ME vector (<< 3, 1/8 pel) it evaluates the centre, then the half-, quarter- and eighth-pel rings around the running best -- each candidate =
// luma motion compensation (svt_aom_simple_luma_unipred: MV clamped to the picture + border, 8-tap EIGHTTAP_REGULAR or bilinear kernels, the
// svt_av1_[highbd_]convolve_{2d,x,y,2d_copy}_sr_c roundings, inter_prediction.c:311-420, 737-790) + the block's variance against the source
// (svt_aom_varianceWxH_c / svt_aom_highbd_10_varianceWxH_c on every (1 << sub_sampling_shift)-th row) -- with the early exits of svt_check_position.
//
// Mapping: ONE WAVE per block (four blocks per workgroup).  The candidate loop is sequential and data dependent, but wave-uniform: the 64 lanes split the
// block's pixels, the distortion is a wave reduction, the accept / early-exit decisions are scalar.  The horizontally filtered intermediate of the 2-D
// case ((rows + 7) x W int16, rounded exactly as the reference's im_block) lives in the wave's private LDS slice; nothing is written to memory but the
// winner.  Only the prediction rows the variance reads are produced (the reference predicts all rows of a non-centre candidate and then reads every
// other one when sub-sampling is on).
#include "svt_hip_common.h"
#include "../../include/svtav1_hip.h"
#include "interp_core.h"

namespace {


// distortion of one candidate MV (1/8 pel) for the block; every lane returns the same value.  NT = 8: the regular kernels (taps 0..7 around x - 3);
// NT = 2: bilinear, whose only non-zero taps are 3 and 4 (x, x + 1)
template <typename PIX, int NT>
__device__ unsigned long long candidate_dist(const SvtHipTfSubpelParams& P, const SvtHipTfSubpelDesc& d, const PIX* __restrict__ src, const PIX* __restrict__ refy,
                                             const int mvx, const int mvy, const int pss, uint32_t* __restrict__ imp, const int l) {
    constexpr bool HBD = sizeof(PIX) == 2;
    const int W = d.bsize, ss = P.subsampling_shift, bd = HBD ? P.bit_depth : 8;
    const int nrows = W >> ss;                  // prediction rows the variance reads
    const int rstep = pss ? 1 : (1 << ss);      // ... every rstep-th row of this candidate's prediction
    const int rmul  = 1 << pss;                 // the centre is predicted with the reference stride doubled (tf_inter_predictor: src_stride << shift)
    const int hpred = pss ? nrows : W;          // block height svt_inter_predictor sees (selects the 4-tap kernel when <= 4)
    // clamp_mv_to_umv_border_sb (enc_inter_prediction.c:30-50) in 1/16 pel
    const int bmi = W >> 2, mirow = d.pu_y >> 2, micol = d.pu_x >> 2;
    const int to_top = -((mirow * 4) * 8), to_bottom = (((int)P.mi_rows - bmi - mirow) * 4) * 8, to_left = -((micol * 4) * 8), to_right = (((int)P.mi_cols - bmi - micol) * 4) * 8;
    const int spel_l = (4 + W) << 4, spel_r = spel_l - 16;
    int row = (int16_t)(mvy * 2), col = (int16_t)(mvx * 2);
    const int min_col = to_left * 2 - spel_l, max_col = to_right * 2 + spel_r, min_row = to_top * 2 - spel_l, max_row = to_bottom * 2 + spel_r;
    col = col < min_col ? min_col : (col > max_col ? max_col : col);
    row = row < min_row ? min_row : (row > max_row ? max_row : row);
    col = (int16_t)col; row = (int16_t)row;
    const int  sx = col & 15, sy = row & 15;
    const long rs = (long)P.ref_stride;
    const PIX* p0 = refy + P.ref_org_x + (long)P.ref_org_y * rs + (d.pu_x + (col >> 4)) + (long)(d.pu_y + (row >> 4)) * rs;
    const long rsm = rs * rmul; // row stride between consecutive rows of this candidate's prediction
    const int mx = (1 << bd) - 1;
    const PackedTaps tx = kTaps.t[NT == 2 ? 2 : 0][sx], ty = kTaps.t[NT == 2 ? 2 : (hpred <= 4 ? 1 : 0)][sy]; // (W >= 8: never the 4-tap kernel horizontally)
    const int lw = 31 - __clz(W); // log2 W
    int      sum_l = 0; // per lane: <= 64 samples of |diff| <= 1023 -> both fit 32 bits
    uint32_t sse_l = 0;
    const uint32_t sstep = (uint32_t)d.src_stride << ss;
    auto finish = [&](const int i, int px) {
        px = px < 0 ? 0 : (px > mx ? mx : px);
        const int diff = px - (int)src[(uint32_t)(i >> lw) * sstep + (uint32_t)(i & (W - 1))];
        sum_l += diff;
        sse_l += (uint32_t)(diff * diff);
    };
    predict_rows<PIX, NT>(p0, rsm, W, nrows, rstep, sx, sy, tx, ty, bd, imp, l, finish);
    const long long          sum = wave_sum_i32(sum_l);        // |total| <= 4096 * 1023
    const unsigned long long sse = wave_sum_u32_wide(sse_l);   // a row of 16 lanes: <= 1024 * 1023^2
    const int                ln  = 2 * lw - ss; // log2(W * nrows): the reference's division by the sample count is a shift of a non-negative square
    unsigned long long var;
    if (!HBD) {
        const uint32_t s32 = (uint32_t)sse;
        const int      su  = (int)sum;
        var = (uint32_t)(s32 - (uint32_t)(((long long)su * su) >> ln)); // svt_aom_varianceWxH_c (variance.c:300-306)
    } else { // highbd_10_variance (svt_psnr.c:160-177)
        const uint32_t s32 = (uint32_t)((sse + 8) >> 4);
        const int      su  = (int)((sum + 2) >> 2);
        const long long v  = (long long)s32 - (((long long)su * su) >> ln);
        var = v >= 0 ? (uint32_t)v : 0;
    }
    return var << ss;
}

constexpr int kSlice = 36 * 64; // dwords

template <typename PIX>
__global__ __launch_bounds__(256) void tf_subpel_kernel(const SvtHipTfSubpelParams P, const PIX* __restrict__ src_base, const PIX* __restrict__ ref_base,
                                                        const SvtHipTfSubpelDesc* __restrict__ descs, const uint32_t n, SvtHipTfSubpelResult* __restrict__ out,
                                                        const uint32_t per_xcd) {
    HIP_DYNAMIC_SHARED(uint32_t, smem)
    const int      l = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t wg = per_xcd ? (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3) : blockIdx.x; // XCD-aware placement (svt_hip_common.h: xcd_band_per)
    const uint32_t item = wg * 4 + (uint32_t)wv;
    if (item >= n) return;
    uint32_t* im = smem + wv * kSlice; // the wave's private intermediate: (rows + 7 + 1) / 2 row pairs x 64 columns
    const SvtHipTfSubpelDesc d = descs[item];
    if (d.bsize == 0) return; // a block its caller decided not to search (tf_picture.hip): no result is written
    const PIX* src  = src_base + d.src_off;
    const PIX* refy = ref_base + d.ref_off;
    const int  hbd = sizeof(PIX) == 2, ss = P.subsampling_shift;
    unsigned long long best = 0x7fffffffull; // INT_MAX: the callers' starting distortion (temporal_filtering.c:1866, :1980, :2117)
    int16_t bx = d.mv_x, by = d.mv_y;
    const int modes[4] = {P.half_pel_mode, P.half_pel_mode, P.quarter_pel_mode, P.eight_pel_mode};
#pragma unroll 1
    for (int ring = 0; ring < 4; ring++) { // centre, half, quarter, eighth (tf_subpel_search :1670-1790)
        if (ring && !modes[ring]) continue;
        const int16_t base_x = bx, base_y = by;
        const int     st = ring == 0 ? 0 : (8 >> ring);
#pragma unroll 1
        for (int c = 0; c < (ring ? 9 : 1); c++) {
            const int i = ring ? (c / 3 - 1) * st : 0, j = ring ? (c % 3 - 1) * st : 0; // xd outer, yd inner
            if (ring && i == 0 && j == 0) continue;
            // svt_check_position (:1561-1664)
            if (modes[ring] >= 2 && i != 0 && j != 0) continue;
            if (best == 0) continue;
            if (P.early_exit_th && best < (((unsigned long long)(d.bsize * d.bsize) * P.early_exit_th) << hbd)) continue;
            const int16_t cx = (int16_t)(base_x + i), cy = (int16_t)(base_y + j);
            const int                pss  = (i == 0 && j == 0) ? ss : 0;
            const unsigned long long dist = d.bilinear ? candidate_dist<PIX, 2>(P, d, src, refy, cx, cy, pss, im, l) : candidate_dist<PIX, 8>(P, d, src, refy, cx, cy, pss, im, l);
            if (dist < best) { best = dist; bx = cx; by = cy; }
        }
    }
    if (l == 0) {
        SvtHipTfSubpelResult r;
        r.dist = best; r.mv_x = bx; r.mv_y = by; r.pad = 0;
        out[item] = r;
    }
}

// ---- final motion compensation (tf_{64x64,32x32,16x16,8x8}_inter_prediction, temporal_filtering.c:2256-2620): one wave per (block, plane) ------------------------
// svt_aom_inter_prediction's uni-directional SIMPLE_TRANSLATION path with MULTITAP_SHARP kernels (the 4-tap regular kernel for a chroma dimension <= 4), the MV
// clamped per plane, the chroma block at ((pu >> 3) << 3) / 2; the prediction lands in a picture-sized plane at the block's position (what
// svt_hip_tf_filter_frame reads).
// (n_dev: the descriptor count lives in device memory -- a list another kernel just appended to; the waves then walk the list with a grid stride instead of one wave per slot)
template <typename PIX>
__global__ __launch_bounds__(256) void tf_mc_kernel(const SvtHipTfSubpelParams P, const SvtHipTfMcPlanes PL, const SvtHipTfMcDesc* __restrict__ descs, const uint32_t n_host,
                                                    const uint32_t* __restrict__ n_dev, const int chroma) {
    HIP_DYNAMIC_SHARED(uint32_t, smem)
    const int      l = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t n = n_dev ? *n_dev : n_host;
    uint32_t*      im = smem + wv * kSlice;
  for (uint32_t item = blockIdx.x * 4 + (uint32_t)wv; item < 3 * n; item += gridDim.x * 4) {
    const uint32_t blk = item / 3;
    const int      pl = (int)(item - blk * 3);
    if (pl && !chroma) continue;
    const SvtHipTfMcDesc d  = descs[blk];
    if (d.bsize == 0) continue; // an unused slot of a descriptor table
    const int ss = pl > 0, W = d.bsize >> ss, bmi = d.bsize >> 2, bd = sizeof(PIX) == 2 ? P.bit_depth : 8, mx = (1 << bd) - 1;
    const int mirow = d.pu_y >> 2, micol = d.pu_x >> 2; // (the MacroBlockD edges are the luma block's, :2318-2324)
    const int to_top = -((mirow * 4) * 8), to_bottom = (((int)P.mi_rows - bmi - mirow) * 4) * 8, to_left = -((micol * 4) * 8), to_right = (((int)P.mi_cols - bmi - micol) * 4) * 8;
    const int spel_l = (4 + W) << 4, spel_r = spel_l - 16, sc = 1 << (1 - ss); // clamp_mv_to_umv_border_sb with the plane's sub-sampling (enc_inter_prediction.c:30-50)
    int row = (int16_t)(d.mv_y * sc), col = (int16_t)(d.mv_x * sc);
    const int min_col = to_left * sc - spel_l, max_col = to_right * sc + spel_r, min_row = to_top * sc - spel_l, max_row = to_bottom * sc + spel_r;
    col = col < min_col ? min_col : (col > max_col ? max_col : col);
    row = row < min_row ? min_row : (row > max_row ? max_row : row);
    col = (int16_t)col; row = (int16_t)row;
    const int  sx = col & 15, sy = row & 15;
    const int  ox = ss ? ((d.pu_x >> 3) << 3) / 2 : d.pu_x, oy = ss ? ((d.pu_y >> 3) << 3) / 2 : d.pu_y;
    const long rs = (long)PL.ref_stride[pl];
    // (ref_off / pred_off are read from the descriptor in memory: indexing the register copy `d` with the runtime plane index put the whole struct in scratch)
    const PIX* p0 = (const PIX*)PL.ref[pl] + descs[blk].ref_off[pl] + (P.ref_org_x >> ss) + (long)(P.ref_org_y >> ss) * rs + ox + (col >> 4) + (long)(oy + (row >> 4)) * rs;
    PIX*       out = (PIX*)PL.pred[pl] + descs[blk].pred_off[pl] + ox + (size_t)oy * PL.pred_stride[pl];
    const uint32_t ps = PL.pred_stride[pl];
    const int      kind = W <= 4 ? 1 : 3, lw = 31 - __clz(W);
    const PackedTaps tx = kTaps.t[kind][sx], ty = kTaps.t[kind][sy];
    predict_rows<PIX, 8>(p0, rs, W, W, 1, sx, sy, tx, ty, bd, im, l, [&](const int i, int px) {
        px = px < 0 ? 0 : (px > mx ? mx : px);
        out[(uint32_t)(i >> lw) * ps + (uint32_t)(i & (W - 1))] = (PIX)px;
    });
    __builtin_amdgcn_wave_barrier(); // (the wave's LDS slice is rewritten by its next item)
  }
}

} // namespace

extern "C" void svt_hip_tf_subpel_search_batch(const SvtHipTfSubpelParams* params, const void* src_base, const void* ref_base, const SvtHipTfSubpelDesc* descs, uint32_t n,
                                               SvtHipTfSubpelResult* results, void* stream) {
    svthip::ensure_device();
    if (n == 0) return;
    if (params->bit_depth != 8 && params->bit_depth != 10) {
        fprintf(stderr, "libsvtav1_hip: svt_hip_tf_subpel_search_batch: bit depth %d (8 and 10 are what svt_aom_mefn_ptr[].vf / vf_hbd_10 cover)\n", params->bit_depth);
        abort();
    }
    const size_t shm = (size_t)4 * kSlice * 4;
    static const bool xcd_off = [] { const char* e = getenv("SVT_HIP_TF_XCD"); return e && *e == '0'; }(); // (A/B measurements)
    const uint32_t n_wg = (n + 3) / 4, per = xcd_off ? 0 : svthip::xcd_band_per(n_wg), grid = per ? 8 * per : n_wg;
    if (params->bit_depth > 8)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(tf_subpel_kernel<uint16_t>), dim3(grid), dim3(256), shm, (hipStream_t)stream, *params, (const uint16_t*)src_base,
                           (const uint16_t*)ref_base, descs, n, results, per);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(tf_subpel_kernel<uint8_t>), dim3(grid), dim3(256), shm, (hipStream_t)stream, *params, (const uint8_t*)src_base,
                           (const uint8_t*)ref_base, descs, n, results, per);
    SVT_LAUNCH_CHECK();
}

static void tf_mc_launch(const SvtHipTfSubpelParams* params, const SvtHipTfMcPlanes* planes, const SvtHipTfMcDesc* descs, uint32_t n, const uint32_t* n_dev, int chroma,
                         void* stream);
extern "C" void svt_hip_tf_inter_pred_batch(const SvtHipTfSubpelParams* params, const SvtHipTfMcPlanes* planes, const SvtHipTfMcDesc* descs, uint32_t n, int chroma,
                                            void* stream) {
    tf_mc_launch(params, planes, descs, n, nullptr, chroma, stream);
}
// the same over a list whose length another kernel left in device memory (at most max_n descriptors)
extern "C" void svt_hip_tf_inter_pred_list(const SvtHipTfSubpelParams* params, const SvtHipTfMcPlanes* planes, const SvtHipTfMcDesc* descs, uint32_t max_n, const uint32_t* n_dev,
                                           int chroma, void* stream) {
    tf_mc_launch(params, planes, descs, max_n, n_dev, chroma, stream);
}
static void tf_mc_launch(const SvtHipTfSubpelParams* params, const SvtHipTfMcPlanes* planes, const SvtHipTfMcDesc* descs, uint32_t n, const uint32_t* n_dev, int chroma,
                         void* stream) {
    svthip::ensure_device();
    if (n == 0) return;
    if (params->bit_depth != 8 && params->bit_depth != 10) {
        fprintf(stderr, "libsvtav1_hip: svt_hip_tf_inter_pred_batch: bit depth %d\n", params->bit_depth);
        abort();
    }
    const size_t   shm = (size_t)4 * kSlice * 4;
    const uint32_t items = n * 3, wgs = (items + 3) / 4, grid = n_dev && wgs > 4096 ? 4096 : wgs; // (a device-side count: enough waves to fill the chip, then a stride)
    if (params->bit_depth > 8)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(tf_mc_kernel<uint16_t>), dim3(grid), dim3(256), shm, (hipStream_t)stream, *params, *planes, descs, n, n_dev, chroma);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(tf_mc_kernel<uint8_t>), dim3(grid), dim3(256), shm, (hipStream_t)stream, *params, *planes, descs, n, n_dev, chroma);
    SVT_LAUNCH_CHECK();
}

// Host-pointer form of the refinement batch (what a seam inside temporal_filtering.c calls once per (central picture, reference picture) pair): src_buf / ref_buf
// = the two pictures' whole padded luma buffers (src_samples / ref_samples samples), descs / results host arrays; the descs' offsets are relative to the buffers.
extern "C" int svt_hip_tf_subpel_search_host(const SvtHipTfSubpelParams* params, const void* src_buf, size_t src_samples, const void* ref_buf, size_t ref_samples,
                                              const SvtHipTfSubpelDesc* descs, uint32_t n, SvtHipTfSubpelResult* results) {
    SVT_HIP_ENTRY_TRY
    svthip::ensure_device();
    if (n == 0) return 0;
    const size_t px = params->bit_depth > 8 ? 2 : 1, db = (size_t)n * sizeof(SvtHipTfSubpelDesc), rb = (size_t)n * sizeof(SvtHipTfSubpelResult);
    svthip::HostCallLease lease; // (a pooled arena: see svt_hip_common.h)
    svthip::HostCall& c = *lease;
    c.begin();
    c.reserve((src_samples + ref_samples) * px + db + rb + 8192, (src_samples + ref_samples) * px + db + rb + 8192);
    void* d_src = c.dalloc(src_samples * px);
    void* d_ref = c.dalloc(ref_samples * px);
    SvtHipTfSubpelDesc*   d_d = (SvtHipTfSubpelDesc*)c.dalloc(db);
    SvtHipTfSubpelResult* d_r = (SvtHipTfSubpelResult*)c.dalloc(rb);
    c.up(d_src, src_buf, src_samples * px);
    c.up(d_ref, ref_buf, ref_samples * px);
    c.up(d_d, descs, db);
    svt_hip_tf_subpel_search_batch(params, d_src, d_ref, d_d, n, d_r, c.stream);
    c.down(results, d_r, rb);
    return 0;
    SVT_HIP_ENTRY_CATCH(SVT_HIP_E_DEVICE)
}

SVT_HIP_DEFINE_WARM(tf_subpel) // (svt_hip_warmup loads this translation unit's code object at encoder initialisation: svt_hip_common.h)
