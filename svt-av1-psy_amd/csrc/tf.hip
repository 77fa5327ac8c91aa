// tf.hip -- the temporal filter's pixel kernels (the DSP part of Codec/temporal_filtering.c; all fixed point):
//   * plane-wise non-local-means accumulation, with motion (svt_av1_apply_temporal_filter_planewise_medium{,_hbd}_c, :1029-1400) and without
//     (svt_av1_apply_zz_based_temporal_filter_planewise_medium{,_hbd}_c, :819-1013);
//   * central-picture initialisation (svt_aom_apply_filtering_central{,_highbd}_c, :350-425) and normalisation
//     (svt_aom_get_final_filtered_pixels_c, :2608-2672), fused with the accumulation in the frame kernel;
//   * noise estimate (svt_estimate_noise{,_highbd}_fp16_c, :3847-3920).
//
// The filter weight is uniform over a 16x16 luma quadrant (and the co-located chroma quadrants) of a 32x32 block: it depends on the
// quadrant's squared error between the central picture and the motion-compensated prediction, on the block's ME error and on the MV
// length.  So one WAVE owns one quadrant: the squared-error sums are wave reductions, the weight is scalar arithmetic, and the only
// per-pixel work is acc += w * pred.  The frame kernel keeps acc in registers across all references and writes the normalised pixel
// directly: HBM traffic is (n_refs + 2) samples per pixel instead of the reference's per-reference read-modify-write of a u32 accumulator
// and a u16 counter (21 B per 8-bit 4:2:0 pixel and reference).
#include "svt_hip_common.h"
#include "../../include/svtav1_hip.h"

#include "tf_tables_gen.h"

namespace {

constexpr uint32_t TF_WEIGHT_SCALE = 1000; // temporal_filtering.h:45 (= TF_PLANEWISE_FILTER_WEIGHT_SCALE, :40)
constexpr uint32_t TF_BALANCE      = 10;   // TF_WINDOW_BLOCK_BALANCE_WEIGHT, :49

template <typename PIX> struct __attribute__((packed, aligned(1))) Px4 { PIX v[4]; };
// the same four samples as raw dwords (kept packed in registers until every load of the kernel has been issued)
template <typename PIX> struct __attribute__((packed, aligned(1))) Raw4 { uint32_t w[sizeof(PIX)]; };
template <typename PIX> __device__ inline uint32_t raw_px(const Raw4<PIX>& r, const int i) {
    return sizeof(PIX) == 1 ? (r.w[0] >> (8 * i)) & 0xff : (r.w[i >> 1] >> (16 * (i & 1))) & 0xffff;
}

__device__ inline uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += (uint32_t)__shfl_xor((int)v, m);
    return v;
}
// sum over each row of 16 lanes, left in every lane of the row (4 VALU DPP steps; the wave total is the sum of lanes 0, 16, 32, 48)
__device__ inline uint32_t row16_sum(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false);  // quad_perm [1,0,3,2]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false);  // quad_perm [2,3,0,1]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xf, 0xf, false); // row_ror:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false); // row_ror:8
    return v;
}
// floor(n / d) for n < 2^28, 0 < d < 2^16 and a quotient < 2^16, given rcp = 1.0f / d: the float estimate is within 1 of the quotient
__device__ inline uint32_t div_by(const uint32_t n, const uint32_t d, const float rcp) {
    uint32_t q = (uint32_t)((float)n * rcp);
    int32_t  r = (int32_t)(n - q * d);
    if (r < 0) { q--; r += (int32_t)d; }
    if (r >= (int32_t)d) q++;
    return q;
}
__device__ inline uint32_t sqrt_fast(const uint32_t x) { // temporal_filtering.c:741-750; svt_log2f = floor(log2)
    if (x > 15) {
        const int log2_half = (31 - __clz((int)x)) >> 1;
        return kTfSqrtQ16[x >> (2 * log2_half - 2)] >> (17 - log2_half);
    }
    return kTfSqrtQ16[x] >> 16;
}
// quadrant q's ME terms: block error in Q8 and the distance factor (>= 1.0 in Q8)
__device__ inline void block_terms(const SvtHipTfParams& P, const SvtHipTfBlock& B, const int q, const bool hbd, uint32_t& blk_err, uint32_t& d_factor) {
    const int      k = B.split ? q : 0;
    const int32_t  col = B.mv_x[k], row = B.mv_y[k];
    const uint32_t th0 = ((uint32_t)P.tf_mv_dist_th << 16) / 10, dist_th = th0 > (1u << 16) ? th0 : (1u << 16);
    const uint32_t d = (sqrt_fast(((uint32_t)(col * col + row * row)) << 8) << 12) / (dist_th >> 8);
    d_factor = d > (1u << 8) ? d : (1u << 8);
    blk_err  = B.split ? (uint32_t)(B.block_error[q] >> (hbd ? 4 : 0)) : (uint32_t)(B.block_error[0] >> (hbd ? 6 : 2));
}
// weight of one plane's quadrant from its window error (Q8); `decay` already doubled for unsplit blocks
__device__ inline uint32_t motion_weight(const uint32_t win, const uint32_t blk_err, const uint32_t d_factor, const uint32_t decay) {
    const uint32_t combined = (win * TF_BALANCE + blk_err) / (TF_BALANCE + 1);
    const uint32_t avg = (combined >> 3) * (d_factor >> 3); // 32-bit product, as the reference
    const uint32_t den = (decay >> 10) > 1 ? (decay >> 10) : 1;
    const uint32_t sd  = avg / den;
    return (uint32_t)(kTfExpQ16[sd < 7 * 16 ? sd : 7 * 16] * (int32_t)TF_WEIGHT_SCALE) >> 16;
}
__device__ inline uint32_t zz_weight(const uint32_t blk_err, const uint32_t decay) {
    const uint32_t den = (decay >> 10) > 1 ? (decay >> 10) : 1;
    const uint32_t sd  = (blk_err << 2) / den;
    return (uint32_t)(kTfExpQ16[sd < 7 * 16 ? sd : 7 * 16] * (int32_t)TF_WEIGHT_SCALE) >> 17;
}
__device__ inline uint32_t window_error(uint32_t sum, const int shift, const uint32_t qw, const uint32_t qh) {
    sum >>= shift;
    return (((sum << 4) / qw) << 4) / qh;
}

// ---- accumulate form: one block (bw x bh luma, any even size) of ONE reference into caller-owned accum / count, addressed with the
// prediction's stride (temporal_filtering.c:1136).  One workgroup, wave q = quadrant q; this is what the four RTCD symbols compute.
struct AccumArgs {
    const void *src[3], *pre[3];
    uint32_t*   accum[3];
    uint16_t*   count[3];
    int         src_stride[2], pre_stride[2];
    unsigned    bw, bh;
    int         zz;
};
template <typename PIX>
__global__ __launch_bounds__(256) void tf_accum_kernel(const SvtHipTfParams P, const SvtHipTfBlock B, const AccumArgs A) {
    const bool hbd = sizeof(PIX) == 2;
    const int  q = threadIdx.x >> 6, lane = threadIdx.x & 63, shift = hbd ? (P.encoder_bit_depth - 8) * 2 : 0;
    uint32_t   blk_err, d_factor, luma_win = 0;
    block_terms(P, B, q, hbd, blk_err, d_factor);
    for (int c = 0; c < (P.tf_chroma ? 3 : 1); c++) {
        const unsigned w = c ? A.bw >> P.ss_x : A.bw, h = c ? A.bh >> P.ss_y : A.bh, qw = w >> 1, qh = h >> 1;
        const int      ss = A.src_stride[c > 0], ps = A.pre_stride[c > 0];
        const PIX*     src = (const PIX*)A.src[c];
        const PIX*     pre = (const PIX*)A.pre[c];
        const int      x0 = (q & 1) * (int)w / 2, y0 = (q >> 1) * (int)h / 2;
        uint32_t       weight;
        if (A.zz) {
            weight = zz_weight(blk_err, P.tf_decay_factor_fp16[c]);
        } else {
            uint32_t sum = 0;
            for (unsigned i = lane; i < qw * qh; i += 64) {
                const unsigned y = i / qw, x = i - y * qw;
                const int d = (int)src[(size_t)(y + (q >> 1) * qh) * ss + x + (q & 1) * qw] - (int)pre[(size_t)(y + (q >> 1) * qh) * ps + x + (q & 1) * qw];
                sum += (uint32_t)(d * d);
            }
            uint32_t win = window_error(wave_sum(sum), shift, qw, qh);
            if (c) win = (win * 5 + luma_win) / 6;
            else luma_win = win;
            weight = motion_weight(win, blk_err, d_factor, B.split ? P.tf_decay_factor_fp16[c] : P.tf_decay_factor_fp16[c] << 1);
        }
        for (unsigned i = lane; i < (w / 2) * (h / 2); i += 64) {
            const unsigned y = i / (w / 2), x = i - y * (w / 2);
            const size_t   k = (size_t)(y + y0) * ps + x + x0;
            A.count[c][k] = (uint16_t)(A.count[c][k] + weight);
            A.accum[c][k] += weight * pre[k];
        }
    }
}

// ---- frame form: central init + every reference + normalisation for all 32x32 blocks of a picture
struct FrameArgs {
    SvtHipTfPlanes central, out, preds[SVT_HIP_TF_MAX_REFS];
    const SvtHipTfBlock* blocks; // [n_refs][nby][nbx]
    uint32_t n_refs, nbx, nby;
    // more reference frames than one launch takes: the accumulators travel between launches (svt_hip_tf_filter_frame_chunked)
    uint32_t* acc[3];      // per pixel
    uint16_t* cnt[3];      // per lane group: written at the group's first pixel
    uint32_t  acc_stride[2], chain; // chain bit 0: start from acc / cnt instead of the central picture; bit 1: store acc / cnt instead of the normalised pixels
};
// The weights are uniform per (reference, plane, quadrant) and cost ~300 instructions of scalar-looking arithmetic each (integer divisions,
// the square-root and exponential tables).  Evaluated per reference in every lane they were 15x the pixel work, so the kernel is staged:
//   1. issue every load of the workgroup (central + all predictions + the block records) before the first use: one memory round trip;
//   2. squared errors per (reference, plane): packed dot products on the raw dwords (sum a^2 + sum b^2 - 2 sum ab), a 4-step DPP row
//      reduction, row sums handed over through LDS;
//   3. lane (plane * 16 + ref) turns its sum into a weight -- all references and planes at once, one pass of the expensive arithmetic;
//   4. broadcast the weights and accumulate, normalise (exact float-reciprocal division), store.
// The kernel is VALU-issue bound (about 600 wave instructions per 32x32 block at 6 references), not HBM bound; NR = reference count rounded
// up to even, so that dead references cost nothing.
template <typename PIX> __device__ inline uint32_t raw_dot(const Raw4<PIX>& a, const Raw4<PIX>& b) { // sum of the 4 products
    if (sizeof(PIX) == 1) return __builtin_amdgcn_udot4(a.w[0], b.w[0], 0u, false);
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    us2 x0, y0, x1, y1;
    __builtin_memcpy(&x0, &a.w[0], 4); __builtin_memcpy(&y0, &b.w[0], 4);
    __builtin_memcpy(&x1, &a.w[sizeof(PIX) - 1], 4); __builtin_memcpy(&y1, &b.w[sizeof(PIX) - 1], 4);
    return __builtin_amdgcn_udot2(x1, y1, __builtin_amdgcn_udot2(x0, y0, 0u, false), false);
}
template <typename PIX, int CPX, int NR> // CPX = chroma pixels per lane and plane: 1 (4:2:0), 2 (4:2:2, 4:4:0), 4 (4:4:4), 0 (luma only)
__global__ __launch_bounds__(256) void tf_frame_kernel(const SvtHipTfParams P, const FrameArgs A) {
    constexpr int  CP = CPX ? CPX : 1;
    constexpr bool hbd = sizeof(PIX) == 2;
    __shared__ uint32_t sh_sum[4][3][NR][4]; // [quadrant][plane][reference][row of 16 lanes]
    const int q = threadIdx.x >> 6, lane = threadIdx.x & 63, shift = hbd ? (P.encoder_bit_depth - 8) * 2 : 0;
    // XCD-aware block order: hardware places workgroup b on XCD b % 8; each XCD gets a contiguous raster run of blocks, so the four
    // horizontally adjacent 32-pixel blocks that share a 128-byte line of every plane meet in one L2 instead of four
    const uint32_t nblk = A.nbx * A.nby, per = nblk >> 3, b = blockIdx.x;
    const uint32_t t = (per == 0 || b >= per * 8) ? b : (b & 7) * per + (b >> 3);
    const int by = (int)(t / A.nbx), bx = (int)(t - (uint32_t)by * A.nbx);
    // luma: 4 consecutive pixels of one row of the 16x16 quadrant
    const int ly = by * 32 + (q >> 1) * 16 + (lane >> 2), lx = bx * 32 + (q & 1) * 16 + (lane & 3) * 4;
    // chroma quadrant (16 >> ss_x) x (16 >> ss_y) = 2^lw x 2^lh, CPX consecutive pixels of a row per lane
    const int lw = 4 - P.ss_x, lh = 4 - P.ss_y;
    const int ci = lane * CP, cyq = ci >> lw, cxq = ci - (cyq << lw);
    const int cy = by * (32 >> P.ss_y) + ((q >> 1) << lh) + cyq, cx = bx * (32 >> P.ss_x) + ((q & 1) << lw) + cxq;
    const int n_refs = (int)A.n_refs;

    // 1. loads.  The weight lanes (plane * 16 + ref) fetch their block record in the same round trip; 32-bit sample offsets from uniform
    // plane pointers (SGPR base + VGPR offset form, no 64-bit address pairs to recycle); references beyond n_refs (at most one) re-read the
    // central picture and get weight 0, so there is no branch between the loads.
    const int  wr = lane & 15, wc = lane >> 4; // weight lanes: plane * 16 + reference (up to 12 references x 3 planes)
    const bool wlane = wr < n_refs && wc < (CPX ? 3 : 1);
    SvtHipTfBlock B = {};
    if (wlane) B = A.blocks[((size_t)wr * A.nby + by) * A.nbx + bx];
    Raw4<PIX> rawS, rawY[NR];
    PIX       rawSU[CP], rawSV[CP], rawU[NR][CP], rawV[NR][CP];
    rawS = *(const Raw4<PIX>*)((const PIX*)A.central.y + ((uint32_t)ly * A.central.y_stride + (uint32_t)lx));
    if (CPX) {
        const uint32_t o = (uint32_t)cy * A.central.uv_stride + (uint32_t)cx;
#pragma unroll
        for (int i = 0; i < CP; i++) {
            rawSU[i] = ((const PIX*)A.central.u)[o + i];
            rawSV[i] = ((const PIX*)A.central.v)[o + i];
        }
    }
#pragma unroll
    for (int r = 0; r < NR; r++) {
        const bool     live = r < n_refs;
        const PIX*     ry = (const PIX*)(live ? A.preds[r].y : A.central.y);
        const PIX*     ru = (const PIX*)(live ? A.preds[r].u : A.central.u);
        const PIX*     rv = (const PIX*)(live ? A.preds[r].v : A.central.v);
        const uint32_t sy = live ? A.preds[r].y_stride : A.central.y_stride, sc = live ? A.preds[r].uv_stride : A.central.uv_stride;
        rawY[r] = *(const Raw4<PIX>*)(ry + ((uint32_t)ly * sy + (uint32_t)lx));
        if (CPX) {
            const uint32_t o = (uint32_t)cy * sc + (uint32_t)cx;
#pragma unroll
            for (int i = 0; i < CP; i++) {
                rawU[r][i] = ru[o + i];
                rawV[r][i] = rv[o + i];
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);

    // 2. squared errors per (reference, plane)
    if (!P.use_zz_based_filter) {
        const uint32_t ssq = raw_dot<PIX>(rawS, rawS);
#pragma unroll
        for (int r = 0; r < NR; r++) {
            uint32_t sY = ssq + raw_dot<PIX>(rawY[r], rawY[r]) - 2 * raw_dot<PIX>(rawS, rawY[r]), sU = 0, sV = 0;
            sY = row16_sum(sY);
            if (CPX) {
#pragma unroll
                for (int i = 0; i < CP; i++) {
                    const int du = (int)rawSU[i] - (int)rawU[r][i], dv = (int)rawSV[i] - (int)rawV[r][i];
                    sU += (uint32_t)(du * du);
                    sV += (uint32_t)(dv * dv);
                }
                sU = row16_sum(sU);
                sV = row16_sum(sV);
            }
            if ((lane & 15) == 0) { sh_sum[q][0][r][lane >> 4] = sY; sh_sum[q][1][r][lane >> 4] = sU; sh_sum[q][2][r][lane >> 4] = sV; }
        }
    }
    // 3. lane = plane * 16 + ref: one weight each (same-wave LDS traffic only: program order suffices on the hardware)
    __builtin_amdgcn_wave_barrier();
    uint32_t weight = 0;
    if (wlane) {
        uint32_t blk_err, d_factor;
        block_terms(P, B, q, hbd, blk_err, d_factor);
        if (P.use_zz_based_filter) {
            weight = zz_weight(blk_err, P.tf_decay_factor_fp16[wc]);
        } else {
            const uint32_t* sy = sh_sum[q][0][wr < NR ? wr : 0];
            const uint32_t* sc = sh_sum[q][wc][wr < NR ? wr : 0];
            // window_error with power-of-two quadrant sides: (((sum << 4) / w) << 4) / h
            const uint32_t winY = (((sy[0] + sy[1] + sy[2] + sy[3]) >> shift) << 4 >> 4) << 4 >> 4;
            uint32_t       win  = winY;
            if (wc) win = ((((((sc[0] + sc[1] + sc[2] + sc[3]) >> shift) << 4) >> lw) << 4 >> lh) * 5 + winY) / 6;
            weight = motion_weight(win, blk_err, d_factor, P.tf_decay_factor_fp16[wc] << (B.split ? 0 : 1));
        }
    }
    // 4. accumulate with broadcast weights, normalise, store
    uint32_t accY[4], accU[CP], accV[CP], cntY = TF_WEIGHT_SCALE, cntU = TF_WEIGHT_SCALE, cntV = TF_WEIGHT_SCALE;
    const uint32_t aoy = (uint32_t)ly * A.acc_stride[0] + (uint32_t)lx, aoc = (uint32_t)cy * A.acc_stride[1] + (uint32_t)cx;
    if (A.chain & 1u) { // (workgroup-uniform) a later chunk of the reference frames: the sums so far
#pragma unroll
        for (int i = 0; i < 4; i++) accY[i] = A.acc[0][aoy + i];
        cntY = A.cnt[0][aoy];
        if (CPX) {
#pragma unroll
            for (int i = 0; i < CP; i++) { accU[i] = A.acc[1][aoc + i]; accV[i] = A.acc[2][aoc + i]; }
            cntU = A.cnt[1][aoc]; cntV = A.cnt[2][aoc];
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++) accY[i] = TF_WEIGHT_SCALE * raw_px<PIX>(rawS, i);
#pragma unroll
        for (int i = 0; i < CP; i++) { accU[i] = CPX ? TF_WEIGHT_SCALE * rawSU[i] : 0; accV[i] = CPX ? TF_WEIGHT_SCALE * rawSV[i] : 0; }
    }
#pragma unroll
    for (int r = 0; r < NR; r++) {
        const uint32_t wY = (uint32_t)__shfl((int)weight, r);
        cntY = (uint16_t)(cntY + wY);
#pragma unroll
        for (int i = 0; i < 4; i++) accY[i] += wY * raw_px<PIX>(rawY[r], i);
        if (CPX) {
            const uint32_t wU = (uint32_t)__shfl((int)weight, 16 + r), wV = (uint32_t)__shfl((int)weight, 32 + r);
            cntU = (uint16_t)(cntU + wU);
            cntV = (uint16_t)(cntV + wV);
#pragma unroll
            for (int i = 0; i < CP; i++) { accU[i] += wU * rawU[r][i]; accV[i] += wV * rawV[r][i]; }
        }
    }
    if (A.chain & 2u) { // not the last chunk: hand the sums on
#pragma unroll
        for (int i = 0; i < 4; i++) A.acc[0][aoy + i] = accY[i];
        A.cnt[0][aoy] = (uint16_t)cntY;
        if (CPX) {
#pragma unroll
            for (int i = 0; i < CP; i++) { A.acc[1][aoc + i] = accU[i]; A.acc[2][aoc + i] = accV[i]; }
            A.cnt[1][aoc] = (uint16_t)cntU; A.cnt[2][aoc] = (uint16_t)cntV;
        }
        return;
    }
    Px4<PIX>    o;
    const float rY = 1.0f / (float)cntY;
#pragma unroll
    for (int i = 0; i < 4; i++) o.v[i] = (PIX)div_by(accY[i] + (cntY >> 1), cntY, rY);
    *(Px4<PIX>*)((PIX*)A.out.y + ((uint32_t)ly * A.out.y_stride + (uint32_t)lx)) = o;
    if (CPX) {
        const float rU = 1.0f / (float)cntU, rV = 1.0f / (float)cntV;
#pragma unroll
        for (int i = 0; i < CP; i++) {
            ((PIX*)A.out.u)[(uint32_t)cy * A.out.uv_stride + (uint32_t)(cx + i)] = (PIX)div_by(accU[i] + (cntU >> 1), cntU, rU);
            ((PIX*)A.out.v)[(uint32_t)cy * A.out.uv_stride + (uint32_t)(cx + i)] = (PIX)div_by(accV[i] + (cntV >> 1), cntV, rV);
        }
    }
}

// ---- noise estimate: mean |Laplacian| over the interior pixels whose Sobel magnitude is below the edge threshold.
// Workgroup = 64 columns x 64 rows (wave = 16 rows); a lane walks down its column with a 3x3 window in registers (3 loads per pixel) and
// the workgroup leaves one (sum, num) pair in the workspace; a one-workgroup second kernel adds the pairs and does the fixed-point division.
constexpr int NZ_ROWS = 16;
template <typename PIX>
__global__ __launch_bounds__(256) void noise_kernel(const PIX* __restrict__ src, const int width, const int height, const int stride, const int bd,
                                                    uint32_t* __restrict__ partial /* [blocks][2] */) {
    __shared__ uint32_t sh[4][2];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x = 1 + blockIdx.x * 64 + lane, y0 = 1 + (blockIdx.y * 4 + wv) * NZ_ROWS;
    uint32_t  sum = 0, num = 0;
    if (x < width - 1 && y0 < height - 1) {
        const PIX* p = src + (size_t)(y0 - 1) * stride + x;
        int a0 = p[-1], a1 = p[0], a2 = p[1];
        p += stride;
        int b0 = p[-1], b1 = p[0], b2 = p[1];
        for (int y = y0; y < y0 + NZ_ROWS && y < height - 1; y++) {
            p += stride;
            const int c0 = p[-1], c1 = p[0], c2 = p[1];
            const int gx = (a0 - a2) + (c0 - c2) + 2 * (b0 - b2);
            const int gy = (a0 - c0) + (a2 - c2) + 2 * (a1 - c1);
            int       ga = (gx < 0 ? -gx : gx) + (gy < 0 ? -gy : gy);
            if (sizeof(PIX) == 2) ga = (ga + ((1 << (bd - 8)) >> 1)) >> (bd - 8);
            if (ga < 50) { // EDGE_THRESHOLD
                int v = 4 * b1 - 2 * (b0 + b2 + a1 + c1) + (a0 + a2 + c0 + c2);
                v     = v < 0 ? -v : v;
                if (sizeof(PIX) == 2) v = (v + ((1 << (bd - 8)) >> 1)) >> (bd - 8);
                sum += (uint32_t)v;
                num++;
            }
            a0 = b0; a1 = b1; a2 = b2;
            b0 = c0; b1 = c1; b2 = c2;
        }
    }
    sum = wave_sum(sum);
    num = wave_sum(num);
    if (lane == 0) { sh[wv][0] = sum; sh[wv][1] = num; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const size_t b = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
        partial[2 * b]     = sh[0][0] + sh[1][0] + sh[2][0] + sh[3][0];
        partial[2 * b + 1] = sh[0][1] + sh[1][1] + sh[2][1] + sh[3][1];
    }
}
__global__ __launch_bounds__(256) void noise_finalize_kernel(const uint32_t* __restrict__ partial, const int n, int32_t* __restrict__ out) {
    __shared__ unsigned long long sh[4][2];
    unsigned long long sum = 0, num = 0;
    for (int i = threadIdx.x; i < n; i += 256) { sum += partial[2 * i]; num += partial[2 * i + 1]; }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        sum += ((unsigned long long)(unsigned)__shfl_xor((int)(unsigned)(sum >> 32), m) << 32) | (unsigned)__shfl_xor((int)(unsigned)sum, m);
        num += ((unsigned long long)(unsigned)__shfl_xor((int)(unsigned)(num >> 32), m) << 32) | (unsigned)__shfl_xor((int)(unsigned)num, m);
    }
    if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6][0] = sum; sh[threadIdx.x >> 6][1] = num; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const long long s = (long long)(sh[0][0] + sh[1][0] + sh[2][0] + sh[3][0]), k = (long long)(sh[0][1] + sh[1][1] + sh[2][1] + sh[3][1]);
        out[0] = k < 16 ? -65536 : (int32_t)((s * 82137) / (6 * k)); // SMOOTH_THRESHOLD -> -1.0; SQRT_PI_BY_2_FP16
    }
}
inline dim3 noise_grid(uint32_t width, uint32_t height) {
    return dim3(width > 2 ? (width - 2 + 63) / 64 : 0, height > 2 ? (height - 2 + 4 * NZ_ROWS - 1) / (4 * NZ_ROWS) : 0);
}

template <typename PIX>
void accum_host(const SvtHipTfParams* P, const SvtHipTfBlock* B, const void* const src[3], const int src_stride[2], const void* const pre[3],
                const int pre_stride[2], unsigned bw, unsigned bh, uint32_t* const accum[3], uint16_t* const count[3], int zz) {
    svthip::ensure_device();
    auto&     hc = svthip::host_call();
    hc.begin();
    const int nc = P->tf_chroma ? 3 : 1;
    size_t    total = 0;
    unsigned  w[3], h[3];
    for (int c = 0; c < nc; c++) {
        w[c] = c ? bw >> P->ss_x : bw;
        h[c] = c ? bh >> P->ss_y : bh;
        total += svthip::align_up((size_t)w[c] * h[c] * (2 * sizeof(PIX) + 6), 256) * 4;
    }
    hc.reserve(total, 2 * total); // every upload and download is staged through the pinned arena
    AccumArgs A;
    memset(&A, 0, sizeof(A));
    A.bw = bw; A.bh = bh; A.zz = zz;
    for (int c = 0; c < nc; c++) {
        const size_t n = (size_t)w[c] * h[c];
        PIX*      ds = (PIX*)hc.dalloc(n * sizeof(PIX));
        PIX*      dp = (PIX*)hc.dalloc(n * sizeof(PIX));
        uint32_t* da = (uint32_t*)hc.dalloc(n * 4);
        uint16_t* dc = (uint16_t*)hc.dalloc(n * 2);
        if (!zz) hc.up2d(ds, w[c] * sizeof(PIX), src[c], (size_t)src_stride[c > 0] * sizeof(PIX), w[c] * sizeof(PIX), h[c]);
        hc.up2d(dp, w[c] * sizeof(PIX), pre[c], (size_t)pre_stride[c > 0] * sizeof(PIX), w[c] * sizeof(PIX), h[c]);
        hc.up2d(da, w[c] * 4, accum[c], (size_t)pre_stride[c > 0] * 4, w[c] * 4, h[c]);
        hc.up2d(dc, w[c] * 2, count[c], (size_t)pre_stride[c > 0] * 2, w[c] * 2, h[c]);
        A.src[c] = ds; A.pre[c] = dp; A.accum[c] = da; A.count[c] = dc;
        A.src_stride[c > 0] = (int)w[c]; A.pre_stride[c > 0] = (int)w[c];
    }
    hipLaunchKernelGGL(tf_accum_kernel<PIX>, dim3(1), dim3(256), 0, hc.stream, *P, *B, A);
    SVT_LAUNCH_CHECK();
    for (int c = 0; c < nc; c++) {
        hc.down2d_later(accum[c], (size_t)pre_stride[c > 0] * 4, A.accum[c], w[c] * 4, w[c] * 4, h[c]); // in-place accumulators: ONE commit point
        hc.down2d_later(count[c], (size_t)pre_stride[c > 0] * 2, A.count[c], w[c] * 2, w[c] * 2, h[c]);
    }
    hc.finish();
}

template <typename PIX> int32_t noise_host(const PIX* src, int width, int height, int stride, int bd) {
    svthip::ensure_device();
    auto& hc = svthip::host_call();
    hc.begin();
    const size_t pitch = svthip::align_up((size_t)width * sizeof(PIX), 256);
    const size_t ws = svt_hip_estimate_noise_workspace((uint32_t)width, (uint32_t)height);
    hc.reserve(pitch * height + ws + 1024, pitch * height + 512);
    PIX*     d   = (PIX*)hc.dalloc(pitch * height);
    uint8_t* w   = (uint8_t*)hc.dalloc(ws);
    int32_t* out = (int32_t*)hc.dalloc(sizeof(int32_t));
    hc.up2d(d, pitch, src, (size_t)stride * sizeof(PIX), (size_t)width * sizeof(PIX), height);
    svt_hip_estimate_noise_batch(d, (uint32_t)width, (uint32_t)height, (uint32_t)(pitch / sizeof(PIX)), bd, out, w, hc.stream);
    int32_t res = 0;
    hc.down(&res, out, sizeof(int32_t));
    return res;
}

} // namespace

extern "C" {

size_t svt_hip_estimate_noise_workspace(uint32_t width, uint32_t height) {
    const dim3 g = noise_grid(width, height);
    return (size_t)g.x * g.y * 8 + 8;
}
void svt_hip_estimate_noise_batch(const void* plane, uint32_t width, uint32_t height, uint32_t stride, int bit_depth, int32_t* noise_out, void* workspace,
                                  void* stream) {
    svthip::ensure_device();
    const dim3 grid = noise_grid(width, height);
    const int  n = (int)(grid.x * grid.y);
    if (n) {
        if (bit_depth > 8)
            hipLaunchKernelGGL(noise_kernel<uint16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)plane, (int)width, (int)height, (int)stride,
                               bit_depth, (uint32_t*)workspace);
        else
            hipLaunchKernelGGL(noise_kernel<uint8_t>, grid, dim3(256), 0, (hipStream_t)stream, (const uint8_t*)plane, (int)width, (int)height, (int)stride, 8,
                               (uint32_t*)workspace);
        SVT_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(noise_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const uint32_t*)workspace, n, noise_out);
    SVT_LAUNCH_CHECK();
}
int32_t svt_estimate_noise_fp16_hip(const uint8_t* src, uint16_t width, uint16_t height, uint16_t stride_y) { return noise_host(src, width, height, stride_y, 8); }
int32_t svt_estimate_noise_highbd_fp16_hip(const uint16_t* src, int width, int height, int stride, int bd) { return noise_host(src, width, height, stride, bd); }

static void tf_frame_launch(const SvtHipTfParams* params, const SvtHipTfPlanes* central, const SvtHipTfPlanes* preds, uint32_t n_refs, const SvtHipTfBlock* blocks,
                            uint32_t nbx, uint32_t nby, const SvtHipTfPlanes* out, void* state, uint32_t chain, hipStream_t st) {
    FrameArgs A;
    memset(&A, 0, sizeof(A));
    A.central = *central; A.out = *out; A.blocks = blocks; A.n_refs = n_refs; A.nbx = nbx; A.nby = nby; A.chain = chain;
    for (uint32_t r = 0; r < n_refs; r++) A.preds[r] = preds[r];
    if (state) { // [acc y][acc u][acc v][cnt y][cnt u][cnt v], luma pitch 32 nbx, chroma pitch (32 nbx) >> ss_x
        const size_t W = (size_t)nbx * 32, H = (size_t)nby * 32, cw = W >> params->ss_x, chh = H >> params->ss_y;
        uint8_t* p = (uint8_t*)state;
        A.acc[0] = (uint32_t*)p; p += W * H * 4;
        A.acc[1] = (uint32_t*)p; p += cw * chh * 4;
        A.acc[2] = (uint32_t*)p; p += cw * chh * 4;
        A.cnt[0] = (uint16_t*)p; p += W * H * 2;
        A.cnt[1] = (uint16_t*)p; p += cw * chh * 2;
        A.cnt[2] = (uint16_t*)p;
        A.acc_stride[0] = (uint32_t)W; A.acc_stride[1] = (uint32_t)cw;
    }
    const int  cpx = !params->tf_chroma ? 0 : (16 >> params->ss_x) * (16 >> params->ss_y) / 64;
    const bool hbd = params->encoder_bit_depth > 8;
    const int  nr  = n_refs <= 2 ? 2 : (int)((n_refs + 1) & ~1u);
    const dim3 grid(nbx * nby), blk(256);
#define LAUNCH(PIX, C, NR) hipLaunchKernelGGL((tf_frame_kernel<PIX, C, NR>), grid, blk, 0, st, *params, A)
#define BY_NR(PIX, C) do { if (nr == 2) LAUNCH(PIX, C, 2); else if (nr == 4) LAUNCH(PIX, C, 4); else if (nr == 6) LAUNCH(PIX, C, 6); else if (nr == 8) LAUNCH(PIX, C, 8); \
                           else if (nr == 10) LAUNCH(PIX, C, 10); else LAUNCH(PIX, C, 12); } while (0)
#define BY_C(PIX) do { if (cpx == 0) BY_NR(PIX, 0); else if (cpx == 1) BY_NR(PIX, 1); else if (cpx == 2) BY_NR(PIX, 2); else BY_NR(PIX, 4); } while (0)
    if (hbd) BY_C(uint16_t); else BY_C(uint8_t);
#undef BY_C
#undef BY_NR
#undef LAUNCH
    SVT_LAUNCH_CHECK();
}

void svt_hip_tf_filter_frame(const SvtHipTfParams* params, const SvtHipTfPlanes* central, const SvtHipTfPlanes* preds, uint32_t n_refs,
                             const SvtHipTfBlock* blocks, uint32_t nbx, uint32_t nby, const SvtHipTfPlanes* out, void* stream) {
    svthip::ensure_device();
    if (n_refs > SVT_HIP_TF_MAX_REFS) { fprintf(stderr, "libsvtav1_hip: svt_hip_tf_filter_frame: n_refs %u > %d\n", n_refs, SVT_HIP_TF_MAX_REFS); abort(); }
    if (!nbx || !nby) return;
    tf_frame_launch(params, central, preds, n_refs, blocks, nbx, nby, out, nullptr, 0, (hipStream_t)stream);
}

size_t svt_hip_tf_filter_frame_workspace(const SvtHipTfParams* params, uint32_t nbx, uint32_t nby) {
    const size_t W = (size_t)nbx * 32, H = (size_t)nby * 32, cw = W >> params->ss_x, chh = H >> params->ss_y;
    return (W * H + 2 * cw * chh) * 6 + 256;
}
void svt_hip_tf_filter_frame_chunked(const SvtHipTfParams* params, const SvtHipTfPlanes* central, const SvtHipTfPlanes* preds, uint32_t n_refs,
                                     const SvtHipTfBlock* blocks, uint32_t nbx, uint32_t nby, const SvtHipTfPlanes* out, void* workspace, void* stream) {
    svthip::ensure_device();
    if (n_refs > SVT_HIP_TF_MAX_FRAMES) { fprintf(stderr, "libsvtav1_hip: svt_hip_tf_filter_frame_chunked: n_refs %u > %d\n", n_refs, SVT_HIP_TF_MAX_FRAMES); abort(); }
    if (!nbx || !nby) return;
    if (n_refs <= SVT_HIP_TF_MAX_REFS) { tf_frame_launch(params, central, preds, n_refs, blocks, nbx, nby, out, nullptr, 0, (hipStream_t)stream); return; }
    for (uint32_t r0 = 0; r0 < n_refs; r0 += SVT_HIP_TF_MAX_REFS) { // the sums of the frames so far travel through the workspace; only the last launch writes pixels
        const uint32_t n = n_refs - r0 < SVT_HIP_TF_MAX_REFS ? n_refs - r0 : SVT_HIP_TF_MAX_REFS;
        const uint32_t chain = (r0 ? 1u : 0u) | (r0 + n < n_refs ? 2u : 0u);
        tf_frame_launch(params, central, preds + r0, n, blocks + (size_t)r0 * nbx * nby, nbx, nby, out, workspace, chain, (hipStream_t)stream);
    }
}

void svt_av1_apply_temporal_filter_planewise_medium_hip(const SvtHipTfParams* params, const SvtHipTfBlock* block, const uint8_t* y_src, int y_src_stride,
                                                        const uint8_t* y_pre, int y_pre_stride, const uint8_t* u_src, const uint8_t* v_src, int uv_src_stride,
                                                        const uint8_t* u_pre, const uint8_t* v_pre, int uv_pre_stride, unsigned int block_width,
                                                        unsigned int block_height, int ss_x, int ss_y, uint32_t* y_accum, uint16_t* y_count, uint32_t* u_accum,
                                                        uint16_t* u_count, uint32_t* v_accum, uint16_t* v_count) {
    SvtHipTfParams P = *params;
    P.ss_x = (uint8_t)ss_x; P.ss_y = (uint8_t)ss_y; P.encoder_bit_depth = 8;
    const void* const src[3] = {y_src, u_src, v_src};
    const void* const pre[3] = {y_pre, u_pre, v_pre};
    const int ss[2] = {y_src_stride, uv_src_stride}, ps[2] = {y_pre_stride, uv_pre_stride};
    uint32_t* const acc[3] = {y_accum, u_accum, v_accum};
    uint16_t* const cnt[3] = {y_count, u_count, v_count};
    accum_host<uint8_t>(&P, block, src, ss, pre, ps, block_width, block_height, acc, cnt, 0);
}
void svt_av1_apply_temporal_filter_planewise_medium_hbd_hip(const SvtHipTfParams* params, const SvtHipTfBlock* block, const uint16_t* y_src, int y_src_stride,
                                                            const uint16_t* y_pre, int y_pre_stride, const uint16_t* u_src, const uint16_t* v_src,
                                                            int uv_src_stride, const uint16_t* u_pre, const uint16_t* v_pre, int uv_pre_stride,
                                                            unsigned int block_width, unsigned int block_height, int ss_x, int ss_y, uint32_t* y_accum,
                                                            uint16_t* y_count, uint32_t* u_accum, uint16_t* u_count, uint32_t* v_accum, uint16_t* v_count,
                                                            uint32_t encoder_bit_depth) {
    SvtHipTfParams P = *params;
    P.ss_x = (uint8_t)ss_x; P.ss_y = (uint8_t)ss_y; P.encoder_bit_depth = (uint8_t)encoder_bit_depth;
    const void* const src[3] = {y_src, u_src, v_src};
    const void* const pre[3] = {y_pre, u_pre, v_pre};
    const int ss[2] = {y_src_stride, uv_src_stride}, ps[2] = {y_pre_stride, uv_pre_stride};
    uint32_t* const acc[3] = {y_accum, u_accum, v_accum};
    uint16_t* const cnt[3] = {y_count, u_count, v_count};
    accum_host<uint16_t>(&P, block, src, ss, pre, ps, block_width, block_height, acc, cnt, 0);
}
void svt_av1_apply_zz_based_temporal_filter_planewise_medium_hip(const SvtHipTfParams* params, const SvtHipTfBlock* block, const uint8_t* y_pre,
                                                                 int y_pre_stride, const uint8_t* u_pre, const uint8_t* v_pre, int uv_pre_stride,
                                                                 unsigned int block_width, unsigned int block_height, int ss_x, int ss_y, uint32_t* y_accum,
                                                                 uint16_t* y_count, uint32_t* u_accum, uint16_t* u_count, uint32_t* v_accum, uint16_t* v_count) {
    SvtHipTfParams P = *params;
    P.ss_x = (uint8_t)ss_x; P.ss_y = (uint8_t)ss_y; P.encoder_bit_depth = 8;
    const void* const pre[3] = {y_pre, u_pre, v_pre};
    const int ps[2] = {y_pre_stride, uv_pre_stride};
    uint32_t* const acc[3] = {y_accum, u_accum, v_accum};
    uint16_t* const cnt[3] = {y_count, u_count, v_count};
    accum_host<uint8_t>(&P, block, pre, ps, pre, ps, block_width, block_height, acc, cnt, 1);
}
void svt_av1_apply_zz_based_temporal_filter_planewise_medium_hbd_hip(const SvtHipTfParams* params, const SvtHipTfBlock* block, const uint16_t* y_pre,
                                                                     int y_pre_stride, const uint16_t* u_pre, const uint16_t* v_pre, int uv_pre_stride,
                                                                     unsigned int block_width, unsigned int block_height, int ss_x, int ss_y,
                                                                     uint32_t* y_accum, uint16_t* y_count, uint32_t* u_accum, uint16_t* u_count,
                                                                     uint32_t* v_accum, uint16_t* v_count, uint32_t encoder_bit_depth) {
    SvtHipTfParams P = *params;
    P.ss_x = (uint8_t)ss_x; P.ss_y = (uint8_t)ss_y; P.encoder_bit_depth = (uint8_t)encoder_bit_depth;
    const void* const pre[3] = {y_pre, u_pre, v_pre};
    const int ps[2] = {y_pre_stride, uv_pre_stride};
    uint32_t* const acc[3] = {y_accum, u_accum, v_accum};
    uint16_t* const cnt[3] = {y_count, u_count, v_count};
    accum_host<uint16_t>(&P, block, pre, ps, pre, ps, block_width, block_height, acc, cnt, 1);
}

} // extern "C"

SVT_HIP_DEFINE_WARM(tf) // (svt_hip_warmup loads this translation unit's code object at encoder initialisation: svt_hip_common.h)
