// tpl_full.hip -- the TPL dispenser with the option set of tpl levels 0-3 (set_tpl_params, initial_rc_process.c:301-342; level 1 = presets M0-M2): every intra mode
// DC_PRED .. PAETH_PRED with the directional modes' edge filter, transform + SATD costs, half- / quarter-pel vectors, the coefficient-rate estimate.
//
// Reference: tpl_mc_flow_dispenser_sb_generic (Codec/src_ops_process.c:519-1198) with
//   * intra: svt_aom_update_neighbor_samples_array_open_loop_mb[_recon] (enc_intra_prediction.c:1127-1310), filter_intra_edge (intra_prediction.c:2521-2576:
//     corner + svt_av1_filter_intra_edge_c, no upsampling at 16x16), svt_aom_intra_prediction_open_loop_mb (:2578-2598: DC by availability, V, H, the three directional
//     zones with dx / dy of eb_dr_intra_derivative, SMOOTH / SMOOTH_V / SMOOTH_H, PAETH);
//   * inter: tpl_subpel_search (:418-517) = svt_av1_find_best_sub_pixel_tree_pruned (mcomp.c:606-686) with the bilinear sub-pixel variance
//     (svt_aom_sub_pixel_variance16x16, C_DEFAULT/variance.c:308-318), no MV cost (MV_COST_NONE), then svt_aom_enc_make_inter_predictor (EIGHTTAP_REGULAR both ways,
//     enc_inter_prediction.c:3158-3389) for a vector with a fractional part;
//   * costs: svt_nxm_sad_kernel, or svt_aom_subtract_block -> svt_av1_wht_fwd_txfm (DCT_DCT 16x16, partial-frequency shape) -> svt_aom_satd;
//   * get_quantize_error (:223-249) and rate_estimator (:251-264).
// Only 16x16 blocks exist at these levels (dispenser_search_level 0, subsample_tx 0).
//
// Mapping: ONE WAVE per 16x16 block.  Lane l owns the samples i = l + 64 k (k = 0..3; row i >> 4, column i & 15 -- the layout predict_rows of interp_core.h produces);
// the block's source, the candidate and the best inter prediction, the neighbour arrays and the transform tile live in the wave's LDS; the candidate loops are
// wave-uniform and sequential exactly as in the reference (first strict minimum wins), reductions are DPP sums.  The presets that select this option set spend
// seconds per frame in mode decision: the stage is about taking the dispenser off the host and keeping the planes resident, not about its kernel time.
#include "svt_hip_common.h"
#include "../../include/svtav1_hip.h"
#include "txfm_core.h"
#include "interp_core.h"

namespace {

constexpr int TPL_PAD = 32, TPL_NEWMV = 16, TPL_COST_SCALE_LOG2 = 4, PROB_COST_SHIFT = 9; // (encode_context.h:43-44, definitions.h:1143, :49, AV1_PROB_COST_SHIFT)
constexpr int NB = 16;   // neighbour arrays: sample k (-1 .. 31) of the row above / the column left of the block lives at [NB + k]
constexpr int TP = 17;   // transform tile pitch (dwords)
constexpr uint32_t FULL_WAIT_POLLS = 1u << 22;

// default_scan_16x16 (coefficients.h:440): anti-diagonals d = r + c, odd d downwards from the top row, even d upwards from the left column
struct Iscan16 { uint8_t v[256]; };
constexpr Iscan16 make_iscan16() {
    Iscan16 t{};
    int n = 0;
    for (int d = 0; d < 31; d++) {
        const int rmin = d > 15 ? d - 15 : 0, rmax = d < 15 ? d : 15;
        if (d & 1) for (int r = rmin; r <= rmax; r++) t.v[r * 16 + (d - r)] = (uint8_t)n++;
        else for (int r = rmax; r >= rmin; r--) t.v[r * 16 + (d - r)] = (uint8_t)n++;
    }
    return t;
}
__device__ constexpr Iscan16 kIscan16 = make_iscan16();
__device__ constexpr uint8_t kSmW16[16] = {255, 225, 196, 170, 145, 123, 102, 84, 68, 54, 43, 33, 26, 20, 17, 16}; // sm_weight_arrays + 16
// per intra mode: p_angle (mode_to_angle_map; 0 = not directional), dx, dy (eb_dr_intra_derivative, intra_prediction.c:245-296)
__device__ constexpr int8_t kEdgeKernel[3][5] = {{0, 4, 8, 4, 0}, {0, 5, 6, 5, 0}, {2, 4, 4, 4, 2}}; // svt_av1_filter_intra_edge_c, by strength - 1
__device__ constexpr int16_t kAngle[13] = {0, 90, 180, 45, 135, 113, 157, 203, 67, 0, 0, 0, 0};
__device__ constexpr int16_t kDx[13]    = {0, 0, 0, 64, 64, 27, 151, 1, 27, 0, 0, 0, 0};
__device__ constexpr int16_t kDy[13]    = {0, 0, 0, 1, 64, 151, 27, 27, 1, 0, 0, 0, 0};

// Every block of this file is ONE wave (64 lanes per 16x16 block), so every barrier in it is between lanes of one wave: a wave barrier (LDS traffic of a wave is served in
// order) -- which is what lets the reconstruction kernel put several independent blocks into one workgroup (one ticket draw per workgroup, see tpl_full_recon_kernel).
#define TPLF_BARRIER() __builtin_amdgcn_wave_barrier()
struct FullLds {
    uint8_t  src[256], pred[256], best[256];
    uint8_t  a0[64], l0[64], a[64], l[64]; // unfiltered / working neighbour arrays
    int32_t  tr[16 * TP];
    uint32_t im[12 * 16 + 16];             // predict_rows' horizontal intermediate: (16 + 7 + 1) / 2 row pairs x 16 columns
    uint16_t bil[17 * 16];                 // first pass of the bilinear sub-pixel variance
    SvtHipTplRef refs[8];
};

__device__ __forceinline__ int log2_floor(const uint32_t x) { return x ? 31 - __clz(x) : 0; } // svt_aom_log2f_32 (utility.c:160-172)

// ---- neighbour arrays (enc_intra_prediction.c:1127-1310, bwidth = bheight = 16, top-right / bottom-left on) as closed forms per sample --------------------------------
// pic: sample (0, 0) of the plane the neighbours come from (the source picture, or the reconstruction); width / height: the picture's
__device__ __forceinline__ void fill_neighbours(const uint8_t* __restrict__ pic, const size_t s, const int x0, const int y0, const int width, const int height,
                                                uint8_t* __restrict__ A, uint8_t* __restrict__ L, const int l) {
    if (l < 33) {
        const int k = l - 1, cw = width - x0 < 32 ? width - x0 : 32, ch = height - y0 < 32 ? height - y0 : 32;
        int a, b;
        // the column on the left (and the corner as the left pass leaves it)
        if (x0 != 0) {
            if (k < 0) b = y0 ? pic[(size_t)(y0 - 1) * s + x0 - 1] : pic[(size_t)y0 * s + x0 - 1];
            else { const int kk = k < 16 ? k : 15; b = kk < ch ? pic[(size_t)(y0 + kk) * s + x0 - 1] : 129; } // rows 16..31: "the value at (-1, 15)"
        } else if (y0 != 0) b = k < ch ? pic[(size_t)(y0 - 1) * s + x0] : 129;
        else b = k < 0 ? 128 : 129;
        // the row above
        if (y0 != 0) {
            if (k < 0) a = x0 ? pic[(size_t)(y0 - 1) * s + x0 - 1] : pic[(size_t)(y0 - 1) * s + x0];
            else if (x0 != 0) { const int kk = k < 16 ? k : 15; a = kk < cw ? pic[(size_t)(y0 - 1) * s + x0 + kk] : 127; } // columns 16..31: "the value at (15, -1)"
            else a = k < cw ? pic[(size_t)(y0 - 1) * s + x0 + k] : 127; // the first block column reads its real top-right neighbour
        } else if (x0 != 0) { // first block row: the left column's sample 32 - count, spread over the corner and the first `count` samples
            const int j = 32 - cw, jj = j < 16 ? j : 15;
            const int v = jj < ch ? pic[(size_t)(y0 + jj) * s + x0 - 1] : 129;
            a = k < cw ? v : 127;
        } else a = k < 0 ? 128 : 127;
        A[NB + k] = (uint8_t)a;
        L[NB + k] = (uint8_t)b;
    }
}

// ---- filter_intra_edge for one directional mode: A / L = filtered copies of A0 / L0 (intra_prediction.c:2521-2576; strengths of svt_aom_intra_edge_filter_strength for
// blk_wh 32, type 0) ----
__device__ __forceinline__ int edge_strength(const int delta) { const int d = delta < 0 ? -delta : delta; return d >= 32 ? 3 : (d >= 4 ? 2 : (d >= 1 ? 1 : 0)); }
__device__ __forceinline__ void filter_edges(const int angle, const int x0, const int y0, const uint8_t* __restrict__ A0, const uint8_t* __restrict__ L0,
                                             uint8_t* __restrict__ A, uint8_t* __restrict__ L, const int l) {
    const bool need_above = angle < 180, need_left = angle > 90;
    // the corner first (both edges needed: 90 < angle < 180), from unfiltered samples; the edge passes copy their input before they write and never write sample -1
    int corner_a = A0[NB - 1], corner_l = L0[NB - 1];
    if (need_above && need_left) corner_a = corner_l = (L0[NB] * 5 + A0[NB - 1] * 6 + A0[NB] * 5 + 8) >> 4;
    if (l < 33) {
        const int k = l - 1; // the sample this lane produces of each array
        auto pass = [&](const uint8_t* E0, const int corner, const int sz, const int strength) -> int { // p = &E[-1], p[i] with i = k + 1
            const int i = k + 1;
            if (i < 1 || i >= sz || !strength) return k < 0 ? corner : E0[NB + k];
            int s = 0;
#pragma unroll
            for (int j = 0; j < 5; j++) {
                int q = i - 2 + j;
                q = q < 0 ? 0 : (q > sz - 1 ? sz - 1 : q);
                s += (q == 0 ? corner : (int)E0[NB + q - 1]) * (int)kEdgeKernel[strength - 1][j];
            }
            return (s + 8) >> 4;
        };
        int a = k < 0 ? corner_a : A0[NB + k], b = k < 0 ? corner_l : L0[NB + k];
        if (need_above && y0 > 0) a = pass(A0, corner_a, 17 + (angle < 90 ? 16 : 0), edge_strength(angle - 90));
        if (need_left && x0 > 0) b = pass(L0, corner_l, 17 + (angle > 180 ? 16 : 0), edge_strength(angle - 180));
        A[NB + k] = (uint8_t)a;
        L[NB + k] = (uint8_t)b;
    }
}

// ---- one intra prediction sample (svt_aom_intra_prediction_open_loop_mb) -------------------------------------------------------------------------------------------------
__device__ __forceinline__ int intra_sample(const int mode, const int r, const int c, const uint8_t* __restrict__ A, const uint8_t* __restrict__ L, const int dc) {
    const uint8_t* a = A + NB;
    const uint8_t* b = L + NB;
    switch (mode) {
    case 0: return dc;
    case 1: return a[c];
    case 2: return b[r];
    case 3: case 8: { // zone 1 (svt_av1_dr_prediction_z1_c)
        const int x = kDx[mode] * (r + 1), base = (x >> 6) + c, sh = (x & 63) >> 1;
        return base < 31 ? (a[base] * (32 - sh) + a[base + 1] * sh + 16) >> 5 : a[31];
    }
    case 7: { // zone 3
        const int y = kDy[mode] * (c + 1), base = (y >> 6) + r, sh = (y & 63) >> 1;
        return base < 31 ? (b[base] * (32 - sh) + b[base + 1] * sh + 16) >> 5 : b[31];
    }
    case 4: case 5: case 6: { // zone 2
        const int x = -kDx[mode] * (r + 1), base1 = (x >> 6) + c;
        if (base1 >= -1) { const int sh = (x & 63) >> 1; return (a[base1] * (32 - sh) + a[base1 + 1] * sh + 16) >> 5; }
        const int y = (r << 6) - kDy[mode] * (c + 1), base2 = y >> 6, sh = (y & 63) >> 1;
        return (b[base2] * (32 - sh) + b[base2 + 1] * sh + 16) >> 5;
    }
    case 9: return (kSmW16[r] * a[c] + (256 - kSmW16[r]) * b[15] + kSmW16[c] * b[r] + (256 - kSmW16[c]) * a[15] + 256) >> 9;
    case 10: return (kSmW16[r] * a[c] + (256 - kSmW16[r]) * b[15] + 128) >> 8;
    case 11: return (kSmW16[c] * b[r] + (256 - kSmW16[c]) * a[15] + 128) >> 8;
    default: { // PAETH
        const int top = a[c], left = b[r], tl = a[-1], base = top + left - tl;
        const int pl = base > left ? base - left : left - base, pt = base > top ? base - top : top - base, ptl = base > tl ? base - tl : tl - base;
        return (pl <= pt && pl <= ptl) ? left : (pt <= ptl ? top : tl);
    }
    }
}
// the whole 16x16 prediction of `mode` into out[256]; A0 / L0 hold the unfiltered neighbours, A / L are scratch for the filtered copies
__device__ __forceinline__ void intra_predict(const int mode, const int x0, const int y0, const uint8_t* A0, const uint8_t* L0, uint8_t* A, uint8_t* L, uint8_t* out,
                                              const int l) {
    const int  angle = kAngle[mode];
    const bool directional = angle != 0 && angle != 90 && angle != 180; // (V and H: p_angle 90 / 180 -- no edge filter, the plain predictors)
    const uint8_t *pa = A0, *pl = L0;
    if (directional) {
        filter_edges(angle, x0, y0, A0, L0, A, L, l);
        TPLF_BARRIER();
        pa = A; pl = L;
    }
    int dc = 128;
    if (mode == 0) { // svt_aom_dc_pred[x > 0][y > 0]
        int s = 0;
        if (l < 16) s = (y0 > 0 ? pa[NB + l] : 0) + (x0 > 0 ? pl[NB + l] : 0);
        s = wave_sum_i32(s);
        if (x0 > 0 && y0 > 0) dc = (s + 16) / 32;
        else if (x0 > 0 || y0 > 0) dc = (s + 8) / 16;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int i = l + 64 * k;
        out[i] = (uint8_t)intra_sample(mode, i >> 4, i & 15, pa, pl, dc);
    }
    TPLF_BARRIER();
}

// ---- residual -> forward DCT_DCT 16x16 (svt_av1_highbd_fwd_txfm, shifts {2, -2, 0}); the coefficients end up in tr[row * TP + column] --------------------------------
__device__ __forceinline__ void fwd16(const uint8_t* __restrict__ src, const uint8_t* __restrict__ pred, int32_t* __restrict__ tr, const int l) {
    constexpr int FS0 = fwd_shift0(16, 16), FS1 = -fwd_shift1(16, 16), FS2 = -fwd_shift2(16, 16);
    constexpr int CBC = kFwdCosCol[2][2], CBR = kFwdCosRow[2][2];
    static_assert(FS2 == 0 && FS1 > 0, "16x16 shifts");
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int i = l + 64 * k;
        tr[(i >> 4) * TP + (i & 15)] = (int32_t)((uint32_t)((int)src[i] - (int)pred[i]) << FS0);
    }
    TPLF_BARRIER();
    if (l < 16) {
        int32_t v[16];
#pragma unroll
        for (int r = 0; r < 16; r++) v[r] = tr[r * TP + l];
        fwd1d<16, CBC>(K_DCT, v);
#pragma unroll
        for (int r = 0; r < 16; r++) tr[r * TP + l] = rshift_round(v[r], FS1);
    }
    TPLF_BARRIER();
    if (l < 16) {
        int32_t v[16];
#pragma unroll
        for (int c = 0; c < 16; c++) v[c] = tr[l * TP + c];
        fwd1d<16, CBR>(K_DCT, v);
#pragma unroll
        for (int c = 0; c < 16; c++) tr[l * TP + c] = v[c];
    }
    TPLF_BARRIER();
}
// svt_aom_satd over the coefficients the partial-frequency shape keeps (the rest is zero: transforms.c:5202-5273)
__device__ __forceinline__ uint32_t satd16(const int32_t* __restrict__ tr, const int keep, const int l) {
    int s = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int i = l + 64 * k, r = i >> 4, c = i & 15;
        if (r < keep && c < keep) { const int32_t x = tr[r * TP + c]; s += x < 0 ? -x : x; }
    }
    return (uint32_t)wave_sum_i32(s);
}
__device__ __forceinline__ uint32_t sad16(const uint8_t* __restrict__ src, const uint8_t* __restrict__ pred, const int l) {
    int s = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int i = l + 64 * k, d = (int)src[i] - (int)pred[i];
        s += d < 0 ? -d : d;
    }
    return (uint32_t)wave_sum_i32(s);
}
// cost of the prediction in `pred` (:729-748, :862-882)
__device__ __forceinline__ uint32_t block_cost(const SvtHipTplSrcParams& P, FullLds& S, const uint8_t* pred, const int l) {
    if (!(P.search_flags & 1)) return sad16(S.src, pred, l);
    fwd16(S.src, pred, S.tr, l);
    const uint32_t c = satd16(S.tr, 16 >> P.pf_shape, l);
    TPLF_BARRIER();
    return c;
}

// get_quantize_error + rate_estimator on the coefficients in tr; the dequantised coefficients replace them (input of the inverse transform)
struct QuantOut { long long err; int eob, rate; };
__device__ __forceinline__ QuantOut quantize16(const SvtHipTplSrcParams& P, int32_t* __restrict__ tr, const int l) {
    const int keep = 16 >> P.pf_shape;
    unsigned long long err = 0;
    int eob = 0, rate = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int i = l + 64 * k, r = i >> 4, c = i & 15;
        int32_t   dq = 0;
        if (r < keep && c < keep) {
            const int32_t x = tr[r * TP + c];
            const int     kk = i != 0;
            const int32_t sign = x < 0 ? -1 : 0, a = (x ^ sign) - sign;
            // svt_av1_quantize_fp (quantize_fp_helper_c, full_loop.c:282-342; log_scale 0, no matrices) and svt_av1_block_error's term
            if (((long long)a << 1) >= (int32_t)P.dequant[kk]) {
                long long tt = (long long)a + P.round_fp[kk];
                tt = tt < -32768 ? -32768 : (tt > 32767 ? 32767 : tt);
                const int32_t q = (int32_t)((tt * P.quant_fp[kk]) >> 16);
                if (q) {
                    dq = (((int32_t)((uint32_t)q * (uint32_t)(int32_t)P.dequant[kk])) ^ sign) - sign;
                    const int pos = (int)kIscan16.v[i] + 1;
                    eob = pos > eob ? pos : eob;
                    rate += log2_floor((uint32_t)q + 1u) + 1; // rate_estimator's term of a non-zero level (zero levels add nothing)
                }
            }
            const long long df = (long long)x - dq;
            err += (unsigned long long)(df * df);
        }
        tr[r * TP + c] = dq;
    }
    const unsigned long long l0 = (uint32_t)wave_sum_i32((int)(err & 0x3fffffu)), l1 = (uint32_t)wave_sum_i32((int)((err >> 22) & 0x3fffffu)),
                             l2 = (uint32_t)wave_sum_i32((int)(err >> 44));
    QuantOut o;
    o.err = (long long)(l0 + (l1 << 22) + (l2 << 44));
    o.err >>= 2; // tx_size != TX_32X32 (:227)
    if (o.err < 1) o.err = 1;
    // the wave maximum of eob: lanes hold values <= 256 -- sum of one-hot is not possible, use the comparison tree on shuffles
    int e = eob;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { const int other = __shfl_xor(e, m); e = other > e ? other : e; }
    o.eob  = e;
    o.rate = (wave_sum_i32(rate) + e + 1) << PROB_COST_SHIFT;
    TPLF_BARRIER();
    return o;
}

// ---- inter prediction of the block at vector (mvr, mvc) (1/8 sample) from the plane `ref` (sample (0, 0) of the picture) into out ---------------------------------------
// a vector without fractional part copies the block (:803-806, :1016-1019, :1058-1062); with one: svt_aom_enc_make_inter_predictor (clamp_mv_to_umv_border_sb, regular kernels)
__device__ __forceinline__ void inter_predict(const SvtHipTplSrcParams& P, const uint8_t* __restrict__ ref, const long rs, const int x0, const int y0, const int mvr,
                                              const int mvc, uint8_t* __restrict__ out, uint32_t* __restrict__ im, const int l) {
    if (!((mvr | mvc) & 7)) {
        const uint8_t* p0 = ref + (long)(y0 + (mvr >> 3)) * rs + x0 + (mvc >> 3);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = l + 64 * k;
            out[i] = p0[(long)(i >> 4) * rs + (i & 15)];
        }
        TPLF_BARRIER();
        return;
    }
    const int aligned_h = (int)((P.height + 7) & ~7u), mi_rows = aligned_h >> 2, mi_cols = (int)P.aligned_width >> 2, mirow = y0 >> 2, micol = x0 >> 2;
    const int to_top = -((mirow * 4) * 8), to_bottom = ((mi_rows - 4 - mirow) * 4) * 8, to_left = -((micol * 4) * 8), to_right = ((mi_cols - 4 - micol) * 4) * 8; // init_xd_tpl (:403-416)
    const int spel_l = (4 + 16) << 4, spel_r = spel_l - 16;
    int row = (int16_t)(mvr * 2), col = (int16_t)(mvc * 2);
    const int min_col = to_left * 2 - spel_l, max_col = to_right * 2 + spel_r, min_row = to_top * 2 - spel_l, max_row = to_bottom * 2 + spel_r;
    col = col < min_col ? min_col : (col > max_col ? max_col : col);
    row = row < min_row ? min_row : (row > max_row ? max_row : row);
    col = (int16_t)col; row = (int16_t)row;
    const int sx = col & 15, sy = row & 15;
    const uint8_t* p0 = ref + (long)(y0 + (row >> 4)) * rs + x0 + (col >> 4);
    auto finish = [&](const int i, int px) { out[i] = (uint8_t)(px < 0 ? 0 : (px > 255 ? 255 : px)); };
    predict_rows<uint8_t, 8>(p0, rs, 16, 16, 1, sx, sy, kTaps.t[0][sx], kTaps.t[0][sy], 8, im, l, finish);
    TPLF_BARRIER();
}

// ---- svt_aom_sub_pixel_variance16x16 (bilinear taps {128 - 16 o, 16 o}) of the reference block at vector (mvr, mvc) against the source block --------------------------
__device__ __forceinline__ uint32_t subpel_variance(const uint8_t* __restrict__ ref, const long rs, const int x0, const int y0, const int mvr, const int mvc,
                                                    const uint8_t* __restrict__ src, uint16_t* __restrict__ bil, const int l) {
    const uint8_t* p0 = ref + (long)(y0 + (mvr >> 3)) * rs + x0 + (mvc >> 3);
    const int xo = mvc & 7, yo = mvr & 7, f0 = 128 - 16 * xo, f1 = 16 * xo, g0 = 128 - 16 * yo, g1 = 16 * yo;
    for (int i = l; i < 17 * 16; i += 64) {
        const uint8_t* q = p0 + (long)(i >> 4) * rs + (i & 15);
        bil[i] = (uint16_t)(((int)q[0] * f0 + (int)q[1] * f1 + 64) >> 7);
    }
    TPLF_BARRIER();
    int sum = 0;
    uint32_t sse = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int i = l + 64 * k;
        const int px = ((int)bil[i] * g0 + (int)bil[i + 16] * g1 + 64) >> 7;
        const int d = px - (int)src[i];
        sum += d; sse += (uint32_t)(d * d);
    }
    const int          su = wave_sum_i32(sum);
    const uint32_t     ss = (uint32_t)wave_sum_i32((int)sse); // <= 256 * 255^2
    TPLF_BARRIER();
    return ss - (uint32_t)(((long long)su * su) >> 8); // svt_aom_variance16x16_c
}

// ---- tpl_subpel_search: svt_av1_find_best_sub_pixel_tree_pruned from the full-pel vector (mvr, mvc); returns the best vector ------------------------------------------------
__device__ __forceinline__ void subpel_search(const SvtHipTplSrcParams& P, const uint8_t* __restrict__ ref, const long rs, const int x0, const int y0, int& mvr, int& mvc,
                                              const uint8_t* __restrict__ src, uint16_t* __restrict__ bil, const int l) {
    const int rounds = (P.search_flags >> 2) & 3;
    if (!rounds) return;
    const bool no_diag = (P.search_flags & 16) != 0; // subpel_diag_refinement 4: org_error = 0 -- no diagonal check, no second level
    // the sub-pel limits: svt_av1_set_mv_search_range + svt_av1_set_subpel_mv_search_range around ref_mv (0, 0) (:448-455)
    const int aligned_h = (int)((P.height + 7) & ~7u);
    auto lim_lo = [](int lo_fp) { lo_fp = lo_fp < -1023 ? -1023 : lo_fp; const int lo = lo_fp * 8 > -8184 ? lo_fp * 8 : -8184; return lo > -16383 ? lo : -16383; };
    auto lim_hi = [](int hi_fp) { hi_fp = hi_fp > 1023 ? 1023 : hi_fp; const int hi = hi_fp * 8 < 8184 ? hi_fp * 8 : 8184; return hi < 16383 ? hi : 16383; };
    const int cmin = lim_lo(-(x0 + 16 + 4)), cmax = lim_hi((int)P.aligned_width - x0 + 4), rmin = lim_lo(-(y0 + 16 + 4)), rmax = lim_hi(aligned_h - y0 + 4);
    int br = mvr, bc = mvc;
    uint32_t besterr = subpel_variance(ref, rs, x0, y0, br, bc, src, bil, l); // svt_upsampled_setup_center_error: the variance at the full-pel start, no MV cost
    auto check = [&](const int r, const int c) -> uint32_t { // svt_check_better_fast
        if (c < cmin || c > cmax || r < rmin || r > rmax) return 0x7fffffffu;
        const uint32_t cost = subpel_variance(ref, rs, x0, y0, r, c, src, bil, l);
        if (cost < besterr) { besterr = cost; br = r; bc = c; }
        return cost;
    };
    int sr = mvr, sc = mvc, hstep = 4;
    const int n_rounds = rounds < 2 ? rounds : 2; // AOMMIN(FULL_PEL - forced_stop, 3 - !allow_hp), allow_hp 0
    for (int iter = 0; iter < n_rounds; iter++) {
        // first_level_check_fast
        const uint32_t left = check(sr, sc - hstep), right = check(sr, sc + hstep), up = check(sr - hstep, sc), down = check(sr + hstep, sc);
        const int dr = up <= down ? -hstep : hstep, dc = left <= right ? -hstep : hstep;
        if (!no_diag) {
            check(sr + dr, sc + dc);
            // second_level_check_fast (iters_per_step 2): around the best so far
            const int tr_ = sr, tc_ = sc, b_r = br, b_c = bc;
            if (tr_ != b_r && tc_ != b_c) { check(b_r, b_c + dc); check(b_r + dr, b_c); }
            else if (tr_ == b_r && tc_ != b_c) { check(b_r + hstep, b_c + dc); check(b_r - hstep, b_c + dc); check(b_r - dr, b_c); }
            else if (tr_ != b_r && tc_ == b_c) { check(b_r + dr, b_c + hstep); check(b_r + dr, b_c - hstep); check(b_r, b_c - dc); }
        }
        hstep >>= 1;
        sr = br; sc = bc;
    }
    mvr = br; mvc = bc;
}

// ======================================================== the source-based half ========================================================================================
__global__ __launch_bounds__(64) void tpl_full_src_kernel(const SvtHipTplSrcParams P, const uint8_t* __restrict__ src_base, const uint8_t* __restrict__ ref_base,
                                                          const uint8_t* __restrict__ tot_base, const uint32_t* __restrict__ mv_base, const uint8_t* __restrict__ cand_base,
                                                          SvtHipTplSrcStats* __restrict__ stats) {
    __shared__ FullLds S;
    const int l = (int)threadIdx.x;
    if (l < 8) S.refs[l] = P.refs[l];
    const uint32_t item = blockIdx.x, sb = item >> 4, k = item & 15;
    const int      sx = (int)(sb % P.sbs_x) * 64, sy = (int)(sb / P.sbs_x) * 64, x0 = sx + (int)(k & 3) * 16, y0 = sy + (int)(k >> 2) * 16;
    if (sb >= P.n_sb || x0 + 8 > (int)P.width || y0 + 8 > (int)P.height) return; // at least half of the block inside (:580)
    const int n_pus = P.enable_me_8x8 ? 85 : (P.enable_me_16x16 ? 21 : 5);
    int       pu    = 5 + (int)(k >> 2) * 4 + (int)(k & 3); // tpl_blk_idx_tab[1] (:355)
    if (!P.enable_me_16x16) pu = (pu - 1) / 4;               // :762-763
    const uint8_t* src = src_base + P.src_off;
    const size_t   ss  = P.src_stride;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int i = l + 64 * q;
        S.src[i] = src[(size_t)(y0 + (i >> 4)) * ss + x0 + (i & 15)];
    }
    fill_neighbours(src, ss, x0, y0, (int)P.width, (int)P.height, S.a0, S.l0, l);
    TPLF_BARRIER();
    // ---- intra (:611-757) ----
    uint32_t best_intra = 0xffffffffu;
    int      best_intra_mode = 0;
    if (!P.disable_intra_pred) {
        for (int mode = 0; mode <= (int)P.intra_mode_end; mode++) {
            intra_predict(mode, x0, y0, S.a0, S.l0, S.a, S.l, S.pred, l);
            const uint32_t cost = block_cost(P, S, S.pred, l);
            if (mode == 0 || cost < best_intra) { best_intra = cost; best_intra_mode = mode; } // (INT64_MAX in the reference: the first mode always wins)
        }
    }
    // ---- inter (:760-890) ----
    uint32_t  best_inter = 0xffffffffu;
    bool      have_inter = false;
    int       best_rf = -1, bmvr = 0, bmvc = 0;
    const int n_cand = P.i_slice ? 0 : (int)tot_base[(size_t)sb * n_pus + pu];
    for (int i = 0; i < n_cand; i++) {
        const uint32_t c = cand_base[((size_t)sb * n_pus + pu) * P.max_cand + i];
        const int dir = (int)(c & 3), r0 = (int)((c >> 2) & 3), r1 = (int)((c >> 4) & 3);
        if (dir > 1) continue;
        const int list = dir & 1, ref = list == 0 ? r0 : r1, rf = list * 4 + ref;
        const SvtHipTplRef& R = S.refs[rf];
        if (!R.valid) continue;
        const uint32_t m = mv_base[((size_t)sb * n_pus + pu) * P.max_refs + (list ? P.max_l0 : 0) + ref];
        int xm = (int)(int16_t)((int16_t)(m & 0xffff) * 8), ym = (int)(int16_t)((int16_t)(m >> 16) * 8);
        if (x0 + (xm >> 3) < -TPL_PAD) xm = (int)(int16_t)((-TPL_PAD - x0) * 8);
        if (x0 + 16 + (xm >> 3) > TPL_PAD + (int)R.max_width - 1) xm = (int)(int16_t)(((TPL_PAD + (int)R.max_width - 1) - (x0 + 16)) * 8);
        if (y0 + (ym >> 3) < -TPL_PAD) ym = (int)(int16_t)((-TPL_PAD - y0) * 8);
        if (y0 + 16 + (ym >> 3) > TPL_PAD + (int)R.max_height - 1) ym = (int)(int16_t)(((TPL_PAD + (int)R.max_height - 1) - (y0 + 16)) * 8);
        const uint8_t* rp = ref_base + R.plane_off + (size_t)R.org_y * R.stride + R.org_x; // sample (0, 0) of the reference picture
        int mvr = ym, mvc = xm;
        subpel_search(P, rp, (long)R.stride, x0, y0, mvr, mvc, S.src, S.bil, l);
        mvr = (int16_t)mvr; mvc = (int16_t)mvc;
        inter_predict(P, rp, (long)R.stride, x0, y0, mvr, mvc, S.pred, S.im, l);
        const uint32_t cost = block_cost(P, S, S.pred, l);
        if (!have_inter || cost < best_inter) {
            have_inter = true; best_inter = cost; best_rf = rf; bmvr = mvr; bmvc = mvc;
#pragma unroll
            for (int q = 0; q < 4; q++) S.best[l + 64 * q] = S.pred[l + 64 * q];
        }
        TPLF_BARRIER();
    }
    // (the costs are below 2^31: "nothing evaluated" = INT64_MAX in the reference is never less than anything)
    const bool newmv = have_inter && (P.disable_intra_pred ? true : best_inter < best_intra);
    SvtHipTplSrcStats o = {};
    o.written = 1;
    o.best_mode = newmv ? TPL_NEWMV : (uint8_t)best_intra_mode;
    o.best_intra_mode = (uint8_t)best_intra_mode;
    o.best_rf_idx = best_rf;
    o.ref_frame_poc = best_rf >= 0 ? S.refs[best_rf].picture_number : 0;
    o.mv_row = (int16_t)bmvr; o.mv_col = (int16_t)bmvc;
    if (newmv) { // the best candidate's coefficients (:895-951) -> quantisation error and rate
        fwd16(S.src, S.best, S.tr, l);
        const QuantOut q = quantize16(P, S.tr, l);
        o.srcrf_dist = q.err << TPL_COST_SCALE_LOG2;
        o.srcrf_rate = (P.search_flags & 2) ? (long long)q.rate << TPL_COST_SCALE_LOG2 : 0;
    }
    if (l == 0) stats[(size_t)(y0 >> 4) * ((P.aligned_width + 15) >> 4) + (x0 >> 4)] = o;
}

// ======================================================== the reconstruction half ======================================================================================
// One wave per block, every block in flight, tickets in anti-diagonal order (as tpl_recon_dep_kernel of tpl.hip: a waiting wave only waits for tickets drawn earlier).
// An intra block reads the reconstruction above, left, above-left and -- in the first block column, whose top-right samples are real -- above-right of it; the cells'
// "reconstructed" flags live in SvtHipTplReconStats.reserved.
constexpr int REC_WAVES = 8; // independent blocks (waves) per workgroup of the reconstruction kernel
__global__ __launch_bounds__(64 * REC_WAVES) void tpl_full_recon_kernel(const SvtHipTplReconParams RP, const uint8_t* __restrict__ src_base, const uint8_t* __restrict__ ref_base,
                                                            const SvtHipTplSrcStats* __restrict__ src_stats, uint8_t* __restrict__ recon_base,
                                                            SvtHipTplReconStats* __restrict__ out, uint32_t* __restrict__ sync, const int cols16, const int rows16, const int wt /* 1: write-through hand-off (svt_hip_common.h), 0: release fence */) {
    __shared__ FullLds  Sw[REC_WAVES];
    __shared__ uint32_t s_ticket;
    const SvtHipTplSrcParams& P = RP.src;
    const int l = (int)threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    FullLds&  S = Sw[wv];
    if (l < 8) S.refs[l] = RP.rec_refs[l];
    // ONE ticket draw per workgroup, its waves take consecutive tickets (8 160 returning atomics on one word -- one per single-wave workgroup -- cost more than the blocks'
    // work: csrc/tpl.hip tpl_recon_dep_kernel); this is the kernel's only workgroup barrier, afterwards the waves are independent blocks
    if (threadIdx.x == 0) s_ticket = atomicAdd(&sync[0], (uint32_t)REC_WAVES);
    __syncthreads();
    const int ticket = (int)s_ticket + wv, u = ticket / rows16, cy = ticket % rows16, cx = u - cy;
    if (cx < 0 || cx >= cols16) return;
    const int    x0 = cx * 16, y0 = cy * 16;
    const size_t cell = (size_t)cy * cols16 + cx;
    const uint8_t* src = src_base + P.src_off;
    uint8_t*       rec = recon_base + RP.recon_off;
    const size_t   ss = P.src_stride, rs = RP.recon_stride;
    bool active = !(x0 + 8 > (int)P.width || y0 + 8 > (int)P.height);
    SvtHipTplSrcStats s = {};
    if (active) s = src_stats[cell];
    active = active && s.written;
    const bool newmv = active && s.best_mode == TPL_NEWMV;
    if (active) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int i = l + 64 * q;
            S.src[i] = src[(size_t)(y0 + (i >> 4)) * ss + x0 + (i & 15)];
        }
        if (newmv) {
            const SvtHipTplRef& R = S.refs[s.best_rf_idx & 7];
            inter_predict(P, ref_base + R.plane_off + (size_t)R.org_y * R.stride + R.org_x, (long)R.stride, x0, y0, s.mv_row, s.mv_col, S.pred, S.im, l);
        } else {
            // the neighbours' reconstruction must be there: one lane per cell polls
            bool timed_out = false;
            if (l < 4) { // lane 0: above, 1: left, 2: above-left, 3: above-right (the first block column only)
                const int nx = cx + (l == 0 ? 0 : (l == 3 ? 1 : -1)), ny = cy + (l == 1 ? 0 : -1);
                if (nx >= 0 && ny >= 0 && nx < cols16 && (l < 3 || cx == 0)) {
                    uint32_t* flag  = &out[(size_t)ny * cols16 + nx].reserved;
                    uint32_t  polls = 0;
                    // (a relaxed agent-scope LOAD per poll: read-modify-writes queue the pollers of one flag at the memory side's atomic unit)
                    while ((wt ? __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : atomicAdd(flag, 0u)) == 0u && polls < FULL_WAIT_POLLS) { polls++; __builtin_amdgcn_s_sleep(1); }
                    timed_out = polls >= FULL_WAIT_POLLS;
                }
            }
            if (timed_out) atomicAdd(&sync[1], 1u);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            TPLF_BARRIER();
            fill_neighbours(rec, rs, x0, y0, (int)P.width, (int)P.height, S.a0, S.l0, l);
            TPLF_BARRIER();
            intra_predict(s.best_intra_mode, x0, y0, S.a0, S.l0, S.a, S.l, S.pred, l);
        }
        // residual -> transform -> quantisation error, rate (:1112-1131)
        fwd16(S.src, S.pred, S.tr, l);
        const QuantOut q = quantize16(P, S.tr, l);
        const bool inverse = q.eob != 0 && (!P.disable_intra_pred || RP.is_ref); // (:1135-1136)
        if (inverse) { // svt_aom_inv_transform_recon8bit, DCT_DCT 16x16: rows, then columns added to the prediction
            constexpr int S0 = -inv_shift0(16, 16);
            if (l < 16) {
                const int32_t rhi = (1 << 15) - 1, rlo = -(1 << 15);
                int32_t v[16];
#pragma unroll
                for (int c = 0; c < 16; c++) v[c] = txfm1d::clamp_i32(S.tr[l * TP + c], rlo, rhi);
                inv1d<16>(K_DCT, v, rlo, rhi);
#pragma unroll
                for (int c = 0; c < 16; c++) S.tr[l * TP + c] = S0 ? rshift_round(v[c], S0 ? S0 : 1) : v[c];
            }
            TPLF_BARRIER();
            if (l < 16) {
                const int32_t chi = (1 << 15) - 1, clo = -(1 << 15);
                int32_t v[16];
#pragma unroll
                for (int r = 0; r < 16; r++) v[r] = txfm1d::clamp_i32(S.tr[r * TP + l], clo, chi);
                inv1d<16>(K_DCT, v, clo, chi);
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int32_t px = (int32_t)S.pred[r * 16 + l] + rshift_round(v[r], 4);
                    S.pred[r * 16 + l] = (uint8_t)(px < 0 ? 0 : (px > 255 ? 255 : px));
                }
            }
            TPLF_BARRIER();
        }
        if (wt) { // a row of sixteen samples per lane, one 16-byte write-through store each (sample-wide write-through stores would be one fabric write apiece)
            if (l < 16) {
                const uint32_t* pr = (const uint32_t*)(S.pred + 16 * l);
                svt_hip_store_x4_wt(rec + (size_t)(y0 + l) * rs + x0, pr[0], pr[1], pr[2], pr[3]);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int i = l + 64 * k;
                rec[(size_t)(y0 + (i >> 4)) * rs + x0 + (i & 15)] = S.pred[i];
            }
        }
        if (l == 0) {
            SvtHipTplReconStats o = {};
            o.written = 1; o.coded = q.eob != 0;
            const long long rate = (P.search_flags & 2) ? (long long)q.rate << TPL_COST_SCALE_LOG2 : 0;
            o.recrf_dist = q.err << TPL_COST_SCALE_LOG2;
            o.recrf_rate = rate;
            o.srcrf_dist = newmv ? s.srcrf_dist : o.recrf_dist;
            o.srcrf_rate = newmv ? s.srcrf_rate : rate;
            if (o.srcrf_dist > o.recrf_dist) o.recrf_dist = o.srcrf_dist;
            if (o.srcrf_rate > o.recrf_rate) o.recrf_rate = o.srcrf_rate;
            if (wt) { // everything but `reserved`: the cell's flag is published by a coherent store a plain store of the whole record must not shadow
                SvtHipTplReconStats* q2 = &out[cell];
                q2->srcrf_dist = o.srcrf_dist; q2->recrf_dist = o.recrf_dist; q2->srcrf_rate = o.srcrf_rate; q2->recrf_rate = o.recrf_rate;
                q2->written = o.written; q2->coded = o.coded; q2->pad[0] = 0; q2->pad[1] = 0;
            } else {
                out[cell] = o;
            }
        }
    }
    if (wt) svt_hip_drain_stores(); // the rows have left this XCD's L2 ...
    else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    TPLF_BARRIER();
    if (l == 0) { // ... before the cell is published
        if (wt) __hip_atomic_store(&out[cell].reserved, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else atomicExch(&out[cell].reserved, 1u);
    }
}

} // namespace

namespace svthip {

bool tpl_full_wanted(const SvtHipTplSrcParams& P) { return P.intra_mode_end != 0 || P.search_flags != 0; }
bool tpl_full_supported(const SvtHipTplSrcParams& P) {
    return P.dispenser_search_level == 0 && P.subsample_tx == 0 && P.pf_shape <= 2 && P.intra_mode_end <= 12 && ((P.search_flags >> 2) & 3) <= 2 && !(P.search_flags & 0xe0) &&
           P.n_sb && P.sbs_x;
}
void tpl_full_src_launch(const SvtHipTplSrcParams& P, const uint8_t* src, const uint8_t* ref, const uint8_t* tot, const uint32_t* mv, const uint8_t* cand,
                         SvtHipTplSrcStats* stats, hipStream_t st) {
    hipLaunchKernelGGL(tpl_full_src_kernel, dim3(P.n_sb * 16), dim3(64), 0, st, P, src, ref, tot, mv, cand, stats);
    SVT_LAUNCH_CHECK();
}
// out's flags cleared and sync[0..1] zero on entry (the caller's reset kernel); the caller's finish kernel reports sync[1]
void tpl_full_recon_launch(const SvtHipTplReconParams& R, const uint8_t* src, const uint8_t* ref, const SvtHipTplSrcStats* ss, uint8_t* rec, SvtHipTplReconStats* out,
                           uint32_t* sync, int cols16, int rows16, int wt, hipStream_t st) {
    hipLaunchKernelGGL(tpl_full_recon_kernel, dim3(((cols16 + rows16 - 1) * rows16 + REC_WAVES - 1) / REC_WAVES), dim3(64 * REC_WAVES), 0, st, R, src, ref, ss, rec, out, sync, cols16, rows16, wt);
    SVT_LAUNCH_CHECK();
}

} // namespace svthip

SVT_HIP_DEFINE_WARM(tpl_full) // (svt_hip_warmup loads this translation unit's code object at encoder initialisation: svt_hip_common.h)
