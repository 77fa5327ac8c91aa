// hme.hip -- one hierarchical-ME level for a whole picture (the callers of a2): hme_level_0 / hme_level_1 / hme_level_2
// (Codec/motion_estimation.c:820-921, 923-1018, 1020-1116) for every (reference, 64x64 SB, search region).  The reference runs these leaf
// drivers per SB on a worker thread; here a descriptor kernel reproduces their search-area placement and clipping (int16_t arithmetic, as the
// reference), the search is svt_hip_sad_loop_batch over all items at once, and a second kernel applies the sub-sampling factor and rescales the
// winning position to the next level's resolution.  Nothing goes through the host between the levels.
#include "svt_hip_common.h"
#include "../../include/svtav1_hip.h"
#include "hme_geom.h"

namespace {

struct HmeItem { // geometry of one item, shared by the two kernels
    int16_t sa_origin_x, sa_origin_y;
};

__device__ __forceinline__ bool hme_item_skipped(const SvtHipHmeLevelParams& P, const uint32_t* __restrict__ zz_sad, const uint32_t i) {
    if (!P.zz_skip_th || !zz_sad || P.level == 2) return false;
    return zz_sad[i / ((uint32_t)P.num_hme_sa_w * P.num_hme_sa_h)] < P.zz_skip_th; // [ref][sb] = item / regions
}
__global__ __launch_bounds__(256) void hme_descs_kernel(const SvtHipHmeLevelParams P, const int16_t* __restrict__ prev_sc, const uint32_t* __restrict__ zz_sad,
                                                        SvtHipSadLoopDesc* __restrict__ descs, HmeItem* __restrict__ items, const uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    SvtHipSadLoopDesc d;
    int16_t           ox, oy;
    hme_item_geometry(P, i, P.level ? prev_sc[2 * i] : (int16_t)0, P.level ? prev_sc[2 * i + 1] : (int16_t)0, d, ox, oy);
    if (hme_item_skipped(P, zz_sad, i)) { d.search_area_width = 1; d.search_area_height = 1; } // result overwritten by the rescale kernel
    descs[i] = d;
    items[i] = HmeItem{ox, oy};
}

__global__ __launch_bounds__(256) void hme_post_kernel(const SvtHipHmeLevelParams P, const uint32_t* __restrict__ zz_sad, const SvtHipSadLoopResult* __restrict__ res,
                                                       const HmeItem* __restrict__ items, const int sub_sampled, const int scale,
                                                       unsigned long long* __restrict__ sad_out, int16_t* __restrict__ sc_out, const uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (hme_item_skipped(P, zz_sad, i)) { sad_out[i] = 0; sc_out[2 * i] = 0; sc_out[2 * i + 1] = 0; return; }
    const SvtHipSadLoopResult r = res[i];
    const int16_t x = r.valid ? r.x_search_center : sc_out[2 * i], y = r.valid ? r.y_search_center : sc_out[2 * i + 1];
    sad_out[i]        = sub_sampled ? r.best_sad * 2 : r.best_sad;
    sc_out[2 * i]     = (int16_t)((int16_t)(x + items[i].sa_origin_x) * scale);
    sc_out[2 * i + 1] = (int16_t)((int16_t)(y + items[i].sa_origin_y) * scale);
}

inline uint32_t hme_items(const SvtHipHmeLevelParams* P) { return P->n_refs * P->sbs_x * P->sbs_y * P->num_hme_sa_w * P->num_hme_sa_h; }
struct HmeWs { size_t descs, res, keys, items, bytes; };
inline HmeWs hme_ws(uint32_t n) {
    HmeWs w;
    w.descs = 0;
    w.res   = svthip::align_up(w.descs + (size_t)n * sizeof(SvtHipSadLoopDesc), 256);
    w.keys  = svthip::align_up(w.res + (size_t)n * sizeof(SvtHipSadLoopResult), 256);
    w.items = svthip::align_up(w.keys + (size_t)n * 8, 256);
    w.bytes = svthip::align_up(w.items + (size_t)n * sizeof(HmeItem), 256);
    return w;
}

// ---- integer ME from the HME results.  One thread per SB walks its reference slots three times, as the reference's per-SB code does: final search
// centre per slot (set_final_seach_centre_sb), then hme_prune_ref_and_adjust_sr over all slots (needs the best HME SAD of the SB), then
// integer_search_b64's area geometry per slot -> SvtHipMeSearchDesc.  Item index = ref * n_sb + sb.
struct MeIntRec { // per item, between the phases of the probing form
    int16_t  cx, cy, w, h; // search centre (after check_00_center), area before the variance scaling
    uint8_t  live, check00, probe, hme_good; // hme_good: is_ref and the slot's HME SAD is below 24 * 24 (sr_adjustment 2, :1350-1351)
};
// PHASE 0: all in one kernel (no probes).  PHASE 1: up to the base area -> records (+ single-point descriptors for the variance probe).
// PHASE 2: records (+ probe tables) -> final descriptors.
template <int PHASE>
__global__ __launch_bounds__(64) void me_int_descs_kernel(const SvtHipMeIntegerSearchParams P, const unsigned long long* __restrict__ hme_sad,
                                                          const int16_t* __restrict__ hme_sc, uint8_t* __restrict__ do_ref, uint32_t* __restrict__ divisor,
                                                          const uint32_t* __restrict__ zz_sad, SvtHipMeSearchDesc* __restrict__ descs,
                                                          int16_t* __restrict__ sc_out, unsigned long long* __restrict__ sad_out,
                                                          MeIntRec* __restrict__ recs, const uint32_t* __restrict__ probe_sad, const uint32_t r_begin,
                                                          const uint32_t r_end) { // [r_begin, r_end): the slots whose descriptors are produced (PHASE 2; the others: all)
    const uint32_t n_sb = P.sbs_x * P.sbs_y, sb = blockIdx.x * blockDim.x + threadIdx.x;
    if (sb >= n_sb) return;
    int16_t            cx[8], cy[8];
    unsigned long long csad[8], best_all = 0xffffffffull; // slots HME never touched keep MAX_U32 (init_me_hme_data, :3061)
    for (uint32_t r = 0; r < P.n_refs && PHASE != 2; r++) {
        const uint32_t i = r * n_sb + sb;
        if (P.list1_no_hme && r >= P.n_refs_list0) { // base layer: no HME for list 1 (:2211, :2362-2373): centre (0, 0); hmeMvSad still holds the last list-0 slot's value
            cx[r] = 0; cy[r] = 0; csad[r] = r ? csad[r - 1] : 0ull;
            sc_out[2 * i] = 0; sc_out[2 * i + 1] = 0; sad_out[i] = csad[r];
            best_all = csad[r] < best_all ? csad[r] : best_all;
            continue;
        }
        // set_final_seach_centre_sb: first strictly smaller SAD, regions in sr_h-outer / sr_w-inner order
        const unsigned long long* ps = hme_sad + (size_t)i * P.regions;
        const int16_t*            pc = hme_sc + (size_t)i * P.regions * 2;
        unsigned long long best = ps[0];
        int16_t            x = pc[0], y = pc[1];
        for (uint32_t k = 1; k < P.regions; k++)
            if (ps[k] < best) { best = ps[k]; x = pc[2 * k]; y = pc[2 * k + 1]; }
        cx[r] = x; cy[r] = y; csad[r] = best;
        sc_out[2 * i] = x; sc_out[2 * i + 1] = y; sad_out[i] = best;
        best_all = best < best_all ? best : best_all;
    }
    for (uint32_t r = r_begin; r < r_end; r++) {
        const uint32_t i = r * n_sb + sb;
        const int      b64_origin_x = (int)(sb % P.sbs_x) * 64, b64_origin_y = (int)(sb / P.sbs_x) * 64;
        const int16_t  pad_width = 63, pad_height = 63, org_x = (int16_t)b64_origin_x, org_y = (int16_t)b64_origin_y;
        const int      picture_width = (int16_t)P.aligned_width, picture_height = (int16_t)P.aligned_height;
        int16_t x_search_center, y_search_center, search_area_width, search_area_height;
        bool    live;
        if (PHASE != 2) {
        const size_t dri = (size_t)sb * 8 + (r < P.n_refs_list0 ? 0 : 4) + P.ref_pic_index[r]; // search_results[list][ref] layout, as svt_hip_me_results_batch
        live = do_ref ? do_ref[dri] != 0 : true;
        if (P.tf_me_exit_th && csad[0] < P.tf_me_exit_th) live = false; // ME_MCTF: search_results[0][0].hme_sad below tf_me_exit_th ends the SB before the search (:3109-3113)
        // hme_prune_ref_and_adjust_sr: references (other than the first of each list) whose HME SAD is th % above the best are dropped ...
        if (P.hme_prune_enabled && P.ref_pic_index[r] != 0 && (csad[r] - best_all) * 100 > (unsigned long long)P.prune_ref_if_hme_sad_dev_bigger_than_th * best_all) {
            live = false;
            if (do_ref) do_ref[dri] = 0;
        }
        // ... and the ME search range shrinks when the HME result is stationary or already good
        uint32_t div = (!P.sr_adjustment && divisor) ? divisor[(size_t)sb * P.n_refs + r] : 1u;
        if (P.sr_adjustment) {
            const int ax = cx[r] < 0 ? -cx[r] : cx[r], ay = cy[r] < 0 ? -cy[r] : cy[r];
            if (ax <= P.reduce_me_sr_based_on_mv_length_th && ay <= P.reduce_me_sr_based_on_mv_length_th && csad[r] < P.stationary_hme_sad_abs_th)
                div = P.stationary_me_sr_divisor;
            else if (csad[r] < P.reduce_me_sr_based_on_hme_sad_abs_th)
                div = P.me_sr_divisor_for_low_hme_sad;
            if (divisor) divisor[(size_t)sb * P.n_refs + r] = div;
        }
        x_search_center = cx[r]; y_search_center = cy[r];
        search_area_width = P.sa_min_width; search_area_height = P.sa_min_height;
        {
            const int w = search_area_width * P.dist[r], h = search_area_height * P.dist[r];
            search_area_width  = (int16_t)(w < (uint16_t)P.sa_max_width ? w : (uint16_t)P.sa_max_width);
            search_area_height = (int16_t)(h < (uint16_t)P.sa_max_height ? h : (uint16_t)P.sa_max_height);
        }
        if (P.mv_adj_enabled && (!P.mv_adj_nearest_ref_only || P.ref_pic_index[r] == 0)) {
            if ((x_search_center < 0 ? -x_search_center : x_search_center) > P.mv_adj_mv_size_th) search_area_width = (int16_t)(search_area_width * P.mv_adj_sa_multiplier);
            if ((y_search_center < 0 ? -y_search_center : y_search_center) > P.mv_adj_mv_size_th) search_area_height = (int16_t)(search_area_height * P.mv_adj_sa_multiplier);
        }
        {
            const uint32_t w = (uint32_t)search_area_width / div, h = (uint32_t)search_area_height / div; // unsigned division, as the reference
            search_area_width  = (int16_t)(((w > 1 ? w : 1) + 7) & ~0x07u);
            search_area_height = (int16_t)(h > 3 ? h : 3);
        }
        if (P.me_early_exit_th && zz_sad[i] < P.me_early_exit_th / 6) { search_area_width = 1; search_area_height = 1; } // :1322-1327
        if (PHASE == 1) {
            MeIntRec rec;
            rec.cx = x_search_center; rec.cy = y_search_center; rec.w = search_area_width; rec.h = search_area_height;
            rec.live = live; rec.check00 = live && !P.me_early_exit_th && P.is_ref && (x_search_center != 0 || y_search_center != 0); rec.probe = 0;
            rec.hme_good = P.is_ref && csad[r] < 24 * 24;
            recs[i] = rec;
            continue;
        }
        } else { // PHASE 2: centre possibly replaced by check_00_center, then the variance scaling of the area (:1388-1420)
            const MeIntRec rec = recs[i];
            x_search_center = rec.cx; y_search_center = rec.cy; search_area_width = rec.w; search_area_height = rec.h; live = rec.live;
            if (rec.probe) {
                const uint32_t* t = probe_sad + (size_t)i * 85;
                const uint32_t  mean = t[0] / 64;
                uint32_t        ssq = 0;
                for (int k = 0; k < 64; k++) { const int32_t diff = (int32_t)t[21 + k] - (int32_t)mean; ssq += (uint32_t)(diff * diff); }
                const uint32_t var = ssq / 64;
                if (var > P.me_sr_mult2_th) {
                    const int w = search_area_width * 3 / 2, h = search_area_height * 3 / 2;
                    search_area_width  = (int16_t)(((w > 1 ? w : 1) + 7) & ~0x7);
                    search_area_height = (int16_t)(h > 1 ? h : 1);
                }
                if (var < P.me_sr_div4_th) {
                    const int w = search_area_width >> 2, h = search_area_height >> 2;
                    search_area_width  = (int16_t)(((w > 1 ? w : 1) + 7) & ~0x7);
                    search_area_height = (int16_t)((h > 1 ? h : 1) > 3 ? (h > 1 ? h : 1) : 3);
                } else if (var < P.me_sr_div2_th) {
                    const int w = search_area_width >> 1 < search_area_width ? search_area_width >> 1 : search_area_width;
                    const int h = search_area_height >> 1 < search_area_height ? search_area_height >> 1 : search_area_height;
                    search_area_width  = (int16_t)((w + 7) & ~0x7);
                    search_area_height = (int16_t)(h > 3 ? h : 3);
                }
            }
        }
        int16_t x_search_area_origin = (int16_t)(x_search_center - (search_area_width >> 1));
        int16_t y_search_area_origin = (int16_t)(y_search_center - (search_area_height >> 1));
        // origin and size are corrected by separate conditionals, the size one evaluated with the corrected origin (:1462-1467): the left / top
        // correction therefore never shrinks the area
        x_search_area_origin = (int16_t)(((org_x + x_search_area_origin) < -pad_width) ? -pad_width - org_x : x_search_area_origin);
        search_area_width    = (int16_t)(((org_x + x_search_area_origin) < -pad_width) ? search_area_width - (-pad_width - (org_x + x_search_area_origin)) : search_area_width);
        x_search_area_origin = (int16_t)(((org_x + x_search_area_origin) > picture_width - 1) ? x_search_area_origin - ((org_x + x_search_area_origin) - (picture_width - 1))
                                                                                                : x_search_area_origin);
        if ((org_x + x_search_area_origin + search_area_width) > picture_width) {
            const int w = search_area_width - ((org_x + x_search_area_origin + search_area_width) - picture_width);
            search_area_width = (int16_t)(w > 1 ? w : 1);
        }
        search_area_width    = (int16_t)(search_area_width < 8 ? search_area_width : search_area_width & ~0x07);
        y_search_area_origin = (int16_t)(((org_y + y_search_area_origin) < -pad_height) ? -pad_height - org_y : y_search_area_origin);
        search_area_height   = (int16_t)(((org_y + y_search_area_origin) < -pad_height) ? search_area_height - (-pad_height - (org_y + y_search_area_origin)) : search_area_height);
        y_search_area_origin = (int16_t)(((org_y + y_search_area_origin) > picture_height - 1) ? y_search_area_origin - ((org_y + y_search_area_origin) - (picture_height - 1))
                                                                                                 : y_search_area_origin);
        if ((org_y + y_search_area_origin + search_area_height) > picture_height) {
            const int h = search_area_height - ((org_y + y_search_area_origin + search_area_height) - picture_height);
            search_area_height = (int16_t)(h > 1 ? h : 1);
        }
        if (!live) { // not searched by the reference: placeholder
            x_search_area_origin = y_search_area_origin = 0;
            search_area_width = search_area_height = 1;
        }
        SvtHipMeSearchDesc d;
        d.src_off    = P.src_off + (uint64_t)b64_origin_y * P.src_stride + (uint64_t)b64_origin_x;
        d.ref_off    = P.ref_off[r] + (uint64_t)((long long)((int)P.ref_org_y + b64_origin_y + y_search_area_origin) * (long long)P.ref_stride +
                                                 (long long)((int)P.ref_org_x + b64_origin_x + x_search_area_origin));
        d.src_stride = P.src_stride;
        d.ref_stride = P.ref_stride;
        d.x_search_area_origin = x_search_area_origin;
        d.y_search_area_origin = y_search_area_origin;
        d.search_area_width    = (uint16_t)search_area_width;
        d.search_area_height   = (uint16_t)search_area_height;
        descs[i] = d;
    }
}

// sub-sampled 64 x h SAD of the SB against the reference at displacement (dx, dy): svt_nxm_sad_kernel(src, stride << 1, ref, stride << 1, h >> 1, w), one wave
__device__ __forceinline__ uint32_t sb_sub_sad(const uint8_t* __restrict__ s, const uint32_t ss, const uint8_t* __restrict__ f, const uint32_t fs, const uint32_t bw,
                                               const uint32_t bh, const uint32_t l) {
    uint32_t sad = 0;
    for (uint32_t y = l >> 1; y < (bh >> 1); y += 32)
        for (uint32_t x = (l & 1) * 32; x < bw && x < (l & 1) * 32 + 32; x++) {
            const int d = (int)s[(size_t)(2 * y) * ss + x] - (int)f[(size_t)(2 * y) * fs + x];
            sad += (uint32_t)(d < 0 ? -d : d);
        }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) sad += (uint32_t)__shfl_xor((int)sad, m);
    return sad;
}
// check_00_center (:1139-1206), one wave per item flagged by phase 1: clip the HME centre to the picture + 63, compare its sub-sampled SAD with the
// zero-motion one, keep (0, 0) when that is not worse; then decide whether the variance probe applies and emit its single-point descriptor.
__global__ __launch_bounds__(256) void me_int_probe_kernel(const SvtHipMeIntegerSearchParams P, const uint8_t* __restrict__ src_base, const uint8_t* __restrict__ ref_base,
                                                           MeIntRec* __restrict__ recs, SvtHipMeSearchDesc* __restrict__ probe_descs, const uint32_t i0, const uint32_t n,
                                                           const uint32_t* __restrict__ slot0_sad) { // items [i0, n); slot0_sad: the first slot's final tables (sr_adjustment 2, i0 > 0)
    const uint32_t i = i0 + blockIdx.x * 4 + (threadIdx.x >> 6), l = threadIdx.x & 63;
    if (i >= n) return;
    const uint32_t n_sb = P.sbs_x * P.sbs_y, sb = i % n_sb, r = i / n_sb;
    const uint32_t fx = (sb % P.sbs_x) * 64, fy = (sb / P.sbs_x) * 64;
    const uint32_t bw = P.aligned_width - fx < 64 ? P.aligned_width - fx : 64, bh = P.aligned_height - fy < 64 ? P.aligned_height - fy : 64;
    MeIntRec rec = recs[i];
    unsigned long long best_hme_sad = ~0ull;
    if (rec.check00) {
        const int16_t org_x = (int16_t)fx, org_y = (int16_t)fy, pad = 63, ref_w = (int16_t)P.ref_width, ref_h = (int16_t)P.ref_height;
        int16_t x = rec.cx, y = rec.cy;
        x = (int16_t)(((org_x + x) < -pad) ? -pad - org_x : x);
        x = (int16_t)(((org_x + x) > ref_w - 1) ? x - ((org_x + x) - (ref_w - 1)) : x);
        y = (int16_t)(((org_y + y) < -pad) ? -pad - org_y : y);
        y = (int16_t)(((org_y + y) > ref_h - 1) ? y - ((org_y + y) - (ref_h - 1)) : y);
        const uint8_t* s  = src_base + P.src_off + (size_t)fy * P.src_stride + fx;
        const uint8_t* f0 = ref_base + P.ref_off[r] + (size_t)(P.ref_org_y + fy) * P.ref_stride + P.ref_org_x + fx;
        const uint8_t* f1 = ref_base + P.ref_off[r] + (size_t)((int)(P.ref_org_y + fy) + y) * P.ref_stride + (int)(P.ref_org_x + fx) + x;
        const uint32_t zero = sb_sub_sad(s, P.src_stride, f0, P.ref_stride, bw, bh, l) << 1, hme = sb_sub_sad(s, P.src_stride, f1, P.ref_stride, bw, bh, l) << 1;
        if (zero <= hme) { x = 0; y = 0; } // MIN(zero cost, hme cost) == zero cost
        rec.cx = x; rec.cy = y;
        best_hme_sad = hme; // check_00_center returns the SAD at the (clipped) HME centre
    }
    if (P.sr_adjustment == 2 && !P.me_early_exit_th) { // (:1349-1364; inside the branch without me_early_exit_th)
        const int16_t h_before = rec.h;
        const bool    hme_is_accurate = !(rec.check00 && rec.cx == 0 && rec.cy == 0);
        if ((hme_is_accurate && best_hme_sad < 24 * 24) || rec.hme_good) rec.h = (int16_t)(rec.h / 2);
        if (r > 0 && slot0_sad[(size_t)sb * 85] < 5000 && rec.h == h_before) { rec.h = (int16_t)(rec.h >> 1); rec.w = (int16_t)(rec.w >> 1); } // (the width was not touched before)
    }
    rec.probe = rec.live && P.me_8x8_var_enabled && (int)rec.w * (int)rec.h > 24;
    if (l == 0) {
        recs[i] = rec;
        SvtHipMeSearchDesc d;
        d.src_off    = P.src_off + (uint64_t)fy * P.src_stride + fx;
        d.ref_off    = P.ref_off[r] + (uint64_t)((long long)((int)(P.ref_org_y + fy) + (rec.probe ? rec.cy : 0)) * (long long)P.ref_stride +
                                                 (long long)((int)(P.ref_org_x + fx) + (rec.probe ? rec.cx : 0)));
        d.src_stride = P.src_stride; d.ref_stride = P.ref_stride;
        d.x_search_area_origin = rec.probe ? rec.cx : 0; d.y_search_area_origin = rec.probe ? rec.cy : 0;
        d.search_area_width = 1; d.search_area_height = 1;
        probe_descs[i] = d;
    }
}
// the probe's single point was searched first and is never re-initialised (:1366, :1398-1404): it stays the winner unless the area search is strictly better
__global__ __launch_bounds__(256) void me_int_merge_kernel(const MeIntRec* __restrict__ recs, const uint32_t* __restrict__ probe_sad, const uint32_t* __restrict__ probe_mv,
                                                           uint32_t* __restrict__ best_sad, uint32_t* __restrict__ best_mv, const uint32_t n) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * 85) return;
    if (!recs[t / 85].probe) return;
    if (probe_sad[t] <= best_sad[t]) { best_sad[t] = probe_sad[t]; best_mv[t] = probe_mv[t]; }
}

// init_zz_sad: one wave per (reference, SB): lane = (row pair of the sub-sampled block, half row); sub-sampled SAD at the co-located position
__global__ __launch_bounds__(256) void me_zz_sad_kernel(const SvtHipMeIntegerSearchParams P, const uint8_t* __restrict__ src_base, const uint8_t* __restrict__ ref_base,
                                                        uint32_t* __restrict__ zz_out, const uint32_t n) {
    const uint32_t i = blockIdx.x * 4 + (threadIdx.x >> 6), l = threadIdx.x & 63;
    if (i >= n) return;
    const uint32_t n_sb = P.sbs_x * P.sbs_y, sb = i % n_sb, r = i / n_sb;
    const uint32_t fx = (sb % P.sbs_x) * 64, fy = (sb / P.sbs_x) * 64;
    const uint32_t bw = P.aligned_width - fx < 64 ? P.aligned_width - fx : 64, bh = P.aligned_height - fy < 64 ? P.aligned_height - fy : 64;
    const uint8_t* s = src_base + P.src_off + (size_t)fy * P.src_stride + fx;
    const uint8_t* f = ref_base + P.ref_off[r] + (size_t)(P.ref_org_y + fy) * P.ref_stride + P.ref_org_x + fx;
    uint32_t sad = 0;
    for (uint32_t y = l >> 1; y < (bh >> 1); y += 32) // rows 0, 2, 4, ... of the block (stride << 1, height >> 1)
        for (uint32_t x = (l & 1) * 32; x < bw && x < (l & 1) * 32 + 32; x++) {
            const int d = (int)s[(size_t)(2 * y) * P.src_stride + x] - (int)f[(size_t)(2 * y) * P.ref_stride + x];
            sad += (uint32_t)(d < 0 ? -d : d);
        }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) sad += (uint32_t)__shfl_xor((int)sad, m);
    if (l == 0) zz_out[i] = (P.list1_no_hme && r >= P.n_refs_list0) ? 0xffffffffu : ((sad << 1) * 64u * 64u) / (bw * bh); // (:2390: list 1 keeps init_me_hme_data's ~0 at the base layer)
}

// upper bounds of the area the geometry above can produce (host side, for the search kernel's tile / workspace sizing)
inline void me_int_max_area(const SvtHipMeIntegerSearchParams* P, uint32_t& mw, uint32_t& mh) {
    uint32_t maxd = 1;
    for (uint32_t r = 0; r < P->n_refs; r++) maxd = P->dist[r] > maxd ? P->dist[r] : maxd;
    uint32_t w = (uint32_t)P->sa_min_width * maxd, h = (uint32_t)P->sa_min_height * maxd;
    w = w < (uint16_t)P->sa_max_width ? w : (uint16_t)P->sa_max_width;
    h = h < (uint16_t)P->sa_max_height ? h : (uint16_t)P->sa_max_height;
    if (P->mv_adj_enabled) { w *= P->mv_adj_sa_multiplier; h *= P->mv_adj_sa_multiplier; }
    mw = ((w > 1 ? w : 1) + 7) & ~7u;
    mh = h > 3 ? h : 3;
    if (P->me_8x8_var_enabled) { mw = ((mw * 3 / 2) + 7) & ~7u; mh = mh * 3 / 2; } // me_sr_mult2_th
}

} // namespace

extern "C" {

size_t svt_hip_hme_level_workspace(const SvtHipHmeLevelParams* params) { return hme_ws(hme_items(params)).bytes; }

void svt_hip_hme_level_batch(const SvtHipHmeLevelParams* params, const uint8_t* src_base, const uint8_t* ref_base, const int16_t* prev_sc,
                             const uint32_t* zz_sad, uint64_t* sad_out, int16_t* sc_out, void* workspace, void* stream) {
    if (params->l0_mv_th_min || params->l0_mv_th_max) { // (the other slots' level-0 areas depend on slot 0's result of the same SB: only the chain form orders that)
        fprintf(stderr, "libsvtav1_hip: svt_hip_hme_level_batch: level-0 resizing from list 0's motion is built in svt_hip_hme_chain_batch only\n");
        abort();
    }
    svthip::ensure_device();
    const uint32_t n = hme_items(params);
    if (n == 0) return;
    if (params->n_refs > 8 || params->level > 2) { fprintf(stderr, "libsvtav1_hip: svt_hip_hme_level_batch: bad parameters\n"); abort(); }
    const HmeWs        w = hme_ws(n);
    uint8_t*           ws = (uint8_t*)workspace;
    SvtHipSadLoopDesc* descs = (SvtHipSadLoopDesc*)(ws + w.descs);
    SvtHipSadLoopResult* res = (SvtHipSadLoopResult*)(ws + w.res);
    HmeItem*           items = (HmeItem*)(ws + w.items);
    hipStream_t        st = (hipStream_t)stream;
    hipLaunchKernelGGL(hme_descs_kernel, dim3((n + 255) / 256), dim3(256), 0, st, *params, prev_sc, zz_sad, descs, items, n);
    SVT_LAUNCH_CHECK();
    const int      shift = params->level == 0 ? 2 : (params->level == 1 ? 1 : 0);
    const uint32_t blk = 64u >> shift, step = params->sub_sampled ? 2 : 1;
    int maw = params->sa_width, mah = params->sa_height;
    if (params->per_ref_area) {
        maw = mah = 1;
        for (uint32_t r = 0; r < params->n_refs; r++) {
            maw = params->sa_width_ref[r] > maw ? params->sa_width_ref[r] : maw;
            mah = params->sa_height_ref[r] > mah ? params->sa_height_ref[r] : mah;
        }
    }
    svt_hip_sad_loop_batch(src_base, ref_base, descs, n, (uint32_t)((maw + 7) & ~7), (uint32_t)mah, blk, blk / step, (int)step, res,
                           (uint64_t*)(ws + w.keys), stream);
    hipLaunchKernelGGL(hme_post_kernel, dim3((n + 255) / 256), dim3(256), 0, st, *params, zz_sad, (const SvtHipSadLoopResult*)res, (const HmeItem*)items,
                       (int)params->sub_sampled, params->level == 0 ? 4 : (params->level == 1 ? 2 : 1), (unsigned long long*)sad_out, sc_out, n);
    SVT_LAUNCH_CHECK();
}


size_t svt_hip_me_integer_search_workspace(const SvtHipMeIntegerSearchParams* params) {
    const uint32_t n = params->n_refs * params->sbs_x * params->sbs_y;
    uint32_t mw, mh;
    me_int_max_area(params, mw, mh);
    return 2 * svthip::align_up((size_t)n * sizeof(SvtHipMeSearchDesc), 256) + svthip::align_up((size_t)n * sizeof(MeIntRec), 256) +
           2 * svthip::align_up((size_t)n * 85 * 4, 256) + svt_hip_me_fullpel_search_workspace(n, mw, mh);
}

void svt_hip_me_zz_sad_batch(const SvtHipMeIntegerSearchParams* params, const uint8_t* src_base, const uint8_t* ref_base, uint32_t* zz_out, void* stream) {
    svthip::ensure_device();
    const uint32_t n = params->n_refs * params->sbs_x * params->sbs_y;
    if (n == 0) return;
    hipLaunchKernelGGL(me_zz_sad_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, *params, src_base, ref_base, zz_out, n);
    SVT_LAUNCH_CHECK();
}

void svt_hip_me_integer_search_batch(const SvtHipMeIntegerSearchParams* params, const uint8_t* src_base, const uint8_t* ref_base, const uint64_t* hme_sad,
                                     const int16_t* hme_sc, uint8_t* do_ref, uint32_t* divisor, const uint32_t* zz_sad, uint32_t* best_sad, uint32_t* best_mv,
                                     int16_t* sc_out, uint64_t* sad_out, void* workspace, void* stream) {
    svthip::ensure_device();
    const uint32_t n = params->n_refs * params->sbs_x * params->sbs_y;
    if (n == 0) return;
    if (params->n_refs > 8 || params->regions == 0 || params->sr_adjustment > 2 || (params->me_early_exit_th && !zz_sad)) {
        fprintf(stderr, "libsvtav1_hip: svt_hip_me_integer_search_batch: bad parameters\n");
        abort();
    }
    uint8_t* w = (uint8_t*)workspace;
    SvtHipMeSearchDesc* descs  = (SvtHipMeSearchDesc*)w;  w += svthip::align_up((size_t)n * sizeof(SvtHipMeSearchDesc), 256);
    SvtHipMeSearchDesc* pdescs = (SvtHipMeSearchDesc*)w;  w += svthip::align_up((size_t)n * sizeof(SvtHipMeSearchDesc), 256);
    MeIntRec*           recs   = (MeIntRec*)w;            w += svthip::align_up((size_t)n * sizeof(MeIntRec), 256);
    uint32_t*           psad   = (uint32_t*)w;            w += svthip::align_up((size_t)n * 85 * 4, 256);
    uint32_t*           pmv    = (uint32_t*)w;            w += svthip::align_up((size_t)n * 85 * 4, 256);
    void*               ws2    = w;
    const uint32_t n_sb = params->sbs_x * params->sbs_y;
    const dim3     gsb((n_sb + 63) / 64), bsb(64);
    hipStream_t    st = (hipStream_t)stream;
    uint32_t mw, mh;
    me_int_max_area(params, mw, mh);
    const bool sr2     = params->sr_adjustment == 2 && !params->me_early_exit_th; // two more area rules, one of which reads the first slot's final SADs
    const bool probing = (params->is_ref && !params->me_early_exit_th) || params->me_8x8_var_enabled || sr2;
    if (!probing) {
        hipLaunchKernelGGL(me_int_descs_kernel<0>, gsb, bsb, 0, st, *params, (const unsigned long long*)hme_sad, hme_sc, do_ref, divisor, zz_sad, descs, sc_out,
                           (unsigned long long*)sad_out, (MeIntRec*)nullptr, (const uint32_t*)nullptr, 0u, params->n_refs);
        SVT_LAUNCH_CHECK();
        svt_hip_me_fullpel_search_batch(src_base, ref_base, descs, n, mw, mh, params->sub_sad, best_sad, best_mv, ws2, stream);
        return;
    }
    hipLaunchKernelGGL(me_int_descs_kernel<1>, gsb, bsb, 0, st, *params, (const unsigned long long*)hme_sad, hme_sc, do_ref, divisor, zz_sad, descs, sc_out,
                       (unsigned long long*)sad_out, recs, (const uint32_t*)nullptr, 0u, params->n_refs);
    SVT_LAUNCH_CHECK();
    // item range [i0, i1) = slots [r0, r1): everything at once, or -- sr_adjustment 2 with more than one slot -- the first slot through to its final tables, then the others
    auto pass = [&](const uint32_t r0, const uint32_t r1) {
        const uint32_t i0 = r0 * n_sb, i1 = r1 * n_sb, m = i1 - i0;
        hipLaunchKernelGGL(me_int_probe_kernel, dim3((m + 3) / 4), dim3(256), 0, st, *params, src_base, ref_base, recs, pdescs, i0, i1, (const uint32_t*)best_sad);
        SVT_LAUNCH_CHECK();
        if (params->me_8x8_var_enabled)
            svt_hip_me_fullpel_search_batch(src_base, ref_base, pdescs + i0, m, 1, 1, params->sub_sad, psad + (size_t)i0 * 85, pmv + (size_t)i0 * 85, ws2, stream);
        hipLaunchKernelGGL(me_int_descs_kernel<2>, gsb, bsb, 0, st, *params, (const unsigned long long*)hme_sad, hme_sc, do_ref, divisor, zz_sad, descs, sc_out,
                           (unsigned long long*)sad_out, recs, (const uint32_t*)psad, r0, r1);
        SVT_LAUNCH_CHECK();
        svt_hip_me_fullpel_search_batch(src_base, ref_base, descs + i0, m, mw, mh, params->sub_sad, best_sad + (size_t)i0 * 85, best_mv + (size_t)i0 * 85, ws2, stream);
        if (params->me_8x8_var_enabled) {
            hipLaunchKernelGGL(me_int_merge_kernel, dim3((m * 85 + 255) / 256), dim3(256), 0, st, (const MeIntRec*)recs + i0, (const uint32_t*)psad + (size_t)i0 * 85,
                               (const uint32_t*)pmv + (size_t)i0 * 85, best_sad + (size_t)i0 * 85, best_mv + (size_t)i0 * 85, m);
            SVT_LAUNCH_CHECK();
        }
    };
    if (sr2 && params->n_refs > 1) { pass(0, 1); pass(1, params->n_refs); }
    else pass(0, params->n_refs);
}

} // extern "C"

SVT_HIP_DEFINE_WARM(hme) // (svt_hip_warmup loads this translation unit's code object at encoder initialisation: svt_hip_common.h)
