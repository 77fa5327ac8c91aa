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

__global__ __launch_bounds__(256) void hme_descs_kernel(const SvtHipHmeLevelParams P, const int16_t* __restrict__ prev_sc, SvtHipSadLoopDesc* __restrict__ descs,
                                                        HmeItem* __restrict__ items, const uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    SvtHipSadLoopDesc d;
    int16_t           ox, oy;
    hme_item_geometry(P, i, P.level ? prev_sc[2 * i] : (int16_t)0, P.level ? prev_sc[2 * i + 1] : (int16_t)0, d, ox, oy);
    descs[i] = d;
    items[i] = HmeItem{ox, oy};
}

__global__ __launch_bounds__(256) void hme_post_kernel(const SvtHipSadLoopResult* __restrict__ res, const HmeItem* __restrict__ items, const int sub_sampled,
                                                       const int scale, unsigned long long* __restrict__ sad_out, int16_t* __restrict__ sc_out, const uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
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

// ---- integer ME from the HME results: final search centre per (reference, SB) + integer_search_b64's area geometry -> SvtHipMeSearchDesc
__global__ __launch_bounds__(256) void me_int_descs_kernel(const SvtHipMeIntegerSearchParams P, const unsigned long long* __restrict__ hme_sad,
                                                           const int16_t* __restrict__ hme_sc, const uint8_t* __restrict__ do_ref,
                                                           const uint32_t* __restrict__ divisor, SvtHipMeSearchDesc* __restrict__ descs,
                                                           int16_t* __restrict__ sc_out, unsigned long long* __restrict__ sad_out, const uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t n_sb = P.sbs_x * P.sbs_y, sb = i % n_sb, r = i / n_sb;
    // set_final_seach_centre_sb: first strictly smaller SAD, regions in sr_h-outer / sr_w-inner order
    const unsigned long long* ps = hme_sad + (size_t)i * P.regions;
    const int16_t*            pc = hme_sc + (size_t)i * P.regions * 2;
    unsigned long long best = ps[0];
    int16_t            x_search_center = pc[0], y_search_center = pc[1];
    for (uint32_t k = 1; k < P.regions; k++)
        if (ps[k] < best) { best = ps[k]; x_search_center = pc[2 * k]; y_search_center = pc[2 * k + 1]; }
    sc_out[2 * i] = x_search_center; sc_out[2 * i + 1] = y_search_center; sad_out[i] = best;

    const int      b64_origin_x = (int)(sb % P.sbs_x) * 64, b64_origin_y = (int)(sb / P.sbs_x) * 64;
    const int16_t  pad_width = 63, pad_height = 63, org_x = (int16_t)b64_origin_x, org_y = (int16_t)b64_origin_y;
    const int      picture_width = (int16_t)P.aligned_width, picture_height = (int16_t)P.aligned_height;
    const uint32_t div = divisor ? divisor[(size_t)sb * P.n_refs + r] : 1u;
    int16_t search_area_width = P.sa_min_width, search_area_height = P.sa_min_height;
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
    if (do_ref && !do_ref[(size_t)sb * P.n_refs + r]) { // not searched by the reference: placeholder
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
}

} // namespace

extern "C" {

size_t svt_hip_hme_level_workspace(const SvtHipHmeLevelParams* params) { return hme_ws(hme_items(params)).bytes; }

void svt_hip_hme_level_batch(const SvtHipHmeLevelParams* params, const uint8_t* src_base, const uint8_t* ref_base, const int16_t* prev_sc, uint64_t* sad_out,
                             int16_t* sc_out, void* workspace, void* stream) {
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
    hipLaunchKernelGGL(hme_descs_kernel, dim3((n + 255) / 256), dim3(256), 0, st, *params, prev_sc, descs, items, n);
    SVT_LAUNCH_CHECK();
    const int      shift = params->level == 0 ? 2 : (params->level == 1 ? 1 : 0);
    const uint32_t blk = 64u >> shift, step = params->sub_sampled ? 2 : 1;
    svt_hip_sad_loop_batch(src_base, ref_base, descs, n, (uint32_t)((params->sa_width + 7) & ~7), (uint32_t)params->sa_height, blk, blk / step, (int)step, res,
                           (uint64_t*)(ws + w.keys), stream);
    hipLaunchKernelGGL(hme_post_kernel, dim3((n + 255) / 256), dim3(256), 0, st, (const SvtHipSadLoopResult*)res, (const HmeItem*)items,
                       (int)params->sub_sampled, params->level == 0 ? 4 : (params->level == 1 ? 2 : 1), (unsigned long long*)sad_out, sc_out, n);
    SVT_LAUNCH_CHECK();
}


size_t svt_hip_me_integer_search_workspace(const SvtHipMeIntegerSearchParams* params) {
    const uint32_t n = params->n_refs * params->sbs_x * params->sbs_y;
    uint32_t mw, mh;
    me_int_max_area(params, mw, mh);
    return svthip::align_up((size_t)n * sizeof(SvtHipMeSearchDesc), 256) + svt_hip_me_fullpel_search_workspace(n, mw, mh);
}

void svt_hip_me_integer_search_batch(const SvtHipMeIntegerSearchParams* params, const uint8_t* src_base, const uint8_t* ref_base, const uint64_t* hme_sad,
                                     const int16_t* hme_sc, const uint8_t* do_ref, const uint32_t* divisor, uint32_t* best_sad, uint32_t* best_mv,
                                     int16_t* sc_out, uint64_t* sad_out, void* workspace, void* stream) {
    svthip::ensure_device();
    const uint32_t n = params->n_refs * params->sbs_x * params->sbs_y;
    if (n == 0) return;
    if (params->n_refs > 8 || params->regions == 0) { fprintf(stderr, "libsvtav1_hip: svt_hip_me_integer_search_batch: bad parameters\n"); abort(); }
    SvtHipMeSearchDesc* descs = (SvtHipMeSearchDesc*)workspace;
    void*               ws2   = (uint8_t*)workspace + svthip::align_up((size_t)n * sizeof(SvtHipMeSearchDesc), 256);
    hipLaunchKernelGGL(me_int_descs_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, *params, (const unsigned long long*)hme_sad, hme_sc, do_ref,
                       divisor, descs, sc_out, (unsigned long long*)sad_out, n);
    SVT_LAUNCH_CHECK();
    uint32_t mw, mh;
    me_int_max_area(params, mw, mh);
    svt_hip_me_fullpel_search_batch(src_base, ref_base, descs, n, mw, mh, params->sub_sad, best_sad, best_mv, ws2, stream);
}

} // extern "C"
