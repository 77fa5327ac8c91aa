// hme_geom.h -- search-area placement and clipping of hme_level_0 / hme_level_1 / hme_level_2 (Codec/motion_estimation.c:820-921, 923-1018,
// 1020-1116) for item i = ((ref * n_sb + sb) * num_hme_sa_h + sr_h) * num_hme_sa_w + sr_w, in the reference's int16_t arithmetic.  Shared by the
// per-level descriptor kernel (hme.hip) and the fused three-level kernel (sad.hip).
#pragma once
#include "../../include/svtav1_hip.h"

// l0_flags (level 0 with l0_mv_th_*): bit 0 = the width takes the (2 + ref index) divisor, bit 1 = the height does, bit 2 = both take (4 + ref index)
// (get_hme_l0_search_area :1809-1850)
__device__ __forceinline__ void hme_item_geometry(const SvtHipHmeLevelParams& P, const uint32_t i, const int16_t prev_x, const int16_t prev_y,
                                                  SvtHipSadLoopDesc& d, int16_t& out_origin_x, int16_t& out_origin_y, const int l0_flags = 0) {
    const uint32_t regions = (uint32_t)P.num_hme_sa_w * P.num_hme_sa_h, n_sb = P.sbs_x * P.sbs_y;
    const uint32_t sr = i % regions, rs = i / regions, sb = rs % n_sb, r = rs / n_sb;
    const int      sr_w = (int)(sr % P.num_hme_sa_w), sr_h = (int)(sr / P.num_hme_sa_w);
    const int      shift = P.level == 0 ? 2 : (P.level == 1 ? 1 : 0);
    const uint32_t fx = (sb % P.sbs_x) * 64, fy = (sb / P.sbs_x) * 64; // full-resolution SB origin
    const uint32_t b64_w = P.aligned_width - fx < 64 ? P.aligned_width - fx : 64, b64_h = P.aligned_height - fy < 64 ? P.aligned_height - fy : 64;
    const int16_t  org_x = (int16_t)((int16_t)fx >> shift), org_y = (int16_t)((int16_t)fy >> shift);
    const uint32_t block_width = b64_w >> shift, block_height = b64_h >> shift;

    int16_t       sa_width  = (int16_t)(((P.per_ref_area ? ((l0_flags & 4) ? P.sa_width_ref4[r] : (l0_flags & 1) ? P.sa_width_ref2[r] : P.sa_width_ref[r]) : P.sa_width) + 7) & ~0x07);
    int16_t       sa_height = P.per_ref_area ? ((l0_flags & 4) ? P.sa_height_ref4[r] : (l0_flags & 2) ? P.sa_height_ref2[r] : P.sa_height_ref[r]) : P.sa_height;
    const int16_t pad_width = (int16_t)(P.level == 2 ? 63 : (int)P.ref_org_x - 1), pad_height = (int16_t)(P.level == 2 ? 63 : (int)P.ref_org_y - 1);
    const int16_t ref_w = (int16_t)P.ref_width, ref_h = (int16_t)P.ref_height;
    int16_t       sa_origin_x, sa_origin_y;
    if (P.level == 0) {
        sa_origin_x = (int16_t)(-(int16_t)((sa_width * P.num_hme_sa_w) >> 1) + (int16_t)(sa_width * sr_w));
        sa_origin_y = (int16_t)(-(int16_t)((sa_height * P.num_hme_sa_h) >> 1) + (int16_t)(sa_height * sr_h));
    } else {
        sa_origin_x = (int16_t)(-(sa_width >> 1) + (int16_t)(prev_x >> P.prev_shift));
        sa_origin_y = (int16_t)(-(sa_height >> 1) + (int16_t)(prev_y >> P.prev_shift));
    }
    // clip to the reference picture, left / right / top / bottom in the reference's order
    if ((org_x + sa_origin_x) < -pad_width) {
        sa_origin_x = (int16_t)(-pad_width - org_x);
        sa_width    = (int16_t)(sa_width - (-pad_width - (org_x + sa_origin_x)));
    }
    if ((org_x + sa_origin_x) > ref_w - 1) sa_origin_x = (int16_t)(sa_origin_x - ((org_x + sa_origin_x) - (ref_w - 1)));
    if ((org_x + sa_origin_x + sa_width) > ref_w) {
        const int w = sa_width - ((org_x + sa_origin_x + sa_width) - ref_w);
        sa_width    = (int16_t)(w > 1 ? w : 1);
    }
    sa_width = (int16_t)(sa_width < 8 ? sa_width : sa_width & ~0x07);
    if ((org_y + sa_origin_y) < -pad_height) {
        sa_origin_y = (int16_t)(-pad_height - org_y);
        sa_height   = (int16_t)(sa_height - (-pad_height - (org_y + sa_origin_y)));
    }
    if ((org_y + sa_origin_y) > ref_h - 1) sa_origin_y = (int16_t)(sa_origin_y - ((org_y + sa_origin_y) - (ref_h - 1)));
    if ((org_y + sa_origin_y + sa_height) > ref_h) {
        const int h = sa_height - ((org_y + sa_origin_y + sa_height) - ref_h);
        sa_height   = (int16_t)(h > 1 ? h : 1);
    }
    const int16_t  x_tl = (int16_t)(((int16_t)P.ref_org_x + org_x) + sa_origin_x), y_tl = (int16_t)(((int16_t)P.ref_org_y + org_y) + sa_origin_y);
    const uint32_t index = (uint32_t)(x_tl + y_tl * (int)P.ref_stride);
    const uint32_t step = P.sub_sampled ? 2 : 1;
    d.src_off            = P.src_off + (uint64_t)org_y * P.src_stride + (uint64_t)org_x;
    d.ref_off            = P.ref_off[r] + index;
    d.src_stride         = P.src_stride * step;
    d.ref_stride         = P.ref_stride * step;
    d.src_stride_raw     = P.ref_stride;
    d.block_width        = (uint16_t)block_width;
    d.block_height       = (uint16_t)(block_height / step);
    d.search_area_width  = sa_width;
    d.search_area_height = sa_height;
    d.skip_search_line   = 0;
    d.pad[0] = d.pad[1] = d.pad[2] = 0;
    out_origin_x = sa_origin_x;
    out_origin_y = sa_origin_y;
}
