/*
 * ref_lr_search.c -- TEST INFRASTRUCTURE, built only into oracle/_ref/libsvtref_me.so (make -C oracle ref).
 *
 * The per-unit half of the reference's loop-restoration search -- search_norestore_seg, search_wiener_seg, search_sgrproj_seg and everything below them
 * (wiener_decompose_sep_sym, finalize_sym_filter, compute_score, finer_tile_search_wiener_seg, search_selfguided_restoration,
 * finer_search_pixel_proj_error, try_restoration_unit_seg; Codec/restoration_pick.c:56-1420) -- is `static`.  This translation unit compiles that reference
 * source file WHERE IT LIES (the #include below resolves through -I$(REF)/Source/Lib/Codec; nothing is copied) and adds plain-C entry points that build the
 * RestSearchCtxt / Av1Common / PictureControlSet fields those functions read, exactly as restoration_seg_search does (:1448-1527), and run them for the
 * units of ONE plane.
 */
#include "restoration_pick.c"

typedef struct RefLrSearchParams { /* = OracleLrSearchParams / SvtHipLrSearchParams */
    const void *dgd, *src;
    uint32_t    dgd_stride, src_stride, width, height, unit_size;
    uint8_t     ss_y, highbd, bit_depth;
    uint8_t     wn_enabled, wiener_win, wn_use_refinement, wn_max_one_refinement_step;
    uint8_t     sg_enabled, sg_start_ep, sg_end_ep, sg_ep_inc, sg_refine;
    uint8_t     pad[3];
} RefLrSearchParams;
typedef struct RefLrSearchUnit {
    int64_t sse[3];
    int16_t vfilter[8], hfilter[8];
    int32_t ep, xqd[2], pad;
} RefLrSearchUnit;
typedef struct RefLrPrevUnit { int32_t use; int16_t vfilter[8], hfilter[8]; } RefLrPrevUnit;

/* wiener_decompose_sep_sym + finalize_sym_filter x 2 + compute_score for one (M, H) pair */
int64_t ref_wiener_solve(int win, int64_t *M, int64_t *H, int16_t *vfilter /* [8] */, int16_t *hfilter /* [8] */, int32_t *vd /* [7] */, int32_t *hd /* [7] */) {
    wiener_decompose_sep_sym(win, M, H, vd, hd);
    memset(vfilter, 0, 16); memset(hfilter, 0, 16); /* (rui is memset in search_wiener_seg :1295) */
    finalize_sym_filter(win, vd, vfilter);
    finalize_sym_filter(win, hd, hfilter);
    return compute_score(win, M, H, vfilter, hfilter);
}

/* rects: n x {h_start, h_end, v_start, v_end} (what svt_aom_foreach_rest_unit_in_frame hands to the visitors) */
void ref_lr_search_plane(const RefLrSearchParams *P, const RefLrPrevUnit *prev, RefLrSearchUnit *out, const int32_t *rects, int n) {
    Av1Common               *cm   = calloc(1, sizeof(*cm));
    PictureControlSet       *pcs  = calloc(1, sizeof(*pcs));
    PictureParentControlSet *ppcs = calloc(1, sizeof(*ppcs));
    RestUnitSearchInfo      *rusi = calloc((size_t)n, sizeof(*rusi));
    Yv12BufferConfig         fts, src, dst;
    const int                plane = 0; /* (chroma differs only through ss_y, carried below, and the resolved wiener_win) */
    const int                bs    = P->highbd ? 2 : 1;
    memset(&fts, 0, sizeof(fts)); memset(&src, 0, sizeof(src)); memset(&dst, 0, sizeof(dst));
    uint8_t *trial = calloc((size_t)(P->height + 16) * (P->width + 64) * bs, 1);
    fts.buffers[0] = P->highbd ? CONVERT_TO_BYTEPTR(P->dgd) : (uint8_t *)P->dgd; fts.strides[0] = (int32_t)P->dgd_stride;
    src.buffers[0] = P->highbd ? CONVERT_TO_BYTEPTR(P->src) : (uint8_t *)P->src; src.strides[0] = (int32_t)P->src_stride;
    dst.buffers[0] = P->highbd ? CONVERT_TO_BYTEPTR(trial) : trial; dst.strides[0] = (int32_t)P->width + 64;
    fts.crop_widths[0] = src.crop_widths[0] = dst.crop_widths[0] = (int32_t)P->width;
    fts.crop_heights[0] = src.crop_heights[0] = dst.crop_heights[0] = (int32_t)P->height;
    cm->child_pcs = pcs; cm->frame_to_show = &fts;
    cm->use_highbitdepth = P->highbd; cm->bit_depth = P->bit_depth;
    cm->subsampling_x = cm->subsampling_y = 0; /* plane 0 */
    cm->use_boundaries_in_rest_search = 0;     /* enc_handle.c:4129 */
    cm->wn_filter_ctrls.enabled = P->wn_enabled; cm->wn_filter_ctrls.filter_tap_lvl = P->wiener_win == 7 ? 1 : (P->wiener_win == 5 ? 2 : 3);
    cm->wn_filter_ctrls.use_refinement = P->wn_use_refinement; cm->wn_filter_ctrls.max_one_refinement_step = P->wn_max_one_refinement_step;
    cm->wn_filter_ctrls.use_prev_frame_coeffs = prev != NULL;
    cm->sg_filter_ctrls.enabled = P->sg_enabled; cm->sg_filter_ctrls.step_range = 16;
    cm->sg_filter_ctrls.start_ep[0] = (int8_t)P->sg_start_ep; cm->sg_filter_ctrls.end_ep[0] = (int8_t)P->sg_end_ep;
    cm->sg_filter_ctrls.ep_inc[0] = (int8_t)P->sg_ep_inc; cm->sg_filter_ctrls.refine[0] = (int8_t)P->sg_refine;
    pcs->ppcs = ppcs; pcs->rest_search_mutex = svt_create_mutex();
    ppcs->frm_hdr.frame_type = INTER_FRAME;
    pcs->rst_info[0].unit_info = calloc((size_t)n, sizeof(RestorationUnitInfo));
    for (int u = 0; prev && u < n; u++)
        if (prev[u].use) {
            pcs->rst_info[0].unit_info[u].restoration_type = RESTORE_WIENER;
            memcpy(pcs->rst_info[0].unit_info[u].wiener_info.vfilter, prev[u].vfilter, 16);
            memcpy(pcs->rst_info[0].unit_info[u].wiener_info.hfilter, prev[u].hfilter, 16);
        }
    RestSearchCtxt rsc;
    memset(&rsc, 0, sizeof(rsc));
    init_rsc_seg(&fts, &src, cm, NULL, plane, rusi, &dst, &rsc);
    rsc.tmpbuf = (int32_t *)svt_aom_memalign(16, RESTORATION_TMPBUF_SIZE);
    rsc.tile_stripe0 = 0;
    const Av1PixelRect tile = {0, 0, (int32_t)P->width, (int32_t)P->height};
    /* the chroma row offset of the units enters through the rects; the stripe offset inside try_restoration_unit_seg only partitions rows (no boundaries) */
    for (int u = 0; u < n; u++) {
        RestorationTileLimits lim = {rects[4 * u], rects[4 * u + 1], rects[4 * u + 2], rects[4 * u + 3]};
        search_norestore_seg(&lim, &tile, u, &rsc);
        if (P->wn_enabled) search_wiener_seg(&lim, &tile, u, &rsc);
        if (P->sg_enabled) search_sgrproj_seg(&lim, &tile, u, &rsc);
        memset(&out[u], 0, sizeof(out[u]));
        out[u].sse[0] = rusi[u].sse[RESTORE_NONE];
        if (P->wn_enabled) {
            out[u].sse[1] = rusi[u].sse[RESTORE_WIENER];
            if (out[u].sse[1] != INT64_MAX) { memcpy(out[u].vfilter, rusi[u].wiener.vfilter, 16); memcpy(out[u].hfilter, rusi[u].wiener.hfilter, 16); }
        }
        if (P->sg_enabled) {
            out[u].sse[2] = rusi[u].sse[RESTORE_SGRPROJ];
            out[u].ep = rusi[u].sgrproj.ep; out[u].xqd[0] = rusi[u].sgrproj.xqd[0]; out[u].xqd[1] = rusi[u].sgrproj.xqd[1];
        }
    }
    svt_aom_free(rsc.tmpbuf);
    free(pcs->rst_info[0].unit_info); free(trial); free(rusi); free(ppcs); free(pcs); free(cm);
}
