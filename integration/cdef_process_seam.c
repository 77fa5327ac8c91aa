/* cdef_process_seam.c -- REFERENCE-SIDE BINDING (what a maintainer of the reference adds; built into the reference encoder by oracle/Makefile for the identity / fps runs): the reference's CDEF process with the frame-apply seam of INTEGRATION.md §3.
 *
 * This translation unit IS Source/Lib/Codec/cdef_process.c of the reference (included below where it lies; nothing is copied).  The one change: the call
 *
 *     svt_av1_cdef_frame(scs, pcs);                                                                                      (cdef_process.c:458)
 *
 * is given a macro name for the duration of the #include and lands in seam_av1_cdef_frame() below.  With SVT_HIP_CDEF_SEAM unset (or the HIP library not loaded)
 * that function IS the reference call.  With SVT_HIP_CDEF_SEAM=1 the whole picture is filtered by ONE svt_hip_cdef_apply_host() call: the per-filter-block
 * strengths are read exactly as svt_av1_cdef_frame reads them (enc_cdef.c:385-395: frm_hdr->cdef_params.cdef_y_strength / cdef_uv_strength indexed by the
 * block's mbmi.cdef_strength), the 8x8 skip map comes from the reference's own svt_sb_compute_cdef_list, filter blocks the reference skips (all strengths zero,
 * or no unit to filter, :396-401) are marked skipped as a whole.  With SB 128 the reference's per-block dirinit rule applies and the seam declines.
 *
 * The strength SEARCH, cdef_seg_search (:106-345), is a `static` function of the same file, called once per segment at :443.  A plain macro would rename its
 * definition and its call alike; the macro below appends __COUNTER__ (unused anywhere else in this translation unit), so the DEFINITION becomes
 * cdef_seg_search_use0 -- the reference's body, untouched -- and the CALL becomes cdef_seg_search_use1, which is the seam: the first segment of a picture to
 * arrive searches ALL filter blocks of the three planes with one svt_hip_cdef_search_host() call (candidate lists = cdef_ctrls->default_first_pass_fs[] +
 * default_second_pass_fs[], the chroma list without the entries flagged -1) and fills pcs->mse_seg[0 / 1][fb][gi], pcs->skip_cdef_seg[fb] and
 * pcs->cdef_dir_data[fb] exactly as the reference's loops do; the other segments find the work done.  finish_cdef_search (the strength decision) is untouched.
 * SVT_HIP_CDEF_SEAM_STATS=<file> receives the counters.
 */
#define _GNU_SOURCE /* RTLD_DEFAULT */
#include "../integration/seam_cpu.h"
#include <dlfcn.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pcs.h"
#include "sequence_control_set.h"
#include "enc_cdef.h"
#include "svtav1_hip.h" /* include/svtav1_hip.h of this repository: the C ABI */

void    svt_av1_cdef_frame(SequenceControlSet *scs, PictureControlSet *pcs);
void    svt_aom_get_recon_pic(PictureControlSet *pcs, EbPictureBufferDesc **recon_ptr, bool is_highbd);
int32_t svt_sb_compute_cdef_list(PictureControlSet *pcs, const Av1Common *const cm, int32_t mi_row, int32_t mi_col, CdefList *dlist, BlockSize bs);

int svt_hip_seam_bind(unsigned long long picture_number); /* integration/enc_handle_binding.c: SVT_HIP_DEVICES sharding */
#include <time.h>
static double seam_ms_now(void) { struct timespec t_; clock_gettime(CLOCK_MONOTONIC, &t_); return 1e3 * (double)t_.tv_sec + 1e-6 * (double)t_.tv_nsec; }
static struct {
    pthread_mutex_t lock;
    int             mode;
    int (*apply_host)(const SvtHipCdefApplyHost *);  /* non-zero: the device path is off (svtav1_hip.h, error policy): the picture takes the reference's own function */
    int (*search_host)(const SvtHipCdefSearchHost *);
    unsigned long long us_stage; /* microseconds inside the two stage calls */
    PictureControlSet *done_pcs[64]; /* pictures whose search the seam has run, and how many of their segments have passed */
    uint64_t           done_num[64];
    uint32_t           seen[64];
    uint64_t n_pictures, n_fbs, n_declined, n_searched, n_search_fbs;
} D = {PTHREAD_MUTEX_INITIALIZER};

static void cdef_seam_stats(void) {
    const char *f = getenv("SVT_HIP_CDEF_SEAM_STATS");
    FILE       *o = f ? fopen(f, "w") : NULL;
    if (!o) return;
    fprintf(o, "ms_in_stage_calls %llu\n", (unsigned long long)(D.us_stage / 1000));
    fprintf(o, "pictures_filtered %llu\nfilter_blocks %llu\npictures_declined %llu\npictures_searched %llu\nfilter_blocks_searched %llu\n",
            (unsigned long long)D.n_pictures, (unsigned long long)D.n_fbs, (unsigned long long)D.n_declined, (unsigned long long)D.n_searched,
            (unsigned long long)D.n_search_fbs);
    fclose(o);
}
static void cdef_seam_init(void) {
    const char *e = getenv("SVT_HIP_CDEF_SEAM");
    if (!e || !atoi(e) || !getenv("SVT_HIP")) return;
    *(void **)&D.apply_host = dlsym(RTLD_DEFAULT, "svt_hip_cdef_apply_host");
    *(void **)&D.search_host = dlsym(RTLD_DEFAULT, "svt_hip_cdef_search_host");
    if (!D.apply_host || !D.search_host) { fprintf(stderr, "SVT_HIP_CDEF_SEAM: libsvtav1_hip is not loaded\n"); abort(); }
    atexit(cdef_seam_stats);
    fprintf(stderr, "SVT_HIP_CDEF_SEAM: CDEF is applied to a picture by one device call\n");
    D.mode = 1;
}
static int cdef_seam_on(void) {
    static pthread_once_t once = PTHREAD_ONCE_INIT;
    pthread_once(&once, cdef_seam_init);
    return D.mode;
}

static void seam_av1_cdef_frame_body(SequenceControlSet *scs, PictureControlSet *pcs) {
    if (!cdef_seam_on() || scs->super_block_size == 128 || av1_num_planes(&scs->seq_header.color_config) != 3) {
        if (D.mode) { pthread_mutex_lock(&D.lock); D.n_declined++; pthread_mutex_unlock(&D.lock); }
        svt_av1_cdef_frame(scs, pcs);
        return;
    }
    struct PictureParentControlSet *ppcs     = pcs->ppcs;
    Av1Common                      *cm       = ppcs->av1_cm;
    FrameHeader                    *frm_hdr  = &ppcs->frm_hdr;
    const bool                      is_16bit = scs->is_16bit_pipeline;
    EbPictureBufferDesc            *recon_pic;
    svt_aom_get_recon_pic(pcs, &recon_pic, is_16bit);
    const int32_t nvfb = (cm->mi_rows + MI_SIZE_64X64 - 1) / MI_SIZE_64X64, nhfb = (cm->mi_cols + MI_SIZE_64X64 - 1) / MI_SIZE_64X64, nfb = nvfb * nhfb;
    uint8_t      *skip = malloc((size_t)nvfb * 8 * nhfb * 8);
    int32_t      *str  = calloc((size_t)nfb * 4, sizeof(int32_t)), *pri_y = str, *sec_y = str + nfb, *pri_uv = str + 2 * nfb, *sec_uv = str + 3 * nfb;
    CdefList      dlist[MI_SIZE_64X64 * MI_SIZE_64X64];
    memset(skip, 1, (size_t)nvfb * 8 * nhfb * 8);
    uint64_t filtered = 0;
    for (int32_t fbr = 0; fbr < nvfb; fbr++)
        for (int32_t fbc = 0; fbc < nhfb; fbc++) { /* enc_cdef.c:385-401 */
            const int32_t s = pcs->mi_grid_base[MI_SIZE_64X64 * fbr * cm->mi_stride + MI_SIZE_64X64 * fbc]->mbmi.cdef_strength;
            int32_t level = frm_hdr->cdef_params.cdef_y_strength[s] / CDEF_SEC_STRENGTHS, sec = frm_hdr->cdef_params.cdef_y_strength[s] % CDEF_SEC_STRENGTHS;
            int32_t uv_level = frm_hdr->cdef_params.cdef_uv_strength[s] / CDEF_SEC_STRENGTHS, uv_sec = frm_hdr->cdef_params.cdef_uv_strength[s] % CDEF_SEC_STRENGTHS;
            sec += sec == 3; uv_sec += uv_sec == 3;
            int32_t count;
            if ((level == 0 && sec == 0 && uv_level == 0 && uv_sec == 0) ||
                (count = svt_sb_compute_cdef_list(pcs, cm, fbr * MI_SIZE_64X64, fbc * MI_SIZE_64X64, dlist, BLOCK_64X64)) == 0)
                continue; /* the whole filter block stays as it is */
            const int32_t fb = fbr * nhfb + fbc;
            pri_y[fb] = level; sec_y[fb] = sec; pri_uv[fb] = uv_level; sec_uv[fb] = uv_sec;
            for (int32_t i = 0; i < count; i++) skip[(size_t)(fbr * 8 + dlist[i].by) * (nhfb * 8) + fbc * 8 + dlist[i].bx] = 0;
            filtered++;
        }
    SvtHipCdefApplyHost A;
    memset(&A, 0, sizeof(A));
    A.plane[0] = recon_pic->buffer_y + ((recon_pic->org_x + recon_pic->org_y * recon_pic->stride_y) << is_16bit);
    A.plane[1] = recon_pic->buffer_cb + (((recon_pic->org_x + recon_pic->org_y * recon_pic->stride_cb) >> 1) << is_16bit);
    A.plane[2] = recon_pic->buffer_cr + (((recon_pic->org_x + recon_pic->org_y * recon_pic->stride_cr) >> 1) << is_16bit);
    A.stride[0] = recon_pic->stride_y; A.stride[1] = recon_pic->stride_cb; A.stride[2] = recon_pic->stride_cr;
    A.width = (uint32_t)cm->mi_cols << MI_SIZE_LOG2; A.height = (uint32_t)cm->mi_rows << MI_SIZE_LOG2;
    A.num_planes = 3; A.is_16bit = is_16bit;
    A.coeff_shift = (uint8_t)AOMMAX(scs->static_config.encoder_bit_depth - 8, 0);
    A.damping = (uint8_t)frm_hdr->cdef_params.cdef_damping;
    A.skip = skip; A.pri_y = pri_y; A.sec_y = sec_y; A.pri_uv = pri_uv; A.sec_uv = sec_uv;
    svt_hip_seam_bind(pcs->picture_number);
    /* SVT_HIP_CDEF_SEAM_VERIFY=1 (diagnostic): the reference's own function filters the same picture as well and the two results are compared sample by sample
     * (differences go to stderr); the picture continues with the reference's result. */
    SEAM_ENV_ONCE(verify, (getenv("SVT_HIP_CDEF_SEAM_VERIFY") && atoi(getenv("SVT_HIP_CDEF_SEAM_VERIFY"))));
    uint8_t *before[3] = {NULL, NULL, NULL}, *device[3] = {NULL, NULL, NULL};
    const size_t vrows[3] = {A.height, (A.height + 1) >> 1, (A.height + 1) >> 1}, vcols[3] = {A.width, (A.width + 1) >> 1, (A.width + 1) >> 1};
    if (verify && filtered)
        for (int p = 0; p < 3; p++) {
            before[p] = malloc(vrows[p] * (vcols[p] << is_16bit));
            device[p] = malloc(vrows[p] * (vcols[p] << is_16bit));
            for (size_t y = 0; y < vrows[p]; y++) memcpy(before[p] + y * (vcols[p] << is_16bit), (uint8_t *)A.plane[p] + y * ((size_t)A.stride[p] << is_16bit), vcols[p] << is_16bit);
        }
    const double ta_ = seam_ms_now();
    const int apply_rc = filtered ? D.apply_host(&A) : 0; /* (the host form writes the planes after its last device operation: a failed call has changed nothing) */
    __atomic_fetch_add(&D.us_stage, (unsigned long long)((seam_ms_now() - ta_) * 1e3), __ATOMIC_RELAXED);
    if (apply_rc) {
        pthread_mutex_lock(&D.lock); D.n_declined++; pthread_mutex_unlock(&D.lock);
        svt_av1_cdef_frame(scs, pcs);
        for (int p = 0; p < 3; p++) { free(before[p]); free(device[p]); }
        free(str); free(skip);
        return;
    }
    if (verify && filtered) {
        for (int p = 0; p < 3; p++)
            for (size_t y = 0; y < vrows[p]; y++) {
                uint8_t *row = (uint8_t *)A.plane[p] + y * ((size_t)A.stride[p] << is_16bit);
                memcpy(device[p] + y * (vcols[p] << is_16bit), row, vcols[p] << is_16bit);
                memcpy(row, before[p] + y * (vcols[p] << is_16bit), vcols[p] << is_16bit);
            }
        svt_av1_cdef_frame(scs, pcs);
        for (int p = 0; p < 3; p++) {
            unsigned long long bad = 0; long first = -1;
            for (size_t y = 0; y < vrows[p]; y++) {
                const uint8_t *row = (const uint8_t *)A.plane[p] + y * ((size_t)A.stride[p] << is_16bit), *dv = device[p] + y * (vcols[p] << is_16bit);
                for (size_t x = 0; x < (vcols[p] << is_16bit); x++)
                    if (row[x] != dv[x]) { if (first < 0) first = (long)(y * 100000 + (x >> is_16bit)); bad++; }
            }
            if (bad) fprintf(stderr, "SVT_HIP_CDEF_SEAM_VERIFY: apply, picture %llu plane %d: %llu bytes differ from the reference (first at row*100000+col %ld)\n", (unsigned long long)pcs->picture_number, p, bad, first);
            free(before[p]); free(device[p]);
        }
        fprintf(stderr, "SVT_HIP_CDEF_SEAM_VERIFY: apply, picture %llu compared\n", (unsigned long long)pcs->picture_number);
    }
    pthread_mutex_lock(&D.lock);
    D.n_pictures++; D.n_fbs += filtered;
    pthread_mutex_unlock(&D.lock);
    free(str); free(skip);
}
static void seam_av1_cdef_frame(SequenceControlSet *scs, PictureControlSet *pcs) {
    seam_test_delay();
    SEAM_CPU_BEGIN();
    seam_av1_cdef_frame_body(scs, pcs);
    SEAM_CPU_END(SEAM_CPU_CDEF);
}


/* ---- the search: cdef_seg_search_use0 = the reference's function (defined by the #include below), cdef_seg_search_use1 = what its call site reaches ---- */
static void cdef_seg_search_use0(PictureControlSet *pcs, SequenceControlSet *scs, uint32_t segment_index);
static int search_picture(PictureControlSet *pcs, SequenceControlSet *scs) {
    struct PictureParentControlSet *ppcs     = pcs->ppcs;
    Av1Common                      *cm       = ppcs->av1_cm;
    const bool                      is_16bit = scs->is_16bit_pipeline;
    const CdefSearchControls       *ctl      = &ppcs->cdef_search_ctrls;
    const int32_t nvfb = (cm->mi_rows + MI_SIZE_64X64 - 1) / MI_SIZE_64X64, nhfb = (cm->mi_cols + MI_SIZE_64X64 - 1) / MI_SIZE_64X64, nfb = nvfb * nhfb;
    const int     n1 = ctl->first_pass_fs_num, n2 = ctl->default_second_pass_fs_num, ncand = n1 + n2;
    EbPictureBufferDesc *input_pic = is_16bit ? pcs->input_frame16bit : ppcs->enhanced_pic, *recon_pic;
    svt_aom_get_recon_pic(pcs, &recon_pic, is_16bit);
    int32_t pri_y[TOTAL_STRENGTHS], sec_y[TOTAL_STRENGTHS], pri_uv[TOTAL_STRENGTHS], sec_uv[TOTAL_STRENGTHS], uv_slot[TOTAL_STRENGTHS], ncand_uv = 0;
    for (int gi = 0; gi < ncand; gi++) { /* :247-300 */
        const int fs = gi < n1 ? ctl->default_first_pass_fs[gi] : ctl->default_second_pass_fs[gi - n1];
        const int uv = gi < n1 ? ctl->default_first_pass_fs_uv[gi] : ctl->default_second_pass_fs_uv[gi - n1];
        pri_y[gi] = fs / CDEF_SEC_STRENGTHS; sec_y[gi] = fs % CDEF_SEC_STRENGTHS; sec_y[gi] += sec_y[gi] == 3;
        uv_slot[gi] = -1;
        if (uv != -1) { uv_slot[gi] = ncand_uv; pri_uv[ncand_uv] = pri_y[gi]; sec_uv[ncand_uv] = sec_y[gi]; ncand_uv++; }
    }
    uint8_t  *skip = malloc((size_t)nvfb * 8 * nhfb * 8);
    int32_t  *count = calloc((size_t)nfb, sizeof(int32_t));
    CdefList  dlist[MI_SIZE_64X64 * MI_SIZE_64X64];
    memset(skip, 1, (size_t)nvfb * 8 * nhfb * 8);
    uint64_t searched = 0;
    for (int32_t fbr = 0; fbr < nvfb; fbr++)
        for (int32_t fbc = 0; fbc < nhfb; fbc++) { /* :196-204 */
            const int32_t fb = fbr * nhfb + fbc;
            count[fb] = svt_sb_compute_cdef_list(pcs, cm, fbr * MI_SIZE_64X64, fbc * MI_SIZE_64X64, dlist, BLOCK_64X64);
            pcs->skip_cdef_seg[fb] = count[fb] == 0;
            for (int32_t i = 0; i < count[fb]; i++) skip[(size_t)(fbr * 8 + dlist[i].by) * (nhfb * 8) + fbc * 8 + dlist[i].bx] = 0;
            searched += count[fb] != 0;
        }
    uint64_t *mse = calloc((size_t)nfb * (ncand + 2 * (size_t)ncand_uv) + 1, 8), *mse_y = mse, *mse_u = mse + (size_t)nfb * ncand, *mse_v = mse_u + (size_t)nfb * ncand_uv;
    uint8_t  *dir = calloc((size_t)nfb * 64, 1);
    int32_t  *var = calloc((size_t)nfb * 64, 4);
    const uint8_t sf = ctl->subsampling_factor;
    SvtHipCdefSearchHost A;
    memset(&A, 0, sizeof(A));
    for (int pli = 0; pli < 3; pli++) {
        A.recon[pli] = pcs->cdef_input_recon[pli]; A.source[pli] = pcs->cdef_input_source[pli];
        A.recon_stride[pli]  = pli == 0 ? recon_pic->stride_y : (pli == 1 ? recon_pic->stride_cb : recon_pic->stride_cr);
        A.source_stride[pli] = pli == 0 ? input_pic->stride_y : (pli == 1 ? input_pic->stride_cb : input_pic->stride_cr);
    }
    A.width = (uint32_t)cm->mi_cols << MI_SIZE_LOG2; A.height = (uint32_t)cm->mi_rows << MI_SIZE_LOG2;
    A.is_16bit = is_16bit; A.coeff_shift = (uint8_t)AOMMAX(scs->static_config.encoder_bit_depth - 8, 0);
    A.damping = (uint8_t)(3 + (ppcs->frm_hdr.quantization_params.base_q_idx >> 6));
    A.subsampling[0] = sf < 4 ? sf : 4; A.subsampling[1] = sf < 1 ? sf : 1; /* :215-219: 8x8 luma units, 4x4 chroma units */
    A.skip = skip; A.ncand_y = (uint32_t)ncand; A.ncand_uv = (uint32_t)ncand_uv;
    A.pri_y = pri_y; A.sec_y = sec_y; A.pri_uv = pri_uv; A.sec_uv = sec_uv;
    A.mse_y = mse_y; A.mse_u = mse_u; A.mse_v = mse_v; A.dir = dir; A.var = var;
    svt_hip_seam_bind(pcs->picture_number);
    const double ts_ = seam_ms_now();
    const int search_rc = searched ? D.search_host(&A) : 0;
    __atomic_fetch_add(&D.us_stage, (unsigned long long)((seam_ms_now() - ts_) * 1e3), __ATOMIC_RELAXED);
    if (search_rc) { free(var); free(dir); free(mse); free(count); free(skip); return search_rc; }
    for (int32_t fb = 0; fb < nfb; fb++) {
        if (!count[fb]) continue;
        for (int gi = 0; gi < ncand; gi++) {
            pcs->mse_seg[0][fb][gi] = mse_y[(size_t)fb * ncand + gi] * A.subsampling[0];
            pcs->mse_seg[1][fb][gi] = uv_slot[gi] < 0 ? (uint64_t)1040400 * 64 /* default_mse_uv * 64 */
                                                      : (mse_u[(size_t)fb * ncand_uv + uv_slot[gi]] + mse_v[(size_t)fb * ncand_uv + uv_slot[gi]]) * A.subsampling[1];
        }
        for (int k = 0; k < 64; k++) {
            if (skip[(size_t)((fb / nhfb) * 8 + (k >> 3)) * (nhfb * 8) + (fb % nhfb) * 8 + (k & 7)]) continue; /* the reference writes the listed units only */
            pcs->cdef_dir_data[fb].dir[k >> 3][k & 7] = dir[(size_t)fb * 64 + k];
            pcs->cdef_dir_data[fb].var[k >> 3][k & 7] = var[(size_t)fb * 64 + k];
        }
    }
    D.n_searched++; D.n_search_fbs += searched;
    free(var); free(dir); free(mse); free(count); free(skip);
    return 0;
}
static void cdef_seg_search_use1_body(PictureControlSet *pcs, SequenceControlSet *scs, uint32_t segment_index) {
    if (!cdef_seam_on() || scs->super_block_size == 128) { cdef_seg_search_use0(pcs, scs, segment_index); return; }
    pthread_mutex_lock(&D.lock);
    int slot = -1, free_slot = -1;
    for (int i = 0; i < 64; i++) {
        if (D.done_pcs[i] == pcs && D.done_num[i] == pcs->picture_number) slot = i;
        if (!D.done_pcs[i] && free_slot < 0) free_slot = i;
    }
    if (slot < 0) {
        if (search_picture(pcs, scs)) { /* the device path is off: the reference's own search of every segment of this picture, here and now (the others find it done) */
            for (uint32_t sg = 0; sg < pcs->cdef_segments_total_count; sg++) cdef_seg_search_use0(pcs, scs, sg);
            D.n_declined++;
        }
        SEAM_ENV_ONCE(verify, (getenv("SVT_HIP_CDEF_SEAM_VERIFY") && atoi(getenv("SVT_HIP_CDEF_SEAM_VERIFY"))));
        if (verify) { /* (diagnostic) the reference's own search of every segment, compared with what the device stage stored */
            const int32_t nvfb = (pcs->ppcs->av1_cm->mi_rows + MI_SIZE_64X64 - 1) / MI_SIZE_64X64, nhfb = (pcs->ppcs->av1_cm->mi_cols + MI_SIZE_64X64 - 1) / MI_SIZE_64X64, nfb = nvfb * nhfb;
            uint64_t (*m0)[TOTAL_STRENGTHS] = malloc(sizeof(*m0) * nfb), (*m1)[TOTAL_STRENGTHS] = malloc(sizeof(*m1) * nfb);
            uint8_t     *sk = malloc((size_t)nfb);
            CdefDirData *dd = malloc(sizeof(*dd) * nfb);
            memcpy(m0, pcs->mse_seg[0], sizeof(*m0) * nfb); memcpy(m1, pcs->mse_seg[1], sizeof(*m1) * nfb); memcpy(sk, pcs->skip_cdef_seg, (size_t)nfb); memcpy(dd, pcs->cdef_dir_data, sizeof(*dd) * nfb);
            for (uint32_t sg = 0; sg < pcs->cdef_segments_total_count; sg++) cdef_seg_search_use0(pcs, scs, sg);
            const CdefSearchControls *ctl = &pcs->ppcs->cdef_search_ctrls;
            const int nc = ctl->first_pass_fs_num + ctl->default_second_pass_fs_num;
            unsigned long long bad_mse = 0, bad_dir = 0, bad_skip = 0;
            for (int32_t fb = 0; fb < nfb; fb++) {
                bad_skip += sk[fb] != pcs->skip_cdef_seg[fb];
                if (pcs->skip_cdef_seg[fb]) continue;
                for (int gi = 0; gi < nc; gi++) bad_mse += (m0[fb][gi] != pcs->mse_seg[0][fb][gi]) + (m1[fb][gi] != pcs->mse_seg[1][fb][gi]);
                bad_dir += memcmp(&dd[fb], &pcs->cdef_dir_data[fb], sizeof(*dd)) != 0;
            }
            if (bad_mse || bad_dir || bad_skip)
                fprintf(stderr, "SVT_HIP_CDEF_SEAM_VERIFY: search, picture %llu: %llu mse entries, %llu direction records, %llu skip flags differ from the reference\n",
                        (unsigned long long)pcs->picture_number, bad_mse, bad_dir, bad_skip);
            fprintf(stderr, "SVT_HIP_CDEF_SEAM_VERIFY: search, picture %llu compared (%d filter blocks, %d candidates)\n", (unsigned long long)pcs->picture_number, nfb, nc);
            free(m0); free(m1); free(sk); free(dd);
        }
        if (free_slot < 0) { fprintf(stderr, "SVT_HIP_CDEF_SEAM: more than 64 pictures in the CDEF stage\n"); abort(); }
        slot = free_slot;
        D.done_pcs[slot] = pcs; D.done_num[slot] = pcs->picture_number; D.seen[slot] = 0;
    }
    if (++D.seen[slot] == pcs->cdef_segments_total_count) D.done_pcs[slot] = NULL;
    pthread_mutex_unlock(&D.lock);
}
static void cdef_seg_search_use1(PictureControlSet *pcs, SequenceControlSet *scs, uint32_t segment_index) {
    seam_test_delay();
    SEAM_CPU_BEGIN();
    cdef_seg_search_use1_body(pcs, scs, segment_index);
    SEAM_CPU_END(SEAM_CPU_CDEF);
}


#define SEAM_CAT_(a, b) a##b
#define SEAM_CAT(a, b) SEAM_CAT_(a, b)
#define cdef_seg_search(pcs, scs, idx) SEAM_CAT(cdef_seg_search_use, __COUNTER__)(pcs, scs, idx)
#define svt_av1_cdef_frame(scs, pcs) seam_av1_cdef_frame(scs, pcs)
#include "cdef_process.c" /* resolves through -I$(REF)/Source/Lib/Codec */
