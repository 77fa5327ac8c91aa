/* cdef_process_seam.c -- TEST / BASELINE INFRASTRUCTURE: the reference's CDEF process with the frame-apply seam of INTEGRATION.md §3.
 *
 * This translation unit IS Source/Lib/Codec/cdef_process.c of the reference (included below where it lies; nothing is copied).  The one change: the call
 *
 *     svt_av1_cdef_frame(scs, pcs);                                                                                      (cdef_process.c:458)
 *
 * is given a macro name for the duration of the #include and lands in seam_av1_cdef_frame() below.  With SVT_HIP_CDEF_SEAM unset (or the HIP library not loaded)
 * that function IS the reference call.  With SVT_HIP_CDEF_SEAM=1 the whole picture is filtered by ONE svt_hip_cdef_apply_host() call: the per-filter-block
 * strengths are read exactly as svt_av1_cdef_frame reads them (enc_cdef.c:385-395: frm_hdr->cdef_params.cdef_y_strength / cdef_uv_strength indexed by the
 * block's mbmi.cdef_strength), the 8x8 skip map comes from the reference's own svt_sb_compute_cdef_list, filter blocks the reference skips (all strengths zero,
 * or no unit to filter, :396-401) are marked skipped as a whole.  The strength SEARCH (cdef_seg_search) is a static function of the same file and stays C; with
 * SB 128 the reference's per-block dirinit rule applies and the seam declines (calls the reference).  SVT_HIP_CDEF_SEAM_STATS=<file> receives the counters.
 */
#define _GNU_SOURCE /* RTLD_DEFAULT */
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

static struct {
    pthread_mutex_t lock;
    int             mode;
    void (*apply_host)(const SvtHipCdefApplyHost *);
    uint64_t n_pictures, n_fbs, n_declined;
} D = {PTHREAD_MUTEX_INITIALIZER};

static void cdef_seam_stats(void) {
    const char *f = getenv("SVT_HIP_CDEF_SEAM_STATS");
    FILE       *o = f ? fopen(f, "w") : NULL;
    if (!o) return;
    fprintf(o, "pictures_filtered %llu\nfilter_blocks %llu\npictures_declined %llu\n", (unsigned long long)D.n_pictures, (unsigned long long)D.n_fbs,
            (unsigned long long)D.n_declined);
    fclose(o);
}
static void cdef_seam_init(void) {
    const char *e = getenv("SVT_HIP_CDEF_SEAM");
    if (!e || !atoi(e) || !getenv("SVT_HIP")) return;
    *(void **)&D.apply_host = dlsym(RTLD_DEFAULT, "svt_hip_cdef_apply_host");
    if (!D.apply_host) { fprintf(stderr, "SVT_HIP_CDEF_SEAM: libsvtav1_hip is not loaded\n"); abort(); }
    atexit(cdef_seam_stats);
    fprintf(stderr, "SVT_HIP_CDEF_SEAM: CDEF is applied to a picture by one device call\n");
    D.mode = 1;
}
static int cdef_seam_on(void) {
    static pthread_once_t once = PTHREAD_ONCE_INIT;
    pthread_once(&once, cdef_seam_init);
    return D.mode;
}

static void seam_av1_cdef_frame(SequenceControlSet *scs, PictureControlSet *pcs) {
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
    if (filtered) D.apply_host(&A);
    pthread_mutex_lock(&D.lock);
    D.n_pictures++; D.n_fbs += filtered;
    pthread_mutex_unlock(&D.lock);
    free(str); free(skip);
}

#define svt_av1_cdef_frame(scs, pcs) seam_av1_cdef_frame(scs, pcs)
#include "cdef_process.c" /* resolves through -I$(REF)/Source/Lib/Codec */
