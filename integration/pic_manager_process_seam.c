/* pic_manager_process_seam.c -- TEST HARNESS inside the reference-side binding (built into the reference encoder by oracle/Makefile for the identity runs; a deployment does
 * not need it): makes the reference's 10-bit encodes reproducible, so that the bitstream-identity tests can demand equality on the FIRST attempt.
 *
 * This translation unit IS Source/Lib/Codec/pic_manager_process.c of the reference (included below where it lies; nothing is copied) with one call renamed for the duration
 * of the #include: svt_get_empty_object(fifo, &wrapper).  With SVT_HIP_TEST_SCRUB_PCS unset the replacement IS that call.  With SVT_HIP_TEST_SCRUB_PCS=1, when the object
 * taken is a child PictureControlSet (the fifo is context_ptr->picture_control_set_fifo_ptr, :602), its 16-bit source picture `input_frame16bit` is zero-filled and then
 * packed in full from the parent's input picture before the picture manager hands it on.
 *
 * Why: mode decision packs the 16-bit source one superblock at a time, just before that superblock is searched (pad_hbd_pictures -> svt_aom_store16bit_input_src,
 * product_coding_loop.c:10057-10137), and the psy-rd distortion of this fork reads 8x8 / 4x4 units of that picture past what has been packed (svt_psy_distortion,
 * svt_sa8d_8x8, svt_satd_4x4, psy_rd.c:94-165 -- MemorySanitizer on the plain C encoder, profiles/r05_reference_msan_10bit.txt).  What it finds there is either what the
 * previous picture that used the same pool object left behind -- WHICH picture that was depends on the order in which pictures return their control sets -- or, with more
 * than one EncDec thread, whatever a neighbouring superblock's thread has or has not packed yet.  Both are thread timing: at `--lp 1` the plain reference flips now and
 * then (profiles/r05_race_probe_10bit.txt).  (Multi-threaded it gives a different 10-bit bitstream in EVERY run -- five md5s in five 720p preset-8 encodes; one md5 with
 * --psy-rd 0, one md5 at 8 bit -- and stays that way under this harness: the same distortion code also reads per-thread scratch buffers, and which thread gets which
 * superblock is timing.  The identity cases therefore run 10 bit single-threaded, or multi-threaded with --psy-rd 0.)  The harness takes the pool-order timing out by doing, when the control set is taken from its pool, exactly what mode decision will do later
 * superblock by superblock: every superblock of the picture is packed into `input_frame16bit` with the reference's own functions (svt_aom_compressed_pack_sb,
 * svt_aom_pad_input_picture_16bit, the store of coding_loop.c:679-718).  The later per-superblock packs then rewrite the same values, and a read past the packed area
 * returns the same thing in every run, whatever the order.  The encoder's defined behaviour -- everything it reads after writing -- is untouched.  Both encodes of a
 * comparison (the reference alone and the reference with the device stages) run with it.
 */
#include <malloc.h>
#include <stdlib.h>
#include <string.h>

#include "enc_handle.h"
#include "pcs.h"
#include "pic_operators.h"
#include "sequence_control_set.h"
#include "sys_resource_manager.h"

static int svt_hip_test_scrub_on(void) {
    static int on = -1;
    if (on < 0) {
        const char *e = getenv("SVT_HIP_TEST_SCRUB_PCS");
        on            = e && *e && *e != '0';
    }
    return on;
}
static void svt_hip_test_scrub_plane(uint8_t *buf) { /* the whole allocation (EB_MALLOC_ALIGNED_ARRAY = posix_memalign, pic_buffer_desc.c:270-292): padding included */
    if (buf)
        memset(buf, 0, malloc_usable_size(buf));
}
static EbErrorType svt_hip_test_get_empty_object(EbFifo *fifo, EbObjectWrapper **wrapper, EbFifo *child_pcs_fifo) {
    const EbErrorType err = svt_get_empty_object(fifo, wrapper);
    if (fifo == child_pcs_fifo && svt_hip_test_scrub_on()) { /* first the whole allocation to zero (the buffer's outer padding is never written by anybody) ... */
        EbPictureBufferDesc *p = ((PictureControlSet *)(*wrapper)->object_ptr)->input_frame16bit;
        if (p) {
            svt_hip_test_scrub_plane(p->buffer_y);
            svt_hip_test_scrub_plane(p->buffer_cb);
            svt_hip_test_scrub_plane(p->buffer_cr);
        }
    }
    return err;
}
/* ... then, once the child control set is linked to its parent (the hook is the first tile call after the link, :742), the picture itself: pad_hbd_pictures
 * (product_coding_loop.c:10057-10137) for every superblock */
static void svt_hip_test_prepack(PictureControlSet *pcs) {
    if (!svt_hip_test_scrub_on() || !pcs || !pcs->input_frame16bit || !pcs->ppcs || !pcs->scs || pcs->scs->static_config.encoder_bit_depth <= EB_EIGHT_BIT)
        return;
    EbPictureBufferDesc *in = pcs->ppcs->enhanced_pic, *out = pcs->input_frame16bit;
    if (!in || !in->buffer_bit_inc_y || in->color_format != EB_YUV420)
        return;
    const uint32_t sb = pcs->scs->sb_size, aw = pcs->ppcs->aligned_width, ah = pcs->ppcs->aligned_height;
    uint16_t *ty = (uint16_t *)malloc((size_t)sb * sb * 2), *tu = (uint16_t *)malloc((size_t)sb * sb / 2), *tv = (uint16_t *)malloc((size_t)sb * sb / 2);
    const uint32_t cs_y = in->stride_y / 4, cs_uv = in->stride_cb / 4;
    for (uint32_t y = 0; y < ah; y += sb)
        for (uint32_t x = 0; x < aw; x += sb) {
            const uint32_t w = sb < aw - x ? sb : aw - x, h = sb < ah - y ? sb : ah - y;
            svt_aom_compressed_pack_sb(in->buffer_y + (y + in->org_y) * in->stride_y + x + in->org_x, in->stride_y,
                                       in->buffer_bit_inc_y + cs_y * in->org_y + in->org_x / 4 + x / 4 + y * cs_y, cs_y, ty, sb, w, h);
            svt_aom_compressed_pack_sb(in->buffer_cb + ((y + in->org_y) >> 1) * in->stride_cb + ((x + in->org_x) >> 1), in->stride_cb,
                                       in->buffer_bit_inc_cb + cs_uv * (in->org_y / 2) + in->org_x / 2 / 4 + x / 4 / 2 + y / 2 * cs_uv, cs_uv, tu, sb / 2, w / 2, h / 2);
            svt_aom_compressed_pack_sb(in->buffer_cr + ((y + in->org_y) >> 1) * in->stride_cr + ((x + in->org_x) >> 1), in->stride_cr,
                                       in->buffer_bit_inc_cr + cs_uv * (in->org_y / 2) + in->org_x / 2 / 4 + x / 4 / 2 + y / 2 * cs_uv, cs_uv, tv, sb / 2, w / 2, h / 2);
            svt_aom_pad_input_picture_16bit(ty, sb, w, h, sb - w, sb - h);
            svt_aom_pad_input_picture_16bit(tu, sb / 2, w >> 1, h >> 1, (sb - w) >> 1, (sb - h) >> 1);
            svt_aom_pad_input_picture_16bit(tv, sb / 2, w >> 1, h >> 1, (sb - w) >> 1, (sb - h) >> 1);
            /* svt_aom_store16bit_input_src(buffer, pcs, x, y, sb, sb): the whole sb x sb block, coding_loop.c:679-718 */
            uint16_t *dy = (uint16_t *)out->buffer_y + (x + out->org_x) + (y + out->org_y) * out->stride_y;
            uint16_t *du = (uint16_t *)out->buffer_cb + (x / 2 + out->org_x / 2) + (y / 2 + out->org_y / 2) * out->stride_cb;
            uint16_t *dv = (uint16_t *)out->buffer_cr + (x / 2 + out->org_x / 2) + (y / 2 + out->org_y / 2) * out->stride_cb;
            for (uint32_t r = 0; r < sb; r++) memcpy(dy + r * out->stride_y, ty + r * sb, sb * 2);
            for (uint32_t r = 0; r < sb / 2; r++) {
                memcpy(du + r * out->stride_cb, tu + r * (sb / 2), sb);
                memcpy(dv + r * out->stride_cr, tv + r * (sb / 2), sb);
            }
        }
    free(ty); free(tu); free(tv);
}
/* every call in the file sits in svt_aom_picture_manager_kernel, where `context_ptr` is the PictureManagerContext (:312-356) */
#define svt_get_empty_object(fifo, wrapper) svt_hip_test_get_empty_object(fifo, wrapper, context_ptr->picture_control_set_fifo_ptr)
/* both uses (:260 after a resolution change, :742 for every new picture) have `child_pcs` in scope, linked to its parent; a macro is not re-expanded inside itself */
#define svt_av1_tile_set_row(a, b, c, d) (svt_hip_test_prepack(child_pcs), svt_av1_tile_set_row(a, b, c, d))
#include "pic_manager_process.c" /* resolves through -I$(REF)/Source/Lib/Codec */
