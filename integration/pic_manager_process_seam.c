/* pic_manager_process_seam.c -- TEST HARNESS inside the reference-side binding (built into the reference encoder by oracle/Makefile for the identity runs; a deployment does
 * not need it): makes the reference's 10-bit encodes reproducible, so that the bitstream-identity tests can demand equality on the FIRST attempt.
 *
 * This translation unit IS Source/Lib/Codec/pic_manager_process.c of the reference (included below where it lies; nothing is copied) with one call renamed for the duration
 * of the #include: svt_get_empty_object(fifo, &wrapper).  With SVT_HIP_TEST_SCRUB_PCS unset the replacement IS that call.  With SVT_HIP_TEST_SCRUB_PCS=1, when the object
 * taken is a child PictureControlSet (the fifo is context_ptr->picture_control_set_fifo_ptr, :602), its 16-bit source picture `input_frame16bit` is zero-filled before the
 * picture manager hands it on.
 *
 * Why: mode decision packs the 16-bit source one superblock at a time, just before that superblock is searched (pad_hbd_pictures -> svt_aom_store16bit_input_src,
 * product_coding_loop.c:10057-10137), and the psy-rd distortion of this fork reads 8x8 / 4x4 units of that picture past what has been packed (svt_psy_distortion,
 * svt_sa8d_8x8, svt_satd_4x4, psy_rd.c:94-165 -- MemorySanitizer on the plain C encoder, profiles/r05_reference_msan_10bit.txt).  What it finds there is whatever the
 * previous picture that used the same pool object left behind, and WHICH picture that was depends on the order in which pictures return their control sets -- i.e. on
 * thread timing: the reference does not always reproduce its own 10-bit bitstream (one flip in eight `--lp 1` runs with every device result replaced by the reference's
 * own, profiles/r05_race_probe_10bit.txt).  Zero-filling at acquisition makes those reads see the same thing in every run, whatever the order; the encoder's defined
 * behaviour -- everything it reads after writing -- is untouched.  Both encodes of a comparison (the reference alone and the reference with the device stages) run with it.
 */
#include <malloc.h>
#include <stdlib.h>
#include <string.h>

#include "enc_handle.h"
#include "pcs.h"
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
    if (fifo == child_pcs_fifo && svt_hip_test_scrub_on()) {
        EbPictureBufferDesc *p = ((PictureControlSet *)(*wrapper)->object_ptr)->input_frame16bit;
        if (p) {
            svt_hip_test_scrub_plane(p->buffer_y);
            svt_hip_test_scrub_plane(p->buffer_cb);
            svt_hip_test_scrub_plane(p->buffer_cr);
        }
    }
    return err;
}
/* every call in the file sits in svt_aom_picture_manager_kernel, where `context_ptr` is the PictureManagerContext (:312-356) */
#define svt_get_empty_object(fifo, wrapper) svt_hip_test_get_empty_object(fifo, wrapper, context_ptr->picture_control_set_fifo_ptr)
#include "pic_manager_process.c" /* resolves through -I$(REF)/Source/Lib/Codec */
