/* temporal_filtering_seam.c -- TEST / BASELINE INFRASTRUCTURE: the reference's temporal filter with its ME call routed through the ME seam.
 *
 * This translation unit IS Source/Lib/Codec/temporal_filtering.c of the reference (included below where it lies; nothing is copied).  The one change: the call
 *
 *     svt_aom_motion_estimation_b64(centre_pcs, blk_row * blk_cols + blk_col, blk_col * BW, blk_row * BH, ctx, input_picture_ptr_central);        (:3180)
 *
 * is given a macro name for the duration of the #include and lands in svt_hip_seam_tf_motion_estimation_b64() (ref_wrap/me_process_seam.c, which owns the
 * device session and the picture ring).  With SVT_HIP_TF_ME_SEAM unset that function IS the reference call.
 */
#include "motion_estimation.h" /* declares svt_aom_motion_estimation_b64 before the macro below exists */
#include "me_context.h"
#include "pcs.h"

EbErrorType svt_hip_seam_tf_motion_estimation_b64(PictureParentControlSet *pcs, uint32_t b64_index, uint32_t b64_origin_x, uint32_t b64_origin_y, MeContext *me_ctx,
                                                  EbPictureBufferDesc *input_ptr);

#define svt_aom_motion_estimation_b64(pcs, i, x, y, ctx, pic) svt_hip_seam_tf_motion_estimation_b64(pcs, i, x, y, ctx, pic)
#include "temporal_filtering.c" /* resolves through -I$(REF)/Source/Lib/Codec */
