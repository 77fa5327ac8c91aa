/* coding_loop_seam.c -- REFERENCE-SIDE BINDING (what a maintainer of the reference adds; built into the reference encoder by oracle/Makefile for the identity / fps runs): the reference's coding loop with the SB-based deblocking call sent to the deblocking seam.
 *
 * This translation unit IS Source/Lib/Codec/coding_loop.c of the reference (included below where it lies; nothing is copied).  One call is renamed for the duration
 * of the #include: svt_aom_loop_filter_sb(recon_buffer, pcs, ...) at :2297 -- the per-SB deblocking that presets >= 7 run inside the EncDec kernel right after an SB
 * is reconstructed (enable_dlf = dlf_ctrls.enabled && dlf_ctrls.sb_based_dlf, :2278) -- lands in svt_hip_seam_loop_filter_sb() (integration/dlf_process_seam.c).
 * With SVT_HIP_DLF_SEAM unset that function IS the reference call; with SVT_HIP_DLF_SEAM=1 the reference's function runs with recording leaf filters and the picture
 * is deblocked by one device call per plane when it reaches the deblocking process.
 */
#include "deblocking_filter.h" /* declares svt_aom_loop_filter_sb before the macro below exists */
void svt_hip_seam_loop_filter_sb(EbPictureBufferDesc *frame_buffer, PictureControlSet *pcs, int32_t mi_row, int32_t mi_col, int32_t plane_start, int32_t plane_end,
                                 uint8_t last_col);
#define svt_aom_loop_filter_sb(a, b, c, d, e, f, g) svt_hip_seam_loop_filter_sb(a, b, c, d, e, f, g)
#include "coding_loop.c" /* resolves through -I$(REF)/Source/Lib/Codec */
