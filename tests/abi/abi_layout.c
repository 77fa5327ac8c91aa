/* abi_layout.c -- TEST INFRASTRUCTURE: the self-contained PODs of include/svtav1_hip.h that stand for a struct of the reference have that struct's layout.
 *
 * Companion of abi_typecheck.c (which compiles the header with SVT_HIP_REFERENCE_TYPES, i.e. with those names typedef'd to the reference's structs).  Here the
 * header is compiled WITHOUT it, next to the reference's headers, and every mirrored POD is compared field by field: same size, same alignment, same offset and
 * size of every field.  Compiled (never run) by tests/test_abi_typecheck.py; needs /root/reference. */
#include <stddef.h>
#include "definitions.h"
#include "aom_dsp_rtcd.h"
#include "common_dsp_rtcd.h"
#include "mcomp.h"
#include "svtav1_hip.h"

#define SAME_STRUCT(ours, theirs) \
    _Static_assert(sizeof(ours) == sizeof(theirs) && _Alignof(ours) == _Alignof(theirs), "size/alignment of " #ours " vs " #theirs)
#define SAME_FIELD(ours, theirs, field)                                                                                                   \
    _Static_assert(offsetof(ours, field) == offsetof(theirs, field) && sizeof(((ours *)0)->field) == sizeof(((theirs *)0)->field), \
                   "field " #field " of " #ours " vs " #theirs)

SAME_STRUCT(SvtHipMv, MV);
SAME_FIELD(SvtHipMv, MV, row);
SAME_FIELD(SvtHipMv, MV, col);

SAME_STRUCT(SvtHipMvCostParams, MV_COST_PARAMS); /* mcomp.h:37-48 */
SAME_FIELD(SvtHipMvCostParams, MV_COST_PARAMS, ref_mv);
SAME_FIELD(SvtHipMvCostParams, MV_COST_PARAMS, full_ref_mv);
SAME_FIELD(SvtHipMvCostParams, MV_COST_PARAMS, mv_cost_type);
SAME_FIELD(SvtHipMvCostParams, MV_COST_PARAMS, mvjcost);
SAME_FIELD(SvtHipMvCostParams, MV_COST_PARAMS, mvcost);
SAME_FIELD(SvtHipMvCostParams, MV_COST_PARAMS, error_per_bit);
SAME_FIELD(SvtHipMvCostParams, MV_COST_PARAMS, early_exit_th);
SAME_FIELD(SvtHipMvCostParams, MV_COST_PARAMS, sad_per_bit);

SAME_STRUCT(SvtHipTxfmParam, TxfmParam); /* definitions.h:1043-1055 */
SAME_FIELD(SvtHipTxfmParam, TxfmParam, tx_type);
SAME_FIELD(SvtHipTxfmParam, TxfmParam, tx_size);
SAME_FIELD(SvtHipTxfmParam, TxfmParam, lossless);
SAME_FIELD(SvtHipTxfmParam, TxfmParam, bd);
SAME_FIELD(SvtHipTxfmParam, TxfmParam, is_hbd);
SAME_FIELD(SvtHipTxfmParam, TxfmParam, tx_set_type);
SAME_FIELD(SvtHipTxfmParam, TxfmParam, eob);

SAME_STRUCT(SvtHipBuf2D, Buf2D); /* definitions.h:243-249 */
SAME_FIELD(SvtHipBuf2D, Buf2D, buf);
SAME_FIELD(SvtHipBuf2D, Buf2D, buf0);
SAME_FIELD(SvtHipBuf2D, Buf2D, width);
SAME_FIELD(SvtHipBuf2D, Buf2D, height);
SAME_FIELD(SvtHipBuf2D, Buf2D, stride);

SAME_STRUCT(SvtHipCdefList, CdefList); /* definitions.h:256-259 */
SAME_FIELD(SvtHipCdefList, CdefList, by);
SAME_FIELD(SvtHipCdefList, CdefList, bx);

SAME_STRUCT(SvtHipSgrParams, SgrParamsType); /* definitions.h:1750-1753 */
SAME_FIELD(SvtHipSgrParams, SgrParamsType, r);
SAME_FIELD(SvtHipSgrParams, SgrParamsType, s);

SAME_STRUCT(SvtHipConvolveParams, ConvolveParams); /* definitions.h:572-585 */
SAME_FIELD(SvtHipConvolveParams, ConvolveParams, ref);
SAME_FIELD(SvtHipConvolveParams, ConvolveParams, do_average);
SAME_FIELD(SvtHipConvolveParams, ConvolveParams, dst);
SAME_FIELD(SvtHipConvolveParams, ConvolveParams, dst_stride);
SAME_FIELD(SvtHipConvolveParams, ConvolveParams, round_0);
SAME_FIELD(SvtHipConvolveParams, ConvolveParams, round_1);
SAME_FIELD(SvtHipConvolveParams, ConvolveParams, plane);
SAME_FIELD(SvtHipConvolveParams, ConvolveParams, is_compound);
SAME_FIELD(SvtHipConvolveParams, ConvolveParams, use_jnt_comp_avg);
SAME_FIELD(SvtHipConvolveParams, ConvolveParams, fwd_offset);
SAME_FIELD(SvtHipConvolveParams, ConvolveParams, bck_offset);
SAME_FIELD(SvtHipConvolveParams, ConvolveParams, use_dist_wtd_comp_avg);

_Static_assert(sizeof(SvtHipBlockSize) == sizeof(BlockSize), "BlockSize is a one-byte packed enum");
_Static_assert(sizeof(EbBitDepth) == sizeof(unsigned int), "EbBitDepth is passed as an unsigned int");
_Static_assert(sizeof(TxType) == 1 && sizeof(TxSize) == 1, "TxType / TxSize are one-byte packed enums");
