/*
 * ref_quant_tables.c -- TEST INFRASTRUCTURE, built only into oracle/_ref/libsvtref_me.so (make -C oracle ref).
 *
 * SURVEY 8(d) config 3 asks for the reference's OWN quantizer tables and scan orders:
 *   - svt_av1_build_quantizer (Codec/md_config_process.c:111-189) fills Quants / Dequants for every q index; it reads two
 *     fields of its picture argument (frm_hdr.quantization_params.base_q_idx, scs->static_config.sharpness);
 *   - av1_scan_orders[TX_SIZES_ALL][TX_TYPES] (Codec/coefficients.h:2197) is `static const`, i.e. private to whichever
 *     translation unit includes that header.
 * This file includes the reference headers where they lie (nothing is copied) and exposes both through plain-C entry points;
 * tools/gen_golden.py freezes their output into tests/golden/quant_tables.npz for the GPU box.
 */
#include <stdlib.h>
#include <string.h>
#include "pcs.h"
#include "sequence_control_set.h"
#include "md_config_process.h"
#include "coefficients.h"
#include "inv_transforms.h"

/* out[7][2] int16: zbin, round, quant, quant_shift, dequant, quant_fp, round_fp -- (DC, AC) of the luma tables at q index `q`. */
void ref_build_quantizer_y(int bit_depth, int base_q_idx, int sharpness, int q, int16_t *out) {
    PictureParentControlSet *pcs = calloc(1, sizeof(*pcs));
    SequenceControlSet      *scs = calloc(1, sizeof(*scs));
    Quants                  *qt  = calloc(1, sizeof(*qt));
    Dequants                *dq  = calloc(1, sizeof(*dq));
    pcs->scs                                  = scs;
    pcs->frm_hdr.quantization_params.base_q_idx = (uint8_t)base_q_idx;
    scs->static_config.sharpness              = (int8_t)sharpness;
    svt_av1_build_quantizer(pcs, (EbBitDepth)bit_depth, 0, 0, 0, 0, 0, qt, dq);
    for (int i = 0; i < 2; i++) {
        out[0 * 2 + i] = qt->y_zbin[q][i];
        out[1 * 2 + i] = qt->y_round[q][i];
        out[2 * 2 + i] = qt->y_quant[q][i];
        out[3 * 2 + i] = qt->y_quant_shift[q][i];
        out[4 * 2 + i] = dq->y_dequant_qtx[q][i];
        out[5 * 2 + i] = qt->y_quant_fp[q][i];
        out[6 * 2 + i] = qt->y_round_fp[q][i];
    }
    free(pcs); free(scs); free(qt); free(dq);
}

/* scan / iscan of av1_scan_orders[tx_size][tx_type]; returns the number of entries (av1_get_max_eob: 32x32-packed for 64-point sizes). */
int ref_scan_order(int tx_size, int tx_type, int16_t *scan, int16_t *iscan) {
    const ScanOrder *so = &av1_scan_orders[tx_size][tx_type];
    const int        n  = av1_get_max_eob((TxSize)tx_size);
    memcpy(scan, so->scan, n * sizeof(int16_t));
    memcpy(iscan, so->iscan, n * sizeof(int16_t));
    return n;
}
