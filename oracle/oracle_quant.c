/*
 * oracle_quant.c -- TEST INFRASTRUCTURE: plain-C restatement of the reference quantizers and of svt_handle_transform*.
 * Pinned against oracle/_ref in tests/test_oracle_pin_quant.py.  Paths relative to /root/reference/Source/Lib/Codec.
 */
#include <stdint.h>
#include <string.h>

#define QM_BITS 5 /* AOM_QM_BITS */
static inline int32_t rpot(int32_t v, int n) { return (v + ((1 << n) >> 1)) >> n; } /* ROUND_POWER_OF_TWO */
static inline int64_t clamp64i(int64_t v, int64_t lo, int64_t hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* mode 0: svt_aom_quantize_b_c_ii (full_loop.c:29-79)      -- the reverse pre-scan only skips coefficients that the
 *                                                             per-coefficient zbin test rejects anyway
 * mode 1: svt_aom_highbd_quantize_b_c (full_loop.c:149-198)
 * mode 2: quantize_fp_helper_c (full_loop.c:282-342)        (svt_av1_quantize_fp / _32x32 / _64x64 / _qm)
 * mode 3: highbd_quantize_fp_helper_c (full_loop.c:387-453) (svt_av1_highbd_quantize_fp / _qm)
 * qm/iqm may be NULL.  eob = 1 + last scan position with a non-zero level. */
void oracle_quantize(int mode, const int32_t *coeff, int n, const int16_t *zbin, const int16_t *round, const int16_t *quant,
                     const int16_t *quant_shift, int32_t *qcoeff, int32_t *dqcoeff, const int16_t *dequant, uint16_t *eob_ptr,
                     const int16_t *scan, const uint8_t *qm, const uint8_t *iqm, int log_scale) {
    int eob = -1;
    memset(qcoeff, 0, sizeof(int32_t) * n);
    memset(dqcoeff, 0, sizeof(int32_t) * n);
    for (int i = 0; i < n; i++) {
        const int     rc = scan[i], k = rc != 0;
        const int32_t c = coeff[rc], sign = c < 0 ? -1 : 0;
        const int32_t a = (c ^ sign) - sign;
        const int32_t wt = qm ? qm[rc] : (1 << QM_BITS), iwt = iqm ? iqm[rc] : (1 << QM_BITS);
        const int32_t deq_w = (dequant[k] * iwt + (1 << (QM_BITS - 1))) >> QM_BITS;
        int32_t       q = 0, dq = 0;
        if (mode == 0) {
            const int32_t zb = rpot(zbin[k], log_scale);
            if ((int32_t)((uint32_t)a * (uint32_t)wt) >= (zb << QM_BITS)) {
                int64_t tmp = clamp64i((int64_t)a + rpot(round[k], log_scale), INT16_MIN, INT16_MAX);
                tmp *= wt;
                q  = (int32_t)(((((tmp * quant[k]) >> 16) + tmp) * quant_shift[k]) >> (16 - log_scale + QM_BITS));
                dq = (int32_t)((uint32_t)q * (uint32_t)deq_w) >> log_scale;
            }
        } else if (mode == 1) {
            const int32_t zb = rpot(zbin[k], log_scale);
            const int32_t cw = (int32_t)((uint32_t)c * (uint32_t)wt);
            if (cw >= zb * (1 << QM_BITS) || cw <= -zb * (1 << QM_BITS)) {
                const int64_t t1 = (int64_t)a + rpot(round[k], log_scale), tw = t1 * wt;
                const int64_t t2 = ((tw * quant[k]) >> 16) + tw;
                q  = (int32_t)((t2 * quant_shift[k]) >> (16 - log_scale + QM_BITS));
                dq = (int32_t)((uint32_t)q * (uint32_t)deq_w) >> log_scale;
            }
        } else if (mode == 2) {
            const int32_t rnd = rpot(round[k], log_scale);
            if (!qm && !iqm) {
                if (((int64_t)a << (1 + log_scale)) >= (int32_t)dequant[k]) {
                    const int64_t t = clamp64i((int64_t)a + rnd, INT16_MIN, INT16_MAX);
                    q = (int32_t)((t * quant[k]) >> (16 - log_scale));
                    if (q) dq = (int32_t)((uint32_t)q * (uint32_t)(int32_t)dequant[k]) >> log_scale;
                }
            } else if ((int64_t)a * wt >= ((int32_t)dequant[k] << (QM_BITS - (1 + log_scale)))) {
                const int64_t t = clamp64i((int64_t)a + rnd, INT16_MIN, INT16_MAX);
                q  = (int32_t)((t * wt * quant[k]) >> (16 - log_scale + QM_BITS));
                dq = (int32_t)((uint32_t)q * (uint32_t)deq_w) >> log_scale;
            }
        } else {
            const int shift = 16 - log_scale;
            if (qm || iqm) {
                if ((int64_t)a * wt >= ((int32_t)dequant[k] << (QM_BITS - (1 + log_scale)))) {
                    const int64_t t = (int64_t)a + rpot(round[k], log_scale);
                    q  = (int32_t)((t * quant[k] * wt) >> (shift + QM_BITS));
                    dq = (int32_t)((uint32_t)q * (uint32_t)deq_w) >> log_scale;
                }
            } else if ((int32_t)((uint32_t)a << (1 + log_scale)) >= (int32_t)dequant[k]) {
                const int64_t t = (int64_t)a + rpot(round[k], log_scale);
                q  = (int32_t)((t * quant[k]) >> shift);
                dq = (int32_t)((uint32_t)q * (uint32_t)(int32_t)dequant[k]) >> log_scale;
            }
        }
        if (q) {
            qcoeff[rc]  = (q ^ sign) - sign;
            dqcoeff[rc] = (dq ^ sign) - sign;
            eob         = i;
        }
    }
    *eob_ptr = (uint16_t)(eob + 1);
}

/* svt_handle_transform{64x64,64x32,32x64,64x16,16x64}[_N2_N4]_c (transforms.c:2362-2542): energy of the discarded
 * high-frequency area (0 for the N2/N4 forms) + in-place repack of 64-wide rows to stride 32. */
uint64_t oracle_handle_transform(int32_t *out, int w, int h, int n2n4) {
    uint64_t e = 0;
    if (!n2n4) {
        if (w == 64)
            for (int r = 0; r < (h > 32 ? 32 : h); r++)
                for (int c = 32; c < 64; c++) e += (uint64_t)((int64_t)out[r * 64 + c] * (int64_t)out[r * 64 + c]);
        if (h == 64)
            for (int r = 32; r < 64; r++)
                for (int c = 0; c < w; c++) e += (uint64_t)((int64_t)out[r * w + c] * (int64_t)out[r * w + c]);
    }
    if (w == 64)
        for (int r = 1; r < (h > 32 ? 32 : h); r++) memmove(out + r * 32, out + r * 64, 32 * sizeof(int32_t));
    return e;
}
