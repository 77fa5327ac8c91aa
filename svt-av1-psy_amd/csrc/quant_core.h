// quant_core.h -- the per-coefficient quantizer shared by quant.hip and txfm_fused.hip (the four reference quantizers, full_loop.c:29-453).
#pragma once
#include <stdint.h>
#include "../../include/svtav1_hip.h"

namespace {

constexpr int QM_BITS = 5; // AOM_QM_BITS
__device__ __forceinline__ int32_t q_rpot(const int32_t v, const int n) { return (v + ((1 << n) >> 1)) >> n; }
__device__ __forceinline__ int64_t clamp_i16(const int64_t v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }

struct QOut {
    int32_t q, dq;
};
// MODE 0: svt_aom_quantize_b_c_ii (full_loop.c:29-79); 1: svt_aom_highbd_quantize_b_c (:149-198);
//      2: quantize_fp_helper_c (:282-342);             3: highbd_quantize_fp_helper_c (:387-453)
template <int MODE, bool QM>
__device__ __forceinline__ QOut quant_one(const int32_t c, const int k, const SvtHipQuantParams& P, const int32_t wt, const int32_t iwt) {
    const int     ls   = P.log_scale;
    const int32_t sign = c < 0 ? -1 : 0;
    const int32_t a    = (c ^ sign) - sign;
    const int32_t dqt  = P.dequant[k];
    const int32_t deqw = QM ? ((dqt * iwt + (1 << (QM_BITS - 1))) >> QM_BITS) : dqt;
    int32_t       q    = 0;
    if (MODE == 0) {
        const int32_t zb = q_rpot(P.zbin[k], ls);
        if ((int32_t)((uint32_t)a * (uint32_t)wt) >= (zb << QM_BITS)) {
            int64_t tmp = clamp_i16((int64_t)a + q_rpot(P.round[k], ls));
            tmp *= wt;
            q = (int32_t)(((((tmp * P.quant[k]) >> 16) + tmp) * P.quant_shift[k]) >> (16 - ls + QM_BITS));
        }
    } else if (MODE == 1) {
        const int32_t zb = q_rpot(P.zbin[k], ls);
        const int32_t cw = (int32_t)((uint32_t)c * (uint32_t)wt);
        if (cw >= zb * (1 << QM_BITS) || cw <= -zb * (1 << QM_BITS)) {
            const int64_t tw = ((int64_t)a + q_rpot(P.round[k], ls)) * wt;
            const int64_t t2 = ((tw * P.quant[k]) >> 16) + tw;
            q = (int32_t)((t2 * P.quant_shift[k]) >> (16 - ls + QM_BITS));
        }
    } else if (MODE == 2) {
        const int32_t rnd = q_rpot(P.round[k], ls);
        if (!QM) {
            if (((int64_t)a << (1 + ls)) >= dqt) q = (int32_t)((clamp_i16((int64_t)a + rnd) * P.quant[k]) >> (16 - ls));
        } else if ((int64_t)a * wt >= (dqt << (QM_BITS - (1 + ls)))) {
            q = (int32_t)((clamp_i16((int64_t)a + rnd) * wt * P.quant[k]) >> (16 - ls + QM_BITS));
        }
    } else {
        if (QM) {
            if ((int64_t)a * wt >= (dqt << (QM_BITS - (1 + ls)))) q = (int32_t)((((int64_t)a + q_rpot(P.round[k], ls)) * P.quant[k] * wt) >> (16 - ls + QM_BITS));
        } else if ((int32_t)((uint32_t)a << (1 + ls)) >= dqt) {
            q = (int32_t)((((int64_t)a + q_rpot(P.round[k], ls)) * P.quant[k]) >> (16 - ls));
        }
    }
    const int32_t dq = (int32_t)((uint32_t)q * (uint32_t)deqw) >> ls;
    QOut          o;
    o.q  = (q ^ sign) - sign;
    o.dq = q ? ((dq ^ sign) - sign) : 0;
    return o;
}

} // namespace
