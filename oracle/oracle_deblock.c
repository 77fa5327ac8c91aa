/* oracle_deblock.c -- TEST INFRASTRUCTURE ONLY: CPU restatement of the AV1 deblocking edge filters:
 * svt_aom_lpf_{horizontal,vertical}_{4,6,8,14}_c and svt_aom_highbd_lpf_* (Codec/deblocking_common.c:141-865).  One routine covers both
 * families: the 8-bit functions are the high-bit-depth ones at bd = 8 (limits << 0, clamp to [-128, 127], `^ 0x80` == `- 0x80` on int8).
 * Pinned against the reference's sixteen `_c` functions in tests/test_deblock.py. */
#include <stdint.h>
#include <stdlib.h>

static int sclamp(int t, int bd) { /* signed_char_clamp_high, deblocking_common.c:28-35 */
    const int lo = -(128 << (bd - 8)), hi = (128 << (bd - 8)) - 1;
    return t < lo ? lo : (t > hi ? hi : t);
}
static int rpot(int v, int n) { return (v + ((1 << n) >> 1)) >> n; }

/* p[k] = k-th pixel on the p side (p[0] next to the edge), q[k] likewise; len in {4, 6, 8, 14} */
static void lpf_px(int *p, int *q, int len, int blimit, int limit, int thresh, int bd) {
    const int sh = bd - 8, limit16 = limit << sh, blimit16 = blimit << sh, thresh16 = thresh << sh, one16 = 1 << sh;
    int mask = 0, flat = 0, flat2 = 0; /* 0 / 1 here; the reference keeps 0 / -1 bytes */
    {   /* filter_mask2 / filter_mask3_chroma / filter_mask (:141-171, :376-401, :659-669) */
        const int taps = len == 4 ? 2 : (len == 6 ? 3 : 4);
        int bad = 0;
        for (int k = 1; k < taps; k++) bad |= (abs(p[k] - p[k - 1]) > limit16) | (abs(q[k] - q[k - 1]) > limit16);
        bad |= abs(p[0] - q[0]) * 2 + abs(p[1] - q[1]) / 2 > blimit16;
        mask = !bad;
    }
    if (len >= 6) { /* flat_mask3_chroma / flat_mask4 with thresh = 1 (:173-205, :403-416) */
        const int taps = len == 6 ? 3 : 4;
        int bad = 0;
        for (int k = 1; k < taps; k++) bad |= (abs(p[k] - p[0]) > one16) | (abs(q[k] - q[0]) > one16);
        flat = !bad;
    }
    if (len == 14) { /* flat_mask4(1, p6, p5, p4, p0, q0, q4, q5, q6) (:797) */
        int bad = 0;
        for (int k = 4; k < 7; k++) bad |= (abs(p[k] - p[0]) > one16) | (abs(q[k] - q[0]) > one16);
        flat2 = !bad;
    }
    if (len == 14 && flat2 && flat && mask) { /* 13-tap filter [1,1,1,1,1,2,2,2,1,1,1,1,1] (:762-785) */
        const int p6 = p[6], p5 = p[5], p4 = p[4], p3 = p[3], p2 = p[2], p1 = p[1], p0 = p[0];
        const int q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3], q4 = q[4], q5 = q[5], q6 = q[6];
        p[5] = rpot(p6 * 7 + p5 * 2 + p4 * 2 + p3 + p2 + p1 + p0 + q0, 4);
        p[4] = rpot(p6 * 5 + p5 * 2 + p4 * 2 + p3 * 2 + p2 + p1 + p0 + q0 + q1, 4);
        p[3] = rpot(p6 * 4 + p5 + p4 * 2 + p3 * 2 + p2 * 2 + p1 + p0 + q0 + q1 + q2, 4);
        p[2] = rpot(p6 * 3 + p5 + p4 + p3 * 2 + p2 * 2 + p1 * 2 + p0 + q0 + q1 + q2 + q3, 4);
        p[1] = rpot(p6 * 2 + p5 + p4 + p3 + p2 * 2 + p1 * 2 + p0 * 2 + q0 + q1 + q2 + q3 + q4, 4);
        p[0] = rpot(p6 + p5 + p4 + p3 + p2 + p1 * 2 + p0 * 2 + q0 * 2 + q1 + q2 + q3 + q4 + q5, 4);
        q[0] = rpot(p5 + p4 + p3 + p2 + p1 + p0 * 2 + q0 * 2 + q1 * 2 + q2 + q3 + q4 + q5 + q6, 4);
        q[1] = rpot(p4 + p3 + p2 + p1 + p0 + q0 * 2 + q1 * 2 + q2 * 2 + q3 + q4 + q5 + q6 * 2, 4);
        q[2] = rpot(p3 + p2 + p1 + p0 + q0 + q1 * 2 + q2 * 2 + q3 * 2 + q4 + q5 + q6 * 3, 4);
        q[3] = rpot(p2 + p1 + p0 + q0 + q1 + q2 * 2 + q3 * 2 + q4 * 2 + q5 + q6 * 4, 4);
        q[4] = rpot(p1 + p0 + q0 + q1 + q2 + q3 * 2 + q4 * 2 + q5 * 2 + q6 * 5, 4);
        q[5] = rpot(p0 + q0 + q1 + q2 + q3 + q4 * 2 + q5 * 2 + q6 * 7, 4);
    } else if (len >= 8 && flat && mask) { /* 7-tap filter [1,1,1,2,1,1,1] (:289-304) */
        const int p3 = p[3], p2 = p[2], p1 = p[1], p0 = p[0], q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
        p[2] = rpot(p3 + p3 + p3 + 2 * p2 + p1 + p0 + q0, 3);
        p[1] = rpot(p3 + p3 + p2 + 2 * p1 + p0 + q0 + q1, 3);
        p[0] = rpot(p3 + p2 + p1 + 2 * p0 + q0 + q1 + q2, 3);
        q[0] = rpot(p2 + p1 + p0 + 2 * q0 + q1 + q2 + q3, 3);
        q[1] = rpot(p1 + p0 + q0 + 2 * q1 + q2 + q3 + q3, 3);
        q[2] = rpot(p0 + q0 + q1 + 2 * q2 + q3 + q3 + q3, 3);
    } else if (len == 6 && flat && mask) { /* 5-tap filter [1,2,2,2,1] (:274-287) */
        const int p2 = p[2], p1 = p[1], p0 = p[0], q0 = q[0], q1 = q[1], q2 = q[2];
        p[1] = rpot(p2 * 3 + p1 * 2 + p0 * 2 + q0, 3);
        p[0] = rpot(p2 + p1 * 2 + p0 * 2 + q0 * 2 + q1, 3);
        q[0] = rpot(p1 + p0 * 2 + q0 * 2 + q1 * 2 + q2, 3);
        q[1] = rpot(p0 + q0 * 2 + q1 * 2 + q2 * 3, 3);
    } else { /* filter4 (:214-240, :426-458) */
        const int off = 0x80 << sh, m = mask ? -1 : 0;
        const int ps1 = p[1] - off, ps0 = p[0] - off, qs0 = q[0] - off, qs1 = q[1] - off;
        const int hev = ((abs(p[1] - p[0]) > thresh16) | (abs(q[1] - q[0]) > thresh16)) ? -1 : 0;
        int f = sclamp(ps1 - qs1, bd) & hev;
        f = sclamp(f + 3 * (qs0 - ps0), bd) & m;
        const int f1 = sclamp(f + 4, bd) >> 3, f2 = sclamp(f + 3, bd) >> 3;
        q[0] = sclamp(qs0 - f1, bd) + off;
        p[0] = sclamp(ps0 + f2, bd) + off;
        f    = rpot(f1, 1) & ~hev;
        q[1] = sclamp(qs1 - f, bd) + off;
        p[1] = sclamp(ps1 + f, bd) + off;
    }
}

/* one 4-pixel edge segment; s points at q0 of the first pixel; vertical != 0: the edge is a column boundary (filter along x) */
void oracle_lpf(void *s, int pitch, int is16, int vertical, int len, int blimit, int limit, int thresh, int bd) {
    const int step = vertical ? pitch : 1, across = vertical ? 1 : pitch, half = len == 14 ? 7 : len / 2;
    for (int i = 0; i < 4; i++) {
        int p[7], q[7];
        for (int k = 0; k < half; k++) {
            const long po = (long)i * step - (long)(k + 1) * across, qo = (long)i * step + (long)k * across;
            p[k] = is16 ? ((uint16_t *)s)[po] : ((uint8_t *)s)[po];
            q[k] = is16 ? ((uint16_t *)s)[qo] : ((uint8_t *)s)[qo];
        }
        lpf_px(p, q, len, blimit, limit, thresh, bd);
        for (int k = 0; k < half; k++) {
            const long po = (long)i * step - (long)(k + 1) * across, qo = (long)i * step + (long)k * across;
            if (is16) { ((uint16_t *)s)[po] = (uint16_t)p[k]; ((uint16_t *)s)[qo] = (uint16_t)q[k]; }
            else { ((uint8_t *)s)[po] = (uint8_t)p[k]; ((uint8_t *)s)[qo] = (uint8_t)q[k]; }
        }
    }
}
