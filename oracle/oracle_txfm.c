/*
 * oracle_txfm.c -- TEST INFRASTRUCTURE: plain-C restatement of the reference's forward / inverse 2-D transforms.
 * Never linked into the product.  Pinned against oracle/_ref (the real reference) in tests/test_oracle_pin_txfm.py
 * (every 1-D kernel, every 2-D size x allowed type x bit depth) and against tests/golden/txfm.npz.
 *
 * 1-D DCT/ADST flow graphs come from tools/gen_txfm.py (oracle_txfm1d_gen.h); the 4-point ADST, the identity
 * transforms and the 2-D drivers are restated here with the reference lines they follow
 * (paths relative to /root/reference/Source/Lib/Codec).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "oracle_txfm1d_gen.h"

/* ---- tables ------------------------------------------------------------------------------------------------- */
/* TxSize order of definitions.h (TX_4X4 .. TX_64X16) */
static const uint8_t TXW[19] = {4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64};
static const uint8_t TXH[19] = {4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16};
/* transforms.h:27-45 */
static const int8_t FWD_SHIFT[19][3] = {{2, 0, 0},  {2, -1, 0}, {2, -2, 0}, {2, -4, 0}, {0, -2, -2}, {2, -1, 0}, {2, -1, 0},
                                        {2, -2, 0}, {2, -2, 0}, {2, -4, 0}, {2, -4, 0}, {0, -2, -2}, {2, -4, -2}, {2, -1, 0},
                                        {2, -1, 0}, {2, -2, 0}, {2, -2, 0}, {0, -2, 0}, {2, -4, 0}};
/* transforms.h:47-50, indexed [log2(w)-2][log2(h)-2] */
static const int8_t FWD_COS_BIT_COL[5][5] = {{13, 13, 13, 0, 0}, {13, 13, 13, 12, 0}, {13, 13, 13, 12, 13}, {0, 13, 13, 12, 13}, {0, 0, 13, 12, 13}};
static const int8_t FWD_COS_BIT_ROW[5][5] = {{13, 13, 12, 0, 0}, {13, 13, 13, 12, 0}, {13, 13, 12, 13, 12}, {0, 12, 13, 12, 11}, {0, 0, 12, 11, 10}};
/* inv_transforms.c:17-35 */
static const int8_t INV_SHIFT[19][2] = {{0, -4},  {-1, -4}, {-2, -4}, {-2, -4}, {-2, -4}, {0, -4},  {0, -4},  {-1, -4}, {-1, -4}, {-1, -4},
                                        {-1, -4}, {-1, -4}, {-1, -4}, {-1, -4}, {-1, -4}, {-2, -4}, {-2, -4}, {-2, -4}, {-2, -4}};
#define INV_COS_BIT 12 /* inv_transforms.h */
/* sinpi[i] = round(sin(i*pi/9) * 2*sqrt(2)/3 * 2^bit), bits 10..16 (svt_aom_eb_av1_sinpi_arr_data, inv_transforms.c:3228) */
static const int32_t SINPI[7][5] = {{0, 330, 621, 836, 951},        {0, 660, 1241, 1672, 1901},     {0, 1321, 2482, 3344, 3803},
                                    {0, 2642, 4964, 6689, 7606},    {0, 5283, 9929, 13377, 15212},  {0, 10566, 19858, 26755, 30424},
                                    {0, 21133, 39716, 53510, 60849}};
const int32_t *oracle_cospi_table(int bit) { return o_cospi_tab[bit - 10]; }
const int32_t *oracle_sinpi_table(int bit) { return SINPI[bit - 10]; }

enum { K_DCT = 0, K_ADST = 1, K_IDTX = 2 };
/* vtx_tab / htx_tab / set_flip_cfg of inv_transforms.h: TxType -> (column kind, row kind, ud_flip, lr_flip) */
static const uint8_t COL_KIND[16] = {K_DCT, K_ADST, K_DCT, K_ADST, K_ADST, K_DCT, K_ADST, K_ADST, K_ADST, K_IDTX, K_DCT, K_IDTX, K_ADST, K_IDTX, K_ADST, K_IDTX};
static const uint8_t ROW_KIND[16] = {K_DCT, K_DCT, K_ADST, K_ADST, K_DCT, K_ADST, K_ADST, K_ADST, K_ADST, K_IDTX, K_IDTX, K_DCT, K_IDTX, K_ADST, K_IDTX, K_ADST};
static const uint8_t UD_FLIP[16]  = {0, 0, 0, 0, 1, 0, 1, 0, 1, 0, 0, 0, 0, 0, 1, 0};
static const uint8_t LR_FLIP[16]  = {0, 0, 0, 0, 0, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 1};

#define NEW_SQRT2 5793     /* inv_transforms.h:247-251 */
#define NEW_INV_SQRT2 2896
#define NEW_SQRT2_BITS 12

static inline int32_t rshift64(int64_t v, int bit) { return (int32_t)((v + ((int64_t)1 << (bit - 1))) >> bit); }

/* ---- 4-point ADST (transforms.c:1415-1502 forward, inv_transforms.c:722-800 inverse) -------------------------- */
#define MUL32(a, b) ((int32_t)((uint32_t)(a) * (uint32_t)(b))) /* the reference multiplies and adds in int32 (wrapping) */
#define ADD32(a, b) ((int32_t)((uint32_t)(a) + (uint32_t)(b)))
#define SUB32(a, b) ((int32_t)((uint32_t)(a) - (uint32_t)(b)))
static void o_fadst4(const int32_t *in, int32_t *out, int bit) {
    const int32_t *s = SINPI[bit - 10];
    const int32_t  x0 = in[0], x1 = in[1], x2 = in[2], x3 = in[3];
    const int32_t  a0 = ADD32(ADD32(MUL32(s[1], x0), MUL32(s[2], x1)), MUL32(s[4], x3)); /* s0 + s2 + s5 */
    const int32_t  a1 = MUL32(s[3], SUB32(ADD32(x0, x1), x3));                            /* sinpi3 * (x0 + x1 - x3) */
    const int32_t  a2 = ADD32(SUB32(MUL32(s[4], x0), MUL32(s[1], x1)), MUL32(s[2], x3)); /* s1 - s3 + s6 */
    const int32_t  a3 = MUL32(s[3], x2);                                                  /* s4 */
    out[0] = rshift64(ADD32(a0, a3), bit);
    out[1] = rshift64(a1, bit);
    out[2] = rshift64(SUB32(a2, a3), bit);
    out[3] = rshift64(ADD32(SUB32(a2, a0), a3), bit);
}
static void o_iadst4(const int32_t *in, int32_t *out, int bit) {
    const int32_t *s = SINPI[bit - 10];
    const int32_t  x0 = in[0], x1 = in[1], x2 = in[2], x3 = in[3];
    const int32_t  s0 = ADD32(ADD32(MUL32(s[1], x0), MUL32(s[4], x2)), MUL32(s[2], x3));
    const int32_t  s1 = SUB32(SUB32(MUL32(s[2], x0), MUL32(s[1], x2)), MUL32(s[4], x3));
    const int32_t  s3 = MUL32(s[3], x1);
    const int32_t  s2 = MUL32(s[3], ADD32(SUB32(x0, x2), x3));
    out[0] = rshift64(ADD32(s0, s3), bit);
    out[1] = rshift64(ADD32(s1, s3), bit);
    out[2] = rshift64(s2, bit);
    out[3] = rshift64(SUB32(ADD32(s0, s1), s3), bit);
}

/* ---- identity (transforms.c:2205-2234, inv_transforms.c:2331-2366): x sqrt2, x2, x2sqrt2, x4, x4sqrt2 ---------- */
static void o_identity(const int32_t *in, int32_t *out, int n) {
    for (int i = 0; i < n; i++) switch (n) {
        case 4: out[i] = rshift64((int64_t)in[i] * NEW_SQRT2, NEW_SQRT2_BITS); break;
        case 8: out[i] = (int32_t)((int64_t)in[i] * 2); break;
        case 16: out[i] = rshift64((int64_t)in[i] * 2 * NEW_SQRT2, NEW_SQRT2_BITS); break;
        case 32: out[i] = (int32_t)((int64_t)in[i] * 4); break;
        default: out[i] = rshift64((int64_t)in[i] * 4 * NEW_SQRT2, NEW_SQRT2_BITS); break;
        }
}

void oracle_fwd_txfm1d(int kind, int n, const int32_t *in, int32_t *out, int cos_bit) {
    if (kind == K_IDTX) { o_identity(in, out, n); return; }
    if (kind == K_DCT) switch (n) {
        case 4: o_fdct4(in, out, cos_bit); return;
        case 8: o_fdct8(in, out, cos_bit); return;
        case 16: o_fdct16(in, out, cos_bit); return;
        case 32: o_fdct32(in, out, cos_bit); return;
        default: o_fdct64(in, out, cos_bit); return;
        }
    switch (n) {
    case 4: o_fadst4(in, out, cos_bit); return;
    case 8: o_fadst8(in, out, cos_bit); return;
    default: o_fadst16(in, out, cos_bit); return;
    }
}
void oracle_inv_txfm1d(int kind, int n, const int32_t *in, int32_t *out, int cos_bit, int clamp_bit) {
    if (kind == K_IDTX) { o_identity(in, out, n); return; }
    if (kind == K_DCT) switch (n) {
        case 4: o_idct4(in, out, cos_bit, clamp_bit); return;
        case 8: o_idct8(in, out, cos_bit, clamp_bit); return;
        case 16: o_idct16(in, out, cos_bit, clamp_bit); return;
        case 32: o_idct32(in, out, cos_bit, clamp_bit); return;
        default: o_idct64(in, out, cos_bit, clamp_bit); return;
        }
    switch (n) {
    case 4: o_iadst4(in, out, cos_bit); return;
    case 8: o_iadst8(in, out, cos_bit, clamp_bit); return;
    case 16: o_iadst16(in, out, cos_bit, clamp_bit); return;
    default: o_iadst32(in, out, cos_bit, clamp_bit); return; /* av1_iadst32_new, inv_transforms.c:1119-1552: never signalled by AV1, accepted by the `_c` functions */
    }
}

static int lg2(int v) { int r = 0; while ((1 << r) < v) r++; return r; }
/* svt_av1_round_shift_array_c, inv_transforms.c:2421-2433 */
static void round_shift_array(int32_t *a, int n, int bit) {
    if (bit == 0) return;
    for (int i = 0; i < n; i++) a[i] = bit > 0 ? rshift64(a[i], bit) : (int32_t)((uint32_t)a[i] * (1u << (-bit)));
}

/* av1_tranform_two_d_core_c (transforms.c:2259-2324) + svt_aom_transform_config (:2344-2360).  `pf` = 0 default,
 * 1 = N2, 2 = N4: only the top-left 1/2 (1/4) x 1/2 (1/4) coefficients are kept, the rest zero (transforms.c:5202-5273,
 * and what test/FwdTxfm2dAsmTest.cc:333-356 compares against). */
void oracle_fwd_txfm2d(const int16_t *input, int32_t *output, uint32_t stride, int tx_type, int tx_size, int bd, int pf) {
    (void)bd; /* bit depth only feeds range asserts in the reference */
    const int w = TXW[tx_size], h = TXH[tx_size];
    const int8_t *shift = FWD_SHIFT[tx_size];
    const int cbc = FWD_COS_BIT_COL[lg2(w) - 2][lg2(h) - 2], cbr = FWD_COS_BIT_ROW[lg2(w) - 2][lg2(h) - 2];
    const int ck = COL_KIND[tx_type], rk = ROW_KIND[tx_type];
    int32_t *buf = (int32_t *)malloc(sizeof(int32_t) * w * h);
    int32_t  tin[64], tout[64];
    for (int c = 0; c < w; c++) {
        for (int r = 0; r < h; r++) tin[r] = input[(UD_FLIP[tx_type] ? (h - 1 - r) : r) * stride + c];
        round_shift_array(tin, h, -shift[0]);
        oracle_fwd_txfm1d(ck, h, tin, tout, cbc);
        round_shift_array(tout, h, -shift[1]);
        const int cc = LR_FLIP[tx_type] ? (w - 1 - c) : c;
        for (int r = 0; r < h; r++) buf[r * w + cc] = tout[r];
    }
    const int rect1 = (w == 2 * h) || (h == 2 * w);
    for (int r = 0; r < h; r++) {
        oracle_fwd_txfm1d(rk, w, buf + r * w, output + r * w, cbr);
        round_shift_array(output + r * w, w, -shift[2]);
        if (rect1)
            for (int c = 0; c < w; c++) output[r * w + c] = rshift64((int64_t)output[r * w + c] * NEW_SQRT2, NEW_SQRT2_BITS);
    }
    if (pf) {
        const int kw = w >> pf, kh = h >> pf;
        for (int r = 0; r < h; r++)
            for (int c = 0; c < w; c++)
                if (r >= kh || c >= kw) output[r * w + c] = 0;
    }
    free(buf);
}

/* inv_txfm2d_add_c (inv_transforms.c:2459-2535), 64-point inputs zero-extended from the packed 32-wide layout
 * (:2567-2580, :2628-2690).  output_r = prediction, output_w = reconstruction (may alias). */
void oracle_inv_txfm2d_add(const int32_t *input, const uint16_t *out_r, int stride_r, uint16_t *out_w, int stride_w, int tx_type,
                           int tx_size, int bd) {
    const int w = TXW[tx_size], h = TXH[tx_size];
    const int8_t *shift = INV_SHIFT[tx_size];
    const int ck = COL_KIND[tx_type], rk = ROW_KIND[tx_type];
    const int row_clamp = bd + 8, col_clamp = (bd + 6 > 16) ? bd + 6 : 16;
    const int rect1 = (w == 2 * h) || (h == 2 * w);
    const int in_w = w > 32 ? 32 : w, in_h = h > 32 ? 32 : h; /* packed input holds at most 32x32 */
    int32_t *buf = (int32_t *)calloc((size_t)w * h, sizeof(int32_t));
    int32_t  tin[64], tout[64];
    const int64_t rmax = ((int64_t)1 << (row_clamp - 1)) - 1, rmin = -((int64_t)1 << (row_clamp - 1));
    const int64_t cmax = ((int64_t)1 << (col_clamp - 1)) - 1, cmin = -((int64_t)1 << (col_clamp - 1));
    for (int r = 0; r < h; r++) {
        for (int c = 0; c < w; c++) {
            int32_t v = (r < in_h && c < in_w) ? input[r * in_w + c] : 0;
            if (rect1) v = rshift64((int64_t)v * NEW_INV_SQRT2, NEW_SQRT2_BITS);
            tin[c] = (int32_t)(v < rmin ? rmin : (v > rmax ? rmax : v));
        }
        oracle_inv_txfm1d(rk, w, tin, buf + r * w, INV_COS_BIT, row_clamp);
        round_shift_array(buf + r * w, w, -shift[0]);
    }
    for (int c = 0; c < w; c++) {
        const int cc = LR_FLIP[tx_type] ? (w - 1 - c) : c;
        for (int r = 0; r < h; r++) {
            const int32_t v = buf[r * w + cc];
            tin[r] = (int32_t)(v < cmin ? cmin : (v > cmax ? cmax : v));
        }
        oracle_inv_txfm1d(ck, h, tin, tout, INV_COS_BIT, col_clamp);
        round_shift_array(tout, h, -shift[1]);
        for (int r = 0; r < h; r++) {
            const int32_t res = tout[UD_FLIP[tx_type] ? (h - 1 - r) : r];
            int32_t px = (int32_t)((uint32_t)out_r[r * stride_r + c] + (uint32_t)res); /* highbd_clip_pixel_add */
            const int32_t mx = (1 << bd) - 1;
            out_w[r * stride_w + c] = (uint16_t)(px < 0 ? 0 : (px > mx ? mx : px));
        }
    }
    free(buf);
}

int oracle_tx_width(int tx_size) { return TXW[tx_size]; }
int oracle_tx_height(int tx_size) { return TXH[tx_size]; }

/* ---- lossless mode: 4x4 Walsh-Hadamard ----------------------------------------------------------------------------------
 * Forward: svt_av1_fwht4x4_c (Source/Lib/Codec/transforms.c:3099-3152).  Each 1-D pass maps (a, b, c, d) to
 * (a', c', d', b') -- the reference's output order -- and the second pass scales by UNIT_QUANT_FACTOR = 4. */
static void wht_fwd_1d(int64_t a, int64_t b, int64_t c, int64_t d, int64_t o[4]) {
    a += b;
    d -= c;
    const int64_t e = (a - d) >> 1;
    b = e - b;
    c = e - c;
    a -= c;
    d += b;
    o[0] = a; o[1] = c; o[2] = d; o[3] = b;
}
void oracle_fwht4x4(const int16_t *input, int32_t *output, uint32_t stride) {
    int64_t t[16], o[4];
    for (int i = 0; i < 4; i++) { /* columns of the input become rows of t */
        wht_fwd_1d(input[0 * stride + i], input[1 * stride + i], input[2 * stride + i], input[3 * stride + i], o);
        for (int k = 0; k < 4; k++) t[4 * i + k] = (int32_t)o[k];
    }
    for (int i = 0; i < 4; i++) {
        wht_fwd_1d(t[i], t[4 + i], t[8 + i], t[12 + i], o);
        for (int k = 0; k < 4; k++) output[4 * k + i] = (int32_t)(o[k] * 4);
    }
}
/* Inverse + reconstruction: svt_av1_highbd_iwht4x4_16_add_c (eob > 1) and svt_av1_highbd_iwht4x4_1_add_c (eob <= 1)
 * (inv_transforms.c:2735-2825), selected by highbd_iwht4x4_add (:2826-2831). */
static void wht_inv_1d(int32_t a, int32_t c, int32_t d, int32_t b, int32_t o[4]) {
    a += c;
    d -= b;
    const int32_t e = (a - d) >> 1;
    b = e - b;
    c = e - c;
    a -= b;
    d += c;
    o[0] = a; o[1] = b; o[2] = c; o[3] = d;
}
void oracle_iwht4x4_add(const int32_t *input, const uint16_t *out_r, int stride_r, uint16_t *out_w, int stride_w, int eob, int bd) {
    int32_t t[16], o[4];
    if (eob > 1) {
        for (int i = 0; i < 4; i++) {
            wht_inv_1d(input[4 * i] >> 2, input[4 * i + 1] >> 2, input[4 * i + 2] >> 2, input[4 * i + 3] >> 2, o);
            for (int k = 0; k < 4; k++) t[4 * i + k] = o[k];
        }
        for (int i = 0; i < 4; i++) {
            wht_inv_1d(t[i], t[4 + i], t[8 + i], t[12 + i], o);
            for (int k = 0; k < 4; k++) t[4 * k + i] = o[k];
        }
    } else {
        int32_t a = input[0] >> 2;
        const int32_t e = a >> 1;
        a -= e;
        const int32_t row[4] = {a, e, e, e};
        for (int i = 0; i < 4; i++) {
            const int32_t e1 = row[i] >> 1;
            t[i] = row[i] - e1;
            t[4 + i] = t[8 + i] = t[12 + i] = e1;
        }
    }
    const int32_t mx = (1 << bd) - 1;
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) {
            const int32_t px = (int32_t)out_r[r * stride_r + c] + t[4 * r + c];
            out_w[r * stride_w + c] = (uint16_t)(px < 0 ? 0 : (px > mx ? mx : px));
        }
}
