/* oracle_picprep.c -- TEST INFRASTRUCTURE ONLY (see oracle/README or DESIGN.md section 3): CPU restatement of the reference's
 * picture-preparation helpers that feed ME.  Pinned against the reference's own objects in
 * tests/test_picprep.py (svt_aom_downsample_2d_c, svt_aom_generate_padding from oracle/_ref/libsvtref.so). */
#include <stdint.h>
#include <string.h>

/* svt_aom_downsample_2d_c (pic_analysis_process.c:130-160): 2x2 zero-phase box at the centre of every decim_step x decim_step cell */
void oracle_downsample_2d(const uint8_t *in, uint32_t in_stride, uint32_t in_w, uint32_t in_h, uint8_t *out, uint32_t out_stride, uint32_t step) {
    const uint32_t half = step >> 1;
    const uint8_t *line = in + half * in_stride;
    for (uint32_t v = half; v < in_h; v += step) {
        const uint8_t *prev = line - in_stride;
        uint32_t       o    = 0;
        for (uint32_t h = half; h < in_w; h += step, o++) out[o] = (uint8_t)((prev[h - 1] + prev[h] + line[h - 1] + line[h] + 2u) >> 2);
        line += in_stride * step;
        out += out_stride;
    }
}

/* svt_aom_generate_padding (pic_operators.c:397-441): rows first (left / right), then whole rows up and down */
void oracle_generate_padding(uint8_t *base, uint32_t stride, uint32_t w, uint32_t h, uint32_t pad_w, uint32_t pad_h) {
    uint8_t *row = base + pad_w + pad_h * stride;
    for (uint32_t y = 0; y < h; y++, row += stride) {
        memset(row - pad_w, row[0], pad_w);
        memset(row + w, row[w - 1], pad_w);
    }
    uint8_t *top = base + pad_h * stride, *bot = base + (pad_h + h - 1) * stride;
    for (uint32_t k = 1; k <= pad_h; k++) {
        memcpy(top - k * stride, top, stride);
        memcpy(bot + k * stride, bot, stride);
    }
}
