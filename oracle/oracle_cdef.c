/*
 * oracle_cdef.c -- TEST INFRASTRUCTURE: plain-C restatement of the reference's CDEF kernels and per-64x64 driver.
 * Pinned against oracle/_ref in tests/test_oracle_pin_cdef.py.  Paths relative to /root/reference/Source/Lib/Codec.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define BSTRIDE 144        /* CDEF_BSTRIDE, cdef.h:35 */
#define VERY_LARGE 0x7f7f  /* CDEF_VERY_LARGE, cdef.h:38 */
#define VBORDER 3
#define HBORDER 8

static int msb(uint32_t n) { int r = 0; while (n >>= 1) r++; return r; }

/* cdef.c:85-91 */
static int constrain(int diff, int threshold, int damping) {
    if (!threshold) return 0;
    int shift = damping - msb((uint32_t)threshold);
    if (shift < 0) shift = 0;
    const int ad = abs(diff);
    int       v  = threshold - (ad >> shift);
    if (v < 0) v = 0;
    if (ad < v) v = ad;
    return diff < 0 ? -v : v;
}

/* Cdef_Directions (AV1 spec 7.15.3; cdef.c:99-120): offsets of the two taps along direction d, as (dy, dx) */
static const int8_t DIR_DY[8][2] = {{-1, -2}, {0, -1}, {0, 0}, {0, 1}, {1, 2}, {1, 2}, {1, 2}, {1, 2}};
static const int8_t DIR_DX[8][2] = {{1, 2}, {1, 2}, {1, 2}, {1, 2}, {1, 2}, {0, 1}, {0, 0}, {0, -1}};
static int dir_off(int dir, int k, int stride) { dir &= 7; return DIR_DY[dir][k] * stride + DIR_DX[dir][k]; }

/* svt_aom_cdef_find_dir_c, cdef.c:150-210 */
uint8_t oracle_cdef_find_dir(const uint16_t *img, int stride, int32_t *var, int coeff_shift) {
    static const int div_table[9] = {0, 840, 420, 280, 210, 168, 140, 120, 105};
    int32_t cost[8] = {0}, partial[8][15];
    memset(partial, 0, sizeof(partial));
    for (int i = 0; i < 8; i++)
        for (int j = 0; j < 8; j++) {
            const int x = (img[i * stride + j] >> coeff_shift) - 128;
            partial[0][i + j] += x;
            partial[1][i + j / 2] += x;
            partial[2][i] += x;
            partial[3][3 + i - j / 2] += x;
            partial[4][7 + i - j] += x;
            partial[5][3 - i / 2 + j] += x;
            partial[6][j] += x;
            partial[7][i / 2 + j] += x;
        }
    for (int i = 0; i < 8; i++) {
        cost[2] += partial[2][i] * partial[2][i];
        cost[6] += partial[6][i] * partial[6][i];
    }
    cost[2] *= div_table[8];
    cost[6] *= div_table[8];
    for (int i = 0; i < 7; i++) {
        cost[0] += (partial[0][i] * partial[0][i] + partial[0][14 - i] * partial[0][14 - i]) * div_table[i + 1];
        cost[4] += (partial[4][i] * partial[4][i] + partial[4][14 - i] * partial[4][14 - i]) * div_table[i + 1];
    }
    cost[0] += partial[0][7] * partial[0][7] * div_table[8];
    cost[4] += partial[4][7] * partial[4][7] * div_table[8];
    for (int i = 1; i < 8; i += 2) {
        for (int j = 0; j < 5; j++) cost[i] += partial[i][3 + j] * partial[i][3 + j];
        cost[i] *= div_table[8];
        for (int j = 0; j < 3; j++) cost[i] += (partial[i][j] * partial[i][j] + partial[i][10 - j] * partial[i][10 - j]) * div_table[2 * j + 2];
    }
    int32_t best = 0;
    int     bd   = 0;
    for (int i = 0; i < 8; i++)
        if (cost[i] > best) { best = cost[i]; bd = i; }
    *var = (best - cost[(bd + 4) & 7]) >> 10;
    return (uint8_t)bd;
}

/* svt_cdef_filter_block_c, cdef.c:253-306.  bw/bh = 4 or 8 (BLOCK_4X4/4X8/8X4/8X8); exactly one of dst8/dst16. */
void oracle_cdef_filter_block(uint8_t *dst8, uint16_t *dst16, int dstride, const uint16_t *in, int pri_strength, int sec_strength,
                              int dir, int pri_damping, int sec_damping, int bw, int bh, int coeff_shift, int subsampling) {
    static const int pri_taps[2][2] = {{4, 2}, {3, 3}}, sec_taps[2][2] = {{2, 1}, {2, 1}};
    const int *pt = pri_taps[(pri_strength >> coeff_shift) & 1], *st = sec_taps[(pri_strength >> coeff_shift) & 1];
    for (int i = 0; i < bh; i += subsampling)
        for (int j = 0; j < bw; j++) {
            const int16_t x   = (int16_t)in[i * BSTRIDE + j];
            int16_t       sum = 0;
            int           mx = x, mn = x;
            for (int k = 0; k < 2; k++) {
                const int     po = dir_off(dir, k, BSTRIDE), s0o = dir_off(dir + 2, k, BSTRIDE), s1o = dir_off(dir - 2 + 8, k, BSTRIDE);
                const int16_t p[2] = {(int16_t)in[i * BSTRIDE + j + po], (int16_t)in[i * BSTRIDE + j - po]};
                const int16_t s[4] = {(int16_t)in[i * BSTRIDE + j + s0o], (int16_t)in[i * BSTRIDE + j - s0o],
                                      (int16_t)in[i * BSTRIDE + j + s1o], (int16_t)in[i * BSTRIDE + j - s1o]};
                for (int t = 0; t < 2; t++) {
                    sum = (int16_t)(sum + (int16_t)(pt[k] * constrain(p[t] - x, pri_strength, pri_damping)));
                    if (p[t] != VERY_LARGE && p[t] > mx) mx = p[t];
                    if (p[t] < mn) mn = p[t];
                }
                for (int t = 0; t < 4; t++) {
                    sum = (int16_t)(sum + (int16_t)(st[k] * constrain(s[t] - x, sec_strength, sec_damping)));
                    if (s[t] != VERY_LARGE && s[t] > mx) mx = s[t];
                    if (s[t] < mn) mn = s[t];
                }
            }
            int y = x + ((8 + sum - (sum < 0)) >> 4);
            y     = y < mn ? mn : (y > mx ? mx : y);
            if (dst8) dst8[i * dstride + j] = (uint8_t)y;
            else dst16[i * dstride + j] = (uint16_t)(int16_t)y;
        }
}

/* adjust_strength, cdef.c:130-134 */
static int adjust_strength(int strength, int32_t var) {
    const int i = (var >> 6) ? (msb((uint32_t)(var >> 6)) < 12 ? msb((uint32_t)(var >> 6)) : 12) : 0;
    return var ? (strength * (4 + i) + 8) >> 4 : 0;
}

/* svt_cdef_filter_fb, cdef.c:339-430.  `in` points at the first pixel of the 64x64 area inside a BSTRIDE-pitched tile
 * whose out-of-frame halo holds VERY_LARGE.  dlist = (by, bx) pairs of the non-skip 8x8 units.  dstride == 0 selects the
 * packed per-block output used by the strength search.  dir/var are [16][16] as in the reference. */
void oracle_cdef_filter_fb(uint8_t *dst8, uint16_t *dst16, int dstride, const uint16_t *in, int xdec, int ydec, uint8_t dir[16][16],
                           int *dirinit, int32_t var[16][16], int pli, const uint8_t *dlist /* by,bx pairs */, int cdef_count, int level,
                           int sec_strength, int pri_damping, int sec_damping, int coeff_shift, int subsampling) {
    const int pri_strength = level << coeff_shift;
    sec_strength <<= coeff_shift;
    sec_damping += coeff_shift - (pli != 0);
    pri_damping += coeff_shift - (pli != 0);
    const int bsx = 3 - xdec, bsy = 3 - ydec, bw = 1 << bsx, bh = 1 << bsy;
    if (!dstride && pri_strength == 0 && sec_strength == 0) { /* zero strength in search mode: plain copy (cdef.c:353-379) */
        for (int bi = 0; bi < cdef_count; bi++) {
            const int       by = dlist[2 * bi] << bsy, bx = dlist[2 * bi + 1] << bsx;
            const uint16_t *s = in + by * BSTRIDE + bx;
            for (int iy = 0; iy < bh; iy += subsampling)
                for (int ix = 0; ix < bw; ix++) {
                    if (dst8) dst8[(bi << (bsx + bsy)) + (iy << bsx) + ix] = (uint8_t)s[iy * BSTRIDE + ix];
                    else dst16[(bi << (bsx + bsy)) + (iy << bsx) + ix] = s[iy * BSTRIDE + ix];
                }
        }
        return;
    }
    if (pli == 0) {
        if (!dirinit || !*dirinit) {
            for (int bi = 0; bi < cdef_count; bi++) {
                const int by = dlist[2 * bi], bx = dlist[2 * bi + 1];
                dir[by][bx]  = oracle_cdef_find_dir(in + 8 * by * BSTRIDE + 8 * bx, BSTRIDE, &var[by][bx], coeff_shift);
            }
            if (dirinit) *dirinit = 1;
        }
    } else if (pli == 1 && xdec != ydec) {
        static const uint8_t conv422[8] = {7, 0, 2, 4, 5, 6, 6, 6}, conv440[8] = {1, 2, 2, 2, 3, 4, 6, 0};
        for (int bi = 0; bi < cdef_count; bi++) {
            const int by = dlist[2 * bi], bx = dlist[2 * bi + 1];
            dir[by][bx]  = (xdec ? conv422 : conv440)[dir[by][bx]];
        }
    }
    for (int bi = 0; bi < cdef_count; bi++) {
        const int by = dlist[2 * bi], bx = dlist[2 * bi + 1];
        const int t  = pli ? pri_strength : adjust_strength(pri_strength, var[by][bx]);
        const int o  = dstride ? (by << bsy) * dstride + (bx << bsx) : bi << (bsx + bsy);
        oracle_cdef_filter_block(dst8 ? dst8 + o : NULL, dst8 ? NULL : dst16 + o, dstride ? dstride : bw, in + (by << bsy) * BSTRIDE + (bx << bsx),
                                 t, sec_strength, pri_strength ? dir[by][bx] : 0, pri_damping, sec_damping, bw, bh, coeff_shift, subsampling);
    }
}

/* svt_aom_compute_cdef_dist_c / _8bit_c, enc_cdef.c:23-219.  `plane` = the ORIGINAL picture (strided), `packed` = the
 * filtered blocks in dlist order.  Luma 8x8 uses the variance-weighted distortion in double precision. */
uint64_t oracle_cdef_dist(const void *plane, int pstride, const void *packed, const uint8_t *dlist, int cdef_count, int bw, int bh,
                          int coeff_shift, int pli, int subsampling, int is16) {
    uint64_t sum = 0;
    for (int bi = 0; bi < cdef_count; bi++) {
        const int by = dlist[2 * bi], bx = dlist[2 * bi + 1];
        uint64_t  ss = 0, sd = 0, ss2 = 0, sd2 = 0, ssd = 0, mse = 0;
        for (int i = 0; i < bh; i += subsampling)
            for (int j = 0; j < bw; j++) {
                const int      po = (by * bh + i) * pstride + bx * bw + j, qo = bi * bw * bh + bw * i + j;
                const uint32_t d  = is16 ? ((const uint16_t *)plane)[po] : ((const uint8_t *)plane)[po];
                const uint32_t s  = is16 ? ((const uint16_t *)packed)[qo] : ((const uint8_t *)packed)[qo];
                ss += s; sd += d; ss2 += s * s; sd2 += d * d; ssd += s * d;
                const int32_t e = (int32_t)d - (int32_t)s;
                mse += (uint64_t)(int64_t)(e * e);
            }
        if (pli == 0 && bw == 8 && bh == 8) {
            const uint64_t svar = ss2 - ((ss * ss + 32) >> 6), dvar = sd2 - ((sd * sd + 32) >> 6);
            sum += (uint64_t)floor(.5 + (sd2 + ss2 - 2 * ssd) * .5 * (svar + dvar + (400 << 2 * coeff_shift)) /
                                            (sqrt((20000 << 4 * coeff_shift) + svar * (double)dvar)));
        } else {
            sum += mse;
        }
    }
    return sum >> 2 * coeff_shift;
}

/* Frame-level drivers = what the batched HIP entry points compute.  Tile construction follows cdef_seg_search
 * (cdef_process.c:208-228): the 64x64 area (clipped to the plane) plus a 3-row / 8-column halo taken from the
 * neighbouring filter blocks where they exist; everything else CDEF_VERY_LARGE. */
static void build_tile(uint16_t *tile /* (64+2*VBORDER) * BSTRIDE */, const void *plane, int stride, int pw, int ph, int fbr, int fbc, int nvfb,
                       int nhfb, int xdec, int ydec, int is16) {
    for (int i = 0; i < (64 + 2 * VBORDER) * BSTRIDE; i++) tile[i] = VERY_LARGE;
    const int bw = 64 >> xdec, bh = 64 >> ydec;
    const int x0 = fbc * bw, y0 = fbr * bh;
    const int xs = x0 - (fbc != 0 ? HBORDER : 0), ys = y0 - (fbr != 0 ? VBORDER : 0);
    int       xe = x0 + bw, ye = y0 + bh;
    if (xe > pw) xe = pw;
    if (ye > ph) ye = ph;
    if (fbc + 1 < nhfb) xe += HBORDER;
    if (fbr + 1 < nvfb) ye += VBORDER;
    /* (the reference copies the full halo out of a padded frame buffer; a 4:2:0 chroma plane's last filter block can be 4 samples wide, and the 4 halo columns beyond
     * the picture are never reached by a tap -- 2 samples at most -- so a checker that is handed tightly allocated planes leaves them VERY_LARGE instead of reading
     * past a row's end: AddressSanitizer, profiles/r05_asan_emulator.txt) */
    if (xe > pw) xe = pw;
    if (ye > ph) ye = ph;
    uint16_t *in = tile + VBORDER * BSTRIDE + HBORDER;
    for (int y = ys; y < ye; y++)
        for (int x = xs; x < xe; x++)
            in[(y - y0) * BSTRIDE + (x - x0)] = is16 ? ((const uint16_t *)plane)[y * stride + x] : ((const uint8_t *)plane)[y * stride + x];
}

/* One plane of one frame.  skip[(fbr*8+by)*(nhfb*8) + fbc*8+bx] != 0 marks a skipped 8x8 luma unit.
 * mode 0 (apply):  out plane <- filtered pixels (skipped units and everything else copied from `recon`), strengths per fb.
 * mode 1 (search): mse[fb * ncand + c] for candidate (pri[c], sec[c]); `source` is the original picture.
 * dir/var: [nfb][64] (by*8+bx), written for pli == 0, read for pli > 0. */
void oracle_cdef_frame(int mode, const void *recon, int rstride, const void *source, int sstride, void *out, int ostride, int pw, int ph, int xdec,
                       int ydec, int pli, int is16, int coeff_shift, int pri_damping, int sec_damping, int subsampling, const uint8_t *skip,
                       const int32_t *pri, const int32_t *sec, int ncand, uint8_t *dir_all, int32_t *var_all, uint64_t *mse) {
    const int bw = 64 >> xdec, bh = 64 >> ydec;
    const int nhfb = (pw + bw - 1) / bw, nvfb = (ph + bh - 1) / bh;
    const int px = is16 ? 2 : 1;
    uint16_t *tile = (uint16_t *)malloc(sizeof(uint16_t) * (64 + 2 * VBORDER) * BSTRIDE);
    uint16_t *tmp  = (uint16_t *)malloc(sizeof(uint16_t) * 64 * 64);
    if (mode == 0)
        for (int y = 0; y < ph; y++) memcpy((uint8_t *)out + (size_t)y * ostride * px, (const uint8_t *)recon + (size_t)y * rstride * px, (size_t)pw * px);
    for (int fbr = 0; fbr < nvfb; fbr++)
        for (int fbc = 0; fbc < nhfb; fbc++) {
            const int fb = fbr * nhfb + fbc;
            uint8_t   dlist[128];
            int       cnt = 0;
            const int u8w = 8 >> xdec, u8h = 8 >> ydec; /* size of one unit in this plane */
            for (int by = 0; by < 8; by++)
                for (int bx = 0; bx < 8; bx++) {
                    if ((fbc * 8 + bx) * u8w >= pw || (fbr * 8 + by) * u8h >= ph) continue;
                    if (skip[(fbr * 8 + by) * (nhfb * 8) + fbc * 8 + bx]) continue;
                    dlist[2 * cnt] = (uint8_t)by; dlist[2 * cnt + 1] = (uint8_t)bx; cnt++;
                }
            if (mode == 1)
                for (int c = 0; c < ncand; c++) mse[(size_t)fb * ncand + c] = 0;
            if (!cnt) continue;
            build_tile(tile, recon, rstride, pw, ph, fbr, fbc, nvfb, nhfb, xdec, ydec, is16);
            uint16_t *in = tile + VBORDER * BSTRIDE + HBORDER;
            uint8_t   dir[16][16];
            int32_t   var[16][16];
            memset(dir, 0, sizeof(dir));
            memset(var, 0, sizeof(var));
            if (pli)
                for (int b = 0; b < 64; b++) { dir[b >> 3][b & 7] = dir_all[fb * 64 + b]; var[b >> 3][b & 7] = var_all[fb * 64 + b]; }
            int dirinit = 0;
            /* Chroma of non-4:2:0 formats: svt_cdef_filter_fb remaps the luma directions IN PLACE on every pli == 1 call (cdef.c:388-395), which is
             * right for the one call per plane of the apply path (plane 2 then sees plane 1's remapped array) but would remap again per candidate
             * in a strength search.  SVT-AV1 only encodes 4:2:0 (xdec == ydec), where the remap never runs; the frame drivers here define the
             * non-4:2:0 case as "every chroma call sees the luma directions remapped exactly once" and therefore call the block driver with
             * pli = 2 (no in-place remap) on a freshly remapped copy. */
            const int pli_call = pli ? 2 : 0;
            if (pli && xdec != ydec) {
                static const uint8_t conv422[8] = {7, 0, 2, 4, 5, 6, 6, 6}, conv440[8] = {1, 2, 2, 2, 3, 4, 6, 0};
                for (int b = 0; b < 64; b++) dir[b >> 3][b & 7] = (xdec ? conv422 : conv440)[dir[b >> 3][b & 7] & 7];
            }
            for (int c = 0; c < (mode == 1 ? ncand : 1); c++) {
                const int lvl = mode == 1 ? pri[c] : pri[fb], sc = mode == 1 ? sec[c] : sec[fb];
                if (mode == 0) {
                    if (lvl == 0 && sc == 0) { /* zero strength leaves the pixels unchanged (enc_cdef.c:571: the call only initialises dir) */
                    } else {
                        uint8_t  *o8  = is16 ? NULL : (uint8_t *)out + (size_t)(fbr * bh) * ostride + fbc * bw;
                        uint16_t *o16 = is16 ? (uint16_t *)out + (size_t)(fbr * bh) * ostride + fbc * bw : NULL;
                        oracle_cdef_filter_fb(o8, o16, ostride, in, xdec, ydec, dir, &dirinit, var, pli_call, dlist, cnt, lvl, sc, pri_damping, sec_damping,
                                              coeff_shift, 1);
                    }
                } else {
                    /* the first non-zero candidate initialises dir/var; a zero-strength candidate does not need them */
                    oracle_cdef_filter_fb(is16 ? NULL : (uint8_t *)tmp, is16 ? tmp : NULL, 0, in, xdec, ydec, dir, &dirinit, var, pli_call, dlist, cnt, lvl, sc,
                                          pri_damping, sec_damping, coeff_shift, subsampling);
                    const uint8_t *sp = (const uint8_t *)source + ((size_t)(fbr * bh) * sstride + fbc * bw) * px;
                    mse[(size_t)fb * ncand + c] = oracle_cdef_dist(sp, sstride, tmp, dlist, cnt, 8 >> xdec, 8 >> ydec, coeff_shift, pli, subsampling, is16);
                }
            }
            if (pli == 0) {
                if (!dirinit) /* every candidate had zero strength: directions are still defined as find_dir of the tile */
                    for (int bi = 0; bi < cnt; bi++)
                        dir[dlist[2 * bi]][dlist[2 * bi + 1]] = oracle_cdef_find_dir(in + 8 * dlist[2 * bi] * BSTRIDE + 8 * dlist[2 * bi + 1], BSTRIDE,
                                                                                     &var[dlist[2 * bi]][dlist[2 * bi + 1]], coeff_shift);
                for (int b = 0; b < 64; b++) { dir_all[fb * 64 + b] = dir[b >> 3][b & 7]; var_all[fb * 64 + b] = var[b >> 3][b & 7]; }
            }
        }
    free(tile);
    free(tmp);
}

/* svt_search_one_dual_c (enc_cdef.c:627-683) over flat [sb_count][64] tables */
uint64_t oracle_search_one_dual(int *lev0, int *lev1, int nb_strengths, const uint64_t *mse0, const uint64_t *mse1, int sb_count, int start_gi, int end_gi) {
    static uint64_t tot[64][64];
    uint64_t        best_tot = (uint64_t)1 << 63;
    int             best0 = 0, best1 = 0;
    memset(tot, 0, sizeof(tot));
    for (int i = 0; i < sb_count; i++) {
        uint64_t best = (uint64_t)1 << 63;
        for (int gi = 0; gi < nb_strengths; gi++) {
            const uint64_t c = mse0[(size_t)i * 64 + lev0[gi]] + mse1[(size_t)i * 64 + lev1[gi]];
            if (c < best) best = c;
        }
        for (int j = start_gi; j < end_gi; j++)
            for (int k = start_gi; k < end_gi; k++) {
                const uint64_t c = mse0[(size_t)i * 64 + j] + mse1[(size_t)i * 64 + k];
                tot[j][k] += c < best ? c : best;
            }
    }
    for (int j = start_gi; j < end_gi; j++)
        for (int k = start_gi; k < end_gi; k++)
            if (tot[j][k] < best_tot) { best_tot = tot[j][k]; best0 = j; best1 = k; }
    lev0[nb_strengths] = best0;
    lev1[nb_strengths] = best1;
    return best_tot;
}

/* joint_strength_search_dual (enc_cdef.c:697-726) */
uint64_t oracle_joint_strength_search(int *lev0, int *lev1, int nb_strengths, const uint64_t *mse0, const uint64_t *mse1, int sb_count, int start_gi, int end_gi) {
    uint64_t best = (uint64_t)1 << 63;
    for (int i = 0; i < nb_strengths; i++) best = oracle_search_one_dual(lev0, lev1, i, mse0, mse1, sb_count, start_gi, end_gi);
    for (int i = 0; i < 4 * nb_strengths; i++) {
        for (int j = 0; j < nb_strengths - 1; j++) { lev0[j] = lev0[j + 1]; lev1[j] = lev1[j + 1]; }
        best = oracle_search_one_dual(lev0, lev1, nb_strengths - 1, mse0, mse1, sb_count, start_gi, end_gi);
    }
    return best;
}
/* finish_cdef_search's per-filter-block assignment (enc_cdef.c:916-931) */
void oracle_assign_fb_strengths(const uint64_t *mse0, const uint64_t *mse1, const int *lev0, const int *lev1, int nb_strengths, int sb_count, int8_t *best_gi) {
    for (int i = 0; i < sb_count; i++) {
        uint64_t best = (uint64_t)1 << 63;
        int      arg  = 0;
        for (int gi = 0; gi < nb_strengths; gi++) {
            const uint64_t c = mse0[(size_t)i * 64 + lev0[gi]] + mse1[(size_t)i * 64 + lev1[gi]];
            if (c < best) { best = c; arg = gi; }
        }
        best_gi[i] = (int8_t)arg;
    }
}
