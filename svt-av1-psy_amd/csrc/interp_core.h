// interp_core.h -- AV1 sub-pel interpolation of one block by one wave, shared by the temporal filter's sub-pel refinement (tf_subpel.hip) and the TPL dispenser's
// sub-pel vectors (tpl_full.hip): the kernels' taps packed for the dot instructions, the four cases of svt_av1_[highbd_]convolve_{2d_copy,x,y,2d}_sr_c with the
// reference's roundings (inter_prediction.c:311-420, 737-790), wave reductions.  Included inside each translation unit's anonymous namespace users (everything here
// is internal).
#pragma once
#include "svt_hip_common.h"

namespace {

// av1 sub_pel_filters_8 (EIGHTTAP_REGULAR), sub_pel_filters_4 (block dimension <= 4: inter_prediction.h:147-153) -- inter_prediction.c:205-254; bilinear =
// {128 - 8p, 8p} at taps 3, 4.  Phase 0 (the tap 128) is never evaluated: a zero phase selects the copy / one-directional kernels instead.
constexpr int8_t kReg8[16][8] = {{0, 0, 0, 0, 0, 0, 0, 0},        {0, 2, -6, 126, 8, -2, 0, 0},    {0, 2, -10, 122, 18, -4, 0, 0},  {0, 2, -12, 116, 28, -8, 2, 0},
                                 {0, 2, -14, 110, 38, -10, 2, 0}, {0, 2, -14, 102, 48, -12, 2, 0}, {0, 2, -16, 94, 58, -12, 2, 0},  {0, 2, -14, 84, 66, -14, 2, 0},
                                 {0, 2, -14, 76, 76, -14, 2, 0},  {0, 2, -14, 66, 84, -14, 2, 0},  {0, 2, -12, 58, 94, -16, 2, 0},  {0, 2, -12, 48, 102, -14, 2, 0},
                                 {0, 2, -10, 38, 110, -14, 2, 0}, {0, 2, -8, 28, 116, -12, 2, 0},  {0, 0, -4, 18, 122, -10, 2, 0},  {0, 0, -2, 8, 126, -6, 2, 0}};
constexpr int8_t kReg4[16][8] = {{0, 0, 0, 0, 0, 0, 0, 0},       {0, 0, -4, 126, 8, -2, 0, 0},    {0, 0, -8, 122, 18, -4, 0, 0},   {0, 0, -10, 116, 28, -6, 0, 0},
                                 {0, 0, -12, 110, 38, -8, 0, 0}, {0, 0, -12, 102, 48, -10, 0, 0}, {0, 0, -14, 94, 58, -10, 0, 0},  {0, 0, -12, 84, 66, -10, 0, 0},
                                 {0, 0, -12, 76, 76, -12, 0, 0}, {0, 0, -10, 66, 84, -12, 0, 0},  {0, 0, -10, 58, 94, -14, 0, 0},  {0, 0, -10, 48, 102, -12, 0, 0},
                                 {0, 0, -8, 38, 110, -12, 0, 0}, {0, 0, -6, 28, 116, -10, 0, 0},  {0, 0, -4, 18, 122, -8, 0, 0},   {0, 0, -2, 8, 126, -4, 0, 0}};
constexpr int8_t kSharp8[16][8] = {{0, 0, 0, 0, 0, 0, 0, 0},           {-2, 2, -6, 126, 8, -2, 2, 0},      {-2, 6, -12, 124, 16, -6, 4, -2},  {-2, 8, -18, 120, 26, -10, 6, -2},
                                   {-4, 10, -22, 116, 38, -14, 6, -2}, {-4, 10, -22, 108, 48, -18, 8, -2}, {-4, 10, -24, 100, 60, -20, 8, -2}, {-4, 10, -24, 90, 70, -22, 10, -2},
                                   {-4, 12, -24, 80, 80, -24, 12, -4}, {-2, 10, -22, 70, 90, -24, 10, -4}, {-2, 8, -20, 60, 100, -24, 10, -4}, {-2, 8, -18, 48, 108, -22, 10, -4},
                                   {-2, 6, -14, 38, 116, -22, 10, -4}, {-2, 6, -10, 26, 120, -18, 8, -2},  {-2, 4, -6, 16, 124, -12, 6, -2},   {0, 2, -2, 8, 126, -6, 2, -2}}; // MULTITAP_SHARP
// the taps of one (kernel, phase), packed for the dot instructions: b4 = four signed bytes per dword, h2 = two signed halves per dword; for the bilinear
// kernel only taps 3 and 4 are stored (b4[0] bytes 0-1, h2[0])
struct PackedTaps { uint32_t b4[2], h2[4]; };
struct TapTables { PackedTaps t[4][16]; }; // [regular 8 | regular 4 | bilinear | sharp 8][phase]
constexpr TapTables make_tap_tables() {
    TapTables T{};
    for (int kind = 0; kind < 4; kind++) {
        if (kind == 2) continue;
        for (int p = 0; p < 16; p++) {
            for (int k = 0; k < 8; k++) {
                const int f = kind == 0 ? kReg8[p][k] : (kind == 1 ? kReg4[p][k] : kSharp8[p][k]);
                T.t[kind][p].b4[k >> 2] |= (uint32_t)(f & 0xff) << (8 * (k & 3));
                T.t[kind][p].h2[k >> 1] |= (uint32_t)(f & 0xffff) << (16 * (k & 1));
            }
        }
    }
    for (int p = 1; p < 16; p++) {
        T.t[2][p].b4[0] = (uint32_t)(128 - 8 * p) | ((uint32_t)(8 * p) << 8);
        T.t[2][p].h2[0] = (uint32_t)(128 - 8 * p) | ((uint32_t)(8 * p) << 16);
    }
    return T;
}
__device__ constexpr TapTables kTaps = make_tap_tables();
__device__ __forceinline__ int tap_of(const PackedTaps& t, const int k) { return (int)(int16_t)(t.h2[k >> 1] >> (16 * (k & 1))); } // tap T0 + k

__device__ __forceinline__ int rpot(const int v, const int n) { return n ? (v + (1 << (n - 1))) >> n : v; }

// sum over each row of 16 lanes (4 DPP steps), then the four row sums through SGPRs: no LDS round trips on the per-candidate critical path
__device__ __forceinline__ uint32_t row16_sum(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false);  // quad_perm [1,0,3,2]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false);  // quad_perm [2,3,0,1]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xf, 0xf, false); // row_ror:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false); // row_ror:8
    return v;
}
__device__ __forceinline__ int wave_sum_i32(const int v) {
    const int r = (int)row16_sum((uint32_t)v);
    return __builtin_amdgcn_readlane(r, 0) + __builtin_amdgcn_readlane(r, 16) + __builtin_amdgcn_readlane(r, 32) + __builtin_amdgcn_readlane(r, 48);
}
__device__ __forceinline__ unsigned long long wave_sum_u32_wide(const uint32_t v) { // each row's sum must fit 32 bits, the total need not
    const int r = (int)row16_sum(v);
    return (unsigned long long)(uint32_t)__builtin_amdgcn_readlane(r, 0) + (uint32_t)__builtin_amdgcn_readlane(r, 16) + (uint32_t)__builtin_amdgcn_readlane(r, 32) +
           (uint32_t)__builtin_amdgcn_readlane(r, 48);
}

__device__ __forceinline__ int sdot2(const uint32_t a, const uint32_t b, const int c) { // c + a.lo * b.lo + a.hi * b.hi, signed 16 bit (v_dot2_i32_i16)
    typedef short s2 __attribute__((ext_vector_type(2)));
    s2 x, y;
    __builtin_memcpy(&x, &a, 4);
    __builtin_memcpy(&y, &b, 4);
    return __builtin_amdgcn_sdot2(x, y, c, false);
}
struct __attribute__((packed, aligned(1))) u16_a1 { uint16_t x; };
struct __attribute__((packed, aligned(1))) u32_a1 { uint32_t x; };
struct __attribute__((packed, aligned(1))) u32x2_a1 { uint32_t x, y; };
struct __attribute__((packed, aligned(1))) u32x4_a1 { uint32_t x, y, z, w; };

// sum over the NT taps of tap * sample for the samples starting at q (= the first evaluated tap's sample), one unaligned load:
// 8 bit: v_dot4_i32_i8 on (sample - 128) -- every AV1 kernel's taps sum to 128, so the bias is the constant 128 * 128; 10 bit: v_dot2_i32_i16
template <int NT> __device__ __forceinline__ int hsum(const uint8_t* q, const PackedTaps& t) {
    if (NT == 8) {
        const u32x2_a1 v = *reinterpret_cast<const u32x2_a1*>(q);
        return __builtin_amdgcn_sdot4((int)(v.x ^ 0x80808080u), (int)t.b4[0], __builtin_amdgcn_sdot4((int)(v.y ^ 0x80808080u), (int)t.b4[1], 16384, false), false);
    }
    const uint32_t v = reinterpret_cast<const u16_a1*>(q)->x;
    return __builtin_amdgcn_sdot4((int)(v ^ 0x8080u), (int)t.b4[0], 16384, false);
}
template <int NT> __device__ __forceinline__ int hsum(const uint16_t* q, const PackedTaps& t) {
    if (NT == 8) {
        const u32x4_a1 v = *reinterpret_cast<const u32x4_a1*>(q);
        return sdot2(v.x, t.h2[0], sdot2(v.y, t.h2[1], sdot2(v.z, t.h2[2], sdot2(v.w, t.h2[3], 0))));
    }
    return sdot2(reinterpret_cast<const u32_a1*>(q)->x, t.h2[0], 0);
}

// The W x nrows samples of one prediction (every rstep-th row of the block at p0, row stride rsm) -> finish(i, unclamped sample), i = row * W + column, split over the
// wave's lanes.  NT = 8: an 8-tap kernel (taps 0..7 around x - 3); NT = 2: bilinear (taps 3, 4).  The four cases are the four kernels of
// svt_av1_[highbd_]convolve_{2d_copy,x,y,2d}_sr_c; the choice is wave-uniform, offsets are unsigned from a per-prediction base.
template <typename PIX, int NT, typename F>
__device__ __forceinline__ void predict_rows(const PIX* __restrict__ p0, const long rsm, const int W, const int nrows, const int rstep, const int sx, const int sy,
                                             const PackedTaps& tx, const PackedTaps& ty, const int bd, uint32_t* __restrict__ imp, const int l, F finish) {
    constexpr bool HBD = sizeof(PIX) == 2;
    constexpr int  T0  = NT == 8 ? 0 : 3; // first tap evaluated
    int r0 = 3, r1 = 11;                  // get_conv_params_no_round (convolve.h:40-64)
    if (bd + 7 - r0 + 2 > 16) { r1 -= bd + 7 - r0 + 2 - 16; r0 += bd + 7 - r0 + 2 - 16; }
    const int      lw = 31 - __clz(W), offset_bits = bd + 14 - r0, npx = nrows * W;
    const uint32_t urs = (uint32_t)rsm;
    if (!sx && !sy) {
        for (int i = l; i < npx; i += 64) finish(i, (int)p0[(uint32_t)((i >> lw) * rstep) * urs + (uint32_t)(i & (W - 1))]);
    } else if (!sy) {
        const PIX* pb = p0 - 3 + T0;
        for (int i = l; i < npx; i += 64) {
            const uint32_t o = (uint32_t)((i >> lw) * rstep) * urs + (uint32_t)(i & (W - 1));
            finish(i, rpot(rpot(hsum<NT>(pb + o, tx), r0), 7 - r0));
        }
    } else if (!sx) {
        const PIX* pb = p0 - (long)(3 - T0) * rsm;
        for (int i = l; i < npx; i += 64) {
            const uint32_t o = (uint32_t)((i >> lw) * rstep) * urs + (uint32_t)(i & (W - 1));
            int            s = 0;
#pragma unroll
            for (int k = 0; k < NT; k++) s += tap_of(ty, k) * (int)(pb + o)[(long)k * rsm];
            finish(i, rpot(s, 7));
        }
    } else {
        // horizontal pass of every row a needed output row touches -> im, rounded like the reference's im_block (int16); im row y <-> prediction row
        // y - 3 + T0.  A lane filters rows 2m and 2m + 1 at one x and stores them as ONE dword, so that the vertical pass is v_dot2_i32_i16 on row pairs.
        const PIX* pb     = p0 - (long)(3 - T0) * rsm - 3 + T0;
        const int  imrows = (nrows - 1) * rstep + NT, npairs = (imrows + 1) >> 1;
        for (int i = l; i < (npairs << lw); i += 64) {
            const int      m = i >> lw;
            const uint32_t o = (uint32_t)(2 * m) * urs + (uint32_t)(i & (W - 1));
            const int      a = rpot(hsum<NT>(pb + o, tx) + (1 << (bd + 6)), r0);
            const int      b = 2 * m + 1 < imrows ? rpot(hsum<NT>(pb + o + rsm, tx) + (1 << (bd + 6)), r0) : 0; // (never read; keeps the loads inside the rows the reference reads)
            imp[i] = (uint32_t)(a & 0xffff) | ((uint32_t)b << 16);
        }
        __builtin_amdgcn_wave_barrier();
        const int sub = (1 << (offset_bits - r1)) + (1 << (offset_bits - r1 - 1));
        for (int i = l; i < npx; i += 64) {
            const int       y = (i >> lw) * rstep;
            const uint32_t* c = imp + ((y >> 1) << lw) + (i & (W - 1));
            const uint32_t  sh = (uint32_t)(y & 1) << 4; // odd rows: the pair (y, y + 1) straddles two stored pairs
            int             s = 1 << offset_bits;
            uint32_t        lo = c[0];
#pragma unroll
            for (int j = 0; j < NT / 2; j++) {
                const uint32_t hi = c[(j + 1) << lw];
                s  = sdot2(__builtin_amdgcn_alignbit(hi, lo, sh), ty.h2[j], s);
                lo = hi;
            }
            int res = rpot(s, r1) - sub;
            if (!HBD) res = (int16_t)res; // the 8-bit kernel narrows to ConvBufType first (inter_prediction.c:345)
            finish(i, rpot(res, 14 - r0 - r1));
        }
        __builtin_amdgcn_wave_barrier(); // the slice is rewritten by the next candidate
    }
}

} // namespace
