// cdef.hip -- CDEF direction search, filter and strength-search distortion for gfx950.
//
// Frame kernel: one workgroup per (64x64 filter block, chunk of strength candidates).  The block and its 3-row /
// 8-column halo are staged once into LDS as u16 (CDEF_VERY_LARGE outside the frame, cdef_process.c:208-228); a quad of
// lanes owns one 8x8 unit: the quad finds the unit's direction (each lane evaluates two orthogonal directions with
// fully static partial-sum indices, so `var = best - cost[orthogonal]` is lane-local), then every lane filters two
// rows.  Search mode never materialises filtered pixels: the variance-weighted luma distortion / chroma MSE
// (enc_cdef.c:23-219) is accumulated from registers, so HBM traffic is one read of the block per candidate chunk
// and 8 bytes out per (block, strength).  Apply mode writes out of place (the reference's in-place line/column buffers,
// enc_cdef.c:334-335, exist only because it overwrites its input).
#include "svt_hip_common.h"
#include "../../include/svtav1_hip.h"

namespace {

constexpr int VERY_LARGE = 0x7f7f; // CDEF_VERY_LARGE (cdef.h:38)
constexpr int VB = 3, HB = 8;      // CDEF_VBORDER / CDEF_HBORDER
constexpr int CAND_CHUNK = 16;

__device__ __forceinline__ int msb_u32(uint32_t n) { return 31 - __clz(n); }

// constrain() of cdef.c:85-91 with the damping shift hoisted: shift = max(0, damping - msb(threshold))
__device__ __forceinline__ int constrain_s(const int diff, const int threshold, const int shift) {
    const int ad = diff < 0 ? -diff : diff;
    int       v  = threshold - (ad >> shift);
    v            = v < 0 ? 0 : v;
    v            = ad < v ? ad : v;
    return diff < 0 ? -v : v;
}
// Cdef_Directions (cdef.c:99-120) as (dy, dx) of tap k
__device__ __forceinline__ int dir_off(const int dir, const int k, const int pitch) {
    const int d  = dir & 7;
    const int dy = k == 0 ? (d == 0 ? -1 : (d >= 4 ? 1 : 0)) : (d == 0 ? -2 : (d == 1 ? -1 : (d == 2 ? 0 : (d == 3 ? 1 : 2))));
    const int dx = k == 0 ? (d <= 4 ? 1 : 0) : (d <= 4 ? 2 : (d == 5 ? 1 : (d == 6 ? 0 : -1)));
    return dy * pitch + dx;
}

struct FilterCtx {
    int pri, sec, pri_shift, sec_shift, pt0, pt1, st0, st1;
    int po[2], s0o[2], s1o[2];
};
__device__ __forceinline__ FilterCtx make_ctx(const int pri_strength, const int sec_strength, const int dir, const int pri_damping, const int sec_damping,
                                              const int coeff_shift, const int pitch) {
    FilterCtx c;
    c.pri = pri_strength;
    c.sec = sec_strength;
    int s = pri_strength ? pri_damping - msb_u32((uint32_t)pri_strength) : 0;
    c.pri_shift = s < 0 ? 0 : s;
    s           = sec_strength ? sec_damping - msb_u32((uint32_t)sec_strength) : 0;
    c.sec_shift = s < 0 ? 0 : s;
    const int odd = (pri_strength >> coeff_shift) & 1; // svt_aom_eb_cdef_pri_taps / sec_taps (cdef.c:249-250)
    c.pt0 = odd ? 3 : 4; c.pt1 = odd ? 3 : 2; c.st0 = 2; c.st1 = 1;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        c.po[k]  = dir_off(dir, k, pitch);
        c.s0o[k] = dir_off(dir + 2, k, pitch);
        c.s1o[k] = dir_off(dir + 6, k, pitch);
    }
    return c;
}
// one pixel of svt_cdef_filter_block_c (cdef.c:253-306); `p` points at the pixel inside a u16 tile
__device__ __forceinline__ int filter_px(const uint16_t* p, const FilterCtx& c) {
    const int x = (int16_t)p[0];
    int sum = 0, mx = x, mn = x;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int pt = k ? c.pt1 : c.pt0, st = k ? c.st1 : c.st0;
        const int v[6] = {(int16_t)p[c.po[k]], (int16_t)p[-c.po[k]], (int16_t)p[c.s0o[k]], (int16_t)p[-c.s0o[k]], (int16_t)p[c.s1o[k]], (int16_t)p[-c.s1o[k]]};
#pragma unroll
        for (int t = 0; t < 6; t++) {
            if (t < 2) { if (c.pri) sum += pt * constrain_s(v[t] - x, c.pri, c.pri_shift); }
            else       { if (c.sec) sum += st * constrain_s(v[t] - x, c.sec, c.sec_shift); }
            if (v[t] != VERY_LARGE) mx = v[t] > mx ? v[t] : mx;
            mn = v[t] < mn ? v[t] : mn;
        }
    }
    sum   = (int16_t)sum; // the reference accumulates in int16 (cdef.c:265); |sum| <= 12 * 4 * 240 so this never truncates
    int y = x + ((8 + sum - (sum < 0)) >> 4);
    return y < mn ? mn : (y > mx ? mx : y);
}
__device__ __forceinline__ int adjust_strength(const int strength, const int var) { // cdef.c:130-134
    const int v6 = var >> 6;
    int       i  = v6 ? msb_u32((uint32_t)v6) : 0;
    i            = i > 12 ? 12 : i;
    return var ? (strength * (4 + i) + 8) >> 4 : 0;
}

// cost of direction D for one 8x8 unit (svt_aom_cdef_find_dir_c, cdef.c:150-199); all partial-sum indices are static
template <int D> __device__ __forceinline__ int dir_cost(const uint16_t* img, const int pitch, const int coeff_shift) {
    int partial[15];
#pragma unroll
    for (int i = 0; i < 15; i++) partial[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int x = ((int)img[i * pitch + j] >> coeff_shift) - 128;
            constexpr int dummy = 0;
            (void)dummy;
            const int idx = D == 0 ? i + j : D == 1 ? i + j / 2 : D == 2 ? i : D == 3 ? 3 + i - j / 2 : D == 4 ? 7 + i - j : D == 5 ? 3 - i / 2 + j : D == 6 ? j : i / 2 + j;
            partial[idx] += x;
        }
    // (unsigned arithmetic: the lanes of units outside the picture or skipped run the same code on whatever their tile holds -- OUTSIDE markers included --, and their
    // results are dropped; a real unit's cost is at most 8 x 1 024^2 x 105 < 2^31 as in the reference)
    auto sq = [](const int p) { return (uint32_t)p * (uint32_t)p; };
    uint32_t cost = 0;
    if (D == 2 || D == 6) {
#pragma unroll
        for (int i = 0; i < 8; i++) cost += sq(partial[i]);
        cost *= 105u;
    } else if (D == 0 || D == 4) {
        constexpr uint32_t div[8] = {840, 420, 280, 210, 168, 140, 120, 105};
#pragma unroll
        for (int i = 0; i < 7; i++) cost += (sq(partial[i]) + sq(partial[14 - i])) * div[i];
        cost += sq(partial[7]) * 105u;
    } else {
#pragma unroll
        for (int j = 0; j < 5; j++) cost += sq(partial[3 + j]);
        cost *= 105u;
        constexpr uint32_t div2[3] = {420, 210, 140};
#pragma unroll
        for (int j = 0; j < 3; j++) cost += (sq(partial[j]) + sq(partial[10 - j])) * div2[j];
    }
    return (int)cost;
}
// lane q of a quad evaluates directions q and q+4; returns best dir / var in every lane of the quad
__device__ __forceinline__ void quad_find_dir(const uint16_t* img, const int pitch, const int coeff_shift, const int q, int& best_dir, int& var) {
    int a, b;
    if (q == 0) { a = dir_cost<0>(img, pitch, coeff_shift); b = dir_cost<4>(img, pitch, coeff_shift); }
    else if (q == 1) { a = dir_cost<1>(img, pitch, coeff_shift); b = dir_cost<5>(img, pitch, coeff_shift); }
    else if (q == 2) { a = dir_cost<2>(img, pitch, coeff_shift); b = dir_cost<6>(img, pitch, coeff_shift); }
    else { a = dir_cost<3>(img, pitch, coeff_shift); b = dir_cost<7>(img, pitch, coeff_shift); }
    // first maximum in index order, strict '>' starting from best_cost = 0 (cdef.c:200-205)
    int c = a, d = q, o = b;
    if (b > a) { c = b; d = q + 4; o = a; }
#pragma unroll
    for (int m = 1; m <= 2; m <<= 1) {
        const int c2 = __shfl_xor(c, m), d2 = __shfl_xor(d, m), o2 = __shfl_xor(o, m);
        if (c2 > c || (c2 == c && d2 < d)) { c = c2; d = d2; o = o2; }
    }
    // (costs are sums of squares, so "nothing beat the initial best_cost = 0" means every cost is 0: dir 0, var 0 -- same result)
    best_dir = d;
    var      = (c - o) >> 10;
}

// ---- packed 16-bit arithmetic: two horizontally adjacent pixels per VGPR (v_pk_*_i16 / v_dot2_u32_u16) ----------------------
// Everything in svt_cdef_filter_block_c fits int16: pixels <= 4095, CDEF_VERY_LARGE = 0x7f7f, |sum| <= 12 * 4 * 240, and the reference itself
// accumulates `sum` in int16 (cdef.c:265).
typedef short          s16x2 __attribute__((vector_size(4)));
typedef unsigned short u16x2 __attribute__((vector_size(4)));
typedef unsigned short us2e __attribute__((ext_vector_type(2)));
struct __attribute__((packed, aligned(2))) PairA2 { uint32_t v; };
struct __attribute__((packed, aligned(2))) Row8A2 { uint32_t v[4]; };
struct __attribute__((packed, aligned(1))) Row8A1 { uint32_t v[2]; };
struct __attribute__((aligned(16))) Row8A16 { uint32_t v[4]; };
struct __attribute__((packed, aligned(1))) U16A1 { uint16_t v; };

__device__ __forceinline__ s16x2    as_pk(const uint32_t v) { s16x2 r; __builtin_memcpy(&r, &v, 4); return r; }
__device__ __forceinline__ uint32_t as_u32(const s16x2 v) { uint32_t r; __builtin_memcpy(&r, &v, 4); return r; }
__device__ __forceinline__ s16x2    splat(const int v) { const short h = (short)v; return s16x2{h, h}; }
__device__ __forceinline__ s16x2    pk_max(const s16x2 a, const s16x2 b) { return a > b ? a : b; }
__device__ __forceinline__ s16x2    pk_min(const s16x2 a, const s16x2 b) { return a < b ? a : b; }
struct __attribute__((aligned(4))) DwPairA4 { uint32_t lo, hi; };
// pixel pair at an even pixel offset of the tile: one aligned ds_read_b32
__device__ __forceinline__ s16x2 ld_pair_even(const uint16_t* p) { return as_pk(*(const uint32_t*)p); }
// A pixel pair at an odd pixel offset is fetched as the two dwords around it (one ds_read2_b32) plus a funnel shift (load_taps): a dword-misaligned
// ds_read_b32 is legal but costs ~45 LDS cycles per wave instruction on gfx950 (SQ_LDS_IDX_ACTIVE / SQ_INSTS_LDS, profiles/r01_call6_cdef_lds.txt).
__device__ __forceinline__ uint32_t dot2(const uint32_t a, const uint32_t b, const uint32_t c) { // c + a.lo * b.lo + a.hi * b.hi
    us2e x, y;
    __builtin_memcpy(&x, &a, 4);
    __builtin_memcpy(&y, &b, 4);
    return __builtin_amdgcn_udot2(x, y, c, false);
}
// constrain() of cdef.c:85-91 on a pixel pair = clamp(diff, -m, m) with m = max(0, threshold - (|diff| >> shift)) = sign(diff) * min(|diff|, m).  |diff| and the
// sign multiplier (+1 / -1) of a tap do not depend on the strength: a pass that evaluates several strengths on the same taps (the search: four primary levels, three
// secondary strengths) shares them, and what is left per (tap, strength) is one shift, one saturating v_pk_sub_u16 for the max(0, .), one unsigned minimum and the
// multiply-add that applies the sign while it accumulates (five packed operations plus an add before).  threshold 0 yields 0 for any shift.
__device__ __forceinline__ s16x2 pk_minu(const s16x2 a, const s16x2 b) { const u16x2 x = (u16x2)a, y = (u16x2)b; return (s16x2)(x < y ? x : y); }
struct TapMag { s16x2 ad, sg; };
__device__ __forceinline__ TapMag tap_mag(const s16x2 t, const s16x2 x) {
    // (unsigned subtractions: an OUTSIDE tap is 0x8000, and 0x8000 - x / the negation of -32768 are meant to wrap -- as the packed 16-bit instructions do)
    const s16x2 one = {1, 1}, diff = (s16x2)((u16x2)t - (u16x2)x);
    TapMag m;
    m.ad = pk_max(diff, (s16x2)(u16x2{0, 0} - (u16x2)diff));
    m.sg = (diff >> 15) | one;
    return m;
}
// the same in one piece, for passes that evaluate ONE strength per tap (the apply pass): two packed operations fewer than magnitude + sign there
__device__ __forceinline__ s16x2 constrain2(const s16x2 diff, const s16x2 thr, const s16x2 shift) {
    const s16x2 z  = {0, 0};
    const s16x2 ad = pk_max(diff, (s16x2)(u16x2{0, 0} - (u16x2)diff)); // (wraps for -32768, see tap_mag)
    const s16x2 m  = (s16x2)__builtin_elementwise_sub_sat((u16x2)thr, (u16x2)(ad >> shift));
    return pk_max(pk_min(diff, m), z - m);
}
__device__ __forceinline__ s16x2 constrain_mag(const s16x2 ad, const s16x2 thr, const s16x2 shift) { // min(|diff|, m); |diff| = 0x8000 (diff = -32768) shifts to >= 0xf800: m = 0
    return pk_minu(ad, (s16x2)__builtin_elementwise_sub_sat((u16x2)thr, (u16x2)(ad >> shift)));
}
// Out-of-frame pixels of the frame kernel's tile.  The reference marks them CDEF_VERY_LARGE (0x7f7f): constrain() of such a tap is 0 and the tap
// is left out of the max (cdef.c:277-301) while it can never win the min.  0x8000 has the same three properties at two packed ops per tap
// instead of three: signed max ignores it (-32768), unsigned min ignores it (32768), and |0x8000 - x| >> shift still exceeds every threshold.
constexpr int OUTSIDE = 0x8000;

// Tap k of a lane: byte offset (dword aligned) from the lane's own pixel pair and funnel-shift amount.  The pair sits at an even pixel index, so
// both depend on the direction only and are computed once per unit, not once per pixel.
struct TapOffs { int oa[12]; uint32_t sh[6]; };
__device__ __forceinline__ TapOffs tap_offs(const int dir, const int pitch) {
    // order: primary k = 0 (+,-), k = 1 (+,-); secondary (dir + 2) k = 0 (+,-), (dir + 6) k = 0 (+,-); secondary (dir + 2) k = 1 (+,-), (dir + 6) k = 1 (+,-)
    const int e[6] = {dir_off(dir, 0, pitch), dir_off(dir, 1, pitch), dir_off(dir + 2, 0, pitch), dir_off(dir + 6, 0, pitch), dir_off(dir + 2, 1, pitch),
                      dir_off(dir + 6, 1, pitch)};
    TapOffs o;
#pragma unroll
    for (int k = 0; k < 6; k++) {
        o.oa[2 * k]     = (2 * e[k]) & ~3;  // floor to a dword, also for negative offsets
        o.oa[2 * k + 1] = (-2 * e[k]) & ~3;
        o.sh[k]         = (uint32_t)(2 * e[k]) & 2u; // same parity for +e and -e
    }
    return o;
}
// t[0..3] primary (k0+, k0-, k1+, k1-), t[4..7] secondary k = 0 (weight 2), t[8..11] secondary k = 1 (weight 1); min / max over all twelve and
// the centre.  `pair` = LDS address of the lane's pixel pair (dword aligned); one ds_read2_b32 + one v_alignbyte_b32 per tap.
__device__ __forceinline__ void load_taps(const uint16_t* pair, const TapOffs& o, const s16x2 x, s16x2 (&t)[12], s16x2& mn, s16x2& mx) {
    mn = x;
    mx = x;
#pragma unroll
    for (int k = 0; k < 12; k++) {
        const DwPairA4 v = *(const DwPairA4*)((const char*)pair + o.oa[k]);
        t[k] = as_pk(__builtin_amdgcn_alignbyte(v.hi, v.lo, o.sh[k >> 1]));
        mn   = pk_minu(mn, t[k]);
        mx   = pk_max(mx, t[k]);
    }
}
// taps K0 .. K1 - 1 only, no min / max: the single-strength forms of apply_pass
template <int K0, int K1> __device__ __forceinline__ void load_taps_range(const uint16_t* pair, const TapOffs& o, s16x2 (&t)[12]) {
#pragma unroll
    for (int k = K0; k < K1; k++) {
        const DwPairA4 v = *(const DwPairA4*)((const char*)pair + o.oa[k]);
        t[k] = as_pk(__builtin_amdgcn_alignbyte(v.hi, v.lo, o.sh[k >> 1]));
    }
}
// SHARED: the caller evaluates several strengths on the same taps (magnitude / sign form, shared through common subexpressions)
template <bool SHARED>
__device__ __forceinline__ s16x2 pri_sum(const s16x2 x, const s16x2 (&t)[12], const s16x2 thr, const s16x2 sh, const s16x2 w0, const s16x2 w1) {
    auto d = [&](const int k) { return (s16x2)((u16x2)t[k] - (u16x2)x); }; // (wrapping, see tap_mag)
    if (!SHARED) return w0 * (constrain2(d(0), thr, sh) + constrain2(d(1), thr, sh)) + w1 * (constrain2(d(2), thr, sh) + constrain2(d(3), thr, sh));
    const TapMag m0 = tap_mag(t[0], x), m1 = tap_mag(t[1], x), m2 = tap_mag(t[2], x), m3 = tap_mag(t[3], x);
    const s16x2  k0 = constrain_mag(m0.ad, thr, sh) * m0.sg + constrain_mag(m1.ad, thr, sh) * m1.sg;
    const s16x2  k1 = constrain_mag(m2.ad, thr, sh) * m2.sg + constrain_mag(m3.ad, thr, sh) * m3.sg;
    return w0 * k0 + w1 * k1;
}
template <bool SHARED>
__device__ __forceinline__ s16x2 sec_sum(const s16x2 x, const s16x2 (&t)[12], const s16x2 thr, const s16x2 sh) {
    s16x2 k0 = {0, 0}, k1 = {0, 0};
#pragma unroll
    for (int k = 4; k < 8; k++) {
        if (SHARED) {
            const TapMag a = tap_mag(t[k], x), b = tap_mag(t[k + 4], x);
            k0 += constrain_mag(a.ad, thr, sh) * a.sg;
            k1 += constrain_mag(b.ad, thr, sh) * b.sg;
        } else {
            k0 += constrain2((s16x2)((u16x2)t[k] - (u16x2)x), thr, sh);
            k1 += constrain2((s16x2)((u16x2)t[k + 4] - (u16x2)x), thr, sh);
        }
    }
    return k0 + k0 + k1;
}
__device__ __forceinline__ s16x2 finish_px(const s16x2 x, const s16x2 sum, const s16x2 mn, const s16x2 mx) { // cdef.c:302-303
    const s16x2 y = x + ((sum + splat(8) + (sum >> 15)) >> 4);
    return pk_min(pk_max(y, mn), mx);
}

// Stage rows x (8 * cpr) pixels as u16 into LDS rows of `pitch` pixels: pixels inside [ys, ye) x [xs, xe) come from the plane, the rest is `fill`.
// A thread owns 8-pixel chunks; chunks that lie fully inside are fetched with one vector load each, all issued before the first LDS store.
template <typename PIX, int NIT>
__device__ __forceinline__ void stage_tile(uint16_t* lds, const int pitch, const int rows, const int cpr, const PIX* plane, const size_t stride, const int gy0, const int gx0,
                                           const int ys, const int ye, const int xs, const int xe, const int fill, const int tid) {
    uint32_t  v[NIT][4];
    const int total = rows * cpr;
#pragma unroll
    for (int k = 0; k < NIT; k++) {
        const int  i = tid + 256 * k, r = i / cpr, c = i - r * cpr;
        const int  gy = gy0 + r, gx = gx0 + 8 * c;
        const bool full = i < total && gy >= ys && gy < ye && gx >= xs && gx + 8 <= xe;
        const PIX* src = plane + (full ? (size_t)gy * stride + gx : (size_t)ys * stride + xs);
        if (sizeof(PIX) == 2) {
            const Row8A2 t = *(const Row8A2*)src;
            v[k][0] = t.v[0]; v[k][1] = t.v[1]; v[k][2] = t.v[2]; v[k][3] = t.v[3];
        } else {
            const Row8A1 t = *(const Row8A1*)src;
            v[k][0] = __builtin_amdgcn_perm(0u, t.v[0], 0x0c010c00u); v[k][1] = __builtin_amdgcn_perm(0u, t.v[0], 0x0c030c02u);
            v[k][2] = __builtin_amdgcn_perm(0u, t.v[1], 0x0c010c00u); v[k][3] = __builtin_amdgcn_perm(0u, t.v[1], 0x0c030c02u);
        }
    }
#pragma unroll
    for (int k = 0; k < NIT; k++) {
        const int i = tid + 256 * k, r = i / cpr, c = i - r * cpr;
        if (i >= total) continue;
        const int  gy = gy0 + r, gx = gx0 + 8 * c;
        const bool rowok = gy >= ys && gy < ye;
        if (!(rowok && gx >= xs && gx + 8 <= xe)) { // frame / filter-block border: pixel-exact
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int  xa = gx + 2 * e, xb = xa + 1;
                const uint32_t lo = (rowok && xa >= xs && xa < xe) ? (uint32_t)plane[(size_t)gy * stride + xa] : (uint32_t)fill;
                const uint32_t hi = (rowok && xb >= xs && xb < xe) ? (uint32_t)plane[(size_t)gy * stride + xb] : (uint32_t)fill;
                v[k][e] = lo | (hi << 16);
            }
        }
        *(Row8A16*)(lds + r * pitch + c * 8) = Row8A16{{v[k][0], v[k][1], v[k][2], v[k][3]}};
    }
}

// 8x8 direction search on registers: px[i][j] = pixel pair (2j, 2j+1) of row i, already (>> coeff_shift) - 128 (svt_aom_cdef_find_dir_c, cdef.c:150-199)
template <int D> __device__ __forceinline__ int dir_cost_regs(const s16x2 (&px)[8][4]) {
    int partial[15];
#pragma unroll
    for (int i = 0; i < 15; i++) partial[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int x   = (int)px[i][j >> 1][j & 1];
            const int idx = D == 0 ? i + j : D == 1 ? i + j / 2 : D == 2 ? i : D == 3 ? 3 + i - j / 2 : D == 4 ? 7 + i - j : D == 5 ? 3 - i / 2 + j : D == 6 ? j : i / 2 + j;
            partial[idx] += x;
        }
    // (unsigned arithmetic: the lanes of units outside the picture or skipped run the same code on whatever their tile holds -- OUTSIDE markers included --, and their
    // results are dropped; a real unit's cost is at most 8 x 1 024^2 x 105 < 2^31 as in the reference)
    auto sq = [](const int p) { return (uint32_t)p * (uint32_t)p; };
    uint32_t cost = 0;
    if (D == 2 || D == 6) {
#pragma unroll
        for (int i = 0; i < 8; i++) cost += sq(partial[i]);
        cost *= 105u;
    } else if (D == 0 || D == 4) {
        constexpr uint32_t div[8] = {840, 420, 280, 210, 168, 140, 120, 105};
#pragma unroll
        for (int i = 0; i < 7; i++) cost += (sq(partial[i]) + sq(partial[14 - i])) * div[i];
        cost += sq(partial[7]) * 105u;
    } else {
#pragma unroll
        for (int j = 0; j < 5; j++) cost += sq(partial[3 + j]);
        cost *= 105u;
        constexpr uint32_t div2[3] = {420, 210, 140};
#pragma unroll
        for (int j = 0; j < 3; j++) cost += (sq(partial[j]) + sq(partial[10 - j])) * div2[j];
    }
    return (int)cost;
}
// Direction search of all 64 units of a filter block by the whole workgroup, free of divergence: wave w evaluates directions w and w + 4
// (a wave-uniform choice) for unit = lane, reading the unit with eight 16-byte LDS loads; the four partial winners meet in LDS.
// Returns, for the unit `b` of the calling lane, the first maximum in index order (strict '>' from best_cost = 0, cdef.c:200-205; costs are sums
// of squares, so "nothing beat 0" means every cost is 0 -> dir 0, var 0, the same result) and var = (best - cost[dir ^ 4]) >> 10.
__device__ __forceinline__ void block_find_dir(const uint16_t* in, const int pitch, const int coeff_shift, const int tid, const int b,
                                               int (*sh_c)[64], int (*sh_d)[64], int (*sh_o)[64], int& best_dir, int& var) {
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), u = tid & 63;
    {
        const uint16_t* img = in + ((u >> 3) * 8) * pitch + (u & 7) * 8;
        s16x2 px[8][4];
        const s16x2 csv = splat(coeff_shift), c128 = splat(128);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const Row8A16 r = *(const Row8A16*)(img + i * pitch);
#pragma unroll
            for (int j = 0; j < 4; j++) px[i][j] = (s16x2)((u16x2)(as_pk(r.v[j]) >> csv) - (u16x2)c128); // (wrapping: an OUTSIDE marker of a unit whose result is dropped)
        }
        int a, o;
        if (w == 0) { a = dir_cost_regs<0>(px); o = dir_cost_regs<4>(px); }
        else if (w == 1) { a = dir_cost_regs<1>(px); o = dir_cost_regs<5>(px); }
        else if (w == 2) { a = dir_cost_regs<2>(px); o = dir_cost_regs<6>(px); }
        else { a = dir_cost_regs<3>(px); o = dir_cost_regs<7>(px); }
        int d = w;
        if (o > a) { const int t = a; a = o; o = t; d = w + 4; }
        sh_c[w][u] = a; sh_d[w][u] = d; sh_o[w][u] = o;
    }
    __syncthreads();
    int c = sh_c[0][b], d = sh_d[0][b], o = sh_o[0][b];
#pragma unroll
    for (int k = 1; k < 4; k++) {
        const int c2 = sh_c[k][b], d2 = sh_d[k][b], o2 = sh_o[k][b];
        if (c2 > c || (c2 == c && d2 < d)) { c = c2; d = d2; o = o2; }
    }
    best_dir = d;
    var      = (c - o) >> 10;
}
__device__ __forceinline__ uint32_t quad_sum(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false); // quad_perm [1,0,3,2]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false); // quad_perm [2,3,0,1]
    return v;
}

// Reduce-scatter over a quad: lane q of the quad receives the quad's total of v_q (q = 0..3) -- three DPP adds and six selects for FOUR values, where summing all
// four values into every lane (quad_sum x 4) and then picking one by lane costs eight DPP adds and a select chain.  Step 1 (partner q ^ 1): a lane keeps the two
// values whose index has its own bit 0 and hands the other two over; step 2 (partner q ^ 2): the same with bit 1.
__device__ __forceinline__ uint32_t quad_reduce_scatter(const uint32_t v0, const uint32_t v1, const uint32_t v2, const uint32_t v3, const int q) {
    const bool b0 = (q & 1) != 0, b1 = (q & 2) != 0;
    uint32_t k0 = b0 ? v1 : v0, k1 = b0 ? v3 : v2;
    const uint32_t s0 = b0 ? v0 : v1, s1 = b0 ? v2 : v3;
    k0 += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s0, 0xB1, 0xf, 0xf, false); // quad_perm [1,0,3,2]
    k1 += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s1, 0xB1, 0xf, 0xf, false);
    uint32_t       kk = b1 ? k1 : k0;
    const uint32_t ss = b1 ? k0 : k1;
    kk += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)ss, 0x4E, 0xf, 0xf, false); // quad_perm [2,3,0,1]
    return kk;
}

typedef uint32_t acc16 __attribute__((vector_size(64)));

// Tile row pitch in pixels: block + two 8-pixel halos, padded by whole 8-pixel chunks until one row of units (uh pixel rows) is half the LDS
// bank space (32 dwords mod 64) away from the next, so the two unit rows a wave touches in one instruction do not share banks.
__host__ __device__ inline int tile_pitch(const int bw, const int uh) {
    int p = bw + 2 * HB;
    while (((uh * p / 2) & 63) != 32) p += 8;
    return p;
}

// Per-lane geometry and strength context of the filter passes
struct LaneCtx {
    const uint16_t* in;  // tile origin (pixel 0,0 of the filter block), u16, pitch `pitch`
    const uint16_t* org; // search mode: source block, pitch bw
    int pitch, bw, by, bx, uw, uh, q, sub, cs, pdamp, sdamp, vm;
};
// SHARED (the search pass): four levels are evaluated on the same taps -- the magnitude / sign form of constrain (see tap_mag)
template <bool SHARED = false>
__device__ __forceinline__ s16x2 pri_sum_level(const LaneCtx& L, const int lvl, const s16x2 x, const s16x2 (&t)[12]) {
    const int t_  = ((lvl << L.cs) * L.vm + 8) >> 4; // adjust_strength (cdef.c:130-134); lane-varying for luma, identity for chroma (vm = 16)
    int       sh  = L.pdamp - msb_u32((uint32_t)t_);
    sh            = sh < 0 ? 0 : sh;
    const int odd = (t_ >> L.cs) & 1; // svt_aom_eb_cdef_pri_taps (cdef.c:249)
    return pri_sum<SHARED>(x, t, splat(t_), splat(sh), splat(4 - odd), splat(2 + odd));
}
template <bool SHARED = false>
__device__ __forceinline__ s16x2 sec_sum_strength(const LaneCtx& L, const int sec, const s16x2 x, const s16x2 (&t)[12]) {
    const int st = sec << L.cs;
    int       sh = L.sdamp - msb_u32((uint32_t)st);
    sh           = sh < 0 ? 0 : sh;
    return sec_sum<SHARED>(x, t, splat(st), splat(sh));
}

// Search: one pass over the lane's pixel pairs for a static grid of NP primary levels (lv[], uniform) x the four secondary strengths
// {0,1,2,4}; NP = 0 is the pass for primary level 0, which the reference filters with dir = 0 (cdef.c:411).  Pixel-pair-major: the twelve
// taps are read from LDS once per pair, the three secondary sums and the NP primary sums are evaluated once, and each of the grid's cells only
// adds, rounds, clamps and updates its three running sums (sum s, sum s^2, sum s*d).  Nothing filtered is written.
// scache (search with several level groups per workgroup): the three secondary sums of a pixel pair depend on the pixel, the direction and the secondary strength
// only -- not on the primary level -- so the first group's pass stores them ([value][row step][thread]: lane-contiguous dwords, conflict-free) and the passes of the
// other groups read them back instead of spending 3 x 59 packed operations per pair again (28 % of the kernel's VALU instructions at four groups).
template <int NP>
__device__ __forceinline__ void search_pass(const LaneCtx& L, const TapOffs& o, const int (&lv)[4], const bool acc_src,
                                            acc16& a_s, acc16& a_s2, acc16& a_sd, uint32_t& d_s, uint32_t& d_s2, uint32_t* scache = nullptr, const bool cached = false) {
    // the quad's lanes sit side by side on one pixel row (lane = pixel pair; 4-wide units: two rows of two pairs) and walk down the unit:
    // a wave instruction then reads two runs of 64 adjacent pixels, which is what the LDS banks like
    const int ppr = L.uw >> 1, jq = L.q & (ppr - 1), r0 = L.uw == 8 ? 0 : L.q >> 1, rstep = 4 / ppr;
    int it = 0;
    for (int r = r0; r < L.uh; r += rstep, ++it) {
        if ((r & (L.sub - 1)) != 0) continue;
        {
            const int       ri   = (L.by * L.uh + r) * L.pitch + L.bx * L.uw + 2 * jq; // pixel index of this lane's pair relative to L.in (even)
            const uint16_t* orow = L.org + (L.by * L.uh + r) * L.bw + L.bx * L.uw + 2 * jq;
            const s16x2 x = ld_pair_even(L.in + ri);
            s16x2 t[12], mn = x, mx = x, S[4];
            // a cell with only ONE of its two strengths non-zero cannot be bound by the clamp to the taps' [min, max] (see apply_pass): the pass for primary level 0
            // (NP = 0) needs no min / max at all and reads the eight secondary taps only, the other passes skip the clamp for their sec = 0 cells
            if (NP) load_taps(L.in + ri, o, x, t, mn, mx);
            else load_taps_range<4, 12>(L.in + ri, o, t);
            S[0] = splat(0);
            uint32_t* sc = scache + it * 256; // (+ the value's plane of 8 x 256 dwords)
            if (NP && scache && cached) {
                S[1] = as_pk(sc[0]); S[2] = as_pk(sc[2048]); S[3] = as_pk(sc[4096]);
            } else {
                S[1] = sec_sum_strength<true>(L, 1, x, t);
                S[2] = sec_sum_strength<true>(L, 2, x, t);
                S[3] = sec_sum_strength<true>(L, 4, x, t);
                if (NP && scache) { sc[0] = as_u32(S[1]); sc[2048] = as_u32(S[2]); sc[4096] = as_u32(S[3]); }
            }
            const uint32_t dpair = as_u32(ld_pair_even(orow));
            if (acc_src) {
                d_s += dpair; // u16 halves: at most 8 pixels of 4095 each per half
                d_s2 = dot2(dpair, dpair, d_s2);
            }
#pragma unroll
            for (int pi = 0; pi < (NP ? NP : 1); pi++) {
                const s16x2 Pv = NP ? pri_sum_level<true>(L, lv[pi], x, t) : splat(0);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const s16x2    sum = Pv + S[k];
                    const uint32_t yu  = as_u32((NP && k) ? finish_px(x, sum, mn, mx) : x + ((sum + splat(8) + (sum >> 15)) >> 4));
                    const int      c  = 4 * pi + k;
                    a_s[c] += yu;
                    a_s2[c] = dot2(yu, yu, a_s2[c]);
                    a_sd[c] = dot2(yu, dpair, a_sd[c]);
                }
            }
        }
    }
}
// quad totals in every lane, then lane q evaluates cells q, q + 4, ... (the double-precision distortion is the long pole) and adds them to
// the block's per-cell totals cells[0 .. 4 * NG)
template <int NG>
__device__ __forceinline__ void search_reduce(acc16& a_s, acc16& a_s2, acc16& a_sd, const uint32_t d_s, const uint32_t d_s2,
                                              const int q, const bool act, const bool weighted, const int cs, unsigned long long* cells) {
    // (every quantity of an 8x8 unit fits 32 bits -- 64 samples of at most 4095: sum <= 262 080, sum of squares / products <= 1.08e9 -- so the integer side stays in
    // dwords and the conversions to double are single instructions; only sum * sum needs 64 bits)
    const uint32_t sd = quad_sum((d_s & 0xffffu) + (d_s >> 16)), sd2 = quad_sum(d_s2);
#pragma unroll
    for (int k = 0; k < NG; k++) {
        // lane q of the quad takes cell 4 k + q: the quad's totals of that cell's three sums by reduce-scatter
        const uint32_t ss  = quad_reduce_scatter((a_s[4 * k] & 0xffffu) + (a_s[4 * k] >> 16), (a_s[4 * k + 1] & 0xffffu) + (a_s[4 * k + 1] >> 16),
                                                 (a_s[4 * k + 2] & 0xffffu) + (a_s[4 * k + 2] >> 16), (a_s[4 * k + 3] & 0xffffu) + (a_s[4 * k + 3] >> 16), q);
        const uint32_t ss2 = quad_reduce_scatter(a_s2[4 * k], a_s2[4 * k + 1], a_s2[4 * k + 2], a_s2[4 * k + 3], q);
        const uint32_t ssd = quad_reduce_scatter(a_sd[4 * k], a_sd[4 * k + 1], a_sd[4 * k + 2], a_sd[4 * k + 3], q);
        if (act) {
            unsigned long long dist = (unsigned long long)sd2 + ss2 - 2ull * ssd; // = sum (d - s)^2
            if (weighted) { // dist_8xn_*_c, enc_cdef.c:23-48: IEEE double, no contraction (-ffp-contract=off)
                const uint32_t svar = ss2 - (uint32_t)(((unsigned long long)ss * ss + 32) >> 6), dvar = sd2 - (uint32_t)(((unsigned long long)sd * sd + 32) >> 6);
                const double num = (double)dist * .5 * (double)((unsigned long long)svar + dvar + (unsigned long long)(400 << 2 * cs));
                const double den = sqrt((double)(20000 << 4 * cs) + (double)svar * (double)dvar);
                dist = (unsigned long long)floor(.5 + num / den);
            }
            atomicAdd(&cells[4 * k + q], dist);
        }
    }
}
// Apply: the block's own (level, secondary strength); a lane writes whole rows (16 / 8 bytes) of its unit
template <typename PIX>
__device__ __forceinline__ void apply_pass(const LaneCtx& L, const TapOffs& o, const int lvl, const int sec, PIX* out, const size_t out_stride,
                                           const int fbr, const int fbc) {
    const int ppr = L.uw >> 1, jq = L.q & (ppr - 1), r0 = L.uw == 8 ? 0 : L.q >> 1, rstep = 4 / ppr; // same lane layout as search_pass
    for (int r = r0; r < L.uh; r += rstep) {
        const int   ri = (L.by * L.uh + r) * L.pitch + L.bx * L.uw + 2 * jq;
        const s16x2 x  = ld_pair_even(L.in + ri);
        s16x2 t[12], yv;
        // With only ONE of the two strengths non-zero the reference's clamp to [min, max] of the taps (cdef.c:300) cannot bind: every constrained difference lies
        // between 0 and its tap's difference, the four primary (eight secondary) weights sum to 12 of 16, and (12 D + 8) >> 4 <= D for every D >= 0 (likewise
        // downwards) -- so the result already lies between the centre and the extreme valid tap.  Those filter blocks (strengths are per filter block, the branch
        // is workgroup-uniform) read 4 or 8 taps instead of 12 and skip the 24 packed min / max operations.  Out-of-frame taps constrain to 0 either way.
        if (sec == 0) {
            load_taps_range<0, 4>(L.in + ri, o, t);
            const s16x2 sum = pri_sum_level(L, lvl, x, t);
            yv = x + ((sum + splat(8) + (sum >> 15)) >> 4);
        } else if (lvl == 0) {
            load_taps_range<4, 12>(L.in + ri, o, t);
            const s16x2 sum = sec_sum_strength(L, sec, x, t);
            yv = x + ((sum + splat(8) + (sum >> 15)) >> 4);
        } else {
            s16x2 mn, mx;
            load_taps(L.in + ri, o, x, t, mn, mx);
            yv = finish_px(x, pri_sum_level(L, lvl, x, t) + sec_sum_strength(L, sec, x, t), mn, mx);
        }
        const uint32_t y  = as_u32(yv);
        const size_t   gy = (size_t)(fbr * (L.uh * 8) + L.by * L.uh + r);
        const int      gx = fbc * L.bw + L.bx * L.uw + 2 * jq;
        PIX*           op = out + gy * out_stride + gx;
        if (sizeof(PIX) == 2) *(PairA2*)op = PairA2{y}; // the quad writes 16 (8) contiguous bytes of the row
        else *(U16A1*)op = U16A1{(uint16_t)((y & 0xffu) | ((y >> 8) & 0xff00u))};
    }
}

// One workgroup = one 64x64 filter block (apply) or one filter block x one group of four primary levels (search: blockIdx.y = g takes the
// 4g-th .. (4g+3)-th distinct non-zero primary levels of the candidate list, each against all four secondary strengths; g = 0 also takes
// primary level 0).  A quad owns an 8x8 (4x4 / 4x8 / 8x4 for subsampled chroma) unit; a lane filters uh / 4 rows, two pixels per packed op.
// MINB = workgroups per CU the register budget is cut for (__launch_bounds__): the apply kernel fits 4 (101 VGPRs); the search pass keeps 48 running sums (16 cells x
// {s, s^2, s*d}) + 12 taps + 7 partial sums per lane and at 4 (128 VGPRs) spilt 172 bytes per lane -- 89 MB of scratch traffic per 4K plane (profiles/r02_kernel_resources.txt,
// r02_reg9_pmc_traffic.json) -- so it is built for 3 (up to 170 VGPRs, no scratch).
template <typename PIX, int MODE, int MINB = (MODE == 1 ? 3 : 4)>
__global__ __launch_bounds__(256, MINB) void cdef_frame_kernel(const SvtHipCdefParams P, const int gpw, const int reuse_dir, const int fb0 /* first filter block of the launch */) {
    HIP_DYNAMIC_SHARED(uint16_t, tile_raw)
    __shared__ int                sh_any, sh_dc[4][64], sh_dd[4][64], sh_do[4][64];
    __shared__ unsigned long long sh_cells[20]; // [4 levels][4 secondary] + [level 0][4 secondary]
    const int tid = threadIdx.x;
    const int xdec = P.xdec, ydec = P.ydec, pli = P.pli, cs = P.coeff_shift;
    const int bw = 64 >> xdec, bh = 64 >> ydec, uw = 8 >> xdec, uh = 8 >> ydec;
    const int pw = (int)P.width, ph = (int)P.height;
    const int nhfb = (pw + bw - 1) / bw, nvfb = (ph + bh - 1) / bh;
    // XCD-aware order: a filter block's tile shares its halo lines with the neighbouring blocks' tiles (the apply pass moved 2.2 x the algorithmic bytes)
    const int fb = fb0 + (int)xcd_remap(blockIdx.x, gridDim.x), fbr = fb / nhfb, fbc = fb % nhfb;
    const int pitch = tile_pitch(bw, uh);
    // search: this workgroup takes the level groups g0 .. g0 + gpw - 1 of its filter block one after the other, on ONE staged tile (large frames: gpw = 2
    // or 4, so the tile, the source block and the direction search are not repeated per group; small frames keep gpw = 1 to fill the chip)
    const int g0 = MODE == 1 ? (int)blockIdx.y * gpw : 0, ncand = MODE == 1 ? (int)P.ncand : 0;
    uint32_t all_levels = 0;
    if (MODE == 1) {
        for (int c = 0; c < ncand; c++) all_levels |= 1u << (P.pri[c] & 15);
        uint32_t rest = all_levels & ~1u;
        for (int k = 0; k < 4 * g0 && rest; k++) rest &= rest - 1;
        if (rest == 0 && !((all_levels & 1u) && g0 == 0) && g0 != 0) return; // nothing left for this workgroup's groups
    }

    const int b = tid >> 2, q = tid & 3, by = b >> 3, bx = b & 7;
    const bool inside = (fbc * 8 + bx) * uw < pw && (fbr * 8 + by) * uh < ph;
    const int  act    = inside && !P.skip[(size_t)(fbr * 8 + by) * (nhfb * 8) + fbc * 8 + bx];
    if (tid == 0) sh_any = 0;
    if (tid < 20) sh_cells[tid] = 0;
    __syncthreads();
    if (act) sh_any = 1;
    __syncthreads();
    if (sh_any == 0) {
        if (MODE == 1 && g0 == 0)
            for (int c = tid; c < ncand; c += 256) P.mse[(size_t)fb * P.ncand + c] = 0;
        return;
    }
    {   // stage the tile (cdef_process.c:208-228): real pixels where the neighbouring filter block exists, else OUTSIDE
        const int x0 = fbc * bw, y0 = fbr * bh;
        const int xs = x0 - (fbc != 0 ? HB : 0), ys = y0 - (fbr != 0 ? VB : 0);
        // (never past the picture: a 4:2:0 chroma plane's last filter block can be 4 samples wide -- narrower than the 8-sample halo its left neighbour stages; the
        // taps reach 2 samples, so what lies beyond the picture is never used, but it must not be READ either: AddressSanitizer on the emulator, last row of a plane)
        const int xe_ = (x0 + bw < pw ? x0 + bw : pw) + (fbc + 1 < nhfb ? HB : 0), xe = xe_ < pw ? xe_ : pw;
        const int ye_ = (y0 + bh < ph ? y0 + bh : ph) + (fbr + 1 < nvfb ? VB : 0), ye = ye_ < ph ? ye_ : ph;
        stage_tile<PIX, 3>(tile_raw, pitch, bh + 2 * VB, (bw + 2 * HB) >> 3, (const PIX*)P.recon, P.recon_stride, y0 - VB, x0 - HB, ys, ye, xs, xe, OUTSIDE, tid);
        if (MODE == 1)
            stage_tile<PIX, 2>(tile_raw + (bh + 2 * VB) * pitch, bw, bh, bw >> 3, (const PIX*)P.source, P.source_stride, y0, x0, y0,
                               y0 + bh < ph ? y0 + bh : ph, x0, x0 + bw < pw ? x0 + bw : pw, 0, tid);
    }
    __syncthreads();
    LaneCtx L;
    L.in  = tile_raw + VB * pitch + HB;
    L.org = tile_raw + (bh + 2 * VB) * pitch; // search mode: the source block, pitch bw
    int dir = 0, var = 0;
    if (pli == 0 && !reuse_dir) {
        block_find_dir(L.in, pitch, cs, tid, b, sh_dc, sh_dd, sh_do, dir, var); // every unit is searched (inactive ones: result unused)
        if (!act) { dir = 0; var = 0; }
        if (q == 0 && g0 == 0) { P.dir[(size_t)fb * 64 + b] = (uint8_t)dir; P.var[(size_t)fb * 64 + b] = var; }
    } else {
        dir = P.dir[(size_t)fb * 64 + b] & 7;
        if (xdec != ydec) // cdef.c:388-395: conv422 = {7,0,2,4,5,6,6,6}, conv440 = {1,2,2,2,3,4,6,0}, one nibble per direction
            dir = (int)(((xdec ? 0x66654207u : 0x06432221u) >> (4 * dir)) & 7u);
        var = P.var[(size_t)fb * 64 + b];
    }
    L.pitch = pitch; L.bw = bw; L.by = by; L.bx = bx; L.uw = uw; L.uh = uh; L.q = q; L.cs = cs;
    L.sub   = MODE == 1 ? P.subsampling : 1;
    L.pdamp = P.pri_damping + cs - (pli != 0);
    L.sdamp = P.sec_damping + cs - (pli != 0);
    // adjust_strength (cdef.c:130-134) = (strength * vm + 8) >> 4 with vm = var ? 4 + min(msb(var >> 6), 12) : 0; chroma: vm = 16 (identity)
    L.vm = 16;
    if (pli == 0) {
        const int v6 = var >> 6;
        int       i  = v6 ? msb_u32((uint32_t)v6) : 0;
        i            = i > 12 ? 12 : i;
        L.vm         = var ? 4 + i : 0;
    }
    if (MODE == 0) {
        const int lvl = P.pri[fb], sec = P.sec[fb];
        if (lvl == 0 && sec == 0) return; // zero strength leaves the block unchanged (enc_cdef.c:571); dir / var are still reported
        if (act) apply_pass<PIX>(L, tap_offs(lvl ? dir : 0, pitch), lvl, sec, (PIX*)P.out, P.out_stride, fbr, fbc);
        return;
    }
    const bool weighted = pli == 0 && uw == 8 && uh == 8;
    // secondary-sum cache of the lane's pixel pairs (3 values x 8 row steps x 256 threads), behind the tile and the source block; only with several groups per workgroup
    uint32_t*  scache  = gpw > 1 ? (uint32_t*)(tile_raw + (((bh + 2 * VB) * pitch + bh * bw + 1) & ~1)) + tid : nullptr;
    bool       s_ready = false; // (uniform) a pass with the block's own direction has filled the cache
    for (int gi = 0; gi < gpw; gi++) {
    const int g = g0 + gi;
    // which primary levels does group g own?  (uniform; the list has at most 64 entries)
    int  lv[4] = {0, 0, 0, 0}, nlv = 0;
    bool do0 = (all_levels & 1u) && g == 0;
    {
        uint32_t levels = all_levels & ~1u;
        for (int k = 0; k < 4 * g && levels; k++) levels &= levels - 1; // drop the levels of the groups before this one
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (levels) { lv[k] = __builtin_ctz(levels); levels &= levels - 1; nlv = k + 1; }
    }
    if (nlv == 0 && !do0) break;
    if (gi) { // the cells of the previous group have been read out
        __syncthreads();
        if (tid < 20) sh_cells[tid] = 0;
        __syncthreads();
    }
    acc16    a_s, a_s2, a_sd; // vector values, not arrays: they must never become stack objects
    uint32_t d_s = 0, d_s2 = 0;
    if (nlv) {
        a_s = a_s2 = a_sd = acc16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (act) search_pass<4>(L, tap_offs(dir, pitch), lv, true, a_s, a_s2, a_sd, d_s, d_s2, scache, s_ready);
        s_ready = true; // (a lane reads back only what it wrote itself: no barrier needed)
        search_reduce<4>(a_s, a_s2, a_sd, d_s, d_s2, q, act, weighted, cs, sh_cells);
    }
    if (do0) {
        a_s = a_s2 = a_sd = acc16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const bool again = nlv != 0; // the source sums are already there
        if (act) search_pass<0>(L, tap_offs(0, pitch), lv, !again, a_s, a_s2, a_sd, d_s, d_s2);
        search_reduce<1>(a_s, a_s2, a_sd, d_s, d_s2, q, act, weighted, cs, sh_cells + 16);
    }
    __syncthreads();
    for (int c = tid; c < ncand; c += 256) { // every candidate is reported by exactly one group
        const int lvl = P.pri[c] & 15, sec = P.sec[c];
        const int k   = sec == 0 ? 0 : sec == 1 ? 1 : sec == 2 ? 2 : 3;
        int cell = -1;
        if (lvl == 0) { if (g == 0) cell = 16 + k; }
        else {
#pragma unroll
            for (int pi = 0; pi < 4; pi++)
                if (pi < nlv && lv[pi] == lvl) cell = 4 * pi + k;
        }
        if (cell >= 0) P.mse[(size_t)fb * P.ncand + c] = sh_cells[cell] >> (2 * cs);
    }
    } // group loop
}

// ---- single-call kernels ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void find_dir_kernel(const uint16_t* img /* n blocks of 8x8, pitch 8 */, int n, int coeff_shift, int* out /* dir,var pairs */) {
    const int tid = threadIdx.x, b = tid >> 2, q = tid & 3;
    int d = 0, v = 0;
    quad_find_dir(img + (b < n ? b : 0) * 64, 8, coeff_shift, q, d, v);
    if (q == 0 && b < n) { out[2 * b] = d; out[2 * b + 1] = v; }
}
template <typename PIX>
__global__ __launch_bounds__(64) void filter_block_kernel(PIX* dst, int dstride, const uint16_t* tile /* (bh+4) x (bw+4), pitch bw+4, origin (2,2) */, int pri,
                                                          int sec, int dir, int pdamp, int sdamp, int bw, int bh, int cs, int sub) {
    const int pitch = bw + 4, i = threadIdx.x >> 3, j = threadIdx.x & 7;
    if (i >= bh || j >= bw || (i % sub) != 0) return;
    const FilterCtx ctx = make_ctx(pri, sec, dir, pdamp, sdamp, cs, pitch);
    dst[i * dstride + j] = (PIX)filter_px(tile + (i + 2) * pitch + 2 + j, ctx);
}
template <typename PIX>
__global__ __launch_bounds__(64) void cdef_dist_kernel(const PIX* plane /* packed per block like `packed` */, const PIX* packed, int count, int bw, int bh, int cs,
                                                       int pli, int sub, unsigned long long* out) {
    __shared__ unsigned long long total;
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    for (int bi = threadIdx.x; bi < count; bi += 64) {
        unsigned long long ss = 0, sd = 0, ss2 = 0, sd2 = 0, ssd = 0, mse = 0;
        for (int i = 0; i < bh; i += sub)
            for (int j = 0; j < bw; j++) {
                const unsigned d = plane[bi * bw * bh + i * bw + j], s = packed[bi * bw * bh + i * bw + j];
                ss += s; sd += d; ss2 += s * s; sd2 += d * d; ssd += s * d;
                const int e = (int)d - (int)s;
                mse += (unsigned long long)(long long)(e * e);
            }
        unsigned long long dist = mse;
        if (pli == 0 && bw == 8 && bh == 8) {
            const unsigned long long svar = ss2 - ((ss * ss + 32) >> 6), dvar = sd2 - ((sd * sd + 32) >> 6);
            const double num = (double)(sd2 + ss2 - 2 * ssd) * .5 * (double)(svar + dvar + (unsigned long long)(400 << 2 * cs));
            const double den = sqrt((double)(20000 << 4 * cs) + (double)svar * (double)dvar);
            dist = (unsigned long long)floor(.5 + num / den);
        }
        atomicAdd(&total, dist);
    }
    __syncthreads();
    if (threadIdx.x == 0) out[0] = total >> (2 * cs);
}
__global__ void copy_rect8_to_16_kernel(uint16_t* dst, const uint8_t* src, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

// fbr0 / fbr1: filter-block rows [fbr0, fbr1) of the plane (the whole plane by default; a strip of rows when a picture is split over several GPUs -- the tiles of
// the strip's first and last row still read their halos from the full input plane)
template <int MODE> void launch_frame(const SvtHipCdefParams& P, hipStream_t st, const int reuse_dir = 0, int fbr0 = 0, int fbr1 = -1) {
    const int bw = 64 >> P.xdec, bh = 64 >> P.ydec;
    const int nhfb = ((int)P.width + bw - 1) / bw, nvfb_all = ((int)P.height + bh - 1) / bh;
    if (fbr1 < 0 || fbr1 > nvfb_all) fbr1 = nvfb_all;
    if (fbr0 < 0) fbr0 = 0;
    if (fbr0 >= fbr1) return;
    const int nvfb = fbr1 - fbr0, fb0 = fbr0 * nhfb;
    size_t shmem = (size_t)((bh + 2 * VB) * tile_pitch(bw, 8 >> P.ydec) + (MODE == 1 ? bh * bw : 0)) * 2 + 64;
    // search: four groups of four primary levels; a workgroup takes gpw of them on one staged tile as long as >= 4096 workgroups remain (256 CUs x 4 x 4 rounds)
    int gpw = 1;
    if (MODE == 1) {
        const int e = svthip::tuning_cdef_groups_per_workgroup(); // SVT_HIP_CDEF_GPW (measurement override, read once)
        gpw = e ? e : (nhfb * nvfb >= 2040 ? 4 : (nhfb * nvfb >= 1020 ? 2 : 1)); // 4K luma: 378 us with 4, 386 with 2, 418 with 1 (profiles/r02_call8_*)
    }
    if (MODE == 1 && gpw > 1) shmem += 3 * 8 * 256 * 4; // the secondary-sum cache (search_pass)
    const dim3 grid(nhfb * nvfb, MODE == 1 ? 4 / gpw : 1);
    // (the round-2 register budget of the search kernel -- 128 VGPRs, 196 B of scratch, 343 us on the 4K luma plane against 288 us -- is gone: profiles/r03_call3_ab_sad_cdef.txt)
    if (false) {
    } else if (MODE == 1 && svthip::tuning_cdef_search_minb() == 2) { // (SVT_HIP_CDEF_MINB=2: 172 VGPRs, no scratch at all, two workgroups per CU)
        if (P.is_16bit) hipLaunchKernelGGL(HIP_KERNEL_NAME(cdef_frame_kernel<uint16_t, MODE, 2>), grid, dim3(256), shmem, st, P, gpw, reuse_dir, fb0);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(cdef_frame_kernel<uint8_t, MODE, 2>), grid, dim3(256), shmem, st, P, gpw, reuse_dir, fb0);
    } else if (P.is_16bit) hipLaunchKernelGGL(HIP_KERNEL_NAME(cdef_frame_kernel<uint16_t, MODE>), grid, dim3(256), shmem, st, P, gpw, reuse_dir, fb0);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(cdef_frame_kernel<uint8_t, MODE>), grid, dim3(256), shmem, st, P, gpw, reuse_dir, fb0);
    SVT_LAUNCH_CHECK();
}
const int kBlkW[4] = {4, 4, 8, 8}, kBlkH[4] = {4, 8, 4, 8}; // BLOCK_4X4, 4X8, 8X4, 8X8 (definitions.h)

} // namespace

extern "C" {

void svt_hip_cdef_frame(int mode, const SvtHipCdefParams* params, void* stream) { svt_hip_cdef_frame_rows(mode, params, 0, -1, stream); }
void svt_hip_cdef_frame_rows(int mode, const SvtHipCdefParams* params, int fb_row_begin, int fb_row_end, void* stream) {
    svthip::ensure_device();
    if (mode == 1 && params->ncand == 0) return;
    if (mode == 0) launch_frame<0>(*params, (hipStream_t)stream, 0, fb_row_begin, fb_row_end);
    else if (mode == 2) launch_frame<0>(*params, (hipStream_t)stream, 1, fb_row_begin, fb_row_end);
    else launch_frame<1>(*params, (hipStream_t)stream, 0, fb_row_begin, fb_row_end);
}

// Host-pointer form of the frame apply for all planes of a 4:2:0 picture (what a seam around svt_av1_cdef_frame, cdef_process.c:458, calls): uploads the
// planes, the 8x8 skip map and the per-filter-block strengths, filters luma (which produces the directions / variances) then chroma out of place on the
// device, downloads the filtered planes IN PLACE.  Every pointer is a host pointer.
int svt_hip_cdef_apply_host(const SvtHipCdefApplyHost* a) {
    SVT_HIP_ENTRY_TRY
    svthip::ensure_device();
    const size_t px = a->is_16bit ? 2 : 1;
    const uint32_t nhfb = (a->width + 63) / 64, nvfb = (a->height + 63) / 64, nfb = nhfb * nvfb;
    size_t pitch[3], rows[3], wid[3], total = 0;
    for (int p = 0; p < a->num_planes; p++) {
        wid[p]   = p ? (a->width + 1) >> 1 : a->width; // 4:2:0 chroma of an odd luma size: (w + ss_x) >> ss_x, as the reference's picture buffers
        rows[p]  = p ? (a->height + 1) >> 1 : a->height;
        pitch[p] = svthip::align_up(wid[p] * px, 16);
        total += 2 * pitch[p] * rows[p];
    }
    const size_t side = (size_t)nfb * 64 * (1 + 4) + (size_t)nvfb * 8 * nhfb * 8 + (size_t)nfb * 4 * 4 + 16384;
    svthip::HostCallLease lease; // (a pooled arena: see svt_hip_common.h)
    svthip::HostCall& c = *lease;
    c.begin();
    c.reserve(total + side, total + side);
    uint8_t* d_skip = (uint8_t*)c.dalloc((size_t)nvfb * 8 * nhfb * 8);
    int32_t* d_str[4];
    const int32_t* h_str[4] = {a->pri_y, a->sec_y, a->pri_uv, a->sec_uv};
    for (int k = 0; k < (a->num_planes > 1 ? 4 : 2); k++) { // a monochrome caller (num_planes == 1) passes no chroma strengths
        d_str[k] = (int32_t*)c.dalloc((size_t)nfb * 4);
        c.up(d_str[k], h_str[k], (size_t)nfb * 4);
    }
    uint8_t* d_dir = (uint8_t*)c.dalloc((size_t)nfb * 64);
    int32_t* d_var = (int32_t*)c.dalloc((size_t)nfb * 64 * 4);
    c.up(d_skip, a->skip, (size_t)nvfb * 8 * nhfb * 8);
    uint8_t *d_in[3], *d_out[3];
    for (int p = 0; p < a->num_planes; p++) {
        d_in[p]  = (uint8_t*)c.dalloc(pitch[p] * rows[p]);
        d_out[p] = (uint8_t*)c.dalloc(pitch[p] * rows[p]);
        c.up2d(d_in[p], pitch[p], a->plane[p], (size_t)a->stride[p] * px, wid[p] * px, rows[p]);
        HIP_CHECK(hipMemcpyAsync(d_out[p], d_in[p], pitch[p] * rows[p], hipMemcpyDeviceToDevice, c.stream));
        SvtHipCdefParams P;
        memset(&P, 0, sizeof(P));
        P.recon = d_in[p]; P.out = d_out[p];
        P.recon_stride = P.out_stride = (uint32_t)(pitch[p] / px);
        P.width = (uint32_t)wid[p]; P.height = (uint32_t)rows[p];
        P.xdec = P.ydec = (uint8_t)(p ? 1 : 0); P.pli = (uint8_t)p; P.is_16bit = a->is_16bit;
        P.coeff_shift = a->coeff_shift; P.pri_damping = P.sec_damping = a->damping; P.subsampling = 1;
        P.skip = d_skip; P.pri = d_str[p ? 2 : 0]; P.sec = d_str[p ? 3 : 1]; P.dir = d_dir; P.var = d_var;
        svthip::cdef_frame_dispatch(0, &P, c.stream);
    }
    for (int p = 0; p < a->num_planes; p++) c.down2d_later(a->plane[p], (size_t)a->stride[p] * px, d_out[p], pitch[p], wid[p] * px, rows[p]);
    c.finish(); // (one synchronisation for the three planes)
    return 0;
    SVT_HIP_ENTRY_CATCH(SVT_HIP_E_DEVICE)
}

// Host-pointer form of the strength SEARCH for the three planes of a 4:2:0 picture (what a seam around cdef_seg_search, cdef_process.c:443, calls once per
// picture): the distortion of every candidate strength for every filter block, luma directions / variances included.  Every pointer is a host pointer.
int svt_hip_cdef_search_host(const SvtHipCdefSearchHost* a) {
    SVT_HIP_ENTRY_TRY
    svthip::ensure_device();
    const size_t   px = a->is_16bit ? 2 : 1;
    const uint32_t nhfb = (a->width + 63) / 64, nvfb = (a->height + 63) / 64, nfb = nhfb * nvfb;
    size_t pitch[3], rows[3], wid[3], total = 0;
    for (int p = 0; p < 3; p++) {
        wid[p]   = p ? (a->width + 1) >> 1 : a->width;
        rows[p]  = p ? (a->height + 1) >> 1 : a->height;
        pitch[p] = svthip::align_up(wid[p] * px, 16);
        total += 2 * pitch[p] * rows[p];
    }
    const size_t nmse = (size_t)nfb * (a->ncand_y + 2 * (size_t)a->ncand_uv);
    const size_t side = (size_t)nfb * 64 * 5 + (size_t)nvfb * 8 * nhfb * 8 + nmse * 8 + ((size_t)a->ncand_y + a->ncand_uv) * 8 + 16384;
    svthip::HostCallLease lease; // (a pooled arena: see svt_hip_common.h)
    svthip::HostCall& c = *lease;
    c.begin();
    c.reserve(total + side, total + side + nmse * 8);
    uint8_t* d_skip = (uint8_t*)c.dalloc((size_t)nvfb * 8 * nhfb * 8);
    c.up(d_skip, a->skip, (size_t)nvfb * 8 * nhfb * 8);
    int32_t *d_pri[2], *d_sec[2];
    const int32_t* h_pri[2] = {a->pri_y, a->pri_uv};
    const int32_t* h_sec[2] = {a->sec_y, a->sec_uv};
    const uint32_t nc[2] = {a->ncand_y, a->ncand_uv};
    for (int k = 0; k < 2; k++) {
        d_pri[k] = (int32_t*)c.dalloc((size_t)(nc[k] ? nc[k] : 1) * 4); d_sec[k] = (int32_t*)c.dalloc((size_t)(nc[k] ? nc[k] : 1) * 4);
        if (nc[k]) { c.up(d_pri[k], h_pri[k], (size_t)nc[k] * 4); c.up(d_sec[k], h_sec[k], (size_t)nc[k] * 4); }
    }
    uint8_t* d_dir = (uint8_t*)c.dalloc((size_t)nfb * 64);
    int32_t* d_var = (int32_t*)c.dalloc((size_t)nfb * 64 * 4);
    HIP_CHECK(hipMemsetAsync(d_dir, 0, (size_t)nfb * 64, c.stream));
    HIP_CHECK(hipMemsetAsync(d_var, 0, (size_t)nfb * 64 * 4, c.stream));
    uint64_t* d_mse[3];
    uint64_t* h_mse[3] = {a->mse_y, a->mse_u, a->mse_v};
    for (int p = 0; p < 3; p++) {
        const uint32_t ncand = nc[p ? 1 : 0];
        d_mse[p] = (uint64_t*)c.dalloc((size_t)nfb * (ncand ? ncand : 1) * 8);
        if (!ncand) continue;
        uint8_t* d_rec = (uint8_t*)c.dalloc(pitch[p] * rows[p]);
        uint8_t* d_src = (uint8_t*)c.dalloc(pitch[p] * rows[p]);
        c.up2d(d_rec, pitch[p], a->recon[p], (size_t)a->recon_stride[p] * px, wid[p] * px, rows[p]);
        c.up2d(d_src, pitch[p], a->source[p], (size_t)a->source_stride[p] * px, wid[p] * px, rows[p]);
        HIP_CHECK(hipMemsetAsync(d_mse[p], 0, (size_t)nfb * ncand * 8, c.stream));
        SvtHipCdefParams P;
        memset(&P, 0, sizeof(P));
        P.recon = d_rec; P.source = d_src;
        P.recon_stride = P.source_stride = (uint32_t)(pitch[p] / px);
        P.width = (uint32_t)wid[p]; P.height = (uint32_t)rows[p];
        P.xdec = P.ydec = (uint8_t)(p ? 1 : 0); P.pli = (uint8_t)p; P.is_16bit = a->is_16bit;
        P.coeff_shift = a->coeff_shift; P.pri_damping = P.sec_damping = a->damping; P.subsampling = a->subsampling[p ? 1 : 0];
        P.ncand = ncand; P.skip = d_skip; P.pri = d_pri[p ? 1 : 0]; P.sec = d_sec[p ? 1 : 0]; P.dir = d_dir; P.var = d_var; P.mse = d_mse[p];
        svthip::cdef_frame_dispatch(1, &P, c.stream);
    }
    for (int p = 0; p < 3; p++)
        if (nc[p ? 1 : 0]) c.down_later(h_mse[p], d_mse[p], (size_t)nfb * nc[p ? 1 : 0] * 8);
    c.down_later(a->dir, d_dir, (size_t)nfb * 64);
    c.down_later(a->var, d_var, (size_t)nfb * 64 * 4);
    c.finish(); // (one synchronisation for the five arrays)
    return 0;
    SVT_HIP_ENTRY_CATCH(SVT_HIP_E_DEVICE)
}

uint8_t svt_aom_cdef_find_dir_hip(const uint16_t* img, int32_t stride, int32_t* var, int32_t coeff_shift) {
    svthip::HostCall& c = svthip::host_call();
    c.begin_small();
    c.reserve(4096, 4096);
    uint16_t* d = (uint16_t*)c.dalloc(128);
    int*      o = (int*)c.dalloc(8);
    c.up2d(d, 16, img, (size_t)stride * 2, 16, 8);
    hipLaunchKernelGGL(find_dir_kernel, dim3(1), dim3(64), 0, c.stream, (const uint16_t*)d, 1, coeff_shift, o);
    SVT_LAUNCH_CHECK();
    int h[2];
    c.down(h, o, 8);
    *var = h[1];
    return (uint8_t)h[0];
}
void svt_aom_cdef_find_dir_dual_hip(const uint16_t* img1, const uint16_t* img2, int stride, int32_t* var1, int32_t* var2, int32_t coeff_shift,
                                    uint8_t* out1, uint8_t* out2) {
    svthip::HostCall& c = svthip::host_call();
    c.begin_small();
    c.reserve(4096, 4096);
    uint16_t* d = (uint16_t*)c.dalloc(256);
    int*      o = (int*)c.dalloc(16);
    c.up2d(d, 16, img1, (size_t)stride * 2, 16, 8);
    c.up2d(d + 64, 16, img2, (size_t)stride * 2, 16, 8);
    hipLaunchKernelGGL(find_dir_kernel, dim3(1), dim3(64), 0, c.stream, (const uint16_t*)d, 2, coeff_shift, o);
    SVT_LAUNCH_CHECK();
    int h[4];
    c.down(h, o, 16);
    *out1 = (uint8_t)h[0]; *var1 = h[1]; *out2 = (uint8_t)h[2]; *var2 = h[3];
}

void svt_cdef_filter_block_hip(uint8_t* dst8, uint16_t* dst16, int32_t dstride, const uint16_t* in, int32_t pri_strength, int32_t sec_strength, int32_t dir,
                               int32_t pri_damping, int32_t sec_damping, int32_t bsize, int32_t coeff_shift, uint8_t subsampling_factor) {
    const int bw = kBlkW[bsize & 3], bh = kBlkH[bsize & 3], pitch = bw + 4;
    svthip::HostCall& c = svthip::host_call();
    c.begin_small();
    c.reserve(8192, 8192);
    uint16_t* dt = (uint16_t*)c.dalloc((size_t)(bh + 4) * pitch * 2);
    void*     dd = c.dalloc(8 * 8 * 2);
    // the taps reach at most 2 pixels in every direction (Cdef_Directions): upload that neighbourhood of the 144-pitch tile
    c.up2d(dt, (size_t)pitch * 2, in - 2 * 144 - 2, 144 * 2, (size_t)pitch * 2, bh + 4);
    const size_t px = dst8 ? 1 : 2;
    c.up2d(dd, 8 * px, dst8 ? (void*)dst8 : (void*)dst16, (size_t)dstride * px, (size_t)bw * px, bh); // rows skipped by subsampling keep their content
    if (dst8)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(filter_block_kernel<uint8_t>), dim3(1), dim3(64), 0, c.stream, (uint8_t*)dd, 8, (const uint16_t*)dt, pri_strength,
                           sec_strength, dir, pri_damping, sec_damping, bw, bh, coeff_shift, (int)subsampling_factor);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(filter_block_kernel<uint16_t>), dim3(1), dim3(64), 0, c.stream, (uint16_t*)dd, 8, (const uint16_t*)dt, pri_strength,
                           sec_strength, dir, pri_damping, sec_damping, bw, bh, coeff_shift, (int)subsampling_factor);
    SVT_LAUNCH_CHECK();
    c.down2d(dst8 ? (void*)dst8 : (void*)dst16, (size_t)dstride * px, dd, 8 * px, (size_t)bw * px, bh);
}

static uint64_t cdef_dist_host(const void* dst, int32_t dstride, const void* src, const uint8_t* dlist, int32_t cdef_count, int bsize, int32_t coeff_shift,
                               int32_t pli, uint8_t sub, int px) {
    if (cdef_count <= 0) return 0;
    const int bw = kBlkW[bsize & 3], bh = kBlkH[bsize & 3];
    svthip::HostCall& c = svthip::host_call();
    c.begin();
    const size_t blk = (size_t)bw * bh * px;
    c.reserve(2 * blk * cdef_count + 4096, 3 * blk * cdef_count + 4096);
    uint8_t* dp = (uint8_t*)c.dalloc(blk * cdef_count);
    uint8_t* ds = (uint8_t*)c.dalloc(blk * cdef_count);
    unsigned long long* o = (unsigned long long*)c.dalloc(8);
    // gather the picture blocks named by dlist into the packed layout of `src`
    uint8_t* hp = (uint8_t*)c.palloc(blk * cdef_count);
    for (int bi = 0; bi < cdef_count; bi++) {
        const int by = dlist[2 * bi], bx = dlist[2 * bi + 1];
        for (int i = 0; i < bh; i++)
            memcpy(hp + bi * blk + (size_t)i * bw * px, (const uint8_t*)dst + ((size_t)(by * bh + i) * dstride + bx * bw) * px, (size_t)bw * px);
    }
    HIP_CHECK(hipMemcpyAsync(dp, hp, blk * cdef_count, hipMemcpyHostToDevice, c.stream));
    c.up(ds, src, blk * cdef_count);
    if (px == 2)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(cdef_dist_kernel<uint16_t>), dim3(1), dim3(64), 0, c.stream, (const uint16_t*)dp, (const uint16_t*)ds, cdef_count, bw, bh,
                           coeff_shift, pli, (int)sub, o);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(cdef_dist_kernel<uint8_t>), dim3(1), dim3(64), 0, c.stream, (const uint8_t*)dp, (const uint8_t*)ds, cdef_count, bw, bh,
                           coeff_shift, pli, (int)sub, o);
    SVT_LAUNCH_CHECK();
    uint64_t r;
    c.down(&r, o, 8);
    return r;
}
uint64_t svt_compute_cdef_dist_16bit_hip(const uint16_t* dst, int32_t dstride, const uint16_t* src, const SvtHipCdefList* dlist, int32_t cdef_count, SvtHipBlockSize bsize,
                                         int32_t coeff_shift, int32_t pli, uint8_t subsampling_factor) {
    return cdef_dist_host(dst, dstride, src, (const uint8_t*)dlist, cdef_count, bsize, coeff_shift, pli, subsampling_factor, 2);
}
uint64_t svt_compute_cdef_dist_8bit_hip(const uint8_t* dst8, int32_t dstride, const uint8_t* src8, const SvtHipCdefList* dlist, int32_t cdef_count, SvtHipBlockSize bsize,
                                        int32_t coeff_shift, int32_t pli, uint8_t subsampling_factor) {
    return cdef_dist_host(dst8, dstride, src8, (const uint8_t*)dlist, cdef_count, bsize, coeff_shift, pli, subsampling_factor, 1);
}
void svt_aom_copy_rect8_8bit_to_16bit_hip(uint16_t* dst, int32_t dstride, const uint8_t* src, int32_t sstride, int32_t v, int32_t h) {
    if (v <= 0 || h <= 0) return;
    svthip::HostCall& c = svthip::host_call();
    c.begin_small();
    const size_t n = (size_t)v * h;
    c.reserve(n * 3 + 4096, n * 3 + 4096);
    uint8_t*  ds = (uint8_t*)c.dalloc(n);
    uint16_t* dd = (uint16_t*)c.dalloc(n * 2);
    c.up2d(ds, h, src, sstride, h, v);
    hipLaunchKernelGGL(copy_rect8_to_16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c.stream, dd, (const uint8_t*)ds, (int)n);
    SVT_LAUNCH_CHECK();
    c.down2d(dst, (size_t)dstride * 2, dd, (size_t)h * 2, (size_t)h * 2, v);
}

} // extern "C"

SVT_HIP_DEFINE_WARM(cdef) // (svt_hip_warmup loads this translation unit's code object at encoder initialisation: svt_hip_common.h)
