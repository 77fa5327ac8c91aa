// lr_core.h -- loop-restoration tile primitives shared by restoration.hip (apply) and lr_search.hip (search): tile staging with the stripe-boundary
// substitution, the Wiener passes and the self-guided A / B tables + 3x3 weighting on a staged LDS tile.  (Moved out of restoration.hip unchanged.)
#pragma once
#include "svt_hip_common.h"
#include "../../include/svtav1_hip.h"

namespace {

constexpr int FILTER_BITS = 7;
constexpr int TW = 64 + 8;          // LDS tile pitch (u16): 64 columns + 3 left + up to 5 right (3 halo + the 8th-tap column)
constexpr int TH = 64 + 6;          // rows
__device__ __forceinline__ int rpot(const int v, const int n) { return (v + ((1 << n) >> 1)) >> n; }
__device__ __forceinline__ uint32_t rpotu(const uint32_t v, const int n) { return (v + ((1u << n) >> 1)) >> n; }
__device__ __forceinline__ int clampi(const int v, const int lo, const int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// svt_aom_eb_sgr_params (restoration.c:85-103)
__device__ constexpr int16_t kSgrR[16][2] = {{2, 1}, {2, 1}, {2, 1}, {2, 1}, {2, 1}, {2, 1}, {2, 1}, {2, 1}, {2, 1}, {2, 1}, {0, 1}, {0, 1}, {0, 1}, {0, 1}, {2, 0}, {2, 0}};
__device__ constexpr int16_t kSgrS[16][2] = {{140, 3236}, {112, 2158}, {93, 1618}, {80, 1438}, {70, 1295}, {58, 1177}, {47, 1079}, {37, 996},
                                             {30, 925},   {25, 863},   {-1, 2589}, {-1, 1618}, {-1, 1177}, {-1, 925},  {56, -1},   {22, -1}};
// svt_aom_eb_x_by_xplus1 = round(256 z / (z + 1)), [0] = 1, [255] = 256; svt_aom_eb_one_by_x = round(4096 / n) (restoration.c:647-667)
__device__ __forceinline__ uint32_t one_by_x(const uint32_t n) { return (4096 + n / 2) / n; }
// the 256 entries as a compile-time table (every workgroup copies it into LDS: one 2-byte load per thread instead of an integer division)
struct XByXplus1 { uint16_t v[256]; };
constexpr XByXplus1 make_x_by_xplus1() {
    XByXplus1 t{};
    for (uint32_t z = 0; z < 256; z++) t.v[z] = (uint16_t)(z == 0 ? 1 : (z >= 255 ? 256 : (256 * z + (z + 1) / 2) / (z + 1)));
    return t;
}
__device__ constexpr XByXplus1 kXByXplus1 = make_x_by_xplus1();

struct TileSrc { // where a processing unit's pixels come from
    const void* data; const void* above; const void* below;
    int stride, bstride, w, h, highbd;     // plane size (frame mode) or unbounded (raw mode: w = h = 0)
    int x0, y0, uw, uh;                    // unit origin and size inside the plane
    int stripe_top, stripe_bot, stripe_idx;
};
__device__ __forceinline__ int rd_px(const void* p, const int highbd, const size_t off) {
    return highbd ? ((const uint16_t*)p)[off] : ((const uint8_t*)p)[off];
}
// Source row of plane row y for this unit (restoration.c:288-332 stripe boundary substitution + svt_extend_frame row replication); the
// column is clamped by the caller.  Raw mode (w == 0): the block's own rows.  Written as selects on (base address, row index, stride): a
// three-way choice between the pointers of the struct made the compiler index the struct through scratch memory.
__device__ __forceinline__ const uint8_t* src_row(const TileSrc& s, const int y) {
    // raw mode has stripe_top = stripe_bot = h = 0, so `up` and `dn` are false there and only the clamp needs the flag
    const bool raw = s.w == 0;
    const bool up = y < s.stripe_top && s.stripe_top != 0, dn = y >= s.stripe_bot && s.stripe_bot < s.h;
    const int  lo = raw ? INT32_MIN : 0, hi = raw ? INT32_MAX : s.h - 1;                       // (wave-uniform)
    const int  k2 = 2 * s.stripe_idx, ku = k2 - s.stripe_top + 2, kd = k2 - s.stripe_bot;     // (wave-uniform)
    const int  ru = y + ku > k2 ? y + ku : k2;            // 2 idx + max(y - stripe_top + 2, 0)
    const int  rd = y + kd < k2 + 1 ? y + kd : k2 + 1;    // 2 idx + min(y - stripe_bot, 1)
    const int  rm = clampi(y, lo, hi);
    const int  row = up ? ru : (dn ? rd : rm);
    // (bit masks, not ?: on the struct's fields: a select between loads of struct members is rewritten into an indexed load of the struct, which then lives in scratch)
    const uintptr_t d = (uintptr_t)s.data, mu = (uintptr_t)0 - (uintptr_t)up, md = (uintptr_t)0 - (uintptr_t)dn;
    const uintptr_t base = d ^ ((d ^ (uintptr_t)s.above) & mu) ^ ((d ^ (uintptr_t)s.below) & md);
    const int       sb = s.highbd ? 2 * s.stride : s.stride, bb = s.highbd ? 2 * s.bstride : s.bstride; // bytes (wave-uniform)
    const int       stride = sb ^ ((sb ^ bb) & (int)(uint32_t)(mu | md));
    return (const uint8_t*)(base + (uintptr_t)((long long)row * (long long)stride));
}
struct __attribute__((packed, aligned(2))) LrRow8A2 { uint32_t v[4]; };
struct __attribute__((packed, aligned(1))) LrRow8A1 { uint32_t v[2]; };
struct __attribute__((aligned(16))) LrRow8A16 { uint32_t v[4]; };
// tile[(r) * TW + c] <- pixel (y0 - 3 + r, x0 - 3 + c), r < uh + 6, c < uw + 6, zero beyond (the extra columns feed tap 7, always x 0).
// A row is nine 8-pixel chunks; the sixteen lanes of a row group own chunk (lane & 15) < 9 of row (tid >> 4) + 16 k, so the (row, chunk) of a
// thread costs no division and the row pointer is computed once per k.  Chunks that lie inside the plane are fetched with one vector load each, all
// issued before the first LDS store (one memory round trip per workgroup); only chunks that cross the plane's left / right edge or the unit's last
// column go pixel by pixel.  NIT * 16 >= rows.  (A wave-uniform row loop -- src_row on the scalar unit, one load per row and wave -- was tried and was
// twice as slow: 1 365 scalar instructions per wave and serialised loads, profiles/r02_call7_lr_row_uniform_staging.txt.)
template <int NIT> struct StageRegs { uint32_t v[NIT][4]; };
// stage_load issues the global loads of a tile into registers (nothing waits on them), stage_commit writes them to LDS (and fetches the few edge
// chunks pixel by pixel); the frame kernel loads the NEXT half-stripe's tile between the two so that its memory latency hides behind the filter.
template <int NIT>
__device__ __forceinline__ void stage_load(StageRegs<NIT>& R, const TileSrc& s, const int tid) {
    uint32_t (&v)[NIT][4] = R.v;
    const int  rows = s.uh + 6, cols = s.uw + 6;
    const int  c = tid & 15, rb = tid >> 4;
    const int  x = s.x0 - 3 + 8 * c;
    // a chunk that only STARTS inside the needed columns is still fetched whole when its 8 pixels exist (inside the plane / the raw block's border): the
    // columns beyond uw + 6 are multiplied by tap 7 = 0 (Wiener) or never read (self-guided), so they need not be zero.  Before, chunk 8 (6 of 8 pixels
    // needed) took the pixel-by-pixel path in EVERY workgroup: ~180 VALU instructions per wave and a second dependent memory round trip.
    const bool cfast = c < 9 && 8 * c < cols && (s.w == 0 || (x >= 0 && x + 8 <= s.w));
    // Rows that are neither substituted (stripe boundary) nor replicated (plane edge) sit at data + y * stride: one 64-bit multiply-add per thread, then
    // a wave-uniform increment per k.  Only the waves that hold one of the <= 6 special rows take the general src_row (40 VALU instructions; before, every
    // wave paid it for every k -- a third of the Wiener kernel's instructions).
    const int      px = s.highbd ? 2 : 1, sbytes = s.stride * px;
    const bool     raw = s.w == 0;
    const int      ilo = raw ? INT32_MIN : (s.stripe_top != 0 ? s.stripe_top : 0), ihi = raw ? INT32_MAX : (s.stripe_bot < s.h ? s.stripe_bot : s.h);
    const uint8_t* pint = (const uint8_t*)s.data + (long long)(s.y0 - 3 + rb) * sbytes + (cfast ? (long long)x * px : 0ll);
#pragma unroll
    for (int k = 0; k < NIT; k++) {
        const int  r = rb + 16 * k, y = s.y0 - 3 + r;
        const uint8_t* p = pint + (long long)(16 * k) * sbytes;
        if (r >= rows) p = (const uint8_t*)s.data + (long long)s.y0 * sbytes;        // idle lanes read the start of the unit's first row: always mapped (also when
                                                                                     // `data` is the virtual base of a band of rows: csrc/partition.hip)
        else if (y < ilo || y >= ihi) p = src_row(s, y) + (cfast ? (long long)x * px : 0ll); // (slow lanes read the row start)
        if (s.highbd) {
            const svt_u32x4_a2 t = svt_hip_global_load_x4(p);
            v[k][0] = t[0]; v[k][1] = t[1]; v[k][2] = t[2]; v[k][3] = t[3];
        } else {
            const svt_u32x2_a1 t = svt_hip_global_load_x2(p);
            v[k][0] = __builtin_amdgcn_perm(0u, t[0], 0x0c010c00u); v[k][1] = __builtin_amdgcn_perm(0u, t[0], 0x0c030c02u);
            v[k][2] = __builtin_amdgcn_perm(0u, t[1], 0x0c010c00u); v[k][3] = __builtin_amdgcn_perm(0u, t[1], 0x0c030c02u);
        }
    }
}
template <int NIT>
__device__ __forceinline__ void stage_commit(uint16_t* tile, StageRegs<NIT>& R, const TileSrc& s, const int tid) {
    uint32_t (&v)[NIT][4] = R.v;
    const int  rows = s.uh + 6, cols = s.uw + 6;
    const int  c = tid & 15, rb = tid >> 4;
    const int  x = s.x0 - 3 + 8 * c;
    const bool cfast = c < 9 && 8 * c < cols && (s.w == 0 || (x >= 0 && x + 8 <= s.w));
#pragma unroll
    for (int k = 0; k < NIT; k++) {
        const int r = rb + 16 * k;
        if (r >= rows || c >= 9) continue;
        if (!cfast) {
            const uint8_t* row = src_row(s, s.y0 - 3 + r);
#pragma unroll
            for (int e = 0; e < 4; e++) {
                uint32_t px[2];
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int cc = 8 * c + 2 * e + h;
                    int       xx = x + 2 * e + h;
                    if (s.w != 0) xx = clampi(xx, 0, s.w - 1);
                    px[h] = cc < cols ? (uint32_t)rd_px(row, s.highbd, (size_t)0 + (long long)xx) : 0u;
                }
                v[k][e] = px[0] | (px[1] << 16);
            }
        }
        *(LrRow8A16*)(tile + r * TW + 8 * c) = LrRow8A16{{v[k][0], v[k][1], v[k][2], v[k][3]}};
    }
}
template <int NIT>
__device__ __forceinline__ void stage_tile(uint16_t* tile, const TileSrc& s, const int tid) {
    StageRegs<NIT> R;
    stage_load<NIT>(R, s, tid);
    stage_commit<NIT>(tile, R, s, tid);
}
struct WienerTaps { int16_t fx[8], fy[8]; };
typedef short lr_s2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int lr_sdot2(const uint32_t a, const uint32_t b, const int c) { // c + a.lo * b.lo + a.hi * b.hi, signed 16-bit (v_dot2_i32_i16)
    lr_s2 x, y;
    __builtin_memcpy(&x, &a, 4);
    __builtin_memcpy(&y, &b, 4);
    return __builtin_amdgcn_sdot2(x, y, c, false);
}
__device__ __forceinline__ uint32_t lr_pack(const int lo, const int hi) { return ((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16); }
struct __attribute__((aligned(4))) LrDw5 { uint32_t d[5]; };
struct __attribute__((aligned(4))) LrDw2 { uint32_t lo, hi; };
struct __attribute__((aligned(8))) LrDw2A8 { uint32_t lo, hi; };
// Wiener on a staged tile: horizontal pass (clamped, convolve.c:63-83 / :156-176) into `mid`, vertical pass to `out(y, x)`.  A thread produces a
// 2 x 2 block of samples per step, everything on packed pairs (v_dot2_i32_i16).  Horizontal: two tile rows, five aligned dwords each, the odd column on
// the same pairs with the taps shifted by one; the four results are stored as ONE b64 with VERTICALLY adjacent rows paired in a dword -- mid2[row pair][column] = (row 2p | row
// 2p + 1 << 16).  Vertical: output rows 2q and 2q + 1 both read row pairs q .. q + 3 (four b64 reads for two columns); the even row takes the taps paired
// (g0,g1)(g2,g3)(g4,g5)(g6,0), the odd row (0,g0)(g1,g2)(g3,g4)(g5,g6), so no regrouping instruction is needed: 16 dot products per four output samples
// (the row-major `mid` of before cost 16 + 12 v_perm and seven LDS reads per TWO samples; 531 -> see profiles/ VALU instructions per wave).  The
// reference's "+ (centre << FILTER_BITS)" is tap 3 plus 128; the rounding constants ride in the accumulator seeds.  Every operand fits int16: pixels
// <= 4095, mid <= 2^15 - 1 (WIENER_CLAMP_LIMIT), taps < 2^8.  A row past the staged ones (uh odd) only ever meets a zero tap or an output row >= uh.
struct LrNoHook { __device__ __forceinline__ void operator()() const {} };
template <typename OUT, typename HOOK = LrNoHook> __device__ __forceinline__ void wiener_tile(const uint16_t* tile, uint16_t* mid, const WienerTaps& t, const int uw, const int uh,
                                                                    const int bd, const int tid, OUT out, HOOK between_passes = HOOK()) { // (between_passes: run by every thread ahead of the barrier that separates the passes)
    int r0 = 3, r1 = 2 * FILTER_BITS - 3; // get_conv_params_wiener, convolve.h:70-88
    const int range = bd + FILTER_BITS - r0 + 2;
    if (range > 16) { r0 += range - 16; r1 -= range - 16; }
    const int lim = (1 << (bd + 1 + FILTER_BITS - r0)) - 1;
    uint32_t* mid2 = (uint32_t*)mid;
    const int      f3 = t.fx[3] + (1 << FILTER_BITS);
    const uint32_t f01 = lr_pack(t.fx[0], t.fx[1]), f23 = lr_pack(t.fx[2], f3), f45 = lr_pack(t.fx[4], t.fx[5]), f67 = lr_pack(t.fx[6], t.fx[7]);
    const uint32_t fz0 = lr_pack(0, t.fx[0]), f12 = lr_pack(t.fx[1], t.fx[2]), f34 = lr_pack(f3, t.fx[4]), f56 = lr_pack(t.fx[5], t.fx[6]), f7z = lr_pack(t.fx[7], 0);
    const int      oh = (1 << (bd + FILTER_BITS - 1)) + ((1 << r0) >> 1);
    const int      nrp = (uh + 7) >> 1; // row pairs of the uh + 6 staged rows
    for (int i = tid; i < nrp * 32; i += 256) {
        const int rp = i >> 5, c = (i & 31) * 2;
        if (c < uw) {
            uint32_t pk[2][2];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const LrDw5 v = *(const LrDw5*)(tile + (2 * rp + h) * TW + c); // pixels c .. c + 9; pixel c + 3 is the centre of output c
                const int   s0 = lr_sdot2(v.d[0], f01, lr_sdot2(v.d[1], f23, lr_sdot2(v.d[2], f45, lr_sdot2(v.d[3], f67, oh))));
                // output c + 1 on the same aligned pairs with the taps shifted by one (as the vertical pass does for odd rows)
                const int   s1 = lr_sdot2(v.d[0], fz0, lr_sdot2(v.d[1], f12, lr_sdot2(v.d[2], f34, lr_sdot2(v.d[3], f56, lr_sdot2(v.d[4], f7z, oh)))));
                pk[h][0] = (uint32_t)clampi(s0 >> r0, 0, lim); pk[h][1] = (uint32_t)clampi(s1 >> r0, 0, lim);
            }
            *(LrDw2A8*)(mid2 + rp * 64 + c) = LrDw2A8{pk[0][0] | (pk[1][0] << 16), pk[0][1] | (pk[1][1] << 16)};
        }
    }
    between_passes();
    __syncthreads();
    const int      g3 = t.fy[3] + (1 << FILTER_BITS);
    const uint32_t g01 = lr_pack(t.fy[0], t.fy[1]), g23 = lr_pack(t.fy[2], g3), g45 = lr_pack(t.fy[4], t.fy[5]), g6z = lr_pack(t.fy[6], 0);
    const uint32_t gz0 = lr_pack(0, t.fy[0]), g12 = lr_pack(t.fy[1], t.fy[2]), g34 = lr_pack(g3, t.fy[4]), g56 = lr_pack(t.fy[5], t.fy[6]);
    const int      ov = ((1 << r1) >> 1) - (1 << (bd + r1 - 1)), pmax = (1 << bd) - 1;
    for (int i = tid; i < ((uh + 1) >> 1) * 32; i += 256) {
        const int q = i >> 5, c = (i & 31) * 2;
        if (c < uw) {
            const uint32_t* p = mid2 + q * 64 + c; // row pairs q .. q + 3 <-> rows 2q .. 2q + 7 <-> y - 3 .. y + 4 of output row 2q
            const LrDw2A8 d0 = *(const LrDw2A8*)p, d1 = *(const LrDw2A8*)(p + 64), d2 = *(const LrDw2A8*)(p + 128), d3 = *(const LrDw2A8*)(p + 192);
            const int a0 = lr_sdot2(d0.lo, g01, lr_sdot2(d1.lo, g23, lr_sdot2(d2.lo, g45, lr_sdot2(d3.lo, g6z, ov))));
            const int a1 = lr_sdot2(d0.hi, g01, lr_sdot2(d1.hi, g23, lr_sdot2(d2.hi, g45, lr_sdot2(d3.hi, g6z, ov))));
            out(2 * q, c, clampi(a0 >> r1, 0, pmax), clampi(a1 >> r1, 0, pmax), c + 1 < uw);
            if (2 * q + 1 < uh) {
                const int b0 = lr_sdot2(d0.lo, gz0, lr_sdot2(d1.lo, g12, lr_sdot2(d2.lo, g34, lr_sdot2(d3.lo, g56, ov))));
                const int b1 = lr_sdot2(d0.hi, gz0, lr_sdot2(d1.hi, g12, lr_sdot2(d2.hi, g34, lr_sdot2(d3.hi, g56, ov))));
                out(2 * q + 1, c, clampi(b0 >> r1, 0, pmax), clampi(b1 >> r1, 0, pmax), c + 1 < uw);
            }
        }
    }
}

typedef unsigned short lr_us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t lr_dot2(const uint32_t a, const uint32_t b, const uint32_t c) { // c + a.lo * b.lo + a.hi * b.hi (v_dot2_u32_u16)
    lr_us2 x, y;
    __builtin_memcpy(&x, &a, 4);
    __builtin_memcpy(&y, &b, 4);
    return __builtin_amdgcn_udot2(x, y, c, false);
}
struct __attribute__((aligned(4))) LrDw3 { uint32_t d0, d1, d2; };
// Box sums for TWO horizontally adjacent A/B positions per thread: the union of their (2r+1)-wide windows is six pixels = three aligned
// dwords of a tile row, so a row costs two LDS reads and a handful of v_dot2_u32_u16 (sum: multiply by {1,1}; sum of squares: by itself)
// instead of 2 x (2r+1) scalar reads.  xlut = svt_aom_eb_x_by_xplus1 built once per workgroup (the table entries need an integer division).
__device__ __forceinline__ void sgr_ab_pass(const uint16_t* tile, uint16_t* A16, int32_t* B32, const uint16_t* xlut, const int pass, const int idx, const int uw,
                                            const int uh, const int bd, const int tid) {
    const int      r = kSgrR[idx][pass];
    const uint32_t s = (uint32_t)kSgrS[idx][pass], n = (uint32_t)((2 * r + 1) * (2 * r + 1)), obx = one_by_x(n);
    const int      nrows = pass == 0 ? 33 : 66; // pass 0 only needs the positions with ii even (i = ii - 1 odd)
    for (int e = tid; e < nrows * 33; e += 256) {
        const int rr = (e * 1986) >> 16, jj = (e - rr * 33) * 2, ii = pass == 0 ? 2 * rr : rr; // rr = e / 33 exactly for e < 2178; i = ii - 1, j = jj - 1 (and jj)
        if (ii > uh + 1 || jj > uw + 1) continue;
        const uint16_t* p = tile + (ii + 2 - r) * TW + jj; // first of the six pixels jj .. jj + 5 of the window's top row (dword aligned)
        uint32_t tot = 0, tot2 = 0, ea = 0, ea2 = 0, eb = 0, eb2 = 0; // r = 2: totals over six pixels minus an edge; r = 1: centre pair plus an edge
        for (int dy = 0; dy <= 2 * r; dy++) {
            const LrDw3 v = *(const LrDw3*)(p + dy * TW);
            if (r == 2) {
                tot  = lr_dot2(v.d0, 0x00010001u, lr_dot2(v.d1, 0x00010001u, lr_dot2(v.d2, 0x00010001u, tot)));
                tot2 = lr_dot2(v.d0, v.d0, lr_dot2(v.d1, v.d1, lr_dot2(v.d2, v.d2, tot2)));
                ea   = lr_dot2(v.d2, 0x00010000u, ea);            // pixel 5: not in the left window
                ea2  = lr_dot2(v.d2 & 0xffff0000u, v.d2, ea2);
                eb   = lr_dot2(v.d0, 0x00000001u, eb);            // pixel 0: not in the right window
                eb2  = lr_dot2(v.d0 & 0x0000ffffu, v.d0, eb2);
            } else {
                tot  = lr_dot2(v.d1, 0x00010001u, tot);           // pixels 2, 3: in both windows
                tot2 = lr_dot2(v.d1, v.d1, tot2);
                ea   = lr_dot2(v.d0, 0x00010000u, ea);            // pixel 1: left window only
                ea2  = lr_dot2(v.d0 & 0xffff0000u, v.d0, ea2);
                eb   = lr_dot2(v.d2, 0x00000001u, eb);            // pixel 4: right window only
                eb2  = lr_dot2(v.d2 & 0x0000ffffu, v.d2, eb2);
            }
        }
        uint32_t av2[2], bv2[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint32_t sum = r == 2 ? tot - (h ? eb : ea) : tot + (h ? eb : ea);
            const uint32_t sq  = r == 2 ? tot2 - (h ? eb2 : ea2) : tot2 + (h ? eb2 : ea2);
            const uint32_t a = rpotu(sq, 2 * (bd - 8)), b = rpotu(sum, bd - 8);
            const uint32_t pp = (a * n < b * b) ? 0 : a * n - b * b;
            const uint32_t z  = rpotu(pp * s, 20);
            av2[h] = xlut[z > 255 ? 255 : z]; // 1 .. 256
            bv2[h] = rpotu((256u - av2[h]) * sum * obx, 12);
        }
        // jj <= 64, so both positions of the pair lie inside the 66-wide tables: one dword / one qword store (ii * 66 + jj is even)
        const int o = ii * 66 + jj;
        *(uint32_t*)(A16 + o) = av2[0] | (av2[1] << 16);
        *(LrDw2A8*)(B32 + o)  = LrDw2A8{bv2[0], bv2[1]};
    }
}
typedef unsigned short lr_u16x2 __attribute__((vector_size(4)));
__device__ __forceinline__ lr_u16x2 lr_as_pk(const uint32_t v) { lr_u16x2 r; __builtin_memcpy(&r, &v, 4); return r; }
__device__ __forceinline__ lr_u16x2 lr_splat(const int v) { const unsigned short h = (unsigned short)v; return lr_u16x2{h, h}; }
struct __attribute__((aligned(8))) LrInt4 { int32_t v[4]; };
// Weighted 3x3 sums of the A / B tables for the output pair (i, j), (i, j + 1), j even (restoration.c:770-800 "fast" r = 2 pass on alternate rows,
// :850-880 r = 1 pass).  A <= 256 and the weights sum to 32, so the A side runs on packed u16 pairs: per table row the two dwords holding columns
// j - 1 .. j + 2 give the "left" pair, the "right" pair and (one funnel shift) the "centre" pair of the two outputs.
__device__ __forceinline__ void sgr_flt_pair(const uint16_t* tile, const uint16_t* A16, const int32_t* B32, const int pass, const int i, const int j, int32_t (&f)[2]) {
    const uint16_t* A = A16 + (i + 1) * 66 + j; // column j - 1 of table row i (dword aligned: j even, 66 even)
    const int32_t*  B = B32 + (i + 1) * 66 + j;
    lr_u16x2 a;
    int32_t  b0, b1, nb;
    auto rowA = [&](const int dr, lr_u16x2& L, lr_u16x2& Cc, lr_u16x2& R) {
        const LrDw2 v = *(const LrDw2*)(A + dr * 66);
        L = lr_as_pk(v.lo); R = lr_as_pk(v.hi); Cc = lr_as_pk(__builtin_amdgcn_alignbyte(v.hi, v.lo, 2));
    };
    if (pass == 0 && (i & 1)) {
        nb = 4;
        lr_u16x2 L, Cc, R;
        rowA(0, L, Cc, R);
        a = Cc * lr_splat(6) + (L + R) * lr_splat(5);
        const LrInt4 q = *(const LrInt4*)B;
        b0 = q.v[1] * 6 + (q.v[0] + q.v[2]) * 5;
        b1 = q.v[2] * 6 + (q.v[1] + q.v[3]) * 5;
    } else {
        nb = 5;
        lr_u16x2 Lm, Cm, Rm, Lp, Cp, Rp;
        rowA(-1, Lm, Cm, Rm);
        rowA(1, Lp, Cp, Rp);
        const LrInt4 qm = *(const LrInt4*)(B - 66), qp = *(const LrInt4*)(B + 66);
        if (pass == 0) {
            a  = (Cm + Cp) * lr_splat(6) + (Lm + Rm + Lp + Rp) * lr_splat(5);
            b0 = (qm.v[1] + qp.v[1]) * 6 + (qm.v[0] + qm.v[2] + qp.v[0] + qp.v[2]) * 5;
            b1 = (qm.v[2] + qp.v[2]) * 6 + (qm.v[1] + qm.v[3] + qp.v[1] + qp.v[3]) * 5;
        } else {
            lr_u16x2 L0, C0, R0;
            rowA(0, L0, C0, R0);
            const LrInt4 q0 = *(const LrInt4*)B;
            a  = (C0 + L0 + R0 + Cm + Cp) * lr_splat(4) + (Lm + Rm + Lp + Rp) * lr_splat(3);
            b0 = (q0.v[1] + q0.v[0] + q0.v[2] + qm.v[1] + qp.v[1]) * 4 + (qm.v[0] + qm.v[2] + qp.v[0] + qp.v[2]) * 3;
            b1 = (q0.v[2] + q0.v[1] + q0.v[3] + qm.v[2] + qp.v[2]) * 4 + (qm.v[1] + qm.v[3] + qp.v[1] + qp.v[3]) * 3;
        }
    }
    const uint16_t* px = tile + (i + 3) * TW + j + 3;
    f[0] = rpot((int32_t)a[0] * (int32_t)px[0] + b0, 8 + nb - 4);
    f[1] = rpot((int32_t)a[1] * (int32_t)px[1] + b1, 8 + nb - 4);
}

// The r = 2 output of a thread's sixteen pixels (i = tid + 256 k) stays in registers until the r = 1 pass has its A / B tables; MODE_APPLY
// combines with xqd (svt_apply_selfguided_restoration_c :957-992)
template <typename OUT0, typename OUT1>
__device__ __forceinline__ void sgr_tile(const uint16_t* tile, uint16_t* A16, int32_t* B32, uint16_t* xlut, const int idx, const int uw, const int uh,
                                         const int bd, const int tid, OUT0 out_flt0, OUT1 out_flt1_or_apply) {
    const bool p0 = kSgrR[idx][0] > 0, p1 = kSgrR[idx][1] > 0;
    xlut[tid] = kXByXplus1.v[tid]; // 256 threads, 256 entries
    __syncthreads();
    int32_t f0[16]; // eight output pairs per thread: k <-> row 8 k + 4 ((tid >> 5) & 1) + (tid >> 6), columns 2 (tid & 31), + 1
#pragma unroll
    for (int k = 0; k < 16; k++) f0[k] = 0;
    if (p0) {
        sgr_ab_pass(tile, A16, B32, xlut, 0, idx, uw, uh, bd, tid);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int r = 8 * k + ((tid >> 5) & 1) * 4 + (tid >> 6), c = (tid & 31) * 2; // a wave holds rows r and r + 4: same parity, so the
                                                                                         // even-row / odd-row forms of the r = 2 pass do not diverge inside it
            if (r < uh && c < uw) {
                int32_t f[2];
                sgr_flt_pair(tile, A16, B32, 0, r, c, f);
                f0[2 * k] = f[0]; f0[2 * k + 1] = f[1];
                out_flt0(r, c, f[0]);
                if (c + 1 < uw) out_flt0(r, c + 1, f[1]);
            }
        }
        __syncthreads();
    }
    if (p1) sgr_ab_pass(tile, A16, B32, xlut, 1, idx, uw, uh, bd, tid);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int r = 8 * k + ((tid >> 5) & 1) * 4 + (tid >> 6), c = (tid & 31) * 2; // (the same pixel pair as above: f0[] stays in registers)
        if (r < uh && c < uw) {
            int32_t f[2] = {0, 0};
            if (p1) sgr_flt_pair(tile, A16, B32, 1, r, c, f);
            out_flt1_or_apply(r, c, f0[2 * k], f[0], f0[2 * k + 1], f[1], c + 1 < uw);
        }
    }
}
__device__ __forceinline__ int sgr_combine(const int px, const int32_t f0, const int32_t f1, const int idx, const int32_t xqd0, const int32_t xqd1, const int bd) {
    int xq0, xq1; // svt_decode_xq, restoration.c:634-645
    if (kSgrR[idx][0] == 0) { xq0 = 0; xq1 = 128 - xqd1; }
    else if (kSgrR[idx][1] == 0) { xq0 = xqd0; xq1 = 0; }
    else { xq0 = xqd0; xq1 = 128 - xq0 - xqd1; }
    const int32_t u = px << 4;
    int32_t       v = u << 7;
    if (kSgrR[idx][0] > 0) v += xq0 * (f0 - u);
    if (kSgrR[idx][1] > 0) v += xq1 * (f1 - u);
    const int16_t w = (int16_t)rpot(v, 11);
    return clampi(w, 0, (1 << bd) - 1);
}

} // namespace
