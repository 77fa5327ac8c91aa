// sad.hip -- SAD family for gfx950.
//
//   me_fullpel_kernel : frame-batched integer full-pel search (open_loop_me_fullpel_search_sblock,
//                       Source/Lib/Codec/motion_estimation.c:781-816 == a3+a4+a5+a6 fused).  One workgroup per
//                       (64x64 SB, reference, <=64x32 search tile).  Lane = one 8x8 block (Morton order, which IS
//                       the reference's p_best_sad_8x8 numbering), wave = one strip of 4 adjacent x positions.
//                       The 8x8 source block lives in 16 VGPRs for the whole search, the reference window in LDS,
//                       an 8-row register ring slides down the strip so each step costs ONE new 12-byte LDS row
//                       and 16 v_qsad_pk_u16_u8 (4 positions x 64 |a-b| each).  16x16/32x32/64x64 SADs are
//                       DPP quad / row-rotate / cross-row sums of the packed u16 lanes; "first minimum in raster
//                       order" is a single v_min_u32 on (sad << 11 | tile_raster_index).
//   sad_nxm_kernel    : n independent WxH SADs, one wave per pair, 16-byte loads (HBM bound).
//   sad_loop_kernel   : generic exhaustive search (svt_sad_loop_kernel, HME levels), 4 positions per lane.
//   ext_*             : single-call forms of the reference's ext_* pointers.
#include "svt_hip_common.h"
#include "../../include/svtav1_hip.h"
#include "hme_geom.h"
#include <vector>

namespace {

struct __attribute__((packed, aligned(4))) U64A4 { unsigned long long v; }; // 8-byte LDS window, dword aligned
struct __attribute__((aligned(4))) u32x4_a4 { uint32_t x, y, z, w; }; // 16-byte load that only promises dword alignment
struct __attribute__((aligned(16))) u32x4_a16 { uint32_t x, y, z, w; };

constexpr uint32_t MAX_SAD_VALUE = 128 * 128 * 255; // motion_estimation.h:85
constexpr int      ME_TW         = 64;              // search tile, positions
constexpr int      ME_TH         = 32;
constexpr int      KEY_POS_BITS  = 11; // tile raster index: yl * 64 + xl < 2048
constexpr uint32_t ME_WAVE_MAX_W = 24, ME_WAVE_MAX_H = 16; // areas up to this size take the one-wave-per-item kernel

// ---- helpers -------------------------------------------------------------------------------------------------
// Copy rows x width bytes (arbitrary global alignment / stride) into LDS rows of pitch_dw dwords, zero padded.
__device__ __forceinline__ void stage_rows_u8(uint32_t* lds, int pitch_dw, const uint8_t* g, uint32_t gstride, int width,
                                              int rows, int tid, int nthreads) {
    const int total = rows * pitch_dw;
    for (int i = tid; i < total; i += nthreads) {
        const int r   = i / pitch_dw;
        const int k   = i - r * pitch_dw;
        const int col = k * 4;
        uint32_t  v   = 0;
        if (col < width) {
            const uint8_t*  p    = g + (size_t)r * gstride + col;
            const uint32_t  sh   = (uint32_t)((uintptr_t)p & 3);
            const uint32_t* ap   = (const uint32_t*)(p - sh);
            const int       need = (width - col) < 4 ? (width - col) : 4;
            const uint32_t  lo   = ap[0];
            uint32_t        hi   = 0;
            if ((int)sh + need > 4) hi = ap[1];
            v = __builtin_amdgcn_alignbyte(hi, lo, sh);
            if (need < 4) v &= (1u << (8 * need)) - 1u;
        }
        lds[i] = v;
    }
}

struct __attribute__((packed, aligned(1))) u32x4_a1 { uint32_t x, y, z, w; }; // 16-byte load at any byte address (gfx950 global loads are unaligned-capable)
struct __attribute__((packed, aligned(1))) u32_a1 { uint32_t x; };
struct __attribute__((aligned(8))) u32x2_a8 { uint32_t x, y; };

// Same contract as stage_rows_u8 for rows of at most 128 bytes, built for latency: a row is cut into eight 16-byte chunks, thread t owns
// chunks t, t+256, ... and issues ALL of its global loads (one unaligned dwordx4 each) before the first LDS store, so a block pays one
// memory round trip instead of one per loop iteration. Only the last, partial chunk of a row takes the dword-exact path (it must not
// read past the end of the row: the caller owns nothing beyond it). pitch_dw must be even (8-byte LDS stores).
template <int NIT, int CL2> // CL2 = log2(chunks per row): 3 for rows up to 128 bytes, 2 for 64-byte rows; width >= 16
__device__ __forceinline__ void stage_rows_load(uint32_t (&v)[NIT][4], const uint8_t* g, uint32_t gstride, int width, int rows, int tid) {
    // issue phase: straight-line, one dwordx4 per chunk. A partial last chunk (left < 16 bytes) is fetched as the 16 bytes that END at
    // the row end and shifted down afterwards; dead chunks re-read byte 0 and are zeroed. Nothing outside [row, row + width) is touched.
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const int  idx = tid + 256 * k, r = idx >> CL2, col = (idx & ((1 << CL2) - 1)) * 16;
        const bool live = r < rows && col < width;
        const int  left = width - col;
        const int  back = (live && left < 16) ? 16 - left : 0;
        const uint8_t* p = g + (live ? (size_t)r * gstride + (size_t)(col - back) : (size_t)0);
        const u32x4_a1 t = *(const u32x4_a1*)p;
        v[k][0] = t.x; v[k][1] = t.y; v[k][2] = t.z; v[k][3] = t.w;
    }
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const int  idx = tid + 256 * k, r = idx >> CL2, col = (idx & ((1 << CL2) - 1)) * 16;
        const bool live = r < rows && col < width;
        const int  left = width - col;
        uint32_t   x0 = live ? v[k][0] : 0u, x1 = live ? v[k][1] : 0u, x2 = live ? v[k][2] : 0u, x3 = live ? v[k][3] : 0u;
        if (live && left < 16) { // shift the 128-bit value right by 16 - left bytes, zero fill
            const int back = 16 - left;
            if (back & 4) { x0 = x1; x1 = x2; x2 = x3; x3 = 0; }
            if (back & 8) { x0 = x2; x1 = x3; x2 = 0; x3 = 0; }
            const uint32_t bs = (uint32_t)back & 3u;
            x0 = __builtin_amdgcn_alignbyte(x1, x0, bs);
            x1 = __builtin_amdgcn_alignbyte(x2, x1, bs);
            x2 = __builtin_amdgcn_alignbyte(x3, x2, bs);
            x3 = __builtin_amdgcn_alignbyte(0u, x3, bs);
        }
        v[k][0] = x0; v[k][1] = x1; v[k][2] = x2; v[k][3] = x3;
    }
}
template <int NIT, int CL2>
__device__ __forceinline__ void stage_rows_store(const uint32_t (&v)[NIT][4], uint32_t* lds, int pitch_dw, int rows, int tid) {
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const int idx = tid + 256 * k, r = idx >> CL2, c4 = (idx & ((1 << CL2) - 1)) * 4;
        if (r < rows) {
            uint32_t* o = lds + r * pitch_dw + c4;
            if (c4 + 0 < pitch_dw) *(u32x2_a8*)(o + 0) = u32x2_a8{v[k][0], v[k][1]};
            if (c4 + 2 < pitch_dw) *(u32x2_a8*)(o + 2) = u32x2_a8{v[k][2], v[k][3]};
        }
    }
}

// stage_rows_u8 for any number of rows / chunks per row, four 16-byte loads in flight per thread (same tail rule as stage_rows_load:
// nothing outside [row, row + width) is read).  Needs width >= 16 and an even pitch_dw.
// U chunks per thread in flight (the body of stage_rows_wide_any): loads first, LDS stores after
template <int NT, int U>
__device__ __forceinline__ void stage_rows_wide_step(uint32_t* lds, const int pitch_dw, const uint8_t* g, const uint32_t gstride, const int width, const int total,
                                                     const int cpr, const int dr, const int dc, const int base, const int tid, int& r, int& c) {
    uint32_t v[U][4];
    int      rk[U], ck[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {
        const int  idx = base + tid + NT * k, col = c * 16;
        rk[k] = r; ck[k] = c;
        const bool live = idx < total && col < width;
        const int  left = width - col;
        const int  back = (live && left < 16) ? 16 - left : 0;
        const uint8_t* p = g + (live ? (uint32_t)r * gstride + (uint32_t)(col - back) : 0u);
        const u32x4_a1 t = *(const u32x4_a1*)p;
        uint32_t x0 = live ? t.x : 0u, x1 = live ? t.y : 0u, x2 = live ? t.z : 0u, x3 = live ? t.w : 0u;
        if (back) {
            if (back & 4) { x0 = x1; x1 = x2; x2 = x3; x3 = 0; }
            if (back & 8) { x0 = x2; x1 = x3; x2 = 0; x3 = 0; }
            const uint32_t bs = (uint32_t)back & 3u;
            x0 = __builtin_amdgcn_alignbyte(x1, x0, bs);
            x1 = __builtin_amdgcn_alignbyte(x2, x1, bs);
            x2 = __builtin_amdgcn_alignbyte(x3, x2, bs);
            x3 = __builtin_amdgcn_alignbyte(0u, x3, bs);
        }
        v[k][0] = x0; v[k][1] = x1; v[k][2] = x2; v[k][3] = x3;
        r += dr; c += dc;
        if (c >= cpr) { c -= cpr; r++; }
    }
#pragma unroll
    for (int k = 0; k < U; ++k) {
        const int idx = base + tid + NT * k, c4 = ck[k] * 4;
        if (idx < total) {
            uint32_t* o = lds + rk[k] * pitch_dw + c4;
            if (c4 + 0 < pitch_dw) *(u32x2_a8*)(o + 0) = u32x2_a8{v[k][0], v[k][1]};
            if (c4 + 2 < pitch_dw) *(u32x2_a8*)(o + 2) = u32x2_a8{v[k][2], v[k][3]};
        }
    }
}
template <int NT = 256> // cooperating threads (256 = workgroup, 64 = one wave)
__device__ __forceinline__ void stage_rows_wide_any(uint32_t* lds, int pitch_dw, const uint8_t* g, uint32_t gstride, int width, int rows, int tid) {
    const int cpr = (pitch_dw + 3) >> 2, total = rows * cpr; // chunks per LDS row (the last may be partial in LDS too)
    // chunk idx = tid + NT * j -> (row, chunk in row), advanced incrementally: one division per call instead of two per chunk
    const int dr = NT / cpr, dc = NT - dr * cpr;
    int       r = tid / cpr, c = tid - r * cpr;
    // four chunks per thread in flight while that many remain, then two / one: the small windows of the HME levels do not pay for idle chunk slots
    int base = 0;
    for (; total - base > 2 * NT; base += 4 * NT) stage_rows_wide_step<NT, 4>(lds, pitch_dw, g, gstride, width, total, cpr, dr, dc, base, tid, r, c);
    if (total - base > NT) stage_rows_wide_step<NT, 2>(lds, pitch_dw, g, gstride, width, total, cpr, dr, dc, base, tid, r, c);
    else if (total - base > 0) stage_rows_wide_step<NT, 1>(lds, pitch_dw, g, gstride, width, total, cpr, dr, dc, base, tid, r, c);
}

// the same with the chunks per row given by the caller (= ceil(width / 16) <= (pitch_dw + 3) / 4): no dead chunks for windows narrower than the LDS pitch
template <int NT>
__device__ __forceinline__ void stage_rows_wide_cpr(uint32_t* lds, int pitch_dw, const uint8_t* g, uint32_t gstride, int width, int rows, int cpr, int tid) {
    const int total = rows * cpr, dr = NT / cpr, dc = NT - dr * cpr;
    int       r = tid / cpr, c = tid - r * cpr, base = 0;
    for (; total - base > 2 * NT; base += 4 * NT) stage_rows_wide_step<NT, 4>(lds, pitch_dw, g, gstride, width, total, cpr, dr, dc, base, tid, r, c);
    if (total - base > NT) stage_rows_wide_step<NT, 2>(lds, pitch_dw, g, gstride, width, total, cpr, dr, dc, base, tid, r, c);
    else if (total - base > 0) stage_rows_wide_step<NT, 1>(lds, pitch_dw, g, gstride, width, total, cpr, dr, dc, base, tid, r, c);
}

__device__ __forceinline__ uint32_t dpp_add_quad_xor1(uint32_t v) {
    return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false);
}
__device__ __forceinline__ uint32_t dpp_add_quad_xor2(uint32_t v) {
    return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false);
}
__device__ __forceinline__ uint32_t dpp_add_row_ror4(uint32_t v) {
    return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xf, 0xf, false);
}
__device__ __forceinline__ uint32_t dpp_add_row_ror8(uint32_t v) {
    return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false);
}
__device__ __forceinline__ uint32_t umin32(uint32_t a, uint32_t b) { return a < b ? a : b; }

// XCD-aware item order: hardware places workgroup b on XCD b % 8; give each XCD a contiguous run of items so
// neighbouring superblocks (whose reference windows overlap) share one L2.

// ---- frame-batched integer full-pel ME search ---------------------------------------------------------------
constexpr int ME_PITCH = 34; // window row pitch in dwords: >= 17 + 64 / 4, even (8-byte LDS stores) and = 2 mod 8, so the 8 block rows x
                             // 8 block columns a half-wave reads fall into 64 distinct LDS banks (8 * 34 = 16 mod 64)

// All strips of one wave. keys: 8x8 -> (sad16 << 16) | pos   (one v_lshl_or / v_and_or per position, straight from the packed u16 lanes)
//                               16x16/32x32/64x64 -> (sad << 11) | pos   (sad64 < 2^20, pos < 2^11)
// FULL: the tile width is a multiple of 4, no strip has invalid positions.
template <bool SUB, bool FULL, int PITCH = ME_PITCH>
__device__ __forceinline__ void me_search_strips(const uint32_t* __restrict__ win, const uint32_t (&s)[8][2], int Wt, int Ht, int g0, int gstep, int l,
                                                 uint32_t& best8, uint32_t& best16, uint32_t& best32, uint32_t& best64) {
    const int bx = (l & 1) | ((l >> 1) & 2) | ((l >> 2) & 4);
    const int by = ((l >> 1) & 1) | ((l >> 2) & 2) | ((l >> 3) & 4);
    const int q  = l & 3;
    const int      G    = (Wt + 3) >> 2;
    const uint32_t qsel = 0x0c0c0100u + 0x0202u * (uint32_t)q; // v_perm_b32 selector: u16 number q of {thi:tlo}, zero extended
    for (int g = g0; g < G; g += gstep) {
        const uint32_t* colp   = win + (by * 8) * PITCH + bx * 2 + g;
        const int       nvalid = (Wt - 4 * g) < 4 ? (Wt - 4 * g) : 4;
        // invalid positions (last strip when Wt % 4 != 0) are pushed to the top of the key space
        const uint32_t inv1 = nvalid > 1 ? 0u : 0xffffffffu, inv2 = nvalid > 2 ? 0u : 0xffffffffu, inv3 = nvalid > 3 ? 0u : 0xffffffffu;
        const uint32_t invq = q < nvalid ? 0u : 0xffffffffu;
        // 8-row ring; each row is kept as the two overlapping 8-byte windows v_qsad_pk_u16_u8 consumes
        U64A4 ra[8], rb[8];
#pragma unroll
        for (int r = 0; r < 7; r++) {
            ra[r] = *(const U64A4*)(colp + r * PITCH);
            rb[r] = *(const U64A4*)(colp + r * PITCH + 1);
        }
        uint32_t pos = (uint32_t)(4 * g), posq = (uint32_t)(4 * g + q);
        for (int yb = 0; yb < Ht; yb += 8) {
            const uint32_t* rowp = colp + (yb + 7) * PITCH;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (yb + i < Ht) {
                    ra[(i + 7) & 7] = *(const U64A4*)(rowp + i * PITCH);
                    rb[(i + 7) & 7] = *(const U64A4*)(rowp + i * PITCH + 1);
                    unsigned long long acc = 0;
#pragma unroll
                    for (int r = 0; r < 8; r += (SUB ? 2 : 1)) {
                        acc = __builtin_amdgcn_qsad_pk_u16_u8(ra[(i + r) & 7].v, s[r][0], acc);
                        acc = __builtin_amdgcn_qsad_pk_u16_u8(rb[(i + r) & 7].v, s[r][1], acc);
                    }
                    if (SUB) acc <<= 1; // 8x4 on even rows, doubled (motion_estimation.c:105-126); u16 lanes cannot carry
                    const uint32_t lo = (uint32_t)acc, hi = (uint32_t)(acc >> 32);
                    // 8x8: this lane's block, 4 positions
                    uint32_t k0 = (lo << 16) | pos, k1 = (lo & 0xffff0000u) | (pos + 1), k2 = (hi << 16) | (pos + 2),
                             k3 = (hi & 0xffff0000u) | (pos + 3);
                    if (!FULL) { k1 |= inv1; k2 |= inv2; k3 |= inv3; }
                    best8 = umin32(umin32(best8, k0), k1);
                    best8 = umin32(umin32(best8, k2), k3);
                    // 16x16 = the quad's four 8x8 (u16 lanes: 4 * 16320 < 65536, so plain adds never carry)
                    const uint32_t tlo = dpp_add_quad_xor2(dpp_add_quad_xor1(lo));
                    const uint32_t thi = dpp_add_quad_xor2(dpp_add_quad_xor1(hi));
                    // lane q of the quad takes position q from here on
                    const uint32_t sad16 = __builtin_amdgcn_perm(thi, tlo, qsel);
                    const uint32_t pq    = FULL ? posq : (posq | invq);
                    best16 = umin32(best16, (sad16 << KEY_POS_BITS) | pq);
                    // 32x32 = 4 quads of a 16-lane row; 64x64 = 4 rows
                    const uint32_t sad32 = dpp_add_row_ror8(dpp_add_row_ror4(sad16));
                    best32 = umin32(best32, (sad32 << KEY_POS_BITS) | pq);
                    // row sums across the wave with the gfx950 row / half swaps (VALU; a ds_bpermute pair sat in the dependent chain before)
                    const auto     x16   = __builtin_amdgcn_permlane16_swap(sad32, sad32, false, false);
                    const uint32_t pair  = x16[0] + x16[1];
                    const auto     x32   = __builtin_amdgcn_permlane32_swap(pair, pair, false, false);
                    const uint32_t sad64 = x32[0] + x32[1];
                    best64 = umin32(best64, (sad64 << KEY_POS_BITS) | pq);
                    pos += ME_TW;
                    posq += ME_TW;
                }
            }
        }
    }
}

template <bool SUB>
__global__ __launch_bounds__(256, 4) void me_fullpel_kernel(const uint8_t* __restrict__ src_base, const uint8_t* __restrict__ ref_base,
                                                         const SvtHipMeSearchDesc* __restrict__ descs, uint32_t n,
                                                         uint32_t tiles_x,
                                                         uint32_t* __restrict__ best_sad, uint32_t* __restrict__ best_mv,
                                                         unsigned long long* __restrict__ keys) {
    HIP_DYNAMIC_SHARED(uint32_t, smem)
    uint32_t* src_lds  = smem;             // 64 rows x 16 dwords
    uint32_t* best_lds = smem + 64 * 16;   // 85 (+3 pad)
    uint32_t* win      = smem + 64 * 16 + 88;

    const int      tid  = threadIdx.x;
    const uint32_t item = xcd_remap(blockIdx.x, n);
    const uint32_t tile = blockIdx.y;
    const SvtHipMeSearchDesc d = descs[item];
    const int W   = d.search_area_width, H = d.search_area_height;
    const int tx0 = (int)(tile % tiles_x) * ME_TW;
    const int ty0 = (int)(tile / tiles_x) * ME_TH;
    if (tx0 >= W || ty0 >= H) {
        // empty tile; tile 0 of an empty search area still reports "nothing found"
        if (tile == 0 && keys == nullptr && tid < SVT_HIP_ME_NUM_BLOCKS) {
            best_sad[(size_t)item * SVT_HIP_ME_NUM_BLOCKS + tid] = MAX_SAD_VALUE;
            best_mv[(size_t)item * SVT_HIP_ME_NUM_BLOCKS + tid]  = 0;
        }
        return;
    }
    const int Wt = (W - tx0) < ME_TW ? (W - tx0) : ME_TW;
    const int Ht = (H - ty0) < ME_TH ? (H - ty0) : ME_TH;

    {
        uint32_t vs[1][4], vw[3][4]; // 64 x 64 source = 256 chunks; window <= 95 rows x 8 chunks
        stage_rows_load<1, 2>(vs, src_base + d.src_off, d.src_stride, 64, 64, tid);
        stage_rows_load<3, 3>(vw, ref_base + d.ref_off + (size_t)ty0 * d.ref_stride + tx0, d.ref_stride, 64 + Wt - 1, 64 + Ht - 1, tid);
        stage_rows_store<1, 2>(vs, src_lds, 16, 64, tid);
        stage_rows_store<3, 3>(vw, win, ME_PITCH, 64 + Ht - 1, tid);
    }
    if (tid < 88) best_lds[tid] = 0xffffffffu;
    __syncthreads();

    const int l  = tid & 63;
    const int wv = tid >> 6;
    uint32_t s[8][2];
    {
        const int bx = (l & 1) | ((l >> 1) & 2) | ((l >> 2) & 4);
        const int by = ((l >> 1) & 1) | ((l >> 2) & 2) | ((l >> 3) & 4);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            s[r][0] = src_lds[(by * 8 + r) * 16 + bx * 2 + 0];
            s[r][1] = src_lds[(by * 8 + r) * 16 + bx * 2 + 1];
        }
    }
    uint32_t best8 = 0xffffffffu, best16 = 0xffffffffu, best32 = 0xffffffffu, best64 = 0xffffffffu;
    if ((Wt & 3) == 0) me_search_strips<SUB, true>(win, s, Wt, Ht, wv, 4, l, best8, best16, best32, best64);
    else me_search_strips<SUB, false>(win, s, Wt, Ht, wv, 4, l, best8, best16, best32, best64);

    atomicMin(&best_lds[21 + l], best8);
    atomicMin(&best_lds[5 + (l >> 2)], best16);
    atomicMin(&best_lds[1 + (l >> 4)], best32);
    atomicMin(&best_lds[0], best64);
    __syncthreads();
    if (tid < SVT_HIP_ME_NUM_BLOCKS) {
        const uint32_t key = best_lds[tid];
        const uint32_t sad = tid >= 21 ? (key >> 16) : (key >> KEY_POS_BITS);
        const int      X   = tx0 + (int)(key & 63u);
        const int      Y   = ty0 + (int)((key >> 6) & 31u);
        const size_t   o   = (size_t)item * SVT_HIP_ME_NUM_BLOCKS + tid;
        if (keys == nullptr) {
            best_sad[o] = sad;
            best_mv[o]  = ((uint32_t)(uint16_t)(Y + d.y_search_area_origin) << 16) | (uint16_t)(X + d.x_search_area_origin);
        } else {
            atomicMin(&keys[o], ((unsigned long long)sad << 32) | (unsigned long long)(uint32_t)(Y * W + X));
        }
    }
}

// ---- the same search with ONE WAVE per (SB, reference) item, for the small areas of the fast presets (16x9 is the preset-8 maximum at 1080p, 8x4 what
// CRF 35 leaves of it; enc_mode_config.c:311-333).  With four cooperating waves per item the fixed work around the search -- staging the source block and the
// window through LDS, two workgroup barriers, LDS atomics to merge the waves -- was ~230 VALU instructions per wave against 9 search steps (25 % of the kernel
// at 16x9).  Here a wave owns its item: the lane's 8x8 source block comes straight from global memory into its 16 VGPRs (eight 8-byte
// loads), the window goes into the wave's private LDS slice (no workgroup barrier: LDS program order within a wave is enough), the wave walks ALL strips of
// the area, and the 16x16 / 32x32 / 64x64 winners are merged with DPP / row-swap minima instead of LDS atomics.  Same keys, same tie-breaks (first minimum in
// raster order = smallest (sad, position) key), bit-identical tables.
struct __attribute__((packed, aligned(1))) u32x2_a1 { uint32_t x, y; };
__device__ __forceinline__ uint32_t dpp_min_quad_xor1(uint32_t v) { return umin32(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0xB1, 0xf, 0xf, false)); }
__device__ __forceinline__ uint32_t dpp_min_quad_xor2(uint32_t v) { return umin32(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x4E, 0xf, 0xf, false)); }
__device__ __forceinline__ uint32_t dpp_min_row_ror4(uint32_t v) { return umin32(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x124, 0xf, 0xf, false)); }
__device__ __forceinline__ uint32_t dpp_min_row_ror8(uint32_t v) { return umin32(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x128, 0xf, 0xf, false)); }

template <bool SUB, int PITCH>
__global__ __launch_bounds__(256) void me_fullpel_wave_kernel(const uint8_t* __restrict__ src_base, const uint8_t* __restrict__ ref_base,
                                                              const SvtHipMeSearchDesc* __restrict__ descs, const uint32_t n, const int win_dw,
                                                              uint32_t* __restrict__ best_sad, uint32_t* __restrict__ best_mv) {
    HIP_DYNAMIC_SHARED(uint32_t, smem)
    const int      l    = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t item = xcd_remap(blockIdx.x, gridDim.x) * 4 + (uint32_t)wv; // four consecutive items (neighbouring SBs of one reference) per workgroup
    if (item >= n) return;
    const SvtHipMeSearchDesc d = descs[item];
    const int    W = d.search_area_width, H = d.search_area_height;
    const size_t o = (size_t)item * SVT_HIP_ME_NUM_BLOCKS;
    if (W <= 0 || H <= 0) { // an empty search area reports "nothing found"
        for (int k = l; k < SVT_HIP_ME_NUM_BLOCKS; k += 64) { best_sad[o + k] = MAX_SAD_VALUE; best_mv[o + k] = 0; }
        return;
    }
    uint32_t* win = smem + wv * win_dw;
    const int bx = (l & 1) | ((l >> 1) & 2) | ((l >> 2) & 4);
    const int by = ((l >> 1) & 1) | ((l >> 2) & 2) | ((l >> 3) & 4);
    uint32_t  s[8][2];
    {
        const uint8_t* sp = src_base + d.src_off + (size_t)(by * 8) * d.src_stride + bx * 8;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const u32x2_a1 t = *(const u32x2_a1*)(sp + (size_t)r * d.src_stride);
            s[r][0] = t.x; s[r][1] = t.y;
        }
    }
    const int width = 64 + W - 1;
    stage_rows_wide_cpr<64>(win, PITCH, ref_base + d.ref_off, d.ref_stride, width, 64 + H - 1, (width + 15) >> 4, l);
    __builtin_amdgcn_wave_barrier(); // the slice belongs to this wave alone: LDS program order is enough on the hardware

    uint32_t best8 = 0xffffffffu, best16 = 0xffffffffu, best32 = 0xffffffffu, best64 = 0xffffffffu;
    if ((W & 3) == 0) me_search_strips<SUB, true, PITCH>(win, s, W, H, 0, 1, l, best8, best16, best32, best64);
    else me_search_strips<SUB, false, PITCH>(win, s, W, H, 0, 1, l, best8, best16, best32, best64);
    // lane q of a quad tracked position q of every strip: the block's winner is the smallest key of the lanes that share the block
    best16 = dpp_min_quad_xor2(dpp_min_quad_xor1(best16));
    best32 = dpp_min_row_ror8(dpp_min_row_ror4(dpp_min_quad_xor2(dpp_min_quad_xor1(best32))));
    best64 = dpp_min_row_ror8(dpp_min_row_ror4(dpp_min_quad_xor2(dpp_min_quad_xor1(best64))));
    best64 = umin32(best64, (uint32_t)__shfl_xor((int)best64, 16));
    best64 = umin32(best64, (uint32_t)__shfl_xor((int)best64, 32));
    const int xo = d.x_search_area_origin, yo = d.y_search_area_origin;
    auto emit = [&](const int idx, const uint32_t key, const bool is8) {
        const uint32_t sad = is8 ? (key >> 16) : (key >> KEY_POS_BITS);
        const int      X = (int)(key & 63u), Y = (int)((key >> 6) & 31u);
        best_sad[o + idx] = sad;
        best_mv[o + idx]  = ((uint32_t)(uint16_t)(Y + yo) << 16) | (uint16_t)(X + xo);
    };
    emit(21 + l, best8, true);
    if ((l & 3) == 0) emit(5 + (l >> 2), best16, false);
    if ((l & 15) == 0) emit(1 + (l >> 4), best32, false);
    if (l == 0) emit(0, best64, false);
}

__global__ void me_finalize_kernel(const SvtHipMeSearchDesc* __restrict__ descs, uint32_t n, const unsigned long long* __restrict__ keys,
                                   uint32_t* __restrict__ best_sad, uint32_t* __restrict__ best_mv) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * SVT_HIP_ME_NUM_BLOCKS) return;
    const SvtHipMeSearchDesc d   = descs[i / SVT_HIP_ME_NUM_BLOCKS];
    const unsigned long long key = keys[i];
    if (key == ~0ull) {
        best_sad[i] = MAX_SAD_VALUE;
        best_mv[i]  = 0;
        return;
    }
    const uint32_t pos = (uint32_t)key;
    const int      W   = d.search_area_width;
    const int      X = (int)(pos % (uint32_t)W), Y = (int)(pos / (uint32_t)W);
    best_sad[i]        = (uint32_t)(key >> 32);
    best_mv[i]         = ((uint32_t)(uint16_t)(Y + d.y_search_area_origin) << 16) | (uint16_t)(X + d.x_search_area_origin);
}

// ---- n independent WxH SADs: one wave per pair -----------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
    v += (uint32_t)__shfl_xor((int)v, 32);
    v += (uint32_t)__shfl_xor((int)v, 16);
    v = dpp_add_row_ror8(v);
    v = dpp_add_row_ror4(v);
    v = dpp_add_quad_xor2(v);
    v = dpp_add_quad_xor1(v);
    return v;
}

// One wave per pair. A block row is cut into 16-byte chunks (widths >= 16) or 4-byte chunks (widths 4, 8, 12); every lane issues up to four
// chunk loads per operand before the first v_sad_u8 so that a 64x64 pair (8 KB) is fully in flight after one issue burst.
template <int CHUNK> // bytes per lane-chunk: 16 or 4
__device__ __forceinline__ uint32_t sad_chunks(const uint8_t* __restrict__ src, const uint8_t* __restrict__ ref, uint32_t ss, uint32_t rs,
                                               int cpr, int cshift, int total, int l) {
    uint32_t sad = 0;
    for (int i0 = l; i0 < total; i0 += 256) {
        uint32_t av[4][CHUNK / 4], bv[4][CHUNK / 4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int  i  = i0 + 64 * k;
            const bool in = i < total;
            const int  r  = cshift >= 0 ? (i >> cshift) : (i / cpr);
            const int  c  = i - r * cpr;
            const uint8_t* pa = src + (size_t)r * ss + c * CHUNK;
            const uint8_t* pb = ref + (size_t)r * rs + c * CHUNK;
            if (CHUNK == 16) {
                u32x4_a1 a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
                if (in) { a = *(const u32x4_a1*)pa; b = *(const u32x4_a1*)pb; }
                av[k][0] = a.x; av[k][CHUNK / 4 > 1 ? 1 : 0] = a.y; av[k][CHUNK / 4 > 2 ? 2 : 0] = a.z; av[k][CHUNK / 4 > 3 ? 3 : 0] = a.w;
                bv[k][0] = b.x; bv[k][CHUNK / 4 > 1 ? 1 : 0] = b.y; bv[k][CHUNK / 4 > 2 ? 2 : 0] = b.z; bv[k][CHUNK / 4 > 3 ? 3 : 0] = b.w;
            } else {
                u32_a1 a = {0}, b = {0};
                if (in) { a = *(const u32_a1*)pa; b = *(const u32_a1*)pb; }
                av[k][0] = a.x; bv[k][0] = b.x;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int j = 0; j < CHUNK / 4; ++j) sad = __builtin_amdgcn_sad_u8(av[k][j], bv[k][j], sad);
    }
    return sad;
}

__global__ __launch_bounds__(256) void sad_nxm_kernel(const uint8_t* __restrict__ src_base, const uint8_t* __restrict__ ref_base,
                                                      const SvtHipSadPair* __restrict__ pairs, uint32_t n, int width, int height,
                                                      uint32_t* __restrict__ sad_out) {
    const int      l    = threadIdx.x & 63;
    // XCD-aware order: consecutive pairs are usually neighbouring blocks of one picture row (64-byte rows at a 2 KB pitch: two neighbours share every
    // 128-byte line, and an unaligned reference row straddles two), so each XCD takes a contiguous run of pair groups and the shared lines hit in ITS L2
    // instead of being fetched by two XCDs (HBM traffic was 1.59 x the algorithmic bytes, profiles/r02_reg6_pmc_traffic.json)
    const uint32_t pair = xcd_remap(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6);
    if (pair >= n) return;
    const SvtHipSadPair p   = pairs[pair];
    const uint8_t*      src = src_base + p.src_off;
    const uint8_t*      ref = ref_base + p.ref_off;
    uint32_t            sad = 0;
    if ((width & 15) == 0) {
        const int cpr = width >> 4;
        sad = sad_chunks<16>(src, ref, p.src_stride, p.ref_stride, cpr, (cpr & (cpr - 1)) ? -1 : __builtin_ctz(cpr), cpr * height, l);
    } else if ((width & 3) == 0) {
        const int cpr = width >> 2;
        sad = sad_chunks<4>(src, ref, p.src_stride, p.ref_stride, cpr, (cpr & (cpr - 1)) ? -1 : __builtin_ctz(cpr), cpr * height, l);
    } else {
        const int total = width * height;
        for (int i = l; i < total; i += 64) {
            const int r = i / width, c = i - r * width;
            const int a = src[(size_t)r * p.src_stride + c], b = ref[(size_t)r * p.ref_stride + c];
            sad += (uint32_t)(a > b ? a - b : b - a);
        }
    }
    sad = wave_sum(sad);
    if (l == 0) sad_out[pair] = sad;
}

// Pipelined form for blocks of at most 256 16-byte chunks (64x64 .. 16xN): a wave walks SADP_PPW pairs and issues the loads of pair k + 1 before it
// reduces pair k, so the descriptor fetch, the wave start-up and the reduction tail of one pair hide behind the data of the next (the one-pair-per-wave
// form above spent about a third of a wave's life with nothing in flight: 0.57 of the HBM roofline with traffic already at 1.07 x the algorithmic
// bytes).  At step k the four waves of a workgroup hold four consecutive pairs, so neighbouring blocks still meet in one L2.
constexpr int SADP_PPW = 4;
struct SadRegs { uint32_t a[4][4], b[4][4]; };
__device__ __forceinline__ void sadp_issue(SadRegs& r, const uint8_t* __restrict__ src, const uint8_t* __restrict__ ref, const SvtHipSadPair& p, const int cpr,
                                           const int cshift, const int total, const int l) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int  i  = l + 64 * k;
        const bool in = i < total;
        const int  rr = i >> cshift, c = i - (rr << cshift);
        u32x4_a1 a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
        if (in) {
            a = *(const u32x4_a1*)(src + p.src_off + (size_t)rr * p.src_stride + c * 16);
            b = *(const u32x4_a1*)(ref + p.ref_off + (size_t)rr * p.ref_stride + c * 16);
        }
        r.a[k][0] = a.x; r.a[k][1] = a.y; r.a[k][2] = a.z; r.a[k][3] = a.w;
        r.b[k][0] = b.x; r.b[k][1] = b.y; r.b[k][2] = b.z; r.b[k][3] = b.w;
    }
}
__device__ __forceinline__ uint32_t sadp_reduce(const SadRegs& r) {
    uint32_t sad = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) sad = __builtin_amdgcn_sad_u8(r.a[k][j], r.b[k][j], sad);
    return wave_sum(sad);
}
__global__ __launch_bounds__(256) void sad_nxm_pipe_kernel(const uint8_t* __restrict__ src_base, const uint8_t* __restrict__ ref_base,
                                                           const SvtHipSadPair* __restrict__ pairs, uint32_t n, int cshift /* log2(width / 16) */, int total,
                                                           uint32_t* __restrict__ sad_out) {
    const int      l     = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t first = xcd_remap(blockIdx.x, gridDim.x) * (4 * SADP_PPW) + (uint32_t)w; // pairs first, first + 4, first + 8, ...
    if (first >= n) return;
    const int cpr = 1 << cshift;
    SadRegs   cur, nxt;
    sadp_issue(cur, src_base, ref_base, pairs[first], cpr, cshift, total, l);
#pragma unroll
    for (int k = 0; k < SADP_PPW; ++k) {
        const uint32_t pair = first + 4u * (uint32_t)k, np = pair + 4u;
        const bool     more = k + 1 < SADP_PPW && np < n;
        if (more) sadp_issue(nxt, src_base, ref_base, pairs[np], cpr, cshift, total, l);
        const uint32_t sad = sadp_reduce(cur);
        if (l == 0) sad_out[pair] = sad;
        if (!more) break;
        cur = nxt;
    }
}

// Strip form (round 3, an experiment that LOST -- see svt_hip_sad_nxm_batch): the wave's 64 lanes are laid over FOUR ROWS x 256 BYTES -- 16 / cpr blocks side by side (four 64-wide blocks), cpr lanes per block row -- and
// walk down the blocks four rows per step.  With horizontally adjacent blocks (the 510 SBs of a picture row by row: config 1) one wave instruction then reads four
// runs of 256 CONTIGUOUS bytes = 12 cache-line requests, two thirds of them fully used, where the pair-per-wave forms read sixteen separate 64-byte row pieces =
// 32 line requests, every line shared with the neighbouring block's wave (rows start at byte 68 of a 2056-byte pitch: each 64-byte piece straddles two 128-byte
// lines; traffic was 1.35 x the algorithmic bytes, profiles/r03_call1_bench_default.json).  Nothing requires the blocks to be adjacent -- every lane addresses its
// own pair's descriptor -- adjacency only decides how well the requests coalesce.  Loads are pipelined two groups of four steps deep across the strips a wave walks.
constexpr int SADS_SPW = 4; // strips per wave
struct StripRegs { uint32_t a[4][4], b[4][4]; };
__device__ __forceinline__ void strip_issue(StripRegs& r, const uint8_t* __restrict__ sp, const uint8_t* __restrict__ rp, const uint32_t ss, const uint32_t rs,
                                            const int row0, const int ro, const int height, const bool live) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int row = row0 + 4 * k; // relative to the lane's own first row `ro` (already in sp / rp)
        u32x4_a1  a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
        if (live && row + ro < height) {
            a = *(const u32x4_a1*)(sp + (size_t)row * ss);
            b = *(const u32x4_a1*)(rp + (size_t)row * rs);
        }
        r.a[k][0] = a.x; r.a[k][1] = a.y; r.a[k][2] = a.z; r.a[k][3] = a.w;
        r.b[k][0] = b.x; r.b[k][1] = b.y; r.b[k][2] = b.z; r.b[k][3] = b.w;
    }
}
__device__ __forceinline__ uint32_t strip_consume(const StripRegs& r, uint32_t sad) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) sad = __builtin_amdgcn_sad_u8(r.a[k][j], r.b[k][j], sad);
    return sad;
}
__global__ __launch_bounds__(256) void sad_nxm_strip_kernel(const uint8_t* __restrict__ src_base, const uint8_t* __restrict__ ref_base,
                                                            const SvtHipSadPair* __restrict__ pairs, const uint32_t n, const int cshift /* log2(width / 16) */,
                                                            const int height, uint32_t* __restrict__ sad_out) {
    const int      l = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int      cpr = 1 << cshift, nb = 16 >> cshift;            // lanes per block row, blocks per strip
    const int      sb = (l & 15) >> cshift, j = l & (cpr - 1), ro = l >> 4;
    const int      groups = (height + 15) >> 4;                    // groups of four steps (16 rows)
    const uint32_t strip0 = (xcd_remap(blockIdx.x, gridDim.x) * 4 + (uint32_t)w) * SADS_SPW;
    if (strip0 * (uint32_t)nb >= n) return;
    // unit u = (strip k, group g); the loads of unit u + 1 are issued before unit u is reduced
    const uint8_t *sp = nullptr, *rp = nullptr;
    uint32_t       ss = 0, rs = 0;
    bool           live = false;
    auto open_strip = [&](const int k) {
        const uint32_t pair = (strip0 + (uint32_t)k) * (uint32_t)nb + (uint32_t)sb;
        live = k < SADS_SPW && pair < n;
        if (live) {
            const SvtHipSadPair p = pairs[pair];
            sp = src_base + p.src_off + (size_t)ro * p.src_stride + j * 16;
            rp = ref_base + p.ref_off + (size_t)ro * p.ref_stride + j * 16;
            ss = p.src_stride; rs = p.ref_stride;
        }
    };
    StripRegs cur, nxt;
    open_strip(0);
    strip_issue(cur, sp, rp, ss, rs, 0, ro, height, live);
    for (int k = 0; k < SADS_SPW; ++k) {
        if ((strip0 + (uint32_t)k) * (uint32_t)nb >= n) break;
        const bool     mine = live; // this lane's block of strip k exists
        uint32_t       sad  = 0;
        const uint32_t out  = (strip0 + (uint32_t)k) * (uint32_t)nb + (uint32_t)sb;
        for (int g = 0; g < groups; ++g) {
            if (g + 1 < groups) {
                strip_issue(nxt, sp, rp, ss, rs, 16 * (g + 1), ro, height, live);
            } else {
                open_strip(k + 1); // (past the last strip: live = false, nothing is loaded)
                strip_issue(nxt, sp, rp, ss, rs, 0, ro, height, live);
            }
            sad = strip_consume(cur, sad);
            cur = nxt;
        }
        // lanes of one block: cpr lanes of a row (low bits) x four row phases (lanes + 16, + 32)
        for (int m = 1; m < cpr; m <<= 1) sad += (uint32_t)__shfl_xor((int)sad, m);
        sad += (uint32_t)__shfl_xor((int)sad, 16);
        sad += (uint32_t)__shfl_xor((int)sad, 32);
        if (mine && j == 0 && ro == 0) sad_out[out] = sad;
    }
}

__global__ __launch_bounds__(64) void sad_16b_kernel(const uint16_t* __restrict__ src, uint32_t src_stride, const uint16_t* __restrict__ ref,
                                                     uint32_t ref_stride, int width, int height, uint32_t* __restrict__ out) {
    const int l     = threadIdx.x;
    uint32_t  sad   = 0;
    const int total = width * height;
    for (int i = l; i < total; i += 64) {
        const int r = i / width, c = i - r * width;
        const int a = src[(size_t)r * src_stride + c], b = ref[(size_t)r * ref_stride + c];
        sad += (uint32_t)(a > b ? a - b : b - a);
    }
    sad = wave_sum(sad);
    if (l == 0) out[0] = sad;
}

// ---- generic exhaustive search (svt_sad_loop_kernel) ----------------------------------------------------------
// One workgroup = one (item, position tile of (4 << LXG) x (256 >> LXG)): 256 lanes = (1 << LXG) x-groups of 4 positions x (256 >> LXG) search
// rows.  The host picks the narrowest tile that covers the widest search area of the batch (16, 32 or 64 positions wide), so the small HME
// areas keep every lane busy.
template <int LXG>
__device__ __forceinline__ void sad_loop_item(uint32_t* smem, unsigned long long& wg_best, const uint8_t* __restrict__ src_base,
                                              const uint8_t* __restrict__ ref_base, const SvtHipSadLoopDesc& d, const uint32_t item, const uint32_t tile,
                                              const uint32_t tiles_x, unsigned long long* __restrict__ keys) {
    constexpr int TW = 4 << LXG, TH = 256 >> LXG;
    const int tid = threadIdx.x;
    const int W = d.search_area_width, H = d.search_area_height;
    const int bw = d.block_width, bh = d.block_height;
    const int tx0 = (int)(tile % tiles_x) * TW, ty0 = (int)(tile / tiles_x) * TH;
    if (tx0 >= W || ty0 >= H) return;
    const int Wt = (W - tx0) < TW ? (W - tx0) : TW;
    const int Ht = (H - ty0) < TH ? (H - ty0) : TH;
    const int src_pitch = ((bw + 15) >> 4) << 2;        // dwords, rows 16-byte aligned
    const int win_w     = bw + Wt - 1;                   // bytes actually read by the reference
    const int win_pitch = (((bw + TW + 3) >> 2) + 3) & ~1; // dwords, even (covers the 4-position over-read of the last group)
    uint32_t* src_lds   = smem;
    uint32_t* win       = smem + src_pitch * bh;
    // source rows are src_stride apart; search rows advance by src_stride_raw, block rows by ref_stride
    // (compute_sad_c.c:72-97: `ref += src_stride_raw` per search line, `ref[... + y * ref_stride ...]` per block line)
    stage_rows_u8(src_lds, src_pitch, src_base + d.src_off, d.src_stride, bw, bh, tid, 256);
    if (tid == 0) wg_best = ~0ull;
    // window row r of the staged tile = reference line ty0 + r (lines are src_stride_raw apart).  Search line yy, block line y reads line
    // yy + rstep * y with rstep = ref_stride / src_stride_raw: 1 for the full SAD, 2 for the sub-sampled HME form
    // (motion_estimation.c:891-908); the lines a tile needs are the contiguous run 0 .. Ht - 1 + rstep * (bh - 1).
    const int rstep    = (int)(d.ref_stride / d.src_stride_raw);
    const int win_rows = Ht + rstep * (bh - 1);
    const uint8_t* wsrc = ref_base + d.ref_off + (size_t)ty0 * d.src_stride_raw + tx0;
    if (win_w >= 16) stage_rows_wide_any(win, win_pitch, wsrc, d.src_stride_raw, win_w, win_rows, tid);
    else stage_rows_u8(win, win_pitch, wsrc, d.src_stride_raw, win_w, win_rows, tid, 256);
    __syncthreads();

    const int gx = tid & ((1 << LXG) - 1), yy = tid >> LXG;
    const bool skip_rule = (bw == 16) && (bh <= 16) && d.skip_search_line; // compute_sad_c.c:74-79: even lines skipped
    unsigned long long best = ~0ull;
    if (yy < Ht && 4 * gx < Wt && !(skip_rule && (((ty0 + yy) & 1) == 0))) {
        uint32_t sad[4] = {0, 0, 0, 0};
        const int full_dw = bw >> 2, tail = bw & 3;
        for (int y = 0; y < bh; y++) {
            const uint32_t* rrow = win + (yy + rstep * y) * win_pitch + gx;
            const uint32_t* srow = src_lds + y * src_pitch;
            unsigned long long acc = 0;
            uint32_t prev = rrow[0];
            int      k    = 0;
            for (; k + 4 <= full_dw; k += 4) { // 16 source bytes per step: one ds_read_b128 (source, wave-uniform) + two ds_read2_b32 (window)
                const u32x4_a16 sv = *(const u32x4_a16*)(srow + k);
                const U64A4     n0 = *(const U64A4*)(rrow + k + 1), n1 = *(const U64A4*)(rrow + k + 3);
                const uint32_t  r1 = (uint32_t)n0.v, r2 = (uint32_t)(n0.v >> 32), r3 = (uint32_t)n1.v, r4 = (uint32_t)(n1.v >> 32);
                acc  = __builtin_amdgcn_qsad_pk_u16_u8(((unsigned long long)r1 << 32) | prev, sv.x, acc);
                acc  = __builtin_amdgcn_qsad_pk_u16_u8(n0.v, sv.y, acc);
                acc  = __builtin_amdgcn_qsad_pk_u16_u8(((unsigned long long)r3 << 32) | r2, sv.z, acc);
                acc  = __builtin_amdgcn_qsad_pk_u16_u8(n1.v, sv.w, acc);
                prev = r4;
                if ((k & 12) == 12) { // every 64 source bytes: flush before a u16 lane can overflow (64 * 255 = 16320)
                    sad[0] += (uint32_t)acc & 0xffffu; sad[1] += (uint32_t)(acc >> 16) & 0xffffu;
                    sad[2] += (uint32_t)(acc >> 32) & 0xffffu; sad[3] += (uint32_t)(acc >> 48);
                    acc = 0;
                }
            }
            for (; k < full_dw; k++) { // at most 3 dwords (48 further bytes: no overflow before the flush below)
                const uint32_t next = rrow[k + 1];
                acc  = __builtin_amdgcn_qsad_pk_u16_u8(((unsigned long long)next << 32) | prev, srow[k], acc);
                prev = next;
            }
            sad[0] += (uint32_t)acc & 0xffffu; sad[1] += (uint32_t)(acc >> 16) & 0xffffu;
            sad[2] += (uint32_t)(acc >> 32) & 0xffffu; sad[3] += (uint32_t)(acc >> 48);
            if (tail) {
                const uint32_t mask = (1u << (8 * tail)) - 1u;
                const uint32_t next = rrow[full_dw + 1];
                const uint32_t sv   = srow[full_dw] & mask;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t rv = __builtin_amdgcn_alignbyte(next, prev, (uint32_t)j) & mask;
                    sad[j]            = __builtin_amdgcn_sad_u8(rv, sv, sad[j]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int X = tx0 + 4 * gx + j;
            if (4 * gx + j < Wt) {
                const unsigned long long key = ((unsigned long long)sad[j] << 32) | (uint32_t)((ty0 + yy) * W + X);
                best = key < best ? key : best;
            }
        }
    }
    atomicMin(&wg_best, best);
    __syncthreads();
    if (tid == 0 && wg_best != ~0ull) atomicMin(&keys[item], wg_best);
}
template <int LXG>
__global__ __launch_bounds__(256) void sad_loop_kernel(const uint8_t* __restrict__ src_base, const uint8_t* __restrict__ ref_base,
                                                       const SvtHipSadLoopDesc* __restrict__ descs, const uint32_t* __restrict__ todo, const uint32_t tiles_x,
                                                       unsigned long long* __restrict__ keys) {
    HIP_DYNAMIC_SHARED(uint32_t, smem)
    __shared__ unsigned long long wg_best;
    const uint32_t count = todo[0]; // items sad_loop_ring_kernel left for this kernel (usually none: the launch is then a few thousand empty workgroups)
    for (uint32_t i = blockIdx.x; i < count; i += gridDim.x) {
        const uint32_t          item = todo[1 + i];
        const SvtHipSadLoopDesc d    = descs[item];
        __syncthreads(); // the previous item's LDS tile and wg_best are no longer in use
        sad_loop_item<LXG>(smem, wg_best, src_base, ref_base, d, item, blockIdx.y, tiles_x, keys);
    }
}

// ---- svt_sad_loop_kernel, ring form: the HME shapes -------------------------------------------------------------------------------------
// Items whose block is 16x16 / 32x32 / 64x64 (or the sub-sampled 16x8 / 32x16 / 64x32 with ref_stride = 2 * src_stride_raw), whose search area has
// at most 4096 positions and whose window fits SLR_WIN_BYTES of LDS take this path; everything else is appended to a work list for sad_loop_kernel.
//
// One WAVE per item.  The block is cut into 8x8 sub-blocks (8x4 used rows in the sub-sampled form) exactly as me_fullpel_kernel does: lane =
// (x-group of 4 positions, sub-block), the lane keeps its sub-block of the source in registers and slides an 8-row (4-row) register ring down the
// search column, 16 (8) v_qsad_pk_u16_u8 per step; the sub-block SADs meet through DPP quad / row adds (and ds_bpermute for 64x64).  A 16x16 item
// therefore searches 64 positions per step, a 32x32 item 16, a 64x64 item 4.
constexpr int SLR_WIN_BYTES = 10240, SLR_SRC_BYTES = 4096; // upper limits per wave; the launch sizes the slices from the batch maxima
// Blocks shorter than they are wide (the last SB row of a picture: 56 of 64 rows, 14 of 16 at the 1/16 level) also qualify: the sub-block rows
// beyond block_height are skipped under an exec mask (PARTIAL instantiation of the wave routine).
__device__ __forceinline__ bool sad_loop_ring_eligible(const SvtHipSadLoopDesc& d, const int win_budget, const int src_budget) {
    const int bw = d.block_width, bh = d.block_height, W = d.search_area_width, H = d.search_area_height;
    if (!(bw == 16 || bw == 32 || bw == 64) || d.src_stride_raw == 0 || d.ref_stride % d.src_stride_raw != 0) return false;
    const int rstep = (int)(d.ref_stride / d.src_stride_raw);
    if ((rstep != 1 && rstep != 2) || bh <= 0 || bh * rstep > bw || W <= 0 || H <= 0 || W * H > 4096) return false;
    const int pitch = (((bw + W + 3) >> 2) + 3) & ~1, lines = H + rstep * (bh - 1);
    return pitch * 4 * lines <= win_budget && bw * bh <= src_budget;
}
// One wave searches one item: stages the source block and the reference window into its own LDS slices, returns (sad << 12) | (yy * W + x) of
// the first raster-order minimum (0xffffffff when every position was skipped).  Only wave-level synchronisation.
template <bool PARTIAL>
__device__ __forceinline__ uint32_t sad_loop_ring_wave(uint32_t* src_lds, uint32_t* win, const uint8_t* __restrict__ src_base,
                                                       const uint8_t* __restrict__ ref_base, const SvtHipSadLoopDesc& d, const int l) {
    const int bw = d.block_width, bh = d.block_height, W = d.search_area_width, H = d.search_area_height;
    const int rstep = (int)(d.ref_stride / d.src_stride_raw);
    const int src_pitch = bw >> 2, win_pitch = (((bw + W + 3) >> 2) + 3) & ~1, lines = H + rstep * (bh - 1);
    stage_rows_wide_any<64>(src_lds, src_pitch, src_base + d.src_off, d.src_stride, bw, bh, l);
    stage_rows_wide_any<64>(win, win_pitch, ref_base + d.ref_off, d.src_stride_raw, bw + W - 1, lines, l);
    __builtin_amdgcn_wave_barrier(); // the slices belong to this wave alone: LDS program order is enough on the hardware

    const int sbc  = bw >> 3, nsb = sbc * sbc;           // sub-block grid sbc x sbc: 4, 16 or 64 lanes per x-group
    const int RB   = 8 / rstep;                          // used rows per sub-block
    const int sub  = l & (nsb - 1), xg = l / nsb, XG = 64 / nsb;
    const int sx   = sub % sbc, sy = sub / sbc;
    const int q    = l & 3;
    const int nrow = PARTIAL ? bh - sy * RB : RB;        // valid rows of this lane's sub-block (<= 0: the sub-block lies below the block)
    uint32_t s[8][2];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int rr = (r < RB && (!PARTIAL || r < nrow)) ? r : 0;
        const int row = PARTIAL && nrow <= 0 ? 0 : sy * RB + rr;
        s[r][0] = src_lds[row * src_pitch + sx * 2 + 0];
        s[r][1] = src_lds[row * src_pitch + sx * 2 + 1];
    }
    const bool     skip_rule = (bw == 16) && (bh <= 16) && d.skip_search_line; // compute_sad_c.c:74-79: even search lines are skipped
    const uint32_t qsel = 0x0c0c0100u + 0x0202u * (uint32_t)q;
    uint32_t best = 0xffffffffu; // (sad << 12) | (yy * W + x): sad < 2^20, position < 2^12
    // The XG lane groups of a wave cover the x-groups of the area; when the area is narrower than 4 * XG positions (the HME shapes: 16 or 8
    // wide against 64 / 16 positions per step) the spare groups take contiguous chunks of the search lines instead of idling.
    const int GX = (W + 3) >> 2;
    int       gxp = 1;
    while (gxp < GX && gxp < XG) gxp <<= 1; // x-groups per pass: a power of two <= XG
    const int YG = XG / gxp, gx = xg & (gxp - 1), yg = xg / gxp;
    for (int g0 = 0; g0 < GX; g0 += gxp) {
        const int  g   = g0 + gx;        // this lane's x-group: positions 4 g .. 4 g + 3
        const bool gok = g < GX;
        const int  x   = 4 * g + q;      // the position this lane reports after the quad reduction
        const uint32_t* colp = win + (rstep * sy * RB) * win_pitch + sx * 2 + (gok ? g : 0);
        for (int par = 0; par < rstep; par++) {
            const int K  = (H - par + rstep - 1) / rstep; // search lines of this parity: yy = par + rstep * k
            const int CH = (K + YG - 1) / YG, k0 = yg * CH; // lines per y-group; this lane starts at line k0
            // window line of block row r at search line k: par + rstep * (k + r) below this lane's first sub-block line; idle y-groups (k >= K)
            // are clamped to the last staged line so that nothing outside the slice is touched
            const int last = lines - 1 - rstep * sy * RB;
            U64A4 ra[8], rb[8];
#pragma unroll
            for (int r = 0; r < 7; r++) {
                const int rr = r < RB - 1 ? r : 0, ln = par + rstep * (k0 + rr);
                ra[r] = *(const U64A4*)(colp + (ln < last ? ln : last) * win_pitch);
                rb[r] = *(const U64A4*)(colp + (ln < last ? ln : last) * win_pitch + 1);
            }
            // ring slot of block row r at step k is (k + r) % RB; RB is 8 or 4, so the 8-way unrolled body indexes registers statically
            for (int kb = 0; kb < CH; kb += 8) {
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    if (kb + i < CH) {
                        const int k = k0 + kb + i, yy = par + rstep * k;
                        const int       ln = par + rstep * (k + RB - 1);
                        const uint32_t* np = colp + (ln < last ? ln : last) * win_pitch;
                        unsigned long long acc = 0;
                        if (RB == 8) {
                            ra[(i + 7) & 7] = *(const U64A4*)(np);
                            rb[(i + 7) & 7] = *(const U64A4*)(np + 1);
#pragma unroll
                            for (int r = 0; r < 8; r++) {
                                if (!PARTIAL || r < nrow) {
                                    acc = __builtin_amdgcn_qsad_pk_u16_u8(ra[(i + r) & 7].v, s[r][0], acc);
                                    acc = __builtin_amdgcn_qsad_pk_u16_u8(rb[(i + r) & 7].v, s[r][1], acc);
                                }
                            }
                        } else {
                            ra[(i + 3) & 3] = *(const U64A4*)(np);
                            rb[(i + 3) & 3] = *(const U64A4*)(np + 1);
#pragma unroll
                            for (int r = 0; r < 4; r++) {
                                if (!PARTIAL || r < nrow) {
                                    acc = __builtin_amdgcn_qsad_pk_u16_u8(ra[(i + r) & 3].v, s[r][0], acc);
                                    acc = __builtin_amdgcn_qsad_pk_u16_u8(rb[(i + r) & 3].v, s[r][1], acc);
                                }
                            }
                        }
                        const uint32_t lo = (uint32_t)acc, hi = (uint32_t)(acc >> 32);
                        // quad = the four sub-blocks of a 16x16 (u16 lanes: 4 * 16320 < 65536); lane q then takes position q
                        const uint32_t tlo = dpp_add_quad_xor2(dpp_add_quad_xor1(lo));
                        const uint32_t thi = dpp_add_quad_xor2(dpp_add_quad_xor1(hi));
                        uint32_t sad = __builtin_amdgcn_perm(thi, tlo, qsel);
                        if (nsb >= 16) sad = dpp_add_row_ror8(dpp_add_row_ror4(sad));
                        if (nsb == 64) {
                            const auto     x16  = __builtin_amdgcn_permlane16_swap(sad, sad, false, false);
                            const uint32_t pair = x16[0] + x16[1];
                            const auto     x32  = __builtin_amdgcn_permlane32_swap(pair, pair, false, false);
                            sad = x32[0] + x32[1];
                        }
                        const bool ok = gok && x < W && k < K && !(skip_rule && ((yy & 1) == 0));
                        const uint32_t key = ok ? ((sad << 12) | (uint32_t)(yy * W + x)) : 0xffffffffu;
                        best = key < best ? key : best;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)best, m);
        best = o < best ? o : best;
    }
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)best); // every lane holds the minimum: tell the compiler it is wave-uniform
}
__global__ __launch_bounds__(256) void sad_loop_ring_kernel(const uint8_t* __restrict__ src_base, const uint8_t* __restrict__ ref_base,
                                                            const SvtHipSadLoopDesc* __restrict__ descs, const uint32_t n,
                                                            unsigned long long* __restrict__ keys, uint32_t* __restrict__ todo, const int win_budget,
                                                            const int src_budget) {
    HIP_DYNAMIC_SHARED(uint32_t, smem)
    const int      l = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); // wave-uniform: descriptor math on the SALU
    const uint32_t item = blockIdx.x * 4 + wv;
    const bool     have = item < n;
    const SvtHipSadLoopDesc d = descs[have ? item : 0];
    const bool     mine = have && sad_loop_ring_eligible(d, win_budget, src_budget);
    if (have && !mine && l == 0) todo[1 + atomicAdd(&todo[0], 1u)] = item; // work list of the generic kernel: todo[0] = count, todo[1..] = items
    if (!mine) return;
    uint32_t* src_lds = smem + wv * ((win_budget + src_budget) / 4);
    uint32_t* win     = src_lds + src_budget / 4;
    const int rstep = (int)(d.ref_stride / d.src_stride_raw);
    const uint32_t best = d.block_height * rstep == d.block_width ? sad_loop_ring_wave<false>(src_lds, win, src_base, ref_base, d, l)
                                                                  : sad_loop_ring_wave<true>(src_lds, win, src_base, ref_base, d, l);
    if (l == 0 && best != 0xffffffffu) keys[item] = ((unsigned long long)(best >> 12) << 32) | (unsigned long long)(best & 0xfffu);
}

// ---- the three HME levels of one (reference, SB, region) item in ONE wave: level N+1 only needs level N's winner of the same item, so the
// chain needs no grid-wide step.  Per level: hme_item_geometry (the reference's placement / clipping), the ring search above (or, for shapes it
// does not take -- block widths other than 16 / 32 / 64 at a ragged right picture edge -- a plain lane-per-position search), the sub-sampling
// factor and the rescale to the next level.  One launch instead of 3 x (descriptors, ring search, generic search, finalize, rescale).
__device__ __forceinline__ unsigned long long sad_loop_plain_wave(const uint8_t* __restrict__ src_base, const uint8_t* __restrict__ ref_base,
                                                                  const SvtHipSadLoopDesc& d, const int l) {
    const int bw = d.block_width, bh = d.block_height, W = d.search_area_width, H = d.search_area_height;
    unsigned long long best = ~0ull; // (sad << 32) | raster position
    for (int p = l; p < W * H; p += 64) {
        const int yy = p / W, xx = p - yy * W;
        const uint8_t* s = src_base + d.src_off;
        const uint8_t* r = ref_base + d.ref_off + (size_t)yy * d.src_stride_raw + xx;
        uint32_t sad = 0;
        for (int y = 0; y < bh; y++)
            for (int x = 0; x < bw; x++) {
                const int df = (int)s[(size_t)y * d.src_stride + x] - (int)r[(size_t)y * d.ref_stride + x];
                sad += (uint32_t)(df < 0 ? -df : df);
            }
        const unsigned long long key = ((unsigned long long)sad << 32) | (uint32_t)p;
        best = key < best ? key : best;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const unsigned long long o = ((unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)(best >> 32), m) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)best, m);
        best = o < best ? o : best;
    }
    return ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(best >> 32)) << 32) |
           (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)best);
}
struct HmeChainArgs {
    SvtHipHmeLevelParams P[3];
    const uint8_t *src[3], *ref[3];
    unsigned long long* sad_out[3];
    int16_t* sc_out[3];
    const uint32_t* zz_sad;
    const uint8_t* do_ref;
    const SvtHipPrehmeResult* prehme;
    uint32_t n, prev_stage_th;
    int win_budget, src_budget;
    int n_levels, list1_skip;
    uint32_t item0; // first item of this launch (the level-0 resizing from list 0's motion runs reference 0 first)
    // XCD-aware placement: workgroups reach the 8 XCDs round-robin by their id, so raster-adjacent SBs -- whose search windows overlap almost entirely on the 1/16 and
    // 1/4 planes -- landed on 8 different L2s and every L2 fetched its own copy of the overlap (7.7x the planes' bytes moved, BENCH_r05).  Workgroup b takes logical
    // position (b % 8) * wg_per_xcd + b / 8 instead: each XCD walks one contiguous band of SB rows.  (An affinity for speed only; results do not depend on it.)  0 = off.
    uint32_t wg_per_xcd, n_wg;
};
__global__ __launch_bounds__(256, 4) void hme_chain_kernel(const HmeChainArgs A) {
    HIP_DYNAMIC_SHARED(uint32_t, smem)
    __shared__ unsigned long long sh_l0[4]; // level-0 SADs of the workgroup's four items (the 2 x 2 regions of one (reference, SB) in the pre-HME form)
    // everything but the pixel work is wave-uniform (item index, geometry, descriptors, winners): kept on the SALU through readfirstlane
    const int      l = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint32_t wg = blockIdx.x;
    if (A.wg_per_xcd) {
        wg = (blockIdx.x & 7u) * A.wg_per_xcd + (blockIdx.x >> 3);
        if (wg >= A.n_wg) return; // (the grid is rounded up to 8 bands: whole workgroups leave, ahead of every barrier)
    }
    const uint32_t item = A.item0 + wg * 4 + wv;
    const bool     have = item < A.n; // (kept alive for the workgroup barrier of the pre-HME form; the grid is exact then)
    uint32_t* src_lds = smem + wv * ((A.win_budget + A.src_budget) / 4);
    uint32_t* win     = src_lds + A.src_budget / 4;
    const uint32_t regions = (uint32_t)A.P[0].num_hme_sa_w * A.P[0].num_hme_sa_h, n_sb = A.P[0].sbs_x * A.P[0].sbs_y;
    const uint32_t rs = (have ? item : 0) / regions, sb = rs % n_sb, r = rs / n_sb; // [ref][sb] index of this item = rs
    // per-(reference, SB) gates, the same for every region: zero-motion SAD below the threshold (centre 0, SAD 0) has priority over a dropped reference
    // (centre 0, SAD MAX_U32); both only at levels 0 and 1 (hme_level0_b64 :1922-1973, hme_level1_b64 :2057-2082)
    const bool zz_skip = have && A.P[0].zz_skip_th && A.zz_sad && A.zz_sad[rs] < A.P[0].zz_skip_th;
    const bool dropped = have && A.do_ref && !A.do_ref[(size_t)sb * 8 + (r < A.P[0].n_refs_list0 ? 0 : 4) + A.P[0].ref_pic_index[r]];
    // base layer: list 1 is not searched by HME at all (:1983, :2055, :2127); uniform over the workgroup in the pre-HME form (its items share the reference)
    if (A.list1_skip && r >= A.P[0].n_refs_list0) return;
    int16_t px = 0, py = 0;
    unsigned long long prev_lsad = 0; // the previous level's stored SAD of this item
#pragma unroll 1
    for (int lv = 0; lv < A.n_levels; lv++) {
        const SvtHipHmeLevelParams& P = A.P[lv];
        // gates in the reference's order: zero-motion exit, (level 0) pre-HME result good enough, dropped reference, (levels 1, 2) previous level good enough
        bool gated = false;
        if (lv < 2 && zz_skip) { px = py = 0; prev_lsad = 0ull; gated = true; }
        if (!gated && lv == 0 && have && A.prev_stage_th && A.prehme) {
            // prev_me_stage_based_exit_th (:1937-1957): every region of the (reference, SB) takes the better pre-HME region, provided that one was searched
            const SvtHipPrehmeResult* pr = A.prehme + (size_t)rs * 2;
            const int sr = pr[0].sad <= pr[1].sad ? 0 : 1;
            if (pr[sr].performed && pr[sr].sad < (A.prev_stage_th >> 4)) { px = pr[sr].mv_x; py = pr[sr].mv_y; prev_lsad = pr[sr].sad; gated = true; }
        }
        if (!gated && lv < 2 && dropped) { px = py = 0; prev_lsad = 0xffffffffull; gated = true; }
        // (:2086-2096, :2144-2154) the previous level's centre and SAD are kept as they are
        if (!gated && lv > 0 && have && A.prev_stage_th && prev_lsad < (A.prev_stage_th >> (lv == 1 ? 5 : 2))) gated = true;
        if (gated) { // (level 0 / 1 gates but the last are uniform over the workgroup in the pre-HME form: its four items share (reference, SB))
            if (have && l == 0) { A.sad_out[lv][item] = prev_lsad; A.sc_out[lv][2 * item] = px; A.sc_out[lv][2 * item + 1] = py; }
            continue;
        }
        uint32_t sad = 0xffffff; // svt_sad_loop_kernel's initial best (compute_sad_c.c:71)
        if (have) {
            SvtHipSadLoopDesc d;
            int16_t ox, oy;
            int l0_flags = 0;
            if (lv == 0 && P.l0_mv_th_min && P.l0_mv_th_max && r > 0) { // the motion list 0 / reference 0 found at level 0 for this SB, region (0, 0): an earlier launch wrote it
                const uint32_t it0 = sb * regions;
                const int      l0x = A.sc_out[0][2 * it0], l0y = A.sc_out[0][2 * it0 + 1], ax = l0x < 0 ? -l0x : l0x, ay = l0y < 0 ? -l0y : l0y;
                const bool     is_ver = ax < (int)P.l0_mv_th_min && ay > (int)P.l0_mv_th_max, is_hor = ax > (int)P.l0_mv_th_max && ay < (int)P.l0_mv_th_min;
                const bool     is_still = P.l0_still_rule && ax < 3 * (int)P.l0_mv_th_min && ay < 3 * (int)P.l0_mv_th_min; // (:1825-1826, :1836-1841)
                l0_flags = (is_hor ? 0 : 1) | (is_ver ? 0 : 2) | (is_still ? 4 : 0);
            }
            hme_item_geometry(P, item, px, py, d, ox, oy, l0_flags);
            const int W = d.search_area_width;
            int       pos = -1;
            if (sad_loop_ring_eligible(d, A.win_budget, A.src_budget)) {
                const int      rstep = (int)(d.ref_stride / d.src_stride_raw);
                const uint32_t best  = d.block_height * rstep == d.block_width ? sad_loop_ring_wave<false>(src_lds, win, A.src[lv], A.ref[lv], d, l)
                                                                               : sad_loop_ring_wave<true>(src_lds, win, A.src[lv], A.ref[lv], d, l);
                if (best != 0xffffffffu && (best >> 12) < sad) { sad = best >> 12; pos = (int)(best & 0xfffu); }
                __builtin_amdgcn_wave_barrier(); // the LDS slices are restaged by the next level
            } else if (W > 0 && d.search_area_height > 0) {
                const unsigned long long best = sad_loop_plain_wave(A.src[lv], A.ref[lv], d, l);
                if ((uint32_t)(best >> 32) < sad) { sad = (uint32_t)(best >> 32); pos = (int)(uint32_t)best; }
            }
            // an empty or all-skipped area leaves the reference's centre variable untouched: sc_out is in/out, like svt_hip_hme_level_batch
            int16_t x = A.sc_out[lv][2 * item], y = A.sc_out[lv][2 * item + 1];
            if (pos >= 0) { y = (int16_t)(pos / W); x = (int16_t)(pos - (pos / W) * W); }
            const int scale = P.level == 0 ? 4 : (P.level == 1 ? 2 : 1);
            px = (int16_t)((int16_t)(x + ox) * scale);
            py = (int16_t)((int16_t)(y + oy) * scale);
        }
        unsigned long long lsad = P.sub_sampled ? (unsigned long long)sad * 2 : (unsigned long long)sad;
        if (lv == 0 && P.prehme_enabled && A.prehme) { // replace the worst of the four regions by the better pre-HME result when that beats it
            if (l == 0) sh_l0[wv] = lsad;
            __syncthreads();
            int worst = 0; // get_worst_quadrant (:1872-1900): first strictly larger SAD in region order (w0h0), (w1h0), (w0h1), (w1h1), starting from 0
            unsigned long long mx = 0;
            for (int k = 0; k < 4; k++) {
                const unsigned long long v = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(sh_l0[k] >> 32)) << 32) |
                                             (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)sh_l0[k]);
                if (v > mx) { mx = v; worst = k; }
            }
            const SvtHipPrehmeResult* pr = A.prehme + (size_t)rs * 2;
            const int sr = pr[0].sad <= pr[1].sad ? 0 : 1;
            if (have && wv == worst && pr[sr].sad < mx) { lsad = pr[sr].sad; px = pr[sr].mv_x; py = pr[sr].mv_y; }
        }
        prev_lsad = lsad;
        if (have && l == 0) {
            A.sad_out[lv][item]        = lsad;
            A.sc_out[lv][2 * item]     = px;
            A.sc_out[lv][2 * item + 1] = py;
        }
    }
}

// ---- pre-HME of one (SB, reference index) pair per workgroup: waves 0, 1 = the two search regions of list 0's reference, waves 2, 3 = list 1's (they
// may shortcut from list 0's result of the same index, so they run second).
struct PrehmeArgs {
    SvtHipPrehmeParams P;
    const uint8_t *src, *ref;
    const uint32_t* zz_sad;
    const uint8_t* do_ref;
    SvtHipPrehmeResult* out;
    int win_budget, src_budget, n_l1;
};
__device__ __forceinline__ void prehme_wave(const PrehmeArgs& A, uint32_t* src_lds, uint32_t* win, const uint32_t sb, const uint32_t slot, const int sr_i, const int l,
                                            SvtHipPrehmeResult& res) {
    const SvtHipHmeLevelParams& G = A.P.plane;
    const uint32_t n_sb = G.sbs_x * G.sbs_y;
    const uint32_t fx = (sb % G.sbs_x) * 64, fy = (sb / G.sbs_x) * 64;
    const uint32_t b64_w = G.aligned_width - fx < 64 ? G.aligned_width - fx : 64, b64_h = G.aligned_height - fy < 64 ? G.aligned_height - fy : 64;
    const int16_t  org_x = (int16_t)((int16_t)fx >> 2), org_y = (int16_t)((int16_t)fy >> 2);
    // prehme_core (:1568-1666): area centred on the co-located block, clipped like integer_search_b64 (origin and size in separate conditionals)
    const uint32_t fw = (uint32_t)A.P.sa_min_width[sr_i] * A.P.hme_sr_factor[slot], fh = (uint32_t)A.P.sa_min_height[sr_i] * A.P.hme_sr_factor[slot];
    int16_t search_area_width  = (int16_t)(uint16_t)(fw < A.P.sa_max_width[sr_i] ? fw : A.P.sa_max_width[sr_i]);
    int16_t search_area_height = (int16_t)(uint16_t)(fh < A.P.sa_max_height[sr_i] ? fh : A.P.sa_max_height[sr_i]);
    const int16_t pad_width = (int16_t)((int)G.ref_org_x - 1), pad_height = (int16_t)((int)G.ref_org_y - 1), ref_w = (int16_t)G.ref_width, ref_h = (int16_t)G.ref_height;
    int16_t xo = (int16_t)(-(int16_t)(search_area_width >> 1)), yo = (int16_t)(-(int16_t)(search_area_height >> 1));
    xo = (int16_t)(((org_x + xo) < -pad_width) ? -pad_width - org_x : xo);
    search_area_width = (int16_t)(((org_x + xo) < -pad_width) ? search_area_width - (-pad_width - (org_x + xo)) : search_area_width);
    xo = (int16_t)(((org_x + xo) > ref_w - 1) ? xo - ((org_x + xo) - (ref_w - 1)) : xo);
    if ((org_x + xo + search_area_width) > ref_w) { const int w = search_area_width - ((org_x + xo + search_area_width) - ref_w); search_area_width = (int16_t)(w > 1 ? w : 1); }
    yo = (int16_t)(((org_y + yo) < -pad_height) ? -pad_height - org_y : yo);
    search_area_height = (int16_t)(((org_y + yo) < -pad_height) ? search_area_height - (-pad_height - (org_y + yo)) : search_area_height);
    yo = (int16_t)(((org_y + yo) > ref_h - 1) ? yo - ((org_y + yo) - (ref_h - 1)) : yo);
    if ((org_y + yo + search_area_height) > ref_h) { const int h = search_area_height - ((org_y + yo + search_area_height) - ref_h); search_area_height = (int16_t)(h > 1 ? h : 1); }
    const int16_t  x_tl = (int16_t)(((int16_t)G.ref_org_x + org_x) + xo), y_tl = (int16_t)(((int16_t)G.ref_org_y + org_y) + yo);
    const uint32_t step = G.sub_sampled ? 2 : 1;
    SvtHipSadLoopDesc d;
    d.src_off = G.src_off + (uint64_t)org_y * G.src_stride + (uint64_t)org_x;
    d.ref_off = G.ref_off[slot] + (uint32_t)(x_tl + y_tl * (int)G.ref_stride);
    d.src_stride = G.src_stride * step; d.ref_stride = G.ref_stride * step; d.src_stride_raw = G.ref_stride;
    d.block_width = (uint16_t)(b64_w >> 2); d.block_height = (uint16_t)((b64_h >> 2) / step);
    d.search_area_width = search_area_width; d.search_area_height = search_area_height; d.skip_search_line = A.P.skip_search_line;
    uint32_t sad = 0xffffff;
    int      pos = -1;
    const int W = search_area_width;
    if (sad_loop_ring_eligible(d, A.win_budget, A.src_budget)) {
        const int      rstep = (int)(d.ref_stride / d.src_stride_raw);
        const uint32_t best  = d.block_height * rstep == d.block_width ? sad_loop_ring_wave<false>(src_lds, win, A.src, A.ref, d, l)
                                                                       : sad_loop_ring_wave<true>(src_lds, win, A.src, A.ref, d, l);
        if (best != 0xffffffffu && (best >> 12) < sad) { sad = best >> 12; pos = (int)(best & 0xfffu); }
    } else if (W > 0 && search_area_height > 0) {
        // lane-per-position search honouring skip_search_line (compute_sad_c.c:74-79)
        unsigned long long best = ~0ull;
        const int bw = d.block_width, bh = d.block_height;
        for (int p = l; p < W * search_area_height; p += 64) {
            const int yy = p / W, xx = p - yy * W;
            if (bw == 16 && bh <= 16 && d.skip_search_line && (yy & 1) == 0) continue;
            const uint8_t* s = A.src + d.src_off;
            const uint8_t* f = A.ref + d.ref_off + (size_t)yy * d.src_stride_raw + xx;
            uint32_t v = 0;
            for (int y = 0; y < bh; y++)
                for (int x = 0; x < bw; x++) { const int df = (int)s[(size_t)y * d.src_stride + x] - (int)f[(size_t)y * d.ref_stride + x]; v += (uint32_t)(df < 0 ? -df : df); }
            const unsigned long long key = ((unsigned long long)v << 32) | (uint32_t)p;
            best = key < best ? key : best;
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const unsigned long long o = ((unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)(best >> 32), m) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)best, m);
            best = o < best ? o : best;
        }
        if (best != ~0ull && (uint32_t)(best >> 32) < sad) { sad = (uint32_t)(best >> 32); pos = (int)(uint32_t)best; }
    }
    // svt_sad_loop_kernel leaves the centre untouched when nothing beat 0xffffff: the reference's SearchInfo then keeps its previous content; start from 0
    const int16_t x = pos >= 0 ? (int16_t)(pos - (pos / W) * W) : (int16_t)0, y = pos >= 0 ? (int16_t)(pos / W) : (int16_t)0;
    res.sad  = G.sub_sampled ? (unsigned long long)sad * 2 : (unsigned long long)sad;
    res.mv_x = (int16_t)((int16_t)(x + xo) * 4);
    res.mv_y = (int16_t)((int16_t)(y + yo) * 4);
    res.valid = 1; res.performed = 1;
    (void)n_sb;
}
__global__ __launch_bounds__(256, 4) void prehme_kernel(const PrehmeArgs A) {
    HIP_DYNAMIC_SHARED(uint32_t, smem)
    __shared__ SvtHipPrehmeResult sh_l0[2];
    const SvtHipHmeLevelParams& G = A.P.plane;
    const int      l = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), list = wv >> 1, sr_i = wv & 1;
    const uint32_t n_sb = G.sbs_x * G.sbs_y, sb = blockIdx.x, ref_i = blockIdx.y;
    const int      n_l0 = G.n_refs_list0, n_l1 = A.n_l1;
    uint32_t* src_lds = smem + sr_i * ((A.win_budget + A.src_budget) / 4); // the lists run in different phases: list 1's waves reuse list 0's slices
    uint32_t* win     = src_lds + A.src_budget / 4;
    const bool have = list == 0 ? (int)ref_i < n_l0 : (int)ref_i < n_l1;
    const uint32_t slot = list == 0 ? ref_i : (uint32_t)n_l0 + ref_i;
    for (int phase = 0; phase < 2; phase++) {
        if (phase == list && have) {
            const size_t o = ((size_t)slot * n_sb + sb) * 2 + sr_i;
            SvtHipPrehmeResult res;
            res.sad = 0; res.mv_x = 0; res.mv_y = 0; res.valid = 0; res.performed = 0; res.pad[0] = res.pad[1] = 0;
            bool done = false;
            if (list == 1 && !A.P.temporal_layer_gt0) { // :1762-1770: list 1 mirrors list 0 when it is not searched at this layer
                const SvtHipPrehmeResult& z = sh_l0[sr_i];
                res.sad = z.sad; res.mv_x = (int16_t)-z.mv_x; res.mv_y = (int16_t)-z.mv_y; done = true;
            }
            // check_prehme_early_exit (:1690-1717)
            if (!done && A.P.me_early_exit_th && A.zz_sad[(size_t)slot * n_sb + sb] < A.P.me_early_exit_th) { res.valid = 1; done = true; } // mv 0, sad 0
            if (!done && A.P.l1_early_exit && list == 1 && (int)ref_i < n_l0) {
                const SvtHipPrehmeResult& z = sh_l0[sr_i];
                const int ax = z.mv_x < 0 ? -z.mv_x : z.mv_x, ay = z.mv_y < 0 ? -z.mv_y : z.mv_y;
                if (z.valid && (z.sad < 32 * 32 || (ax < 16 && ay < 16))) { res.sad = z.sad; res.mv_x = (int16_t)-z.mv_x; res.mv_y = (int16_t)-z.mv_y; res.valid = 1; done = true; }
            }
            if (!done && A.do_ref && !A.do_ref[(size_t)sb * 8 + list * 4 + ref_i]) { res.sad = 0xffffffffull; done = true; } // :1737-1742
            if (!done) prehme_wave(A, src_lds, win, sb, slot, sr_i, l, res);
            if (l == 0) {
                A.out[o] = res;
                if (list == 0) sh_l0[sr_i] = res;
            }
        }
        __syncthreads();
    }
}
// reference pruning on the pre-HME SADs (:1781-1797) and on the zero-motion SADs (init_zz_sad :2402-2417): one thread per SB
__global__ void ref_prune_pct_kernel(const SvtHipHmeLevelParams G, const SvtHipPrehmeResult* __restrict__ pre, const uint32_t* __restrict__ zz, const uint32_t th,
                                     const uint32_t pct, const int tl_gt0, uint8_t* __restrict__ do_ref) {
    const uint32_t n_sb = G.sbs_x * G.sbs_y, sb = blockIdx.x * blockDim.x + threadIdx.x;
    if (sb >= n_sb || !tl_gt0) return;
    uint32_t v[8], best = 0xffffffffu;
    for (uint32_t r = 0; r < G.n_refs; r++) {
        if (pre) { const SvtHipPrehmeResult* p = pre + ((size_t)r * n_sb + sb) * 2; v[r] = (uint32_t)(p[0].sad < p[1].sad ? p[0].sad : p[1].sad); }
        else v[r] = zz[(size_t)r * n_sb + sb];
        best = v[r] < best ? v[r] : best;
    }
    if (!(best < th)) return;
    for (uint32_t r = 0; r < G.n_refs; r++) {
        const size_t dri = (size_t)sb * 8 + (r < G.n_refs_list0 ? 0 : 4) + G.ref_pic_index[r];
        if (G.ref_pic_index[r] == 0) continue;
        if (pre && !do_ref[dri]) continue;
        if ((uint32_t)((v[r] - best) * 100) > (uint32_t)(pct * best)) do_ref[dri] = 0;
    }
}

__global__ void sad_loop_finalize_kernel(const SvtHipSadLoopDesc* __restrict__ descs, uint32_t n, const unsigned long long* __restrict__ keys,
                                         SvtHipSadLoopResult* __restrict__ res) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long key = keys[i];
    SvtHipSadLoopResult      r;
    const uint32_t sad = (uint32_t)(key >> 32);
    // reference: *best_sad = 0xffffff, update only when sad < best (compute_sad_c.c:71,91)
    if (key == ~0ull || sad >= 0xffffffu) {
        r.best_sad = 0xffffff; r.x_search_center = 0; r.y_search_center = 0; r.valid = 0;
    } else {
        const uint32_t pos = (uint32_t)key, W = (uint32_t)descs[i].search_area_width;
        r.best_sad = sad; r.x_search_center = (int16_t)(pos % W); r.y_search_center = (int16_t)(pos / W); r.valid = 1;
    }
    res[i] = r;
}

// ---- single-call ext_* kernels ---------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t sad8x8_bytes(const uint8_t* s, uint32_t ss, const uint8_t* r, uint32_t rs, bool sub) {
    uint32_t sad = 0;
    for (int y = 0; y < 8; y += (sub ? 2 : 1))
        for (int x = 0; x < 8; x++) {
            const int a = s[y * ss + x], b = r[y * rs + x];
            sad += (uint32_t)(a > b ? a - b : b - a);
        }
    return sub ? sad << 1 : sad;
}
// 512 threads: (8x8 block b8 in reference numbering, search_index p)
__global__ __launch_bounds__(512) void ext_all_sad_kernel(const uint8_t* src, uint32_t ss, const uint8_t* ref, uint32_t rs, uint32_t mv,
                                                          uint32_t* best8, uint32_t* best16, uint32_t* mv8, uint32_t* mv16,
                                                          uint32_t* eight16 /*[16][8]*/, int sub) {
    __shared__ uint32_t s8[64][8];
    __shared__ uint32_t s16[16][8];
    const int tid = threadIdx.x, b8 = tid >> 3, p = tid & 7;
    const int bx = (b8 & 1) | ((b8 >> 1) & 2) | ((b8 >> 2) & 4), by = ((b8 >> 1) & 1) | ((b8 >> 2) & 2) | ((b8 >> 3) & 4);
    s8[b8][p] = sad8x8_bytes(src + by * 8 * ss + bx * 8, ss, ref + by * 8 * rs + bx * 8 + p, rs, sub != 0);
    __syncthreads();
    if (tid < 128) {
        const int i16 = tid >> 3;
        const uint32_t v = s8[4 * i16][p] + s8[4 * i16 + 1][p] + s8[4 * i16 + 2][p] + s8[4 * i16 + 3][p];
        s16[i16][p] = v;
        eight16[i16 * 8 + p] = v;
    }
    __syncthreads();
    const int16_t xm = (int16_t)(mv & 0xffff), ym = (int16_t)(mv >> 16);
    if (tid < 64) {
        uint32_t b = best8[tid], m = mv8[tid];
        for (int k = 0; k < 8; k++)
            if (s8[tid][k] < b) { b = s8[tid][k]; m = ((uint32_t)(uint16_t)ym << 16) | (uint16_t)(int16_t)(xm + k); }
        best8[tid] = b; mv8[tid] = m;
    } else if (tid < 80) {
        const int i = tid - 64;
        uint32_t  b = best16[i], m = mv16[i];
        for (int k = 0; k < 8; k++)
            if (s16[i][k] < b) { b = s16[i][k]; m = ((uint32_t)(uint16_t)ym << 16) | (uint16_t)(int16_t)(xm + k); }
        best16[i] = b; mv16[i] = m;
    }
}
__global__ __launch_bounds__(64) void ext_eight_32_64_kernel(const uint32_t* sad16 /*[16][8]*/, uint32_t* best32, uint32_t* best64,
                                                             uint32_t* mv32, uint32_t* mv64, uint32_t mv, uint32_t* sad32 /*[4][8]*/) {
    __shared__ uint32_t s32[4][8];
    const int tid = threadIdx.x;
    if (tid < 32) {
        const int k = tid >> 3, p = tid & 7;
        const uint32_t v = sad16[(4 * k) * 8 + p] + sad16[(4 * k + 1) * 8 + p] + sad16[(4 * k + 2) * 8 + p] + sad16[(4 * k + 3) * 8 + p];
        s32[k][p] = v; sad32[k * 8 + p] = v;
    }
    __syncthreads();
    const int16_t xm = (int16_t)(mv & 0xffff), ym = (int16_t)(mv >> 16);
    if (tid < 4) {
        uint32_t b = best32[tid], m = mv32[tid];
        for (int k = 0; k < 8; k++)
            if (s32[tid][k] < b) { b = s32[tid][k]; m = ((uint32_t)(uint16_t)ym << 16) | (uint16_t)(int16_t)(xm + k); }
        best32[tid] = b; mv32[tid] = m;
    } else if (tid == 4) {
        uint32_t b = best64[0], m = mv64[0];
        for (int k = 0; k < 8; k++) {
            const uint32_t v = s32[0][k] + s32[1][k] + s32[2][k] + s32[3][k];
            if (v < b) { b = v; m = ((uint32_t)(uint16_t)ym << 16) | (uint16_t)(int16_t)(xm + k); }
        }
        best64[0] = b; mv64[0] = m;
    }
}
__global__ __launch_bounds__(64) void ext_one_8_16_kernel(const uint8_t* src, uint32_t ss, const uint8_t* ref, uint32_t rs, uint32_t* best8,
                                                          uint32_t* best16, uint32_t* mv8, uint32_t* mv16, uint32_t mv, uint32_t* sad16,
                                                          uint32_t* sad8, int sub) {
    __shared__ uint32_t s[4];
    const int tid = threadIdx.x;
    if (tid < 4) {
        const int      dx = (tid & 1) * 8, dy = (tid >> 1) * 8;
        const uint32_t v  = sad8x8_bytes(src + dy * ss + dx, ss, ref + dy * rs + dx, rs, sub != 0);
        s[tid] = v; sad8[tid] = v;
        if (v < best8[tid]) { best8[tid] = v; mv8[tid] = mv; }
    }
    __syncthreads();
    if (tid == 0) {
        const uint32_t v = s[0] + s[1] + s[2] + s[3];
        if (v < best16[0]) { best16[0] = v; mv16[0] = mv; }
        sad16[0] = v;
    }
}
__global__ __launch_bounds__(64) void ext_one_32_64_kernel(const uint32_t* sad16, uint32_t* best32, uint32_t* best64, uint32_t* mv32,
                                                           uint32_t* mv64, uint32_t mv, uint32_t* sad32) {
    __shared__ uint32_t s[4];
    const int tid = threadIdx.x;
    if (tid < 4) {
        const uint32_t v = sad16[4 * tid] + sad16[4 * tid + 1] + sad16[4 * tid + 2] + sad16[4 * tid + 3];
        s[tid] = v; sad32[tid] = v;
        if (v < best32[tid]) { best32[tid] = v; mv32[tid] = mv; }
    }
    __syncthreads();
    if (tid == 0) {
        const uint32_t v = s[0] + s[1] + s[2] + s[3];
        if (v < best64[0]) { best64[0] = v; mv64[0] = mv; }
    }
}
__global__ void fill_u32_kernel(uint32_t* p, uint32_t n, uint32_t v) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

} // namespace

// =================================================================================================================
// C ABI
// =================================================================================================================
extern "C" {

void svt_hip_sad_nxm_batch(const uint8_t* src_base, const uint8_t* ref_base, const SvtHipSadPair* pairs, uint32_t n, uint32_t width,
                           uint32_t height, uint32_t* sad_out, void* stream) {
    svthip::ensure_device();
    if (n == 0) return;
    const uint32_t cpr = width >> 4;
    // SVT_HIP_SAD_FORM (measurement knob): 0 = pair-per-wave forms (default), 1 = the strip form, which measured SLOWER on the MI355X (221 vs 204 us per 122 400 64x64
    // pairs, 1.53 vs 1.35 x the algorithmic bytes moved: profiles/r03_call3_ab_sad_cdef.txt) and is kept for that comparison only
    const int form = svthip::tuning_sad_form();
    if (form == 1 && (width & 15) == 0 && cpr != 0 && !(cpr & (cpr - 1)) && cpr <= 16 && n >= 4 * SADS_SPW * (16 / cpr) * 64) {
        const uint32_t per_wg = 4 * SADS_SPW * (16 / cpr); // pairs per workgroup
        hipLaunchKernelGGL(sad_nxm_strip_kernel, dim3((n + per_wg - 1) / per_wg), dim3(256), 0, (hipStream_t)stream, src_base, ref_base, pairs, n, __builtin_ctz(cpr),
                           (int)height, sad_out);
    } else if ((width & 15) == 0 && cpr != 0 && !(cpr & (cpr - 1)) && cpr * height <= 256 && n >= 4 * SADP_PPW * 64) { // enough pairs to fill the chip with walking waves
        hipLaunchKernelGGL(sad_nxm_pipe_kernel, dim3((n + 4 * SADP_PPW - 1) / (4 * SADP_PPW)), dim3(256), 0, (hipStream_t)stream, src_base, ref_base, pairs, n,
                           __builtin_ctz(cpr), (int)(cpr * height), sad_out);
    } else {
        hipLaunchKernelGGL(sad_nxm_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, src_base, ref_base, pairs, n, (int)width,
                           (int)height, sad_out);
    }
    SVT_LAUNCH_CHECK();
}

void svt_hip_sad_loop_batch(const uint8_t* src_base, const uint8_t* ref_base, const SvtHipSadLoopDesc* descs, uint32_t n,
                            uint32_t max_area_width, uint32_t max_area_height, uint32_t max_block_width, uint32_t max_block_height,
                            int max_ref_step, SvtHipSadLoopResult* results, uint64_t* keys, void* stream) {
    svthip::ensure_device();
    if (n == 0) return;
    const int max_w = max_area_width ? (int)max_area_width : 1, max_h = max_area_height ? (int)max_area_height : 1;
    const int max_bw = max_block_width < 4 ? 4 : (int)max_block_width, max_bh = max_block_height ? (int)max_block_height : 1;
    // narrowest tile that covers the widest area; widen (= fewer search rows per tile) while the window does not fit 60 KB of LDS
    int lxg = max_w <= 16 ? 2 : (max_w <= 32 ? 3 : 4);
    size_t shmem = 0;
    for (;; lxg++) {
        const int tw = 4 << lxg, th = 256 >> lxg;
        const int src_pitch = ((max_bw + 15) >> 4) << 2, win_pitch = (((max_bw + tw + 3) >> 2) + 3) & ~1;
        const int win_rows  = th + (max_ref_step < 1 ? 1 : max_ref_step) * (max_bh - 1);
        shmem = (size_t)(src_pitch * max_bh + win_pitch * win_rows) * 4 + 64;
        if (shmem <= 60 * 1024 || lxg == 4) break;
    }
    const int      tw = 4 << lxg, th = 256 >> lxg;
    const uint32_t tiles_x = (max_w + tw - 1) / tw, tiles_y = (max_h + th - 1) / th;
    HIP_CHECK(hipMemsetAsync(keys, 0xff, (size_t)n * 8, (hipStream_t)stream));
    uint32_t* todo = (uint32_t*)results; // (n + 1) dwords of the result array (16 bytes per item) serve as the work list until the finalize kernel fills it
    HIP_CHECK(hipMemsetAsync(todo, 0, 4, (hipStream_t)stream));
    // per-wave LDS slices of the ring kernel, from the batch maxima (small HME shapes then leave room for 4 workgroups per CU)
    int src_budget = max_bw * max_bh, win_budget = ((((max_bw + max_w + 3) >> 2) + 3) & ~1) * 4 * (max_h + (max_ref_step < 1 ? 1 : max_ref_step) * (max_bh - 1));
    src_budget = (src_budget > SLR_SRC_BYTES ? SLR_SRC_BYTES : src_budget + 15) & ~15;
    win_budget = (win_budget > SLR_WIN_BYTES ? SLR_WIN_BYTES : win_budget + 15) & ~15;
    hipLaunchKernelGGL(sad_loop_ring_kernel, dim3((n + 3) / 4), dim3(256), 4 * (size_t)(src_budget + win_budget) + 64, (hipStream_t)stream, src_base, ref_base, descs, n,
                       (unsigned long long*)keys, todo, win_budget, src_budget);
    SVT_LAUNCH_CHECK();
    const dim3 grid(n < 2048 ? n : 2048, tiles_x * tiles_y);
    if (lxg == 2)
        hipLaunchKernelGGL(sad_loop_kernel<2>, grid, dim3(256), shmem, (hipStream_t)stream, src_base, ref_base, descs, (const uint32_t*)todo, tiles_x, (unsigned long long*)keys);
    else if (lxg == 3)
        hipLaunchKernelGGL(sad_loop_kernel<3>, grid, dim3(256), shmem, (hipStream_t)stream, src_base, ref_base, descs, (const uint32_t*)todo, tiles_x, (unsigned long long*)keys);
    else
        hipLaunchKernelGGL(sad_loop_kernel<4>, grid, dim3(256), shmem, (hipStream_t)stream, src_base, ref_base, descs, (const uint32_t*)todo, tiles_x, (unsigned long long*)keys);
    SVT_LAUNCH_CHECK();
    hipLaunchKernelGGL(sad_loop_finalize_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, descs, n,
                       (const unsigned long long*)keys, results);
    SVT_LAUNCH_CHECK();
}

void svt_hip_hme_chain_batch(const SvtHipHmeLevelParams* params, const uint8_t* const* src_base, const uint8_t* const* ref_base,
                             const SvtHipHmeChainInputs* inputs, uint64_t* const* sad_out, int16_t* const* sc_out, void* stream) {
    svthip::ensure_device();
    const uint32_t n = params[0].n_refs * params[0].sbs_x * params[0].sbs_y * params[0].num_hme_sa_w * params[0].num_hme_sa_h;
    if (n == 0) return;
    HmeChainArgs A;
    memset(&A, 0, sizeof(A));
    int src_budget = 0, win_budget = 0;
    const int n_levels = (inputs && inputs->n_levels) ? inputs->n_levels : 3;
    if (n_levels < 1 || n_levels > 3) {
        fprintf(stderr, "libsvtav1_hip: svt_hip_hme_chain_batch: n_levels must be 1, 2 or 3\n");
        abort();
    }
    A.n_levels = n_levels;
    A.list1_skip = inputs ? inputs->list1_no_hme : 0;
    for (int lv = 0; lv < n_levels; lv++) {
        const SvtHipHmeLevelParams& P = params[lv];
        if (P.level != lv || P.n_refs != params[0].n_refs || P.sbs_x != params[0].sbs_x || P.sbs_y != params[0].sbs_y ||
            P.num_hme_sa_w != params[0].num_hme_sa_w || P.num_hme_sa_h != params[0].num_hme_sa_h || P.n_refs > 8) {
            fprintf(stderr, "libsvtav1_hip: svt_hip_hme_chain_batch: the three levels must describe the same items (level %d)\n", lv);
            abort();
        }
        A.P[lv] = P; A.src[lv] = src_base[lv]; A.ref[lv] = ref_base[lv]; A.sad_out[lv] = (unsigned long long*)sad_out[lv]; A.sc_out[lv] = sc_out[lv];
        int maw = P.sa_width, mah = P.sa_height;
        if (P.per_ref_area) {
            maw = mah = 1;
            for (uint32_t r = 0; r < P.n_refs; r++) { maw = P.sa_width_ref[r] > maw ? P.sa_width_ref[r] : maw; mah = P.sa_height_ref[r] > mah ? P.sa_height_ref[r] : mah; }
        }
        const int bw = 64 >> (2 - lv), step = P.sub_sampled ? 2 : 1, bh = bw / step, mw = (maw + 7) & ~7, mh = mah;
        const int sb = bw * bh, wb = ((((bw + mw + 3) >> 2) + 3) & ~1) * 4 * (mh + step * (bh - 1));
        src_budget = sb > src_budget ? sb : src_budget;
        win_budget = wb > win_budget ? wb : win_budget;
    }
    src_budget = (src_budget > SLR_SRC_BYTES ? SLR_SRC_BYTES : src_budget + 15) & ~15;
    win_budget = (win_budget > SLR_WIN_BYTES ? SLR_WIN_BYTES : win_budget + 15) & ~15;
    A.n = n; A.win_budget = win_budget; A.src_budget = src_budget;
    if (inputs) { A.zz_sad = inputs->zz_sad; A.do_ref = inputs->do_ref; A.prehme = inputs->prehme; A.prev_stage_th = inputs->prev_me_stage_based_exit_th; }
    if (params[0].prehme_enabled && A.prehme && (params[0].num_hme_sa_w != 2 || params[0].num_hme_sa_h != 2)) {
        fprintf(stderr, "libsvtav1_hip: svt_hip_hme_chain_batch: the pre-HME replacement needs 2 x 2 search regions (get_worst_quadrant)\n");
        abort();
    }
    const size_t   shm  = 4 * (size_t)(src_budget + win_budget) + 64;
    const uint32_t per0 = params[0].sbs_x * params[0].sbs_y * params[0].num_hme_sa_w * params[0].num_hme_sa_h; // the items of slot 0 (list 0, reference 0)
    static const bool xcd_off = [] { const char* e = getenv("SVT_HIP_HME_XCD"); return e && *e == '0'; }(); // (A/B measurements)
    auto launch = [&](const uint32_t first, const uint32_t end) {
        A.n = end; A.item0 = first;
        A.n_wg = (end - first + 3) / 4;
        A.wg_per_xcd = (!xcd_off && A.n_wg >= 64) ? (A.n_wg + 7) / 8 : 0;
        hipLaunchKernelGGL(hme_chain_kernel, dim3(A.wg_per_xcd ? 8 * A.wg_per_xcd : A.n_wg), dim3(256), shm, (hipStream_t)stream, A);
    };
    if (params[0].l0_mv_th_min && params[0].l0_mv_th_max && params[0].per_ref_area && n > per0) {
        // level-0 areas of the other slots depend on slot 0's level-0 result of the same SB: slot 0 first (whole chain), then the rest
        launch(0, per0);
        launch(per0, n);
    } else {
        launch(0, n);
    }
    SVT_LAUNCH_CHECK();
}

// me_safe_limit_zz_th (init_zz_sad :2416-2436): when both lists' nearest references are (almost) static for the SB, every other reference is dropped
__global__ void ref_safe_limit_kernel(const SvtHipHmeLevelParams G, const uint32_t* __restrict__ zz, const uint32_t th, uint8_t* __restrict__ do_ref) {
    const uint32_t n_sb = G.sbs_x * G.sbs_y, sb = blockIdx.x * blockDim.x + threadIdx.x;
    if (sb >= n_sb || G.n_refs_list0 == 0 || G.n_refs <= G.n_refs_list0) return; // two lists are searched
    if (!(zz[sb] < th && zz[(size_t)G.n_refs_list0 * n_sb + sb] < th)) return;    // zz_sad[0][0], zz_sad[1][0]
    for (uint32_t r = 0; r < G.n_refs; r++)
        if (G.ref_pic_index[r] > 0) do_ref[(size_t)sb * 8 + (r < G.n_refs_list0 ? 0 : 4) + G.ref_pic_index[r]] = 0;
}
void svt_hip_me_ref_safe_limit_batch(const SvtHipHmeLevelParams* plane, const uint32_t* zz_sad, uint32_t safe_limit_zz_th, uint8_t* do_ref, void* stream) {
    svthip::ensure_device();
    const uint32_t n_sb = plane->sbs_x * plane->sbs_y;
    if (!n_sb || !safe_limit_zz_th) return;
    hipLaunchKernelGGL(ref_safe_limit_kernel, dim3((n_sb + 63) / 64), dim3(64), 0, (hipStream_t)stream, *plane, zz_sad, safe_limit_zz_th, do_ref);
    SVT_LAUNCH_CHECK();
}

void svt_hip_me_ref_gate_batch(const SvtHipHmeLevelParams* plane, const uint32_t* zz_sad, uint32_t zz_sad_th, uint32_t zz_sad_pct, int temporal_layer_gt0,
                               uint8_t* do_ref, void* stream) {
    svthip::ensure_device();
    const uint32_t n_sb = plane->sbs_x * plane->sbs_y;
    if (!n_sb || !zz_sad_th) return;
    hipLaunchKernelGGL(ref_prune_pct_kernel, dim3((n_sb + 63) / 64), dim3(64), 0, (hipStream_t)stream, *plane, (const SvtHipPrehmeResult*)nullptr, zz_sad, zz_sad_th,
                       zz_sad_pct, temporal_layer_gt0, do_ref);
    SVT_LAUNCH_CHECK();
}

void svt_hip_prehme_batch(const SvtHipPrehmeParams* params, const uint8_t* src_base, const uint8_t* ref_base, const uint32_t* zz_sad, uint8_t* do_ref,
                          SvtHipPrehmeResult* out, void* stream) {
    svthip::ensure_device();
    const SvtHipHmeLevelParams& G = params->plane;
    const uint32_t n_sb = G.sbs_x * G.sbs_y;
    if (!n_sb || !G.n_refs) return;
    if (params->me_early_exit_th && !zz_sad) { fprintf(stderr, "libsvtav1_hip: svt_hip_prehme_batch: zz_sad is required with me_early_exit_th\n"); abort(); }
    PrehmeArgs A;
    memset(&A, 0, sizeof(A));
    A.P = *params; A.src = src_base; A.ref = ref_base; A.zz_sad = zz_sad; A.do_ref = do_ref; A.out = out;
    const int n_l0 = G.n_refs_list0, n_l1 = (int)G.n_refs - n_l0;
    A.n_l1 = n_l1;
    // per-wave LDS slices for the largest region: 16 x 16 (16 x 8 sub-sampled) block, window (16 + W - 1) x (H + rows - 1)
    const int step = G.sub_sampled ? 2 : 1, bh = 16 / step;
    int win = 0;
    for (int k = 0; k < 2; k++) {
        const int W = params->sa_max_width[k], H = params->sa_max_height[k];
        const int wb = ((((16 + W + 3) >> 2) + 3) & ~1) * 4 * (H + step * (bh - 1));
        win = wb > win ? wb : win;
    }
    A.src_budget = (16 * bh + 15) & ~15;
    A.win_budget = ((win > 36 * 1024 ? 36 * 1024 : win) + 15) & ~15; // 2 slices x 36 KB + sources; larger areas take the plain path
    const dim3 grid(n_sb, n_l0 > n_l1 ? n_l0 : n_l1);
    hipLaunchKernelGGL(prehme_kernel, grid, dim3(256), 2 * (size_t)(A.src_budget + A.win_budget) + 64, (hipStream_t)stream, A);
    SVT_LAUNCH_CHECK();
    if (params->phme_sad_th && do_ref) {
        hipLaunchKernelGGL(ref_prune_pct_kernel, dim3((n_sb + 63) / 64), dim3(64), 0, (hipStream_t)stream, G, (const SvtHipPrehmeResult*)out, (const uint32_t*)nullptr,
                           params->phme_sad_th, (uint32_t)params->phme_sad_pct, (int)params->temporal_layer_gt0, do_ref);
        SVT_LAUNCH_CHECK();
    }
}

size_t svt_hip_me_fullpel_search_workspace(uint32_t n, uint32_t max_w, uint32_t max_h) {
    if (max_w <= (uint32_t)ME_TW && max_h <= (uint32_t)ME_TH) return 0;
    return (size_t)n * SVT_HIP_ME_NUM_BLOCKS * sizeof(uint64_t);
}

void svt_hip_me_fullpel_search_batch(const uint8_t* src_base, const uint8_t* ref_base, const SvtHipMeSearchDesc* descs, uint32_t n,
                                     uint32_t max_w, uint32_t max_h, int sub_sad, uint32_t* best_sad, uint32_t* best_mv,
                                     void* workspace, void* stream) {
    svthip::ensure_device();
    if (n == 0) return;
    if (max_w == 0) max_w = 1;
    if (max_h == 0) max_h = 1;
    if (max_w <= ME_WAVE_MAX_W && max_h <= ME_WAVE_MAX_H) { // small areas: one wave per item (see me_fullpel_wave_kernel)
        const int    G = (int)(max_w + 3) >> 2, pitch = 16 + G <= 18 ? 18 : 26; // dwords a ring load may touch: 16 + G; pitch = 2 mod 8 (bank spread of the 8 x 8 block grid)
        const int    win_dw = pitch * (64 + (int)max_h - 1);
        const size_t shm    = (size_t)4 * win_dw * 4;
        const dim3   grid((n + 3) / 4);
        hipStream_t  st = (hipStream_t)stream;
#define ME_WAVE_LAUNCH(SUBV, PITCHV) hipLaunchKernelGGL(HIP_KERNEL_NAME(me_fullpel_wave_kernel<SUBV, PITCHV>), grid, dim3(256), shm, st, src_base, ref_base, descs, n, win_dw, best_sad, best_mv)
        if (pitch == 18) { if (sub_sad) ME_WAVE_LAUNCH(true, 18); else ME_WAVE_LAUNCH(false, 18); }
        else             { if (sub_sad) ME_WAVE_LAUNCH(true, 26); else ME_WAVE_LAUNCH(false, 26); }
#undef ME_WAVE_LAUNCH
        SVT_LAUNCH_CHECK();
        return;
    }
    const uint32_t tiles_x = (max_w + ME_TW - 1) / ME_TW, tiles_y = (max_h + ME_TH - 1) / ME_TH;
    const int      tw = max_w < (uint32_t)ME_TW ? (int)max_w : ME_TW, th = max_h < (uint32_t)ME_TH ? (int)max_h : ME_TH;
    const int      win_rows = 64 + th - 1;
    const size_t   shmem    = (size_t)(64 * 16 + 88 + ME_PITCH * win_rows) * 4;
    const bool     multi    = tiles_x * tiles_y > 1;
    unsigned long long* keys = multi ? (unsigned long long*)workspace : nullptr;
    if (multi) {
        if (!workspace) { fprintf(stderr, "libsvtav1_hip: me_fullpel_search_batch needs a workspace for areas > 64x32\n"); abort(); }
        HIP_CHECK(hipMemsetAsync(keys, 0xff, (size_t)n * SVT_HIP_ME_NUM_BLOCKS * 8, (hipStream_t)stream));
    }
    if (sub_sad)
        hipLaunchKernelGGL(me_fullpel_kernel<true>, dim3(n, tiles_x * tiles_y), dim3(256), shmem, (hipStream_t)stream, src_base, ref_base,
                           descs, n, tiles_x, best_sad, best_mv, keys);
    else
        hipLaunchKernelGGL(me_fullpel_kernel<false>, dim3(n, tiles_x * tiles_y), dim3(256), shmem, (hipStream_t)stream, src_base, ref_base,
                           descs, n, tiles_x, best_sad, best_mv, keys);
    SVT_LAUNCH_CHECK();
    if (multi) {
        const uint32_t tot = n * SVT_HIP_ME_NUM_BLOCKS;
        hipLaunchKernelGGL(me_finalize_kernel, dim3((tot + 255) / 256), dim3(256), 0, (hipStream_t)stream, descs, n, keys, best_sad, best_mv);
        SVT_LAUNCH_CHECK();
    }
}

// ---- RTCD-signature single-call forms (host pointers) --------------------------------------------------------------
uint32_t svt_nxm_sad_kernel_hip(const uint8_t* src, uint32_t src_stride, const uint8_t* ref, uint32_t ref_stride, uint32_t height,
                                uint32_t width) {
    svthip::HostCall& c = svthip::host_call();
    c.begin();
    const size_t pitch = svthip::align_up(width, 16);
    c.reserve(2 * pitch * height + 1024, 2 * pitch * height + 1024);
    uint8_t*       ds = (uint8_t*)c.dalloc(pitch * height);
    uint8_t*       dr = (uint8_t*)c.dalloc(pitch * height);
    SvtHipSadPair* dp = (SvtHipSadPair*)c.dalloc(sizeof(SvtHipSadPair));
    uint32_t*      dout = (uint32_t*)c.dalloc(4);
    c.up2d(ds, pitch, src, src_stride, width, height);
    c.up2d(dr, pitch, ref, ref_stride, width, height);
    SvtHipSadPair p = {0, 0, (uint32_t)pitch, (uint32_t)pitch};
    c.up(dp, &p, sizeof(p));
    svt_hip_sad_nxm_batch(ds, dr, dp, 1, width, height, dout, c.stream);
    uint32_t out;
    c.down(&out, dout, 4);
    return out;
}

uint32_t svt_aom_sad_16b_kernel_hip(uint16_t* src, uint32_t src_stride, uint16_t* ref, uint32_t ref_stride, uint32_t height,
                                    uint32_t width) {
    svthip::HostCall& c = svthip::host_call();
    c.begin();
    const size_t pitch = svthip::align_up(width * 2, 16);
    c.reserve(2 * pitch * height + 1024, 2 * pitch * height + 1024);
    uint16_t* ds   = (uint16_t*)c.dalloc(pitch * height);
    uint16_t* dr   = (uint16_t*)c.dalloc(pitch * height);
    uint32_t* dout = (uint32_t*)c.dalloc(4);
    c.up2d(ds, pitch, src, src_stride * 2, width * 2, height);
    c.up2d(dr, pitch, ref, ref_stride * 2, width * 2, height);
    hipLaunchKernelGGL(sad_16b_kernel, dim3(1), dim3(64), 0, c.stream, (const uint16_t*)ds, (uint32_t)(pitch / 2), (const uint16_t*)dr,
                       (uint32_t)(pitch / 2), (int)width, (int)height, dout);
    SVT_LAUNCH_CHECK();
    uint32_t out;
    c.down(&out, dout, 4);
    return out;
}

uint32_t svt_aom_sad_wxh_hip(const uint8_t* src, int src_stride, const uint8_t* ref, int ref_stride, int w, int h) {
    return svt_nxm_sad_kernel_hip(src, (uint32_t)src_stride, ref, (uint32_t)ref_stride, (uint32_t)h, (uint32_t)w);
}
void svt_aom_sad_wxh_x4d_hip(const uint8_t* src, int src_stride, const uint8_t* const ref_array[4], int ref_stride, uint32_t* sad_array,
                             int w, int h) {
    svthip::HostCall& c = svthip::host_call();
    c.begin();
    const size_t pitch = svthip::align_up((size_t)w, 16);
    c.reserve(5 * pitch * h + 1024, 5 * pitch * h + 1024);
    uint8_t*       ds = (uint8_t*)c.dalloc(pitch * h);
    uint8_t*       dr = (uint8_t*)c.dalloc(4 * pitch * h);
    SvtHipSadPair* dp = (SvtHipSadPair*)c.dalloc(4 * sizeof(SvtHipSadPair));
    uint32_t*      dout = (uint32_t*)c.dalloc(16);
    c.up2d(ds, pitch, src, src_stride, w, h);
    SvtHipSadPair p[4];
    for (int i = 0; i < 4; i++) {
        c.up2d(dr + i * pitch * h, pitch, ref_array[i], ref_stride, w, h);
        p[i] = {0, (uint64_t)(i * pitch * h), (uint32_t)pitch, (uint32_t)pitch};
    }
    c.up(dp, p, sizeof(p));
    svt_hip_sad_nxm_batch(ds, dr, dp, 4, w, h, dout, c.stream);
    c.down(sad_array, dout, 16);
}
#define SVT_SAD_MXN(m, n)                                                                                                         \
    uint32_t svt_aom_sad##m##x##n##_hip(const uint8_t* src, int src_stride, const uint8_t* ref, int ref_stride) {                \
        return svt_aom_sad_wxh_hip(src, src_stride, ref, ref_stride, m, n);                                                      \
    }                                                                                                                             \
    void svt_aom_sad##m##x##n##x4d_hip(const uint8_t* src, int src_stride, const uint8_t* const ref_array[], int ref_stride,     \
                                       uint32_t* sad_array) {                                                                     \
        svt_aom_sad_wxh_x4d_hip(src, src_stride, ref_array, ref_stride, sad_array, m, n);                                        \
    }
SVT_SAD_MXN(128, 128) SVT_SAD_MXN(128, 64) SVT_SAD_MXN(64, 128) SVT_SAD_MXN(64, 64) SVT_SAD_MXN(64, 32) SVT_SAD_MXN(32, 64)
SVT_SAD_MXN(32, 32) SVT_SAD_MXN(32, 16) SVT_SAD_MXN(16, 32) SVT_SAD_MXN(16, 16) SVT_SAD_MXN(16, 8) SVT_SAD_MXN(8, 16)
SVT_SAD_MXN(8, 8) SVT_SAD_MXN(8, 4) SVT_SAD_MXN(4, 8) SVT_SAD_MXN(4, 4) SVT_SAD_MXN(4, 16) SVT_SAD_MXN(16, 4)
SVT_SAD_MXN(8, 32) SVT_SAD_MXN(32, 8) SVT_SAD_MXN(16, 64) SVT_SAD_MXN(64, 16)

void svt_sad_loop_kernel_hip(uint8_t* src, uint32_t src_stride, uint8_t* ref, uint32_t ref_stride, uint32_t block_height,
                             uint32_t block_width, uint64_t* best_sad, int16_t* x_search_center, int16_t* y_search_center,
                             uint32_t src_stride_raw, uint8_t skip_search_line, int16_t search_area_width, int16_t search_area_height) {
    *best_sad = 0xffffff;
    if (search_area_width <= 0 || search_area_height <= 0) return;
    svthip::HostCall& c = svthip::host_call();
    c.begin();
    // Upload the exact byte ranges the reference reads. Source rows: block_height rows src_stride apart.
    // Reference: rows r = yy*raw + y*ref_stride, yy < H, y < bh; upload the covering span as one rectangle of
    // `raw`-strided lines when ref_stride is a multiple of raw (always true: ref_stride is raw or 2*raw).
    const size_t sp = svthip::align_up(block_width, 16);
    const size_t ww = (size_t)block_width + search_area_width - 1;
    const size_t rp = svthip::align_up(ww, 16);
    const uint32_t step  = ref_stride / src_stride_raw; // 1 (full) or 2 (sub-sampled HME)
    const size_t   lines = (size_t)(search_area_height - 1) + (size_t)(block_height - 1) * step + 1;
    c.reserve(sp * block_height + rp * lines + 4096, sp * block_height + rp * lines + 4096);
    uint8_t* ds = (uint8_t*)c.dalloc(sp * block_height);
    uint8_t* dr = (uint8_t*)c.dalloc(rp * lines);
    SvtHipSadLoopDesc*   dd = (SvtHipSadLoopDesc*)c.dalloc(sizeof(SvtHipSadLoopDesc));
    SvtHipSadLoopResult* dres = (SvtHipSadLoopResult*)c.dalloc(sizeof(SvtHipSadLoopResult));
    uint64_t*            dk = (uint64_t*)c.dalloc(8);
    c.up2d(ds, sp, src, src_stride, block_width, block_height);
    c.up2d(dr, rp, ref, src_stride_raw, ww, lines);
    SvtHipSadLoopDesc d;
    memset(&d, 0, sizeof(d));
    d.src_stride = (uint32_t)sp; d.ref_stride = (uint32_t)(rp * step); d.src_stride_raw = (uint32_t)rp;
    d.block_width = (uint16_t)block_width; d.block_height = (uint16_t)block_height;
    d.search_area_width = search_area_width; d.search_area_height = search_area_height; d.skip_search_line = skip_search_line;
    c.up(dd, &d, sizeof(d));
    svt_hip_sad_loop_batch(ds, dr, dd, 1, (uint32_t)search_area_width, (uint32_t)search_area_height, block_width, block_height, (int)step, dres, dk,
                           c.stream);
    SvtHipSadLoopResult r;
    c.down(&r, dres, sizeof(r));
    *best_sad = r.best_sad;
    if (r.valid) { *x_search_center = r.x_search_center; *y_search_center = r.y_search_center; }
}

void svt_ext_all_sad_calculation_8x8_16x16_hip(uint8_t* src, uint32_t src_stride, uint8_t* ref, uint32_t ref_stride, uint32_t mv,
                                               uint32_t* p_best_sad_8x8, uint32_t* p_best_sad_16x16, uint32_t* p_best_mv8x8,
                                               uint32_t* p_best_mv16x16, uint32_t p_eight_sad16x16[16][8],
                                               uint32_t p_eight_sad8x8[64][8], bool sub_sad) {
    (void)p_eight_sad8x8; // unused by the reference as well (motion_estimation.c:218)
    svthip::HostCall& c = svthip::host_call();
    c.begin();
    c.reserve(64 * 64 + 64 * 80 + 4096, 64 * 64 + 64 * 80 + 4096);
    uint8_t*  ds = (uint8_t*)c.dalloc(64 * 64);
    uint8_t*  dr = (uint8_t*)c.dalloc(64 * 80);
    uint32_t* dv = (uint32_t*)c.dalloc((64 + 16 + 64 + 16 + 128) * 4);
    uint32_t *b8 = dv, *b16 = dv + 64, *m8 = dv + 80, *m16 = dv + 144, *e16 = dv + 160;
    c.up2d(ds, 64, src, src_stride, 64, 64);
    c.up2d(dr, 80, ref, ref_stride, 71, 64);
    c.up(b8, p_best_sad_8x8, 64 * 4); c.up(b16, p_best_sad_16x16, 16 * 4);
    c.up(m8, p_best_mv8x8, 64 * 4);   c.up(m16, p_best_mv16x16, 16 * 4);
    hipLaunchKernelGGL(ext_all_sad_kernel, dim3(1), dim3(512), 0, c.stream, (const uint8_t*)ds, 64u, (const uint8_t*)dr, 80u, mv, b8, b16, m8,
                       m16, e16, sub_sad ? 1 : 0);
    SVT_LAUNCH_CHECK();
    uint32_t h[64 + 16 + 64 + 16 + 128];
    c.down(h, dv, sizeof(h));
    memcpy(p_best_sad_8x8, h, 64 * 4); memcpy(p_best_sad_16x16, h + 64, 16 * 4);
    memcpy(p_best_mv8x8, h + 80, 64 * 4); memcpy(p_best_mv16x16, h + 144, 16 * 4);
    memcpy(p_eight_sad16x16, h + 160, 128 * 4);
}

void svt_ext_eight_sad_calculation_32x32_64x64_hip(uint32_t p_sad16x16[16][8], uint32_t* p_best_sad_32x32, uint32_t* p_best_sad_64x64,
                                                   uint32_t* p_best_mv32x32, uint32_t* p_best_mv64x64, uint32_t mv,
                                                   uint32_t p_sad32x32[4][8]) {
    svthip::HostCall& c = svthip::host_call();
    c.begin();
    c.reserve(4096, 4096);
    uint32_t* dv = (uint32_t*)c.dalloc((128 + 4 + 1 + 4 + 1 + 32) * 4);
    uint32_t *s16 = dv, *b32 = dv + 128, *b64 = dv + 132, *m32 = dv + 133, *m64 = dv + 137, *s32 = dv + 138;
    c.up(s16, p_sad16x16, 128 * 4); c.up(b32, p_best_sad_32x32, 16); c.up(b64, p_best_sad_64x64, 4);
    c.up(m32, p_best_mv32x32, 16);  c.up(m64, p_best_mv64x64, 4);
    hipLaunchKernelGGL(ext_eight_32_64_kernel, dim3(1), dim3(64), 0, c.stream, (const uint32_t*)s16, b32, b64, m32, m64, mv, s32);
    SVT_LAUNCH_CHECK();
    uint32_t h[170];
    c.down(h, dv, sizeof(h));
    memcpy(p_best_sad_32x32, h + 128, 16); p_best_sad_64x64[0] = h[132];
    memcpy(p_best_mv32x32, h + 133, 16);   p_best_mv64x64[0] = h[137];
    memcpy(p_sad32x32, h + 138, 32 * 4);
}

void svt_ext_sad_calculation_8x8_16x16_hip(uint8_t* src, uint32_t src_stride, uint8_t* ref, uint32_t ref_stride, uint32_t* p_best_sad_8x8,
                                           uint32_t* p_best_sad_16x16, uint32_t* p_best_mv8x8, uint32_t* p_best_mv16x16, uint32_t mv,
                                           uint32_t* p_sad16x16, uint32_t* p_sad8x8, bool sub_sad) {
    svthip::HostCall& c = svthip::host_call();
    c.begin();
    c.reserve(4096, 4096);
    uint8_t*  ds = (uint8_t*)c.dalloc(16 * 16);
    uint8_t*  dr = (uint8_t*)c.dalloc(16 * 16);
    uint32_t* dv = (uint32_t*)c.dalloc(16 * 4);
    c.up2d(ds, 16, src, src_stride, 16, 16);
    c.up2d(dr, 16, ref, ref_stride, 16, 16);
    uint32_t h[16] = {0};
    memcpy(h, p_best_sad_8x8, 16); h[4] = p_best_sad_16x16[0]; memcpy(h + 5, p_best_mv8x8, 16); h[9] = p_best_mv16x16[0];
    c.up(dv, h, sizeof(h));
    hipLaunchKernelGGL(ext_one_8_16_kernel, dim3(1), dim3(64), 0, c.stream, (const uint8_t*)ds, 16u, (const uint8_t*)dr, 16u, dv, dv + 4,
                       dv + 5, dv + 9, mv, dv + 10, dv + 11, sub_sad ? 1 : 0);
    SVT_LAUNCH_CHECK();
    c.down(h, dv, sizeof(h));
    memcpy(p_best_sad_8x8, h, 16); p_best_sad_16x16[0] = h[4]; memcpy(p_best_mv8x8, h + 5, 16); p_best_mv16x16[0] = h[9];
    p_sad16x16[0] = h[10]; memcpy(p_sad8x8, h + 11, 16);
}

void svt_ext_sad_calculation_32x32_64x64_hip(uint32_t* p_sad16x16, uint32_t* p_best_sad_32x32, uint32_t* p_best_sad_64x64,
                                             uint32_t* p_best_mv32x32, uint32_t* p_best_mv64x64, uint32_t mv, uint32_t* p_sad32x32) {
    svthip::HostCall& c = svthip::host_call();
    c.begin();
    c.reserve(4096, 4096);
    uint32_t* dv = (uint32_t*)c.dalloc(32 * 4);
    uint32_t  h[32] = {0};
    memcpy(h, p_sad16x16, 64); memcpy(h + 16, p_best_sad_32x32, 16); h[20] = p_best_sad_64x64[0];
    memcpy(h + 21, p_best_mv32x32, 16); h[25] = p_best_mv64x64[0];
    c.up(dv, h, sizeof(h));
    hipLaunchKernelGGL(ext_one_32_64_kernel, dim3(1), dim3(64), 0, c.stream, (const uint32_t*)dv, dv + 16, dv + 20, dv + 21, dv + 25, mv,
                       dv + 26);
    SVT_LAUNCH_CHECK();
    c.down(h, dv, sizeof(h));
    memcpy(p_best_sad_32x32, h + 16, 16); p_best_sad_64x64[0] = h[20]; memcpy(p_best_mv32x32, h + 21, 16); p_best_mv64x64[0] = h[25];
    memcpy(p_sad32x32, h + 26, 16);
}

void svt_initialize_buffer_32bits_hip(uint32_t* pointer, uint32_t count128, uint32_t count32, uint32_t value) {
    const uint32_t n = count128 * 4 + count32;
    if (n == 0) return;
    svthip::HostCall& c = svthip::host_call();
    c.begin();
    c.reserve((size_t)n * 4 + 1024, (size_t)n * 4 + 1024);
    uint32_t* d = (uint32_t*)c.dalloc((size_t)n * 4);
    hipLaunchKernelGGL(fill_u32_kernel, dim3((n + 255) / 256), dim3(256), 0, c.stream, d, n, value);
    SVT_LAUNCH_CHECK();
    c.down(pointer, d, (size_t)n * 4);
}

} // extern "C"

SVT_HIP_DEFINE_WARM(sad) // (svt_hip_warmup loads this translation unit's code object at encoder initialisation: svt_hip_common.h)
