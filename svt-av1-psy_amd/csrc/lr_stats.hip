// lr_stats.hip -- Wiener search statistics on the matrix cores: svt_av1_compute_stats / _highbd ->
// svt_av1_compute_stats_c / svt_av1_compute_stats_highbd_c (Codec/restoration_pick.c:659-745).
//
// This is the one kernel of the path that IS a dense contraction: with z_t(p) = dgd(p + offset_t) - avg for the win^2 taps and
// z_x(p) = src(p) - avg, the outputs are H = Z^T Z (win^2 x win^2) and M = Z^T z_x, sums over every pixel p of the restoration unit.
// It must be exact (int64 in the reference), so the operands are integers: every centred sample v (|v| < 4096) is split into two int8
// digits, v = 128 h + l with h = v >> 7 in [-32, 31] and l = v & 127, and
//     sum z_a z_b = 16384 sum h_a h_b + 128 (sum h_a l_b + sum l_a h_b) + sum l_a l_b
// is three int8 matrix products accumulated in int32 by v_mfma_i32_16x16x64_i8 (exact: a workgroup sees 4096 pixels, 4096 * 127^2 < 2^31)
// and combined in int64.  Taps + the source column are padded to 64 (win 7) or 32 (win 5) columns = 4 or 2 groups of 16; HH and LL are
// symmetric (upper tile triangle only), HL is not.
//
// MFMA operand for (tap t, 64 consecutive pixels of a row): lane (t & 15, kg = lane >> 4) holds the 16 bytes of digit plane samples of pixels
// 16 kg .. 16 kg + 15 displaced by the tap's offset (dx, dy).  The tile's digit planes are kept in LDS once per horizontal displacement dx (WIN copies, copy dx holding
// plane[c + dx] at column c), so those sixteen bytes are ONE 16-byte-aligned ds_read_b128 of copy dx, row + dy -- no funnel shifts, masks or address arithmetic beyond one
// add in the K loop (until round 6 every operand was five dwords + eight v_alignbyte: the VALU work of the loop kept the matrix pipe at 12 % busy).  The copies are made
// when the tile is staged: 7 shifted dwords per staged dword, once per tile instead of once per (tap, chunk).
// The same registers serve as A (rows of Z^T) and as B (columns of Z): both use the lane <-> (tap, pixel) assignment, so the sum over K
// visits every pixel exactly once whatever order the hardware walks K in.
#include "svt_hip_common.h"
#include "../../include/svtav1_hip.h"

#include <type_traits>

namespace {

typedef int i32x4 __attribute__((vector_size(16)));

constexpr int TR = 8, TC = 256;       // pixel tile staged at a time: 8 rows x 256 columns = 32 chunks of 64 pixels, 8 chunks per wave
constexpr int BANDS = 8;              // a workgroup walks 8 such tiles (64 rows) with its accumulators in registers: 16384 pixels, still exact in
                                      // int32 (16384 * 127^2 < 2^31), a quarter of the merge traffic
constexpr int NR = TR + 6;            // plane rows of a tile: 3 rows of halo above and below (win 7; smaller windows leave the outer ones unused)
constexpr int PP = TC + 16;           // pitch of the staging rows (bytes); tile pixel (r, c) sits at byte (r + 3) * PP + c + 4
constexpr int STAGE_PLANE = NR * PP;  // one digit of the unshifted tile with its halo columns
constexpr int SRC_PLANE = TR * TC;    // src digit plane: [kg][row][chunk] cells of 16 bytes
// The displacement copies as the MFMA operand reads see them.  A ds_read_b128 is served in four groups of sixteen lanes, each group = sixteen DIFFERENT taps (lane & 15) at
// two of the four pixel quarters kg (lane >> 4) of a 64-pixel chunk, and a group is conflict-free when its sixteen 16-byte cells fall into the sixteen different 16-byte slots
// of the 256-byte bank row.  So: one sub-plane per kg at a multiple of 256 bytes (kg never moves the slot); inside it 64-byte rows (four chunk cells), row index
// Rr = A * plane_row + B * copy with (A, B) chosen so that Rr = (odd constant) * tap (mod 16) -- sixteen consecutive taps, sixteen different Rr mod 16; and the cell of
// chunk ch stored at position ch ^ ((Rr >> 2) & 3), so that slot = 4 (Rr & 3) + (ch ^ ((Rr >> 2) & 3)) runs through all sixteen values.  (With all pitches 256 -- the
// layout until round 6 -- every lane of a group hit one of two slots: eight LDS passes per group instead of one.)
template <int WIN> struct CopyLayout {
    static constexpr int A = WIN == 7 ? 7 : 1;                        // WIN 7: the seven copies of a plane row are adjacent rows (7 * 7 = 1 mod 16)
    static constexpr int B = WIN == 7 ? 1 : (WIN == 5 ? 21 : 19);     // WIN 5 / 3: copy-major with a row count = WIN (mod 16)
    static constexpr int ROWS = WIN == 7 ? NR * 7 : (WIN - 1) * B + NR;
    static constexpr int KP = (ROWS * 64 + 255) / 256 * 256;          // one kg sub-plane
    static constexpr int DP = 4 * KP;                                 // one digit
};
constexpr int ZERO_BYTES = 64;        // four zero cells: what the padding columns of the last operand group read
template <int WIN> constexpr int lds_bytes() { return 2 * STAGE_PLANE + 2 * CopyLayout<WIN>::DP + 4 * SRC_PLANE + ZERO_BYTES; } // (two sets of source planes: see the band loop)

__device__ __forceinline__ long long wave_sum_ll(long long v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
__device__ __forceinline__ int rd(const void* p, const int is16, const size_t off) { return is16 ? ((const uint16_t*)p)[off] : ((const uint8_t*)p)[off]; }

// find_average / find_average_highbd (restoration_pick.h): floor(sum / count) of the degraded unit.  Workgroup (b, unit) sums band b of SUM_BANDS row bands and STORES
// its partial in a lower-triangle slot of H (entry (1 + b, 0): the matrix kernel only touches the upper triangle, the finalize kernel overwrites the lower one with the
// mirror image), so there is no atomic and nothing to clear first; the same workgroups zero the unit's M and the rest of its H -- the two fill launches that used to
// precede this kernel are gone.  Rows are read as aligned 16-byte words where a word lies wholly inside the row (per-sample loads at the two ragged ends only).
constexpr int SUM_BANDS = 8;
struct alignas(16) Word16 { uint32_t x, y, z, w; };
__device__ __forceinline__ int sum_slot(const int b, const int w2) { return (1 + b) * w2; }

__global__ __launch_bounds__(256) void stats_sum_kernel(const void* dgd, const SvtHipRect* rects, const int dgd_stride, const int is16, const int w2, const int rows_per_band,
                                                        long long* Mout, long long* Hout) {
    __shared__ long long wsum[4];
    const int        tid = threadIdx.x, b = blockIdx.x, unit = blockIdx.y;
    const SvtHipRect R   = rects[unit];
    const int        W = R.h_end - R.h_start, Hh = R.v_end - R.v_start, r0 = b * rows_per_band;
    long long*       H = Hout + (size_t)unit * 49 * 49;
    {   // this workgroup's share of the clearing: every entry but the SUM_BANDS partial slots
        const int n = w2 * w2, per = (n + SUM_BANDS - 1) / SUM_BANDS, hi = (b + 1) * per < n ? (b + 1) * per : n;
        for (int e = b * per + tid; e < hi; e += 256)
            if (!(e % w2 == 0 && e >= w2 && e <= SUM_BANDS * w2)) H[e] = 0;
        if (b == 0 && tid < w2) Mout[(size_t)unit * 49 + tid] = 0;
    }
    const int rows = W <= 0 ? 0 : (Hh - r0 < rows_per_band ? Hh - r0 : rows_per_band);
    const int px = is16 ? 2 : 1, l = tid & 63;
    auto word_sum = [&](const Word16 q) -> unsigned { // <= 8 samples of 16 bits or 16 of 8 bits
        return is16 ? (q.x & 0xffff) + (q.x >> 16) + (q.y & 0xffff) + (q.y >> 16) + (q.z & 0xffff) + (q.z >> 16) + (q.w & 0xffff) + (q.w >> 16)
                    : __builtin_amdgcn_sad_u8(q.w, 0u, __builtin_amdgcn_sad_u8(q.z, 0u, __builtin_amdgcn_sad_u8(q.y, 0u, __builtin_amdgcn_sad_u8(q.x, 0u, 0u))));
    };
    unsigned long long s = 0;
    constexpr int RU = 4; // rows in flight per wave: a row is one aligned 16-byte load per lane (+ one sample per lane at the two ragged ends), issued before any is summed
    for (int rb = tid >> 6; rb < rows; rb += 4 * RU) {
        uintptr_t a0[RU], v0[RU], v1[RU], a1[RU];
        Word16    q[RU];
        unsigned  e[RU];
#pragma unroll
        for (int u = 0; u < RU; u++) {
            const int r = rb + 4 * u;
            a0[u] = (uintptr_t)dgd + (size_t)px * (size_t)((long long)(R.v_start + r0 + (r < rows ? r : rows - 1)) * dgd_stride + R.h_start);
            a1[u] = r < rows ? a0[u] + (size_t)px * W : a0[u]; // (a row beyond the band: empty)
            v0[u] = (a0[u] + 15) & ~(uintptr_t)15;
            v1[u] = a1[u] & ~(uintptr_t)15; // whole 16-byte words of the row: [v0, v1)
            q[u]  = Word16{0, 0, 0, 0};
            e[u]  = 0;
            const uintptr_t a = v0[u] + 16 * (uintptr_t)l;
            if (a < v1[u]) q[u] = *(const Word16*)a;
            // ragged ends (< 16 bytes each; everything when the row holds no whole word): lane k < head takes sample k, the next `tail` lanes the samples from v1 on
            const uintptr_t hend = v0[u] < a1[u] ? v0[u] : a1[u], tbeg = v1[u] > hend ? v1[u] : hend;
            const int       head = (int)((hend - a0[u]) / px), tail = (int)((a1[u] - tbeg) / px);
            if (l < head) e[u] = is16 ? ((const uint16_t*)a0[u])[l] : ((const uint8_t*)a0[u])[l];
            else if (l - head < tail) e[u] = is16 ? ((const uint16_t*)tbeg)[l - head] : ((const uint8_t*)tbeg)[l - head];
        }
#pragma unroll
        for (int u = 0; u < RU; u++) {
            unsigned acc = word_sum(q[u]) + e[u];
            for (uintptr_t a = v0[u] + 16 * (uintptr_t)(l + 64); a < v1[u]; a += 16 * 64) acc += word_sum(*(const Word16*)a); // (rows wider than 64 words)
            s += acc;
        }
    }
    s = (unsigned long long)wave_sum_ll((long long)s);
    if ((tid & 63) == 0) wsum[tid >> 6] = (long long)s;
    __syncthreads();
    if (tid == 0) H[sum_slot(b, w2)] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__device__ __forceinline__ uint32_t pack_digits_hi(const int v0, const int v1, const int v2, const int v3) {
    return (uint32_t)((v0 >> 7) & 255) | ((uint32_t)((v1 >> 7) & 255) << 8) | ((uint32_t)((v2 >> 7) & 255) << 16) | ((uint32_t)((v3 >> 7) & 255) << 24);
}
__device__ __forceinline__ uint32_t pack_digits_lo(const int v0, const int v1, const int v2, const int v3) {
    return (uint32_t)(v0 & 127) | ((uint32_t)(v1 & 127) << 8) | ((uint32_t)(v2 & 127) << 16) | ((uint32_t)(v3 & 127) << 24);
}


// Four consecutive samples of a picture row as the digit planes want them.  The 8-byte (4-byte at 8 bits) load is ONE instruction whatever the alignment, it is issued
// unconditionally -- a quad that sticks out of the readable range [lo, hi) of its row is loaded from the nearest window inside the range and shifted back, a quad that lies
// wholly outside (or in a row outside the band) from the clamped position and dropped -- so a tile's loads are all in flight together.  (Until round 6 every sample was a
// guarded 2-byte load and the compiler placed an s_waitcnt vmcnt(0) behind each: 24 memory round trips in series per tile.)
struct QuadWhere { int col; int shift; uint32_t keep; }; // first column actually loaded, its distance from the quad's own first column, byte mask of the wanted samples
__device__ __forceinline__ QuadWhere quad_where(const int tc0, const int lo, const int hi, const bool row_ok) {
    QuadWhere q;
    const int top = hi - 4;
    q.col = hi - lo < 4 ? tc0 : (tc0 < lo ? lo : (tc0 > top ? top : tc0)); // (a range narrower than a quad has no window: its samples are loaded one by one, load_narrow)
    q.shift = q.col - tc0;                                            // > 0: loaded window starts right of the quad, < 0: left of it
    const int e0 = lo - tc0 > 0 ? lo - tc0 : 0, e1 = hi - tc0 < 4 ? hi - tc0 : 4; // wanted samples e0 .. e1 - 1
    q.keep = (row_ok && e1 > e0) ? (e1 - e0 >= 4 ? ~0u : (((1u << (8 * (e1 - e0))) - 1u) << (8 * e0))) : 0u;
    return q;
}
// digits of four centred samples held as two packed pairs (x = [p0, p1], y = [p2, p3] in the order given by `sel`): v = 128 h + l, h = v >> 7 (arithmetic), l = v & 127;
// v_pk_sub_i16 / v_pk_ashrrev_i16 / v_and / v_perm: eight instructions per quad
typedef short s16x2 __attribute__((vector_size(4)));
__device__ __forceinline__ s16x2    as_pk(const uint32_t v) { s16x2 r; __builtin_memcpy(&r, &v, 4); return r; }
__device__ __forceinline__ uint32_t as_u32(const s16x2 v) { uint32_t r; __builtin_memcpy(&r, &v, 4); return r; }
__device__ __forceinline__ void pair_digits(const uint32_t x, const uint32_t y, const int avg, const uint32_t sel, const uint32_t keep, uint32_t& hi, uint32_t& lo) {
    const short a16 = (short)avg;
    const s16x2 av = {a16, a16}, cx = as_pk(x) - av, cy = as_pk(y) - av;
    hi = __builtin_amdgcn_perm(as_u32(cy >> 7), as_u32(cx >> 7), sel) & keep;
    lo = __builtin_amdgcn_perm(as_u32(cy) & 0x007f007fu, as_u32(cx) & 0x007f007fu, sel) & keep;
}
template <bool IS16> struct RawQuad;
template <> struct RawQuad<true> {
    uint32_t a, b;
    __device__ __forceinline__ void load(const void* base, const size_t off) { const svt_u32x2_a1 v = svt_hip_global_load_x2((const uint16_t*)base + off); a = v[0]; b = v[1]; }
    __device__ __forceinline__ void load_narrow(const void* base, const size_t row_off, const int tc0, const int lo, const int hi) { // sample e from column clamp(tc0 + e)
        uint32_t v[4];
#pragma unroll
        for (int e = 0; e < 4; e++) { const int c = tc0 + e < lo ? lo : (tc0 + e > hi - 1 ? hi - 1 : tc0 + e); v[e] = ((const uint16_t*)base)[(long long)row_off + c]; }
        a = v[0] | (v[1] << 16); b = v[2] | (v[3] << 16);
    }
    __device__ __forceinline__ void digits(const QuadWhere q, const int avg, uint32_t& hi, uint32_t& lo) const {
        uint32_t x = a, y = b;
        if (q.shift != 0 && q.keep != 0) { // a quad at the edge of the readable range (one or two lanes of a tile row): move the loaded window back over the quad
            unsigned long long w = ((unsigned long long)b << 32) | a;
            w = q.shift > 0 ? w << (16 * q.shift) : w >> (16 * -q.shift);
            x = (uint32_t)w; y = (uint32_t)(w >> 32);
        }
        pair_digits(x, y, avg, 0x06040200u, q.keep, hi, lo); // x = [s0, s1], y = [s2, s3]
    }
};
template <> struct RawQuad<false> {
    uint32_t a;
    __device__ __forceinline__ void load(const void* base, const size_t off) { uint32_t v; __builtin_memcpy(&v, (const uint8_t*)base + off, 4); a = v; }
    __device__ __forceinline__ void load_narrow(const void* base, const size_t row_off, const int tc0, const int lo, const int hi) {
        a = 0;
#pragma unroll
        for (int e = 0; e < 4; e++) { const int c = tc0 + e < lo ? lo : (tc0 + e > hi - 1 ? hi - 1 : tc0 + e); a |= (uint32_t)((const uint8_t*)base)[(long long)row_off + c] << (8 * e); }
    }
    __device__ __forceinline__ void digits(const QuadWhere q, const int avg, uint32_t& hi, uint32_t& lo) const {
        uint32_t w = a;
        if (q.shift != 0 && q.keep != 0) w = q.shift > 0 ? w << (8 * q.shift) : w >> (8 * -q.shift);
        pair_digits(w & 0x00ff00ffu, (w >> 8) & 0x00ff00ffu, avg, 0x06020400u, q.keep, hi, lo); // x = [s0, s2], y = [s1, s3]
    }
};

// WIN 7: 49 taps + source = 50 columns in NG = 4 groups of 16; WIN 5: 26 columns, NG = 2; WIN 3: 10 columns, NG = 1.  With the window a template parameter every
// group but the last is known to hold taps only (16 (NG - 1) <= WIN^2), so only the last group keeps per-lane plane / pitch / mask registers.
template <int WIN, bool IS16>
__global__ __launch_bounds__(256, 2) void stats_mfma_kernel(const void* dgd, const void* src, const SvtHipRect* rects, const int dgd_stride, const int src_stride,
                                                         long long* Mout, long long* Hout, const int nbands) {
    using L = CopyLayout<WIN>;
    constexpr int win = WIN, NG = (WIN * WIN + 1 + 15) / 16;
    static_assert(16 * (NG - 1) <= WIN * WIN, "only the last group may hold the source column or padding");
    // accumulator tiles: HH, LL and X over the upper triangle of group pairs, X(ga, gb) = H_ga L_gb^T (+ L_ga H_gb^T off the diagonal: both cross terms of an
    // entry carry the same weight, so they share an accumulator -- 30 tiles instead of 36 for WIN 7, same 36 MFMAs per chunk)
    constexpr int NTRI = NG * (NG + 1) / 2, NTILE = 3 * NTRI, NMFMA = 2 * NTRI + NG * NG;
    HIP_DYNAMIC_SHARED(uint32_t, smem)
    // LDS: [staging hi][staging lo] (unshifted tile rows with halo columns) | [hi copies: 4 kg sub-planes][lo copies] | [src hi][src lo]; later reused as int32 [NTILE][256]
    uint8_t* stage  = (uint8_t*)smem;
    uint8_t* planes = stage + 2 * STAGE_PLANE;
    constexpr int LO = L::DP, SRC_HI = 2 * L::DP, SRC_LO = SRC_HI + SRC_PLANE;
    const int        tid = threadIdx.x, l = tid & 63, wv = tid >> 6;
    const int        unit = blockIdx.y, bgroup = blockIdx.z; // band groups are the SLOWEST grid dimension: the empty workgroups of short units sit at the end of the dispatch
    const SvtHipRect R = rects[unit];
    constexpr int w2 = win * win, hw = win >> 1;
    const int W = R.h_end - R.h_start, Hh = R.v_end - R.v_start;
    const int c0 = blockIdx.x * TC, rb0 = bgroup * TR * nbands; // origin of the workgroup's band group (nbands tiles of 8 rows, <= BANDS) inside the unit
    if (c0 >= W || rb0 >= Hh) return;
    const int tw = W - c0 < TC ? W - c0 : TC;
    long long* H = Hout + (size_t)unit * 49 * 49;
    long long* M = Mout + (size_t)unit * 49;
#ifdef SVT_HIP_STATS_CENSUS // (measurement build only, tools/stats_census.py: when and where each workgroup ran, and thread 0's time per phase; the finalize launch is skipped, the results are not usable)
    const unsigned long long census_t0 = wall_clock64();
    long long census_ph[6] = {0, 0, 0, 0, 0, 0}, census_c = clock64(); // prologue, top barrier + digits, copies, prefetch issue, K loop, merge
#define CENSUS_PHASE(k) { const long long now_ = clock64(); census_ph[k] += now_ - census_c; census_c = now_; }
#else
#define CENSUS_PHASE(k)
#endif

    // ---- per-lane operand addressing: group g -> tap t = 16 g + (l & 15); kg = l >> 4 selects pixels 16 kg .. 16 kg + 15 of a chunk ----
    // A wave owns tile rows 2 wv and 2 wv + 1.  Operand of (row 2 wv + ri, chunk ch) = the 16-byte cell (see CopyLayout) at  obase[g] + ri * ostep + ((16 ch) ^ oswz[g][ri])
    // -- one v_xad_u32 per operand in the K loop; hi and lo digit of a cell differ by a constant (the ds_read offset field).  The last group also holds the source
    // column (its own [kg][row][chunk] planes, no swizzle) and padding columns, which read the zero cells: no masks.
    constexpr int SRC_SET = 2 * SRC_PLANE, ZERO = SRC_HI + 2 * SRC_SET; // source planes of even tiles | of odd tiles | zero cells
    int obase[NG], oswz[NG][2], last_step = 64 * L::A, last_lo = LO, src_flip = 0; // src_flip: the source-column lanes alternate between the two sets of source planes
#pragma unroll
    for (int g = 0; g < NG; g++) {
        const int t = 16 * g + (l & 15), kg = l >> 4;
        if (g < NG - 1 || t < w2) { // tap index = (dx + hw) * win + (dy + hw)   (restoration_pick.c:673-679: k over columns, l over rows)
            const int dxi = t / win, dyi = t % win, Rr = L::A * (2 * wv + 3 - hw + dyi) + L::B * dxi;
            obase[g]   = kg * L::KP + 64 * Rr;
            oswz[g][0] = 16 * ((Rr >> 2) & 3);
            oswz[g][1] = 16 * (((Rr + L::A) >> 2) & 3);
        } else if (t == w2) { // the source column
            obase[g] = SRC_HI + kg * (TR * 64) + 64 * 2 * wv; oswz[g][0] = oswz[g][1] = 0; last_step = 64; last_lo = SRC_PLANE; src_flip = SRC_SET;
        } else { // padding
            obase[g] = ZERO; oswz[g][0] = oswz[g][1] = 0; last_step = 0; last_lo = 0;
        }
    }
    if (tid < ZERO_BYTES / 4) ((uint32_t*)(planes + ZERO))[tid] = 0; // (made visible by the first tile's barriers)
    i32x4 accHH[NTRI], accLL[NTRI], accX[NTRI];
#pragma unroll
    for (int i = 0; i < NTRI; i++) { accHH[i] = i32x4{0, 0, 0, 0}; accLL[i] = i32x4{0, 0, 0, 0}; accX[i] = i32x4{0, 0, 0, 0}; }

    // ---- the tile loads: quad i = tid + 256 k of the degraded tile (plane row i / SLOTS, staging dword i % SLOTS = tile columns 4 s - 4 ..), quad j of the source tile ----
    constexpr int SLOTS = PP / 4, NIT = (NR * SLOTS + 255) / 256, SSLOTS = TC / 4, SNIT = (TR * SSLOTS + 255) / 256;
    RawQuad<IS16> rawd[NIT], raws[SNIT];
    // readable columns: [-hw, tw + hw) of the degraded rows, [0, tw) of the source rows (a tile of fewer than four columns loads sample by sample)
    auto issue = [&](const int r0, const int th, const int tid) { // (tid: the caller's per-tile copy, see btid)
#pragma unroll
        for (int k = 0; k < NIT; k++) {
            const int i = tid + 256 * k, r = i / SLOTS, s = i - r * SLOTS;
            int tr = r - 3;
            tr = tr < -hw ? -hw : (tr > th + hw - 1 ? th + hw - 1 : tr);
            const size_t row_off = (size_t)((long long)(R.v_start + r0 + tr) * dgd_stride + (R.h_start + c0));
            if (tw + 2 * hw >= 4) rawd[k].load(dgd, row_off + quad_where(4 * s - 4, -hw, tw + hw, true).col);
            else rawd[k].load_narrow(dgd, row_off, 4 * s - 4, -hw, tw + hw);
        }
#pragma unroll
        for (int k = 0; k < SNIT; k++) {
            const int i = tid + 256 * k, r = i / SSLOTS, s = i - r * SSLOTS;
            const int tr = r > th - 1 ? th - 1 : r;
            const size_t row_off = (size_t)((long long)(R.v_start + r0 + tr) * src_stride + (R.h_start + c0));
            if (tw >= 4) raws[k].load(src, row_off + quad_where(4 * s, 0, tw, true).col);
            else raws[k].load_narrow(src, row_off, 4 * s, 0, tw);
        }
    };
    issue(rb0, Hh - rb0 < TR ? Hh - rb0 : TR, tid);
    unsigned long long total = 0; // sum of the degraded unit: stats_sum_kernel's partials (read behind the first tile's loads: one memory round trip, not two in series)
#pragma unroll
    for (int b = 0; b < SUM_BANDS; b++) total += (unsigned long long)H[sum_slot(b, w2)];
    const int avg = (int)(total / (unsigned long long)((long long)W * Hh));

    CENSUS_PHASE(0)
    for (int band = 0; band < nbands; band++) {
    const int r0 = rb0 + band * TR;
    if (r0 >= Hh) break;
    const int th = Hh - r0 < TR ? Hh - r0 : TR;
    // Two barriers per tile: what a wave writes ahead of the first one -- the staging rows and THIS tile's set of source planes -- is not read by the waves still in the
    // previous tile's matrix loop (they read the displacement copies and the other set of source planes), so the digit extraction overlaps their tail; the first barrier
    // then says both "the staging rows are complete" and "everyone is done with the previous tile's copies".
    const int src_set = (band & 1) * SRC_SET;
    int btid = tid; // (re-defined per tile: the staging / copy addresses derived from it are loop invariant and would be hoisted out of the band loop -- into scratch, 480 B)
    SVT_HIP_OPAQUE_I32(btid);
    // ---- the digit planes of the tile from the loaded quads ----
#pragma unroll
    for (int k = 0; k < NIT; k++) {
        const int i = btid + 256 * k, r = i / SLOTS, s = i - r * SLOTS, tr = r - 3;
        const QuadWhere q = quad_where(4 * s - 4, -hw, tw + hw, tr >= -hw && tr < th + hw);
        uint32_t dh, dl;
        rawd[k].digits(q, avg, dh, dl);
        if (i < NR * SLOTS) {
            ((uint32_t*)stage)[i]                 = dh;
            ((uint32_t*)(stage + STAGE_PLANE))[i] = dl;
        }
    }
#pragma unroll
    for (int k = 0; k < SNIT; k++) { // source quad s of row r: chunk s / 16, quarter (s / 4) & 3, dword s & 3 of the cell
        const int i = btid + 256 * k, r = i / SSLOTS, s = i - r * SSLOTS;
        const QuadWhere q = quad_where(4 * s, 0, tw, r < th);
        uint32_t xh, xl;
        raws[k].digits(q, avg, xh, xl);
        const int cell = ((s >> 2) & 3) * (TR * 64) + r * 64 + (s >> 4) * 16 + (s & 3) * 4;
        *(uint32_t*)(planes + SRC_HI + src_set + cell) = xh;
        *(uint32_t*)(planes + SRC_LO + src_set + cell) = xl;
    }
    __syncthreads();
    CENSUS_PHASE(1)
    {   // ---- the displacement copies, one 16-byte cell (plane row r, chunk ch, quarter kg: columns c = 64 ch + 16 kg .. c + 15) per thread: copy dx of the cell = staging
        // bytes r * PP + 4 + c + dx .. + 15, i.e. six staging dwords funnel-shifted by dx; hi and lo digit share the address.  Item -> (ch, r & 1, kg, r >> 1): the eight
        // lanes a ds_write_b128 serves together then hit the eight different 16-byte slots of the 128-byte bank row.
        const int i = btid, ch = i & 3, r = 2 * (i >> 5) + ((i >> 2) & 1), kg = (i >> 3) & 3;
        if (i < NR * 16) {
            uint32_t wd[2][6];
#pragma unroll
            for (int d = 0; d < 2; d++) {
                const uint32_t* row = (const uint32_t*)(stage + d * STAGE_PLANE + r * PP) + 16 * ch + 4 * kg; // dword k of a staging row = tile columns 4 k - 4 .. 4 k - 1
                const i32x4     v = *(const i32x4*)row;
                wd[d][0] = (uint32_t)v[0]; wd[d][1] = (uint32_t)v[1]; wd[d][2] = (uint32_t)v[2]; wd[d][3] = (uint32_t)v[3];
                wd[d][4] = row[4]; wd[d][5] = row[5];
            }
            uint8_t* out = planes + kg * L::KP;
#pragma unroll
            for (int dx = -hw; dx <= hw; dx++) { // columns c + dx ..: dx < 0 starts inside dword 0 (= columns c - 4 ..), dx >= 0 inside dword 1
                const int Rr = L::A * r + L::B * (dx + hw);
                uint8_t*  cell = out + 64 * Rr + 16 * (ch ^ ((Rr >> 2) & 3));
#pragma unroll
                for (int d = 0; d < 2; d++) {
                    i32x4 v;
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        v[j] = (int)(dx < 0 ? __builtin_amdgcn_alignbyte(wd[d][j + 1], wd[d][j], (uint32_t)(4 + dx))
                                            : (dx == 0 ? wd[d][j + 1] : __builtin_amdgcn_alignbyte(wd[d][j + 2], wd[d][j + 1], (uint32_t)dx)));
                    *(i32x4*)(cell + d * LO) = v;
                }
            }
        }
    }
    __syncthreads();
    CENSUS_PHASE(2)
    if (band + 1 < nbands && r0 + TR < Hh) issue(r0 + TR, Hh - (r0 + TR) < TR ? Hh - (r0 + TR) : TR, btid); // the next tile's loads fly while the matrix pipe works on this one
    CENSUS_PHASE(3)
    {   // this wave's chunks of the tile: rows 2 wv, 2 wv + 1, nch chunks each, flattened; software pipelined: the operands of chunk it + 1
        // are fetched while the matrix pipe works through the 36 (10) MFMAs of chunk it
        const int nch = (tw + 63) >> 6;
        int nrow = th - wv * (TR / 4); // (tile rows 2 wv, 2 wv + 1 of th)
        nrow = nrow < 0 ? 0 : (nrow > TR / 4 ? TR / 4 : nrow);
        auto fetch = [&](const int ri, const int ch, i32x4 (&oH)[NG], i32x4 (&oL)[NG], auto full_tag) { // chunk ch of this wave's tile row ri (0 / 1)
            constexpr bool FULL = decltype(full_tag)::value; // every chunk of the tile is 64 pixels wide: no per-byte masks
            const int nvalid = tw - ch * 64; // pixels of this chunk inside the unit (>= 64: all)
#pragma unroll
            for (int g = 0; g < NG; g++) {
                const bool  last = g == NG - 1;
                const int   a = obase[g] + (last && (band & 1) ? src_flip : 0) + ri * (last ? last_step : 64 * L::A) + ((16 * ch) ^ (ri ? oswz[g][1] : oswz[g][0]));
                const i32x4 vh = *(const i32x4*)(planes + a), vl = *(const i32x4*)(planes + a + (last ? last_lo : LO));
                if (FULL) { // whole chunk: the LDS words ARE the operands
                    oH[g] = vh;
                    oL[g] = vl;
                    continue;
                }
#pragma unroll
                for (int d = 0; d < 4; d++) { // pixel 16 kg + 4 d + e of the chunk is outside the unit -> its byte must be zero in every operand
                    const int      first = 16 * (l >> 4) + 4 * d, left = nvalid - first;
                    const uint32_t m = left >= 4 ? ~0u : (left <= 0 ? 0u : ((1u << (8 * left)) - 1u));
                    oH[g][d] = (int)((uint32_t)vh[d] & m);
                    oL[g][d] = (int)((uint32_t)vl[d] & m);
                }
            }
        };
        auto multiply = [&](const i32x4 (&oH)[NG], const i32x4 (&oL)[NG]) {
            int ti = 0;
#pragma unroll
            for (int ga = 0; ga < NG; ga++)
#pragma unroll
                for (int gb = ga; gb < NG; gb++, ti++) {
                    accHH[ti] = __builtin_amdgcn_mfma_i32_16x16x64_i8(oH[ga], oH[gb], accHH[ti], 0, 0, 0);
                    accLL[ti] = __builtin_amdgcn_mfma_i32_16x16x64_i8(oL[ga], oL[gb], accLL[ti], 0, 0, 0);
                }
            ti = 0;
#pragma unroll
            for (int ga = 0; ga < NG; ga++)
#pragma unroll
                for (int gb = ga; gb < NG; gb++, ti++) accX[ti] = __builtin_amdgcn_mfma_i32_16x16x64_i8(oH[ga], oL[gb], accX[ti], 0, 0, 0);
            ti = 0;
#pragma unroll
            for (int ga = 0; ga < NG; ga++)
#pragma unroll
                for (int gb = ga; gb < NG; gb++, ti++)
                    if (gb != ga) accX[ti] = __builtin_amdgcn_mfma_i32_16x16x64_i8(oL[ga], oH[gb], accX[ti], 0, 0, 0);
        };
        i32x4 aH[NG], aL[NG], bH[NG], bL[NG];
        const int nit = nrow * nch;
        if ((tw & 63) == 0 && nit > 0) {
            // Straight-line two-chunk body (the (row, chunk) of the next fetch advances with scalar compares and stops at the last chunk instead of branching; an odd
            // tail multiplies zeros) so that the scheduler can place the next chunk's LDS reads in the shadow of the current chunk's MFMAs: a wave issues in order, and
            // 36 back-to-back MFMAs would otherwise keep the LDS pipe idle for their whole latencies.  An operand address is a select and a v_xad_u32.
            int  fr = 0, fc = 0; // the chunk the next fetch reads
            auto advance = [&]() {
                if (fc + 1 < nch) fc++;
                else if (fr + 1 < nrow) { fc = 0; fr++; }
            };
            fetch(0, 0, aH, aL, std::true_type{});
            advance();
            for (int it = 0; it < nit; it += 2) {
                fetch(fr, fc, bH, bL, std::true_type{});
                advance();
                if (it + 1 >= nit) {
#pragma unroll
                    for (int g = 0; g < NG; g++) { bH[g] = i32x4{0, 0, 0, 0}; bL[g] = i32x4{0, 0, 0, 0}; }
                }
                multiply(aH, aL);
                fetch(fr, fc, aH, aL, std::true_type{});
                advance();
                multiply(bH, bL);
#pragma unroll
                for (int k = 0; k < 2 * NMFMA; k += 4) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); // four MFMAs
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); // one LDS read (sixteen 16-byte reads per chunk pair against 72 MFMAs)
                    __builtin_amdgcn_sched_group_barrier(0x002, 1, 0); // one VALU (an address)
                }
            }
        } else {
            for (int ri = 0; ri < nrow; ri++)
                for (int ch = 0; ch < nch; ch++) {
                    fetch(ri, ch, aH, aL, std::false_type{});
                    multiply(aH, aL);
                }
        }
    }
    CENSUS_PHASE(4)
    }

    // ---- merge the four waves in LDS (int32 is still exact: 2 waves x 16384 pixels x 2 x 128 x 127 < 2^31 for the shared cross-term tiles of a copy), then one int64 atomic per entry ----
    __syncthreads();
    int* part = (int*)smem; // [NTILE][4 registers][64 lanes] (a wave's atomic touches 64 consecutive banks): HH tiles, LL tiles, X tiles
    // two copies: waves 0 and 1 store theirs, waves 2 and 3 add theirs on top (LDS atomics of two waves in a row instead of three), the read-out below adds the copies
    int* mine = part + (wv & 1) * (NTILE * 256);
    if (wv < 2) {
#pragma unroll
        for (int i = 0; i < NTRI; i++)
#pragma unroll
            for (int d = 0; d < 4; d++) {
                mine[(i * 4 + d) * 64 + l]                = accHH[i][d];
                mine[((NTRI + i) * 4 + d) * 64 + l]       = accLL[i][d];
                mine[((2 * NTRI + i) * 4 + d) * 64 + l]   = accX[i][d];
            }
    }
    __syncthreads();
    if (wv >= 2) {
#pragma unroll
        for (int i = 0; i < NTRI; i++)
#pragma unroll
            for (int d = 0; d < 4; d++) {
                atomicAdd(&mine[(i * 4 + d) * 64 + l], accHH[i][d]);
                atomicAdd(&mine[((NTRI + i) * 4 + d) * 64 + l], accLL[i][d]);
                atomicAdd(&mine[((2 * NTRI + i) * 4 + d) * 64 + l], accX[i][d]);
            }
    }
    __syncthreads();
    // C/D layout of the 16x16 MFMAs: lane = col + 16 * (row >> 2), register = row & 3
    constexpr int ncol = w2 + 1; // taps + source column
    for (int e = tid; e < ncol * ncol; e += 256) { // (a, b) over the square, the upper triangle kept: a division by a constant instead of a search for the row
        const int a = e / ncol, b = e - a * ncol;
        if (b < a || a == w2) continue; // (source, source) is not an output
        const int ga = a >> 4, ra = a & 15, gb = b >> 4, cb = b & 15, rb = b & 15, ca = a & 15;
        const int tri = ga * NG - ga * (ga - 1) / 2 + (gb - ga);
        const int eab = (ra & 3) * 64 + cb + 16 * (ra >> 2), eba = (rb & 3) * 64 + ca + 16 * (rb >> 2);
        constexpr int C2 = NTILE * 256; // the second copy
        const long long hh = (long long)part[tri * 256 + eab] + part[C2 + tri * 256 + eab], ll = (long long)part[(NTRI + tri) * 256 + eab] + part[C2 + (NTRI + tri) * 256 + eab];
        // the cross terms sum_p H_a L_b + L_a H_b: already summed in an off-diagonal tile; entries (a, b) and (b, a) of a diagonal one
        const long long x = (long long)part[(2 * NTRI + tri) * 256 + eab] + part[C2 + (2 * NTRI + tri) * 256 + eab] +
                            (ga == gb ? (long long)part[(2 * NTRI + tri) * 256 + eba] + part[C2 + (2 * NTRI + tri) * 256 + eba] : 0);
        const long long v = 16384 * hh + 128 * x + ll;
        unsigned long long* dst = (unsigned long long*)(b == w2 ? &M[a] : &H[a * w2 + b]);
        atomicAdd(dst, (unsigned long long)v);
    }
#ifdef SVT_HIP_STATS_CENSUS
    CENSUS_PHASE(5)
    if (tid == 0 && bgroup < 30) {
        long long* c = H + (10 + bgroup) * w2;
        c[0] = (long long)census_t0; c[1] = (long long)wall_clock64();
        c[2] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)); c[3] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11)); // HW_ID, XCC_ID
        for (int k = 0; k < 6; k++) c[4 + k] = census_ph[k];
    }
#endif
}

// upper triangle / M -> divide as the reference's `/=` (truncation toward zero, restoration_pick.c:733-742), mirror to the lower triangle
__global__ __launch_bounds__(256) void stats_finalize_kernel(const int win, const int bit_depth, long long* Mout, long long* Hout) {
    // `/=` by 1, 4 or 16 truncates toward zero: add (divisor - 1) to negative values, then shift (a 64-bit division costs a few hundred instructions per entry)
    const int       w2 = win * win, sh = bit_depth == 12 ? 4 : (bit_depth == 10 ? 2 : 0);
    const long long bias = (1ll << sh) - 1;
    auto            quot = [&](const long long v) { return (v + ((v >> 63) & bias)) >> sh; };
    long long* H = Hout + (size_t)blockIdx.x * 49 * 49;
    long long* M = Mout + (size_t)blockIdx.x * 49;
    for (int e = blockIdx.y * 256 + threadIdx.x; e < w2 * w2; e += 256 * gridDim.y) {
        const int k = e / w2, l2 = e - k * w2;
        if (k < l2) {
            const long long v = quot(H[e]);
            H[e]              = v;
            H[l2 * w2 + k]    = v;
        } else if (k == l2) {
            H[e] = quot(H[e]);
        }
    }
    if (blockIdx.y == 0)
        for (int k = threadIdx.x; k < w2; k += 256) M[k] = quot(M[k]);
}

} // namespace

extern "C" {

void svt_hip_lr_compute_stats_batch(const void* dgd, const void* src, const SvtHipRect* rects, uint32_t n, int max_rect_width, int max_rect_height, int dgd_stride,
                                    int src_stride, int wiener_win, int bit_depth, int64_t* M, int64_t* H, void* stream) {
    svt_hip_lr_compute_stats_batch_samples(dgd, src, rects, n, max_rect_width, max_rect_height, dgd_stride, src_stride, wiener_win, bit_depth, bit_depth > 8 ? 2 : 1, M, H, stream);
}
void svt_hip_lr_compute_stats_batch_samples(const void* dgd, const void* src, const SvtHipRect* rects, uint32_t n, int max_rect_width, int max_rect_height, int dgd_stride,
                                            int src_stride, int wiener_win, int bit_depth, int sample_bytes, int64_t* M, int64_t* H, void* stream) {
    svthip::ensure_device();
    if (n == 0) return;
    hipStream_t st = (hipStream_t)stream;
    const int   is16 = sample_bytes == 2, w2 = wiener_win * wiener_win;
    const int rows_per_band = (max_rect_height + SUM_BANDS - 1) / SUM_BANDS > 0 ? (max_rect_height + SUM_BANDS - 1) / SUM_BANDS : 1;
    hipLaunchKernelGGL(stats_sum_kernel, dim3(SUM_BANDS, n), dim3(256), 0, st, dgd, rects, dgd_stride, is16, w2, rows_per_band, (long long*)M, (long long*)H);
    SVT_LAUNCH_CHECK();
    // Rows per workgroup: 8 nbands.  More bands per workgroup = fewer merges (120 LDS atomics per lane + 1 275 int64 atomics per workgroup), fewer = shorter workgroups.
    // The launch should be many times the 512 resident workgroups of the chip: with one wave of 510 64-row workgroups (a 4K plane) the dispatcher left a dozen CUs
    // with three and some with one, and the kernel took two workgroup lifetimes (gpurun_out/r06_call9/census.txt).
    static const int env_bands = [] { const char* e = getenv("SVT_HIP_STATS_BANDS"); const int v = e ? atoi(e) : 0; return v >= 1 && v <= BANDS ? v : 0; }(); // (A/B measurement)
    int nbands = env_bands ? env_bands : BANDS;
    if (!env_bands) {
        const long long wg8 = (long long)((max_rect_width + TC - 1) / TC) * ((max_rect_height + TR * BANDS - 1) / (TR * BANDS)) * n;
        nbands = wg8 >= 4096 ? 8 : (wg8 >= 256 ? 4 : 2); // (a 4K plane, 120 units: 85 us at 4, 93 at 2, 97 at 8: gpurun_out/r06_call12)
    }
    const dim3 grid((max_rect_width + TC - 1) / TC, n, (max_rect_height + TR * nbands - 1) / (TR * nbands));
    if (grid.x && grid.z) {
        auto go = [&](auto win_tag, auto is16_tag) {
            constexpr int  WIN = decltype(win_tag)::value;
            constexpr bool IS16 = decltype(is16_tag)::value;
            constexpr int  NG = (WIN * WIN + 1 + 15) / 16, merge = 2 * 3 * (NG * (NG + 1) / 2) * 256 * 4; // the final merge (two copies of the tiles) reuses the tile's LDS
            const size_t   shmem = (size_t)(merge > lds_bytes<WIN>() ? merge : lds_bytes<WIN>()) + 64;
            hipLaunchKernelGGL((stats_mfma_kernel<WIN, IS16>), grid, dim3(256), shmem, st, dgd, src, rects, dgd_stride, src_stride, (long long*)M, (long long*)H, nbands);
        };
        auto by_depth = [&](auto win_tag) { is16 ? go(win_tag, std::true_type{}) : go(win_tag, std::false_type{}); };
        if (wiener_win == 7) by_depth(std::integral_constant<int, 7>{});
        else if (wiener_win == 5) by_depth(std::integral_constant<int, 5>{});
        else by_depth(std::integral_constant<int, 3>{}); // WIENER_WIN_3TAP (restoration_pick.c:1289)
        SVT_LAUNCH_CHECK();
    }
#ifndef SVT_HIP_STATS_CENSUS
    hipLaunchKernelGGL(stats_finalize_kernel, dim3(n, 4), dim3(256), 0, st, wiener_win, bit_depth, (long long*)M, (long long*)H);
#endif
    SVT_LAUNCH_CHECK();
}

} // extern "C"

SVT_HIP_DEFINE_WARM(lr_stats) // (svt_hip_warmup loads this translation unit's code object at encoder initialisation: svt_hip_common.h)
