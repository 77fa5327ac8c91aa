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
constexpr int COPY_PLANE = NR * TC;   // one digit of one displacement copy: pitch TC, no halo columns (the displacement is in the copy)
constexpr int SRC_PLANE = TR * TC;    // src digit plane, pitch TC, no halo
template <int WIN> constexpr int lds_bytes() { return 2 * STAGE_PLANE + 2 * WIN * COPY_PLANE + 2 * SRC_PLANE; }

__device__ __forceinline__ long long wave_sum_ll(long long v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
__device__ __forceinline__ int rd(const void* p, const int is16, const size_t off) { return is16 ? ((const uint16_t*)p)[off] : ((const uint8_t*)p)[off]; }

// find_average / find_average_highbd (restoration_pick.h): floor(sum / count) of the degraded unit.  One workgroup per (unit, 16-row band) adds its
// partial sum to a lower-triangle slot of H (free until the finalize kernel mirrors the upper triangle); the consumers divide.
__global__ __launch_bounds__(256) void stats_sum_kernel(const void* dgd, const SvtHipRect* rects, const int dgd_stride, const int is16, const int w2,
                                                        long long* H) {
    __shared__ long long wsum[4];
    const int        tid = threadIdx.x;
    const SvtHipRect R   = rects[blockIdx.y];
    const int        W = R.h_end - R.h_start, Hh = R.v_end - R.v_start, r0 = blockIdx.x * 16;
    if (r0 >= Hh || W <= 0) return;
    const int rows = Hh - r0 < 16 ? Hh - r0 : 16;
    long long s = 0;
    for (int r = tid >> 6; r < rows; r += 4) { // a wave per row: coalesced
        const size_t base = (size_t)((long long)(R.v_start + r0 + r) * dgd_stride + R.h_start);
        for (int x = tid & 63; x < W; x += 64) s += rd(dgd, is16, base + x);
    }
    s = wave_sum_ll(s);
    if ((tid & 63) == 0) wsum[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) atomicAdd((unsigned long long*)&H[(size_t)blockIdx.y * 49 * 49 + w2], (unsigned long long)(wsum[0] + wsum[1] + wsum[2] + wsum[3]));
}

__device__ __forceinline__ uint32_t pack_digits_hi(const int v0, const int v1, const int v2, const int v3) {
    return (uint32_t)((v0 >> 7) & 255) | ((uint32_t)((v1 >> 7) & 255) << 8) | ((uint32_t)((v2 >> 7) & 255) << 16) | ((uint32_t)((v3 >> 7) & 255) << 24);
}
__device__ __forceinline__ uint32_t pack_digits_lo(const int v0, const int v1, const int v2, const int v3) {
    return (uint32_t)(v0 & 127) | ((uint32_t)(v1 & 127) << 8) | ((uint32_t)(v2 & 127) << 16) | ((uint32_t)(v3 & 127) << 24);
}


// WIN 7: 49 taps + source = 50 columns in NG = 4 groups of 16; WIN 5: 26 columns, NG = 2; WIN 3: 10 columns, NG = 1.  With the window a template parameter every
// group but the last is known to hold taps only (16 (NG - 1) <= WIN^2), so only the last group keeps per-lane plane / pitch / mask registers.
template <int WIN>
__global__ __launch_bounds__(256, 2) void stats_mfma_kernel(const void* dgd, const void* src, const SvtHipRect* rects, const int dgd_stride, const int src_stride,
                                                         const int is16, long long* Mout, long long* Hout) {
    constexpr int win = WIN, NG = (WIN * WIN + 1 + 15) / 16;
    static_assert(16 * (NG - 1) <= WIN * WIN, "only the last group may hold the source column or padding");
    // accumulator tiles: HH, LL and X over the upper triangle of group pairs, X(ga, gb) = H_ga L_gb^T (+ L_ga H_gb^T off the diagonal: both cross terms of an
    // entry carry the same weight, so they share an accumulator -- 30 tiles instead of 36 for WIN 7, same 36 MFMAs per chunk)
    constexpr int NTRI = NG * (NG + 1) / 2, NTILE = 3 * NTRI, NMFMA = 2 * NTRI + NG * NG;
    HIP_DYNAMIC_SHARED(uint32_t, smem)
    // LDS: [staging hi][staging lo] (unshifted tile rows with halo columns) | [hi copies: WIN x NR x TC][lo copies] | [src hi][src lo]; later reused as int32 [NTILE][256]
    uint8_t* stage  = (uint8_t*)smem;
    uint8_t* planes = stage + 2 * STAGE_PLANE;
    constexpr int LO = WIN * COPY_PLANE, SRC_HI = 2 * WIN * COPY_PLANE, SRC_LO = SRC_HI + SRC_PLANE;
    const int        tid = threadIdx.x, l = tid & 63, wv = tid >> 6;
    const int        unit = blockIdx.z;
    const SvtHipRect R = rects[unit];
    constexpr int w2 = win * win, hw = win >> 1;
    const int W = R.h_end - R.h_start, Hh = R.v_end - R.v_start;
    const int c0 = blockIdx.x * TC, rb0 = blockIdx.y * TR * BANDS; // origin of the workgroup's 64-row band group inside the unit
    if (c0 >= W || rb0 >= Hh) return;
    const int tw = W - c0 < TC ? W - c0 : TC;
    long long* H = Hout + (size_t)unit * 49 * 49;
    long long* M = Mout + (size_t)unit * 49;
    const int  avg = (int)((unsigned long long)H[w2] / (unsigned long long)((long long)W * Hh)); // H[w2] = sum of the degraded unit (stats_sum_kernel)

    // ---- per-lane operand addressing: group g -> tap t = 16 g + (l & 15); kg = l >> 4 selects pixels 16 kg .. 16 kg + 15 of a chunk ----
    // every operand is sixteen bytes at a 16-byte-aligned address: displacement copy dx, plane row (row + 3 + dy), column 64 ch + 16 kg -- all pitches are TC
    int      opoff[NG], last_hi = 0, last_lo = LO;
    uint32_t last_mask = ~0u;
#pragma unroll
    for (int g = 0; g < NG; g++) {
        const int t = 16 * g + (l & 15), kg = l >> 4;
        if (g < NG - 1 || t < w2) { // tap index = (dx + hw) * win + (dy + hw)   (restoration_pick.c:673-679: k over columns, l over rows)
            const int dx = t / win - hw, dy = t % win - hw;
            opoff[g] = ((dx + hw) * NR + 3 + dy) * TC + 16 * kg;
        } else { // the source column (t == w2) or padding (contributes zeros)
            opoff[g] = 16 * kg; last_hi = SRC_HI; last_lo = SRC_LO; last_mask = t == w2 ? ~0u : 0u;
        }
    }
    i32x4 accHH[NTRI], accLL[NTRI], accX[NTRI];
#pragma unroll
    for (int i = 0; i < NTRI; i++) { accHH[i] = i32x4{0, 0, 0, 0}; accLL[i] = i32x4{0, 0, 0, 0}; accX[i] = i32x4{0, 0, 0, 0}; }

    for (int band = 0; band < BANDS; band++) {
    const int r0 = rb0 + band * TR;
    if (r0 >= Hh) break;
    const int th = Hh - r0 < TR ? Hh - r0 : TR;
    if (band) __syncthreads(); // everyone is done reading the previous tile
    // ---- stage the digit planes: four samples per step, loads issued before the stores ----
    {
        constexpr int SLOTS = PP / 4, NIT = (NR * SLOTS + 255) / 256, HALF = (NIT + 1) / 2;
#pragma unroll
        for (int k0 = 0; k0 < NIT; k0 += HALF) { // two batches: the accumulators leave no room for all NIT quads at once
            int v[HALF][4];
#pragma unroll
            for (int k = 0; k < HALF; k++) {
                const int i = tid + 256 * (k0 + k), r = i / SLOTS, s = i - r * SLOTS; // plane row r <-> tile row r - 3, slot s <-> tile columns 4 s - 4 ..
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int tr = r - 3, tc = 4 * s - 4 + e;
                    const bool ok = k0 + k < NIT && i < NR * SLOTS && tr >= -hw && tr < th + hw && tc >= -hw && tc < tw + hw;
                    v[k][e] = ok ? rd(dgd, is16, (size_t)((long long)(R.v_start + r0 + tr) * dgd_stride + (R.h_start + c0 + tc))) - avg : 0;
                }
            }
#pragma unroll
            for (int k = 0; k < HALF; k++) {
                const int i = tid + 256 * (k0 + k);
                if (k0 + k < NIT && i < NR * SLOTS) {
                    ((uint32_t*)stage)[i]                 = pack_digits_hi(v[k][0], v[k][1], v[k][2], v[k][3]);
                    ((uint32_t*)(stage + STAGE_PLANE))[i] = pack_digits_lo(v[k][0], v[k][1], v[k][2], v[k][3]);
                }
            }
        }
        constexpr int SSLOTS = TC / 4, SNIT = (TR * SSLOTS + 255) / 256;
        int x[SNIT][4];
#pragma unroll
        for (int k = 0; k < SNIT; k++) {
            const int i = tid + 256 * k, r = i / SSLOTS, s = i - r * SSLOTS;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int tc = 4 * s + e;
                x[k][e] = (r < th && tc < tw) ? rd(src, is16, (size_t)((long long)(R.v_start + r0 + r) * src_stride + (R.h_start + c0 + tc))) - avg : 0;
            }
        }
#pragma unroll
        for (int k = 0; k < SNIT; k++) {
            const int i = tid + 256 * k;
            ((uint32_t*)(planes + SRC_HI))[i] = pack_digits_hi(x[k][0], x[k][1], x[k][2], x[k][3]);
            ((uint32_t*)(planes + SRC_LO))[i] = pack_digits_lo(x[k][0], x[k][1], x[k][2], x[k][3]);
        }
    }
    __syncthreads();
    {   // ---- the displacement copies: copy dx, row r, columns 4 q .. 4 q + 3 = staging bytes (r * PP + 4 + 4 q + dx ..): two adjacent staging dwords, one funnel shift ----
        constexpr int Q = TC / 4;
        for (int i = tid; i < NR * Q; i += 256) {
            const int r = i / Q, q = i - r * Q;
#pragma unroll
            for (int d = 0; d < 2; d++) {
                const uint32_t* row = (const uint32_t*)(stage + d * STAGE_PLANE + r * PP); // dword k of the row = tile columns 4 k - 4 .. 4 k - 1
                const uint32_t  w0 = row[q], w1 = row[q + 1], w2_ = row[q + 2];
                uint32_t*       out = (uint32_t*)(planes + d * LO + r * TC) + q;
#pragma unroll
                for (int dx = -hw; dx <= hw; dx++) { // columns 4 q + dx ..: dx < 0 starts in dword q (= columns 4 q - 4 ..), dx >= 0 in dword q + 1
                    const uint32_t v = dx < 0 ? __builtin_amdgcn_alignbyte(w1, w0, (uint32_t)(4 + dx)) : (dx == 0 ? w1 : __builtin_amdgcn_alignbyte(w2_, w1, (uint32_t)dx));
                    out[(dx + hw) * (COPY_PLANE / 4)] = v;
                }
            }
        }
    }
    __syncthreads();
    {   // this wave's chunks of the tile: rows 4 wv .. 4 wv + 3, nch chunks each, flattened; software pipelined: the operands of chunk it + 1
        // are fetched (LDS + funnel shifts) while the matrix pipe works through the 36 (10) MFMAs of chunk it
        const int nch = (tw + 63) >> 6;
        int nrow = th - wv * (TR / 4);
        nrow = nrow < 0 ? 0 : (nrow > TR / 4 ? TR / 4 : nrow);
        const int nit = nrow * nch;
        auto fetch = [&](const int it, i32x4 (&oH)[NG], i32x4 (&oL)[NG], auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value; // every chunk of the tile is 64 pixels wide: no per-byte masks
            const int row = wv * (TR / 4) + it / nch, ch = it % nch;
            const int nvalid = tw - ch * 64; // pixels of this chunk inside the unit (>= 64: all)
#pragma unroll
            for (int g = 0; g < NG; g++) {
                const bool  last = g == NG - 1;
                const int   a = row * TC + 64 * ch + opoff[g]; // a multiple of 16
                const i32x4 vh = *(const i32x4*)(planes + (last ? last_hi : 0) + a), vl = *(const i32x4*)(planes + (last ? last_lo : LO) + a);
                if (FULL && !last) { // whole chunk, taps only: the LDS words ARE the operands
                    oH[g] = vh;
                    oL[g] = vl;
                    continue;
                }
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    uint32_t m = last ? last_mask : ~0u; // only the last group holds the source column and padding
                    if (!FULL && nvalid < 64) { // pixel 16 kg + 4 d + e of the chunk is outside the unit -> its byte must be zero in every operand
                        const int first = 16 * (l >> 4) + 4 * d, left = nvalid - first;
                        m &= left >= 4 ? ~0u : (left <= 0 ? 0u : ((1u << (8 * left)) - 1u));
                    }
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
        if ((tw & 63) == 0 && nit > 0) {
            // Straight-line two-chunk body (fetches clamp to the last chunk instead of branching; an odd tail multiplies zeros) so that the
            // scheduler can place the next chunk's LDS reads and funnel shifts in the shadow of the current chunk's MFMAs: a wave issues in
            // order, and 36 back-to-back MFMAs would otherwise keep the VALU idle for their whole 16-cycle latencies.
            fetch(0, aH, aL, std::true_type{});
            for (int it = 0; it < nit; it += 2) {
                const int i1 = it + 1 < nit ? it + 1 : nit - 1, i2 = it + 2 < nit ? it + 2 : nit - 1;
                fetch(i1, bH, bL, std::true_type{});
                if (it + 1 >= nit) {
#pragma unroll
                    for (int g = 0; g < NG; g++) { bH[g] = i32x4{0, 0, 0, 0}; bL[g] = i32x4{0, 0, 0, 0}; }
                }
                multiply(aH, aL);
                fetch(i2, aH, aL, std::true_type{});
                multiply(bH, bL);
#pragma unroll
                for (int k = 0; k < 2 * NMFMA; k++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); // four MFMAs
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); // one LDS read (eight 16-byte reads per chunk against 36 MFMAs)
                    __builtin_amdgcn_sched_group_barrier(0x002, 1, 0); // one VALU (addresses, the last group's mask)
                }
            }
        } else {
            for (int it = 0; it < nit; it++) {
                fetch(it, aH, aL, std::false_type{});
                multiply(aH, aL);
            }
        }
    }
    }

    // ---- merge the four waves in LDS (int32 is still exact: 4 waves x 16384 pixels x 2 x 128 x 127 < 2^31 for the shared cross-term tiles), then one int64 atomic per entry ----
    __syncthreads();
    int* part = (int*)smem; // [NTILE][64 lanes][4]: HH tiles, LL tiles, X tiles
    for (int i = tid; i < NTILE * 256; i += 256) part[i] = 0;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NTRI; i++)
#pragma unroll
        for (int d = 0; d < 4; d++) {
            atomicAdd(&part[(i * 64 + l) * 4 + d], accHH[i][d]);
            atomicAdd(&part[((NTRI + i) * 64 + l) * 4 + d], accLL[i][d]);
            atomicAdd(&part[((2 * NTRI + i) * 64 + l) * 4 + d], accX[i][d]);
        }
    __syncthreads();
    // C/D layout of the 16x16 MFMAs: lane = col + 16 * (row >> 2), register = row & 3
    constexpr int ncol = w2 + 1; // taps + source column
    for (int e = tid; e < ncol * (ncol + 1) / 2; e += 256) {
        int a = 0, rem = e;
        while (rem >= ncol - a) { rem -= ncol - a; a++; }
        const int b = a + rem; // a <= b
        if (a == w2) continue;  // (source, source) is not an output
        const int ga = a >> 4, ra = a & 15, gb = b >> 4, cb = b & 15, rb = b & 15, ca = a & 15;
        const int tri = ga * NG - ga * (ga - 1) / 2 + (gb - ga);
        const int eab = (cb + 16 * (ra >> 2)) * 4 + (ra & 3), eba = (ca + 16 * (rb >> 2)) * 4 + (rb & 3);
        const long long hh = part[tri * 256 + eab], ll = part[(NTRI + tri) * 256 + eab];
        // the cross terms sum_p H_a L_b + L_a H_b: already summed in an off-diagonal tile; entries (a, b) and (b, a) of a diagonal one
        const long long x = (long long)part[(2 * NTRI + tri) * 256 + eab] + (ga == gb ? (long long)part[(2 * NTRI + tri) * 256 + eba] : 0);
        const long long v = 16384 * hh + 128 * x + ll;
        unsigned long long* dst = (unsigned long long*)(b == w2 ? &M[a] : &H[a * w2 + b]);
        atomicAdd(dst, (unsigned long long)v);
    }
}

// upper triangle / M -> divide as the reference's `/=` (truncation toward zero, restoration_pick.c:733-742), mirror to the lower triangle
__global__ __launch_bounds__(256) void stats_finalize_kernel(const int win, const int bit_depth, long long* Mout, long long* Hout) {
    const int w2 = win * win, div = bit_depth == 12 ? 16 : (bit_depth == 10 ? 4 : 1);
    long long* H = Hout + (size_t)blockIdx.x * 49 * 49;
    long long* M = Mout + (size_t)blockIdx.x * 49;
    for (int e = threadIdx.x; e < w2 * w2; e += 256) {
        const int k = e / w2, l2 = e - k * w2;
        if (k < l2) {
            const long long v = H[e] / div;
            H[e]              = v;
            H[l2 * w2 + k]    = v;
        } else if (k == l2) {
            H[e] = H[e] / div;
        }
    }
    for (int k = threadIdx.x; k < w2; k += 256) M[k] = M[k] / div;
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
    HIP_CHECK(hipMemsetAsync(M, 0, (size_t)n * 49 * 8, st));
    HIP_CHECK(hipMemsetAsync(H, 0, (size_t)n * 49 * 49 * 8, st));
    hipLaunchKernelGGL(stats_sum_kernel, dim3((max_rect_height + 15) / 16, n), dim3(256), 0, st, dgd, rects, dgd_stride, is16, w2, (long long*)H);
    SVT_LAUNCH_CHECK();
    const dim3 grid((max_rect_width + TC - 1) / TC, (max_rect_height + TR * BANDS - 1) / (TR * BANDS), n);
    if (grid.x && grid.y) {
        if (wiener_win == 7) {
            const size_t shmem = (size_t)(30 * 256 * 4 > lds_bytes<7>() ? 30 * 256 * 4 : lds_bytes<7>()) + 64;
            hipLaunchKernelGGL(stats_mfma_kernel<7>, grid, dim3(256), shmem, st, dgd, src, rects, dgd_stride, src_stride, is16, (long long*)M, (long long*)H);
        } else if (wiener_win == 5) {
            const size_t shmem = (size_t)(9 * 256 * 4 > lds_bytes<5>() ? 9 * 256 * 4 : lds_bytes<5>()) + 64;
            hipLaunchKernelGGL(stats_mfma_kernel<5>, grid, dim3(256), shmem, st, dgd, src, rects, dgd_stride, src_stride, is16, (long long*)M, (long long*)H);
        } else { // WIENER_WIN_3TAP (restoration_pick.c:1289)
            const size_t shmem = (size_t)(3 * 256 * 4 > lds_bytes<3>() ? 3 * 256 * 4 : lds_bytes<3>()) + 64;
            hipLaunchKernelGGL(stats_mfma_kernel<3>, grid, dim3(256), shmem, st, dgd, src, rects, dgd_stride, src_stride, is16, (long long*)M, (long long*)H);
        }
        SVT_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(stats_finalize_kernel, dim3(n), dim3(256), 0, st, wiener_win, bit_depth, (long long*)M, (long long*)H);
    SVT_LAUNCH_CHECK();
}

} // extern "C"

SVT_HIP_DEFINE_WARM(lr_stats) // (svt_hip_warmup loads this translation unit's code object at encoder initialisation: svt_hip_common.h)
