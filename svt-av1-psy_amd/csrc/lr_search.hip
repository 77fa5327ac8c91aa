// lr_search.hip -- the per-unit half of the loop-restoration SEARCH of one plane as a resident device stage (the caller of a21-a24).
//
// Reference: restoration_seg_search (Source/Lib/Codec/restoration_pick.c:1448-1527) runs, for every restoration unit,
//   search_norestore_seg (:1409)  SSE of the unrestored unit;
//   search_wiener_seg (:1281)     svt_av1_compute_stats -> wiener_decompose_sep_sym (:894, integer alternating least squares over linsolve_wiener :754)
//                                 -> finalize_sym_filter (:962) -> compute_score (:925) -> finer_tile_search_wiener_seg (:1027: coordinate descent on the
//                                 taps, every trial = restore the unit + SSE against the source, try_restoration_unit_seg :129);
//   search_sgrproj_seg (:1205)    search_selfguided_restoration (:542): per parameter set the two self-guided filters, svt_get_proj_subspace, encode_xq,
//                                 finer_search_pixel_proj_error (:320); then the SSE of the unit restored with the winner.
// The picture-level decisions (search_*_finish: rate against the previous unit's coefficients, serial) stay with the encoder.
// scs->use_boundaries_in_rest_search is 0 (enc_handle.c:4129): trials filter the plain edge-extended plane, no stripe-boundary substitution.
//
// Mapping.  Units are independent; within a unit both searches are greedy sequences of "evaluate a candidate over the whole unit, keep it if not worse".
//   * Wiener: statistics on the matrix cores (lr_stats.hip); the solve is one wave per unit (the 49 x 49 accumulations spread over the lanes, the 3 x 3
//     integer elimination on lane 0, all int64 with the reference's truncating divisions); the refinement runs all units in LOCK STEP: a step kernel (one
//     thread per unit = the reference's nested loops as a resumable state machine) proposes each unit's next candidate, a trial kernel (one workgroup per
//     64 x 64 tile of every active unit: the Wiener passes of lr_core.h on a staged tile, squared error against the source, one 64-bit atomic per workgroup)
//     evaluates them; finished units drop out.  The host reads the number of active units back every few steps.
//   * self-guided: one launch filters the plane with every parameter set of the range (flt0 / flt1 kept as int32 planes in the workspace), one workgroup
//     per (unit, parameter set) does projection + refinement (each evaluation = a pass over the unit's flt0 / flt1 / dgd / src), a per-unit thread
//     picks the first-best set, the trial kernel restores with it.
#include <type_traits>

#include "lr_core.h"

extern "C" void svt_hip_lr_compute_stats_batch(const void* dgd, const void* src, const SvtHipRect* rects, uint32_t n, int max_rect_width, int max_rect_height,
                                               int dgd_stride, int src_stride, int wiener_win, int bit_depth, int64_t* M, int64_t* H, void* stream);

namespace {

constexpr long long kScale = 1ll << 16; // WIENER_TAP_SCALE_FACTOR
constexpr int       kStep  = 128;       // WIENER_FILT_STEP
__device__ constexpr int kTapMin[3] = {-5, -23, -17}, kTapMax[3] = {10, 8, 46}; // WIENER_FILT_TAPn_MINV / MAXV (restoration.h:143-149)
__device__ constexpr int kInitFilt[7] = {3, -7, 15, 106, 15, -7, 3};            // WIENER_FILT_TAPn_MIDV

struct WnState { // one unit's position in finer_tile_search_wiener_seg
    int16_t   v[8], h[8];
    long long err;
    int32_t   s, dir, p, sign, skip;
    int32_t   active;  // 1: a trial with the taps above is wanted / in flight
    int32_t   started; // the initial evaluation has been issued
    int32_t   trials;
};
struct SgResult { long long err; int32_t xqd[2]; };

struct Ws { // workspace carving (device pointers)
    SvtHipRect*         rects;
    unsigned long long* acc;     // [n] trial accumulators (RESTORE_NONE, Wiener)
    unsigned long long* acc2;    // [n] the self-guided branch's own (it runs beside the Wiener refinement on a second stream)
    WnState*            wn;      // [n]
    long long*          M;       // [n][49]
    long long*          H;       // [n][49 * 49]
    SgResult*           sg;      // [n][slots]
    long long*          sgsum;   // [n][slots][5] the projection's normal equations (lr_sgr_flt_kernel -> lr_sgr_proj_kernel)
    int32_t*            counter; // [1] units still refining
    int32_t*            flt;     // [slots][2][height * width]
};
__host__ __device__ inline int n_units_1d(const int size, const int us) { const int v = (size + (us >> 1)) / us; return v > 0 ? v : 1; }

// unit grid of svt_aom_foreach_rest_unit_in_frame (restoration.c:1240-1330)
// (also clears what the searches accumulate into -- the refinement's counters, the self-guided branch's accumulators and normal-equation sums: three fill launches less between
//  the caller and the fork, each a host call on the stage's critical path)
__global__ void lr_rects_kernel(const SvtHipLrSearchParams P, SvtHipRect* rects, unsigned long long* acc, WnState* wn, SvtHipLrSearchUnit* out, const int n, int32_t* counter,
                                unsigned long long* acc2, long long* sgsum, const int sgsum_n) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u < 32) counter[u] = 0;
    if (sgsum)
        for (int i = u; i < sgsum_n; i += (int)(gridDim.x * blockDim.x)) sgsum[i] = 0;
    if (u >= n) return;
    if (acc2) acc2[u] = 0;
    const int us = (int)P.unit_size, w = (int)P.width, h = (int)P.height, off = 8 >> P.ss_y;
    const int nvu = n_units_1d(h, us), nhu = n_units_1d(w, us), ur = u / nhu, uc = u % nhu;
    SvtHipRect r;
    r.h_start = uc * us; r.h_end = uc == nhu - 1 ? w : (uc + 1) * us;
    r.v_start = ur == 0 ? 0 : ur * us - off; r.v_end = ur == nvu - 1 ? h : (ur + 1) * us - off;
    rects[u] = r;
    acc[u]   = 0;
    WnState z = {};
    wn[u]     = z;
    SvtHipLrSearchUnit o = {};
    o.sse[1] = o.sse[2] = INT64_MAX; // overwritten only by a search that ran to its end: a disabled tool must never read as a perfect restoration
    out[u]               = o;
}

// ---- trial kernel: restore one 64 x 64 tile of a unit with the unit's candidate and add its squared error against the source -------------------------
// KIND 0: unrestored; 1: Wiener with wn[u] (active units only); 2: self-guided with out[u].ep / xqd
constexpr size_t LRS_A_BYTES = (size_t)66 * 66 * 2 + 8;
constexpr size_t LRS_SMEM    = (size_t)TH * TW * 2 + LRS_A_BYTES + (size_t)66 * 66 * 4 + 512;
constexpr size_t LRS_WN_MID_BYTES = (size_t)36 * 64 * 4;                                   // the Wiener passes' row-pair plane (35 row pairs of 64 dwords)
constexpr size_t LRS_WN_SMEM = (size_t)TH * TW * 2 + LRS_WN_MID_BYTES + (size_t)64 * 64 * 2; // KIND 1: tile + row-pair plane + source tile = 27.5 KB, five workgroups per CU
constexpr unsigned long long WN_TICKET = 1ull << 48;
__device__ void lr_wiener_step(const SvtHipLrSearchParams& P, WnState* wn, unsigned long long* acc, SvtHipLrSearchUnit* out, int32_t* counter, const int u, const long long err2);
template <int KIND>
__global__ __launch_bounds__(256) void lr_trial_kernel(const SvtHipLrSearchParams P, const SvtHipRect* __restrict__ rects, WnState* wn,
                                                       const SvtHipLrSearchUnit* __restrict__ units, unsigned long long* __restrict__ acc,
                                                       SvtHipLrSearchUnit* out_units, int32_t* counter) {
    HIP_DYNAMIC_SHARED(uint16_t, smem)
    __shared__ unsigned long long part[4];
    const int        u = blockIdx.z, tid = threadIdx.x;
    const SvtHipRect r = rects[u];
    if (KIND == 1 && !wn[u].active) return;
    TileSrc s;
    s.data = P.dgd; s.above = s.below = nullptr; s.stride = (int)P.dgd_stride; s.bstride = 0; s.w = 0; s.h = 0; s.highbd = P.highbd;
    s.stripe_idx = 0; s.stripe_top = 0; s.stripe_bot = 0;
    s.x0 = r.h_start + (int)blockIdx.x * 64; s.y0 = r.v_start + (int)blockIdx.y * 64;
    if (s.x0 >= r.h_end || s.y0 >= r.v_end) return;
    s.uw = r.h_end - s.x0 < 64 ? r.h_end - s.x0 : 64; s.uh = r.v_end - s.y0 < 64 ? r.v_end - s.y0 : 64;
    uint16_t* tile = smem;
    uint16_t* mid  = tile + TH * TW;
    uint16_t* A16  = mid;
    int32_t*  B32  = (int32_t*)((uint8_t*)mid + LRS_A_BYTES);
    uint16_t* xlut = (uint16_t*)(B32 + 66 * 66);
    const int highbd = P.highbd, bd = P.bit_depth, x0 = s.x0, y0 = s.y0;
    const void* src = P.src;
    const size_t sstride = P.src_stride;
    unsigned long long sse = 0;
    auto score = [&](int rr, int c, int v0, int v1, bool has1) {
        const size_t o = (size_t)(y0 + rr) * sstride + x0 + c;
        const int d0 = v0 - rd_px(src, highbd, o);
        sse += (unsigned long long)(uint32_t)(d0 * d0);
        if (has1) { const int d1 = v1 - rd_px(src, highbd, o + 1); sse += (unsigned long long)(uint32_t)(d1 * d1); }
    };
    // KIND 1 (a refinement round: a latency chain of tens of these launches): the tile's source samples are fetched with the degraded ones -- two 16-byte loads per
    // thread instead of sixteen 2-byte loads inside the vertical pass -- and parked in LDS while the horizontal pass runs
    uint16_t*    srct = mid + LRS_WN_MID_BYTES / 2; // [64][64]
    svt_u32x4_a2 sq[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    if (KIND == 1) {
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int j = tid + 256 * k, rr = j >> 3, cx = (j & 7) * 8;
            if (rr < s.uh && cx < s.uw) {
                const size_t o = (size_t)(y0 + rr) * sstride + x0 + cx;
                if (cx + 8 <= s.uw) {
                    if (highbd) sq[k] = svt_hip_global_load_x4((const uint16_t*)src + o);
                    else { const svt_u32x2_a1 b = svt_hip_global_load_x2((const uint8_t*)src + o); // eight bytes -> eight 16-bit samples
                           sq[k][0] = (b[0] & 0xffu) | ((b[0] & 0xff00u) << 8); sq[k][1] = ((b[0] >> 16) & 0xffu) | ((b[0] >> 8) & 0xff0000u);
                           sq[k][2] = (b[1] & 0xffu) | ((b[1] & 0xff00u) << 8); sq[k][3] = ((b[1] >> 16) & 0xffu) | ((b[1] >> 8) & 0xff0000u); }
                } else { // the ragged right edge of a unit: sample by sample
                    for (int e = 0; e < 8 && cx + e < s.uw; e++) sq[k][e >> 1] |= (uint32_t)rd_px(src, highbd, o + e) << (16 * (e & 1));
                }
            }
        }
    }
    stage_tile<(TH + 15) / 16>(tile, s, tid);
    __syncthreads();
    if (KIND == 1) {
        WienerTaps t;
#pragma unroll
        for (int k = 0; k < 8; k++) { t.fx[k] = wn[u].h[k]; t.fy[k] = wn[u].v[k]; }
        wiener_tile(tile, mid, t, s.uw, s.uh, bd, tid,
                    [&](int rr, int c, int v0, int v1, bool has1) {
                        const uint32_t sp = *(const uint32_t*)(srct + rr * 64 + c);
                        const int      d0 = v0 - (int)(sp & 0xffffu);
                        sse += (unsigned long long)(uint32_t)(d0 * d0);
                        if (has1) { const int d1 = v1 - (int)(sp >> 16); sse += (unsigned long long)(uint32_t)(d1 * d1); }
                    },
                    [&]() {
#pragma unroll
                        for (int k = 0; k < 2; k++) { const int j = tid + 256 * k; typedef uint32_t u32x4_a16 __attribute__((vector_size(16))); // (a 16-byte aligned LDS address: one ds_write_b128)
                                                      *(u32x4_a16*)(srct + (j >> 3) * 64 + (j & 7) * 8) = u32x4_a16{sq[k][0], sq[k][1], sq[k][2], sq[k][3]}; }
                    });
    } else if (KIND == 2) {
        const int idx = units[u].ep & 15, q0 = units[u].xqd[0], q1 = units[u].xqd[1];
        sgr_tile(tile, A16, B32, xlut, idx, s.uw, s.uh, bd, tid, [](int, int, int32_t) {},
                 [&](int rr, int c, int32_t f0a, int32_t f1a, int32_t f0b, int32_t f1b, bool has1) {
                     score(rr, c, sgr_combine(tile[(rr + 3) * TW + c + 3], f0a, f1a, idx, q0, q1, bd), sgr_combine(tile[(rr + 3) * TW + c + 4], f0b, f1b, idx, q0, q1, bd), has1);
                 });
    } else {
        for (int i = tid; i < s.uh * 32; i += 256) {
            const int rr = i >> 5, c = (i & 31) * 2;
            if (c < s.uw) score(rr, c, tile[(rr + 3) * TW + c + 3], tile[(rr + 3) * TW + c + 4], c + 1 < s.uw);
        }
    }
    // workgroup sum -> one atomic
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) sse += ((unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)(sse >> 32), m) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)sse, m);
    if ((tid & 63) == 0) part[tid >> 6] = sse;
    __syncthreads();
    if (tid == 0) {
        if (KIND == 1) {
            // the unit's last workgroup to arrive also consumes the result and proposes the next trial (lr_wiener_step): a round of the refinement is ONE launch.
            // The arrival count rides in the accumulator's top 16 bits (a unit's squared error stays far below 2^48), so the atomic that adds this workgroup's
            // share also hands back everyone else's.
            const unsigned long long mine = part[0] + part[1] + part[2] + part[3], old = atomicAdd(&acc[u], mine + WN_TICKET);
            const int tiles = ((r.h_end - r.h_start + 63) >> 6) * ((r.v_end - r.v_start + 63) >> 6);
            if ((int)(old >> 48) == tiles - 1) lr_wiener_step(P, wn, acc, out_units, counter, u, (long long)((old + mine) & (WN_TICKET - 1)));
        } else {
            atomicAdd(&acc[u], part[0] + part[1] + part[2] + part[3]);
        }
    }
}

__global__ void lr_take_sse_kernel(unsigned long long* acc, SvtHipLrSearchUnit* out, const int which, const int n) { // acc -> out[u].sse[which], acc = 0
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n) return;
    out[u].sse[which] = (int64_t)acc[u];
    acc[u]            = 0;
}

// ---- Wiener solve: one wave per unit ----------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int wrap_index(const int i, const int win) { const int h1 = (win >> 1) + 1; return i >= h1 ? win - 1 - i : i; }
// The unit's statistics are read ~20 times each by the eight alternating updates and the score: they are copied to LDS once (256 threads, coalesced), so the
// dependent chain of the solve waits on LDS instead of on 49 global loads in series per update (217 / 360 us per launch at win 5 / 7 before: profiles/r06_lr_search_timeline.txt).
__global__ __launch_bounds__(256) void lr_wiener_solve_kernel(const SvtHipLrSearchParams P, const long long* __restrict__ Mall, const long long* __restrict__ Hall,
                                                              const SvtHipLrPrevUnit* __restrict__ prev, WnState* __restrict__ wn, SvtHipLrSearchUnit* __restrict__ out) {
    __shared__ long long     A[4], B[16], PQ[2];
    __shared__ int32_t       a[7], b[7];
    __shared__ long long     M[49], H[49 * 49];
    __shared__ int           solved;
    const int                u = blockIdx.x, l = threadIdx.x, win = P.wiener_win, win2 = win * win, h1 = (win >> 1) + 1, plane_off = (7 - win) >> 1;
    WnState st = {};
    if (prev && prev[u].use) { // use_prev_frame_coeffs (:1297-1302): the co-located unit's taps, no solve
        if (l == 0) {
            for (int k = 0; k < 8; k++) { st.v[k] = prev[u].vfilter[k]; st.h[k] = prev[u].hfilter[k]; }
            st.active = 1;
            wn[u] = st;
        }
        return;
    }
    for (int i = l; i < win2 * win2; i += 256) H[i] = Hall[(size_t)u * 49 * 49 + i];
    if (l < win2) M[l] = Mall[(size_t)u * 49 + l];
    if (l < win) a[l] = b[l] = (int32_t)(kScale / kStep * kInitFilt[l + plane_off]);
    __syncthreads();
    for (int iter = 1; iter < 5; iter++) // NUM_WIENER_ITERS
        for (int fix_b = 1; fix_b >= 0; fix_b--) { // update_a_sep_sym (:793), then update_b_sep_sym (:845)
            if (l < 4) A[l] = 0;
            if (l < 16) B[l] = 0;
            __syncthreads();
            if (l < win2) {
                // lane = (i, j) for the right-hand side A
                const int i = l / win, j = l - i * win;
                const long long ta = fix_b ? M[i * win + j] * b[i] / kScale : M[i * win + j] * a[j] / kScale;
                atomicAdd((unsigned long long*)&A[wrap_index(fix_b ? j : i, win)], (unsigned long long)ta);
                // the matrix B: with b fixed the bin depends on (k, l') -> lane = (k, l'), loop over (i, j); with a fixed on (i, j) -> lane = (i, j), loop over (k, l')
                long long tb = 0;
                if (fix_b) {
                    const int k = i, ll = j; // (reusing the lane's pair as (k, l'))
                    for (int ii = 0; ii < win; ii++)
                        for (int jj = 0; jj < win; jj++) tb += H[(size_t)jj * win * win2 + ii * win + k * win2 + ll] * b[ii] / kScale * b[jj] / kScale;
                    atomicAdd((unsigned long long*)&B[wrap_index(ll, win) * h1 + wrap_index(k, win)], (unsigned long long)tb);
                } else {
                    for (int k = 0; k < win; k++)
                        for (int ll = 0; ll < win; ll++) tb += H[(size_t)i * win * win2 + j * win + k * win2 + ll] * a[k] / kScale * a[ll] / kScale;
                    atomicAdd((unsigned long long*)&B[wrap_index(j, win) * h1 + wrap_index(i, win)], (unsigned long long)tb);
                }
            }
            __syncthreads();
            // the reduced system (h1 - 1 unknowns) in place, then linsolve_wiener (restoration_pick.c:754-790) with the rows of an elimination step -- and the
            // columns of a row -- on different lanes: every 64-bit division by a run-time value is a few hundred dependent instructions, and the serial form
            // had fifteen of them in a row per update (eight updates per unit), the row-parallel form five.  Same operations on the same values in the same order
            // per entry: bit-identical.
            const int n = h1 - 1;
            if (l == 0) {
                const long long a_last = A[h1 - 1];
                for (int i = 0; i < n; i++) A[i] -= a_last * 2 + B[i * h1 + h1 - 1] - 2 * B[(h1 - 1) * h1 + (h1 - 1)];
                for (int i = 0; i < n; i++)
                    for (int j = 0; j < n; j++) B[i * h1 + j] -= 2 * (B[i * h1 + (h1 - 1)] + B[(h1 - 1) * h1 + j] - 2 * B[(h1 - 1) * h1 + (h1 - 1)]);
                solved = 1;
            }
            __syncthreads();
            for (int k = 0; k < n - 1; k++) {
                if (l == 0) { // partial pivoting by neighbour swaps, bottom up (:756-766)
                    for (int i = n - 1; i > k; i--) {
                        const long long p0 = B[(i - 1) * h1 + k], p1 = B[i * h1 + k];
                        if ((p0 < 0 ? -p0 : p0) < (p1 < 0 ? -p1 : p1)) {
                            for (int j = 0; j < n; j++) { const long long c = B[i * h1 + j]; B[i * h1 + j] = B[(i - 1) * h1 + j]; B[(i - 1) * h1 + j] = c; }
                            const long long c = A[i]; A[i] = A[i - 1]; A[i - 1] = c;
                        }
                    }
                    if (B[k * h1 + k] == 0) solved = 0;
                }
                __syncthreads();
                if (!solved) break; // (uniform: read after the barrier)
                // lane = (row i + 1 below the pivot row, column j; j == n: the right-hand side)
                const bool mine = l < (n - 1 - k) * (n + 1);
                long long  val = 0;
                const int  i = k + l / (n + 1), j = l % (n + 1);
                if (mine) {
                    const long long c = B[(i + 1) * h1 + k], cd = B[k * h1 + k];
                    val = j < n ? B[(i + 1) * h1 + j] - c / 256 * B[k * h1 + j] / cd * 256 : A[i + 1] - c * A[k] / cd;
                }
                __syncthreads(); // every lane has read the column of multipliers before it is overwritten
                if (mine) { if (j < n) B[(i + 1) * h1 + j] = val; else A[i + 1] = val; }
                __syncthreads();
            }
            if (l == 0 && solved) {
                int32_t S[7];
                for (int i = n - 1; i >= 0; i--) {
                    if (B[i * h1 + i] == 0) { solved = 0; break; }
                    long long c = 0;
                    for (int j = i + 1; j <= n - 1; j++) c += B[i * h1 + j] * S[j] / kScale;
                    S[i] = (int32_t)(kScale * (A[i] - c) / B[i * h1 + i]);
                }
                if (solved) {
                    S[h1 - 1] = (int32_t)kScale;
                    for (int i = h1; i < win; i++) { S[i] = S[win - 1 - i]; S[h1 - 1] -= 2 * S[i]; }
                    for (int i = 0; i < win; i++) (fix_b ? a : b)[i] = S[i];
                }
            }
            __syncthreads();
        }
    // finalize_sym_filter (:962-991) x 2 and compute_score (:925-960)
    __shared__ int16_t vf[8], hf[8], sc[2][7];
    if (l < 2) {
        const int32_t* f  = l ? b : a;
        int16_t*       fi = l ? hf : vf;
        for (int k = 0; k < 8; k++) fi[k] = 0;
        for (int i = 0; i < (win >> 1); i++) {
            const long long dividend = (long long)f[i] * kStep, divisor = kScale;
            fi[i] = (int16_t)(dividend < 0 ? (dividend - divisor / 2) / divisor : (dividend + divisor / 2) / divisor);
        }
        if (win == 7) {
            fi[0] = (int16_t)clampi(fi[0], kTapMin[0], kTapMax[0]); fi[1] = (int16_t)clampi(fi[1], kTapMin[1], kTapMax[1]); fi[2] = (int16_t)clampi(fi[2], kTapMin[2], kTapMax[2]);
        } else {
            fi[2] = (int16_t)clampi(fi[1], kTapMin[2], kTapMax[2]); fi[1] = (int16_t)clampi(fi[0], kTapMin[1], kTapMax[1]); fi[0] = 0;
        }
        fi[6] = fi[0]; fi[5] = fi[1]; fi[4] = fi[2];
        fi[3] = (int16_t)(-2 * (fi[0] + fi[1] + fi[2]));
        // the full symmetric taps compute_score multiplies (indexed per lane at run time: LDS, not a private array)
        int16_t mid = kStep;
        for (int i = 0; i < 3; i++) { sc[l][i] = sc[l][6 - i] = fi[i]; mid = (int16_t)(mid - 2 * fi[i]); }
        sc[l][3] = mid;
    }
    if (l < 2) PQ[l] = 0;
    __syncthreads();
    if (l < win2) {
        const int16_t *ca = sc[0], *cb = sc[1];
        auto ab = [&](const int k) { const int kk = k / win, ll = k - kk * win; return (int32_t)(ca[ll + plane_off] * cb[kk + plane_off]); };
        const int       k  = l;
        const long long pk = ab(k) * M[k] / kStep / kStep;
        long long       qk = 0;
        for (int ll = 0; ll < win2; ll++) qk += ab(k) * H[k * win2 + ll] * ab(ll) / kStep / kStep / kStep / kStep;
        atomicAdd((unsigned long long*)&PQ[0], (unsigned long long)pk);
        atomicAdd((unsigned long long*)&PQ[1], (unsigned long long)qk);
    }
    __syncthreads();
    if (l == 0) {
        const long long score = (PQ[1] - 2 * PQ[0]) - (H[(win2 >> 1) * win2 + (win2 >> 1)] - 2 * M[win2 >> 1]);
        if (score > 0) { // no reduction of the quadratic form: the unit keeps RESTORE_NONE for the Wiener type (:1344-1347)
            out[u].sse[1] = INT64_MAX;
        } else {
            for (int k = 0; k < 8; k++) { st.v[k] = vf[k]; st.h[k] = hf[k]; }
            st.active = 1;
        }
        wn[u] = st;
    }
}

// ---- Wiener refinement: finer_tile_search_wiener_seg (:1027-1131) as a resumable state machine, one thread per unit ----------------------------------
__device__ __forceinline__ void wn_move(int16_t* f, const int p, const int d) { f[p] = (int16_t)(f[p] + d); f[6 - p] = (int16_t)(f[6 - p] + d); f[3] = (int16_t)(f[3] - 2 * d); }
__device__ void lr_wiener_step(const SvtHipLrSearchParams& P, WnState* wn, unsigned long long* acc, SvtHipLrSearchUnit* out, int32_t* counter, const int u, const long long err2) {
    WnState st = wn[u];
    const int plane_off = (7 - P.wiener_win) >> 1, start_step = 4, end_step = P.wn_max_one_refinement_step ? 4 : 1;
    enum { NEXT, AFTER_SIGN, AFTER_DIR, DONE, TRIAL } go;
    acc[u] = 0;
    st.trials++;
    if (st.s == 0) { // result of the initial evaluation
        st.err = err2;
        if (!P.wn_use_refinement) go = DONE;
        else { st.s = start_step; st.dir = 0; st.p = plane_off; st.sign = -1; st.skip = 0; go = NEXT; }
    } else { // result of a move of tap p by sign * s
        int16_t* f = st.dir ? st.v : st.h;
        if (err2 > st.err) { wn_move(f, st.p, -st.sign * st.s); go = AFTER_SIGN; }
        else {
            st.err = err2;
            if (st.sign < 0) st.skip = 1;
            go = (st.s == start_step && !P.wn_max_one_refinement_step) ? NEXT : AFTER_SIGN; // at the largest step keep moving in the same direction
        }
    }
    while (go != DONE && go != TRIAL) {
        if (go == NEXT) {
            int16_t* f = st.dir ? st.v : st.h;
            if (st.sign < 0 ? f[st.p] - st.s >= kTapMin[st.p] : f[st.p] + st.s <= kTapMax[st.p]) { wn_move(f, st.p, st.sign * st.s); go = TRIAL; }
            else go = AFTER_SIGN;
        } else if (go == AFTER_SIGN) {
            if (st.sign < 0) {
                if (st.skip) go = AFTER_DIR; // (:1062-1063: a successful downward move ends the loop over the taps of this direction)
                else { st.sign = 1; go = NEXT; }
            } else {
                st.p++;
                if (st.p >= 3) go = AFTER_DIR;
                else { st.sign = -1; st.skip = 0; go = NEXT; }
            }
        } else { // AFTER_DIR
            st.dir++;
            if (st.dir >= 2) { st.dir = 0; st.s >>= 1; }
            if (st.s < end_step) go = DONE;
            else { st.p = plane_off; st.sign = -1; st.skip = 0; go = NEXT; }
        }
    }
    if (go == DONE) {
        st.active = 0;
        out[u].sse[1] = st.err;
        for (int k = 0; k < 8; k++) { out[u].vfilter[k] = st.v[k]; out[u].hfilter[k] = st.h[k]; }
        atomicAdd(counter, -1);
    }
    wn[u] = st;
}

// the first step of a unit (nothing evaluated yet): ask for the evaluation of the initial taps
__global__ void lr_wiener_start_kernel(WnState* __restrict__ wn, int32_t* __restrict__ counter, const int n) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n) return;
    if (!wn[u].active || wn[u].started) return;
    wn[u].started = 1;
    atomicAdd(counter, 1); // (acc[u] is 0; the trial kernel runs next)
}

// Wave sum of 32-bit values without an LDS round trip: the 16-bit halves are summed separately -- a row of 16 lanes by four DPP adds (each half's row sum < 2^20), the four
// rows through scalar registers -- and recombined in 64 bits.  ~20 instructions; the ds_bpermute butterfly of wave_sum_i64 is twelve dependent LDS-pipe round trips.
__device__ __forceinline__ uint32_t lrs_row16_sum(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false);  // quad_perm [1,0,3,2]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false);  // quad_perm [2,3,0,1]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xf, 0xf, false); // row_ror:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false); // row_ror:8
    return v;
}
__device__ __forceinline__ long long wave_sum_u32(const uint32_t v) {
    const int lo = (int)lrs_row16_sum(v & 0xffffu), hi = (int)lrs_row16_sum(v >> 16);
    const uint32_t L = (uint32_t)__builtin_amdgcn_readlane(lo, 0) + (uint32_t)__builtin_amdgcn_readlane(lo, 16) + (uint32_t)__builtin_amdgcn_readlane(lo, 32) + (uint32_t)__builtin_amdgcn_readlane(lo, 48);
    const uint32_t H = (uint32_t)__builtin_amdgcn_readlane(hi, 0) + (uint32_t)__builtin_amdgcn_readlane(hi, 16) + (uint32_t)__builtin_amdgcn_readlane(hi, 32) + (uint32_t)__builtin_amdgcn_readlane(hi, 48);
    return (long long)((unsigned long long)L + ((unsigned long long)H << 16));
}
__device__ __forceinline__ long long wave_sum_i64(long long v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const unsigned long long uv = (unsigned long long)v;
        v += (long long)(((unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)(uv >> 32), m) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)uv, m));
    }
    return v;
}
__host__ __device__ inline size_t lrs_dq_dwords(const int w, const int h) { return (((size_t)w * h + 1) / 2 + 63) & ~(size_t)63; } // the int16 plane, in dwords
// ---- self-guided: flt0 / flt1 of the whole plane for one parameter set per blockIdx.z ----------------------------------------------------------------
// COMPACT (bit depth <= 10): what the projection search needs of a sample is q1 = flt0 - (dgd << 4), q2 = flt1 - (dgd << 4) and dgd - src, and at <= 10 bits
// all three fit int16 (flt <= 32 (256 * 1.003 pmax + .5) / 512 < 16.05 pmax + 1 = 16 420 for pmax = 1023, so |q| < 2^15).  The plane-sized buffers then hold
// one packed dword per sample and parameter set plus one int16 per sample shared by all sets -- 4 + 2 bytes instead of 4 + 4 + 2 + 2 per sample and pass of
// lr_sgr_proj_kernel, which re-reads its unit about nine times per parameter set (14 GB per 4K plane before, profiles/r02_reg7_pmc_traffic.json).
template <bool COMPACT>
__global__ __launch_bounds__(256) void lr_sgr_flt_kernel(const SvtHipLrSearchParams P, int32_t* __restrict__ flt /* the shared dgd - src plane */, int32_t* __restrict__ qbase /* this group's planes */,
                                                         const int slot0 /* first parameter set of this group */,
                                                         long long* __restrict__ sums /* [unit][set][5]: svt_get_proj_subspace's sums, accumulated here */, const int slots) {
    HIP_DYNAMIC_SHARED(uint16_t, smem)
    __shared__ long long sh_sum[4][10];
    uint16_t* tile = smem;
    uint16_t* A16  = tile + TH * TW;
    int32_t*  B32  = (int32_t*)((uint8_t*)A16 + LRS_A_BYTES);
    uint16_t* xlut = (uint16_t*)(B32 + 66 * 66);
    // (slot = the set's place in the GROUP's buffers; the parameter set itself is slot0 + slot)
    const int tid = threadIdx.x, slot = blockIdx.z, idx = P.sg_start_ep + (slot0 + slot) * P.sg_ep_inc, w = (int)P.width, h = (int)P.height;
    TileSrc s;
    s.data = P.dgd; s.above = s.below = nullptr; s.stride = (int)P.dgd_stride; s.bstride = 0; s.w = 0; s.h = 0; s.highbd = P.highbd;
    s.stripe_idx = 0; s.stripe_top = 0; s.stripe_bot = 0;
    s.x0 = blockIdx.x * 64; s.y0 = blockIdx.y * 64;
    s.uw = w - s.x0 < 64 ? w - s.x0 : 64; s.uh = h - s.y0 < 64 ? h - s.y0 : 64;
    const int x0 = s.x0, y0 = s.y0;
    const bool p0 = kSgrR[idx][0] > 0, p1 = kSgrR[idx][1] > 0;
    // The projection's normal equations (svt_get_proj_subspace, :413-498: sums of q1 q1, q2 q2, q1 q2, q1 s, q2 s over the unit; q = flt - (dgd << 4), s = (src - dgd) << 4)
    // are taken HERE, while flt0 / flt1 of a sample are in registers: one pass of lr_sgr_proj_kernel over the unit less.  A 64 x 64 tile lies in one column of units
    // (unit sizes are multiples of 64) and in at most two rows of units (a unit row starts 8 >> ss_y rows above a multiple of the unit size): two sets of sums per
    // thread, one 64-bit atomic per sum and workgroup.  Integer sums: the order of accumulation does not matter.
    const int us = (int)P.unit_size, uoff = 8 >> P.ss_y, nvu = n_units_1d(h, us), nhu = n_units_1d(w, us);
    const int uc = x0 / us < nhu - 1 ? x0 / us : nhu - 1, ur_lo = (y0 + uoff) / us < nvu - 1 ? (y0 + uoff) / us : nvu - 1;
    const int y_split = ur_lo < nvu - 1 ? (ur_lo + 1) * us - uoff - y0 : 1 << 30; // tile rows >= y_split belong to the next unit row
    long long acc[10];
#pragma unroll
    for (int k = 0; k < 10; k++) acc[k] = 0;
    // acc[0..4]: the whole tile's sums; acc[5..9]: the share of the rows that belong to the next unit row (a branch only tiles across a unit-row boundary take).  Every
    // factor fits 32 bits, so a term is one v_mad_i64_i32 into its 64-bit sum (the per-term 64-bit selects between two sets of sums cost eight times that).
    auto add_sums = [&](const int r, const int d, const int sp, const int32_t f0v, const int32_t f1v) {
        const int q1 = p0 ? f0v - (d << 4) : 0, q2 = p1 ? f1v - (d << 4) : 0, sd = (sp - d) * 16;
        acc[0] += (long long)q1 * q1; acc[1] += (long long)q2 * q2; acc[2] += (long long)q1 * q2; acc[3] += (long long)q1 * sd; acc[4] += (long long)q2 * sd;
        if (r >= y_split) { acc[5] += (long long)q1 * q1; acc[6] += (long long)q2 * q2; acc[7] += (long long)q1 * q2; acc[8] += (long long)q1 * sd; acc[9] += (long long)q2 * sd; }
    };
    stage_tile<(TH + 15) / 16>(tile, s, tid);
    __syncthreads();
    if (COMPACT) {
        int16_t*  dq = (int16_t*)flt;                                                        // [h][w]: dgd - src
        uint32_t* q  = (uint32_t*)qbase + (size_t)slot * w * h;                              // [h][w]: q1 | q2 << 16
        const int highbd = P.highbd;
        sgr_tile(tile, A16, B32, xlut, idx, s.uw, s.uh, P.bit_depth, tid, [](int, int, int32_t) {},
                 [&](int r, int c, int32_t f0a, int32_t f1a, int32_t f0b, int32_t f1b, bool has1) {
                     const size_t o = (size_t)(y0 + r) * w + x0 + c;
                     const int    da = tile[(r + 3) * TW + c + 3], db = tile[(r + 3) * TW + c + 4];
                     const uint32_t qa = (uint32_t)((p0 ? f0a - (da << 4) : 0) & 0xffff) | ((uint32_t)(p1 ? f1a - (da << 4) : 0) << 16);
                     const size_t   so = (size_t)(y0 + r) * P.src_stride + x0 + c;
                     if (has1) { // the pair as one access each: an 8-byte store of the two packed samples, one load of the two source samples, one 4-byte store of the two differences
                         const uint32_t qb = (uint32_t)((p0 ? f0b - (db << 4) : 0) & 0xffff) | ((uint32_t)(p1 ? f1b - (db << 4) : 0) << 16);
                         *(LrDw2*)(q + o) = LrDw2{qa, qb};
                         int sa, sb;
                         if (highbd) { uint32_t v; __builtin_memcpy(&v, (const uint16_t*)P.src + so, 4); sa = (int)(v & 0xffffu); sb = (int)(v >> 16); }
                         else { uint16_t v; __builtin_memcpy(&v, (const uint8_t*)P.src + so, 2); sa = v & 0xff; sb = v >> 8; }
                         add_sums(r, da, sa, f0a, f1a);
                         add_sums(r, db, sb, f0b, f1b);
                         if (slot0 + slot == 0) { // (shared by every set and group: written once)
                             const uint32_t dd = (uint32_t)((da - sa) & 0xffff) | ((uint32_t)(db - sb) << 16);
                             __builtin_memcpy(dq + o, &dd, 4);
                         }
                     } else {
                         q[o] = qa;
                         const int sa = rd_px(P.src, highbd, so);
                         add_sums(r, da, sa, f0a, f1a);
                         if (slot0 + slot == 0) dq[o] = (int16_t)(da - sa);
                     }
                 });
    } else {
        int32_t* f0 = qbase + (size_t)slot * 2 * w * h;
        int32_t* f1 = f0 + (size_t)w * h;
        const int highbd = P.highbd;
        sgr_tile(tile, A16, B32, xlut, idx, s.uw, s.uh, P.bit_depth, tid, [&](int r, int c, int32_t v) { if (p0) f0[(size_t)(y0 + r) * w + x0 + c] = v; },
                 [&](int r, int c, int32_t a0v, int32_t b0, int32_t a1v, int32_t b1, bool has1) {
                     if (p1) { f1[(size_t)(y0 + r) * w + x0 + c] = b0; if (has1) f1[(size_t)(y0 + r) * w + x0 + c + 1] = b1; }
                     const size_t so = (size_t)(y0 + r) * P.src_stride + x0 + c;
                     add_sums(r, tile[(r + 3) * TW + c + 3], rd_px(P.src, highbd, so), a0v, b0);
                     if (has1) add_sums(r, tile[(r + 3) * TW + c + 4], rd_px(P.src, highbd, so + 1), a1v, b1);
                 });
    }
    // workgroup sums -> the units' accumulators
    if (!sums) return; // (workgroup-uniform)
    const bool two = y_split < s.uh;
#pragma unroll
    for (int k = 0; k < 5; k++) acc[k] -= acc[5 + k]; // whole tile -> the rows of the upper unit row
#pragma unroll
    for (int k = 0; k < 10; k++) {
        if (k >= 5 && !two) break;
        const long long v = wave_sum_i64(acc[k]);
        if ((tid & 63) == 0) sh_sum[tid >> 6][k] = v;
    }
    __syncthreads();
    if (tid < (two ? 10 : 5)) {
        const long long v = sh_sum[0][tid] + sh_sum[1][tid] + sh_sum[2][tid] + sh_sum[3][tid];
        const int       u = (ur_lo + (tid >= 5 ? 1 : 0)) * nhu + uc;
        atomicAdd((unsigned long long*)&sums[((size_t)u * slots + slot0 + slot) * 5 + (tid >= 5 ? tid - 5 : tid)], (unsigned long long)v);
    }
}

// ---- self-guided: projection + refinement of one (unit, parameter set) per workgroup (search_selfguided_restoration's loop body, :582-630) ------------
constexpr int PROJ_T = 1024; // threads per (unit, parameter set): sixteen waves; a pass = 64 samples per thread x the candidates of its shape (VALU), one reduction, two barriers
// The workgroup's sums of N per-thread values (uint32_t or long long), value k written to *dst(k) in LDS (valid for every thread on return): one wave reduction per
// value, TWO barriers for the whole group.  use(k) says whether value k is wanted (workgroup-uniform).  part: [PROJ_T / 64][PROJ_ROW].  No barrier at the start: `part`
// is read only between the two barriers, and a thread has read whatever it wanted of an earlier call's results before it gets here (program order).
constexpr int PROJ_W = PROJ_T / 64, PROJ_ROW = 24;
__device__ __forceinline__ long long wave_sum_any(const uint32_t v) { return wave_sum_u32(v); }
__device__ __forceinline__ long long wave_sum_any(const long long v) { return wave_sum_i64(v); }
template <typename T, int N, typename U, typename D>
__device__ __forceinline__ void block_sums_to(const T (&v)[N], long long (*part)[PROJ_ROW], const int tid, U use, D dst) {
    static_assert(N <= PROJ_ROW, "part row");
#pragma unroll
    for (int k = 0; k < N; k++)
        if (use(k)) {
            const long long w = wave_sum_any(v[k]);
            if ((tid & 63) == 0) part[tid >> 6][k] = w;
        }
    __syncthreads();
    if (tid < N && use(tid)) {
        long long t = 0;
#pragma unroll
        for (int w = 0; w < PROJ_W; w++) t += part[w][tid];
        *dst(tid) = t;
    }
    __syncthreads();
}
template <typename T, int N, typename U>
__device__ __forceinline__ void block_sums_i64(const T (&v)[N], long long (*part)[PROJ_ROW], long long* out, const int tid, U use) {
    block_sums_to(v, part, tid, use, [&](const int k) { return out + k; });
}
struct __attribute__((aligned(16))) LrsQuad { uint32_t v[4]; };
struct __attribute__((aligned(8))) LrsPair { uint32_t v[2]; };
// the samples of a unit in raster order, PROJ_T apart, four in flight per thread (all loads of a group issued before the first use); (x, y) advance
// incrementally -- no division per sample.  body(uu, spx, a0, a1): uu = dgd << 4, spx = src, a0 = flt0 - uu, a1 = flt1 - uu (0 for a pass that is off).  The
// compact buffers deliver the same projection with uu = 0 and spx = src - dgd: ((dgd << 11) + y + 1024 >> 11) - src = (y + 1024 >> 11) - (src - dgd) exactly.
#ifndef LRS_DEPTH
#define LRS_DEPTH 4
#endif
#ifndef SVT_HIP_EMU
#define LRS_NOW() wall_clock64() /* SVT_HIP_LR_SG_STATS: the constant-rate 100 MHz clock */
#else
#define LRS_NOW() 0ull
#endif
template <bool COMPACT, int DEPTH = LRS_DEPTH, typename F>
__device__ __forceinline__ void for_unit_samples(const SvtHipLrSearchParams& P, const SvtHipRect& r, const int32_t* f0, const int32_t* f1, const int r0, const int r1,
                                                 const int tid, F body) {
    const int uw = r.h_end - r.h_start, uh = r.v_end - r.v_start, npx = uw * uh, w = (int)P.width, highbd = P.highbd;
    if (COMPACT && !((w | uw | r.h_start) & 3)) { // four consecutive samples per load pair: one b128 of packed (q1, q2) + one b64 of dgd - src
        const int qpr = uw >> 2, nq = qpr * uh, qy = PROJ_T / qpr, rx = PROJ_T - qy * qpr;
        int       y = tid / qpr, x = tid - y * qpr;
        for (int i = tid; i < nq; i += DEPTH * PROJ_T) {
            LrsQuad Q[DEPTH];
            LrsPair D[DEPTH];
            bool    ok[DEPTH];
#pragma unroll
            for (int k = 0; k < DEPTH; k++) {
                ok[k] = i + k * PROJ_T < nq;
                const size_t fo = (size_t)(r.v_start + (ok[k] ? y : 0)) * w + r.h_start + 4 * (ok[k] ? x : 0);
                Q[k] = *(const LrsQuad*)((const uint32_t*)f1 + fo);
                D[k] = *(const LrsPair*)((const int16_t*)f0 + fo);
                x += rx; y += qy;
                if (x >= qpr) { x -= qpr; y++; }
            }
#pragma unroll
            for (int k = 0; k < DEPTH; k++)
                if (ok[k]) {
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const uint32_t pk = Q[k].v[j], dw = D[k].v[j >> 1];
                        body(0, -(int)(int16_t)((j & 1) ? dw >> 16 : dw & 0xffffu), (int)(int16_t)(pk & 0xffffu), (int)pk >> 16);
                    }
                }
        }
        return;
    }
    const int qy = PROJ_T / uw, rx = PROJ_T - qy * uw;
    int       y = tid / uw, x = tid - y * uw;
    for (int i = tid; i < npx; i += DEPTH * PROJ_T) {
        int d[DEPTH], sp[DEPTH], g0[DEPTH], g1[DEPTH];
        bool ok[DEPTH];
#pragma unroll
        for (int k = 0; k < DEPTH; k++) {
            ok[k] = i + k * PROJ_T < npx;
            const int    yy = ok[k] ? y : 0, xx = ok[k] ? x : 0;
            const size_t fo = (size_t)(r.v_start + yy) * w + r.h_start + xx;
            if (COMPACT) {
                const uint32_t pk = ((const uint32_t*)f1)[fo];
                d[k] = 0; sp[k] = -(int)((const int16_t*)f0)[fo];
                g0[k] = (int)(int16_t)(pk & 0xffffu); g1[k] = (int)pk >> 16;
            } else {
                d[k]  = rd_px(P.dgd, highbd, (size_t)(r.v_start + yy) * P.dgd_stride + r.h_start + xx) << 4;
                sp[k] = rd_px(P.src, highbd, (size_t)(r.v_start + yy) * P.src_stride + r.h_start + xx);
                g0[k] = r0 > 0 ? f0[fo] - d[k] : 0;
                g1[k] = r1 > 0 ? f1[fo] - d[k] : 0;
            }
            x += rx; y += qy;
            if (x >= uw) { x -= uw; y++; }
        }
#pragma unroll
        for (int k = 0; k < DEPTH; k++)
            if (ok[k]) body(d[k], sp[k], g0[k], g1[k]);
    }
}
// ---- finer_search_pixel_proj_error for a parameter set with BOTH passes on: a whole step size per pass --------------------------------------------------------
// The walk at step s moves xqd[0] down or up along a line, then -- only if no downward move of xqd[0] was accepted (:372-373) -- xqd[1] along a line from wherever
// xqd[0] ended.  With xq0 = xqd[0], xq1 = 128 - xqd[0] - xqd[1] (svt_decode_xq) a move of (d0, d1) changes a sample's projection by d0 * (a0 - a1) - d1 * a1, so ONE
// pass over the unit gives the error of every point the walk can reach within K moves per line:
//     E[0]                               the current point
//     E[1 + j], j < K                    xqd[0] - (j + 1) s                     (the downward line of tap 0)
//     E[1 + K + i], i < K                xqd[0] + (i + 1) s                     (its upward line)
//     E[1 + 2K + i 2K + j], j < K        (xqd[0] + i s, xqd[1] - (j + 1) s)     i = 0 .. K: tap 1's downward line from each end point of tap 0's upward walk
//     E[1 + 2K + i 2K + K + j], j < K    (xqd[0] + i s, xqd[1] + (j + 1) s)     ... and its upward line
// and the reference's accept / reject sequence is then replayed on the table (sgr_grid_replay).  Errors are per-sample-rounded, so the table cannot come from sums --
// it is the same arithmetic as get_pixel_proj_error at each point.  K = 2 at the largest step (which keeps moving while it improves; a walk that wants a third move
// in one line leaves the table and the caller redoes that step the line-by-line way), K = 1 below it (one move per line, always inside the table).
template <int K> struct SgrGrid { static constexpr int N = 1 + 2 * K + (K + 1) * 2 * K; };
template <bool COMPACT, int K>
__device__ __forceinline__ void sgr_grid_pass(const SvtHipLrSearchParams& P, const SvtHipRect& r, const int32_t* f0, const int32_t* f1, const int r0, const int r1, const int tid,
                                              const int xq0, const int xq1, const int s, long long (*part)[PROJ_ROW], long long* out) {
    constexpr int N = SgrGrid<K>::N;
    // COMPACT (bit depth <= 10): a thread's sums fit 32 bits.  |q1|, |q2| <= 16 420 (lr_sgr_flt_kernel's bound), the candidates' |xq0| <= 100 and |xq1| <= 264, so
    // |v >> 11| <= 2 929, |e| <= 2 929 + 1 023 and e * e < 15.7 M; a unit has at most 383 x 391 samples (a last row / column absorbs up to half a unit, plus the 8-row
    // offset), 147 per thread: 147 x 15.7 M < 2^32.  One v_mad_u32_u24 per candidate instead of a 64-bit multiply-add, and half the accumulator registers.
    typedef typename std::conditional<COMPACT, uint32_t, long long>::type Acc;
    Acc acc[N];
#pragma unroll
    for (int k = 0; k < N; k++) acc[k] = 0;
    for_unit_samples<COMPACT>(P, r, f0, f1, r0, r1, tid, [&](const int uu, const int sp, const int a0, const int a1) {
        // ((v >> 11) - sp == (v - (sp << 11)) >> 11 exactly: the source sample is folded into v once, a candidate is then an add, a shift and a multiply-add)
        const int v = uu * 128 + xq0 * a0 + xq1 * a1 + (1 << 10) - sp * 2048, dA = s * (a0 - a1), dB = s * a1; // (* 128, * 2048: uu << 7, sp << 11 for values of either sign)
        auto sq = [&](const int vv) -> Acc { const int e = vv >> 11; return COMPACT ? (Acc)((uint32_t)e * (uint32_t)e) : (Acc)((long long)e * e); };
        acc[0] += sq(v);
#pragma unroll
        for (int j = 0; j < K; j++) { acc[1 + j] += sq(v - (j + 1) * dA); acc[1 + K + j] += sq(v + (j + 1) * dA); }
#pragma unroll
        for (int i = 0; i <= K; i++) {
            const int b = v + i * dA;
#pragma unroll
            for (int j = 0; j < K; j++) { acc[1 + 2 * K + i * 2 * K + j] += sq(b + (j + 1) * dB); acc[1 + 2 * K + i * 2 * K + K + j] += sq(b - (j + 1) * dB); }
        }
    });
    block_sums_i64(acc, part, out, tid, [](int) { return true; });
}
// a workgroup-uniform value read from LDS: tell the compiler (scalar registers, scalar control flow -- the walk's state then costs no vector registers)
__device__ __forceinline__ int       lrs_uni(const int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ long long lrs_uni64(const long long v) {
    return (long long)(((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((unsigned long long)v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v));
}
// the walk of one step size over the table; workgroup-uniform.  false: a line wanted a move beyond the table (xqd / err are then partly advanced: the caller restores them)
template <int K>
__device__ __forceinline__ bool sgr_grid_replay(const long long* E, const int s, const bool keep_moving, int* xqd, long long& err, const int* tap_min, const int* tap_max) {
    const int cap = keep_moving ? (1 << 30) : 1; // only the largest step keeps moving in the same direction (:359-361)
    int       k = 0, i = 0;
    bool      skip = false;
    for (const int room = (xqd[0] - tap_min[0]) / s; k < room && k < cap;) { // tap 0 down
        if (k == K) return false;
        const long long e = lrs_uni64(E[1 + k]);
        if (e > err) break;
        err = e; k++; skip = true;
    }
    xqd[0] -= s * k;
    if (skip) return true; // (:372-373: a successful downward move ends the loop over the taps)
    for (const int room = (tap_max[0] - xqd[0]) / s; i < room && i < cap;) { // tap 0 up
        if (i == K) return false;
        const long long e = lrs_uni64(E[1 + K + i]);
        if (e > err) break;
        err = e; i++;
    }
    xqd[0] += s * i;
    const long long* row = E + 1 + 2 * K + i * 2 * K;
    k = 0;
    for (const int room = (xqd[1] - tap_min[1]) / s; k < room && k < cap;) { // tap 1 down, from where tap 0 ended
        if (k == K) return false;
        const long long e = lrs_uni64(row[k]);
        if (e > err) break;
        err = e; k++; skip = true;
    }
    xqd[1] -= s * k;
    if (skip) return true;
    int j = 0;
    for (const int room = (tap_max[1] - xqd[1]) / s; j < room && j < cap;) { // tap 1 up
        if (j == K) return false;
        const long long e = lrs_uni64(row[K + j]);
        if (e > err) break;
        err = e; j++;
    }
    xqd[1] += s * j;
    return true;
}
// (96 VGPRs -- five waves per SIMD, four of them this workgroup's: the Wiener trial workgroups of the other stream fit beside it)
template <bool COMPACT>
__global__ __launch_bounds__(PROJ_T) SVT_HIP_WAVES_PER_EU(5, 5) void lr_sgr_proj_kernel(const SvtHipLrSearchParams P, const SvtHipRect* __restrict__ rects, const int32_t* __restrict__ flt,
                                                             const int32_t* __restrict__ qbase, SgResult* __restrict__ res, const int slots, const int slot0, const int line_walk,
                                                             const long long* __restrict__ sums, int32_t* __restrict__ dbg /* nullptr, or [1..3]: table passes, line passes, tables left */) {
    __shared__ long long part[PROJ_W][PROJ_ROW];
    __shared__ long long sh_t[5], sh_err;
    __shared__ long long sh_e[3][9]; // the candidate errors of a line (down, up, the first pass's up run; [8] of the first row: the current point): workgroup-uniform and indexed at run time -- LDS, not registers
    __shared__ long long sh_g[SgrGrid<2>::N]; // the error table of a whole step size (sgr_grid_pass)
    __shared__ int32_t   sh_xq[2];
    const int        u = blockIdx.x, slot = blockIdx.y, tid = threadIdx.x, idx = P.sg_start_ep + (slot0 + slot) * P.sg_ep_inc, w = (int)P.width, h = (int)P.height;
    const SvtHipRect r = rects[u];
    const int        npx = (r.h_end - r.h_start) * (r.v_end - r.v_start);
    // compact: f0 = the shared int16 plane (dgd - src), f1 = this parameter set's packed (q1, q2) plane
    const int32_t*   f0 = COMPACT ? flt : qbase + (size_t)slot * 2 * w * h;
    const int32_t*   f1 = COMPACT ? qbase + (size_t)slot * w * h : f0 + (size_t)w * h;
    const int        r0 = kSgrR[idx][0], r1 = kSgrR[idx][1];
    const unsigned long long t_wg0 = (dbg && threadIdx.x == 0) ? LRS_NOW() : 0;
    // svt_get_proj_subspace (:413-498): the integer sums equal the reference's double sums exactly (every partial sum < 2^53); lr_sgr_flt_kernel accumulated them
    // (sums == nullptr -- a unit size that is not a multiple of the filter kernel's 64 x 64 tiles --: one pass over the unit here)
    if (sums) {
        if (tid < 5) sh_t[tid] = sums[((size_t)u * slots + slot0 + slot) * 5 + tid];
        __syncthreads();
    } else {
        long long a[5] = {0, 0, 0, 0, 0};
        for_unit_samples<COMPACT>(P, r, f0, f1, r0, r1, tid, [&](const int uu, const int sp, const int a0, const int a1) {
            const long long sd = (long long)sp * 16 - uu, q1 = a0, q2 = a1;
            a[0] += q1 * q1; a[1] += q2 * q2; a[2] += q1 * q2; a[3] += q1 * sd; a[4] += q2 * sd;
        });
        block_sums_i64(a, part, sh_t, tid, [](int) { return true; });
    }
    if (tid == 0) {
        const long long* t = sh_t;
        const double size = (double)npx;
        const double H00 = (double)t[0] / size, H11 = (double)t[1] / size, H01 = (double)t[2] / size, C0 = (double)t[3] / size, C1 = (double)t[4] / size;
        int x0 = 0, x1 = 0;
        if (r0 == 0) {
            if (!(H11 < 1e-8)) x1 = (int)rint(C1 / H11 * 128);
        } else if (r1 == 0) {
            if (!(H00 < 1e-8)) x0 = (int)rint(C0 / H00 * 128);
        } else {
            const double det = H00 * H11 - H01 * H01;
            if (!(det < 1e-8)) { x0 = (int)rint((H11 * C0 - H01 * C1) / det * 128); x1 = (int)rint((H00 * C1 - H01 * C0) / det * 128); }
        }
        sh_xq[0] = x0; sh_xq[1] = x1;
    }
    __syncthreads();
    const int tap_min[2] = {-96, -32}, tap_max[2] = {31, 95}; // SGRPROJ_PRJ_MIN0 / MAX0, MIN1 / MAX1
    int       xqd[2];
    {         // encode_xq (:500-511)
        const int e0 = lrs_uni(sh_xq[0]), e1 = lrs_uni(sh_xq[1]);
        if (r0 == 0) { xqd[0] = 0; xqd[1] = clampi(128 - e1, tap_min[1], tap_max[1]); }
        else if (r1 == 0) { xqd[0] = clampi(e0, tap_min[0], tap_max[0]); xqd[1] = clampi(128 - xqd[0], tap_min[1], tap_max[1]); }
        else { xqd[0] = clampi(e0, tap_min[0], tap_max[0]); xqd[1] = clampi(128 - xqd[0] - e1, tap_min[1], tap_max[1]); }
    }
    auto proj_err = [&]() -> long long { // get_pixel_proj_error (:305-318): every thread returns the unit's error for the current xqd
        int xq0, xq1;
        if (r0 == 0) { xq0 = 0; xq1 = 128 - xqd[1]; }
        else if (r1 == 0) { xq0 = xqd[0]; xq1 = 0; }
        else { xq0 = xqd[0]; xq1 = 128 - xq0 - xqd[1]; }
        long long e2 = 0;
        for_unit_samples<COMPACT>(P, r, f0, f1, r0, r1, tid, [&](const int uu, const int sp, const int a0, const int a1) {
            const int v = (uu << 7) + xq0 * a0 + xq1 * a1;
            const int e = ((v + (1 << 10)) >> 11) - sp;
            e2 += (long long)e * e;
        });
        const long long one[1] = {e2};
        block_sums_i64(one, part, &sh_err, tid, [](int) { return true; });
        return lrs_uni64(sh_err);
    };
    // finer_search_pixel_proj_error (:320-411), start_step 2.  The greedy walk only ever moves ONE tap along a line, and a move of tap p by delta changes
    // every sample's projection by delta * (a per-sample constant): so one pass over the unit evaluates a whole run of candidates -- up to PROJ_K steps
    // down and PROJ_K steps up from the current point -- and the reference's accept / reject sequence is then replayed on those errors.  (The up candidates
    // of the first pass stay valid exactly when no down move was accepted, which is when the reference evaluates them.)  Control flow is workgroup-uniform.
    constexpr int PROJ_K = 8; // moves per continuation pass (round 2 measured 4: more passes on long chains, 7.0 ms instead of 6.4 ms on the 4K bench plane)
    static_assert(PROJ_K <= 8, "sh_e rows");
    long long err      = 0;
    bool      have_err = false; // err holds the error of the current xqd
    // one pass: the errors of nd moves down and nu moves up of tap p by st each -- and, if wanted, of the current point itself (ed[PROJ_K]).  ND / NU (compile-time) bound
    // nd / nu: a candidate costs an add, a shift and a multiply-add per sample whether its sum is wanted or not (the `k < nd` below is a select on the sum), so a pass
    // is compiled for the shapes the walk really asks for -- (1, 1) the only look of a line at step 1, (PROJ_KF, PROJ_KF) the first look at step 2, (PROJ_K, 0) and
    // (0, PROJ_K) the continuations -- instead of evaluating 2 PROJ_K + 1 candidates every time.
    auto eval_line = [&](const int p, const int st, const int nd, const int nu, const bool want0, long long* ed, long long* eu, auto ndc, auto nuc) __attribute__((always_inline)) { // ed, eu: rows of sh_e
        constexpr int ND = decltype(ndc)::value, NU = decltype(nuc)::value, NK = ND > NU ? ND : NU;
        int xq0, xq1;
        if (r0 == 0) { xq0 = 0; xq1 = 128 - xqd[1]; }
        else if (r1 == 0) { xq0 = xqd[0]; xq1 = 0; }
        else { xq0 = xqd[0]; xq1 = 128 - xq0 - xqd[1]; }
        // d(xq0) / d(xqd[p]) and d(xq1) / d(xqd[p]) (svt_decode_xq, restoration.c:634-645)
        const int c0 = (p == 0 && r0 > 0) ? 1 : 0, c1 = (p == 1 || (r0 > 0 && r1 > 0)) ? -1 : 0;
        // COMPACT (bit depth <= 10): a thread's sums fit 32 bits -- |q1|, |q2| <= 16 420, the candidates' |xq0| <= 112 and |xq1| <= 280, so |e| <= 3 143 + 1 023 and
        // e * e < 17.4 M; at most 147 samples of a unit per thread (383 x 391 / 1 024): 147 x 17.4 M < 2^32.  The source sample is folded into v
        // ((v >> 11) - sp == (v - (sp << 11)) >> 11 exactly), the candidates of a line are successive adds: per candidate an add, a shift, a multiply-add.
        typedef typename std::conditional<COMPACT, uint32_t, long long>::type Acc;
        Acc ad[PROJ_K + 1], au[PROJ_K];
#pragma unroll
        for (int k = 0; k < PROJ_K; k++) ad[k] = au[k] = 0;
        ad[PROJ_K] = 0;
        const unsigned long long t_s0 = (dbg && tid == 0) ? LRS_NOW() : 0;
        for_unit_samples<COMPACT>(P, r, f0, f1, r0, r1, tid, [&](const int uu, const int sp, const int a0, const int a1) {
            const int v = uu * 128 + xq0 * a0 + xq1 * a1 + (1 << 10) - sp * 2048, dv = st * (c0 * a0 + c1 * a1);
            auto sq = [&](const int vv) -> Acc { const int e = vv >> 11; return COMPACT ? (Acc)((uint32_t)e * (uint32_t)e) : (Acc)((long long)e * e); };
            int vd = v, vu = v;
#pragma unroll
            for (int k = 0; k < NK; k++) {
                vd -= dv; vu += dv;
                if (k < ND && k < nd) ad[k] += sq(vd);
                if (k < NU && k < nu) au[k] += sq(vu);
            }
            if (want0) ad[PROJ_K] += sq(v);
        });
        if (dbg && tid == 0) atomicAdd(&dbg[24], (int)(LRS_NOW() - t_s0)); // (diagnostic: thread 0's time inside the sample loops of the line passes)
        // one reduction for both directions: values 0 .. PROJ_K are the downward sums (+ the current point), PROJ_K + 1 .. the upward ones
        Acc all[2 * PROJ_K + 1];
#pragma unroll
        for (int k = 0; k <= PROJ_K; k++) all[k] = ad[k];
#pragma unroll
        for (int k = 0; k < PROJ_K; k++) all[PROJ_K + 1 + k] = au[k];
        block_sums_to(all, part, tid,
                      [&](const int k) { return k < PROJ_K ? (k < ND && k < nd) : k == PROJ_K ? want0 : (k - PROJ_K - 1 < NU && k - PROJ_K - 1 < nu); },
                      [&](const int k) { return k <= PROJ_K ? ed + k : eu + (k - PROJ_K - 1); });
    };
    constexpr int PROJ_KF = 2; // how far the FIRST pass of a step-2 line looks in each direction (both directions and the start point in one pass)
    typedef std::integral_constant<int, 0> K0_; typedef std::integral_constant<int, 1> K1_; typedef std::integral_constant<int, PROJ_KF> KF_; typedef std::integral_constant<int, PROJ_K> KK_;
    // one step size the line-by-line way (a pass per line; the first pass of a set also delivers the error of the starting point)
    auto line_stage = [&](const int st) __attribute__((always_inline)) {
        for (int p = 0; p < 2; p++) {
            if ((r0 == 0 && p == 0) || (r1 == 0 && p == 1)) continue;
            // only the largest step keeps moving in the same direction (:359-361).  The FIRST pass of such a line looks PROJ_KF moves each way (both directions and the start
            // point in one pass); a walk that accepts them all goes on PROJ_K moves per pass in the direction it has taken
            long long *ed = sh_e[0], *eu = sh_e[1], *eu0 = sh_e[2];
            int       skip = 0, nu0 = 0;
            for (bool first = true;; first = false) { // the downward moves
                const int cap = st == 2 ? (first ? PROJ_KF : PROJ_K) : 1;
                const int roomd = (xqd[p] - tap_min[p]) / st, nd = roomd < cap ? roomd : cap;
                int       nu = 0;
                if (first) { const int roomu = (tap_max[p] - xqd[p]) / st; nu = roomu < cap ? roomu : cap; }
                if (nd == 0 && nu == 0) break;
                // (a first pass leaves its upward sums in eu0, where the upward loop below looks for them; the continuation passes of the downward walk have none)
                if (st != 2) eval_line(p, st, nd, nu, !have_err, ed, eu0, K1_{}, K1_{});
                else if (first) eval_line(p, st, nd, nu, !have_err, ed, eu0, KF_{}, KF_{});
                else eval_line(p, st, nd, 0, !have_err, ed, eu, KK_{}, K0_{});
                if (dbg && tid == 0) atomicAdd(&dbg[2], 1);
                if (!have_err) { err = lrs_uni64(ed[PROJ_K]); have_err = true; }
                if (first) nu0 = nu;
                int  k = 0;
                bool rejected = false;
                while (k < nd) {
                    const long long e = lrs_uni64(ed[k]);
                    if (e > err) { rejected = true; break; }
                    err = e; k++; skip = 1;
                    if (st != 2) break;
                }
                xqd[p] -= st * k;
                if (dbg && tid == 0 && st == 2 && first) atomicAdd(&dbg[4 + (k < 9 ? k : 9)], 1); // (diagnostic: moves accepted down in a line's first pass)
                if (rejected || st != 2 || k < nd) break; // (all accepted: look further -- the next pass finds out whether there is room left)
            }
            if (skip) break; // (:372-373: a successful downward move ends the loop over p)
            int nu = nu0;
            for (int pass = 0; nu > 0; pass++) { // the upward moves, from the unchanged point
                if (pass > 0) {
                    const int cap = st == 2 ? PROJ_K : 1;
                    const int roomu = (tap_max[p] - xqd[p]) / st;
                    nu = roomu < cap ? roomu : cap;
                    if (nu == 0) break;
                    eval_line(p, st, 0, nu, false, ed, eu0, K0_{}, KK_{});
                    if (dbg && tid == 0) atomicAdd(&dbg[2], 1);
                }
                int  k = 0;
                bool rejected = false;
                while (k < nu) {
                    const long long e = lrs_uni64(eu0[k]);
                    if (e > err) { rejected = true; break; }
                    err = e; k++;
                    if (st != 2) break;
                }
                xqd[p] += st * k;
                if (dbg && tid == 0 && st == 2 && pass == 0) atomicAdd(&dbg[14 + (k < 9 ? k : 9)], 1); // (... and up, where no downward move was accepted)
                if (rejected || st != 2 || k < nu) break;
            }
        }
    };
    // finer_search_pixel_proj_error (:320-411), start_step 2.  Control flow is workgroup-uniform throughout.
    //   * both passes on (10 of the 16 parameter sets): one pass per step size (sgr_grid_pass: the starting point's error comes with the first) -- 2 passes after the
    //     projection sums instead of 5-6;
    //   * one pass off: the walk only ever moves ONE tap, a whole run of candidates per pass (eval_line), the starting point's error folded into the first.
    if (P.sg_refine)
        for (int st = 2; st >= 1; st >>= 1) { // (ONE call site of line_stage: it is inlined once, its candidate sums stay in registers)
            bool by_line = true;
            if (r0 > 0 && r1 > 0 && line_walk != 1 && (st == 1 || line_walk >= 2)) {
                if (st == 2) sgr_grid_pass<COMPACT, 2>(P, r, f0, f1, r0, r1, tid, xqd[0], 128 - xqd[0] - xqd[1], 2, part, sh_g);
                else sgr_grid_pass<COMPACT, 1>(P, r, f0, f1, r0, r1, tid, xqd[0], 128 - xqd[0] - xqd[1], 1, part, sh_g);
                if (!have_err) { err = lrs_uni64(sh_g[0]); have_err = true; }
                const int       sx0 = xqd[0], sx1 = xqd[1];
                const long long serr = err;
                by_line = st == 2 ? (!sgr_grid_replay<2>(sh_g, 2, true, xqd, err, tap_min, tap_max) || line_walk == 2) : !sgr_grid_replay<1>(sh_g, 1, false, xqd, err, tap_min, tap_max);
                if (dbg && tid == 0) { atomicAdd(&dbg[1], 1); if (by_line) atomicAdd(&dbg[3], 1); }
                if (by_line) { xqd[0] = sx0; xqd[1] = sx1; err = serr; } // a line wants a third move at the largest step: this step size line by line (rare)
            }
            if (by_line) line_stage(st);
        }
    if (!have_err) err = proj_err(); // no refinement, or no room to move at all
    if (tid == 0) {
        SgResult o;
        o.err = err; o.xqd[0] = xqd[0]; o.xqd[1] = xqd[1];
        res[(size_t)u * slots + slot0 + slot] = o;
        if (dbg) { atomicAdd(&dbg[25], (int)(LRS_NOW() - t_wg0)); atomicAdd(&dbg[26], 1); }
    }
}
__global__ void lr_sgr_pick_kernel(const SvtHipLrSearchParams P, const SgResult* __restrict__ res, SvtHipLrSearchUnit* __restrict__ out, const int slots, const int n) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n) return;
    long long besterr = -1;
    int       bestep = 0, q0 = 0, q1 = 0;
    for (int sl = 0; sl < slots; sl++) { // first strictly smaller error wins (:631-636)
        const SgResult r = res[(size_t)u * slots + sl];
        if (besterr == -1 || r.err < besterr) { besterr = r.err; bestep = P.sg_start_ep + sl * P.sg_ep_inc; q0 = r.xqd[0]; q1 = r.xqd[1]; }
    }
    out[u].ep = bestep; out[u].xqd[0] = q0; out[u].xqd[1] = q1;
}

inline int sg_slots(const SvtHipLrSearchParams& P) {
    if (!P.sg_enabled || P.sg_ep_inc == 0 || P.sg_end_ep <= P.sg_start_ep) return 0;
    return ((int)P.sg_end_ep - (int)P.sg_start_ep + (int)P.sg_ep_inc - 1) / (int)P.sg_ep_inc;
}
// The self-guided planes are kept for a GROUP of parameter sets at a time (the filter launch of a group, then its projection launch, in stream order): all 16
// sets of a 4K plane were 1.06 GB of workspace.  Group size: as many sets as fit 240 MB, at least one, then evened out over the groups
// (16 sets of a 4K 10-bit plane: 6 + 6 + 4, 216 MB; grouping costs ~11 % of the stage's time whichever way the sets are split: profiles/r04_call4_*, r04_final_*).
inline int sg_group(const SvtHipLrSearchParams& P, const int slots) {
    if (slots <= 0) return 0;
    const size_t per_set = (size_t)P.width * P.height * (P.bit_depth <= 10 ? 4 : 8);
    size_t       g = per_set ? ((size_t)140 << 20) / per_set : (size_t)slots; // (per buffer: there are two, see svt_hip_lr_search_plane)
    const char*  e = getenv("SVT_HIP_LR_SG_GROUP"); // (tests: force small groups on small planes; read per call -- a picture-sized stage)
    if (e && atoi(e) > 0) g = (size_t)atoi(e);
    g = g < 1 ? 1 : (g < (size_t)slots ? g : (size_t)slots);
    const size_t groups = ((size_t)slots + g - 1) / g;
    return (int)(((size_t)slots + groups - 1) / groups);
}
inline size_t carve(const SvtHipLrSearchParams& P, void* base, Ws* ws) {
    const size_t n = (size_t)n_units_1d((int)P.height, (int)P.unit_size) * n_units_1d((int)P.width, (int)P.unit_size), slots = (size_t)sg_slots(P);
    size_t       off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return base ? (uint8_t*)base + o : nullptr; };
    Ws w;
    w.rects = (SvtHipRect*)take(n * sizeof(SvtHipRect));
    w.acc = (unsigned long long*)take(n * 8);
    w.acc2 = (unsigned long long*)take(n * 8);
    w.wn = (WnState*)take(n * sizeof(WnState));
    w.M = (long long*)take(n * 49 * 8);
    w.H = (long long*)take(n * 49 * 49 * 8);
    w.sg = (SgResult*)take(n * (slots ? slots : 1) * sizeof(SgResult));
    w.sgsum = (long long*)take(n * (slots ? slots : 1) * 5 * 8);
    w.counter = (int32_t*)take(256);
    const size_t group = (size_t)sg_group(P, (int)slots), wh = (size_t)P.width * P.height;
    // <= 10 bit: one int16 plane (dgd - src) shared by every set + one packed dword plane per set of the group; 12 bit: two int32 planes per set of the group
    w.flt = (int32_t*)take(P.bit_depth <= 10 ? (lrs_dq_dwords((int)P.width, (int)P.height) + 2 * group * wh) * 4 : 2 * group * 2 * wh * 4); // two group buffers
    if (ws) *ws = w;
    return off;
}

} // namespace

extern "C" {

size_t svt_hip_lr_search_workspace(const SvtHipLrSearchParams* params) { return carve(*params, nullptr, nullptr); }

int svt_hip_lr_search_plane(const SvtHipLrSearchParams* params, const SvtHipLrPrevUnit* prev, SvtHipLrSearchUnit* units, void* workspace, void* stream) {
    SVT_HIP_ENTRY_TRY
    svthip::ensure_device();
    const SvtHipLrSearchParams& P = *params;
    if (!P.width || !P.height || !P.unit_size || (P.wn_enabled && P.wiener_win != 7 && P.wiener_win != 5 && P.wiener_win != 3) || (P.sg_enabled && P.sg_end_ep > 16)) return -1;
    hipStream_t st = (hipStream_t)stream;
    Ws          W;
    carve(P, workspace, &W);
    const int nvu = n_units_1d((int)P.height, (int)P.unit_size), nhu = n_units_1d((int)P.width, (int)P.unit_size), n = nvu * nhu, slots = sg_slots(P);
    const int us = (int)P.unit_size, max_uw = (int)P.width - (nhu - 1) * us, max_uh = (int)P.height - (nvu - 1) * us + (nvu > 1 ? (8 >> P.ss_y) : 0);
    const int mw = max_uw > us ? max_uw : (us < (int)P.width ? us : (int)P.width), mh = max_uh > us ? max_uh : (us < (int)P.height ? us : (int)P.height);
    const dim3 tgrid((mw + 63) / 64, (mh + 63) / 64, n);
    const int   slots0 = sg_slots(P);
    const bool  sg_on0 = P.sg_enabled && slots0 > 0;
    long long*  sgsum0 = (sg_on0 && !((int)P.unit_size & 63)) ? W.sgsum : nullptr; // (the tile-wise accumulation needs unit boundaries on the tile grid)
    hipLaunchKernelGGL(lr_rects_kernel, dim3((n + 63) / 64 > 1 ? (n + 63) / 64 : 1), dim3(64), 0, st, P, W.rects, W.acc, W.wn, units, n, W.counter, sg_on0 ? W.acc2 : nullptr, sgsum0,
                       sgsum0 ? n * slots0 * 5 : 0);
    // RESTORE_NONE: the unrestored unit's error.  Nothing of the searches depends on it: with the Wiener search on it runs at the head of that chain (same accumulator,
    // stream order) instead of ahead of the fork, where it held back the self-guided chain -- the longer one of the fast settings -- by its 30 us.
    auto restore_none = [&](hipStream_t s_) {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(lr_trial_kernel<0>), tgrid, dim3(256), LRS_SMEM, s_, P, W.rects, W.wn, units, W.acc, (SvtHipLrSearchUnit*)nullptr, (int32_t*)nullptr);
        hipLaunchKernelGGL(lr_take_sse_kernel, dim3((n + 63) / 64), dim3(64), 0, s_, W.acc, units, 0, n);
        SVT_LAUNCH_CHECK();
    };
    if (!P.wn_enabled) restore_none(st);
    // Three independent launch sequences behind the unit rectangles / the RESTORE_NONE pass, on the calling thread's side streams (fork here, join before returning):
    //   * the self-guided search, group after group of parameter sets (filter launch, projection launch), the groups ALTERNATING between two streams and two sets of
    //     plane buffers: a projection launch ends in a long tail -- one 1 024-thread workgroup per CU, 4 to 12 passes each --, which the other stream's group fills;
    //   * the Wiener chain -- RESTORE_NONE, statistics, solve, then tens of short dependent refinement launches with a read-back of the number of units still
    //     searching every few rounds -- on the highest-priority stream, so that its launches are not queued behind the long self-guided workgroups.
    // They write different fields of a unit's record and keep separate accumulators.
    const bool  sg_on = P.sg_enabled && slots > 0;
    svthip::StreamSetLease       ts_lease; // (handed back when the call returns, its work possibly still in flight: see runtime.hip)
    const svthip::ThreadStreams& TS = *ts_lease.set;
    hipStream_t sg_st[2] = {TS.st[0], TS.st[1]}, wn_st = TS.st[2];
    hipEvent_t  ev_fork = TS.ev[0], ev_sg0 = TS.ev[1], ev_sg1 = TS.ev[2], ev_wn = TS.ev[3], ev_dq = TS.ev[4];
    const char* lw = getenv("SVT_HIP_LR_SG_WALK"); // (A/B measurement, read per call -- a picture-sized stage)
    int32_t*    sg_dbg = getenv("SVT_HIP_LR_SG_STATS") ? W.counter : nullptr; // (diagnostic: pass counts of the projection walk, printed below)
    // default: the largest step line by line (its walks are long where the projection's least-squares start is clamped: 8 moves per pass and direction), the last step
    // from one table pass (always inside its table).  "line": both steps line by line (the round-4 form); "table": the largest step from a K = 2 table as well (a walk
    // that leaves it is redone line by line -- measured: every both-pass set of the bench plane does, profiles/r05_lr_search_walk_forms.txt); "fallback": as "table",
    // every walk treated as if it had left the table (tests)
    const int   line_walk = !lw ? 0 : lw[0] == 'l' ? 1 : lw[0] == 'f' ? 2 : lw[0] == 't' ? 3 : 0;
    long long*  sgsum = ((int)P.unit_size & 63) ? nullptr : W.sgsum; // (the tile-wise accumulation needs unit boundaries on the tile grid)
    HIP_CHECK(hipEventRecord(ev_fork, st));
    unsigned long long* sg_acc = W.acc2;
    // The self-guided launches in two parts, so that the host can slip the head of the Wiener chain in between: part 0 = the filter launches of the first two groups
    // (hundreds of microseconds of work: everything enqueued behind them is early), part 1 = the rest.
    auto self_guided = [&](const int part) {
        const int    group = sg_group(P, slots);
        const size_t wh = (size_t)P.width * P.height;
        const bool   compact = P.bit_depth <= 10; // int16 differences (see lr_sgr_flt_kernel); 12-bit keeps the int32 planes
        if (part == 0) {
            HIP_CHECK(hipStreamWaitEvent(sg_st[0], ev_fork, 0));
            HIP_CHECK(hipStreamWaitEvent(sg_st[1], ev_fork, 0));
        }
        int gi = 0;
        for (int s0 = 0; s0 < slots; s0 += group, gi++) { // group gi: stream gi & 1, buffer gi & 1 (a buffer's next filter launch follows its projection launch in stream order)
            const int   gs = slots - s0 < group ? slots - s0 : group, b = gi & 1;
            const dim3  fgrid(((int)P.width + 63) / 64, ((int)P.height + 63) / 64, gs);
            int32_t*    qb = compact ? W.flt + lrs_dq_dwords((int)P.width, (int)P.height) + (size_t)b * group * wh : W.flt + (size_t)b * group * 2 * wh;
            hipStream_t s_ = sg_st[b];
            const bool  flt_now = (gi < 2) == (part == 0), proj_now = part == 1;
            if (compact) {
                if (flt_now) hipLaunchKernelGGL(HIP_KERNEL_NAME(lr_sgr_flt_kernel<true>), fgrid, dim3(256), LRS_SMEM, s_, P, W.flt, qb, s0, sgsum, slots);
                if (flt_now && gi == 0) HIP_CHECK(hipEventRecord(ev_dq, s_)); // (the first group's filter launch also writes the dgd - src plane every projection launch reads)
                if (proj_now && gi == 1) HIP_CHECK(hipStreamWaitEvent(s_, ev_dq, 0));
                if (proj_now) hipLaunchKernelGGL(HIP_KERNEL_NAME(lr_sgr_proj_kernel<true>), dim3(n, gs), dim3(PROJ_T), 0, s_, P, W.rects, W.flt, qb, W.sg, slots, s0, line_walk, sgsum, sg_dbg);
            } else {
                if (flt_now) hipLaunchKernelGGL(HIP_KERNEL_NAME(lr_sgr_flt_kernel<false>), fgrid, dim3(256), LRS_SMEM, s_, P, W.flt, qb, s0, sgsum, slots);
                if (proj_now) hipLaunchKernelGGL(HIP_KERNEL_NAME(lr_sgr_proj_kernel<false>), dim3(n, gs), dim3(PROJ_T), 0, s_, P, W.rects, W.flt, qb, W.sg, slots, s0, line_walk, sgsum, sg_dbg);
            }
        }
        if (part == 0) return;
        HIP_CHECK(hipEventRecord(ev_sg1, sg_st[1]));
        HIP_CHECK(hipStreamWaitEvent(sg_st[0], ev_sg1, 0));
        hipLaunchKernelGGL(lr_sgr_pick_kernel, dim3((n + 63) / 64), dim3(64), 0, sg_st[0], P, W.sg, units, slots, n);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(lr_trial_kernel<2>), tgrid, dim3(256), LRS_SMEM, sg_st[0], P, W.rects, W.wn, units, sg_acc, (SvtHipLrSearchUnit*)nullptr, (int32_t*)nullptr);
        hipLaunchKernelGGL(lr_take_sse_kernel, dim3((n + 63) / 64), dim3(64), 0, sg_st[0], sg_acc, units, 2, n);
        SVT_LAUNCH_CHECK();
        HIP_CHECK(hipEventRecord(ev_sg0, sg_st[0]));
    };
    if (sg_on) self_guided(0);
    // The Wiener chain is the longer one (statistics -> solve -> tens of dependent refinement rounds): its head is enqueued before the self-guided launches, the
    // refinement rounds after them (the host steps through those while everything else already runs).
    int rc_wn = 0;
    if (P.wn_enabled) {
        HIP_CHECK(hipStreamWaitEvent(wn_st, ev_fork, 0));
        restore_none(wn_st);
        svt_hip_lr_compute_stats_batch(P.dgd, P.src, W.rects, (uint32_t)n, mw, mh, (int)P.dgd_stride, (int)P.src_stride, P.wiener_win, P.highbd ? P.bit_depth : 8,
                                       (int64_t*)W.M, (int64_t*)W.H, wn_st);
        hipLaunchKernelGGL(lr_wiener_solve_kernel, dim3(n), dim3(256), 0, wn_st, P, W.M, W.H, prev, W.wn, units);
        hipLaunchKernelGGL(lr_wiener_start_kernel, dim3((n + 63) / 64), dim3(64), 0, wn_st, W.wn, W.counter, n);
        SVT_LAUNCH_CHECK();
    }
    if (sg_on) self_guided(1);
    if (P.wn_enabled) {
        // Lock-step refinement: a round = one trial launch whose last workgroup per unit consumes the result and proposes the next move.  The number of units still
        // searching comes back every WN_BATCH rounds through a page-locked word, and the NEXT batch is enqueued before the host waits for it: the read-back costs no
        // gap in the stream; the price is one batch of empty launches (every workgroup leaves at once) after the last unit has finished.
        constexpr int WN_BATCH = 4; // (an empty launch costs ~5 us; the fast settings need 8 rounds, the full one ~40)
        volatile int32_t* active = TS.pinned;
        hipEvent_t        ev_cnt = TS.ev[5];
        bool              done = false;
        int               rounds = 0;
        auto batch = [&]() {
            for (int k = 0; k < WN_BATCH; k++) hipLaunchKernelGGL(HIP_KERNEL_NAME(lr_trial_kernel<1>), tgrid, dim3(256), LRS_WN_SMEM, wn_st, P, W.rects, W.wn, units, W.acc, units, W.counter);
            rounds += WN_BATCH;
        };
        batch();
        while (!done && rounds < 4096) {
            HIP_CHECK(hipMemcpyAsync((void*)active, W.counter, 4, hipMemcpyDeviceToHost, wn_st));
            HIP_CHECK(hipEventRecord(ev_cnt, wn_st));
            batch();
            HIP_CHECK(hipEventSynchronize(ev_cnt));
            done = *active <= 0;
        }
        SVT_LAUNCH_CHECK();
        if (!done) { // the cap: was the last batch enough?
            HIP_CHECK(hipMemcpyAsync((void*)active, W.counter, 4, hipMemcpyDeviceToHost, wn_st));
            HIP_CHECK(hipStreamSynchronize(wn_st));
            if (*active > 0) rc_wn = -2; // units still searching: their sse[1] is not final -- the caller must not use this plane's result
        }
        HIP_CHECK(hipEventRecord(ev_wn, wn_st));
        HIP_CHECK(hipStreamWaitEvent(st, ev_wn, 0)); // (also on the failure path: the side streams' work must not outlive the call's ordering)
    }
    if (sg_dbg && sg_on) {
        int32_t c[32];
        HIP_CHECK(hipStreamSynchronize(sg_st[0]));
        HIP_CHECK(hipMemcpy(c, W.counter, 128, hipMemcpyDeviceToHost));
        fprintf(stderr, "SVT_HIP_LR_SG_STATS: %d units x %d sets: %d table passes (%d left their table), %d line passes\n", n, slots, c[1], c[3], c[2]);
        fprintf(stderr, "SVT_HIP_LR_SG_STATS: step-2 lines, moves accepted in the first pass, down 0..8,9+:");
        for (int k = 0; k < 10; k++) fprintf(stderr, " %d", c[4 + k]);
        fprintf(stderr, "; up (no down move) 0..8,9+:");
        for (int k = 0; k < 10; k++) fprintf(stderr, " %d", c[14 + k]);
        fprintf(stderr, "\n");
        fprintf(stderr, "SVT_HIP_LR_SG_STATS: %d projection workgroups, %.1f us each on average, of which thread 0 spent %.1f us inside the sample loops of its line passes\n", c[26],
                c[26] ? c[25] * 0.01 / c[26] : 0.0, c[26] ? c[24] * 0.01 / c[26] : 0.0);
    }
    if (sg_on) HIP_CHECK(hipStreamWaitEvent(st, ev_sg0, 0));
    return rc_wn;
    SVT_HIP_ENTRY_CATCH(SVT_HIP_E_DEVICE)
}

// Host-pointer form (what a seam in rest_process.c calls): uploads the plane with the 3 (+1 right) sample border the filters read and the source plane
// through the calling thread's pinned arena, runs the stage on that thread's stream, downloads the per-unit results.
int svt_hip_lr_search_plane_host(const SvtHipLrSearchParams* params, const SvtHipLrPrevUnit* prev, SvtHipLrSearchUnit* units) {
    SVT_HIP_ENTRY_TRY
    svthip::ensure_device();
    SvtHipLrSearchParams P = *params;
    const size_t px = P.highbd ? 2 : 1, w = P.width, h = P.height;
    const size_t dpitch = svthip::align_up((w + 8) * px, 16), spitch = svthip::align_up(w * px, 16);
    const int    n = n_units_1d((int)h, (int)P.unit_size) * n_units_1d((int)w, (int)P.unit_size);
    svthip::HostCallLease lease; // (a pooled arena: see svt_hip_common.h)
    svthip::HostCall& c = *lease;
    c.begin();
    P.dgd_stride = (uint32_t)(dpitch / px); P.src_stride = (uint32_t)(spitch / px);
    const size_t wsb = carve(P, nullptr, nullptr);
    const size_t dev = dpitch * (h + 6) + spitch * h + wsb + (size_t)n * (sizeof(SvtHipLrSearchUnit) + sizeof(SvtHipLrPrevUnit)) + 8192;
    c.reserve(dev, dpitch * (h + 6) + spitch * h + (size_t)n * (2 * sizeof(SvtHipLrSearchUnit) + sizeof(SvtHipLrPrevUnit)) + 8192);
    uint8_t* d_dgd = (uint8_t*)c.dalloc(dpitch * (h + 6));
    uint8_t* d_src = (uint8_t*)c.dalloc(spitch * h);
    void*    d_ws  = c.dalloc(wsb);
    SvtHipLrSearchUnit* d_units = (SvtHipLrSearchUnit*)c.dalloc((size_t)n * sizeof(SvtHipLrSearchUnit));
    SvtHipLrPrevUnit*   d_prev  = prev ? (SvtHipLrPrevUnit*)c.dalloc((size_t)n * sizeof(SvtHipLrPrevUnit)) : nullptr;
    c.up2d(d_dgd, dpitch, (const uint8_t*)params->dgd - ((size_t)3 * params->dgd_stride + 3) * px, (size_t)params->dgd_stride * px, (w + 7) * px, h + 6);
    c.up2d(d_src, spitch, params->src, (size_t)params->src_stride * px, w * px, h);
    if (prev) c.up(d_prev, prev, (size_t)n * sizeof(SvtHipLrPrevUnit));
    P.dgd = d_dgd + 3 * dpitch + 3 * px;
    P.src = d_src;
    const int rc = svt_hip_lr_search_plane(&P, d_prev, d_units, d_ws, c.stream);
    if (rc) return rc;
    c.down(units, d_units, (size_t)n * sizeof(SvtHipLrSearchUnit));
    return 0;
    SVT_HIP_ENTRY_CATCH(SVT_HIP_E_DEVICE)
}

} // extern "C"

SVT_HIP_DEFINE_WARM(lr_search) // (svt_hip_warmup loads this translation unit's code object at encoder initialisation: svt_hip_common.h)
