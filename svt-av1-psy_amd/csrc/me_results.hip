// me_results.hip -- ME result formatting for a whole picture: the tail of svt_aom_motion_estimation_b64
// (motion_estimation.c:3121-3152) that turns the integer-search tables p_sb_best_sad / p_sb_best_mv[list][ref][85] into what mode decision and
// rate control consume -- reference pruning on the ME SADs (me_prune_ref, :1522-1566), the per-PU candidate lists of MeSbResults
// (construct_me_candidate_array{,_mrp_off,_single_ref}, :2532-2828), the per-SB distortion statistics (compute_distortion, :2964-3008) and
// the global-motion detection flags (perform_gm_detection, :2833-2961).
//
// One 128-thread workgroup per 64x64 SB; thread n < 85 owns PU n of the search tables (n_idx order: 64x64 @0, 32x32 @1-4, 16x16 @5-20,
// 8x8 @21-84, z-order inside a level) and writes MeSbResults at the raster position of that PU.  The work is tiny (5.4 KB in, ~1 KB out per
// SB) and latency-bound; it exists so that the ME stage can hand over its final product without a round trip through the host.
#include "svt_hip_common.h"
#include "../../include/svtav1_hip.h"

namespace {

constexpr int NPU           = SVT_HIP_ME_NUM_BLOCKS;
constexpr int MAX_SAD_VALUE = 128 * 128 * 255; // motion_estimation.h:85

// z_to_raster (motion_estimation.c:2520-2531): de-interleave the z-order bits into (row, col) of the level
__device__ inline int z_to_raster(const int n) {
    if (n < 5) return n;
    if (n < 21) {
        const int k = n - 5;
        return 5 + (((k >> 3) & 1) * 2 + ((k >> 1) & 1)) * 4 + ((k >> 2) & 1) * 2 + (k & 1);
    }
    const int k = n - 21;
    return 21 + (((k >> 5) & 1) * 4 + ((k >> 3) & 1) * 2 + ((k >> 1) & 1)) * 8 + ((k >> 4) & 1) * 4 + ((k >> 2) & 1) * 2 + (k & 1);
}
// me_idx_85_8x8_to_16x16_conversion / me_idx_16x16_to_parent_32x32_conversion (definitions.h:2613-2632), raster child -> raster parent
__device__ inline int parent16(const int n) { const int k = n - 21; return 5 + (k >> 4) * 4 + ((k & 7) >> 1); }
__device__ inline int parent32(const int n) { const int k = n - 5; return 1 + (k >> 3) * 2 + ((k & 3) >> 1); }
// MeCandidate (me_sb_results.h:28-34): direction:2, ref_idx_l0:2, ref_idx_l1:2, ref0_list:1, ref1_list:1, LSB first
__device__ inline uint8_t cand(const int dir, const int l0, const int l1, const int r0, const int r1) {
    return (uint8_t)((dir & 3) | (l0 & 3) << 2 | (l1 & 3) << 4 | (r0 & 1) << 6 | (r1 & 1) << 7);
}
__device__ inline uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += (uint32_t)__shfl_xor((int)v, m);
    return v;
}
__device__ inline unsigned long long wave_sum64(unsigned long long v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)v, m), hi = (unsigned)__shfl_xor((int)(unsigned)(v >> 32), m);
        v += ((unsigned long long)hi << 32) | lo;
    }
    return v;
}

__global__ __launch_bounds__(128) void me_results_kernel(const SvtHipMeResultsParams P, const uint32_t* __restrict__ best_sad,
                                                         const uint32_t* __restrict__ best_mv, uint8_t* __restrict__ do_ref_g,
                                                         const uint8_t* __restrict__ sb_size, uint8_t* __restrict__ total_g, uint32_t* __restrict__ mv_g,
                                                         uint8_t* __restrict__ cand_g, SvtHipMeSbStats* __restrict__ stats) {
    __shared__ uint32_t sh_sum[8];
    __shared__ uint8_t  sh_do[8];
    __shared__ uint32_t sh_dist[NPU];
    __shared__ uint8_t  sh_first[NPU];
    __shared__ uint32_t sh_cnt[32];
    const int    sb = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int    nr0 = P.num_of_ref_pic_to_search[0], nr1 = P.num_of_ref_pic_to_search[1];
    const size_t ref_pitch = (size_t)P.n_sb * NPU;
    const uint32_t* sad_sb = best_sad + (size_t)sb * NPU;
    const uint32_t* mv_sb  = best_mv + (size_t)sb * NPU;
    const int    n_pus = P.enable_me_16x16 ? (P.enable_me_8x8 ? 85 : 21) : 5;
    uint8_t*     total = total_g + (size_t)sb * n_pus;
    uint32_t*    mvs   = mv_g + (size_t)sb * n_pus * P.max_refs;
    uint8_t*     cands = cand_g + (size_t)sb * n_pus * P.max_cand;
#define SLOT(l, r) ((size_t)(((l) ? nr0 : 0) + (r)) * ref_pitch)
#define SAD(l, r, n) sad_sb[SLOT(l, r) + (n)]
#define MV(l, r, n) mv_sb[SLOT(l, r) + (n)]
#define NR(l) ((l) ? nr1 : nr0)

    if (t < 8) sh_do[t] = do_ref_g[(size_t)sb * 8 + t];
    if (t < 32) sh_cnt[t] = 0;
    __syncthreads();

    // ---- me_prune_ref: wave l sums the 64 8x8 SADs of each reference of list l
    if (P.prune_ref) {
        if (wv < P.num_of_list_to_search)
            for (int r = 0; r < NR(wv); r++) {
                const uint32_t s = sh_do[wv * 4 + r] ? wave_sum(SAD(wv, r, 21 + lane)) : 0;
                if (lane == 0) sh_sum[wv * 4 + r] = s;
            }
        __syncthreads();
        if (t == 0 && P.prune_ref_if_me_sad_dev_bigger_than_th != 0xffff) {
            unsigned long long sum[8], best = ~0ull;
            for (int i = 0; i < 8; i++) {
                const int l = i >> 2, r = i & 3;
                const bool searched = l < P.num_of_list_to_search && r < NR(l);
                sum[i] = !searched ? 0xffffffffull : sh_do[i] ? sh_sum[i] : (unsigned long long)MAX_SAD_VALUE * 64; // hme_sad keeps MAX_U32 where unused (:3061)
                best   = sum[i] < best ? sum[i] : best;
            }
            for (int i = 0; i < 8; i++)
                if ((i & 3) && (sum[i] - best) * 100 > (unsigned long long)P.prune_ref_if_me_sad_dev_bigger_than_th * best) sh_do[i] = 0;
        }
        __syncthreads();
        if (t < 8) do_ref_g[(size_t)sb * 8 + t] = sh_do[t];
    }

    // ---- candidate lists: thread n = PU n of the search tables
    if (t < NPU) {
        const int      n = t;
        const bool     use = P.enable_me_16x16 ? (P.enable_me_8x8 || n < 21) : n < 5;
        const int      pu = z_to_raster(n);
        const uint32_t cand_th = (uint32_t)P.prune_me_candidates_th;
        uint8_t*       pc = cands + pu * P.max_cand;
        uint32_t*      pm = mvs + pu * P.max_refs;
        uint32_t       dist;
        int            first = -1; // value written to the PU's first candidate slot, if any
        if (nr0 == 1 && nr1 == 0) { // construct_me_candidate_array_single_ref
            dist = SAD(0, 0, n);
            if (use) total[pu] = 1;
            if (sh_do[0] && use) {
                pc[0] = cand(0, 0, 0, 0, 0);
                first = pc[0];
                pm[0] = MV(0, 0, n);
            }
        } else if (nr0 == 1 && nr1 == 1) { // construct_me_candidate_array_mrp_off
            uint32_t      nlist = P.num_of_list_to_search;
            const uint8_t org0 = sh_do[0], org1 = nlist == 1 ? 0 : sh_do[4];
            if (nlist < 2 || !sh_do[4]) nlist = 1;
            const uint32_t th = (org0 && org1) ? cand_th : 0;
            const uint32_t s0 = SAD(0, 0, n), s1 = SAD(1, 0, n);
            const uint32_t best = (org0 && org1) ? (s0 < s1 ? s0 : s1) : org0 ? s0 : s1;
            dist = best;
            int blk0 = org0, blk1 = org1, off = 0, tot = 1;
            const int min_list = (P.use_best_unipred_cand_only && blk0 && blk1) ? (s0 < s1 ? 0 : 1) : -1;
            for (uint32_t l = 0; l < nlist && (use || off == 0); l++) {
                const uint32_t s = l ? s1 : s0;
                if (!(l ? blk1 : blk0)) continue;
                if (th > 0 && (uint32_t)((s - best) * 100) > (uint32_t)(best * th)) {
                    if (l) blk1 = 0; else blk0 = 0;
                    continue;
                }
                const uint32_t mv = MV(l, 0, n);
                if (min_list != -1 && min_list != (int)l) { // only the MV is kept, for the bi-pred candidate
                    if (use) pm[l ? P.max_l0 : 0] = mv;
                    continue;
                }
                if (use) {
                    pc[off] = cand((int)l, 0, 0, 0, l == 1);
                    if (off == 0) first = pc[0];
                    pm[l ? P.max_l0 : 0] = mv;
                }
                off++;
            }
            if (blk0 && blk1 && use) {
                pc[off] = cand(2, 0, 0, 0, 1);
                if (off == 0) first = pc[0];
                tot = off + 1;
            }
            if (use) total[pu] = (uint8_t)tot;
        } else { // construct_me_candidate_array
            const uint32_t nlist = P.num_of_list_to_search;
            uint32_t       blk = 0, best = ~0u;
            int            off = 0;
            for (uint32_t l = 0; l < nlist; l++)
                for (int r = 0; r < NR(l); r++)
                    if (sh_do[l * 4 + r]) {
                        blk |= 1u << (l * 4 + r);
                        const uint32_t s = SAD(l, r, n);
                        best = s < best ? s : best;
                    }
            dist = best;
            for (uint32_t l = 0; l < nlist && (use || off == 0); l++)
                for (int r = 0; r < NR(l) && (use || off == 0); r++) {
                    const uint32_t bit = 1u << (l * 4 + r);
                    if (!(blk & bit)) continue;
                    if (cand_th > 0 && (uint32_t)((SAD(l, r, n) - best) * 100) > (uint32_t)(best * cand_th)) {
                        blk &= ~bit;
                        continue;
                    }
                    if (use) {
                        pc[off] = cand((int)l, r, r, 0, l == 1);
                        if (off == 0) first = pc[0];
                        pm[(l ? P.max_l0 : 0) + r] = MV(l, r, n);
                    }
                    off++;
                }
            if (nlist == 2 && use) {
                for (int a = 0; a < nr0; a++) // (L0[a], L1[b])
                    for (int b = 0; b < nr1; b++) {
                        if (P.only_l_bwd && (a > 0 || b > 0)) continue;
                        if ((blk >> a & 1) && (blk >> (4 + b) & 1)) {
                            pc[off] = cand(2, a, b, 0, 1);
                            if (off == 0) first = pc[0];
                            off++;
                        }
                    }
                if (!P.only_l_bwd) {
                    for (int a = 1; a < nr0; a++) // (LAST, L0[a])
                        if ((blk & 1) && (blk >> a & 1)) {
                            pc[off] = cand(2, 0, a, 0, 0);
                            if (off == 0) first = pc[0];
                            off++;
                        }
                    if (nr1 == 3 && (blk >> 4 & 1) && (blk >> 6 & 1)) { // (BWD, ALT)
                        pc[off] = cand(2, 0, 2, 1, 1);
                        if (off == 0) first = pc[0];
                        off++;
                    }
                }
            }
            if (use) total[pu] = (uint8_t)off;
        }
        sh_dist[pu] = dist;
        if (use) sh_first[pu] = first >= 0 ? (uint8_t)first : pc[0]; // GM detection reads the slot even when nothing was written to it
    }
    __syncthreads();

    // ---- compute_distortion + perform_gm_detection: wave 0
    if (wv == 0) {
        const uint32_t d    = sh_dist[21 + lane];
        const uint32_t d8   = wave_sum(d);
        const uint32_t d16  = wave_sum(lane < 16 ? sh_dist[5 + lane] : 0);
        const uint32_t d32  = wave_sum(lane < 4 ? sh_dist[1 + lane] : 0);
        const long long diff = (long long)d - (long long)(d8 / 64);
        const unsigned long long ssq = wave_sum64((unsigned long long)(diff * diff));
        uint32_t stationary = 0;
        const int nblk = P.low_resolution ? 64 : 16;
        if (P.gm_enabled) {
            bool still = false;
            if (lane < nblk) {
                int n = (P.low_resolution ? 21 : 5) + lane;
                if (P.low_resolution) {
                    if (!P.enable_me_8x8) {
                        n = parent16(n);
                        if (!P.enable_me_16x16) n = parent32(n);
                    }
                } else if (!P.enable_me_16x16)
                    n = parent32(n);
                const uint8_t c = sh_first[n];
                const int     dir = c & 3;
                const bool    fwd = dir == 0 || dir == 2;
                const int     l = fwd ? (c >> 6) & 1 : (c >> 7) & 1, r = fwd ? (c >> 2) & 3 : (c >> 4) & 3;
                const unsigned long long a = P.picture_number, b = P.ref_picture_number[l][r];
                int dd, active_th;
                if (P.low_resolution) {
                    dd = (int16_t)((a > b ? a : b) - (a > b ? b : a));
                    dd = (uint16_t)(dd < 0 ? -dd : dd);
                    active_th = P.gm_use_distance_based_active_th ? ((dd >> 1) > 4 ? (dd >> 1) : 4) : 4;
                } else {
                    dd = (int16_t)(a - b);
                    dd = (uint16_t)(dd < 0 ? -dd : dd);
                    active_th = P.gm_use_distance_based_active_th ? (dd * 16 > 32 ? dd * 16 : 32) : 32;
                }
                const bool     searched = l < P.num_of_list_to_search && r < NR(l);
                const uint32_t mv = searched ? MV(l, r, n) : 0; // untouched p_sb_best_mv slots are zero (:3049-3052)
                const int      mx = (int)(int16_t)(mv & 0xffff) * 4, my = (int)(int16_t)(mv >> 16) * 4;
                uint32_t* cnt = &sh_cnt[(l * 4 + r) * 4];
                if (mx < -active_th) atomicAdd(&cnt[0], 1u); else if (mx > active_th) atomicAdd(&cnt[1], 1u);
                if (my < -active_th) atomicAdd(&cnt[2], 1u); else if (my > active_th) atomicAdd(&cnt[3], 1u);
                const int sth = P.low_resolution ? 0 : 4;
                still = (mx < 0 ? -mx : mx) <= sth && (my < 0 ? -my : my) <= sth;
            }
            stationary = (uint32_t)__popcll(__ballot(still));
        }
        uint32_t over = lane < 32 ? (sh_cnt[lane] > (uint32_t)(nblk / 2)) : 0; // same-wave LDS atomics above are complete (in-order LDS queue)
        over = __ballot(over) != 0;
        if (lane == 0) {
            const uint32_t pix = (uint32_t)sb_size[2 * sb] * sb_size[2 * sb + 1];
            SvtHipMeSbStats s;
            s.me_8x8_cost_variance = (uint32_t)(ssq / 64);
            s.rc_me_distortion     = P.low_resolution ? d8 : d16;
            s.me_64x64_distortion  = (uint32_t)(sh_dist[0] * 4096u) / pix;
            s.me_32x32_distortion  = (uint32_t)(d32 * 4096u) / pix;
            s.me_16x16_distortion  = (uint32_t)(d16 * 4096u) / pix;
            s.me_8x8_distortion    = (uint32_t)(d8 * 4096u) / pix;
            s.stationary_block_present_sb = P.gm_enabled && stationary > (uint32_t)(nblk * 5) / 100;
            s.rc_me_allow_gm              = P.gm_enabled && over;
            s.pad[0] = s.pad[1] = 0;
            stats[sb] = s;
        }
    }
#undef SLOT
#undef NR
#undef SAD
#undef MV
}

} // namespace

extern "C" void svt_hip_me_results_batch(const SvtHipMeResultsParams* params, const uint32_t* best_sad, const uint32_t* best_mv, uint8_t* do_ref,
                                         const uint8_t* sb_size, uint8_t* total_me_candidate_index, uint32_t* me_mv_array, uint8_t* me_candidate_array,
                                         SvtHipMeSbStats* sb_stats, void* stream) {
    svthip::ensure_device();
    if (params->n_sb == 0) return;
    hipLaunchKernelGGL(me_results_kernel, dim3(params->n_sb), dim3(128), 0, (hipStream_t)stream, *params, best_sad, best_mv, do_ref, sb_size,
                       total_me_candidate_index, me_mv_array, me_candidate_array, sb_stats);
    SVT_LAUNCH_CHECK();
}

SVT_HIP_DEFINE_WARM(me_results) // (svt_hip_warmup loads this translation unit's code object at encoder initialisation: svt_hip_common.h)
