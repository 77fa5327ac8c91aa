// tpl.hip -- the SOURCE-BASED half of the TPL dispenser as one device stage per picture.
//
// Reference: tpl_mc_flow_dispenser_sb_generic, Codec/src_ops_process.c:519-969 (tpl levels 4 / 5 of initial_rc_process.c:343-378: SAD costs, DC intra only,
// full-pel vectors, no rate).  Per 16x16 / 32x32 block: DC intra cost from source neighbours, SAD of every uni-directional ME candidate at its clamped vector,
// the winner, and for NEWMV the forward transform (partial-frequency shape, subsampled rows) + svt_av1_quantize_fp reconstruction error of the residual.
// Blocks only read source pictures, the ME tables and the quantizer row: they are independent, so a picture is ONE launch (two at level 1: 32x32 blocks of the
// complete SBs, 16x16 blocks of the SBs the picture edge cuts).
//
// Mapping: SIZE lanes per block (lane t owns row t of the block, then column t / row t of the transform), 256 / SIZE blocks per workgroup.  A row is one or two
// unaligned 16-byte loads kept in registers for the whole block; costs are v_sad_u8 on the raw dwords reduced over the block's lanes with shuffles (the lanes of a
// block are contiguous inside one wave); the best candidate's reference row stays in registers, so the residual needs no second fetch.  The residual goes through
// the transform tile in LDS exactly like fwd_txfm2d_kernel (txfm.hip) -- same 1-D flow graphs (txfm_core.h), same shifts -- and the kept corner is quantised and
// compared in registers (quant_core.h is not needed: quantize_fp at log_scale 0 without matrices is three lines).  HBM traffic = the picture once + one block
// row set per candidate; at 1080p (8 160 blocks x ~5 candidates) the launch is latency-sized -- the point of the stage is to take the work off the host.
#include "svt_hip_common.h"
#include "../../include/svtav1_hip.h"
#include "txfm_core.h"
#include <type_traits>

namespace {

struct __attribute__((packed, aligned(1))) tpl_u32x4_a1 { uint32_t x, y, z, w; }; // 16 bytes at any byte address
// The write-through hand-off of SVT_HIP_TPL_RECON_FORM=7 (svt_hip_common.h: svt_hip_store_x4_wt / svt_hip_drain_stores)
__device__ __forceinline__ void tpl_store_x4_wt(uint8_t* p, const tpl_u32x4_a1 v) { svt_hip_store_x4_wt(p, v.x, v.y, v.z, v.w); }

constexpr int TPL_PAD = 32;          // TPL_PADX / TPL_PADY (encode_context.h:43-44)
constexpr int TPL_NEWMV = 16;        // PredictionMode NEWMV (definitions.h:1143); DC_PRED = 0
constexpr int TPL_COST_SCALE_LOG2 = 4; // TPL_DEP_COST_SCALE_LOG2 (definitions.h:49)

template <int T> __device__ __forceinline__ uint32_t group_sum(uint32_t v) { // sum over the T contiguous lanes of a block; every lane gets the total
#pragma unroll
    for (int m = T >> 1; m >= 1; m >>= 1) v += (uint32_t)__shfl_xor((int)v, m);
    return v;
}
template <int NW> __device__ __forceinline__ void load_row(uint32_t (&r)[NW], const uint8_t* p) {
#pragma unroll
    for (int i = 0; i < NW / 4; i++) {
        const tpl_u32x4_a1 v = *(const tpl_u32x4_a1*)(p + 16 * i);
        r[4 * i] = v.x; r[4 * i + 1] = v.y; r[4 * i + 2] = v.z; r[4 * i + 3] = v.w;
    }
}
template <int NW> __device__ __forceinline__ uint32_t row_sad(const uint32_t (&a)[NW], const uint32_t (&b)[NW]) {
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < NW; i++) s = __builtin_amdgcn_sad_u8(a[i], b[i], s);
    return s;
}

// SIZE: block edge (16 / 32); TXH: rows the transform sees (SIZE >> subsample_tx); ONLY_INCOMPLETE: 0 = every SB (level 0), 1 = SBs the picture edge cuts
// (level 1, second launch); for SIZE 32 only complete SBs are processed.
template <int SIZE, int TXH>
__global__ __launch_bounds__(256) void tpl_src_kernel(const SvtHipTplSrcParams P, const uint8_t* __restrict__ src_base, const uint8_t* __restrict__ ref_base,
                                                      const uint8_t* __restrict__ tot_base, const uint32_t* __restrict__ mv_base, const uint8_t* __restrict__ cand_base,
                                                      SvtHipTplSrcStats* __restrict__ stats, const int only_incomplete) {
    constexpr int T = SIZE, BPW = 256 / T, W = SIZE, PITCH = W + 1, NW = SIZE / 4, PER = 64 / SIZE, ST = SIZE == 2 * TXH ? 1 : (SIZE == 4 * TXH ? 2 : 0);
    constexpr int FS0 = fwd_shift0(W, TXH), FS1 = -fwd_shift1(W, TXH), FS2 = -fwd_shift2(W, TXH);
    constexpr int CBC = kFwdCosCol[ilog2c(W) - 2][ilog2c(TXH) - 2], CBR = kFwdCosRow[ilog2c(W) - 2][ilog2c(TXH) - 2];
    constexpr bool RECT1 = (W == 2 * TXH) || (TXH == 2 * W);
    HIP_DYNAMIC_SHARED(int32_t, smem)
    __shared__ SvtHipTplRef s_refs[8]; // indexed per lane below: a run-time index into the kernel-argument struct would put the table in scratch
    if (threadIdx.x < 8) s_refs[threadIdx.x] = P.refs[threadIdx.x];
    __syncthreads();
    const int      tid = threadIdx.x, sub = tid / T, t = tid % T;
    const uint32_t item = blockIdx.x * BPW + sub, sb = item / (PER * PER), k = item % (PER * PER);
    int32_t*       buf = smem + sub * (TXH * PITCH);
    const int      aligned_h = (int)((P.height + 7) & ~7u);
    bool           active = sb < P.n_sb;
    const int      sx = (int)(sb % P.sbs_x) * 64, sy = (int)(sb / P.sbs_x) * 64;
    const bool     complete = ((int)P.aligned_width - sx >= 64) && (aligned_h - sy >= 64);
    if (SIZE == 32) active = active && complete;
    else if (only_incomplete) active = active && !complete;
    const int bx = (int)k % PER, by = (int)k / PER, x0 = sx + bx * SIZE, y0 = sy + by * SIZE;
    active = active && !(x0 + (SIZE >> 1) > (int)P.width || y0 + (SIZE >> 1) > (int)P.height); // at least half of the block inside (:580)
    const int n_pus = P.enable_me_8x8 ? 85 : (P.enable_me_16x16 ? 21 : 5);
    int       pu    = SIZE == 32 ? 1 + by * 2 + bx : 5 + by * 4 + bx; // tpl_blk_idx_tab[1] (:355): z-order block -> raster PU of the ME tables
    if (!P.enable_me_16x16) pu = (pu - 1) / 4;                          // :762-763
    const uint32_t ss  = P.src_stride;
    const uint8_t* src = src_base + P.src_off;
    uint32_t       srow[NW], brow[NW];
#pragma unroll
    for (int i = 0; i < NW; i++) { srow[i] = 0; brow[i] = 0; }
    // ---- intra: DC from the source's neighbours (:620-657; border forms of svt_aom_update_neighbor_samples_array_open_loop_mb: fill values 127 / 129) ----
    uint32_t best_intra = 0xffffffffu; // (INT64_MAX in the reference; every SAD is < 2^24)
    if (active) load_row<NW>(srow, src + (size_t)(y0 + t) * ss + x0);
    {
        uint32_t a = 0, l = 0;
        if (active && !P.disable_intra_pred) {
            if (y0 > 0) a = (x0 + t < (int)P.width) ? src[(size_t)(y0 - 1) * ss + x0 + t] : 127u;
            if (x0 > 0) l = (y0 + t < (int)P.height) ? src[(size_t)(y0 + t) * ss + x0 - 1] : 129u;
        }
        const uint32_t sa = group_sum<T>(a), sl = group_sum<T>(l);
        uint32_t dc;
        if (x0 > 0 && y0 > 0) dc = (sa + sl + SIZE) / (2 * SIZE);
        else if (x0 > 0) dc = (sl + (SIZE >> 1)) / SIZE;
        else if (y0 > 0) dc = (sa + (SIZE >> 1)) / SIZE;
        else dc = 128;
        uint32_t dcrow[NW];
#pragma unroll
        for (int i = 0; i < NW; i++) dcrow[i] = dc * 0x01010101u;
        const uint32_t s = group_sum<T>(row_sad<NW>(srow, dcrow));
        if (active && !P.disable_intra_pred) best_intra = s;
    }
    // ---- inter: the PU's uni-directional candidates in table order, first strict minimum (:761-890) ----
    uint32_t best_inter = 0xffffffffu;
    int      best_rf = -1, mvr = 0, mvc = 0;
    const int n_cand = (active && !P.i_slice) ? (int)tot_base[(size_t)sb * n_pus + pu] : 0;
    for (int i = 0; i < (int)P.max_cand; i++) { // uniform trip count: the shuffles below are executed by every lane of the workgroup
        bool     ok = i < n_cand;
        uint32_t rrow[NW];
#pragma unroll
        for (int j = 0; j < NW; j++) rrow[j] = 0;
        int rf = 0, xm = 0, ym = 0;
        if (ok) {
            const uint32_t c = cand_base[((size_t)sb * n_pus + pu) * P.max_cand + i];
            const int dir = (int)(c & 3), r0 = (int)((c >> 2) & 3), r1 = (int)((c >> 4) & 3);
            const int list = dir & 1, ref = list == 0 ? r0 : r1;
            rf = list * 4 + ref;
            ok = dir <= 1 && s_refs[rf].valid;
            if (ok) {
                const SvtHipTplRef& R = s_refs[rf];
                const uint32_t m = mv_base[((size_t)sb * n_pus + pu) * P.max_refs + (list ? P.max_l0 : 0) + ref];
                xm = (int)(int16_t)((int16_t)(m & 0xffff) * 8);
                ym = (int)(int16_t)((int16_t)(m >> 16) * 8);
                if (x0 + (xm >> 3) < -TPL_PAD) xm = (int)(int16_t)((-TPL_PAD - x0) * 8);
                if (x0 + SIZE + (xm >> 3) > TPL_PAD + (int)R.max_width - 1) xm = (int)(int16_t)(((TPL_PAD + (int)R.max_width - 1) - (x0 + SIZE)) * 8);
                if (y0 + (ym >> 3) < -TPL_PAD) ym = (int)(int16_t)((-TPL_PAD - y0) * 8);
                if (y0 + SIZE + (ym >> 3) > TPL_PAD + (int)R.max_height - 1) ym = (int)(int16_t)(((TPL_PAD + (int)R.max_height - 1) - (y0 + SIZE)) * 8);
                load_row<NW>(rrow, ref_base + R.plane_off + (size_t)((int)R.org_y + y0 + t + ym / 8) * R.stride + (int)R.org_x + x0 + xm / 8);
            }
        }
        const uint32_t cost = group_sum<T>(row_sad<NW>(srow, rrow));
        if (ok && cost < best_inter) {
            best_inter = cost; best_rf = rf; mvr = ym; mvc = xm;
#pragma unroll
            for (int j = 0; j < NW; j++) brow[j] = rrow[j];
        }
    }
    const bool newmv = active && best_inter < best_intra; // :892 (both INT64_MAX in the reference when nothing was evaluated: not less)
    // ---- NEWMV: residual rows -> forward DCT_DCT (svt_av1_wht_fwd_txfm = svt_av1_highbd_fwd_txfm[_n2/_n4], transforms.c:3640) -> quantize_fp error ----
    if (newmv && (t & ((1 << ST) - 1)) == 0) {
        const int r = t >> ST;
#pragma unroll
        for (int j = 0; j < NW; j++)
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int d = (int)((srow[j] >> (8 * b)) & 255u) - (int)((brow[j] >> (8 * b)) & 255u);
                buf[r * PITCH + 4 * j + b] = (int32_t)((uint32_t)d << FS0);
            }
    }
    __syncthreads();
    if (newmv) { // column t (T == W)
        int32_t v[TXH];
#pragma unroll
        for (int r = 0; r < TXH; r++) v[r] = buf[r * PITCH + t];
        fwd1d<TXH, CBC>(K_DCT, v);
#pragma unroll
        for (int r = 0; r < TXH; r++) buf[r * PITCH + t] = FS1 ? rshift_round(v[r], FS1 ? FS1 : 1) : v[r];
    }
    __syncthreads();
    uint32_t err_lo = 0, err_hi = 0;
    if (newmv && t < TXH) {
        int32_t v[W];
#pragma unroll
        for (int c = 0; c < W; c++) v[c] = buf[t * PITCH + c];
        fwd1d<W, CBR>(K_DCT, v);
        const int kw = W >> P.pf_shape, kh = TXH >> P.pf_shape; // the partial-frequency shapes keep the top-left corner, the rest is 0 (transforms.c:5202-5273)
        unsigned long long err = 0;
        if (t < kh) {
#pragma unroll
            for (int c = 0; c < W; c++) {
                if (c < kw) {
                    int32_t x = FS2 ? rshift_round(v[c], FS2 ? FS2 : 1) : v[c];
                    if (RECT1) x = mul_sqrt2_like(x, 5793);
                    // svt_av1_quantize_fp (quantize_fp_helper_c, full_loop.c:282-342; log_scale 0, no matrices) and svt_av1_block_error's term
                    const int     kk = (t | c) != 0;
                    const int32_t sign = x < 0 ? -1 : 0, a = (x ^ sign) - sign;
                    int32_t       dq = 0;
                    if (((long long)a << 1) >= (int32_t)P.dequant[kk]) {
                        long long tt = (long long)a + P.round_fp[kk];
                        tt = tt < -32768 ? -32768 : (tt > 32767 ? 32767 : tt);
                        const int32_t q = (int32_t)((tt * P.quant_fp[kk]) >> 16);
                        if (q) dq = (((int32_t)((uint32_t)q * (uint32_t)(int32_t)P.dequant[kk])) ^ sign) - sign;
                    }
                    const long long df = (long long)x - dq;
                    err += (unsigned long long)(df * df);
                }
            }
        }
        err_lo = (uint32_t)err; err_hi = (uint32_t)(err >> 32);
    }
    // 64-bit sum over the block's lanes (carry through a 3 x 22-bit split is not needed: add the halves separately and propagate the carry count)
    {
        // split into three 22-bit limbs so that T <= 32 partial sums cannot overflow 32 bits
        const unsigned long long e = ((unsigned long long)err_hi << 32) | err_lo;
        const uint32_t l0 = group_sum<T>((uint32_t)(e & 0x3fffffu)), l1 = group_sum<T>((uint32_t)((e >> 22) & 0x3fffffu)), l2 = group_sum<T>((uint32_t)(e >> 44));
        const unsigned long long tot = (unsigned long long)l0 + ((unsigned long long)l1 << 22) + ((unsigned long long)l2 << 44);
        err_lo = (uint32_t)tot; err_hi = (uint32_t)(tot >> 32);
    }
    if (active && t == 0) {
        SvtHipTplSrcStats o = {};
        o.written = 1;
        o.best_mode = newmv ? TPL_NEWMV : 0;
        o.best_intra_mode = 0;
        o.best_rf_idx = best_rf;
        o.ref_frame_poc = best_rf >= 0 ? s_refs[best_rf].picture_number : 0;
        o.mv_row = (int16_t)mvr; o.mv_col = (int16_t)mvc;
        if (newmv) {
            long long e = (long long)(((unsigned long long)err_hi << 32) | err_lo);
            e >>= (SIZE == 32 && TXH == 32) ? 0 : 2; // get_quantize_error: shift = tx_size == TX_32X32 ? 0 : 2 (:227)
            if (e < 1) e = 1;
            o.srcrf_dist = (e << TPL_COST_SCALE_LOG2) << ST;
        }
        stats[(size_t)(y0 >> 4) * ((P.aligned_width + 15) >> 4) + (x0 >> 4)] = o;
    }
}

template <int SIZE, int TXH>
void launch_tpl(const SvtHipTplSrcParams& P, const uint8_t* src, const uint8_t* ref, const uint8_t* tot, const uint32_t* mv, const uint8_t* cand,
                SvtHipTplSrcStats* stats, int only_incomplete, hipStream_t st) {
    constexpr int BPW = 256 / SIZE, PER = 64 / SIZE;
    const uint32_t items = P.n_sb * PER * PER;
    const size_t   shmem = (size_t)BPW * TXH * (SIZE + 1) * 4;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(tpl_src_kernel<SIZE, TXH>), dim3((items + BPW - 1) / BPW), dim3(256), shmem, st, P, src, ref, tot, mv, cand, stats, only_incomplete);
    SVT_LAUNCH_CHECK();
}

// ---- the reconstruction half -------------------------------------------------------------------------------------------------------------------------------------
// One launch per anti-diagonal d of the 16x16 cell grid: the blocks whose top-left cell (cx, cy) has cx + cy == d.  A block reads the reconstruction of its left and
// upper neighbours only (DC prediction), and those start on an earlier diagonal -- for the mixed grid too: SBs the picture edge cuts (16x16 blocks) lie right of /
// below the complete ones (32x32 blocks), so a 32x32 block never has 16x16 neighbours on its left or above.  Same lane mapping as the source-based kernel; the
// prediction tile lives in LDS (the column pass of the inverse adds it transposed), the quantised corner goes back into the transform tile for the inverse
// (inv_txfm2d_kernel's two passes, txfm.hip).
// A block at cell (cx, cy) in two steps.  tpl_recon_fetch: everything that does not depend on the neighbours -- the block's statistics, its source row and (NEWMV)
// the reference row of lane t.  tpl_recon_compute: the rest; every thread of the workgroup calls it (it contains workgroup barriers).  t = the lane's row / column
// inside the block, live = the lane group has a block to do; buf: the block's transform tile [TXH][SIZE + 1] int32, ptile: its prediction / reconstruction tile
// [SIZE][SIZE + 4] bytes.
template <int SIZE> struct TplBlockIn {
    SvtHipTplSrcStats s;
    uint32_t          srow[SIZE / 4], prow[SIZE / 4];
    bool              active, newmv;
};
template <int SIZE>
__device__ __forceinline__ void tpl_recon_fetch(const SvtHipTplReconParams& RP, const SvtHipTplRef* s_refs, const uint8_t* __restrict__ src_base,
                                                const uint8_t* __restrict__ ref_base, const SvtHipTplSrcStats* __restrict__ src_stats, const int cx, const int cy,
                                                const bool live, const int t, TplBlockIn<SIZE>& B) {
    const SvtHipTplSrcParams& P = RP.src;
    constexpr int NW = SIZE / 4;
    const int x0 = cx * 16, y0 = cy * 16;
    const int aligned_h = (int)((P.height + 7) & ~7u);
    bool      active = live && cx >= 0;
    const int sx = x0 & ~63, sy = y0 & ~63;
    const bool complete = ((int)P.aligned_width - sx >= 64) && (aligned_h - sy >= 64);
    const int  bsize = (complete && P.dispenser_search_level) ? 32 : 16;
    active = active && bsize == SIZE && (SIZE == 16 || (((cx | cy) & 1) == 0));
    active = active && x0 < (int)P.aligned_width && y0 < aligned_h && !(x0 + (SIZE >> 1) > (int)P.width || y0 + (SIZE >> 1) > (int)P.height);
    const size_t cell = (size_t)cy * ((P.aligned_width + 15) >> 4) + (size_t)(cx < 0 ? 0 : cx);
    B.s = SvtHipTplSrcStats{};
    if (active) B.s = src_stats[cell];
    active = active && B.s.written;
    B.active = active;
    B.newmv  = active && B.s.best_mode == TPL_NEWMV;
#pragma unroll
    for (int i = 0; i < NW; i++) { B.srow[i] = 0; B.prow[i] = 0; }
    if (active) load_row<NW>(B.srow, src_base + P.src_off + (size_t)(y0 + t) * P.src_stride + x0);
    if (B.newmv) {
        const SvtHipTplRef& R = s_refs[B.s.best_rf_idx & 7];
        load_row<NW>(B.prow, ref_base + R.plane_off + (size_t)((int)R.org_y + y0 + t + (B.s.mv_row >> 3)) * R.stride + (int)R.org_x + x0 + (B.s.mv_col >> 3));
    }
}
template <int SIZE, int TXH, bool WT = false, bool WSYNC = false> // WSYNC: the caller's workgroup holds several independent single-wave blocks -- every barrier below is
// between lanes of ONE wave anyway (a block's lanes never span two waves), so it becomes a wave barrier (LDS traffic of a wave is served in order).  WT: the reconstruction is handed to the neighbouring blocks' waves of the SAME launch with write-through stores / coherent loads
__device__ __forceinline__ void tpl_recon_compute(const SvtHipTplReconParams& RP, uint8_t* __restrict__ recon_base, SvtHipTplReconStats* __restrict__ out, const int cx,
                                                  const int cy, const TplBlockIn<SIZE>& B, const int t, int32_t* buf, uint8_t* ptile) {
    const SvtHipTplSrcParams& P = RP.src;
    constexpr int T = SIZE, W = SIZE, PITCH = W + 1, NW = SIZE / 4, ST = SIZE == 2 * TXH ? 1 : (SIZE == 4 * TXH ? 2 : 0), PP = SIZE + 4;
    constexpr int FS0 = fwd_shift0(W, TXH), FS1 = -fwd_shift1(W, TXH), FS2 = -fwd_shift2(W, TXH);
    constexpr int CBC = kFwdCosCol[ilog2c(W) - 2][ilog2c(TXH) - 2], CBR = kFwdCosRow[ilog2c(W) - 2][ilog2c(TXH) - 2];
    constexpr bool RECT1 = (W == 2 * TXH) || (TXH == 2 * W);
    constexpr int  S0 = -inv_shift0(W, TXH);
    const int x0 = cx * 16, y0 = cy * 16;
    const bool active = B.active, newmv = B.newmv;
    const SvtHipTplSrcStats& s = B.s;
    const size_t cell = (size_t)cy * ((P.aligned_width + 15) >> 4) + (size_t)(cx < 0 ? 0 : cx);
    const uint32_t rs = RP.recon_stride;
    uint8_t*       rec = recon_base + RP.recon_off;
    uint32_t       srow[NW], prow[NW];
#pragma unroll
    for (int i = 0; i < NW; i++) { srow[i] = B.srow[i]; prow[i] = B.prow[i]; }
    { // DC from the reconstructed neighbours (fill values 127 / 129 beyond the picture, forms by availability: as the source-based kernel, on the reconstruction)
        uint32_t a = 0, l = 0;
        if (active && !newmv) {
            if (y0 > 0) a = (x0 + t < (int)P.width) ? rec[(size_t)(y0 - 1) * rs + x0 + t] : 127u;
            if (x0 > 0) l = (y0 + t < (int)P.height) ? rec[(size_t)(y0 + t) * rs + x0 - 1] : 129u;
        }
        const uint32_t sa = group_sum<T>(a), sl = group_sum<T>(l);
        uint32_t dc;
        if (x0 > 0 && y0 > 0) dc = (sa + sl + SIZE) / (2 * SIZE);
        else if (x0 > 0) dc = (sl + (SIZE >> 1)) / SIZE;
        else if (y0 > 0) dc = (sa + (SIZE >> 1)) / SIZE;
        else dc = 128;
        if (!newmv) {
#pragma unroll
            for (int i = 0; i < NW; i++) prow[i] = dc * 0x01010101u;
        }
    }
#pragma unroll
    for (int j = 0; j < NW; j++) *(uint32_t*)(ptile + t * PP + 4 * j) = prow[j];
    if (active && (t & ((1 << ST) - 1)) == 0) { // residual on the rows the transform sees
        const int r = t >> ST;
#pragma unroll
        for (int j = 0; j < NW; j++)
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int d = (int)((srow[j] >> (8 * b)) & 255u) - (int)((prow[j] >> (8 * b)) & 255u);
                buf[r * PITCH + 4 * j + b] = (int32_t)((uint32_t)d << FS0);
            }
    }
    if (WSYNC) __builtin_amdgcn_wave_barrier(); else __syncthreads();
    if (active) { // forward, column t
        int32_t v[TXH];
#pragma unroll
        for (int r = 0; r < TXH; r++) v[r] = buf[r * PITCH + t];
        fwd1d<TXH, CBC>(K_DCT, v);
#pragma unroll
        for (int r = 0; r < TXH; r++) buf[r * PITCH + t] = FS1 ? rshift_round(v[r], FS1 ? FS1 : 1) : v[r];
    }
    if (WSYNC) __builtin_amdgcn_wave_barrier(); else __syncthreads();
    unsigned long long err = 0;
    uint32_t           nz  = 0;
    if (active && t < TXH) { // forward, row t; quantize_fp; the dequantised row goes back into the tile
        int32_t v[W];
#pragma unroll
        for (int c = 0; c < W; c++) v[c] = buf[t * PITCH + c];
        fwd1d<W, CBR>(K_DCT, v);
        const int kw = W >> P.pf_shape, kh = TXH >> P.pf_shape;
#pragma unroll
        for (int c = 0; c < W; c++) {
            int32_t dq = 0;
            if (t < kh && c < kw) {
                int32_t x = FS2 ? rshift_round(v[c], FS2 ? FS2 : 1) : v[c];
                if (RECT1) x = mul_sqrt2_like(x, 5793);
                const int     kk = (t | c) != 0;
                const int32_t sign = x < 0 ? -1 : 0, a = (x ^ sign) - sign;
                if (((long long)a << 1) >= (int32_t)P.dequant[kk]) {
                    long long tt = (long long)a + P.round_fp[kk];
                    tt = tt < -32768 ? -32768 : (tt > 32767 ? 32767 : tt);
                    const int32_t q = (int32_t)((tt * P.quant_fp[kk]) >> 16);
                    if (q) { dq = (((int32_t)((uint32_t)q * (uint32_t)(int32_t)P.dequant[kk])) ^ sign) - sign; nz = 1; }
                }
                const long long df = (long long)x - dq;
                err += (unsigned long long)(df * df);
            }
            buf[t * PITCH + c] = dq;
        }
    }
    const uint32_t l0 = group_sum<T>((uint32_t)(err & 0x3fffffu)), l1 = group_sum<T>((uint32_t)((err >> 22) & 0x3fffffu)), l2 = group_sum<T>((uint32_t)(err >> 44));
    const unsigned long long tot = (unsigned long long)l0 + ((unsigned long long)l1 << 22) + ((unsigned long long)l2 << 44);
    const bool coded = group_sum<T>(nz) != 0;
    const bool inverse = active && coded && (!P.disable_intra_pred || RP.is_ref); // (:1135-1136)
    if (WSYNC) __builtin_amdgcn_wave_barrier(); else __syncthreads();
    if (inverse && t < TXH) { // inverse, row t (inv_txfm2d_kernel's first pass, bd 8)
        const int32_t rhi = (1 << 15) - 1, rlo = -(1 << 15);
        int32_t v[W];
#pragma unroll
        for (int c = 0; c < W; c++) {
            int32_t x = buf[t * PITCH + c];
            if (RECT1) x = mul_sqrt2_like(x, 2896);
            v[c] = txfm1d::clamp_i32(x, rlo, rhi);
        }
        inv1d<W>(K_DCT, v, rlo, rhi);
#pragma unroll
        for (int c = 0; c < W; c++) buf[t * PITCH + c] = S0 ? rshift_round(v[c], S0 ? S0 : 1) : v[c];
    }
    if (WSYNC) __builtin_amdgcn_wave_barrier(); else __syncthreads();
    if (inverse) { // inverse, column t, added to the prediction rows the transform saw
        const int32_t chi = (1 << 15) - 1, clo = -(1 << 15);
        int32_t v[TXH];
#pragma unroll
        for (int r = 0; r < TXH; r++) v[r] = txfm1d::clamp_i32(buf[r * PITCH + t], clo, chi);
        inv1d<TXH>(K_DCT, v, clo, chi);
#pragma unroll
        for (int r = 0; r < TXH; r++) {
            int32_t px = (int32_t)ptile[(r << ST) * PP + t] + rshift_round(v[r], 4);
            ptile[(r << ST) * PP + t] = (uint8_t)(px < 0 ? 0 : (px > 255 ? 255 : px));
        }
    }
    if (WSYNC) __builtin_amdgcn_wave_barrier(); else __syncthreads();
    if (active) { // row t of the block: the reconstructed row the transform saw (the rows between are copies, :1149-1167), or the prediction
        const int rr = inverse ? (t & ~((1 << ST) - 1)) : t;
        uint8_t*  d  = rec + (size_t)(y0 + t) * rs + x0;
#pragma unroll
        for (int i = 0; i < NW / 4; i++) {
            tpl_u32x4_a1 o;
            o.x = *(const uint32_t*)(ptile + rr * PP + 16 * i); o.y = *(const uint32_t*)(ptile + rr * PP + 16 * i + 4);
            o.z = *(const uint32_t*)(ptile + rr * PP + 16 * i + 8); o.w = *(const uint32_t*)(ptile + rr * PP + 16 * i + 12);
            if (WT) tpl_store_x4_wt(d + 16 * i, o);
            else *(tpl_u32x4_a1*)(d + 16 * i) = o;
        }
        if (t == 0) {
            long long e = (long long)tot;
            e >>= (SIZE == 32 && TXH == 32) ? 0 : 2;
            if (e < 1) e = 1;
            SvtHipTplReconStats o = {};
            o.written = 1; o.coded = coded;
            o.recrf_dist = (e << TPL_COST_SCALE_LOG2) << ST;
            o.srcrf_dist = newmv ? s.srcrf_dist : o.recrf_dist;
            o.srcrf_rate = newmv ? s.srcrf_rate : 0;
            if (o.srcrf_dist > o.recrf_dist) o.recrf_dist = o.srcrf_dist;
            if (o.srcrf_rate > o.recrf_rate) o.recrf_rate = o.srcrf_rate;
            if (WT) { // everything but `reserved`: in this form the cell's flag is published by a write-through store a plain store of the whole record must not shadow
                SvtHipTplReconStats* q = &out[cell];
                q->srcrf_dist = o.srcrf_dist; q->recrf_dist = o.recrf_dist; q->srcrf_rate = o.srcrf_rate; q->recrf_rate = o.recrf_rate;
                q->written = o.written; q->coded = o.coded; q->pad[0] = 0; q->pad[1] = 0;
            } else {
                out[cell] = o;
            }
        }
    }
}

template <int SIZE, int TXH>
__global__ __launch_bounds__(256) void tpl_recon_kernel(const SvtHipTplReconParams RP, const uint8_t* __restrict__ src_base, const uint8_t* __restrict__ ref_base,
                                                        const SvtHipTplSrcStats* __restrict__ src_stats, uint8_t* __restrict__ recon_base,
                                                        SvtHipTplReconStats* __restrict__ out, const int diag, const int cy_first, const int n_items) {
    constexpr int T = SIZE, BPW = 256 / T, PITCH = SIZE + 1, PP = SIZE + 4;
    HIP_DYNAMIC_SHARED(int32_t, smem)
    __shared__ SvtHipTplRef s_refs[8];
    if (threadIdx.x < 8) s_refs[threadIdx.x] = RP.rec_refs[threadIdx.x];
    __syncthreads();
    const int tid = threadIdx.x, sub = tid / T, t = tid % T;
    const int item = (int)blockIdx.x * BPW + sub;
    int32_t*  buf  = smem + sub * (TXH * PITCH);
    uint8_t*  ptile = (uint8_t*)(smem + BPW * (TXH * PITCH)) + sub * (SIZE * PP);
    const int cy = cy_first + item;
    TplBlockIn<SIZE> B;
    tpl_recon_fetch<SIZE>(RP, s_refs, src_base, ref_base, src_stats, diag - cy, cy, item < n_items, t, B);
    tpl_recon_compute<SIZE, TXH>(RP, recon_base, out, diag - cy, cy, B, t, buf, ptile);
}

// ---- the same blocks as a row wavefront in ONE launch (SVT_HIP_TPL_RECON_FORM=1; not the default until it has been timed on the device) -------------------------
// One single-wave workgroup per block row walks its blocks left to right; before block cx it waits until the row above has published cells [cx, cx + SIZE / 16)
// (progress counter of that cell row, in cells; kept in SvtHipTplReconStats.reserved of the row's first cell), after the block it publishes its own.  Row r only ever
// waits for row r - 1 and the grid (<= 135 waves at 4K) is resident at once, so the wait always ends; it is bounded all the same (a timed-out row marks pad[0] of its
// first cell and goes on: the caller sees it).  Stores are fenced at device scope before the counter moves, the reader fences after it saw the counter.
constexpr uint32_t TPL_WAIT_POLLS = 1u << 22;
template <int SIZE, int TXH>
__global__ __launch_bounds__(64) void tpl_recon_rows_kernel(const SvtHipTplReconParams RP, const uint8_t* __restrict__ src_base, const uint8_t* __restrict__ ref_base,
                                                            const SvtHipTplSrcStats* __restrict__ src_stats, uint8_t* __restrict__ recon_base,
                                                            SvtHipTplReconStats* __restrict__ out, const int cy_first, const int xcds, const int rel_acq) {
    constexpr int PITCH = SIZE + 1, PP = SIZE + 4, CELLS = SIZE / 16;
    HIP_DYNAMIC_SHARED(int32_t, smem)
    __shared__ SvtHipTplRef s_refs[8];
    if (threadIdx.x < 8) s_refs[threadIdx.x] = RP.rec_refs[threadIdx.x];
    __syncthreads();
    constexpr int SUBS = 64 / SIZE; // the wave's other lane groups idle along (their own tiles: the block code stores its prediction row unconditionally)
    const int t = (int)threadIdx.x % SIZE, sub = (int)threadIdx.x / SIZE;
    int32_t*  buf = smem + sub * (TXH * PITCH);
    uint8_t*  ptile = (uint8_t*)(smem + SUBS * (TXH * PITCH)) + sub * (SIZE * PP);
    // xcds > 1 (form 2): workgroup ids go round-robin over the XCDs, so consecutive ROWS are given to ids of the same XCD (contiguous chunks of rows per XCD): the
    // counter and the reconstruction a row hands to the next then stay inside one L2 except at the chunk borders
    int row = (int)blockIdx.x;
    if (xcds > 1) {
        const int nb = (int)gridDim.x, x = row % xcds, k = row / xcds, q = nb / xcds, r = nb % xcds;
        row = x * q + (x < r ? x : r) + k;
    }
    const int cols16 = (int)((RP.src.aligned_width + 15) >> 4), cy = cy_first + row * CELLS;
    bool      timed_out = false;
    TplBlockIn<SIZE> B;
    tpl_recon_fetch<SIZE>(RP, s_refs, src_base, ref_base, src_stats, 0, cy, sub == 0, t, B);
    for (int cx = 0; cx < cols16; cx += CELLS) {
        // a NEWMV block reads nothing of the row above; only a DC block waits for it (the group's lanes agree on the mode; the other lane groups have no block)
        const bool need_above = __shfl((int)(B.active && !B.newmv), 0) != 0; // (lane 0 belongs to the live group)
        if (cy > 0 && need_above) {
            const uint32_t need = (uint32_t)(cx + CELLS < cols16 ? cx + CELLS : cols16);
            uint32_t*      flag = &out[(size_t)(cy - 1) * cols16].reserved;
            uint32_t       polls = 0;
            while (atomicAdd(flag, 0u) < need && polls < TPL_WAIT_POLLS) { polls++; __builtin_amdgcn_s_sleep(2); }
            timed_out = timed_out || polls >= TPL_WAIT_POLLS;
            if (rel_acq) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); // form 3: invalidate only
            else __threadfence();
        }
        tpl_recon_compute<SIZE, TXH>(RP, recon_base, out, cx, cy, B, t, buf, ptile);
        if (cx + CELLS < cols16) tpl_recon_fetch<SIZE>(RP, s_refs, src_base, ref_base, src_stats, cx + CELLS, cy, sub == 0, t, B); // in flight while the stores drain
        if (rel_acq) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); // form 3: write back only (the sequentially-consistent fence also invalidates the L2: DESIGN 4.16)
        else __threadfence();
        __syncthreads();
        if (threadIdx.x == 0)
            for (int k = 0; k < CELLS; k++) atomicExch(&out[(size_t)(cy + k) * cols16].reserved, (uint32_t)(cx + CELLS < cols16 ? cx + CELLS : cols16));
    }
    if (threadIdx.x == 0 && timed_out) out[(size_t)cy * cols16].pad[0] = 0xEE;
}
__global__ void tpl_recon_rows_reset_kernel(SvtHipTplReconStats* __restrict__ out, const int cols16, const int rows16) {
    const int cy = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (cy < rows16) { out[(size_t)cy * cols16].reserved = 0; out[(size_t)cy * cols16].pad[0] = 0; }
}

// ---- the same blocks with the wavefront expressed as DATA dependencies, everything else in parallel (form 4, the default) ----------------------------------------------
// The anti-diagonal and the row forms both serialise blocks that do not depend on each other: only a DC (intra) block reads its neighbours' reconstruction, a NEWMV
// block reads nothing of this picture.  Here every block is in flight at once, ONE WAVE PER BLOCK (a block waiting for its neighbours then holds up nobody else: with
// sixteen blocks per workgroup, a third of them intra, every workgroup waited and the launch degenerated into the full wavefront -- 1.64 ms, profiles/r04_call2_*).
// The waves of the launch take tickets in anti-diagonal order (ticket -> diagonal u of the block grid, block row by; block (u - by, by)), so that the upper and the
// left neighbour of any block belong to a ticket drawn EARLIER: a waiting wave can only wait for waves that are already running, whatever the residency of the grid.
// Each 16x16 cell of the statistics grid carries a "reconstructed" flag (SvtHipTplReconStats.reserved, cleared by tpl_recon_rows_reset_kernel): a DC block polls the
// flags of the cells above and left of it, everybody publishes its own cells after its stores are fenced.  The dependency chains that remain are runs of adjacent intra
// blocks -- the critical path of an inter picture is a few blocks, not cols + rows of them; an all-intra picture degenerates to the wavefront the other forms always
// pay.  Level 1 (32x32 blocks in complete SBs, 16x16 blocks in SBs cut by the picture edge) is two launches in stream order: a 32x32 block never has a 16x16 neighbour
// on its left or above, so the first launch owns the cells of the complete SBs and the second one finds them published.
constexpr int DEP_WAVES = 8; // waves (= blocks in flight) per workgroup of the dependency form: ONE ticket draw per workgroup -- 8 160 returning atomics on one word, one per
                             // single-wave workgroup, were 126 of the kernel's 213 us (the same kernel with the workgroup id as its ticket: 87 us, gpurun call 39).  2 / 4 / 8 / 16 waves: 140 / 109 /
                             // 97 / 123 us (call 42)
template <int SIZE, int TXH>
__global__ __launch_bounds__(64 * DEP_WAVES) void tpl_recon_dep_kernel(const SvtHipTplReconParams RP, const uint8_t* __restrict__ src_base, const uint8_t* __restrict__ ref_base,
                                                           const SvtHipTplSrcStats* __restrict__ src_stats, uint8_t* __restrict__ recon_base,
                                                           SvtHipTplReconStats* __restrict__ out, uint32_t* __restrict__ sync, const int ticket_slot, const int cols_b,
                                                           const int rows_b, const int rel_acq) {
    constexpr int PITCH = SIZE + 1, PP = SIZE + 4, UNIT = SIZE / 16, SUBS = 64 / SIZE;
    HIP_DYNAMIC_SHARED(int32_t, smem)
    __shared__ SvtHipTplRef s_refs[8];
    __shared__ uint32_t     s_ticket;
    if (threadIdx.x < 8) s_refs[threadIdx.x] = RP.rec_refs[threadIdx.x];
    if (threadIdx.x == 0) s_ticket = atomicAdd(&sync[ticket_slot], (uint32_t)DEP_WAVES); // the workgroup's waves take consecutive tickets
    __syncthreads(); // the ONLY workgroup barrier: from here on the waves are independent blocks (a wave waiting for its neighbours holds up nobody else)
    const SvtHipTplSrcParams& P = RP.src;
    const int lane = (int)threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int t = lane % SIZE, sub = lane / SIZE; // (lane groups 1 .. SUBS - 1 idle along on tiles of their own: the block code stores its prediction row unconditionally)
    const int ticket = (int)s_ticket + wv, u = ticket / rows_b, by = ticket % rows_b, bx = u - by;
    if (bx < 0 || bx >= cols_b) return; // (wave-uniform) no block under this ticket
    const int cx = bx * UNIT, cy = by * UNIT;
    constexpr int WAVE_DW = SUBS * (TXH * PITCH) + (SUBS * SIZE * PP + 3) / 4; // the wave's own LDS slice, in dwords
    int32_t*  wsm   = smem + wv * WAVE_DW;
    int32_t*  buf   = wsm + sub * (TXH * PITCH);
    uint8_t*  ptile = (uint8_t*)(wsm + SUBS * (TXH * PITCH)) + sub * (SIZE * PP);
    const int aligned_h = (int)((P.height + 7) & ~7u), cols16 = (int)((P.aligned_width + 15) >> 4), rows16 = (aligned_h + 15) >> 4;
    // the cells this wave answers for in THIS launch: the blocks of its size class, processed or not (a skipped block's neighbours must not wait for ever)
    const bool complete = ((int)P.aligned_width - ((cx * 16) & ~63) >= 64) && (aligned_h - ((cy * 16) & ~63) >= 64);
    if (((complete && P.dispenser_search_level) ? 32 : 16) != SIZE) return; // (uniform) the other launch's block
    TplBlockIn<SIZE> B;
    tpl_recon_fetch<SIZE>(RP, s_refs, src_base, ref_base, src_stats, cx, cy, sub == 0, t, B);
    const bool dc = __shfl((int)(B.active && !B.newmv), 0) != 0; // (lane 0 belongs to the live group)
    if (dc) { // its upper and left cells must be reconstructed; one lane per cell polls
        const int  n_up = cy > 0 ? (cols16 - cx < UNIT ? cols16 - cx : UNIT) : 0, n_left = cx > 0 ? (rows16 - cy < UNIT ? rows16 - cy : UNIT) : 0;
        const int  l = lane;
        bool       timed_out = false;
        if (l < n_up + n_left) {
            uint32_t* flag  = l < n_up ? &out[(size_t)(cy - 1) * cols16 + cx + l].reserved : &out[(size_t)(cy + l - n_up) * cols16 + cx - 1].reserved;
            uint32_t  polls = 0;
            // (rel_acq == 2, form 6: the poll is an atomic LOAD at agent scope -- many waves poll the same neighbours' flags, and a read-modify-write per poll queues them at
            // the L2's atomic unit, a load does not)
            while ((rel_acq >= 2 ? __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : atomicAdd(flag, 0u)) == 0u && polls < TPL_WAIT_POLLS) { polls++; __builtin_amdgcn_s_sleep(1); }
            timed_out = polls >= TPL_WAIT_POLLS;
        }
        if (timed_out) atomicAdd(&sync[1], 1u);
        if (rel_acq) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); // (also form 7: the hand-off's consumer side is ONE acquire, then plain loads)
        else __threadfence();
        __builtin_amdgcn_wave_barrier(); // (nothing on the device, where the polling lanes hold their wave; the CPU emulator's lanes are fibers: the others must not read ahead)
    }
    if (rel_acq == 3) {
        tpl_recon_compute<SIZE, TXH, true, true>(RP, recon_base, out, cx, cy, B, t, buf, ptile);
        svt_hip_drain_stores(); // every lane's write-through stores have left (a wave executes the wait as one) before lane 0 publishes the cells
    } else {
        tpl_recon_compute<SIZE, TXH, false, true>(RP, recon_base, out, cx, cy, B, t, buf, ptile);
        if (rel_acq) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        else __threadfence();
    }
    __builtin_amdgcn_wave_barrier();
    if (lane == 0)
        for (int j = 0; j < UNIT && cy + j < rows16; j++)
            for (int i = 0; i < UNIT && cx + i < cols16; i++) {
                if (rel_acq >= 2) __hip_atomic_store(&out[(size_t)(cy + j) * cols16 + cx + i].reserved, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else atomicExch(&out[(size_t)(cy + j) * cols16 + cx + i].reserved, 1u);
            }
}
__global__ void tpl_recon_dep_finish_kernel(SvtHipTplReconStats* __restrict__ out, const uint32_t* __restrict__ sync) {
    if (threadIdx.x == 0 && blockIdx.x == 0 && sync[1]) out[0].pad[0] = 0xEE; // a block gave up waiting: the caller sees it (as the row form's marker)
}
template <int SIZE, int TXH>
void launch_tpl_recon_dep(const SvtHipTplReconParams& P, const uint8_t* src, const uint8_t* ref, const SvtHipTplSrcStats* ss, uint8_t* rec, SvtHipTplReconStats* out,
                          uint32_t* sync, int ticket_slot, int cols16, int rows16, int rel_acq, hipStream_t st) {
    constexpr int SUBS = 64 / SIZE, UNIT = SIZE / 16;
    const int     cols_b = (cols16 + UNIT - 1) / UNIT, rows_b = (rows16 + UNIT - 1) / UNIT;
    const size_t  wave_dw = (size_t)SUBS * TXH * (SIZE + 1) + ((size_t)SUBS * SIZE * (SIZE + 4) + 3) / 4, shmem = DEP_WAVES * wave_dw * 4;
    const int     tickets = (cols_b + rows_b - 1) * rows_b;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(tpl_recon_dep_kernel<SIZE, TXH>), dim3((tickets + DEP_WAVES - 1) / DEP_WAVES), dim3(64 * DEP_WAVES), shmem, st, P, src, ref, ss, rec, out, sync,
                       ticket_slot, cols_b, rows_b, rel_acq);
    SVT_LAUNCH_CHECK();
}

template <int SIZE, int TXH>
void launch_tpl_recon(const SvtHipTplReconParams& P, const uint8_t* src, const uint8_t* ref, const SvtHipTplSrcStats* ss, uint8_t* rec, SvtHipTplReconStats* out,
                      int diag, int cy_first, int n_items, hipStream_t st) {
    constexpr int BPW = 256 / SIZE;
    const size_t  shmem = (size_t)BPW * TXH * (SIZE + 1) * 4 + (size_t)BPW * SIZE * (SIZE + 4);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(tpl_recon_kernel<SIZE, TXH>), dim3((n_items + BPW - 1) / BPW), dim3(256), shmem, st, P, src, ref, ss, rec, out, diag, cy_first,
                       n_items);
    SVT_LAUNCH_CHECK();
}

bool tpl_supported(const SvtHipTplSrcParams& P) {
    if (svthip::tpl_full_wanted(P)) return svthip::tpl_full_supported(P); // the option set of tpl levels 0-3: tpl_full.hip
    if (P.dispenser_search_level > 1 || P.subsample_tx > 2 || P.pf_shape > 2 || !P.n_sb || !P.sbs_x) return false;
    if (P.dispenser_search_level == 1 && P.subsample_tx != 2) return false; // 32x32 blocks exist with TX_32X8 only (tpl level 5)
    if (P.dispenser_search_level == 0 && P.subsample_tx == 1) return false; // TX_16X8: no tpl level uses it
    return true;
}

} // namespace

extern "C" {

void svt_hip_tpl_src_stage(const SvtHipTplSrcParams* params, const uint8_t* src_base, const uint8_t* ref_base, const uint8_t* total_me_candidate_index,
                           const uint32_t* me_mv_array, const uint8_t* me_candidate_array, SvtHipTplSrcStats* stats, void* stream) {
    svthip::ensure_device();
    const SvtHipTplSrcParams& P = *params;
    if (!tpl_supported(P)) {
        fprintf(stderr, "libsvtav1_hip: svt_hip_tpl_src_stage: option set not covered (level %d, subsample_tx %d, pf_shape %d)\n", P.dispenser_search_level,
                P.subsample_tx, P.pf_shape);
        abort();
    }
    hipStream_t st = (hipStream_t)stream;
    if (svthip::tpl_full_wanted(P)) {
        svthip::tpl_full_src_launch(P, src_base, ref_base, total_me_candidate_index, me_mv_array, me_candidate_array, stats, st);
        return;
    }
    if (P.dispenser_search_level == 0) {
        if (P.subsample_tx == 0) launch_tpl<16, 16>(P, src_base, ref_base, total_me_candidate_index, me_mv_array, me_candidate_array, stats, 0, st);
        else launch_tpl<16, 4>(P, src_base, ref_base, total_me_candidate_index, me_mv_array, me_candidate_array, stats, 0, st);
    } else {
        launch_tpl<32, 8>(P, src_base, ref_base, total_me_candidate_index, me_mv_array, me_candidate_array, stats, 0, st);
        launch_tpl<16, 4>(P, src_base, ref_base, total_me_candidate_index, me_mv_array, me_candidate_array, stats, 1, st); // SBs the picture edge cuts: level 0 (:2048-2051)
    }
}

int svt_hip_tpl_src_stage_host(const SvtHipTplSrcParams* params, const SvtHipTplHostPlanes* planes, const uint8_t* total_me_candidate_index,
                               const uint32_t* me_mv_array, const uint8_t* me_candidate_array, SvtHipTplSrcStats* stats) {
    SVT_HIP_ENTRY_TRY
    svthip::ensure_device();
    SvtHipTplSrcParams P = *params;
    if (!tpl_supported(P)) return -1;
    const int    n_pus = P.enable_me_8x8 ? 85 : (P.enable_me_16x16 ? 21 : 5);
    const size_t cols16 = (P.aligned_width + 15) >> 4, rows16 = ((((size_t)P.height + 7) & ~(size_t)7) + 15) >> 4, cells = cols16 * rows16;
    const size_t tot_b = (size_t)P.n_sb * n_pus, mv_b = tot_b * P.max_refs * 4, cand_b = tot_b * P.max_cand;
    // distinct picture buffers: the source, then every valid reference whose buffer was not seen before
    const uint8_t* bufs[9];
    size_t         bytes[9], doff[9];
    int            nb = 0, ref_slot[8];
    bufs[nb] = planes->src_buf; bytes[nb] = (size_t)P.src_stride * planes->src_rows; nb++;
    for (int r = 0; r < 8; r++) {
        ref_slot[r] = -1;
        if (!P.refs[r].valid || P.i_slice) continue;
        for (int b = 0; b < nb; b++)
            if (bufs[b] == planes->ref_buf[r]) ref_slot[r] = b;
        if (ref_slot[r] < 0) { bufs[nb] = planes->ref_buf[r]; bytes[nb] = (size_t)P.refs[r].stride * planes->ref_rows[r]; ref_slot[r] = nb++; }
    }
    size_t total = 0;
    for (int b = 0; b < nb; b++) { doff[b] = total; total += svthip::align_up(bytes[b], 256); }
    svthip::HostCallLease lease; // (a pooled arena: see svt_hip_common.h)
    svthip::HostCall& c = *lease;
    c.begin();
    const size_t side = tot_b + mv_b + cand_b + cells * sizeof(SvtHipTplSrcStats) + 8192;
    c.reserve(total + side + 4096, total + side + cells * sizeof(SvtHipTplSrcStats) + 4096);
    uint8_t* d_planes = (uint8_t*)c.dalloc(total);
    for (int b = 0; b < nb; b++) c.up(d_planes + doff[b], bufs[b], bytes[b]);
    uint8_t*  d_tot  = (uint8_t*)c.dalloc(tot_b);
    uint32_t* d_mv   = (uint32_t*)c.dalloc(mv_b ? mv_b : 4);
    uint8_t*  d_cand = (uint8_t*)c.dalloc(cand_b ? cand_b : 4);
    SvtHipTplSrcStats* d_stats = (SvtHipTplSrcStats*)c.dalloc(cells * sizeof(SvtHipTplSrcStats));
    c.up(d_tot, total_me_candidate_index, tot_b);
    if (mv_b) c.up(d_mv, me_mv_array, mv_b);
    if (cand_b) c.up(d_cand, me_candidate_array, cand_b);
    HIP_CHECK(hipMemsetAsync(d_stats, 0, cells * sizeof(SvtHipTplSrcStats), c.stream));
    for (int r = 0; r < 8; r++)
        if (ref_slot[r] >= 0) P.refs[r].plane_off += doff[ref_slot[r]];
    svt_hip_tpl_src_stage(&P, d_planes, d_planes, d_tot, d_mv, d_cand, d_stats, c.stream);
    c.down(stats, d_stats, cells * sizeof(SvtHipTplSrcStats));
    return 0;
    SVT_HIP_ENTRY_CATCH(SVT_HIP_E_DEVICE)
}

void svt_hip_tpl_recon_stage(const SvtHipTplReconParams* params, const uint8_t* src_base, const uint8_t* rec_ref_base, const SvtHipTplSrcStats* src_stats,
                             uint8_t* recon_base, SvtHipTplReconStats* out, void* stream) {
    svthip::ensure_device();
    const SvtHipTplReconParams& R = *params;
    const SvtHipTplSrcParams&   P = R.src;
    if (!tpl_supported(P)) {
        fprintf(stderr, "libsvtav1_hip: svt_hip_tpl_recon_stage: option set not covered (level %d, subsample_tx %d, pf_shape %d)\n", P.dispenser_search_level,
                P.subsample_tx, P.pf_shape);
        abort();
    }
    hipStream_t st = (hipStream_t)stream;
    const int   aligned_h = (int)((P.height + 7) & ~7u), cols16 = (int)((P.aligned_width + 15) >> 4), rows16 = (aligned_h + 15) >> 4;
    const bool  edge_sbs = (P.aligned_width & 63) || (aligned_h & 63); // SBs the picture edge cuts run at level 0 (:2048-2051)
    const char* form_env = getenv("SVT_HIP_TPL_RECON_FORM"); // (read per call: a picture-sized stage, and the tests switch it inside one process)
    const int   form = form_env ? atoi(form_env) : 7;
    if (svthip::tpl_full_wanted(P)) { // tpl levels 0-3: one launch, dependencies as data (the only form of that option set)
        uint32_t* sync = svthip::stream_scratch_u32x4(st);
        hipLaunchKernelGGL(tpl_recon_rows_reset_kernel, dim3((cols16 * rows16 + 255) / 256), dim3(256), 0, st, out, 1, cols16 * rows16);
        SVT_LAUNCH_CHECK();
        svthip::tpl_full_recon_launch(R, src_base, rec_ref_base, src_stats, recon_base, out, sync, cols16, rows16, form == 5 || form == 4 ? 0 : 1, st);
        hipLaunchKernelGGL(tpl_recon_dep_finish_kernel, dim3(1), dim3(64), 0, st, out, sync);
        SVT_LAUNCH_CHECK();
        return;
    }
    if (form == 4 || form == 5 || form == 6 || form == 7) { // (6: as 5, polling with atomic loads instead of read-modify-writes) dependencies as data, every block in flight: 7 (the default) = 6 with write-through stores + drained flag instead of the release fence (213 us against 339: gpurun call 37); 5 with release / acquire fences at agent scope, 4 with sequentially-consistent
                                  // ones (416 us against 339 us for a 1080p picture with a third of its blocks intra: profiles/r04_call3_tpl_forms.txt)
        uint32_t* sync = svthip::stream_scratch_u32x4(st); // [0] tickets of the first launch, [1] blocks that gave up waiting, [2] tickets of the second launch
        hipLaunchKernelGGL(tpl_recon_rows_reset_kernel, dim3((cols16 * rows16 + 255) / 256), dim3(256), 0, st, out, 1, cols16 * rows16); // every cell's flag
        SVT_LAUNCH_CHECK();
        if (P.dispenser_search_level == 0) {
            if (P.subsample_tx == 0) launch_tpl_recon_dep<16, 16>(R, src_base, rec_ref_base, src_stats, recon_base, out, sync, 0, cols16, rows16, form == 7 ? 3 : (form == 6 ? 2 : (form == 5)), st);
            else launch_tpl_recon_dep<16, 4>(R, src_base, rec_ref_base, src_stats, recon_base, out, sync, 0, cols16, rows16, form == 7 ? 3 : (form == 6 ? 2 : (form == 5)), st);
        } else {
            launch_tpl_recon_dep<32, 8>(R, src_base, rec_ref_base, src_stats, recon_base, out, sync, 0, cols16, rows16, form == 7 ? 3 : (form == 6 ? 2 : (form == 5)), st);
            if (edge_sbs) launch_tpl_recon_dep<16, 4>(R, src_base, rec_ref_base, src_stats, recon_base, out, sync, 2, cols16, rows16, form == 7 ? 3 : (form == 6 ? 2 : (form == 5)), st);
        }
        hipLaunchKernelGGL(tpl_recon_dep_finish_kernel, dim3(1), dim3(64), 0, st, out, sync);
        SVT_LAUNCH_CHECK();
        return;
    }
    if (form >= 1 && form <= 3 && (P.dispenser_search_level == 0 || (P.aligned_width & 63) == 0)) { // the row wavefront: uniform block size per row (no SB column cut by the right edge)
        hipLaunchKernelGGL(tpl_recon_rows_reset_kernel, dim3((rows16 + 63) / 64), dim3(64), 0, st, out, cols16, rows16);
        SVT_LAUNCH_CHECK();
        auto rows = [&](auto size_tag, auto txh_tag, int cy_first, int n_rows) {
            constexpr int SIZE = decltype(size_tag)::value, TXH = decltype(txh_tag)::value, SUBS = 64 / SIZE;
            if (n_rows <= 0) return;
            const size_t shmem = (size_t)SUBS * TXH * (SIZE + 1) * 4 + (size_t)SUBS * SIZE * (SIZE + 4);
            hipLaunchKernelGGL(HIP_KERNEL_NAME(tpl_recon_rows_kernel<SIZE, TXH>), dim3(n_rows), dim3(64), shmem, st, R, src_base, rec_ref_base, src_stats, recon_base, out,
                               cy_first, form == 2 ? 8 : 1, form == 3 ? 1 : 0);
            SVT_LAUNCH_CHECK();
        };
        using I16 = std::integral_constant<int, 16>; using I32 = std::integral_constant<int, 32>; using I4 = std::integral_constant<int, 4>; using I8 = std::integral_constant<int, 8>;
        if (P.dispenser_search_level == 0) {
            if (P.subsample_tx == 0) rows(I16{}, I16{}, 0, rows16);
            else rows(I16{}, I4{}, 0, rows16);
        } else {
            const int full_sb_rows = aligned_h / 64; // complete SB rows hold 32x32 blocks, the SB row the bottom edge cuts 16x16 blocks (a later launch: stream order)
            rows(I32{}, I8{}, 0, full_sb_rows * 2);
            rows(I16{}, I4{}, full_sb_rows * 4, rows16 - full_sb_rows * 4);
        }
        return;
    }
    for (int d = 0; d < cols16 + rows16 - 1; d++) {
        const int cy0 = d - (cols16 - 1) > 0 ? d - (cols16 - 1) : 0, cy1 = d < rows16 - 1 ? d : rows16 - 1, n = cy1 - cy0 + 1;
        if (P.dispenser_search_level == 0) {
            if (P.subsample_tx == 0) launch_tpl_recon<16, 16>(R, src_base, rec_ref_base, src_stats, recon_base, out, d, cy0, n, st);
            else launch_tpl_recon<16, 4>(R, src_base, rec_ref_base, src_stats, recon_base, out, d, cy0, n, st);
        } else {
            if (!(d & 1)) launch_tpl_recon<32, 8>(R, src_base, rec_ref_base, src_stats, recon_base, out, d, cy0, n, st); // 32x32 blocks start on even cells
            // 16x16 blocks live in the SBs the picture edge cuts: cell rows from the last complete SB row on, cell columns from the last complete SB column on
            const int edge_row = (aligned_h & 63) ? (aligned_h / 64) * 4 : rows16, edge_col = ((int)P.aligned_width & 63) ? ((int)P.aligned_width / 64) * 4 : cols16;
            if (edge_sbs && (cy1 >= edge_row || d - cy0 >= edge_col)) launch_tpl_recon<16, 4>(R, src_base, rec_ref_base, src_stats, recon_base, out, d, cy0, n, st);
        }
    }
}

int svt_hip_tpl_recon_stage_host(const SvtHipTplReconParams* params, const SvtHipTplHostPlanes* planes, const SvtHipTplSrcStats* src_stats, uint8_t* recon_buf,
                                 uint32_t recon_rows, SvtHipTplReconStats* out) {
    SVT_HIP_ENTRY_TRY
    svthip::ensure_device();
    SvtHipTplReconParams R = *params;
    const SvtHipTplSrcParams& P = R.src;
    if (!tpl_supported(P)) return -1;
    const size_t cols16 = (P.aligned_width + 15) >> 4, rows16 = ((((size_t)P.height + 7) & ~(size_t)7) + 15) >> 4, cells = cols16 * rows16;
    bool used[8] = {};
    for (size_t i = 0; i < cells; i++)
        if (src_stats[i].written && src_stats[i].best_mode == TPL_NEWMV) {
            if (src_stats[i].best_rf_idx < 0 || src_stats[i].best_rf_idx > 7) return -2;
            used[src_stats[i].best_rf_idx] = true;
        }
    const uint8_t* bufs[9];
    size_t         bytes[9], doff[9];
    int            nb = 0, ref_slot[8];
    bufs[nb] = planes->src_buf; bytes[nb] = (size_t)P.src_stride * planes->src_rows; nb++;
    for (int r = 0; r < 8; r++) {
        ref_slot[r] = -1;
        if (!used[r]) continue;
        if (!planes->ref_buf[r]) return -3;
        for (int b = 0; b < nb; b++)
            if (bufs[b] == planes->ref_buf[r]) ref_slot[r] = b;
        if (ref_slot[r] < 0) { bufs[nb] = planes->ref_buf[r]; bytes[nb] = (size_t)R.rec_refs[r].stride * planes->ref_rows[r]; ref_slot[r] = nb++; }
    }
    size_t total = 0;
    for (int b = 0; b < nb; b++) { doff[b] = total; total += svthip::align_up(bytes[b], 256); }
    const size_t rec_b = (size_t)R.recon_stride * recon_rows;
    svthip::HostCallLease lease;
    svthip::HostCall& c = *lease;
    c.begin();
    const size_t side = cells * (sizeof(SvtHipTplSrcStats) + sizeof(SvtHipTplReconStats)) + 8192;
    c.reserve(total + rec_b + side + 4096, total + 2 * rec_b + 2 * side + 8192);
    uint8_t* d_planes = (uint8_t*)c.dalloc(total);
    for (int b = 0; b < nb; b++) c.up(d_planes + doff[b], bufs[b], bytes[b]);
    uint8_t*             d_rec = (uint8_t*)c.dalloc(rec_b);
    SvtHipTplSrcStats*   d_ss  = (SvtHipTplSrcStats*)c.dalloc(cells * sizeof(SvtHipTplSrcStats));
    SvtHipTplReconStats* d_out = (SvtHipTplReconStats*)c.dalloc(cells * sizeof(SvtHipTplReconStats));
    // (the plane's content is never read: a DC block's neighbours are blocks of this picture, written earlier in the call; beyond the picture the fill values apply)
    HIP_CHECK(hipMemsetAsync(d_rec, 0, rec_b, c.stream));
    c.up(d_ss, src_stats, cells * sizeof(SvtHipTplSrcStats));
    HIP_CHECK(hipMemsetAsync(d_out, 0, cells * sizeof(SvtHipTplReconStats), c.stream));
    for (int r = 0; r < 8; r++)
        if (ref_slot[r] >= 0) R.rec_refs[r].plane_off += doff[ref_slot[r]];
    svt_hip_tpl_recon_stage(&R, d_planes, d_planes, d_ss, d_rec, d_out, c.stream);
    // Only the rectangle the blocks of this picture wrote comes back: a block is processed when at least half of it lies inside the picture (:580), so the written
    // area is [0, covered_w) x [0, covered_h) with covered = ((size + 8) >> 4) << 4; everything else of the caller's plane -- its borders (tpl_mc_flow_dispenser pads
    // the plane afterwards, :1400-1406) and, for picture sizes with (size % 16) in 1..7, the columns / rows of the skipped blocks -- keeps the caller's content,
    // as it does in the reference.
    const size_t covered_w = ((size_t)P.width + 8) >> 4 << 4, covered_h = ((size_t)P.height + 8) >> 4 << 4;
    size_t       rows_down = covered_h;
    while (rows_down && R.recon_off + (rows_down - 1) * R.recon_stride + covered_w > rec_b) rows_down--; // (never the case for a plane with the reference's borders)
    if (rows_down) { // one DMA of the rows' span into the pinned arena, then the covered columns of each row
        const size_t span = (rows_down - 1) * R.recon_stride + covered_w;
        uint8_t*     pin  = (uint8_t*)c.palloc(span);
        HIP_CHECK(hipMemcpyAsync(pin, d_rec + R.recon_off, span, hipMemcpyDeviceToHost, c.stream));
        c.sync();
        for (size_t y = 0; y < rows_down; y++) memcpy(recon_buf + R.recon_off + y * R.recon_stride, pin + y * R.recon_stride, covered_w);
    }
    c.down(out, d_out, cells * sizeof(SvtHipTplReconStats));
    for (size_t r = 0; r < rows16; r++)
        if (out[r * cols16].pad[0] == 0xEE) return -4; // the row-wavefront form gave up waiting (see tpl_recon_rows_kernel)
    return 0;
    SVT_HIP_ENTRY_CATCH(SVT_HIP_E_DEVICE)
}

// Both halves of the dispenser for one picture in ONE host call: every distinct picture buffer -- the source, the source pictures of its references (source-based
// half), the TPL reconstructions of its references (reconstruction half) -- is uploaded once, the source-based statistics stay on the device between the halves (they
// come back for the caller's TplSrcStats buffer all the same), one synchronisation at the end.  What the seam calls for a picture whose statistics are not stored yet.
//
// `ids` (the _resident form; NULL otherwise) names the CONTENT of every buffer -- picture number + 1 for a source picture, the same for a picture's TPL
// reconstruction -- so that a plane stays on the device across calls (svthip::plane_cache_*): a picture of a TPL group is the source of one call and a reference of
// several others, and its reconstruction is PRODUCED on the device; with ids a call uploads one new plane (its source) instead of up to nine.  The reconstruction
// kept on the device gets the borders the host's copy gets from svt_aom_generate_padding after the dispenser (src_ops_process.c:1400-1406) -- later pictures' vectors
// reach up to 32 samples past the picture --, which is why ids carries its geometry.
static int tpl_stage_host_impl(const SvtHipTplReconParams* params, const SvtHipTplHostPlanes* src_planes, const SvtHipTplHostPlanes* rec_planes, const SvtHipTplPlaneIds* ids,
                               const uint8_t* total_me_candidate_index, const uint32_t* me_mv_array, const uint8_t* me_candidate_array, SvtHipTplSrcStats* src_stats,
                               uint8_t* recon_buf, uint32_t recon_rows, SvtHipTplReconStats* out) {
    svthip::ensure_device();
    SvtHipTplReconParams R = *params;
    SvtHipTplSrcParams&  P = R.src;
    if (!tpl_supported(P)) return -1;
    const int    n_pus = P.enable_me_8x8 ? 85 : (P.enable_me_16x16 ? 21 : 5);
    const size_t cols16 = (P.aligned_width + 15) >> 4, rows16 = ((((size_t)P.height + 7) & ~(size_t)7) + 15) >> 4, cells = cols16 * rows16;
    const size_t tot_b = (size_t)P.n_sb * n_pus, mv_b = tot_b * P.max_refs * 4, cand_b = tot_b * P.max_cand;
    const uint8_t* bufs[17];
    size_t         bytes[17];
    uint64_t       bid[17];
    int            nb = 0, src_slot[8], rec_slot[8];
    auto slot_of = [&](const uint8_t* b, size_t n, uint64_t id) {
        for (int i = 0; i < nb; i++)
            if (bufs[i] == b) return i;
        bufs[nb] = b; bytes[nb] = n; bid[nb] = id;
        return nb++;
    };
    const bool stored = total_me_candidate_index == nullptr; // the source-based statistics come from the caller (an earlier TPL group's: :969-977): reconstruction half only
    slot_of(src_planes->src_buf, (size_t)P.src_stride * src_planes->src_rows, ids ? ids->src : 0);
    for (int r = 0; r < 8; r++) {
        src_slot[r] = rec_slot[r] = -1;
        if (!P.refs[r].valid || P.i_slice) continue;
        if ((!stored && !src_planes->ref_buf[r]) || !rec_planes->ref_buf[r]) return -3;
        if (!stored) src_slot[r] = slot_of(src_planes->ref_buf[r], (size_t)P.refs[r].stride * src_planes->ref_rows[r], ids ? ids->src_ref[r] : 0); // (only the source-based half reads them)
        rec_slot[r] = slot_of(rec_planes->ref_buf[r], (size_t)R.rec_refs[r].stride * rec_planes->ref_rows[r], ids ? ids->rec_ref[r] : 0);
    }
    const size_t rec_b = (size_t)R.recon_stride * recon_rows;
    // resident planes first (pinned for the call), the rest into the call's arena
    uint8_t* dptr[17];
    int      token[17], rec_token = -1;
    bool     hit[17], rec_hit = false;
    size_t   staged = 0;
    for (int b = 0; b < nb; b++) {
        dptr[b] = svthip::plane_cache_acquire(bufs[b], bid[b], bytes[b], &hit[b], &token[b]);
        if (!dptr[b]) staged += svthip::align_up(bytes[b], 256);
    }
    uint8_t* d_rec = (ids && ids->recon) ? svthip::plane_cache_acquire(recon_buf, ids->recon, rec_b, &rec_hit, &rec_token) : nullptr;
    svthip::HostCallLease lease;
    svthip::HostCall& c = *lease;
    c.begin();
    const size_t side = tot_b + mv_b + cand_b + cells * (sizeof(SvtHipTplSrcStats) + sizeof(SvtHipTplReconStats)) + 16384;
    size_t       pin_need = 2 * rec_b + 2 * side + 8192;
    for (int b = 0; b < nb; b++) pin_need += svthip::align_up(bytes[b], 256); // (a staged upload of a plane that is neither page-locked nor resident)
    c.reserve(staged + rec_b + side + 4096, pin_need);
    for (int b = 0; b < nb; b++) {
        if (!dptr[b]) dptr[b] = (uint8_t*)c.dalloc(bytes[b]);
        if (!hit[b]) c.up(dptr[b], bufs[b], bytes[b]);
    }
    uint8_t*             d_tot  = (uint8_t*)c.dalloc(tot_b);
    uint32_t*            d_mv   = (uint32_t*)c.dalloc(mv_b ? mv_b : 4);
    uint8_t*             d_cand = (uint8_t*)c.dalloc(cand_b ? cand_b : 4);
    SvtHipTplSrcStats*   d_ss   = (SvtHipTplSrcStats*)c.dalloc(cells * sizeof(SvtHipTplSrcStats));
    if (!d_rec) d_rec = (uint8_t*)c.dalloc(rec_b);
    SvtHipTplReconStats* d_out  = (SvtHipTplReconStats*)c.dalloc(cells * sizeof(SvtHipTplReconStats));
    if (stored) c.up(d_ss, src_stats, cells * sizeof(SvtHipTplSrcStats));
    else {
        c.up(d_tot, total_me_candidate_index, tot_b);
        if (mv_b) c.up(d_mv, me_mv_array, mv_b);
        if (cand_b) c.up(d_cand, me_candidate_array, cand_b);
        HIP_CHECK(hipMemsetAsync(d_ss, 0, cells * sizeof(SvtHipTplSrcStats), c.stream));
    }
    HIP_CHECK(hipMemsetAsync(d_out, 0, cells * sizeof(SvtHipTplReconStats), c.stream));
    HIP_CHECK(hipMemsetAsync(d_rec, 0, rec_b, c.stream)); // (never read before it is written: a DC block's neighbours are blocks of this picture; only the written rectangle comes back)
    // the kernels address every plane as base + offset: with planes in separate allocations the base is 0 and the offsets are the addresses
    const uint8_t* base0 = nullptr;
    P.src_off += (uint64_t)(uintptr_t)dptr[0];
    for (int r = 0; r < 8; r++) {
        if (src_slot[r] >= 0) P.refs[r].plane_off += (uint64_t)(uintptr_t)dptr[src_slot[r]];
        if (rec_slot[r] >= 0) R.rec_refs[r].plane_off += (uint64_t)(uintptr_t)dptr[rec_slot[r]];
    }
    if (!stored) svt_hip_tpl_src_stage(&P, base0, base0, d_tot, d_mv, d_cand, d_ss, c.stream);
    svt_hip_tpl_recon_stage(&R, base0, base0, d_ss, d_rec, d_out, c.stream);
    const size_t covered_w = ((size_t)P.width + 8) >> 4 << 4, covered_h = ((size_t)P.height + 8) >> 4 << 4;
    size_t       rows_down = covered_h;
    while (rows_down && R.recon_off + (rows_down - 1) * R.recon_stride + covered_w > rec_b) rows_down--;
    const size_t span = rows_down ? (rows_down - 1) * R.recon_stride + covered_w : 0;
    uint8_t*     pin_rec = span ? (uint8_t*)c.palloc(span) : nullptr;
    uint8_t*     pin_ss  = (uint8_t*)c.palloc(cells * sizeof(SvtHipTplSrcStats));
    uint8_t*     pin_out = (uint8_t*)c.palloc(cells * sizeof(SvtHipTplReconStats));
    if (span) HIP_CHECK(hipMemcpyAsync(pin_rec, d_rec + R.recon_off, span, hipMemcpyDeviceToHost, c.stream));
    if (!stored) HIP_CHECK(hipMemcpyAsync(pin_ss, d_ss, cells * sizeof(SvtHipTplSrcStats), hipMemcpyDeviceToHost, c.stream));
    HIP_CHECK(hipMemcpyAsync(pin_out, d_out, cells * sizeof(SvtHipTplReconStats), hipMemcpyDeviceToHost, c.stream));
    const bool keep_rec = rec_token >= 0 && ids->recon_width && ids->recon_height && covered_w >= ids->recon_width && covered_h >= ids->recon_height;
    if (keep_rec) // the device copy's borders, as svt_aom_generate_padding leaves the host's after the dispenser
        svt_hip_generate_padding(d_rec + R.recon_off - (size_t)ids->recon_org_y * R.recon_stride - ids->recon_org_x, R.recon_stride, ids->recon_width, ids->recon_height,
                                 ids->recon_org_x, ids->recon_org_y, c.stream);
    c.sync(); // the one synchronisation of the call
    for (size_t y = 0; y < rows_down; y++) memcpy(recon_buf + R.recon_off + y * R.recon_stride, pin_rec + y * R.recon_stride, covered_w);
    if (!stored) memcpy(src_stats, pin_ss, cells * sizeof(SvtHipTplSrcStats));
    memcpy(out, pin_out, cells * sizeof(SvtHipTplReconStats));
    int rc = 0;
    for (size_t r = 0; r < rows16; r++)
        if (out[r * cols16].pad[0] == 0xEE) rc = -4;
    for (int b = 0; b < nb; b++) svthip::plane_cache_release(token[b], rc == 0); // (uploaded planes are valid from here on; hits just lose their pin)
    svthip::plane_cache_release(rec_token, rc == 0 && keep_rec);
    return rc;
}
int svt_hip_tpl_stage_host(const SvtHipTplReconParams* params, const SvtHipTplHostPlanes* src_planes, const SvtHipTplHostPlanes* rec_planes,
                           const uint8_t* total_me_candidate_index, const uint32_t* me_mv_array, const uint8_t* me_candidate_array, SvtHipTplSrcStats* src_stats,
                           uint8_t* recon_buf, uint32_t recon_rows, SvtHipTplReconStats* out) {
    SVT_HIP_ENTRY_TRY
    if (!total_me_candidate_index) return -1;
    return tpl_stage_host_impl(params, src_planes, rec_planes, nullptr, total_me_candidate_index, me_mv_array, me_candidate_array, src_stats, recon_buf, recon_rows, out);
    SVT_HIP_ENTRY_CATCH(SVT_HIP_E_DEVICE)
}
int svt_hip_tpl_stage_host_resident(const SvtHipTplReconParams* params, const SvtHipTplHostPlanes* src_planes, const SvtHipTplHostPlanes* rec_planes,
                                    const SvtHipTplPlaneIds* ids, const uint8_t* total_me_candidate_index, const uint32_t* me_mv_array, const uint8_t* me_candidate_array,
                                    SvtHipTplSrcStats* src_stats, uint8_t* recon_buf, uint32_t recon_rows, SvtHipTplReconStats* out) {
    SVT_HIP_ENTRY_TRY
    return tpl_stage_host_impl(params, src_planes, rec_planes, ids, total_me_candidate_index, me_mv_array, me_candidate_array, src_stats, recon_buf, recon_rows, out);
    SVT_HIP_ENTRY_CATCH(SVT_HIP_E_DEVICE)
}
void svt_hip_tpl_plane_drop(const void* host_buffer) {
    SVT_HIP_ENTRY_TRY svthip::plane_cache_drop(host_buffer);     SVT_HIP_ENTRY_CATCH((void)0)
}
void svt_hip_tpl_plane_counts(uint64_t* hits, uint64_t* misses) {
    SVT_HIP_ENTRY_TRY svthip::ensure_device(); svthip::plane_cache_counts(hits, misses);     SVT_HIP_ENTRY_CATCH((void)0)
}

} // extern "C"

SVT_HIP_DEFINE_WARM(tpl) // (svt_hip_warmup loads this translation unit's code object at encoder initialisation: svt_hip_common.h)
