// The temporal filter of one central picture as a device stage (include/svtav1_hip.h: svt_hip_tf_picture_host): the glue between the three batched pieces that
// already exist -- svt_hip_tf_subpel_search_batch, svt_hip_tf_inter_pred_batch, svt_hip_tf_filter_frame -- i.e. what produce_temporally_filtered_pic
// (temporal_filtering.c:2782-3400) does per 64x64 block and reference between them: which blocks are searched from which vectors, the 64x64 / 32x32 / 16x16 / 8x8
// decision tree on the search results, the descriptors of the final motion compensation, the per-32x32 records the filter reads, and the 32x32 errors of a 64x64
// prediction.  Nothing returns to the host between the steps: the decisions are made by kernels, on device-resident results.  The sub-pel refinement runs in THREE
// passes -- 64x64, then the 32x32 blocks of the (block, reference) pairs the 64x64-only tests let through, then the 16x16 (and 8x8) blocks of the 32x32 blocks that
// need them -- so that, like the reference, a size is searched only where its result can be used: at preset 8 that is one search per pair instead of 21.
//
// Slots of a 64x64 block, everywhere in this file, follow the ME tables' order: 0 = 64x64; 1 + i32; 5 + 4 i32 + i16; 21 + 16 i32 + 4 i16 + i8 (z-order: index bit 0 =
// right half, bit 1 = lower half at every level) -- tab16x16 / tab8x8 (motion_estimation.h:101-116) and idx_32x32_to_idx_16x16 / _8x8 (temporal_filtering.c:60-80) are
// the two directions of that order.
#include "../../include/svtav1_hip.h"
#include "svt_hip_common.h"

namespace {

constexpr int MC_SLOTS = 64; // motion-compensation descriptors per (block, reference) at most: 1 (64x64) .. 64 (all 8x8) -- the capacity of the picture's list

struct PicArgs {
    SvtHipTfPictureParams P;
    uint32_t n_refs, n_sb, per_sb;
    uint64_t pic0;       // samples from a luma buffer's first sample to picture sample (0, 0)
    uint64_t ref_pitch;  // samples between consecutive references' luma buffers (descriptor offsets are relative to reference 0)
    uint64_t sp_ref_pitch; // ... of the planes the sub-pel searches read (the 8-bit luma copies with subpel_8bit, else = ref_pitch)
    uint64_t uv_pitch;   // ... chroma buffers
    uint64_t pred_y_pitch, pred_uv_pitch; // samples between consecutive references' prediction planes
    uint32_t pred_y_stride, pred_uv_stride;
    const uint32_t*           best_sad;   // [n_refs][n_sb][85]
    const uint32_t*           best_mv;
    const int16_t*            hme_sc;     // [n_refs][n_sb][2]
    const unsigned long long* hme_sad;    // [n_refs][n_sb]
    SvtHipTfSubpelDesc*       sp_descs;   // [n_refs][n_sb][per_sb]: three regions, one per pass -- [pair] | [pair][4] | [pair][per_sb - 5]
    SvtHipTfSubpelResult*     sp_res;     // same layout
    SvtHipTfMcDesc*           mc_descs;   // a list (capacity n_refs * n_sb * MC_SLOTS) the decision kernel appends to
    uint32_t*                 mc_count;   // its length
    SvtHipTfBlock*            blocks;     // [n_refs][2 pic_h_sb][2 pic_w_sb]
    uint8_t*                  path64;     // [n_refs][n_sb]
    SvtHipTfPictureStats*     stats;
};

__device__ __forceinline__ void slot_geometry(const int slot, int& bs, int& lx, int& ly) {
    if (slot == 0) { bs = 64; lx = ly = 0; }
    else if (slot < 5) { const int i = slot - 1; bs = 32; lx = (i & 1) * 32; ly = (i >> 1) * 32; }
    else if (slot < 21) { const int z = slot - 5; bs = 16; lx = ((z >> 2) & 1) * 32 + (z & 1) * 16; ly = ((z >> 3) & 1) * 32 + ((z >> 1) & 1) * 16; }
    else { const int z = slot - 21; bs = 8; lx = ((z >> 4) & 1) * 32 + ((z >> 2) & 1) * 16 + (z & 1) * 8; ly = ((z >> 5) & 1) * 32 + ((z >> 3) & 1) * 16 + ((z >> 1) & 1) * 8; }
}

// where the results of a pair live: slot 0 -> region 0, slots 1-4 -> region 1, slots 5.. -> region 2
__device__ __forceinline__ size_t res_index(const PicArgs& A, const uint32_t pair, const int slot) {
    const size_t np = (size_t)A.n_refs * A.n_sb;
    if (slot == 0) return pair;
    if (slot < 5) return np + (size_t)pair * 4 + (slot - 1);
    return np * 5 + (size_t)pair * (A.per_sb - 5) + (slot - 5);
}
// tf_use_64x64_pred (:2676-2690) on the pair's ME SADs, and the early exit of the ME call (motion_estimation.c:3110): both known before any search
__device__ __forceinline__ bool only_64x64_before_search(const PicArgs& A, const uint32_t pair) {
    const uint8_t th = A.hme_sad[pair] < A.P.me_exit_th ? (uint8_t)0xff : A.P.use_pred_64x64_only_th;
    if (!th) return false;
    if (th == 0xff) return true;
    const uint32_t* sd = A.best_sad + (size_t)pair * 85;
    uint32_t d32 = 0;
    for (int i = 0; i < 4; i++) d32 += sd[1 + i];
    const long long a = (long long)(sd[0] > 1u ? sd[0] : 1u), b = (long long)(d32 > 1u ? d32 : 1u);
    return (a - b) * 100 / b < (long long)th;
}
// (:3263-3270) the 64x64 prediction wins against the four 32x32
__device__ __forceinline__ bool pred_64x64_wins(const PicArgs& A, const uint32_t pair) {
    const unsigned long long e64 = A.sp_res[res_index(A, pair, 0)].dist;
    unsigned long long s32 = 0;
    for (int i = 0; i < 4; i++) s32 += A.sp_res[res_index(A, pair, 1 + i)].dist;
    return e64 * 14 < s32 * 16 && e64 < (1ull << 18);
}

// The descriptors of one pass (level 0: 64x64; 1: 32x32; 2: 16x16 and 8x8) with the starting vector the reference's caller would pass (:1866-1870, :1980-1982,
// :2113-2115, :2232-2234); a block the reference would not search at this point gets bsize 0 and the search kernel skips it.
__global__ __launch_bounds__(256) void tf_pic_descs_kernel(const PicArgs A, const int level) {
    const uint32_t per = level == 0 ? 1u : (level == 1 ? 4u : A.per_sb - 5u), first = level == 0 ? 0u : (level == 1 ? 1u : 5u);
    const uint32_t i = blockIdx.x * 256 + threadIdx.x, total = A.n_refs * A.n_sb * per;
    if (i >= total) return;
    const uint32_t pair = i / per, slot = first + (i - pair * per), ref = pair / A.n_sb, sb = pair - ref * A.n_sb;
    const uint32_t x0 = (sb % A.P.pic_w_sb) * 64, y0 = (sb / A.P.pic_w_sb) * 64;
    int bs, lx, ly;
    slot_geometry((int)slot, bs, lx, ly);
    bool wanted = true;
    if (level >= 1) wanted = !only_64x64_before_search(A, pair);
    if (level == 2 && wanted) {
        const int i32 = slot < 21 ? (int)(slot - 5) >> 2 : (int)(slot - 21) >> 4;
        wanted = !pred_64x64_wins(A, pair) && !(A.sp_res[res_index(A, pair, 1 + i32)].dist < A.P.pred_error_32x32_th); // (:3292-3296)
    }
    // the ME call's early exit leaves no tables: 64x64 only, from the HME centre (motion_estimation.c:3110; temporal_filtering.c:1866-1870)
    const bool from_sc = slot == 0 && (A.hme_sad[pair] < A.P.me_exit_th || A.P.use_pred_64x64_only_th == 0xff);
    const uint32_t mv = A.best_mv[(size_t)pair * 85 + slot];
    SvtHipTfSubpelDesc d;
    d.src_off    = A.pic0 + (uint64_t)(y0 + ly) * A.P.sp.ref_stride + x0 + lx;
    d.ref_off    = (uint64_t)ref * A.sp_ref_pitch;
    d.src_stride = A.P.sp.ref_stride;
    d.pu_x = (uint16_t)(x0 + lx); d.pu_y = (uint16_t)(y0 + ly);
    d.bsize = (uint8_t)(wanted ? bs : 0);
    d.bilinear = (uint8_t)(bs >= 32 ? A.P.use_2tap : 0); // (:1801-1804, :1911-1914; the 16x16 and 8x8 searches always take the regular kernels)
    d.mv_x = (int16_t)((from_sc ? A.hme_sc[2 * pair] : (int16_t)(mv & 0xffffu)) * 8);
    d.mv_y = (int16_t)((from_sc ? A.hme_sc[2 * pair + 1] : (int16_t)(mv >> 16)) * 8);
    d.pad = 0;
    A.sp_descs[res_index(A, pair, (int)slot)] = d;
}

// the decision tree of :3183-3340 for one (block, reference); writes the motion-compensation descriptors and the four 32x32 records of the filter
__global__ __launch_bounds__(64) void tf_pic_decide_kernel(const PicArgs A) {
    const uint32_t pair = blockIdx.x * 64 + threadIdx.x;
    if (pair >= A.n_refs * A.n_sb) return;
    const uint32_t ref = pair / A.n_sb, sb = pair - ref * A.n_sb, sbx = sb % A.P.pic_w_sb, sby = sb / A.P.pic_w_sb, x0 = sbx * 64, y0 = sby * 64;
    auto R = [&](const int slot) -> const SvtHipTfSubpelResult& { return A.sp_res[res_index(A, pair, slot)]; }; // (only slots the passes searched are read)
    const bool zm     = A.P.zero_motion != 0; // the low-delay form: nothing was searched, every block is one 64x64 prediction at vector (0, 0)
    const bool exited = !zm && A.hme_sad[pair] < A.P.me_exit_th;
    const bool p64    = zm || only_64x64_before_search(A, pair) || pred_64x64_wins(A, pair);
    const int16_t mv64x = zm ? (int16_t)0 : R(0).mv_x, mv64y = zm ? (int16_t)0 : R(0).mv_y;
    A.path64[pair] = p64 ? 1 : 0;
    // The decisions run twice: a first pass only COUNTS the prediction blocks of this (reference, SB), one atomic reserves their places in the picture's list, the second pass
    // writes them -- instead of one returning atomic per entry, up to 21 (85 with 8x8 blocks) round trips to the memory side in series per thread.
    uint32_t mc_n = 0, mc_at = 0;
    bool     emit = false;
    auto mc = [&](const int, const int slot, const int16_t mvx, const int16_t mvy) { // appends to the picture's list: the order of the entries does not matter
        if (!emit) { mc_n++; return; }
        int bs, lx, ly;
        slot_geometry(slot, bs, lx, ly);
        SvtHipTfMcDesc d;
        d.ref_off[0] = (uint64_t)ref * A.ref_pitch; d.ref_off[1] = d.ref_off[2] = (uint64_t)ref * A.uv_pitch;
        d.pred_off[0] = (uint64_t)ref * A.pred_y_pitch; d.pred_off[1] = d.pred_off[2] = (uint64_t)ref * A.pred_uv_pitch;
        d.pu_x = (uint16_t)(x0 + lx); d.pu_y = (uint16_t)(y0 + ly); d.bsize = (uint8_t)bs; d.pad = 0; d.mv_x = mvx; d.mv_y = mvy;
        d.pad2[0] = d.pad2[1] = d.pad2[2] = 0;
        A.mc_descs[mc_at++] = d;
    };
    const uint32_t nbx = 2 * A.P.pic_w_sb, nby = 2 * A.P.pic_h_sb;
    uint32_t n64 = 0, n32 = 0, n16 = 0, n8 = 0;
  for (int pass = 0; pass < 2; pass++) {
    emit = pass == 1;
    if (emit) mc_at = mc_n ? atomicAdd(A.mc_count, mc_n) : 0u;
    n64 = n32 = n16 = n8 = 0;
    if (p64) { mc(0, 0, mv64x, mv64y); n64 = 1; }
    for (int i32 = 0; i32 < 4; i32++) {
        SvtHipTfBlock B;
        for (int k = 0; k < 4; k++) { B.block_error[k] = 0; B.mv_x[k] = 0; B.mv_y[k] = 0; }
        B.split = 0;
        for (int k = 0; k < 7; k++) B.pad[k] = 0;
        if (p64) { // convert_64x64_info_to_32x32_info (:2691-2758): the 64x64 vector; the error comes from tf_pic_var32_kernel
            B.mv_x[0] = mv64x; B.mv_y[0] = mv64y;
        } else if (R(1 + i32).dist < A.P.pred_error_32x32_th) { // (:3292-3296)
            B.block_error[0] = R(1 + i32).dist; B.mv_x[0] = R(1 + i32).mv_x; B.mv_y[0] = R(1 + i32).mv_y;
            mc(16 * i32, 1 + i32, R(1 + i32).mv_x, R(1 + i32).mv_y); n32++;
        } else { // derive_tf_32x32_block_split_flag (:237-286), int arithmetic as there
            int  sum = 0, sub[4];
            bool split16[4];
            for (int i = 0; i < 4; i++) {
                sub[i]     = (int)R(5 + 4 * i32 + i).dist;
                split16[i] = false;
                if (A.P.enable_8x8_pred) {
                    int e8 = 0;
                    for (int j = 0; j < 4; j++) e8 += (int)R(21 + 16 * i32 + 4 * i + j).dist;
                    if (!(sub[i] * 8 < e8 * 16)) { split16[i] = true; sub[i] = e8; }
                }
                sum += sub[i];
            }
            const int  e32   = (int)R(1 + i32).dist;
            const bool split = !(e32 * 14 < sum * 16);
            if (!split) {
                B.block_error[0] = R(1 + i32).dist; B.mv_x[0] = R(1 + i32).mv_x; B.mv_y[0] = R(1 + i32).mv_y;
                mc(16 * i32, 1 + i32, R(1 + i32).mv_x, R(1 + i32).mv_y); n32++;
            } else {
                B.split = 1;
                for (int i = 0; i < 4; i++) {
                    const SvtHipTfSubpelResult r16 = R(5 + 4 * i32 + i);
                    B.block_error[i] = split16[i] ? (unsigned long long)(long long)sub[i] : r16.dist; // (a split 16x16 carries the sum of its 8x8 errors, :262-264)
                    B.mv_x[i] = r16.mv_x; B.mv_y[i] = r16.mv_y;
                    if (split16[i]) {
                        for (int j = 0; j < 4; j++) { const SvtHipTfSubpelResult r8 = R(21 + 16 * i32 + 4 * i + j); mc(16 * i32 + 4 * i + j, 21 + 16 * i32 + 4 * i + j, r8.mv_x, r8.mv_y); }
                        n8 += 4;
                    } else { mc(16 * i32 + 4 * i, 5 + 4 * i32 + i, r16.mv_x, r16.mv_y); n16++; }
                }
            }
        }
        if (emit) A.blocks[((size_t)ref * nby + 2 * sby + (i32 >> 1)) * nbx + 2 * sbx + (i32 & 1)] = B;
    }
  }
    if (A.stats) {
        if (n64) atomicAdd(&A.stats->blocks_64x64, n64);
        if (n32) atomicAdd(&A.stats->blocks_32x32, n32);
        if (n16) atomicAdd(&A.stats->blocks_16x16, n16);
        if (n8) atomicAdd(&A.stats->blocks_8x8, n8);
        if (exited) atomicAdd(&A.stats->early_exit_blocks, 1u);
    }
}

// convert_64x64_info_to_32x32_info's block errors (:2718-2756): the variance of the 64x64 prediction against the source per 32x32 block, on every
// (1 << subsampling_shift)-th row, << subsampling_shift (svt_aom_mefn_ptr[BLOCK_32X32 / 32X16].vf / vf_hbd_10).  One wave per (block, reference, 32x32).
template <typename PIX>
__global__ __launch_bounds__(256) void tf_pic_var32_kernel(const PicArgs A, const PIX* __restrict__ central_y, const PIX* __restrict__ pred_y) {
    const int      l = threadIdx.x & 63;
    const uint32_t item = blockIdx.x * 4 + (threadIdx.x >> 6), pair = item >> 2, i32 = item & 3;
    if (pair >= A.n_refs * A.n_sb || !A.path64[pair]) return;
    const uint32_t ref = pair / A.n_sb, sb = pair - ref * A.n_sb, sbx = sb % A.P.pic_w_sb, sby = sb / A.P.pic_w_sb;
    const uint32_t x0 = sbx * 64 + (i32 & 1) * 32, y0 = sby * 64 + (i32 >> 1) * 32;
    const int      ss = A.P.sp.subsampling_shift, rows = 32 >> ss;
    const PIX*     s  = central_y + A.pic0 + (size_t)y0 * A.P.sp.ref_stride + x0;
    const PIX*     p  = pred_y + (size_t)ref * A.pred_y_pitch + (size_t)y0 * A.pred_y_stride + x0;
    int      sum = 0;
    uint32_t sse = 0;
    for (int i = l; i < 32 * rows; i += 64) {
        const int r = (i >> 5) << ss, c = i & 31;
        const int d = (int)p[(size_t)r * A.pred_y_stride + c] - (int)s[(size_t)r * A.P.sp.ref_stride + c];
        sum += d;
        sse += (uint32_t)(d * d); // per lane <= 16 * 1023^2
    }
    long long          ts = sum;
    unsigned long long tq = sse;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        ts += (long long)__shfl_xor((int)ts, m); // |total| <= 1024 * 1023: fits 32 bits
        const unsigned long long o = ((unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)(tq >> 32), m) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)tq, m);
        tq += o;
    }
    const int ln = 10 - ss; // log2(32 * rows)
    unsigned long long var;
    if (sizeof(PIX) == 1) {
        const int su = (int)ts;
        var = (uint32_t)((uint32_t)tq - (uint32_t)(((long long)su * su) >> ln)); // svt_aom_varianceWxH_c (variance.c:300-306)
    } else { // highbd_10_variance (svt_psnr.c:160-177)
        const uint32_t  s32 = (uint32_t)((tq + 8) >> 4);
        const int       su  = (int)((ts + 2) >> 2);
        const long long v   = (long long)s32 - (((long long)su * su) >> ln);
        var = v >= 0 ? (uint32_t)v : 0;
    }
    if (l == 0) {
        const uint32_t nbx = 2 * A.P.pic_w_sb, nby = 2 * A.P.pic_h_sb;
        A.blocks[((size_t)ref * nby + 2 * sby + (i32 >> 1)) * nbx + 2 * sbx + (i32 & 1)].block_error[0] = var << ss;
    }
}

} // namespace

namespace {
struct PicSizes {
    size_t px, n_sb, per_sb, n_pairs, n_sp, nblk, pysz, pcsz, state_bytes;
    uint32_t pw, ph;
};
PicSizes pic_sizes(const SvtHipTfPictureParams& P, uint32_t n_refs) {
    PicSizes z;
    z.px = P.sp.bit_depth > 8 ? 2 : 1;
    z.n_sb = (size_t)P.pic_w_sb * P.pic_h_sb; z.per_sb = 21 + (P.enable_8x8_pred ? 64 : 0);
    z.pw = 64 * P.pic_w_sb; z.ph = 64 * P.pic_h_sb;
    z.pysz = svthip::align_up((size_t)z.pw * z.ph * z.px, 256); z.pcsz = svthip::align_up((size_t)(z.pw / 2) * (z.ph / 2) * z.px, 256);
    z.n_pairs = n_refs * z.n_sb; z.n_sp = z.n_pairs * z.per_sb; z.nblk = (size_t)n_refs * 4 * z.n_sb;
    SvtHipTfParams TW = P.tf;
    z.state_bytes = n_refs > SVT_HIP_TF_MAX_REFS ? svt_hip_tf_filter_frame_workspace(&TW, 2 * P.pic_w_sb, 2 * P.pic_h_sb) : 0;
    return z;
}
bool pic_params_ok(const SvtHipTfPictureParams& P, uint32_t n_refs) {
    if (n_refs == 0 || n_refs > SVT_HIP_TF_MAX_FRAMES || !P.pic_w_sb || !P.pic_h_sb) return false;
    if (P.tf.tf_chroma && (P.tf.ss_x != 1 || P.tf.ss_y != 1)) return false; // (the final motion compensation is built for 4:2:0)
    return P.sp.bit_depth == 8 || P.sp.bit_depth == 10;
}
} // namespace

extern "C" size_t svt_hip_tf_picture_workspace(const SvtHipTfPictureParams* params, uint32_t n_refs) {
    if (!pic_params_ok(*params, n_refs)) return 0;
    const PicSizes z = pic_sizes(*params, n_refs);
    return z.state_bytes + n_refs * (z.pysz + 2 * z.pcsz) + z.n_sp * (sizeof(SvtHipTfSubpelDesc) + sizeof(SvtHipTfSubpelResult)) + z.n_pairs * (MC_SLOTS * sizeof(SvtHipTfMcDesc) + 1) +
           z.nblk * sizeof(SvtHipTfBlock) + sizeof(SvtHipTfPictureStats) + 16 * 256;
}

// Device-resident form: the pictures and the pairs' ME tables are in HBM already (e.g. left there by the ME stage), the filtered 64x64 blocks replace the central picture
// in place, nothing touches the host.
extern "C" int svt_hip_tf_picture(const SvtHipTfPictureParams* params, const SvtHipTfDevicePictures* pics, const SvtHipTfMeTables* me, uint32_t n_refs, void* workspace,
                                  SvtHipTfPictureStats* stats_dev, void* stream) {
    svthip::ensure_device();
    const SvtHipTfPictureParams& P = *params;
    if (!pic_params_ok(P, n_refs)) return -1;
    const bool hbd = P.sp.bit_depth > 8, chroma = P.tf.tf_chroma != 0, sp8 = hbd && P.subpel_8bit;
    if (sp8 && (!pics->central_y8 || !pics->refs_y8)) return -1;
    const PicSizes z = pic_sizes(P, n_refs);
    const size_t px = z.px, n_pairs = z.n_pairs;
    uint8_t* w = (uint8_t*)workspace;
    auto take = [&](size_t bytes) { uint8_t* p = w; w += svthip::align_up(bytes, 256); return p; };
    uint8_t *d_cy = (uint8_t*)pics->central[0], *d_cu = (uint8_t*)pics->central[1], *d_cv = (uint8_t*)pics->central[2];
    uint8_t *d_ry = (uint8_t*)pics->refs[0], *d_ru = (uint8_t*)pics->refs[1], *d_rv = (uint8_t*)pics->refs[2];
    uint8_t* d_py = take(z.pysz * n_refs);
    uint8_t* d_pu = take(z.pcsz * n_refs);
    uint8_t* d_pv = take(z.pcsz * n_refs);
    PicArgs A;
    memset(&A, 0, sizeof(A));
    A.P = P; A.n_refs = n_refs; A.n_sb = (uint32_t)z.n_sb; A.per_sb = (uint32_t)z.per_sb;
    A.pic0 = (uint64_t)P.sp.ref_org_y * P.sp.ref_stride + P.sp.ref_org_x;
    A.ref_pitch = pics->ref_pitch; A.sp_ref_pitch = sp8 ? pics->ref_y8_pitch : pics->ref_pitch; A.uv_pitch = pics->ref_uv_pitch;
    A.pred_y_pitch = z.pysz / px; A.pred_uv_pitch = z.pcsz / px; A.pred_y_stride = z.pw; A.pred_uv_stride = z.pw / 2;
    if (!P.zero_motion) { A.best_sad = me->best_sad; A.best_mv = me->best_mv; A.hme_sc = me->hme_sc; A.hme_sad = (const unsigned long long*)me->hme_sad; }
    A.sp_descs = (SvtHipTfSubpelDesc*)take(z.n_sp * sizeof(SvtHipTfSubpelDesc));
    A.sp_res   = (SvtHipTfSubpelResult*)take(z.n_sp * sizeof(SvtHipTfSubpelResult));
    A.mc_descs = (SvtHipTfMcDesc*)take(n_pairs * MC_SLOTS * sizeof(SvtHipTfMcDesc));
    A.blocks   = (SvtHipTfBlock*)take(z.nblk * sizeof(SvtHipTfBlock));
    A.path64   = take(n_pairs);
    SvtHipTfPictureStats* own_stats = (SvtHipTfPictureStats*)take(sizeof(SvtHipTfPictureStats));
    A.stats    = stats_dev ? stats_dev : own_stats;
    void* d_state = z.state_bytes ? take(z.state_bytes) : nullptr; // the filter's accumulators between its launches (more than 12 frames)
    hipStream_t st = (hipStream_t)stream;
    A.mc_count = (uint32_t*)take(256);
    HIP_CHECK(hipMemsetAsync(A.mc_count, 0, 4, st));
    HIP_CHECK(hipMemsetAsync(A.stats, 0, sizeof(SvtHipTfPictureStats), st));
    // 1. sub-pel refinement in three passes: a size is searched only where the reference would search it
    SvtHipTfSubpelParams SP = P.sp;
    if (sp8) SP.bit_depth = 8;
    for (int level = 0; level < 3 && !P.zero_motion; level++) {
        const size_t per = level == 0 ? 1 : (level == 1 ? 4 : z.per_sb - 5), first = level == 0 ? 0 : (level == 1 ? n_pairs : 5 * n_pairs), cnt = n_pairs * per;
        hipLaunchKernelGGL(tf_pic_descs_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, A, level);
        SVT_LAUNCH_CHECK();
        svt_hip_tf_subpel_search_batch(&SP, sp8 ? pics->central_y8 : (const void*)d_cy, sp8 ? pics->refs_y8 : (const void*)d_ry, A.sp_descs + first, (uint32_t)cnt, A.sp_res + first, st);
    }
    // 2. decisions
    hipLaunchKernelGGL(tf_pic_decide_kernel, dim3((unsigned)((n_pairs + 63) / 64)), dim3(64), 0, st, A);
    SVT_LAUNCH_CHECK();
    // 3. final motion compensation into picture-sized planes
    SvtHipTfMcPlanes PL;
    memset(&PL, 0, sizeof(PL));
    PL.ref[0] = d_ry; PL.ref[1] = d_ru; PL.ref[2] = d_rv; PL.pred[0] = d_py; PL.pred[1] = d_pu; PL.pred[2] = d_pv;
    PL.ref_stride[0] = P.sp.ref_stride; PL.ref_stride[1] = PL.ref_stride[2] = P.uv_stride;
    PL.pred_stride[0] = z.pw; PL.pred_stride[1] = PL.pred_stride[2] = z.pw / 2;
    svt_hip_tf_inter_pred_list(&P.sp, &PL, A.mc_descs, (uint32_t)(n_pairs * MC_SLOTS), A.mc_count, chroma ? 1 : 0, st);
    // 4. the 32x32 errors of the 64x64 predictions
    if (hbd) hipLaunchKernelGGL(HIP_KERNEL_NAME(tf_pic_var32_kernel<uint16_t>), dim3((unsigned)n_pairs), dim3(256), 0, st, A, (const uint16_t*)d_cy, (const uint16_t*)d_py);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(tf_pic_var32_kernel<uint8_t>), dim3((unsigned)n_pairs), dim3(256), 0, st, A, (const uint8_t*)d_cy, (const uint8_t*)d_py);
    SVT_LAUNCH_CHECK();
    // 5. the filter, in place on the central picture
    SvtHipTfParams T = P.tf;
    T.encoder_bit_depth = P.sp.bit_depth;
    const size_t cpic0 = (size_t)(P.sp.ref_org_y >> 1) * P.uv_stride + (P.sp.ref_org_x >> 1);
    SvtHipTfPlanes cen = {d_cy + A.pic0 * px, d_cu ? d_cu + cpic0 * px : nullptr, d_cv ? d_cv + cpic0 * px : nullptr, P.sp.ref_stride, P.uv_stride};
    SvtHipTfPlanes preds[SVT_HIP_TF_MAX_FRAMES];
    for (uint32_t r = 0; r < n_refs; r++) preds[r] = SvtHipTfPlanes{d_py + r * z.pysz, d_pu + r * z.pcsz, d_pv + r * z.pcsz, z.pw, z.pw / 2};
    svt_hip_tf_filter_frame_chunked(&T, &cen, preds, n_refs, A.blocks, 2 * P.pic_w_sb, 2 * P.pic_h_sb, &cen, d_state, st);
    return 0;
}

extern "C" int svt_hip_tf_picture_host(const SvtHipTfPictureParams* params, const SvtHipTfHostPicture* central, const SvtHipTfHostPicture* refs, const SvtHipTfMeTables* me,
                                       uint32_t n_refs, void* out_y, void* out_u, void* out_v, SvtHipTfPictureStats* stats) {
    SVT_HIP_ENTRY_TRY
    svthip::ensure_device();
    const SvtHipTfPictureParams& P = *params;
    if (!pic_params_ok(P, n_refs)) return -1;
    const bool   hbd = P.sp.bit_depth > 8, chroma = P.tf.tf_chroma != 0, sp8 = hbd && P.subpel_8bit;
    if (sp8 && !central->y8) return -1;
    const PicSizes z = pic_sizes(P, n_refs);
    const size_t px = z.px, n_sb = z.n_sb, n_pairs = z.n_pairs;
    const size_t ysz = svthip::align_up(central->y_samples * px, 256), csz = svthip::align_up(central->uv_samples * px, 256);
    const size_t tables = n_pairs * (85 * 4 * 2 + 4 + 8) + 4 * 256;
    for (uint32_t r = 0; r < n_refs; r++)
        if (refs[r].y_samples != central->y_samples || refs[r].uv_samples != central->uv_samples || (sp8 && !refs[r].y8)) return -1;
    const size_t y8sz = sp8 ? svthip::align_up(central->y_samples, 256) : 0;
    const size_t wsb  = svt_hip_tf_picture_workspace(params, n_refs);
    const size_t dev = wsb + (1 + n_refs) * (ysz + 2 * csz + y8sz) + tables + 65536;
    const size_t pin = (2 + n_refs) * (ysz + 2 * csz + y8sz) + tables + 65536; // uploads + the three downloads
    svthip::HostCallLease lease; // (a pooled arena: see svt_hip_common.h)
    svthip::HostCall& c = *lease;
    c.begin();
    c.reserve(dev, pin);
    // pictures: central, then the references back to back (descriptor offsets are relative to reference 0)
    SvtHipTfDevicePictures D;
    memset(&D, 0, sizeof(D));
    uint8_t* d_cy = (uint8_t*)c.dalloc(ysz);
    uint8_t* d_cu = (uint8_t*)c.dalloc(csz);
    uint8_t* d_cv = (uint8_t*)c.dalloc(csz);
    uint8_t* d_ry = (uint8_t*)c.dalloc(ysz * n_refs);
    uint8_t* d_ru = (uint8_t*)c.dalloc(csz * n_refs);
    uint8_t* d_rv = (uint8_t*)c.dalloc(csz * n_refs);
    uint8_t* d_c8 = sp8 ? (uint8_t*)c.dalloc(y8sz) : nullptr;
    uint8_t* d_r8 = sp8 ? (uint8_t*)c.dalloc(y8sz * n_refs) : nullptr;
    if (sp8) {
        c.up(d_c8, central->y8, central->y_samples);
        for (uint32_t r = 0; r < n_refs; r++) c.up(d_r8 + r * y8sz, refs[r].y8, refs[r].y_samples);
    }
    c.up(d_cy, central->y, central->y_samples * px);
    if (chroma) { c.up(d_cu, central->u, central->uv_samples * px); c.up(d_cv, central->v, central->uv_samples * px); }
    for (uint32_t r = 0; r < n_refs; r++) {
        c.up(d_ry + r * ysz, refs[r].y, refs[r].y_samples * px);
        if (chroma) { c.up(d_ru + r * csz, refs[r].u, refs[r].uv_samples * px); c.up(d_rv + r * csz, refs[r].v, refs[r].uv_samples * px); }
    }
    D.central[0] = d_cy; D.central[1] = d_cu; D.central[2] = d_cv; D.refs[0] = d_ry; D.refs[1] = d_ru; D.refs[2] = d_rv;
    D.ref_pitch = ysz / px; D.ref_uv_pitch = csz / px; D.central_y8 = d_c8; D.refs_y8 = d_r8; D.ref_y8_pitch = y8sz;
    uint32_t* d_sad = (uint32_t*)c.dalloc(n_pairs * 85 * 4);
    uint32_t* d_mv  = (uint32_t*)c.dalloc(n_pairs * 85 * 4);
    int16_t*  d_sc  = (int16_t*)c.dalloc(n_pairs * 4);
    uint64_t* d_hs  = (uint64_t*)c.dalloc(n_pairs * 8);
    for (uint32_t r = 0; r < n_refs && !P.zero_motion; r++) {
        c.up(d_sad + r * n_sb * 85, me[r].best_sad, n_sb * 85 * 4);
        c.up(d_mv + r * n_sb * 85, me[r].best_mv, n_sb * 85 * 4);
        c.up(d_sc + r * n_sb * 2, me[r].hme_sc, n_sb * 4);
        c.up(d_hs + r * n_sb, me[r].hme_sad, n_sb * 8);
    }
    SvtHipTfMeTables M = {d_sad, d_mv, d_sc, d_hs};
    SvtHipTfPictureStats* d_stats = (SvtHipTfPictureStats*)c.dalloc(sizeof(SvtHipTfPictureStats));
    void* d_ws = c.dalloc(wsb);
    const int rc = svt_hip_tf_picture(params, &D, &M, n_refs, d_ws, d_stats, c.stream);
    if (rc) { c.sync(); return rc; }
    // the filtered blocks (every 64x64 block in full, as get_final_filtered_pixels writes them, :2608-2672)
    const size_t pic0 = (size_t)P.sp.ref_org_y * P.sp.ref_stride + P.sp.ref_org_x, cpic0 = (size_t)(P.sp.ref_org_y >> 1) * P.uv_stride + (P.sp.ref_org_x >> 1);
    c.down2d_later((uint8_t*)out_y + pic0 * px, (size_t)P.sp.ref_stride * px, d_cy + pic0 * px, (size_t)P.sp.ref_stride * px, (size_t)z.pw * px, z.ph);
    if (chroma) {
        c.down2d_later((uint8_t*)out_u + cpic0 * px, (size_t)P.uv_stride * px, d_cu + cpic0 * px, (size_t)P.uv_stride * px, (size_t)(z.pw / 2) * px, z.ph / 2);
        c.down2d_later((uint8_t*)out_v + cpic0 * px, (size_t)P.uv_stride * px, d_cv + cpic0 * px, (size_t)P.uv_stride * px, (size_t)(z.pw / 2) * px, z.ph / 2);
    }
    if (stats) c.down_later(stats, d_stats, sizeof(SvtHipTfPictureStats));
    c.finish(); // (one synchronisation for the three planes and the statistics)
    return 0;
    SVT_HIP_ENTRY_CATCH(SVT_HIP_E_DEVICE)
}

SVT_HIP_DEFINE_WARM(tf_picture) // (svt_hip_warmup loads this translation unit's code object at encoder initialisation: svt_hip_common.h)
