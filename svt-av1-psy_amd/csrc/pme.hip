// pme.hip -- svt_pme_sad_loop_kernel: SAD + motion-vector rate search of MD's predictive ME
// (svt_pme_sad_loop_kernel_c, product_coding_loop.c:1900-1951; cost model svt_mv_err_cost, mcomp.c:44-68).
//
// The reference walks the search area in groups of eight consecutive x positions that are `search_step` apart (groups that do not
// fit are skipped) and every `search_step`-th row; a candidate replaces the incumbent only on a strictly smaller cost.  Here one
// wave evaluates one visited position (v_sad_u8 over the block, DPP wave reduction), adds the rate term and competes with a 64-bit
// atomicMin key (cost << 32 | visiting order), which reproduces "first strictly smaller in visiting order".
#include "svt_hip_common.h"
#include "../../include/svtav1_hip.h"

#include <vector>

namespace {

struct PmeParams {
    int      bw, bh, src_pitch, ref_pitch, npos, ngx;
    int      cost_type, error_per_bit; // MV_COST_TYPE (mcomp.h:29-36)
    int      joint_cost[4];
};

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += (uint32_t)__shfl_xor((int)v, m);
    return v;
}

// xs[i] / ys[j]: visited x / y offsets; rowterm[j] / colterm[i]: per-row / per-column rate inputs prepared by the caller from the
// MV_COST_PARAMS tables (ENTROPY: comp_cost[0][clip(diff.row)], comp_cost[1][clip(diff.col)]; L1 / OPT: |diff.row|, |diff.col|);
// rowzero / colzero: diff component == 0 (mv joint, rd_cost.c:55-60).
__global__ __launch_bounds__(256) void pme_sad_loop_kernel(const uint8_t* __restrict__ src, const uint8_t* __restrict__ ref, const PmeParams P,
                                                           const int16_t* __restrict__ xs, const int16_t* __restrict__ ys,
                                                           const int32_t* __restrict__ rowterm, const int32_t* __restrict__ colterm,
                                                           const uint8_t* __restrict__ rowzero, const uint8_t* __restrict__ colzero,
                                                           unsigned long long* __restrict__ key) {
    const int l = threadIdx.x & 63;
    const int nwaves = gridDim.x * 4;
    for (int p = blockIdx.x * 4 + (threadIdx.x >> 6); p < P.npos; p += nwaves) {
        const int iy = p / P.ngx, ix = p - iy * P.ngx;
        const uint8_t* r = ref + (size_t)ys[iy] * P.ref_pitch + xs[ix];
        uint32_t sad = 0;
        if ((P.bw & 3) == 0) {
            const int cpr = P.bw >> 2, total = cpr * P.bh;
            for (int i = l; i < total; i += 64) {
                const int y = i / cpr, c = i - y * cpr;
                const uint32_t a = *(const uint32_t*)(src + (size_t)y * P.src_pitch + c * 4); // staged copy: pitch multiple of 16
                uint32_t b;
                __builtin_memcpy(&b, r + (size_t)y * P.ref_pitch + c * 4, 4); // any byte alignment
                sad = __builtin_amdgcn_sad_u8(a, b, sad);
            }
        } else {
            const int total = P.bw * P.bh;
            for (int i = l; i < total; i += 64) {
                const int y = i / P.bw, x = i - y * P.bw;
                const int a = src[(size_t)y * P.src_pitch + x], b = r[(size_t)y * P.ref_pitch + x];
                sad += (uint32_t)(a > b ? a - b : b - a);
            }
        }
        sad = wave_sum_u32(sad);
        if (l == 0) {
            long long rate = 0;
            switch (P.cost_type) {
            case 0: { // MV_COST_ENTROPY: ROUND_POWER_OF_TWO_64(mv_cost * error_per_bit, RDDIV_BITS + AV1_PROB_COST_SHIFT - RD_EPB_SHIFT + 4 = 14)
                const int joint = (rowzero[iy] ? 0 : 2) | (colzero[ix] ? 0 : 1); // MV_JOINT_ZERO, HNZVZ, HZVNZ, HNZVNZ
                const long long v = (long long)(P.joint_cost[joint] + rowterm[iy] + colterm[ix]) * P.error_per_bit;
                rate = (v + (1ll << 13)) >> 14;
                break;
            }
            case 1: rate = (2 * (rowterm[iy] + colterm[ix])) >> 3; break; // MV_COST_L1_LOWRES (SSE_LAMBDA_LOWRES 2)
            case 2: rate = 0; break;                                       // MV_COST_L1_MIDRES (lambda 0)
            case 3: rate = (rowterm[iy] + colterm[ix]) >> 3; break;        // MV_COST_L1_HDRES (lambda 1)
            case 4: {                                                      // MV_COST_OPT
                const long long v = (long long)((rowterm[iy] + colterm[ix]) << 8) * P.error_per_bit;
                rate = (v + (1ll << 13)) >> 14;
                break;
            }
            default: rate = 0; // MV_COST_NONE
            }
            const uint32_t cost = sad + (uint32_t)(int)rate; // the reference adds the int rate to a uint32_t cost
            atomicMin(key, ((unsigned long long)cost << 32) | (unsigned long long)(uint32_t)p);
        }
    }
}

} // namespace

extern "C" {

void svt_pme_sad_loop_kernel_hip(const SvtHipMvCostParams* mv_cost_params, uint8_t* src, uint32_t src_stride, uint8_t* ref, uint32_t ref_stride,
                                 uint32_t block_height, uint32_t block_width, uint32_t* best_cost, int16_t* best_mvx, int16_t* best_mvy,
                                 int16_t search_position_start_x, int16_t search_position_start_y, int16_t search_area_width,
                                 int16_t search_area_height, int16_t search_step, int16_t mvx, int16_t mvy) {
    if (search_area_width < 8 || search_area_height <= 0 || search_step <= 0) return;
    // visited offsets: groups of 8 consecutive x starting every (7 + search_step), complete groups only; every search_step-th row
    std::vector<int16_t> xs, ys;
    for (int x0 = 0; x0 + 8 <= search_area_width; x0 += 7 + search_step)
        for (int k = 0; k < 8; k++) xs.push_back((int16_t)(x0 + k));
    for (int y = 0; y < search_area_height; y += search_step) ys.push_back((int16_t)y);
    const int ngx = (int)xs.size(), ngy = (int)ys.size(), npos = ngx * ngy;
    // rate inputs per visited row / column, exactly the table entries (or magnitudes) svt_mv_err_cost would read
    const SvtHipMvCostParams& M = *mv_cost_params;
    const int type = M.mv_cost_type;
    std::vector<int32_t> rowterm(ngy), colterm(ngx);
    std::vector<uint8_t> rowzero(ngy), colzero(ngx);
    const bool tables = type == 0 && M.mvcost[0] && M.mvcost[1] && M.mvjcost;
    auto clip_mv = [](int v) { return v < -(1 << 14) ? -(1 << 14) : (v > (1 << 14) ? (1 << 14) : v); };
    for (int j = 0; j < ngy; j++) {
        const int16_t mvr = (int16_t)(mvy + (int)((uint32_t)(search_position_start_y + ys[j]) * 8u));
        const int16_t d   = (int16_t)(mvr - M.ref_mv->row);
        rowzero[j] = d == 0;
        rowterm[j] = type == 0 ? (tables ? M.mvcost[0][clip_mv(d)] : 0) : (d < 0 ? -d : d);
    }
    for (int i = 0; i < ngx; i++) {
        const int16_t mvc = (int16_t)(mvx + (int)((uint32_t)(search_position_start_x + xs[i]) * 8u));
        const int16_t d   = (int16_t)(mvc - M.ref_mv->col);
        colzero[i] = d == 0;
        colterm[i] = type == 0 ? (tables ? M.mvcost[1][clip_mv(d)] : 0) : (d < 0 ? -d : d);
    }
    svthip::HostCall& c = svthip::host_call();
    c.begin();
    const size_t sp = svthip::align_up(block_width, 16);
    const size_t ww = (size_t)block_width + xs.back(), rp = svthip::align_up(ww, 16);
    const size_t lines = (size_t)ys.back() + block_height;
    const size_t small = (size_t)(ngx + ngy) * 8 + 4096;
    c.reserve(sp * block_height + rp * lines + small, sp * block_height + rp * lines + small);
    uint8_t* ds = (uint8_t*)c.dalloc(sp * block_height);
    uint8_t* dr = (uint8_t*)c.dalloc(rp * lines);
    int16_t* dxs = (int16_t*)c.dalloc(ngx * 2); int16_t* dys = (int16_t*)c.dalloc(ngy * 2);
    int32_t* drt = (int32_t*)c.dalloc(ngy * 4); int32_t* dct = (int32_t*)c.dalloc(ngx * 4);
    uint8_t* drz = (uint8_t*)c.dalloc(ngy);     uint8_t* dcz = (uint8_t*)c.dalloc(ngx);
    unsigned long long* dk = (unsigned long long*)c.dalloc(8);
    c.up2d(ds, sp, src, src_stride, block_width, block_height);
    c.up2d(dr, rp, ref, ref_stride, ww, lines);
    c.up(dxs, xs.data(), ngx * 2); c.up(dys, ys.data(), ngy * 2);
    c.up(drt, rowterm.data(), ngy * 4); c.up(dct, colterm.data(), ngx * 4);
    c.up(drz, rowzero.data(), ngy); c.up(dcz, colzero.data(), ngx);
    const unsigned long long init = ~0ull;
    c.up(dk, &init, 8);
    PmeParams P;
    P.bw = (int)block_width; P.bh = (int)block_height; P.src_pitch = (int)sp; P.ref_pitch = (int)rp; P.npos = npos; P.ngx = ngx;
    P.cost_type = (type == 0 && !tables) ? 5 : type; // ENTROPY without tables costs 0 (mcomp.c:50-56)
    P.error_per_bit = M.error_per_bit;
    for (int k = 0; k < 4; k++) P.joint_cost[k] = tables ? M.mvjcost[k] : 0;
    const int blocks = (npos + 3) / 4 < 1024 ? (npos + 3) / 4 : 1024;
    hipLaunchKernelGGL(pme_sad_loop_kernel, dim3(blocks), dim3(256), 0, c.stream, ds, dr, P, dxs, dys, drt, dct, drz, dcz, dk);
    SVT_LAUNCH_CHECK();
    unsigned long long k = 0;
    c.down(&k, dk, 8);
    const uint32_t cost = (uint32_t)(k >> 32);
    if (cost < *best_cost) {
        const uint32_t p = (uint32_t)k;
        const int iy = (int)(p / (uint32_t)ngx), ix = (int)(p % (uint32_t)ngx);
        *best_mvx  = (int16_t)(mvx + (int)((uint32_t)(search_position_start_x + xs[ix]) * 8u));
        *best_mvy  = (int16_t)(mvy + (int)((uint32_t)(search_position_start_y + ys[iy]) * 8u));
        *best_cost = cost;
    }
}

} // extern "C"

SVT_HIP_DEFINE_WARM(pme) // (svt_hip_warmup loads this translation unit's code object at encoder initialisation: svt_hip_common.h)
