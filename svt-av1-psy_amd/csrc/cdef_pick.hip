// cdef_pick.hip -- CDEF strength selection over a frame's per-filter-block distortion tables (the part that consumes
// svt_hip_cdef_frame's search output): svt_search_one_dual -> svt_search_one_dual_c (aom_dsp_rtcd.h:242, enc_cdef.c:627-683).
//
// tot[j][k] = sum over filter blocks i of min(best_i, mse0[i][j] + mse1[i][k]) with best_i = min over the already selected pairs; the winner
// is the first strictly smaller total in (j, k) raster order.  One workgroup per luma strength j, thread = (slice of filter blocks, chroma
// strength k): mse1 rows are read as 512 contiguous bytes per filter block, mse0[i][j] and best_i are wave-uniform.
#include "svt_hip_common.h"
#include "../../include/svtav1_hip.h"

#include <vector>

namespace {

constexpr int NS = 64; // TOTAL_STRENGTHS (cdef.h:47)

__global__ __launch_bounds__(256) void search_one_dual_kernel(const unsigned long long* __restrict__ mse0, const unsigned long long* __restrict__ mse1,
                                                              const int* __restrict__ lev0, const int* __restrict__ lev1, const int nb_strengths,
                                                              const int sb_count, const int start_gi, const int end_gi,
                                                              unsigned long long* __restrict__ row_best /* [64] */, int* __restrict__ row_arg /* [64] */) {
    __shared__ unsigned long long part[4][NS];
    const int j = start_gi + blockIdx.x, k = threadIdx.x & 63, sl = threadIdx.x >> 6;
    unsigned long long acc = 0;
    for (int i = sl; i < sb_count; i += 4) {
        unsigned long long best = 1ull << 63;
        for (int gi = 0; gi < nb_strengths; gi++) {
            const unsigned long long c = mse0[(size_t)i * NS + lev0[gi]] + mse1[(size_t)i * NS + lev1[gi]];
            best = c < best ? c : best;
        }
        const unsigned long long c = mse0[(size_t)i * NS + j] + mse1[(size_t)i * NS + k];
        acc += c < best ? c : best;
    }
    part[sl][k] = acc;
    __syncthreads();
    if (sl == 0) {
        unsigned long long tot = part[0][k] + part[1][k] + part[2][k] + part[3][k];
        int                arg = k;
        if (k < start_gi || k >= end_gi) tot = ~0ull; // outside the searched range: can never win (real totals are < 2^63 + ...)
        // first minimum over k: wave-wide (value, index) reduction, ties to the smaller index
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const unsigned long long t2 = (unsigned long long)__shfl_xor((long long)tot, m);
            const int                a2 = __shfl_xor(arg, m);
            if (t2 < tot || (t2 == tot && a2 < arg)) { tot = t2; arg = a2; }
        }
        if (k == 0) { row_best[blockIdx.x] = tot; row_arg[blockIdx.x] = arg; }
    }
}

// first strictly smaller total in raster order == smallest (total, j) with ties to the smaller j; also appends the winner to lev0 / lev1
__global__ __launch_bounds__(64) void search_one_dual_final_kernel(const unsigned long long* __restrict__ row_best, const int* __restrict__ row_arg, const int nrows,
                                                                   const int start_gi, int* __restrict__ lev0, int* __restrict__ lev1, const int nb_strengths,
                                                                   unsigned long long* __restrict__ out_best) {
    const int          l   = threadIdx.x;
    unsigned long long tot = l < nrows ? row_best[l] : ~0ull;
    int                arg = l;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const unsigned long long t2 = (unsigned long long)__shfl_xor((long long)tot, m);
        const int                a2 = __shfl_xor(arg, m);
        if (t2 < tot || (t2 == tot && a2 < arg)) { tot = t2; arg = a2; }
    }
    if (l == 0) {
        // the reference starts from best_tot_mse = 1 << 63 and ids (0, 0): totals >= 2^63 never win
        const bool won = nrows > 0 && tot < (1ull << 63);
        lev0[nb_strengths] = won ? start_gi + arg : 0;
        lev1[nb_strengths] = won ? row_arg[arg] : 0;
        out_best[0]        = won ? tot : (1ull << 63);
    }
}

} // namespace

extern "C" {

void svt_hip_cdef_search_one_dual(const uint64_t* mse0, const uint64_t* mse1, int* lev0, int* lev1, int nb_strengths, int sb_count, int start_gi,
                                  int end_gi, uint64_t* best_tot_mse, void* workspace, void* stream) {
    svthip::ensure_device();
    const int nrows = end_gi - start_gi;
    unsigned long long* row_best = (unsigned long long*)workspace;
    int*                row_arg  = (int*)(row_best + NS);
    if (nrows > 0) {
        hipLaunchKernelGGL(search_one_dual_kernel, dim3(nrows), dim3(256), 0, (hipStream_t)stream, (const unsigned long long*)mse0,
                           (const unsigned long long*)mse1, (const int*)lev0, (const int*)lev1, nb_strengths, sb_count, start_gi, end_gi, row_best, row_arg);
        SVT_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(search_one_dual_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const unsigned long long*)row_best, (const int*)row_arg,
                       nrows > 0 ? nrows : 0, start_gi, lev0, lev1, nb_strengths, (unsigned long long*)best_tot_mse);
    SVT_LAUNCH_CHECK();
}

uint64_t svt_search_one_dual_hip(int* lev0, int* lev1, int nb_strengths, uint64_t** mse[2], int sb_count, int start_gi, int end_gi) {
    svthip::HostCall& c = svthip::host_call();
    c.begin();
    const size_t tbl = (size_t)sb_count * NS * 8;
    c.reserve(2 * tbl + 8192, 2 * tbl + 8192);
    uint64_t* d0 = (uint64_t*)c.dalloc(tbl ? tbl : 8);
    uint64_t* d1 = (uint64_t*)c.dalloc(tbl ? tbl : 8);
    int*      dl0 = (int*)c.dalloc((NS + 1) * 4);
    int*      dl1 = (int*)c.dalloc((NS + 1) * 4);
    uint64_t* dbest = (uint64_t*)c.dalloc(8);
    void*     ws = c.dalloc(NS * 12);
    // the reference's tables are one allocation per filter block (pcs->mse_seg rows): gather them, then one upload per plane class
    std::vector<uint64_t> h((size_t)sb_count * NS + 1);
    for (int p = 0; p < 2; p++) {
        for (int i = 0; i < sb_count; i++) memcpy(h.data() + (size_t)i * NS, mse[p][i], NS * 8);
        if (tbl) c.up(p ? d1 : d0, h.data(), tbl);
    }
    c.up(dl0, lev0, nb_strengths * 4);
    c.up(dl1, lev1, nb_strengths * 4);
    svt_hip_cdef_search_one_dual(d0, d1, dl0, dl1, nb_strengths, sb_count, start_gi, end_gi, dbest, ws, c.stream);
    uint64_t best;
    int      a, b;
    c.down(&best, dbest, 8);
    c.down(&a, dl0 + nb_strengths, 4);
    c.down(&b, dl1 + nb_strengths, 4);
    lev0[nb_strengths] = a;
    lev1[nb_strengths] = b;
    return best;
}

} // extern "C"
