// cdef_pick.hip -- CDEF strength selection over a frame's per-filter-block distortion tables (the part that consumes
// svt_hip_cdef_frame's search output): svt_search_one_dual -> svt_search_one_dual_c (aom_dsp_rtcd.h:242, enc_cdef.c:627-683).
//
// tot[j][k] = sum over filter blocks i of min(best_i, mse0[i][j] + mse1[i][k]) with best_i = min over the already selected pairs; the winner
// is the first strictly smaller total in (j, k) raster order.  Three small kernels: best_i per filter block, the 64 x 64 totals accumulated by
// (strength j, slice of filter blocks) workgroups with 64-bit atomics, and the pick.
#include "svt_hip_common.h"
#include "../../include/svtav1_hip.h"

#include <vector>

namespace {

constexpr int NS = 64; // TOTAL_STRENGTHS (cdef.h:47)

// best_i = smallest luma + chroma distortion of filter block i over the pairs selected so far (1 << 63 when there are none)
__global__ __launch_bounds__(256) void search_best_kernel(const unsigned long long* __restrict__ mse0, const unsigned long long* __restrict__ mse1,
                                                          const int* __restrict__ lev0, const int* __restrict__ lev1, const int nb_strengths, const int sb_count,
                                                          unsigned long long* __restrict__ best, unsigned long long* __restrict__ tot) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < NS * NS) tot[i] = 0; // the launch also clears the accumulation table (grid covers >= 4096 threads)
    if (i >= sb_count) return;
    unsigned long long b = 1ull << 63;
    for (int gi = 0; gi < nb_strengths; gi++) {
        const unsigned long long c = mse0[(size_t)i * NS + lev0[gi]] + mse1[(size_t)i * NS + lev1[gi]];
        b = c < b ? c : b;
    }
    best[i] = b;
}
// tot[j][k] += sum over a slice of filter blocks of min(best_i, mse0[i][j] + mse1[i][k]).  Workgroup = (luma strength j, slice of SLICE filter
// blocks); thread = (one of four block lanes, chroma strength k): mse1 rows are read as 512 contiguous bytes, mse0[i][j] / best_i are uniform.
constexpr int SLICE = 128;
__global__ __launch_bounds__(256) void search_tot_kernel(const unsigned long long* __restrict__ mse0, const unsigned long long* __restrict__ mse1,
                                                         const unsigned long long* __restrict__ best, const int sb_count, const int start_gi,
                                                         unsigned long long* __restrict__ tot) {
    __shared__ unsigned long long part[4][NS];
    const int j = start_gi + blockIdx.x, k = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int i0 = blockIdx.y * SLICE, i1 = i0 + SLICE < sb_count ? i0 + SLICE : sb_count;
    unsigned long long acc = 0;
    for (int i = i0 + sl; i < i1; i += 4) {
        const unsigned long long c = mse0[(size_t)i * NS + j] + mse1[(size_t)i * NS + k], b = best[i];
        acc += c < b ? c : b;
    }
    part[sl][k] = acc;
    __syncthreads();
    if (sl == 0) atomicAdd(&tot[j * NS + k], part[0][k] + part[1][k] + part[2][k] + part[3][k]);
}
// first strictly smaller total in (j, k) raster order over [start_gi, end_gi)^2, starting from 1 << 63 with ids (0, 0) (enc_cdef.c:668-682); appends
// the winner to lev0 / lev1
__global__ __launch_bounds__(256) void search_pick_kernel(const unsigned long long* __restrict__ tot, const int start_gi, const int end_gi, int* __restrict__ lev0,
                                                          int* __restrict__ lev1, const int nb_strengths, unsigned long long* __restrict__ out_best) {
    __shared__ unsigned long long wv_tot[4];
    __shared__ int                wv_arg[4];
    const int tid = threadIdx.x;
    unsigned long long t = ~0ull;
    int                a = NS * NS;
    for (int e = tid; e < NS * NS; e += 256) {
        const int j = e >> 6, k = e & 63;
        if (j >= start_gi && j < end_gi && k >= start_gi && k < end_gi) {
            const unsigned long long v = tot[e];
            if (v < t || (v == t && e < a)) { t = v; a = e; }
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const unsigned long long t2 = (unsigned long long)__shfl_xor((long long)t, m);
        const int                a2 = __shfl_xor(a, m);
        if (t2 < t || (t2 == t && a2 < a)) { t = t2; a = a2; }
    }
    if ((tid & 63) == 0) { wv_tot[tid >> 6] = t; wv_arg[tid >> 6] = a; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; w++)
            if (wv_tot[w] < t || (wv_tot[w] == t && wv_arg[w] < a)) { t = wv_tot[w]; a = wv_arg[w]; }
        const bool won = a < NS * NS && t < (1ull << 63);
        lev0[nb_strengths] = won ? a >> 6 : 0;
        lev1[nb_strengths] = won ? a & 63 : 0;
        out_best[0]        = won ? t : (1ull << 63);
    }
}

// joint_strength_search_dual's refinement step (enc_cdef.c:717-722): drop the oldest pair
__global__ void shift_levels_kernel(int* lev0, int* lev1, const int nb_strengths) {
    if (threadIdx.x == 0 && blockIdx.x == 0)
        for (int j = 0; j < nb_strengths - 1; j++) { lev0[j] = lev0[j + 1]; lev1[j] = lev1[j + 1]; }
}
// finish_cdef_search's per-filter-block choice (enc_cdef.c:916-931): first strictly smaller sum over the selected pairs
__global__ __launch_bounds__(256) void assign_fb_kernel(const unsigned long long* __restrict__ mse0, const unsigned long long* __restrict__ mse1,
                                                        const int* __restrict__ lev0, const int* __restrict__ lev1, const int nb_strengths, const int sb_count,
                                                        int8_t* __restrict__ best_gi) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= sb_count) return;
    unsigned long long best = 1ull << 63;
    int                arg  = 0;
    for (int gi = 0; gi < nb_strengths; gi++) {
        const unsigned long long c = mse0[(size_t)i * NS + lev0[gi]] + mse1[(size_t)i * NS + lev1[gi]];
        if (c < best) { best = c; arg = gi; }
    }
    best_gi[i] = (int8_t)arg;
}

} // namespace

extern "C" {

void svt_hip_cdef_search_one_dual(const uint64_t* mse0, const uint64_t* mse1, int* lev0, int* lev1, int nb_strengths, int sb_count, int start_gi,
                                  int end_gi, uint64_t* best_tot_mse, void* workspace, void* stream) {
    svthip::ensure_device();
    hipStream_t st = (hipStream_t)stream;
    unsigned long long* tot  = (unsigned long long*)workspace; // [64][64]
    unsigned long long* best = tot + NS * NS;                   // [sb_count]
    const int nrows = end_gi - start_gi, cover = sb_count > NS * NS ? sb_count : NS * NS;
    hipLaunchKernelGGL(search_best_kernel, dim3((cover + 255) / 256), dim3(256), 0, st, (const unsigned long long*)mse0, (const unsigned long long*)mse1,
                       (const int*)lev0, (const int*)lev1, nb_strengths, sb_count, best, tot);
    SVT_LAUNCH_CHECK();
    if (nrows > 0 && sb_count > 0) {
        hipLaunchKernelGGL(search_tot_kernel, dim3(nrows, (sb_count + SLICE - 1) / SLICE), dim3(256), 0, st, (const unsigned long long*)mse0,
                           (const unsigned long long*)mse1, (const unsigned long long*)best, sb_count, start_gi, tot);
        SVT_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(search_pick_kernel, dim3(1), dim3(256), 0, st, (const unsigned long long*)tot, start_gi, end_gi, lev0, lev1, nb_strengths,
                       (unsigned long long*)best_tot_mse);
    SVT_LAUNCH_CHECK();
}

void svt_hip_cdef_joint_strength_search(const uint64_t* mse0, const uint64_t* mse1, int* lev0, int* lev1, int nb_strengths, int sb_count, int start_gi,
                                        int end_gi, uint64_t* best_tot_mse, void* workspace, void* stream) {
    for (int i = 0; i < nb_strengths; i++) // greedy: add one pair at a time (enc_cdef.c:714-715)
        svt_hip_cdef_search_one_dual(mse0, mse1, lev0, lev1, i, sb_count, start_gi, end_gi, best_tot_mse, workspace, stream);
    for (int i = 0; i < 4 * nb_strengths; i++) { // refinement: reconsider each selected pair in turn (enc_cdef.c:719-725)
        hipLaunchKernelGGL(shift_levels_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, lev0, lev1, nb_strengths);
        SVT_LAUNCH_CHECK();
        svt_hip_cdef_search_one_dual(mse0, mse1, lev0, lev1, nb_strengths - 1, sb_count, start_gi, end_gi, best_tot_mse, workspace, stream);
    }
}

void svt_hip_cdef_assign_fb_strengths(const uint64_t* mse0, const uint64_t* mse1, const int* lev0, const int* lev1, int nb_strengths, int sb_count,
                                      int8_t* best_gi, void* stream) {
    svthip::ensure_device();
    if (sb_count <= 0) return;
    hipLaunchKernelGGL(assign_fb_kernel, dim3((sb_count + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const unsigned long long*)mse0,
                       (const unsigned long long*)mse1, lev0, lev1, nb_strengths, sb_count, best_gi);
    SVT_LAUNCH_CHECK();
}

uint64_t svt_search_one_dual_hip(int* lev0, int* lev1, int nb_strengths, uint64_t** mse[2], int sb_count, int start_gi, int end_gi) {
    svthip::HostCall& c = svthip::host_call();
    c.begin();
    const size_t tbl = (size_t)sb_count * NS * 8;
    c.reserve(2 * tbl + (size_t)(NS * NS + sb_count + 1) * 8 + 8192, 2 * tbl + 8192);
    uint64_t* d0 = (uint64_t*)c.dalloc(tbl ? tbl : 8);
    uint64_t* d1 = (uint64_t*)c.dalloc(tbl ? tbl : 8);
    int*      dl0 = (int*)c.dalloc((NS + 1) * 4);
    int*      dl1 = (int*)c.dalloc((NS + 1) * 4);
    uint64_t* dbest = (uint64_t*)c.dalloc(8);
    void*     ws = c.dalloc((size_t)(NS * NS + sb_count + 1) * 8);
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
    c.down_later(&best, dbest, 8); // ONE commit point: the caller's memory is written after the last HIP operation has succeeded (rtcd_hook.hip: Guard)
    c.down_later(&a, dl0 + nb_strengths, 4);
    c.down_later(&b, dl1 + nb_strengths, 4);
    c.finish();
    lev0[nb_strengths] = a;
    lev1[nb_strengths] = b;
    return best;
}

} // extern "C"

SVT_HIP_DEFINE_WARM(cdef_pick) // (svt_hip_warmup loads this translation unit's code object at encoder initialisation: svt_hip_common.h)
