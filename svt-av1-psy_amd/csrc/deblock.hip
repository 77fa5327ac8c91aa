// deblock.hip -- AV1 deblocking edge filters: svt_aom_lpf_{horizontal,vertical}_{4,6,8,14} and the svt_aom_highbd_lpf_*
// family (common_dsp_rtcd.h:1037-1067; C: Codec/deblocking_common.c:141-865).  One thread filters one pixel position of an edge: it loads
// the 2 / 3 / 4 / 7 samples on either side, evaluates the filter / flat / flat2 masks and writes the modified samples back; one thread
// handles the four positions of a 4-sample segment with vector accesses.  The 8-bit
// functions are the high-bit-depth ones at bd = 8, so one routine serves both families.
//
// Batched form: a list of 4-sample edge segments over a device plane.  The caller orders the passes exactly as the reference's frame driver
// does (all vertical edges of the picture, then all horizontal edges, deblocking_filter.c): segments inside ONE launch must not overlap.
#include "svt_hip_common.h"
#include "../../include/svtav1_hip.h"

namespace {

__device__ __forceinline__ int sclamp(const int t, const int bd) { // signed_char_clamp_high (deblocking_common.c:28-35)
    const int lo = -(128 << (bd - 8)), hi = (128 << (bd - 8)) - 1;
    return t < lo ? lo : (t > hi ? hi : t);
}
__device__ __forceinline__ int rpot(const int v, const int n) { return (v + ((1 << n) >> 1)) >> n; }
__device__ __forceinline__ int iabs(const int v) { return v < 0 ? -v : v; }

// p[k] = k-th sample on the p side (p[0] next to the edge), q[k] likewise; LEN in {4, 6, 8, 14}
template <int LEN> __device__ __forceinline__ void lpf_px(int (&p)[7], int (&q)[7], const int blimit, const int limit, const int thresh, const int bd) {
    constexpr int TAPS = LEN == 4 ? 2 : (LEN == 6 ? 3 : 4);
    const int sh = bd - 8, limit16 = limit << sh, blimit16 = blimit << sh, thresh16 = thresh << sh, one16 = 1 << sh;
    bool bad = iabs(p[0] - q[0]) * 2 + iabs(p[1] - q[1]) / 2 > blimit16; // filter_mask* (:141-171)
#pragma unroll
    for (int k = 1; k < TAPS; k++) bad |= (iabs(p[k] - p[k - 1]) > limit16) | (iabs(q[k] - q[k - 1]) > limit16);
    const bool mask = !bad;
    bool flat = false, flat2 = false;
    if (LEN >= 6) { // flat_mask3_chroma / flat_mask4 with thresh = 1 (:173-205)
        bool b = false;
#pragma unroll
        for (int k = 1; k < TAPS; k++) b |= (iabs(p[k] - p[0]) > one16) | (iabs(q[k] - q[0]) > one16);
        flat = !b;
    }
    if (LEN == 14) { // flat_mask4(1, p6, p5, p4, p0, q0, q4, q5, q6) (:797)
        bool b = false;
#pragma unroll
        for (int k = 4; k < 7; k++) b |= (iabs(p[k] - p[0]) > one16) | (iabs(q[k] - q[0]) > one16);
        flat2 = !b;
    }
    if (LEN == 14 && flat2 && flat && mask) { // 13-tap filter (:762-785)
        const int p6 = p[6], p5 = p[5], p4 = p[4], p3 = p[3], p2 = p[2], p1 = p[1], p0 = p[0];
        const int q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3], q4 = q[4], q5 = q[5], q6 = q[6];
        p[5] = rpot(p6 * 7 + p5 * 2 + p4 * 2 + p3 + p2 + p1 + p0 + q0, 4);
        p[4] = rpot(p6 * 5 + p5 * 2 + p4 * 2 + p3 * 2 + p2 + p1 + p0 + q0 + q1, 4);
        p[3] = rpot(p6 * 4 + p5 + p4 * 2 + p3 * 2 + p2 * 2 + p1 + p0 + q0 + q1 + q2, 4);
        p[2] = rpot(p6 * 3 + p5 + p4 + p3 * 2 + p2 * 2 + p1 * 2 + p0 + q0 + q1 + q2 + q3, 4);
        p[1] = rpot(p6 * 2 + p5 + p4 + p3 + p2 * 2 + p1 * 2 + p0 * 2 + q0 + q1 + q2 + q3 + q4, 4);
        p[0] = rpot(p6 + p5 + p4 + p3 + p2 + p1 * 2 + p0 * 2 + q0 * 2 + q1 + q2 + q3 + q4 + q5, 4);
        q[0] = rpot(p5 + p4 + p3 + p2 + p1 + p0 * 2 + q0 * 2 + q1 * 2 + q2 + q3 + q4 + q5 + q6, 4);
        q[1] = rpot(p4 + p3 + p2 + p1 + p0 + q0 * 2 + q1 * 2 + q2 * 2 + q3 + q4 + q5 + q6 * 2, 4);
        q[2] = rpot(p3 + p2 + p1 + p0 + q0 + q1 * 2 + q2 * 2 + q3 * 2 + q4 + q5 + q6 * 3, 4);
        q[3] = rpot(p2 + p1 + p0 + q0 + q1 + q2 * 2 + q3 * 2 + q4 * 2 + q5 + q6 * 4, 4);
        q[4] = rpot(p1 + p0 + q0 + q1 + q2 + q3 * 2 + q4 * 2 + q5 * 2 + q6 * 5, 4);
        q[5] = rpot(p0 + q0 + q1 + q2 + q3 + q4 * 2 + q5 * 2 + q6 * 7, 4);
    } else if (LEN >= 8 && flat && mask) { // 7-tap filter (:289-304)
        const int p3 = p[3], p2 = p[2], p1 = p[1], p0 = p[0], q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
        p[2] = rpot(p3 + p3 + p3 + 2 * p2 + p1 + p0 + q0, 3);
        p[1] = rpot(p3 + p3 + p2 + 2 * p1 + p0 + q0 + q1, 3);
        p[0] = rpot(p3 + p2 + p1 + 2 * p0 + q0 + q1 + q2, 3);
        q[0] = rpot(p2 + p1 + p0 + 2 * q0 + q1 + q2 + q3, 3);
        q[1] = rpot(p1 + p0 + q0 + 2 * q1 + q2 + q3 + q3, 3);
        q[2] = rpot(p0 + q0 + q1 + 2 * q2 + q3 + q3 + q3, 3);
    } else if (LEN == 6 && flat && mask) { // 5-tap filter (:274-287)
        const int p2 = p[2], p1 = p[1], p0 = p[0], q0 = q[0], q1 = q[1], q2 = q[2];
        p[1] = rpot(p2 * 3 + p1 * 2 + p0 * 2 + q0, 3);
        p[0] = rpot(p2 + p1 * 2 + p0 * 2 + q0 * 2 + q1, 3);
        q[0] = rpot(p1 + p0 * 2 + q0 * 2 + q1 * 2 + q2, 3);
        q[1] = rpot(p0 + q0 * 2 + q1 * 2 + q2 * 3, 3);
    } else { // filter4 (:214-240, :426-458)
        const int off = 0x80 << sh, m = mask ? -1 : 0;
        const int ps1 = p[1] - off, ps0 = p[0] - off, qs0 = q[0] - off, qs1 = q[1] - off;
        const int hev = ((iabs(p[1] - p[0]) > thresh16) | (iabs(q[1] - q[0]) > thresh16)) ? -1 : 0;
        int f = sclamp(ps1 - qs1, bd) & hev;
        f = sclamp(f + 3 * (qs0 - ps0), bd) & m;
        const int f1 = sclamp(f + 4, bd) >> 3, f2 = sclamp(f + 3, bd) >> 3;
        q[0] = sclamp(qs0 - f1, bd) + off;
        p[0] = sclamp(ps0 + f2, bd) + off;
        f    = rpot(f1, 1) & ~hev;
        q[1] = sclamp(qs1 - f, bd) + off;
        p[1] = sclamp(ps1 + f, bd) + off;
    }
}

// N consecutive samples as one (possibly unaligned) vector access
template <typename PIX, int N> struct __attribute__((packed, aligned(1))) PxN { PIX v[N]; };

// One thread filters a whole 4-sample segment (4 independent evaluations of lpf_px).  The rows of a vertical edge are read as one span
// centred on the edge (16 samples for the 14-tap filters, 8 otherwise) and only the span that the filter may modify is written back --
// 12 / 6 / 4 samples -- because the samples beyond it belong to the neighbouring edges of the same pass.  A horizontal edge is read and
// written as 4-sample row vectors.  This replaced one thread per sample with up to 14 + 12 scalar accesses each.
template <typename PIX, int LEN>
__device__ __forceinline__ void lpf_segment(PIX* s, const uint32_t stride, const bool vertical, const int blimit, const int limit, const int thresh,
                                            const int bd) {
    constexpr int HALF = LEN == 14 ? 7 : LEN / 2, WR = LEN == 14 ? 6 : (LEN == 8 ? 3 : 2); // samples read / possibly modified per side
    constexpr int RD = LEN == 14 ? 8 : 4;                                                  // samples fetched per side of a vertical edge
    if (vertical) {
        PxN<PIX, 2 * RD> in[4]; // all four rows are fetched before the (branchy) filter runs: one memory round trip per segment
#pragma unroll
        for (int r = 0; r < 4; r++) in[r] = *(const PxN<PIX, 2 * RD>*)(s + (size_t)r * stride - RD);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            int p[7], q[7];
#pragma unroll
            for (int k = 0; k < HALF; k++) { p[k] = in[r].v[RD - 1 - k]; q[k] = in[r].v[RD + k]; }
            lpf_px<LEN>(p, q, blimit, limit, thresh, bd);
            PxN<PIX, 2 * WR> out;
#pragma unroll
            for (int k = 0; k < WR; k++) { out.v[WR - 1 - k] = (PIX)p[k]; out.v[WR + k] = (PIX)q[k]; }
            *(PxN<PIX, 2 * WR>*)(s + (size_t)r * stride - WR) = out;
        }
    } else {
        int p[4][7], q[4][7];
#pragma unroll
        for (int k = 0; k < HALF; k++) {
            const PxN<PIX, 4> a = *(const PxN<PIX, 4>*)(s - (size_t)(k + 1) * stride), b = *(const PxN<PIX, 4>*)(s + (size_t)k * stride);
#pragma unroll
            for (int c = 0; c < 4; c++) { p[c][k] = a.v[c]; q[c][k] = b.v[c]; }
        }
#pragma unroll
        for (int c = 0; c < 4; c++) lpf_px<LEN>(p[c], q[c], blimit, limit, thresh, bd);
#pragma unroll
        for (int k = 0; k < WR; k++) {
            PxN<PIX, 4> a, b;
#pragma unroll
            for (int c = 0; c < 4; c++) { a.v[c] = (PIX)p[c][k]; b.v[c] = (PIX)q[c][k]; }
            *(PxN<PIX, 4>*)(s - (size_t)(k + 1) * stride) = a;
            *(PxN<PIX, 4>*)(s + (size_t)k * stride)       = b;
        }
    }
}

// one thread per 4-sample segment
template <typename PIX>
__global__ __launch_bounds__(256) void lpf_edges_kernel(PIX* __restrict__ plane, const uint32_t stride, const int bd, const SvtHipLpfEdge* __restrict__ edges,
                                                        const uint32_t n) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n) return;
    const SvtHipLpfEdge e = edges[id];
    PIX* s = plane + (size_t)e.y * stride + e.x;
    switch (e.length) {
    case 4: lpf_segment<PIX, 4>(s, stride, e.vertical != 0, e.blimit, e.limit, e.thresh, bd); break;
    case 6: lpf_segment<PIX, 6>(s, stride, e.vertical != 0, e.blimit, e.limit, e.thresh, bd); break;
    case 8: lpf_segment<PIX, 8>(s, stride, e.vertical != 0, e.blimit, e.limit, e.thresh, bd); break;
    default: lpf_segment<PIX, 14>(s, stride, e.vertical != 0, e.blimit, e.limit, e.thresh, bd); break;
    }
}

// RTCD single-call form: s = q0 of the first sample, host memory.  Uploads the 4 x (2 * 7) neighbourhood, filters, downloads.
void lpf_host(void* s, int pitch, int is16, int vertical, int len, int blimit, int limit, int thresh, int bd) {
    svthip::HostCall& c = svthip::host_call();
    c.begin_small();
    const size_t px = is16 ? 2 : 1;
    // rectangle in host memory: vertical edge -> 4 rows x 14 columns starting 7 left of s; horizontal -> 14 rows x 4 columns starting 7 above
    const int    half = len == 14 ? 7 : len / 2;
    const int    rw = vertical ? 2 * half : 4, rh = vertical ? 4 : 2 * half;
    const size_t dp = 64; // device pitch in bytes
    const int    mx = vertical ? 8 - half : 0; // the kernel fetches 4 / 8 samples per side of a vertical edge: margin inside the staging rectangle
    c.reserve(dp * rh + 256, 2 * dp * rh + 256);
    uint8_t* d = (uint8_t*)c.dalloc(dp * rh);
    SvtHipLpfEdge* de = (SvtHipLpfEdge*)c.dalloc(sizeof(SvtHipLpfEdge));
    uint8_t* h0 = (uint8_t*)s - (vertical ? (size_t)half * px : (size_t)half * pitch * px);
    if (c.zc) memset(d, 0, dp * rh); // (the staging rectangle is pinned host memory in small-call mode)
    else HIP_CHECK(hipMemsetAsync(d, 0, dp * rh, c.stream));
    c.up2d(d + mx * px, dp, h0, (size_t)pitch * px, rw * px, rh);
    SvtHipLpfEdge e;
    memset(&e, 0, sizeof(e));
    e.x = vertical ? mx + half : 0; e.y = vertical ? 0 : half; e.vertical = (uint8_t)vertical; e.length = (uint8_t)len;
    e.blimit = (uint8_t)blimit; e.limit = (uint8_t)limit; e.thresh = (uint8_t)thresh;
    c.up(de, &e, sizeof(e));
    svt_hip_lpf_edges_batch(d, (uint32_t)(dp / px), is16, bd, de, 1, c.stream);
    // write back only the samples this filter length can modify -- the C kernels' store footprint: 2 per side for lengths 4 and 6 (p1 .. q1), 3 for 8
    // (p2 .. q2), 6 for 14 (p5 .. q5) -- so that neighbouring edges filtered concurrently by other DLF threads are never rewritten with stale values
    const int mod = len == 14 ? 6 : (len == 8 ? 3 : 2), skip = half - mod;
    if (vertical) c.down2d(h0 + (size_t)skip * px, (size_t)pitch * px, d + (mx + skip) * px, dp, (size_t)2 * mod * px, rh);
    else c.down2d(h0 + (size_t)skip * pitch * px, (size_t)pitch * px, d + (size_t)skip * dp, dp, rw * px, (size_t)2 * mod);
}

} // namespace

extern "C" {

void svt_hip_lpf_edges_batch(void* plane, uint32_t stride, int is_16bit, int bd, const SvtHipLpfEdge* edges, uint32_t n, void* stream) {
    svthip::ensure_device();
    if (n == 0) return;
    const dim3 grid((n + 255) / 256);
    if (is_16bit) hipLaunchKernelGGL(HIP_KERNEL_NAME(lpf_edges_kernel<uint16_t>), grid, dim3(256), 0, (hipStream_t)stream, (uint16_t*)plane, stride, bd, edges, n);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(lpf_edges_kernel<uint8_t>), grid, dim3(256), 0, (hipStream_t)stream, (uint8_t*)plane, stride, 8, edges, n);
    SVT_LAUNCH_CHECK();
}

// Host-pointer form for one plane of a picture (what a seam around svt_av1_loop_filter_frame, dlf_process.c:122, calls after recording the edge segments the
// reference's own driver would filter): uploads the plane with a 16-sample margin either side (the reference's picture padding), runs all vertical-edge
// segments, then all horizontal-edge segments, downloads the plane in place.  x / y of a segment = its first q0 sample inside the plane.
int svt_hip_lpf_plane_host(void* plane, uint32_t stride, uint32_t width, uint32_t height, int is_16bit, int bd, const SvtHipLpfEdge* vert, uint32_t n_vert,
                            const SvtHipLpfEdge* horz, uint32_t n_horz) {
    SVT_HIP_ENTRY_TRY
    svthip::ensure_device();
    if (n_vert + n_horz == 0) return 0;
    const size_t px = is_16bit ? 2 : 1, M = 16, pitch = svthip::align_up((width + 2 * M) * px, 16), nb = (size_t)(n_vert + n_horz) * sizeof(SvtHipLpfEdge);
    svthip::HostCallLease lease; // (a pooled arena: see svt_hip_common.h)
    svthip::HostCall& c = *lease;
    c.begin();
    c.reserve(pitch * height + nb + 8192, 2 * pitch * height + nb + 8192);
    uint8_t*       d  = (uint8_t*)c.dalloc(pitch * height);
    SvtHipLpfEdge* de = (SvtHipLpfEdge*)c.dalloc(nb ? nb : 16);
    c.up2d(d, pitch, (const uint8_t*)plane - M * px, (size_t)stride * px, (width + 2 * M) * px, height);
    SvtHipLpfEdge* he = (SvtHipLpfEdge*)c.palloc(nb);
    for (uint32_t i = 0; i < n_vert; i++) { he[i] = vert[i]; he[i].x += (uint32_t)M; }
    for (uint32_t i = 0; i < n_horz; i++) { he[n_vert + i] = horz[i]; he[n_vert + i].x += (uint32_t)M; }
    HIP_CHECK(hipMemcpyAsync(de, he, nb, hipMemcpyHostToDevice, c.stream));
    if (n_vert) svt_hip_lpf_edges_batch(d, (uint32_t)(pitch / px), is_16bit, bd, de, n_vert, c.stream);
    if (n_horz) svt_hip_lpf_edges_batch(d, (uint32_t)(pitch / px), is_16bit, bd, de + n_vert, n_horz, c.stream);
    c.down2d(plane, (size_t)stride * px, d + M * px, pitch, width * px, height);
    return 0;
    SVT_HIP_ENTRY_CATCH(SVT_HIP_E_DEVICE)
}

#define LPF_PAIR(LEN)                                                                                                                                        \
    void svt_aom_lpf_horizontal_##LEN##_hip(uint8_t* s, int32_t pitch, const uint8_t* blimit, const uint8_t* limit, const uint8_t* thresh) {               \
        lpf_host(s, pitch, 0, 0, LEN, *blimit, *limit, *thresh, 8);                                                                                         \
    }                                                                                                                                                        \
    void svt_aom_lpf_vertical_##LEN##_hip(uint8_t* s, int32_t pitch, const uint8_t* blimit, const uint8_t* limit, const uint8_t* thresh) {                 \
        lpf_host(s, pitch, 0, 1, LEN, *blimit, *limit, *thresh, 8);                                                                                         \
    }                                                                                                                                                        \
    void svt_aom_highbd_lpf_horizontal_##LEN##_hip(uint16_t* s, int32_t pitch, const uint8_t* blimit, const uint8_t* limit, const uint8_t* thresh,        \
                                                   int32_t bd) {                                                                                             \
        lpf_host(s, pitch, 1, 0, LEN, *blimit, *limit, *thresh, bd);                                                                                        \
    }                                                                                                                                                        \
    void svt_aom_highbd_lpf_vertical_##LEN##_hip(uint16_t* s, int32_t pitch, const uint8_t* blimit, const uint8_t* limit, const uint8_t* thresh,          \
                                                 int32_t bd) {                                                                                               \
        lpf_host(s, pitch, 1, 1, LEN, *blimit, *limit, *thresh, bd);                                                                                        \
    }
LPF_PAIR(4)
LPF_PAIR(6)
LPF_PAIR(8)
LPF_PAIR(14)
#undef LPF_PAIR

} // extern "C"

SVT_HIP_DEFINE_WARM(deblock) // (svt_hip_warmup loads this translation unit's code object at encoder initialisation: svt_hip_common.h)
