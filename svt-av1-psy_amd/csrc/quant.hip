// quant.hip -- quantize / dequantize and svt_handle_transform* (a11) for gfx950.
//
// All four reference quantizers are element-wise in the coefficient position rc, plus eob = 1 + max scan index with a
// non-zero level.  (The low-bit-depth quantize_b's reverse pre-scan, full_loop.c:41-51, only skips coefficients that
// its own per-coefficient zbin test would zero anyway, so it has no effect on the outputs.)  Mapping: 4 coefficients
// per lane (16-byte loads/stores), min(64, n/4) lanes per block, eob by a DPP/shuffle max over iscan[rc]+1.
// HBM bound: 4 B in + 8 B out per coefficient (+2 B iscan, cache resident).
#include "svt_hip_common.h"
#include "../../include/svtav1_hip.h"
#include "quant_core.h"

namespace {

__device__ __forceinline__ int32_t rpot(const int32_t v, const int n) { return (v + ((1 << n) >> 1)) >> n; }

struct __attribute__((aligned(16))) i32x4 { int32_t v[4]; };
struct __attribute__((aligned(8))) i16x4 { int16_t v[4]; };
struct __attribute__((aligned(4))) u8x4 { uint8_t v[4]; };

template <int MODE, bool QM>
__global__ __launch_bounds__(256) void quant_kernel(const int32_t* __restrict__ coeff, const uint32_t n, const uint32_t n_coeffs, const int lpb_log2,
                                                    const SvtHipQuantParams* __restrict__ qparams, const int16_t* __restrict__ iscan_tables,
                                                    const uint8_t* __restrict__ qm_tables, const uint8_t* __restrict__ iqm_tables,
                                                    const SvtHipQuantDesc* __restrict__ descs, int32_t* __restrict__ qcoeff,
                                                    int32_t* __restrict__ dqcoeff, uint16_t* __restrict__ eob_out) {
    const int      lpb  = 1 << lpb_log2; // lanes per block
    const uint32_t gt   = blockIdx.x * 256 + threadIdx.x;
    const uint32_t blk  = gt >> lpb_log2;
    const int      sub  = (int)(gt & (uint32_t)(lpb - 1));
    const bool     live = blk < n;
    uint32_t       eob  = 0;
    if (live) {
        const SvtHipQuantDesc   d = descs[blk];
        const SvtHipQuantParams P = qparams[d.qparam_idx];
        const int16_t*          isc = iscan_tables + (size_t)d.iscan_idx * n_coeffs;
        const uint8_t*          qm  = QM ? qm_tables + (size_t)d.qm_idx * n_coeffs : nullptr;
        const uint8_t*          iqm = QM ? iqm_tables + (size_t)d.qm_idx * n_coeffs : nullptr;
        const size_t            base = (size_t)blk * n_coeffs;
        for (uint32_t rc0 = (uint32_t)sub * 4; rc0 < n_coeffs; rc0 += (uint32_t)lpb * 4) {
            const i32x4 c  = *(const i32x4*)(coeff + base + rc0);
            const i16x4 is = *(const i16x4*)(isc + rc0);
            u8x4        w, iw;
            if (QM) {
                w  = *(const u8x4*)(qm + rc0);
                iw = *(const u8x4*)(iqm + rc0);
            }
            i32x4 q, dq;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const QOut o = quant_one<MODE, QM>(c.v[j], (rc0 + j) != 0, P, QM ? w.v[j] : (1 << QM_BITS), QM ? iw.v[j] : (1 << QM_BITS));
                q.v[j]  = o.q;
                dq.v[j] = o.dq;
                if (o.q != 0) {
                    const uint32_t e = (uint32_t)is.v[j] + 1u;
                    eob              = e > eob ? e : eob;
                }
            }
            *(i32x4*)(qcoeff + base + rc0)  = q;
            *(i32x4*)(dqcoeff + base + rc0) = dq;
        }
    }
    // max over the lpb lanes of this block (lpb is a power of two <= 64, blocks never straddle a wave)
    for (int m = lpb >> 1; m >= 1; m >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)eob, m);
        eob              = o > eob ? o : eob;
    }
    if (live && sub == 0) eob_out[blk] = (uint16_t)eob;
}

// ---- svt_handle_transform* --------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void handle_transform_kernel(int32_t* __restrict__ coeff, const int w, const int h, const int n2n4,
                                                               unsigned long long* __restrict__ energy) {
    __shared__ unsigned long long part[4];
    const int          tid = threadIdx.x;
    int32_t*           blk = coeff + (size_t)blockIdx.x * (w * h);
    unsigned long long e   = 0;
    const int          hh  = h > 32 ? 32 : h;
    if (!n2n4) {
        if (w == 64)
            for (int i = tid; i < hh * 32; i += 256) {
                const long long v = blk[(i >> 5) * 64 + 32 + (i & 31)];
                e += (unsigned long long)(v * v);
            }
        if (h == 64)
            for (int i = tid; i < 32 * w; i += 256) {
                const long long v = blk[32 * w + i];
                e += (unsigned long long)(v * v);
            }
    }
    // repack 64-wide rows to stride 32 in place: read everything first, then write
    int32_t keep[4];
    if (w == 64) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = tid + 256 * k; // hh * 32 <= 1024 elements
            keep[k]     = i < hh * 32 ? blk[(i >> 5) * 64 + (i & 31)] : 0;
        }
    }
    for (int m = 32; m >= 1; m >>= 1) {
        const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)e, m), hi = (unsigned)__shfl_xor((int)(unsigned)(e >> 32), m);
        e += ((unsigned long long)hi << 32) | lo;
    }
    if ((tid & 63) == 0) part[tid >> 6] = e;
    __syncthreads();
    if (w == 64) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = tid + 256 * k;
            if (i < hh * 32) blk[i] = keep[k];
        }
    }
    if (tid == 0) energy[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

template <int MODE>
void launch_quant(bool qm, uint32_t grid, hipStream_t st, const int32_t* coeff, uint32_t n, uint32_t n_coeffs, int lpb_log2,
                  const SvtHipQuantParams* qp, const int16_t* isc, const uint8_t* qmt, const uint8_t* iqmt, const SvtHipQuantDesc* descs,
                  int32_t* q, int32_t* dq, uint16_t* eob) {
    if (qm)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(quant_kernel<MODE, true>), dim3(grid), dim3(256), 0, st, coeff, n, n_coeffs, lpb_log2, qp, isc, qmt, iqmt, descs, q, dq, eob);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(quant_kernel<MODE, false>), dim3(grid), dim3(256), 0, st, coeff, n, n_coeffs, lpb_log2, qp, isc, qmt, iqmt, descs, q, dq, eob);
    SVT_LAUNCH_CHECK();
}

constexpr int kTxW[19] = {4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64};
constexpr int kTxH[19] = {4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16};

} // namespace

extern "C" {

void svt_hip_quantize_batch(int mode, const int32_t* coeff, uint32_t n, uint32_t n_coeffs, const SvtHipQuantParams* qparams,
                            const int16_t* iscan_tables, const uint8_t* qm_tables, const uint8_t* iqm_tables, const SvtHipQuantDesc* descs,
                            int32_t* qcoeff, int32_t* dqcoeff, uint16_t* eob, void* stream) {
    svthip::ensure_device();
    if (n == 0) return;
    if (n_coeffs < 16 || (n_coeffs & (n_coeffs - 1))) { fprintf(stderr, "libsvtav1_hip: quantize n_coeffs must be a power of two >= 16\n"); abort(); }
    int lpb_log2 = 0;
    while ((4u << lpb_log2) < n_coeffs && lpb_log2 < 6) lpb_log2++;
    const uint64_t threads = (uint64_t)n << lpb_log2;
    const uint32_t grid    = (uint32_t)((threads + 255) / 256);
    const bool     qm      = qm_tables != nullptr && iqm_tables != nullptr;
    hipStream_t    st      = (hipStream_t)stream;
    switch (mode) {
    case 0: launch_quant<0>(qm, grid, st, coeff, n, n_coeffs, lpb_log2, qparams, iscan_tables, qm_tables, iqm_tables, descs, qcoeff, dqcoeff, eob); break;
    case 1: launch_quant<1>(qm, grid, st, coeff, n, n_coeffs, lpb_log2, qparams, iscan_tables, qm_tables, iqm_tables, descs, qcoeff, dqcoeff, eob); break;
    case 2: launch_quant<2>(qm, grid, st, coeff, n, n_coeffs, lpb_log2, qparams, iscan_tables, qm_tables, iqm_tables, descs, qcoeff, dqcoeff, eob); break;
    case 3: launch_quant<3>(qm, grid, st, coeff, n, n_coeffs, lpb_log2, qparams, iscan_tables, qm_tables, iqm_tables, descs, qcoeff, dqcoeff, eob); break;
    default: fprintf(stderr, "libsvtav1_hip: bad quantize mode %d\n", mode); abort();
    }
}

void svt_hip_handle_transform_batch(int32_t* coeff, uint32_t n, int tx_size, int n2_n4, uint64_t* energy, void* stream) {
    svthip::ensure_device();
    if (n == 0) return;
    hipLaunchKernelGGL(handle_transform_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, coeff, kTxW[tx_size], kTxH[tx_size], n2_n4,
                       (unsigned long long*)energy);
    SVT_LAUNCH_CHECK();
}

// ---- RTCD-signature single-call forms -----------------------------------------------------------------------------
// One entry for all ten quantizer pointers: svt_aom_quantize_b / svt_aom_highbd_quantize_b / svt_av1_quantize_b_qm /
// svt_av1_highbd_quantize_b_qm (mode 0/1), svt_av1_quantize_fp[_32x32|_64x64|_qm] (mode 2),
// svt_av1_highbd_quantize_fp[_qm] (mode 3)  -- aom_dsp_rtcd.c:216-225.
void svt_quantize_hip(int mode, const int32_t* coeff_ptr, intptr_t n_coeffs, const int16_t* zbin_ptr, const int16_t* round_ptr,
                      const int16_t* quant_ptr, const int16_t* quant_shift_ptr, int32_t* qcoeff_ptr, int32_t* dqcoeff_ptr,
                      const int16_t* dequant_ptr, uint16_t* eob_ptr, const int16_t* scan, const int16_t* iscan, const uint8_t* qm_ptr,
                      const uint8_t* iqm_ptr, int log_scale) {
    (void)scan; // the kernel works position-wise and needs only the inverse scan
    const size_t n = (size_t)n_coeffs;
    svthip::HostCall& c = svthip::host_call();
    c.begin_small();
    c.reserve(n * 16 + 8192, n * 24 + 8192);
    int32_t*           dco = (int32_t*)c.dalloc(n * 4);
    int32_t*           dq  = (int32_t*)c.dalloc(n * 4);
    int32_t*           ddq = (int32_t*)c.dalloc(n * 4);
    int16_t*           dis = (int16_t*)c.dalloc(n * 2);
    uint8_t*           dqm = (uint8_t*)c.dalloc(n);
    uint8_t*           diq = (uint8_t*)c.dalloc(n);
    SvtHipQuantParams* dp  = (SvtHipQuantParams*)c.dalloc(sizeof(SvtHipQuantParams));
    SvtHipQuantDesc*   dd  = (SvtHipQuantDesc*)c.dalloc(sizeof(SvtHipQuantDesc));
    uint16_t*          de  = (uint16_t*)c.dalloc(4);
    c.up(dco, coeff_ptr, n * 4);
    c.up(dis, iscan, n * 2);
    const bool qm = (qm_ptr != nullptr) || (iqm_ptr != nullptr);
    if (qm) { // a missing matrix means flat weights (1 << AOM_QM_BITS), full_loop.c:312-313
        uint8_t* t = (uint8_t*)c.palloc(n);
        memset(t, 1 << QM_BITS, n);
        c.up(dqm, qm_ptr ? qm_ptr : t, n);
        c.up(diq, iqm_ptr ? iqm_ptr : t, n);
    }
    SvtHipQuantParams P;
    memset(&P, 0, sizeof(P));
    for (int k = 0; k < 2; k++) {
        P.zbin[k] = zbin_ptr ? zbin_ptr[k] : 0; P.round[k] = round_ptr[k]; P.quant[k] = quant_ptr[k];
        P.quant_shift[k] = quant_shift_ptr ? quant_shift_ptr[k] : 0; P.dequant[k] = dequant_ptr[k];
    }
    P.log_scale = log_scale;
    c.up(dp, &P, sizeof(P));
    SvtHipQuantDesc d = {0, 0, 0, 0};
    c.up(dd, &d, sizeof(d));
    svt_hip_quantize_batch(mode, dco, 1, (uint32_t)n, dp, dis, qm ? dqm : nullptr, qm ? diq : nullptr, dd, dq, ddq, de, c.stream);
    c.down_later(qcoeff_ptr, dq, n * 4); // ONE commit point (rtcd_hook.hip: a failed call is finished through the saved pointer, so nothing may be half written)
    c.down_later(dqcoeff_ptr, ddq, n * 4);
    c.down_later(eob_ptr, de, 2);
    c.finish();
}

#define QARGS const int32_t *coeff_ptr, intptr_t n_coeffs, const int16_t *zbin_ptr, const int16_t *round_ptr, const int16_t *quant_ptr, \
              const int16_t *quant_shift_ptr, int32_t *qcoeff_ptr, int32_t *dqcoeff_ptr, const int16_t *dequant_ptr, uint16_t *eob_ptr,     \
              const int16_t *scan, const int16_t *iscan
#define QPASS coeff_ptr, n_coeffs, zbin_ptr, round_ptr, quant_ptr, quant_shift_ptr, qcoeff_ptr, dqcoeff_ptr, dequant_ptr, eob_ptr, scan, iscan
void svt_aom_quantize_b_hip(QARGS, const uint8_t* qm_ptr, const uint8_t* iqm_ptr, const int32_t log_scale) { svt_quantize_hip(0, QPASS, qm_ptr, iqm_ptr, log_scale); }
void svt_aom_highbd_quantize_b_hip(QARGS, const uint8_t* qm_ptr, const uint8_t* iqm_ptr, const int32_t log_scale) { svt_quantize_hip(1, QPASS, qm_ptr, iqm_ptr, log_scale); }
void svt_av1_quantize_fp_hip(QARGS) { svt_quantize_hip(2, QPASS, nullptr, nullptr, 0); }
void svt_av1_quantize_fp_32x32_hip(QARGS) { svt_quantize_hip(2, QPASS, nullptr, nullptr, 1); }
void svt_av1_quantize_fp_64x64_hip(QARGS) { svt_quantize_hip(2, QPASS, nullptr, nullptr, 2); }
void svt_av1_quantize_fp_qm_hip(QARGS, const uint8_t* qm_ptr, const uint8_t* iqm_ptr, int16_t log_scale) { svt_quantize_hip(2, QPASS, qm_ptr, iqm_ptr, log_scale); }
void svt_av1_highbd_quantize_fp_hip(QARGS, int16_t log_scale) { svt_quantize_hip(3, QPASS, nullptr, nullptr, log_scale); }
void svt_av1_highbd_quantize_fp_qm_hip(QARGS, const uint8_t* qm_ptr, const uint8_t* iqm_ptr, int16_t log_scale) { svt_quantize_hip(3, QPASS, qm_ptr, iqm_ptr, log_scale); }

uint64_t svt_handle_transform_hip(int32_t* output, int tx_size, int n2_n4) {
    const size_t n = (size_t)kTxW[tx_size] * kTxH[tx_size];
    svthip::HostCall& c = svthip::host_call();
    c.begin_small();
    c.reserve(n * 4 + 1024, n * 8 + 1024);
    int32_t*  d  = (int32_t*)c.dalloc(n * 4);
    uint64_t* de = (uint64_t*)c.dalloc(8);
    c.up(d, output, n * 4);
    svt_hip_handle_transform_batch(d, 1, tx_size, n2_n4, de, c.stream);
    uint64_t e;
    c.down_later(&e, de, 8);
    c.down_later(output, d, n * 4); // in place: written only once every HIP operation has succeeded
    c.finish();
    return e;
}
#define HT(W, H, ID)                                                                                      \
    uint64_t svt_handle_transform##W##x##H##_hip(int32_t* output) { return svt_handle_transform_hip(output, ID, 0); } \
    uint64_t svt_handle_transform##W##x##H##_N2_N4_hip(int32_t* output) { return svt_handle_transform_hip(output, ID, 1); }
HT(64, 64, 4) HT(32, 64, 11) HT(64, 32, 12) HT(16, 64, 17) HT(64, 16, 18)

} // extern "C"

SVT_HIP_DEFINE_WARM(quant) // (svt_hip_warmup loads this translation unit's code object at encoder initialisation: svt_hip_common.h)
