// txfm_fused.hip -- BASELINE config 3 as ONE launch: forward 2-D transform -> (svt_handle_transform for 64-point sizes) -> quantize / dequantize ->
// inverse 2-D transform + reconstruction, per TX block ("coeffs never leave LDS / registers").
//
// The four-launch chain (svt_hip_fwd_txfm2d_batch, svt_hip_handle_transform_batch, svt_hip_quantize_batch, svt_hip_inv_txfm2d_add_batch) moves
// 6 + 12 + 8 B per pixel through HBM because every stage hands its int32 coefficients to the next one through memory.  Here the block's coefficient tile
// stays in the LDS tile the forward transform wrote its result to: the quantizer reads it there, writes qcoeff (the product the entropy coder needs: 4 B
// per kept coefficient) and puts dqcoeff back into the same LDS cells, and the inverse transform starts from them.  Algorithmic HBM bytes per pixel:
// 2 (residual) + 4 (qcoeff) + 2 (prediction) + 2 (reconstruction) = 10, + 4 when the caller also wants dqcoeff.
// Same arithmetic, same order, same rounding as the separate kernels (txfm_core.h / quant_core.h are shared): results are bit-identical to the chain.
#include "svt_hip_common.h"
#include "../../include/svtav1_hip.h"
#include "txfm_core.h"
#include "quant_core.h"

namespace {

template <typename PIX, int W, int H>
__global__ __launch_bounds__(256) void txfm_roundtrip_kernel(const int16_t* __restrict__ res_base, const PIX* __restrict__ pred_base, PIX* __restrict__ recon_base,
                                                             const SvtHipRoundtripDesc* __restrict__ descs, const uint32_t n, const int bd, const int qmode,
                                                             const SvtHipQuantParams* __restrict__ qparams, const int16_t* __restrict__ iscan_tables,
                                                             const uint8_t* __restrict__ qm_tables, const uint8_t* __restrict__ iqm_tables,
                                                             int32_t* __restrict__ qcoeff, int32_t* __restrict__ dqcoeff, uint16_t* __restrict__ eob_out) {
    constexpr int T = W > H ? W : H, BPW = 256 / T, PITCH = W + 1;
    constexpr int IW = W > 32 ? 32 : W, IH = H > 32 ? 32 : H, NCOEF = IW * IH; // what svt_handle_transform keeps of a 64-point block (transforms.c:2374-2542)
    constexpr int FS0 = fwd_shift0(W, H), FS1 = -fwd_shift1(W, H), FS2 = -fwd_shift2(W, H);
    constexpr int CBC = kFwdCosCol[ilog2c(W) - 2][ilog2c(H) - 2], CBR = kFwdCosRow[ilog2c(W) - 2][ilog2c(H) - 2];
    constexpr int IS0 = -inv_shift0(W, H);
    constexpr bool RECT1 = (W == 2 * H) || (H == 2 * W);
    HIP_DYNAMIC_SHARED(int32_t, smem)
    const int      tid = threadIdx.x, sub = tid / T, t = tid % T;
    const uint32_t blk = blockIdx.x * BPW + sub;
    const bool     active = blk < n;
    int32_t*       buf = smem + sub * (H * PITCH);
    const SvtHipRoundtripDesc d = descs[active ? blk : 0];
    const int tx = d.tx_type & 15;
    // ---- forward (av1_tranform_two_d_core_c, transforms.c:2259-2324): columns, transpose through LDS, rows
    if (active && t < W) {
        int32_t        v[H];
        const int16_t* in = res_base + d.in_off + t;
#pragma unroll
        for (int r = 0; r < H; r++) {
            const int rr = kUdFlip[tx] ? (H - 1 - r) : r;
            v[r]         = (int32_t)((uint32_t)(int32_t)in[(size_t)rr * d.in_stride] << FS0);
        }
        fwd1d<H, CBC>(kColKind[tx], v);
        const int cc = kLrFlip[tx] ? (W - 1 - t) : t;
#pragma unroll
        for (int r = 0; r < H; r++) buf[r * PITCH + cc] = FS1 ? rshift_round(v[r], FS1 ? FS1 : 1) : v[r];
    }
    __syncthreads();
    if (active && t < H) {
        int32_t v[W];
#pragma unroll
        for (int c = 0; c < W; c++) v[c] = buf[t * PITCH + c];
        fwd1d<W, CBR>(kRowKind[tx], v);
#pragma unroll
        for (int c = 0; c < W; c++) {
            int32_t x = FS2 ? rshift_round(v[c], FS2 ? FS2 : 1) : v[c];
            if (RECT1) x = mul_sqrt2_like(x, 5793);
            buf[t * PITCH + c] = x;
        }
    }
    __syncthreads();
    // ---- quantize / dequantize the kept IW x IH corner in place (full_loop.c:29-453); position rc = row * IW + column of the packed block
    uint32_t eob = 0;
    if (active) {
        const SvtHipQuantParams P   = qparams[d.qparam_idx];
        const int16_t*          isc = iscan_tables + (size_t)d.iscan_idx * NCOEF;
        const uint8_t*          qm  = qm_tables ? qm_tables + (size_t)d.qm_idx * NCOEF : nullptr;
        const uint8_t*          iqm = qm_tables ? iqm_tables + (size_t)d.qm_idx * NCOEF : nullptr;
        const size_t            base = (size_t)blk * NCOEF;
        for (int rc = t; rc < NCOEF; rc += T) {
            const int     r = rc / IW, c = rc % IW;
            const int32_t co = buf[r * PITCH + c];
            const int32_t wt = qm ? qm[rc] : (1 << QM_BITS), iwt = qm ? iqm[rc] : (1 << QM_BITS); // unit weights = the plain quantizers (exactly: see DESIGN.md 4.3)
            QOut o;
            if (qmode == 0) o = quant_one<0, true>(co, rc != 0, P, wt, iwt);
            else if (qmode == 1) o = quant_one<1, true>(co, rc != 0, P, wt, iwt);
            else if (qmode == 2) o = quant_one<2, true>(co, rc != 0, P, wt, iwt);
            else o = quant_one<3, true>(co, rc != 0, P, wt, iwt);
            qcoeff[base + rc] = o.q;
            if (dqcoeff) dqcoeff[base + rc] = o.dq;
            buf[r * PITCH + c] = o.dq;
            if (o.q != 0) {
                const uint32_t e = (uint32_t)isc[rc] + 1u;
                eob              = e > eob ? e : eob;
            }
        }
    }
#pragma unroll
    for (int m = T >> 1; m >= 1; m >>= 1) { // the T lanes of a block are contiguous inside one wave
        const uint32_t o = (uint32_t)__shfl_xor((int)eob, m);
        eob              = o > eob ? o : eob;
    }
    if (active && t == 0) eob_out[blk] = (uint16_t)eob;
    __syncthreads();
    // ---- inverse + reconstruction (inv_txfm2d_add_c, inv_transforms.c:2459-2535) from the dequantised corner
    const int32_t rhi = (1 << (bd + 7)) - 1, rlo = -(1 << (bd + 7));
    const int     cb  = (bd + 6 > 16) ? bd + 6 : 16;
    const int32_t chi = (1 << (cb - 1)) - 1, clo = -(1 << (cb - 1));
    if (active && t < H) { // (row t of the tile is read and rewritten by lane t alone)
        int32_t    vrow[W];
        const bool have = t < IH;
#pragma unroll
        for (int c = 0; c < W; c++) {
            int32_t x = (have && c < IW) ? buf[t * PITCH + c] : 0;
            if (RECT1) x = mul_sqrt2_like(x, 2896);
            vrow[c] = txfm1d::clamp_i32(x, rlo, rhi);
        }
        inv1d<W>(kRowKind[tx], vrow, rlo, rhi);
#pragma unroll
        for (int c = 0; c < W; c++) buf[t * PITCH + c] = IS0 ? rshift_round(vrow[c], IS0 ? IS0 : 1) : vrow[c];
    }
    __syncthreads();
    if (active && t < W) {
        int32_t   v[H];
        const int cc = kLrFlip[tx] ? (W - 1 - t) : t;
#pragma unroll
        for (int r = 0; r < H; r++) v[r] = txfm1d::clamp_i32(buf[r * PITCH + cc], clo, chi);
        inv1d<H>(kColKind[tx], v, clo, chi);
        const PIX*    pr = pred_base + d.pred_off + t;
        PIX*          rc = recon_base + d.recon_off + t;
        const int32_t mx = (1 << bd) - 1;
#pragma unroll
        for (int r = 0; r < H; r++) {
            const int     rr  = kUdFlip[tx] ? (H - 1 - r) : r;
            const int32_t res = rshift_round(v[r], 4);
            int32_t       px  = (int32_t)((uint32_t)pr[(size_t)rr * d.pred_stride] + (uint32_t)res);
            px                = px < 0 ? 0 : (px > mx ? mx : px);
            rc[(size_t)rr * d.recon_stride] = (PIX)px;
        }
    }
}

template <typename PIX, int W, int H>
void launch_rt(const int16_t* res, const PIX* pred, PIX* recon, const SvtHipRoundtripDesc* descs, uint32_t n, int bd, int qmode, const SvtHipQuantParams* qp,
               const int16_t* iscan, const uint8_t* qm, const uint8_t* iqm, int32_t* q, int32_t* dq, uint16_t* eob, hipStream_t st) {
    constexpr int T = W > H ? W : H, BPW = 256 / T;
    const size_t  shmem = (size_t)BPW * H * (W + 1) * 4;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(txfm_roundtrip_kernel<PIX, W, H>), dim3((n + BPW - 1) / BPW), dim3(256), shmem, st, res, pred, recon, descs, n, bd, qmode, qp,
                       iscan, qm, iqm, q, dq, eob);
    SVT_LAUNCH_CHECK();
}
#define FOR_ALL_TX_SIZES(X) \
    X(0, 4, 4) X(1, 8, 8) X(2, 16, 16) X(3, 32, 32) X(4, 64, 64) X(5, 4, 8) X(6, 8, 4) X(7, 8, 16) X(8, 16, 8) X(9, 16, 32) X(10, 32, 16) \
    X(11, 32, 64) X(12, 64, 32) X(13, 4, 16) X(14, 16, 4) X(15, 8, 32) X(16, 32, 8) X(17, 16, 64) X(18, 64, 16)

} // namespace

extern "C" void svt_hip_txfm_quant_roundtrip_batch(const int16_t* residual_base, const void* pred_base, void* recon_base, const SvtHipRoundtripDesc* descs, uint32_t n,
                                                   int tx_size, int bd, int quant_mode, const SvtHipQuantParams* qparams, const int16_t* iscan_tables,
                                                   const uint8_t* qm_tables, const uint8_t* iqm_tables, int32_t* qcoeff, int32_t* dqcoeff, uint16_t* eob, void* stream) {
    svthip::ensure_device();
    if (n == 0) return;
    if (quant_mode < 0 || quant_mode > 3 || (bd > 8) != ((quant_mode & 1) != 0)) {
        fprintf(stderr, "libsvtav1_hip: svt_hip_txfm_quant_roundtrip_batch: quant_mode %d does not match bit depth %d (0 / 2 = 8-bit, 1 / 3 = high bit depth)\n", quant_mode, bd);
        abort();
    }
    hipStream_t st = (hipStream_t)stream;
    if (bd > 8) {
        switch (tx_size) {
#define X(ID, W, H) case ID: launch_rt<uint16_t, W, H>(residual_base, (const uint16_t*)pred_base, (uint16_t*)recon_base, descs, n, bd, quant_mode, qparams, iscan_tables, qm_tables, iqm_tables, qcoeff, dqcoeff, eob, st); break;
            FOR_ALL_TX_SIZES(X)
#undef X
        default: fprintf(stderr, "libsvtav1_hip: bad tx_size %d\n", tx_size); abort();
        }
    } else {
        switch (tx_size) {
#define X(ID, W, H) case ID: launch_rt<uint8_t, W, H>(residual_base, (const uint8_t*)pred_base, (uint8_t*)recon_base, descs, n, 8, quant_mode, qparams, iscan_tables, qm_tables, iqm_tables, qcoeff, dqcoeff, eob, st); break;
            FOR_ALL_TX_SIZES(X)
#undef X
        default: fprintf(stderr, "libsvtav1_hip: bad tx_size %d\n", tx_size); abort();
        }
    }
}

SVT_HIP_DEFINE_WARM(txfm_fused) // (svt_hip_warmup loads this translation unit's code object at encoder initialisation: svt_hip_common.h)
