// txfm.hip -- forward / inverse 2-D AV1 transforms for gfx950.
//
// Bit-exactness forces the reference's butterfly networks (one rounding shift per rotation, half_btf), so the 1-D
// kernels are the generated straight-line flow graphs of txfm1d_gen.h (tools/gen_txfm.py), NOT dense contractions:
// MFMA would round once per output instead of once per rotation and produce different LSBs.
//
// Mapping: one lane owns one column (pass 1) and then one row (pass 2) with the whole 1-D vector in VGPRs; the
// transpose goes through an LDS tile with row pitch W+1 (conflict-free both ways).  max(W,H) lanes per TX block,
// 256/max(W,H) blocks per workgroup; global loads/stores are lane-contiguous (coefficients leave through LDS so the
// stores are linear).  Forward: av1_tranform_two_d_core_c (Source/Lib/Codec/transforms.c:2259-2324); inverse:
// inv_txfm2d_add_c (inv_transforms.c:2459-2535).
#include "svt_hip_common.h"
#include "../../include/svtav1_hip.h"
#include "txfm_core.h"

namespace {

// ---- forward ----------------------------------------------------------------------------------------------------
template <int W, int H>
__global__ __launch_bounds__(256) void fwd_txfm2d_kernel(const int16_t* __restrict__ base, const SvtHipFwdTxfmDesc* __restrict__ descs,
                                                         const uint32_t n, const int pf, int32_t* __restrict__ out) {
    constexpr int T = W > H ? W : H, BPW = 256 / T, PITCH = W + 1;
    constexpr int S0 = fwd_shift0(W, H), S1 = -fwd_shift1(W, H), S2 = -fwd_shift2(W, H);
    constexpr int CBC = kFwdCosCol[ilog2c(W) - 2][ilog2c(H) - 2], CBR = kFwdCosRow[ilog2c(W) - 2][ilog2c(H) - 2];
    constexpr bool RECT1 = (W == 2 * H) || (H == 2 * W);
    HIP_DYNAMIC_SHARED(int32_t, smem)
    const int      tid = threadIdx.x, sub = tid / T, t = tid % T;
    const uint32_t blk = blockIdx.x * BPW + sub;
    const bool     active = blk < n;
    int32_t*       buf = smem + sub * (H * PITCH);
    const SvtHipFwdTxfmDesc d = descs[active ? blk : 0];
    const int tx = d.tx_type & 15;
    if (active && t < W) {
        int32_t        v[H];
        const int16_t* in = base + d.in_off + t;
#pragma unroll
        for (int r = 0; r < H; r++) {
            const int rr = kUdFlip[tx] ? (H - 1 - r) : r;
            v[r]         = (int32_t)((uint32_t)(int32_t)in[(size_t)rr * d.in_stride] << S0);
        }
        fwd1d<H, CBC>(kColKind[tx], v);
        const int cc = kLrFlip[tx] ? (W - 1 - t) : t;
#pragma unroll
        for (int r = 0; r < H; r++) buf[r * PITCH + cc] = S1 ? rshift_round(v[r], S1 ? S1 : 1) : v[r];
    }
    __syncthreads();
    if (active && t < H) {
        int32_t v[W];
#pragma unroll
        for (int c = 0; c < W; c++) v[c] = buf[t * PITCH + c];
        fwd1d<W, CBR>(kRowKind[tx], v);
        const int kw = W >> pf, kh = H >> pf; // partial-frequency shapes keep the top-left corner only (transforms.c:5202-5273)
#pragma unroll
        for (int c = 0; c < W; c++) {
            int32_t x = S2 ? rshift_round(v[c], S2 ? S2 : 1) : v[c];
            if (RECT1) x = mul_sqrt2_like(x, 5793);
            buf[t * PITCH + c] = (t < kh && c < kw) ? x : 0;
        }
    }
    __syncthreads();
    if (active) {
        int32_t* o = out + (size_t)blk * (W * H);
        for (int i = t; i < W * H; i += T) o[i] = buf[(i / W) * PITCH + (i % W)];
    }
}

// ---- inverse -----------------------------------------------------------------------------------------------------
template <typename PIX, int W, int H, bool ADST32 = false>
__global__ __launch_bounds__(256) void inv_txfm2d_kernel(const int32_t* __restrict__ coeff_base, const PIX* __restrict__ pred_base,
                                                         PIX* __restrict__ recon_base, const SvtHipInvTxfmDesc* __restrict__ descs,
                                                         const uint32_t n, const int bd) {
    constexpr int T = W > H ? W : H, BPW = 256 / T, PITCH = W + 1;
    constexpr int IW = W > 32 ? 32 : W, IH = H > 32 ? 32 : H; // 64-point inputs arrive packed 32 wide / 32 high (inv_transforms.c:2567-2580)
    constexpr int S0 = -inv_shift0(W, H);
    constexpr bool RECT1 = (W == 2 * H) || (H == 2 * W);
    HIP_DYNAMIC_SHARED(int32_t, smem)
    const int      tid = threadIdx.x, sub = tid / T, t = tid % T;
    const uint32_t blk = blockIdx.x * BPW + sub;
    const bool     active = blk < n;
    int32_t*       buf = smem + sub * (H * PITCH);
    const SvtHipInvTxfmDesc d = descs[active ? blk : 0];
    const int tx = d.tx_type & 15;
    const int32_t rhi = (1 << (bd + 7)) - 1, rlo = -(1 << (bd + 7));                   // clamp to bd + 8 bits
    const int     cb  = (bd + 6 > 16) ? bd + 6 : 16;
    const int32_t chi = (1 << (cb - 1)) - 1, clo = -(1 << (cb - 1));
    if (active) {
        const int32_t* in = coeff_base + d.coeff_off;
        for (int i = t; i < IW * IH; i += T) buf[(i / IW) * PITCH + (i % IW)] = in[i];
    }
    __syncthreads();
    if (active && t < H) {
        int32_t v[W];
        const bool have = t < IH;
#pragma unroll
        for (int c = 0; c < W; c++) {
            int32_t x = (have && c < IW) ? buf[t * PITCH + c] : 0;
            if (RECT1) x = mul_sqrt2_like(x, 2896);
            v[c] = txfm1d::clamp_i32(x, rlo, rhi);
        }
        inv1d<W, ADST32>(kRowKind[tx], v, rlo, rhi);
#pragma unroll
        for (int c = 0; c < W; c++) buf[t * PITCH + c] = S0 ? rshift_round(v[c], S0 ? S0 : 1) : v[c];
    }
    __syncthreads();
    if (active && t < W) {
        int32_t   v[H];
        const int cc = kLrFlip[tx] ? (W - 1 - t) : t;
#pragma unroll
        for (int r = 0; r < H; r++) v[r] = txfm1d::clamp_i32(buf[r * PITCH + cc], clo, chi);
        inv1d<H, ADST32>(kColKind[tx], v, clo, chi);
        const PIX*    pr = pred_base + d.pred_off + t;
        PIX*          rc = recon_base + d.recon_off + t;
        const int32_t mx = (1 << bd) - 1;
#pragma unroll
        for (int r = 0; r < H; r++) {
            // up-down flip (inv_transforms.c:2509-2527): output row rr takes v[H - 1 - rr]; flip the ADDRESS, never the register index
            // (a run-time index into v[] turns into an H-way select chain per row)
            const int     rr  = kUdFlip[tx] ? (H - 1 - r) : r;
            const int32_t res = rshift_round(v[r], 4);
            int32_t       px  = (int32_t)((uint32_t)pr[(size_t)rr * d.pred_stride] + (uint32_t)res);
            px                = px < 0 ? 0 : (px > mx ? mx : px);
            rc[(size_t)rr * d.recon_stride] = (PIX)px;
        }
    }
}


// ---- 4x4 Walsh-Hadamard (lossless mode) ---------------------------------------------------------------------------
// svt_av1_fwht4x4_c (transforms.c:3099-3152) and svt_av1_highbd_iwht4x4_{16,1}_add_c (inv_transforms.c:2735-2825): sixteen values,
// a handful of adds -- one lane owns a whole block.  The forward keeps the reference's odd output order (a, c, d, b).
__global__ __launch_bounds__(64) void fwht4x4_kernel(const int16_t* __restrict__ base, const SvtHipFwdTxfmDesc* __restrict__ descs, const uint32_t n,
                                                     int32_t* __restrict__ out) {
    const uint32_t blk = blockIdx.x * 64 + threadIdx.x;
    if (blk >= n) return;
    const SvtHipFwdTxfmDesc d  = descs[blk];
    const int16_t*          in = base + d.in_off;
    int32_t                 t[16];
#pragma unroll
    for (int i = 0; i < 4; i++) { // pass 0: column i of the input -> row i of t
        int32_t a1 = in[0 * (size_t)d.in_stride + i], b1 = in[1 * (size_t)d.in_stride + i], c1 = in[2 * (size_t)d.in_stride + i],
                d1 = in[3 * (size_t)d.in_stride + i];
        a1 += b1;
        d1 -= c1;
        const int32_t e1 = (a1 - d1) >> 1;
        b1 = e1 - b1;
        c1 = e1 - c1;
        a1 -= c1;
        d1 += b1;
        t[4 * i + 0] = a1; t[4 * i + 1] = c1; t[4 * i + 2] = d1; t[4 * i + 3] = b1;
    }
    int32_t* o = out + (size_t)blk * 16;
#pragma unroll
    for (int i = 0; i < 4; i++) { // pass 1: down column i of t
        int32_t a1 = t[i], b1 = t[4 + i], c1 = t[8 + i], d1 = t[12 + i];
        a1 += b1;
        d1 -= c1;
        const int32_t e1 = (a1 - d1) >> 1;
        b1 = e1 - b1;
        c1 = e1 - c1;
        a1 -= c1;
        d1 += b1;
        o[i] = a1 * 4; o[4 + i] = c1 * 4; o[8 + i] = d1 * 4; o[12 + i] = b1 * 4; // UNIT_QUANT_FACTOR
    }
}

template <typename PIX>
__global__ __launch_bounds__(64) void iwht4x4_add_kernel(const int32_t* __restrict__ coeff_base, const PIX* __restrict__ pred_base, PIX* __restrict__ recon_base,
                                                         const SvtHipInvTxfmDesc* __restrict__ descs, const uint32_t n, const int bd) {
    const uint32_t blk = blockIdx.x * 64 + threadIdx.x;
    if (blk >= n) return;
    const SvtHipInvTxfmDesc d  = descs[blk];
    const int32_t*          ip = coeff_base + d.coeff_off;
    int32_t                 t[16];
    if (d.wht_full) { // eob > 1: svt_av1_highbd_iwht4x4_16_add_c
#pragma unroll
        for (int i = 0; i < 4; i++) {
            int32_t a1 = ip[4 * i + 0] >> 2, c1 = ip[4 * i + 1] >> 2, d1 = ip[4 * i + 2] >> 2, b1 = ip[4 * i + 3] >> 2; // UNIT_QUANT_SHIFT
            a1 += c1;
            d1 -= b1;
            const int32_t e1 = (a1 - d1) >> 1;
            b1 = e1 - b1;
            c1 = e1 - c1;
            a1 -= b1;
            d1 += c1;
            t[4 * i + 0] = a1; t[4 * i + 1] = b1; t[4 * i + 2] = c1; t[4 * i + 3] = d1;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            int32_t a1 = t[i], c1 = t[4 + i], d1 = t[8 + i], b1 = t[12 + i];
            a1 += c1;
            d1 -= b1;
            const int32_t e1 = (a1 - d1) >> 1;
            b1 = e1 - b1;
            c1 = e1 - c1;
            a1 -= b1;
            d1 += c1;
            t[i] = a1; t[4 + i] = b1; t[8 + i] = c1; t[12 + i] = d1;
        }
    } else { // eob <= 1: svt_av1_highbd_iwht4x4_1_add_c -- only the DC term, which matters (not an optimisation) for lossless
        int32_t       a1 = ip[0] >> 2;
        const int32_t e1 = a1 >> 1;
        a1 -= e1;
        const int32_t row[4] = {a1, e1, e1, e1};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int32_t e = row[i] >> 1, a = row[i] - e;
            t[i] = a; t[4 + i] = e; t[8 + i] = e; t[12 + i] = e;
        }
    }
    const int32_t mx = (1 << bd) - 1;
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int c = 0; c < 4; c++) {
            int32_t px = (int32_t)pred_base[d.pred_off + (size_t)r * d.pred_stride + c] + t[4 * r + c];
            px         = px < 0 ? 0 : (px > mx ? mx : px);
            recon_base[d.recon_off + (size_t)r * d.recon_stride + c] = (PIX)px;
        }
}

constexpr int kTxW[19] = {4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64};
constexpr int kTxH[19] = {4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16};

template <int W, int H> void launch_fwd(const int16_t* base, const SvtHipFwdTxfmDesc* descs, uint32_t n, int pf, int32_t* out, hipStream_t st) {
    constexpr int T = W > H ? W : H, BPW = 256 / T;
    const size_t  shmem = (size_t)BPW * H * (W + 1) * 4;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(fwd_txfm2d_kernel<W, H>), dim3((n + BPW - 1) / BPW), dim3(256), shmem, st, base, descs, n, pf, out);
    SVT_LAUNCH_CHECK();
}
template <typename PIX, int W, int H, bool ANY>
void launch_inv(const int32_t* coeff, const PIX* pred, PIX* recon, const SvtHipInvTxfmDesc* descs, uint32_t n, int bd, hipStream_t st) {
    constexpr int  T = W > H ? W : H, BPW = 256 / T;
    constexpr bool ADST32 = ANY && (W == 32 || H == 32); // the other sizes have one kernel: every type their `_c` function computes is in it
    const size_t   shmem = (size_t)BPW * H * (W + 1) * 4;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(inv_txfm2d_kernel<PIX, W, H, ADST32>), dim3((n + BPW - 1) / BPW), dim3(256), shmem, st, coeff, pred, recon, descs, n, bd);
    SVT_LAUNCH_CHECK();
}
#define FOR_ALL_TX_SIZES(X) \
    X(0, 4, 4) X(1, 8, 8) X(2, 16, 16) X(3, 32, 32) X(4, 64, 64) X(5, 4, 8) X(6, 8, 4) X(7, 8, 16) X(8, 16, 8) X(9, 16, 32) X(10, 32, 16) \
    X(11, 32, 64) X(12, 64, 32) X(13, 4, 16) X(14, 16, 4) X(15, 8, 32) X(16, 32, 8) X(17, 16, 64) X(18, 64, 16)

template <typename PIX, bool ANY = false> void inv_dispatch(const int32_t* coeff, const PIX* pred, PIX* recon, const SvtHipInvTxfmDesc* descs, uint32_t n,
                                                            int tx_size, int bd, hipStream_t st) {
    switch (tx_size) {
#define X(ID, W, H) case ID: launch_inv<PIX, W, H, ANY>(coeff, pred, recon, descs, n, bd, st); break;
        FOR_ALL_TX_SIZES(X)
#undef X
    default: fprintf(stderr, "libsvtav1_hip: bad tx_size %d\n", tx_size); abort();
    }
}

} // namespace

extern "C" {

void svt_hip_fwd_txfm2d_batch(const int16_t* residual_base, const SvtHipFwdTxfmDesc* descs, uint32_t n, int tx_size, int bit_depth,
                              int pf_shape, int32_t* coeff_out, void* stream) {
    (void)bit_depth; // only feeds range asserts in the reference (transforms.c:2275)
    svthip::ensure_device();
    if (n == 0) return;
    hipStream_t st = (hipStream_t)stream;
    switch (tx_size) {
#define X(ID, W, H) case ID: launch_fwd<W, H>(residual_base, descs, n, pf_shape, coeff_out, st); break;
        FOR_ALL_TX_SIZES(X)
#undef X
    default: fprintf(stderr, "libsvtav1_hip: bad tx_size %d\n", tx_size); abort();
    }
}

void svt_hip_inv_txfm2d_add_batch(const int32_t* coeff_base, const uint16_t* pred_base, uint16_t* recon_base, const SvtHipInvTxfmDesc* descs,
                                  uint32_t n, int tx_size, int bd, void* stream) {
    svthip::ensure_device();
    if (n == 0) return;
    inv_dispatch<uint16_t>(coeff_base, pred_base, recon_base, descs, n, tx_size, bd, (hipStream_t)stream);
}
void svt_hip_inv_txfm2d_add_batch_u8(const int32_t* coeff_base, const uint8_t* pred_base, uint8_t* recon_base, const SvtHipInvTxfmDesc* descs,
                                     uint32_t n, int tx_size, void* stream) {
    svthip::ensure_device();
    if (n == 0) return;
    inv_dispatch<uint8_t>(coeff_base, pred_base, recon_base, descs, n, tx_size, 8, (hipStream_t)stream);
}

void svt_hip_inv_txfm2d_add_batch_any_type(const int32_t* coeff_base, const uint16_t* pred_base, uint16_t* recon_base, const SvtHipInvTxfmDesc* descs,
                                           uint32_t n, int tx_size, int bd, void* stream) {
    svthip::ensure_device();
    if (n == 0) return;
    inv_dispatch<uint16_t, true>(coeff_base, pred_base, recon_base, descs, n, tx_size, bd, (hipStream_t)stream);
}
void svt_hip_inv_txfm2d_add_batch_any_type_u8(const int32_t* coeff_base, const uint8_t* pred_base, uint8_t* recon_base, const SvtHipInvTxfmDesc* descs,
                                              uint32_t n, int tx_size, void* stream) {
    svthip::ensure_device();
    if (n == 0) return;
    inv_dispatch<uint8_t, true>(coeff_base, pred_base, recon_base, descs, n, tx_size, 8, (hipStream_t)stream);
}

void svt_hip_fwht4x4_batch(const int16_t* residual_base, const SvtHipFwdTxfmDesc* descs, uint32_t n, int32_t* coeff_out, void* stream) {
    svthip::ensure_device();
    if (n == 0) return;
    hipLaunchKernelGGL(fwht4x4_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, residual_base, descs, n, coeff_out);
    SVT_LAUNCH_CHECK();
}
void svt_hip_iwht4x4_add_batch(const int32_t* coeff_base, const uint16_t* pred_base, uint16_t* recon_base, const SvtHipInvTxfmDesc* descs, uint32_t n,
                               int bd, void* stream) {
    svthip::ensure_device();
    if (n == 0) return;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(iwht4x4_add_kernel<uint16_t>), dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, coeff_base, pred_base,
                       recon_base, descs, n, bd);
    SVT_LAUNCH_CHECK();
}
void svt_hip_iwht4x4_add_batch_u8(const int32_t* coeff_base, const uint8_t* pred_base, uint8_t* recon_base, const SvtHipInvTxfmDesc* descs, uint32_t n,
                                  void* stream) {
    svthip::ensure_device();
    if (n == 0) return;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(iwht4x4_add_kernel<uint8_t>), dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, coeff_base, pred_base,
                       recon_base, descs, n, 8);
    SVT_LAUNCH_CHECK();
}

// ---- RTCD-signature single-call forms --------------------------------------------------------------------------------
// svt_av1_fwht4x4 (aom_dsp_rtcd.h:208) -> svt_av1_fwht4x4_c (transforms.c:3099)
void svt_av1_fwht4x4_hip(int16_t* input, int32_t* output, uint32_t stride) {
    svthip::HostCall& c = svthip::host_call();
    c.begin_small();
    c.reserve(4096, 4096);
    int16_t*           din  = (int16_t*)c.dalloc(4 * 16);
    SvtHipFwdTxfmDesc* dd   = (SvtHipFwdTxfmDesc*)c.dalloc(sizeof(SvtHipFwdTxfmDesc));
    int32_t*           dout = (int32_t*)c.dalloc(64);
    c.up2d(din, 16, input, (size_t)stride * 2, 8, 4);
    SvtHipFwdTxfmDesc d;
    memset(&d, 0, sizeof(d));
    d.in_stride = 8;
    c.up(dd, &d, sizeof(d));
    svt_hip_fwht4x4_batch(din, dd, 1, dout, c.stream);
    c.down(output, dout, 64);
}

void svt_av1_fwd_txfm2d_hip(int16_t* input, int32_t* output, uint32_t input_stride, int tx_type, int tx_size, uint8_t bit_depth, int pf) {
    const int w = kTxW[tx_size], h = kTxH[tx_size];
    svthip::HostCall& c = svthip::host_call();
    c.begin_small();
    const size_t pitch = svthip::align_up((size_t)w * 2, 16);
    c.reserve(pitch * h + (size_t)w * h * 4 + 4096, pitch * h + (size_t)w * h * 4 + 4096);
    int16_t*           din = (int16_t*)c.dalloc(pitch * h);
    SvtHipFwdTxfmDesc* dd  = (SvtHipFwdTxfmDesc*)c.dalloc(sizeof(SvtHipFwdTxfmDesc));
    int32_t*           dout = (int32_t*)c.dalloc((size_t)w * h * 4);
    c.up2d(din, pitch, input, (size_t)input_stride * 2, (size_t)w * 2, h);
    SvtHipFwdTxfmDesc d;
    memset(&d, 0, sizeof(d));
    d.in_stride = (uint32_t)(pitch / 2);
    d.tx_type   = (uint8_t)tx_type;
    c.up(dd, &d, sizeof(d));
    svt_hip_fwd_txfm2d_batch(din, dd, 1, tx_size, bit_depth, pf, dout, c.stream);
    c.down(output, dout, (size_t)w * h * 4);
}

void svt_av1_inv_txfm2d_add_hip(const int32_t* input, uint16_t* output_r, int32_t stride_r, uint16_t* output_w, int32_t stride_w, int tx_type,
                                int tx_size, int32_t bd) {
    const int w = kTxW[tx_size], h = kTxH[tx_size];
    const int iw = w > 32 ? 32 : w, ih = h > 32 ? 32 : h;
    svthip::HostCall& c = svthip::host_call();
    c.begin_small();
    const size_t pitch = svthip::align_up((size_t)w * 2, 16);
    c.reserve((size_t)iw * ih * 4 + 2 * pitch * h + 4096, (size_t)iw * ih * 4 + 3 * pitch * h + 4096);
    int32_t*           dco = (int32_t*)c.dalloc((size_t)iw * ih * 4);
    uint16_t*          dpr = (uint16_t*)c.dalloc(pitch * h);
    uint16_t*          drc = (uint16_t*)c.dalloc(pitch * h);
    SvtHipInvTxfmDesc* dd  = (SvtHipInvTxfmDesc*)c.dalloc(sizeof(SvtHipInvTxfmDesc));
    c.up(dco, input, (size_t)iw * ih * 4);
    c.up2d(dpr, pitch, output_r, (size_t)stride_r * 2, (size_t)w * 2, h);
    SvtHipInvTxfmDesc d;
    memset(&d, 0, sizeof(d));
    d.pred_stride = d.recon_stride = (uint32_t)(pitch / 2);
    d.tx_type = (uint8_t)tx_type;
    c.up(dd, &d, sizeof(d));
    svt_hip_inv_txfm2d_add_batch_any_type(dco, dpr, drc, dd, 1, tx_size, bd, c.stream);
    c.down2d(output_w, (size_t)stride_w * 2, drc, pitch, (size_t)w * 2, h);
}

// svt_av1_inv_txfm_add -> svt_av1_inv_txfm_add_c (inv_transforms.c:3177-3192): 8-bit destination form.  lossless != 0 with TX_4X4 selects the
// Walsh-Hadamard inverse (svt_av1_highbd_inv_txfm_add_4x4, inv_transforms.c:2833-2848), whose eob <= 1 form differs from the eob > 1 form.
void svt_av1_inv_txfm_add_u8_hip(const int32_t* dqcoeff, uint8_t* dst_r, int32_t stride_r, uint8_t* dst_w, int32_t stride_w, int tx_type,
                                 int tx_size, int lossless, int eob) {
    const int w = kTxW[tx_size], h = kTxH[tx_size];
    const int iw = w > 32 ? 32 : w, ih = h > 32 ? 32 : h;
    svthip::HostCall& c = svthip::host_call();
    c.begin_small();
    const size_t pitch = svthip::align_up((size_t)w, 16);
    c.reserve((size_t)iw * ih * 4 + 2 * pitch * h + 4096, (size_t)iw * ih * 4 + 3 * pitch * h + 4096);
    int32_t*           dco = (int32_t*)c.dalloc((size_t)iw * ih * 4);
    uint8_t*           dpr = (uint8_t*)c.dalloc(pitch * h);
    uint8_t*           drc = (uint8_t*)c.dalloc(pitch * h);
    SvtHipInvTxfmDesc* dd  = (SvtHipInvTxfmDesc*)c.dalloc(sizeof(SvtHipInvTxfmDesc));
    c.up(dco, dqcoeff, (size_t)iw * ih * 4);
    c.up2d(dpr, pitch, dst_r, (size_t)stride_r, (size_t)w, h);
    SvtHipInvTxfmDesc d;
    memset(&d, 0, sizeof(d));
    d.pred_stride = d.recon_stride = (uint32_t)pitch;
    d.tx_type  = (uint8_t)tx_type;
    d.wht_full = eob > 1;
    c.up(dd, &d, sizeof(d));
    if (lossless && tx_size == 0) svt_hip_iwht4x4_add_batch_u8(dco, dpr, drc, dd, 1, c.stream);
    else svt_hip_inv_txfm2d_add_batch_any_type_u8(dco, dpr, drc, dd, 1, tx_size, c.stream);
    c.down2d(dst_w, (size_t)stride_w, drc, pitch, (size_t)w, h);
}
// the pointer's exact prototype (common_dsp_rtcd.h:144); SvtHipTxfmParam == TxfmParam (definitions.h:1043-1055).  bd / is_hbd are fixed by the
// 8-bit destination (svt_av1_inv_txfm_add_c widens to u16 and narrows again: identical to clipping at 255 for bd = 8, which is what every caller
// passes -- svt_aom_inv_transform_recon8bit, inv_transforms.c:3089-3113); tx_set_type is unused by the reference as well.
void svt_av1_inv_txfm_add_hip(const int32_t* dqcoeff, uint8_t* dst_r, int32_t stride_r, uint8_t* dst_w, int32_t stride_w, const SvtHipTxfmParam* p) {
    if (p->bd != 8) {
        // No caller passes this (svt_aom_inv_transform_recon8bit sets bd = 8), but the reference function is defined for it: widen the 8-bit destination
        // to u16, run the high-bit-depth inverse with p->bd (clip at 2^bd - 1), narrow by truncation -- exactly svt_av1_inv_txfm_add_c's three steps.
        const int w = kTxW[p->tx_size], h = kTxH[p->tx_size];
        const int iw = w > 32 ? 32 : w, ih = h > 32 ? 32 : h;
        svthip::HostCall& c = svthip::host_call();
        c.begin();
        const size_t pitch = svthip::align_up((size_t)w * 2, 16);
        c.reserve((size_t)iw * ih * 4 + 2 * pitch * h + 4096, (size_t)iw * ih * 4 + 3 * pitch * h + 4096);
        int32_t*           dco = (int32_t*)c.dalloc((size_t)iw * ih * 4);
        uint16_t*          dpr = (uint16_t*)c.dalloc(pitch * h);
        uint16_t*          drc = (uint16_t*)c.dalloc(pitch * h);
        SvtHipInvTxfmDesc* dd  = (SvtHipInvTxfmDesc*)c.dalloc(sizeof(SvtHipInvTxfmDesc));
        c.up(dco, dqcoeff, (size_t)iw * ih * 4);
        uint16_t* wide = (uint16_t*)c.palloc(pitch * h);
        for (int r = 0; r < h; r++)
            for (int x = 0; x < w; x++) wide[r * (pitch / 2) + x] = dst_r[r * stride_r + x];
        HIP_CHECK(hipMemcpyAsync(dpr, wide, pitch * h, hipMemcpyHostToDevice, c.stream));
        SvtHipInvTxfmDesc d;
        memset(&d, 0, sizeof(d));
        d.pred_stride = d.recon_stride = (uint32_t)(pitch / 2);
        d.tx_type  = (uint8_t)p->tx_type;
        d.wht_full = p->eob > 1;
        c.up(dd, &d, sizeof(d));
        if (p->lossless && p->tx_size == 0) svt_hip_iwht4x4_add_batch(dco, dpr, drc, dd, 1, p->bd, c.stream);
        else svt_hip_inv_txfm2d_add_batch_any_type(dco, dpr, drc, dd, 1, p->tx_size, p->bd, c.stream);
        uint16_t* back = (uint16_t*)c.palloc(pitch * h);
        HIP_CHECK(hipMemcpyAsync(back, drc, pitch * h, hipMemcpyDeviceToHost, c.stream));
        c.sync();
        for (int r = 0; r < h; r++)
            for (int x = 0; x < w; x++) dst_w[r * stride_w + x] = (uint8_t)back[r * (pitch / 2) + x];
        return;
    }
    svt_av1_inv_txfm_add_u8_hip(dqcoeff, dst_r, stride_r, dst_w, stride_w, p->tx_type, p->tx_size, p->lossless, p->eob);
}

// the 19 (+38 partial-frequency) forward and 19 inverse fixed-size symbols of the RTCD tables
#define X(ID, W, H)                                                                                                                       \
    void svt_av1_fwd_txfm2d_##W##x##H##_hip(int16_t* input, int32_t* output, uint32_t stride, uint8_t tx_type, uint8_t bd) {                    \
        svt_av1_fwd_txfm2d_hip(input, output, stride, tx_type, ID, bd, 0);                                                               \
    }                                                                                                                                     \
    void svt_av1_fwd_txfm2d_##W##x##H##_N2_hip(int16_t* input, int32_t* output, uint32_t stride, uint8_t tx_type, uint8_t bd) {                 \
        svt_av1_fwd_txfm2d_hip(input, output, stride, tx_type, ID, bd, 1);                                                               \
    }                                                                                                                                     \
    void svt_av1_fwd_txfm2d_##W##x##H##_N4_hip(int16_t* input, int32_t* output, uint32_t stride, uint8_t tx_type, uint8_t bd) {                 \
        svt_av1_fwd_txfm2d_hip(input, output, stride, tx_type, ID, bd, 2);                                                               \
    }
FOR_ALL_TX_SIZES(X)
#undef X
// inverse: squares (input, r, stride_r, w, stride_w, tx_type, bd); 4x8/8x4/4x16/16x4 add tx_size; the rest add tx_size, eob
// (common_dsp_rtcd.h:106-116, inv_transforms.c:2545-2716)
#define INV_SQ(ID, N)                                                                                                                     \
    void svt_av1_inv_txfm2d_add_##N##x##N##_hip(const int32_t* in, uint16_t* r, int32_t sr, uint16_t* w, int32_t sw, uint8_t tx_type, int32_t bd) { \
        svt_av1_inv_txfm2d_add_hip(in, r, sr, w, sw, tx_type, ID, bd);                                                                    \
    }
INV_SQ(0, 4) INV_SQ(1, 8) INV_SQ(2, 16) INV_SQ(3, 32) INV_SQ(4, 64)
#define INV_R1(ID, W, H)                                                                                                                  \
    void svt_av1_inv_txfm2d_add_##W##x##H##_hip(const int32_t* in, uint16_t* r, int32_t sr, uint16_t* w, int32_t sw, uint8_t tx_type, uint8_t tx_size, \
                                                int32_t bd) {                                                                             \
        (void)tx_size;                                                                                                                    \
        svt_av1_inv_txfm2d_add_hip(in, r, sr, w, sw, tx_type, ID, bd);                                                                    \
    }
INV_R1(5, 4, 8) INV_R1(6, 8, 4) INV_R1(13, 4, 16) INV_R1(14, 16, 4)
#define INV_R2(ID, W, H)                                                                                                                  \
    void svt_av1_inv_txfm2d_add_##W##x##H##_hip(const int32_t* in, uint16_t* r, int32_t sr, uint16_t* w, int32_t sw, uint8_t tx_type, uint8_t tx_size, \
                                                int32_t eob, int32_t bd) {                                                                \
        (void)tx_size; (void)eob;                                                                                                         \
        svt_av1_inv_txfm2d_add_hip(in, r, sr, w, sw, tx_type, ID, bd);                                                                    \
    }
INV_R2(7, 8, 16) INV_R2(8, 16, 8) INV_R2(9, 16, 32) INV_R2(10, 32, 16) INV_R2(11, 32, 64) INV_R2(12, 64, 32) INV_R2(15, 8, 32)
INV_R2(16, 32, 8) INV_R2(17, 16, 64) INV_R2(18, 64, 16)

} // extern "C"

SVT_HIP_DEFINE_WARM(txfm) // (svt_hip_warmup loads this translation unit's code object at encoder initialisation: svt_hip_common.h)
