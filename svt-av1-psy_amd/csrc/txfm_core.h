// txfm_core.h -- shared device pieces of the 2-D transform kernels (txfm.hip, txfm_fused.hip): TxType tables, the per-size shift / cos_bit constants of
// transforms.h / inv_transforms.c, the 4-point ADST and identity kernels, and the dispatch into the generated 1-D flow graphs (txfm1d_gen.h).
#pragma once
#include "txfm1d_gen.h"

namespace {

enum { K_DCT = 0, K_ADST = 1, K_IDTX = 2 };
// TxType -> column kind / row kind / flips (vtx_tab, htx_tab, set_flip_cfg in inv_transforms.h)
__device__ constexpr uint8_t kColKind[16] = {K_DCT, K_ADST, K_DCT, K_ADST, K_ADST, K_DCT, K_ADST, K_ADST, K_ADST, K_IDTX, K_DCT, K_IDTX, K_ADST, K_IDTX, K_ADST, K_IDTX};
__device__ constexpr uint8_t kRowKind[16] = {K_DCT, K_DCT, K_ADST, K_ADST, K_DCT, K_ADST, K_ADST, K_ADST, K_ADST, K_IDTX, K_IDTX, K_DCT, K_IDTX, K_ADST, K_IDTX, K_ADST};
__device__ constexpr uint8_t kUdFlip[16]  = {0, 0, 0, 0, 1, 0, 1, 0, 1, 0, 0, 0, 0, 0, 1, 0};
__device__ constexpr uint8_t kLrFlip[16]  = {0, 0, 0, 0, 0, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 1};
// sinpi (svt_aom_eb_av1_sinpi_arr_data, inv_transforms.c:3228), bits 10..13
__device__ constexpr int32_t kSinpi[4][5] = {{0, 330, 621, 836, 951}, {0, 660, 1241, 1672, 1901}, {0, 1321, 2482, 3344, 3803}, {0, 2642, 4964, 6689, 7606}};

constexpr int ilog2c(int v) { return v <= 1 ? 0 : 1 + ilog2c(v >> 1); }
// transforms.h:27-50 as functions of (W, H)
constexpr int fwd_shift0(int w, int h) { return ((w == 64 && h == 64) || (w == 32 && h == 64) || (w == 16 && h == 64)) ? 0 : 2; }
constexpr int fwd_shift1(int w, int h) {
    const int m = w > h ? w : h, s = w < h ? w : h;
    if (w == 64 && h == 64) return -2;
    if ((w == 32 && h == 64) || (w == 16 && h == 64)) return -2;
    if (w == 64 && (h == 32 || h == 16)) return -4;
    if (m == 4) return 0;
    if (m == 8) return -1;
    if (m == 16) return s == 4 ? -1 : -2;
    return s == 8 ? -2 : -4; // m == 32
}
constexpr int fwd_shift2(int w, int h) { return ((w == 64 && h == 64) || (w == 32 && h == 64) || (w == 64 && h == 32)) ? -2 : 0; }
constexpr int kFwdCosCol[5][5] = {{13, 13, 13, 0, 0}, {13, 13, 13, 12, 0}, {13, 13, 13, 12, 13}, {0, 13, 13, 12, 13}, {0, 0, 13, 12, 13}};
constexpr int kFwdCosRow[5][5] = {{13, 13, 12, 0, 0}, {13, 13, 13, 12, 0}, {13, 13, 12, 13, 12}, {0, 12, 13, 12, 11}, {0, 0, 12, 11, 10}};
// inv_transforms.c:17-35
constexpr int inv_shift0(int w, int h) {
    const int m = w > h ? w : h, s = w < h ? w : h;
    if (w == h) return w == 4 ? 0 : (w == 8 ? -1 : -2);
    if (m == 8) return 0;                 // 4x8, 8x4
    if (m == 16) return -1;               // 8x16, 16x8, 4x16, 16x4
    if (m == 32) return s == 16 ? -1 : -2; // 16x32/32x16: -1 ; 8x32/32x8: -2
    return s == 32 ? -1 : -2;             // 32x64/64x32: -1 ; 16x64/64x16: -2
}
constexpr int INV_COS_BIT = 12;

__device__ __forceinline__ int32_t rshift_round(const int32_t x, const int bit) { // round_shift(), exact for every int32
    return (x >> bit) + ((x >> (bit - 1)) & 1);
}
__device__ __forceinline__ int32_t mul_sqrt2_like(const int32_t x, const int32_t k) { // round_shift((int64)x * k, 12)
    return (int32_t)(((int64_t)x * k + 2048) >> 12);
}
template <int CB> __device__ __forceinline__ int32_t rshift64_i32(const int32_t v) { return (int32_t)(((int64_t)v + ((int64_t)1 << (CB - 1))) >> CB); }

template <int CB> __device__ __forceinline__ void fadst4(int32_t (&v)[4]) { // transforms.c:1415-1502
    const int32_t  s1 = kSinpi[CB - 10][1], s2 = kSinpi[CB - 10][2], s3 = kSinpi[CB - 10][3], s4 = kSinpi[CB - 10][4];
    const uint32_t x0 = (uint32_t)v[0], x1 = (uint32_t)v[1], x2 = (uint32_t)v[2], x3 = (uint32_t)v[3];
    const uint32_t a0 = s1 * x0 + s2 * x1 + s4 * x3, a1 = s3 * (x0 + x1 - x3), a2 = s4 * x0 - s1 * x1 + s2 * x3, a3 = s3 * x2;
    v[0] = rshift64_i32<CB>((int32_t)(a0 + a3));
    v[1] = rshift64_i32<CB>((int32_t)a1);
    v[2] = rshift64_i32<CB>((int32_t)(a2 - a3));
    v[3] = rshift64_i32<CB>((int32_t)(a2 - a0 + a3));
}
template <int CB> __device__ __forceinline__ void iadst4(int32_t (&v)[4]) { // inv_transforms.c:722-800
    const int32_t  s1 = kSinpi[CB - 10][1], s2 = kSinpi[CB - 10][2], s3 = kSinpi[CB - 10][3], s4 = kSinpi[CB - 10][4];
    const uint32_t x0 = (uint32_t)v[0], x1 = (uint32_t)v[1], x2 = (uint32_t)v[2], x3 = (uint32_t)v[3];
    const uint32_t t0 = s1 * x0 + s4 * x2 + s2 * x3, t1 = s2 * x0 - s1 * x2 - s4 * x3, t3 = s3 * x1, t2 = s3 * (x0 - x2 + x3);
    v[0] = rshift64_i32<CB>((int32_t)(t0 + t3));
    v[1] = rshift64_i32<CB>((int32_t)(t1 + t3));
    v[2] = rshift64_i32<CB>((int32_t)t2);
    v[3] = rshift64_i32<CB>((int32_t)(t0 + t1 - t3));
}
template <int N> __device__ __forceinline__ void identity(int32_t (&v)[N]) { // transforms.c:2205-2234 / inv_transforms.c:2331-2366
#pragma unroll
    for (int i = 0; i < N; i++) {
        if (N == 4) v[i] = mul_sqrt2_like(v[i], 5793);
        else if (N == 8) v[i] = (int32_t)((uint32_t)v[i] * 2u);
        else if (N == 16) v[i] = mul_sqrt2_like(v[i], 2 * 5793);
        else if (N == 32) v[i] = (int32_t)((uint32_t)v[i] * 4u);
        else v[i] = mul_sqrt2_like(v[i], 4 * 5793);
    }
}
template <int N, int CB> __device__ __forceinline__ void fwd1d(const int kind, int32_t (&v)[N]) {
    if (kind == K_IDTX) {
        identity<N>(v);
    } else if (kind == K_DCT) {
        if constexpr (N == 4) txfm1d::fdct4<CB>(v);
        else if constexpr (N == 8) txfm1d::fdct8<CB>(v);
        else if constexpr (N == 16) txfm1d::fdct16<CB>(v);
        else if constexpr (N == 32) txfm1d::fdct32<CB>(v);
        else txfm1d::fdct64<CB>(v);
    } else {
        if constexpr (N == 4) fadst4<CB>(v);
        else if constexpr (N == 8) txfm1d::fadst8<CB>(v);
        else if constexpr (N == 16) txfm1d::fadst16<CB>(v);
    }
}
// ADST32: also compile the 32-point inverse ADST (av1_iadst32_new, inv_transforms.c:1119-1552) -- a kernel no AV1 stream reaches (the 32-point dimensions allow
// DCT and identity only, TxfmCommon.h:160-209) but that the reference's `_c` functions compute and its InvTxfm2dAddTest fixture feeds.  Its live set takes the
// 32x32 inverse kernel from 71 to 125 VGPRs (7 -> 4 waves per SIMD), so the throughput kernels are built without it and the `_any_type` entry points with it.
template <int N, bool ADST32 = false> __device__ __forceinline__ void inv1d(const int kind, int32_t (&v)[N], const int32_t lo, const int32_t hi) {
    if (kind == K_IDTX) {
        identity<N>(v);
    } else if (kind == K_DCT) {
        if constexpr (N == 4) txfm1d::idct4<INV_COS_BIT>(v, lo, hi);
        else if constexpr (N == 8) txfm1d::idct8<INV_COS_BIT>(v, lo, hi);
        else if constexpr (N == 16) txfm1d::idct16<INV_COS_BIT>(v, lo, hi);
        else if constexpr (N == 32) txfm1d::idct32<INV_COS_BIT>(v, lo, hi);
        else txfm1d::idct64<INV_COS_BIT>(v, lo, hi);
    } else {
        if constexpr (N == 4) iadst4<INV_COS_BIT>(v);
        else if constexpr (N == 8) txfm1d::iadst8<INV_COS_BIT>(v, lo, hi);
        else if constexpr (N == 16) txfm1d::iadst16<INV_COS_BIT>(v, lo, hi);
        else if constexpr (N == 32 && ADST32) txfm1d::iadst32<INV_COS_BIT>(v, lo, hi);
    }
}

} // namespace
