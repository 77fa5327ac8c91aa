// cdef.hip -- CDEF direction search, filter and strength-search distortion for gfx950.
//
// Frame kernel: one workgroup per (64x64 filter block, chunk of strength candidates).  The block and its 3-row /
// 8-column halo are staged once into LDS as u16 (CDEF_VERY_LARGE outside the frame, cdef_process.c:208-228); a quad of
// lanes owns one 8x8 unit: the quad finds the unit's direction (each lane evaluates two orthogonal directions with
// fully static partial-sum indices, so `var = best - cost[orthogonal]` is lane-local), then every lane filters two
// rows.  Search mode never materialises filtered pixels: the variance-weighted luma distortion / chroma MSE
// (enc_cdef.c:23-219) is accumulated from registers, so HBM traffic is one read of the block per candidate chunk
// and 8 bytes out per (block, strength).  Apply mode writes out of place (the reference's in-place line/column buffers,
// enc_cdef.c:334-335, exist only because it overwrites its input).
#include "svt_hip_common.h"
#include "../../include/svtav1_hip.h"

namespace {

constexpr int VERY_LARGE = 0x7f7f; // CDEF_VERY_LARGE (cdef.h:38)
constexpr int VB = 3, HB = 8;      // CDEF_VBORDER / CDEF_HBORDER
constexpr int CAND_CHUNK = 8;

__device__ __forceinline__ int msb_u32(uint32_t n) { return 31 - __clz(n); }

// constrain() of cdef.c:85-91 with the damping shift hoisted: shift = max(0, damping - msb(threshold))
__device__ __forceinline__ int constrain_s(const int diff, const int threshold, const int shift) {
    const int ad = diff < 0 ? -diff : diff;
    int       v  = threshold - (ad >> shift);
    v            = v < 0 ? 0 : v;
    v            = ad < v ? ad : v;
    return diff < 0 ? -v : v;
}
// Cdef_Directions (cdef.c:99-120) as (dy, dx) of tap k
__device__ __forceinline__ int dir_off(const int dir, const int k, const int pitch) {
    const int d  = dir & 7;
    const int dy = k == 0 ? (d == 0 ? -1 : (d >= 4 ? 1 : 0)) : (d == 0 ? -2 : (d == 1 ? -1 : (d == 2 ? 0 : (d == 3 ? 1 : 2))));
    const int dx = k == 0 ? (d <= 4 ? 1 : 0) : (d <= 4 ? 2 : (d == 5 ? 1 : (d == 6 ? 0 : -1)));
    return dy * pitch + dx;
}

struct FilterCtx {
    int pri, sec, pri_shift, sec_shift, pt0, pt1, st0, st1;
    int po[2], s0o[2], s1o[2];
};
__device__ __forceinline__ FilterCtx make_ctx(const int pri_strength, const int sec_strength, const int dir, const int pri_damping, const int sec_damping,
                                              const int coeff_shift, const int pitch) {
    FilterCtx c;
    c.pri = pri_strength;
    c.sec = sec_strength;
    int s = pri_strength ? pri_damping - msb_u32((uint32_t)pri_strength) : 0;
    c.pri_shift = s < 0 ? 0 : s;
    s           = sec_strength ? sec_damping - msb_u32((uint32_t)sec_strength) : 0;
    c.sec_shift = s < 0 ? 0 : s;
    const int odd = (pri_strength >> coeff_shift) & 1; // svt_aom_eb_cdef_pri_taps / sec_taps (cdef.c:249-250)
    c.pt0 = odd ? 3 : 4; c.pt1 = odd ? 3 : 2; c.st0 = 2; c.st1 = 1;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        c.po[k]  = dir_off(dir, k, pitch);
        c.s0o[k] = dir_off(dir + 2, k, pitch);
        c.s1o[k] = dir_off(dir + 6, k, pitch);
    }
    return c;
}
// one pixel of svt_cdef_filter_block_c (cdef.c:253-306); `p` points at the pixel inside a u16 tile
__device__ __forceinline__ int filter_px(const uint16_t* p, const FilterCtx& c) {
    const int x = (int16_t)p[0];
    int sum = 0, mx = x, mn = x;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int pt = k ? c.pt1 : c.pt0, st = k ? c.st1 : c.st0;
        const int v[6] = {(int16_t)p[c.po[k]], (int16_t)p[-c.po[k]], (int16_t)p[c.s0o[k]], (int16_t)p[-c.s0o[k]], (int16_t)p[c.s1o[k]], (int16_t)p[-c.s1o[k]]};
#pragma unroll
        for (int t = 0; t < 6; t++) {
            if (t < 2) { if (c.pri) sum += pt * constrain_s(v[t] - x, c.pri, c.pri_shift); }
            else       { if (c.sec) sum += st * constrain_s(v[t] - x, c.sec, c.sec_shift); }
            if (v[t] != VERY_LARGE) mx = v[t] > mx ? v[t] : mx;
            mn = v[t] < mn ? v[t] : mn;
        }
    }
    sum   = (int16_t)sum; // the reference accumulates in int16 (cdef.c:265); |sum| <= 12 * 4 * 240 so this never truncates
    int y = x + ((8 + sum - (sum < 0)) >> 4);
    return y < mn ? mn : (y > mx ? mx : y);
}
__device__ __forceinline__ int adjust_strength(const int strength, const int var) { // cdef.c:130-134
    const int v6 = var >> 6;
    int       i  = v6 ? msb_u32((uint32_t)v6) : 0;
    i            = i > 12 ? 12 : i;
    return var ? (strength * (4 + i) + 8) >> 4 : 0;
}

// cost of direction D for one 8x8 unit (svt_aom_cdef_find_dir_c, cdef.c:150-199); all partial-sum indices are static
template <int D> __device__ __forceinline__ int dir_cost(const uint16_t* img, const int pitch, const int coeff_shift) {
    int partial[15];
#pragma unroll
    for (int i = 0; i < 15; i++) partial[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int x = ((int)img[i * pitch + j] >> coeff_shift) - 128;
            constexpr int dummy = 0;
            (void)dummy;
            const int idx = D == 0 ? i + j : D == 1 ? i + j / 2 : D == 2 ? i : D == 3 ? 3 + i - j / 2 : D == 4 ? 7 + i - j : D == 5 ? 3 - i / 2 + j : D == 6 ? j : i / 2 + j;
            partial[idx] += x;
        }
    int cost = 0;
    if (D == 2 || D == 6) {
#pragma unroll
        for (int i = 0; i < 8; i++) cost += partial[i] * partial[i];
        cost *= 105;
    } else if (D == 0 || D == 4) {
        constexpr int div[8] = {840, 420, 280, 210, 168, 140, 120, 105};
#pragma unroll
        for (int i = 0; i < 7; i++) cost += (partial[i] * partial[i] + partial[14 - i] * partial[14 - i]) * div[i];
        cost += partial[7] * partial[7] * 105;
    } else {
#pragma unroll
        for (int j = 0; j < 5; j++) cost += partial[3 + j] * partial[3 + j];
        cost *= 105;
        constexpr int div2[3] = {420, 210, 140};
#pragma unroll
        for (int j = 0; j < 3; j++) cost += (partial[j] * partial[j] + partial[10 - j] * partial[10 - j]) * div2[j];
    }
    return cost;
}
// lane q of a quad evaluates directions q and q+4; returns best dir / var in every lane of the quad
__device__ __forceinline__ void quad_find_dir(const uint16_t* img, const int pitch, const int coeff_shift, const int q, int& best_dir, int& var) {
    int a, b;
    if (q == 0) { a = dir_cost<0>(img, pitch, coeff_shift); b = dir_cost<4>(img, pitch, coeff_shift); }
    else if (q == 1) { a = dir_cost<1>(img, pitch, coeff_shift); b = dir_cost<5>(img, pitch, coeff_shift); }
    else if (q == 2) { a = dir_cost<2>(img, pitch, coeff_shift); b = dir_cost<6>(img, pitch, coeff_shift); }
    else { a = dir_cost<3>(img, pitch, coeff_shift); b = dir_cost<7>(img, pitch, coeff_shift); }
    // first maximum in index order, strict '>' starting from best_cost = 0 (cdef.c:200-205)
    int c = a, d = q, o = b;
    if (b > a) { c = b; d = q + 4; o = a; }
#pragma unroll
    for (int m = 1; m <= 2; m <<= 1) {
        const int c2 = __shfl_xor(c, m), d2 = __shfl_xor(d, m), o2 = __shfl_xor(o, m);
        if (c2 > c || (c2 == c && d2 < d)) { c = c2; d = d2; o = o2; }
    }
    // (costs are sums of squares, so "nothing beat the initial best_cost = 0" means every cost is 0: dir 0, var 0 -- same result)
    best_dir = d;
    var      = (c - o) >> 10;
}

template <typename PIX, int MODE>
__global__ __launch_bounds__(256) void cdef_frame_kernel(const SvtHipCdefParams P) {
    HIP_DYNAMIC_SHARED(uint16_t, tile_raw)
    __shared__ int                sh_dir[64], sh_var[64], sh_active[64], sh_count;
    __shared__ unsigned long long sh_mse[CAND_CHUNK];
    const int tid = threadIdx.x;
    const int xdec = P.xdec, ydec = P.ydec, pli = P.pli, cs = P.coeff_shift;
    const int bw = 64 >> xdec, bh = 64 >> ydec, uw = 8 >> xdec, uh = 8 >> ydec;
    const int pw = (int)P.width, ph = (int)P.height;
    const int nhfb = (pw + bw - 1) / bw, nvfb = (ph + bh - 1) / bh;
    const int fb = blockIdx.x, fbr = fb / nhfb, fbc = fb % nhfb;
    const int pitch = bw + 2 * HB;
    uint16_t* in = tile_raw + VB * pitch + HB;
    const int c0 = blockIdx.y * CAND_CHUNK;
    const int c1 = MODE == 1 ? ((c0 + CAND_CHUNK) < (int)P.ncand ? (c0 + CAND_CHUNK) : (int)P.ncand) : 1;

    if (tid == 0) sh_count = 0;
    if (tid < CAND_CHUNK) sh_mse[tid] = 0;
    __syncthreads();
    if (tid < 64) {
        const int by = tid >> 3, bx = tid & 7;
        const bool inside = (fbc * 8 + bx) * uw < pw && (fbr * 8 + by) * uh < ph;
        const int  act    = inside && !P.skip[(size_t)(fbr * 8 + by) * (nhfb * 8) + fbc * 8 + bx];
        sh_active[tid]    = act;
        if (act) atomicAdd(&sh_count, 1);
    }
    __syncthreads();
    if (sh_count == 0) {
        if (MODE == 1)
            for (int c = c0 + tid; c < c1; c += 256) P.mse[(size_t)fb * P.ncand + c] = 0;
        return;
    }
    {   // stage the tile (cdef_process.c:208-228): real pixels where the neighbouring filter block exists, else VERY_LARGE
        const PIX* plane = (const PIX*)P.recon;
        const int x0 = fbc * bw, y0 = fbr * bh;
        const int xs = x0 - (fbc != 0 ? HB : 0), ys = y0 - (fbr != 0 ? VB : 0);
        int       xe = (x0 + bw < pw ? x0 + bw : pw) + (fbc + 1 < nhfb ? HB : 0);
        int       ye = (y0 + bh < ph ? y0 + bh : ph) + (fbr + 1 < nvfb ? VB : 0);
        const int total = (bh + 2 * VB) * pitch;
        for (int i = tid; i < total; i += 256) {
            const int r = i / pitch, c = i - r * pitch;
            const int gy = y0 - VB + r, gx = x0 - HB + c;
            const bool ok = gx >= xs && gx < xe && gy >= ys && gy < ye;
            tile_raw[i]   = ok ? (uint16_t)plane[(size_t)gy * P.recon_stride + gx] : (uint16_t)VERY_LARGE;
        }
    }
    __syncthreads();
    const int b = tid >> 2, q = tid & 3, by = b >> 3, bx = b & 7;
    const int act = sh_active[b];
    if (pli == 0) {
        int d = 0, v = 0;
        // a whole quad is either active or not, so the quad shuffles inside are convergent per quad; inactive quads
        // still execute them (results unused) to keep the wave convergent
        quad_find_dir(in + (by * 8) * pitch + bx * 8, pitch, cs, q, d, v);
        if (q == 0) {
            sh_dir[b] = act ? d : 0;
            sh_var[b] = act ? v : 0;
            if (blockIdx.y == 0) { P.dir[(size_t)fb * 64 + b] = (uint8_t)(act ? d : 0); P.var[(size_t)fb * 64 + b] = act ? v : 0; }
        }
    } else if (q == 0) {
        int d = P.dir[(size_t)fb * 64 + b];
        if (xdec != ydec) { // cdef.c:388-395
            const int conv422[8] = {7, 0, 2, 4, 5, 6, 6, 6}, conv440[8] = {1, 2, 2, 2, 3, 4, 6, 0};
            d = xdec ? conv422[d & 7] : conv440[d & 7];
        }
        sh_dir[b] = d;
        sh_var[b] = P.var[(size_t)fb * 64 + b];
    }
    __syncthreads();
    const int sub = MODE == 1 ? P.subsampling : 1;
    for (int c = c0; c < c1; c++) {
        const int level = MODE == 1 ? P.pri[c] : P.pri[fb];
        const int secl  = MODE == 1 ? P.sec[c] : P.sec[fb];
        if (MODE == 0 && level == 0 && secl == 0) break; // zero strength leaves the block unchanged (enc_cdef.c:571)
        const int pri_strength = level << cs, sec_strength = secl << cs;
        const int pdamp = P.pri_damping + cs - (pli != 0), sdamp = P.sec_damping + cs - (pli != 0);
        const int t = pli ? pri_strength : adjust_strength(pri_strength, sh_var[b]);
        const FilterCtx ctx = make_ctx(t, sec_strength, pri_strength ? sh_dir[b] : 0, pdamp, sdamp, cs, pitch);
        unsigned long long ss = 0, sd = 0, ss2 = 0, sd2 = 0, ssd = 0, mse = 0;
        if (act) {
            const int rows = uh >> 2; // rows per lane: 2 (8-row unit) or 1 (4-row unit)
            for (int rr = 0; rr < rows; rr++) {
                const int r = uh == 8 ? 2 * q + rr : q;
                if (r >= uh || (r % sub) != 0) continue;
                const uint16_t* row = in + (by * uh + r) * pitch + bx * uw;
                const size_t    gy  = (size_t)(fbr * bh + by * uh + r);
                const int       gx  = fbc * bw + bx * uw;
                for (int j = 0; j < uw; j++) {
                    const int y = filter_px(row + j, ctx);
                    if (MODE == 0) {
                        ((PIX*)P.out)[gy * P.out_stride + gx + j] = (PIX)y;
                    } else {
                        const unsigned dpx = ((const PIX*)P.source)[gy * P.source_stride + gx + j], s = (unsigned)y;
                        ss += s; sd += dpx; ss2 += s * s; sd2 += dpx * dpx; ssd += s * dpx;
                        const int e = (int)dpx - (int)s;
                        mse += (unsigned long long)(long long)(e * e);
                    }
                }
            }
        }
        if (MODE == 1) {
            // reduce the five sums over the quad (values < 2^32: 64 * 4095^2), then one lane evaluates the block distortion
#pragma unroll
            for (int m = 1; m <= 2; m <<= 1) {
                ss += (unsigned)__shfl_xor((int)(unsigned)ss, m); sd += (unsigned)__shfl_xor((int)(unsigned)sd, m);
                ss2 += (unsigned)__shfl_xor((int)(unsigned)ss2, m); sd2 += (unsigned)__shfl_xor((int)(unsigned)sd2, m);
                ssd += (unsigned)__shfl_xor((int)(unsigned)ssd, m); mse += (unsigned)__shfl_xor((int)(unsigned)mse, m);
            }
            if (act && q == 0) {
                unsigned long long dist = mse;
                if (pli == 0 && uw == 8 && uh == 8) { // dist_8xn_*_c, enc_cdef.c:23-48: IEEE double, no contraction (-ffp-contract=off)
                    const unsigned long long svar = ss2 - ((ss * ss + 32) >> 6), dvar = sd2 - ((sd * sd + 32) >> 6);
                    const double num = (double)(sd2 + ss2 - 2 * ssd) * .5 * (double)(svar + dvar + (unsigned long long)(400 << 2 * cs));
                    const double den = sqrt((double)(20000 << 4 * cs) + (double)svar * (double)dvar);
                    dist = (unsigned long long)floor(.5 + num / den);
                }
                atomicAdd(&sh_mse[c - c0], dist);
            }
        }
    }
    if (MODE == 1) {
        __syncthreads();
        if (tid < c1 - c0) P.mse[(size_t)fb * P.ncand + c0 + tid] = sh_mse[tid] >> (2 * cs);
    }
}

// ---- single-call kernels ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void find_dir_kernel(const uint16_t* img /* n blocks of 8x8, pitch 8 */, int n, int coeff_shift, int* out /* dir,var pairs */) {
    const int tid = threadIdx.x, b = tid >> 2, q = tid & 3;
    int d = 0, v = 0;
    quad_find_dir(img + (b < n ? b : 0) * 64, 8, coeff_shift, q, d, v);
    if (q == 0 && b < n) { out[2 * b] = d; out[2 * b + 1] = v; }
}
template <typename PIX>
__global__ __launch_bounds__(64) void filter_block_kernel(PIX* dst, int dstride, const uint16_t* tile /* (bh+4) x (bw+4), pitch bw+4, origin (2,2) */, int pri,
                                                          int sec, int dir, int pdamp, int sdamp, int bw, int bh, int cs, int sub) {
    const int pitch = bw + 4, i = threadIdx.x >> 3, j = threadIdx.x & 7;
    if (i >= bh || j >= bw || (i % sub) != 0) return;
    const FilterCtx ctx = make_ctx(pri, sec, dir, pdamp, sdamp, cs, pitch);
    dst[i * dstride + j] = (PIX)filter_px(tile + (i + 2) * pitch + 2 + j, ctx);
}
template <typename PIX>
__global__ __launch_bounds__(64) void cdef_dist_kernel(const PIX* plane /* packed per block like `packed` */, const PIX* packed, int count, int bw, int bh, int cs,
                                                       int pli, int sub, unsigned long long* out) {
    __shared__ unsigned long long total;
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    for (int bi = threadIdx.x; bi < count; bi += 64) {
        unsigned long long ss = 0, sd = 0, ss2 = 0, sd2 = 0, ssd = 0, mse = 0;
        for (int i = 0; i < bh; i += sub)
            for (int j = 0; j < bw; j++) {
                const unsigned d = plane[bi * bw * bh + i * bw + j], s = packed[bi * bw * bh + i * bw + j];
                ss += s; sd += d; ss2 += s * s; sd2 += d * d; ssd += s * d;
                const int e = (int)d - (int)s;
                mse += (unsigned long long)(long long)(e * e);
            }
        unsigned long long dist = mse;
        if (pli == 0 && bw == 8 && bh == 8) {
            const unsigned long long svar = ss2 - ((ss * ss + 32) >> 6), dvar = sd2 - ((sd * sd + 32) >> 6);
            const double num = (double)(sd2 + ss2 - 2 * ssd) * .5 * (double)(svar + dvar + (unsigned long long)(400 << 2 * cs));
            const double den = sqrt((double)(20000 << 4 * cs) + (double)svar * (double)dvar);
            dist = (unsigned long long)floor(.5 + num / den);
        }
        atomicAdd(&total, dist);
    }
    __syncthreads();
    if (threadIdx.x == 0) out[0] = total >> (2 * cs);
}
__global__ void copy_rect8_to_16_kernel(uint16_t* dst, const uint8_t* src, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

template <int MODE> void launch_frame(const SvtHipCdefParams& P, hipStream_t st) {
    const int bw = 64 >> P.xdec, bh = 64 >> P.ydec;
    const int nhfb = ((int)P.width + bw - 1) / bw, nvfb = ((int)P.height + bh - 1) / bh;
    const size_t shmem = (size_t)(bh + 2 * VB) * (bw + 2 * HB) * 2 + 64;
    const dim3 grid(nhfb * nvfb, MODE == 1 ? (P.ncand + CAND_CHUNK - 1) / CAND_CHUNK : 1);
    if (P.is_16bit) hipLaunchKernelGGL(HIP_KERNEL_NAME(cdef_frame_kernel<uint16_t, MODE>), grid, dim3(256), shmem, st, P);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(cdef_frame_kernel<uint8_t, MODE>), grid, dim3(256), shmem, st, P);
    SVT_LAUNCH_CHECK();
}
const int kBlkW[4] = {4, 4, 8, 8}, kBlkH[4] = {4, 8, 4, 8}; // BLOCK_4X4, 4X8, 8X4, 8X8 (definitions.h)

} // namespace

extern "C" {

void svt_hip_cdef_frame(int mode, const SvtHipCdefParams* params, void* stream) {
    svthip::ensure_device();
    if (mode == 1 && params->ncand == 0) return;
    if (mode == 0) launch_frame<0>(*params, (hipStream_t)stream);
    else launch_frame<1>(*params, (hipStream_t)stream);
}

uint8_t svt_aom_cdef_find_dir_hip(const uint16_t* img, int32_t stride, int32_t* var, int32_t coeff_shift) {
    svthip::HostCall& c = svthip::host_call();
    c.begin();
    c.reserve(4096, 4096);
    uint16_t* d = (uint16_t*)c.dalloc(128);
    int*      o = (int*)c.dalloc(8);
    c.up2d(d, 16, img, (size_t)stride * 2, 16, 8);
    hipLaunchKernelGGL(find_dir_kernel, dim3(1), dim3(64), 0, c.stream, (const uint16_t*)d, 1, coeff_shift, o);
    SVT_LAUNCH_CHECK();
    int h[2];
    c.down(h, o, 8);
    *var = h[1];
    return (uint8_t)h[0];
}
void svt_aom_cdef_find_dir_dual_hip(const uint16_t* img1, const uint16_t* img2, int stride, int32_t* var1, int32_t* var2, int32_t coeff_shift,
                                    uint8_t* out1, uint8_t* out2) {
    svthip::HostCall& c = svthip::host_call();
    c.begin();
    c.reserve(4096, 4096);
    uint16_t* d = (uint16_t*)c.dalloc(256);
    int*      o = (int*)c.dalloc(16);
    c.up2d(d, 16, img1, (size_t)stride * 2, 16, 8);
    c.up2d(d + 64, 16, img2, (size_t)stride * 2, 16, 8);
    hipLaunchKernelGGL(find_dir_kernel, dim3(1), dim3(64), 0, c.stream, (const uint16_t*)d, 2, coeff_shift, o);
    SVT_LAUNCH_CHECK();
    int h[4];
    c.down(h, o, 16);
    *out1 = (uint8_t)h[0]; *var1 = h[1]; *out2 = (uint8_t)h[2]; *var2 = h[3];
}

void svt_cdef_filter_block_hip(uint8_t* dst8, uint16_t* dst16, int32_t dstride, const uint16_t* in, int32_t pri_strength, int32_t sec_strength, int32_t dir,
                               int32_t pri_damping, int32_t sec_damping, int32_t bsize, int32_t coeff_shift, uint8_t subsampling_factor) {
    const int bw = kBlkW[bsize & 3], bh = kBlkH[bsize & 3], pitch = bw + 4;
    svthip::HostCall& c = svthip::host_call();
    c.begin();
    c.reserve(8192, 8192);
    uint16_t* dt = (uint16_t*)c.dalloc((size_t)(bh + 4) * pitch * 2);
    void*     dd = c.dalloc(8 * 8 * 2);
    // the taps reach at most 2 pixels in every direction (Cdef_Directions): upload that neighbourhood of the 144-pitch tile
    c.up2d(dt, (size_t)pitch * 2, in - 2 * 144 - 2, 144 * 2, (size_t)pitch * 2, bh + 4);
    const size_t px = dst8 ? 1 : 2;
    c.up2d(dd, 8 * px, dst8 ? (void*)dst8 : (void*)dst16, (size_t)dstride * px, (size_t)bw * px, bh); // rows skipped by subsampling keep their content
    if (dst8)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(filter_block_kernel<uint8_t>), dim3(1), dim3(64), 0, c.stream, (uint8_t*)dd, 8, (const uint16_t*)dt, pri_strength,
                           sec_strength, dir, pri_damping, sec_damping, bw, bh, coeff_shift, (int)subsampling_factor);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(filter_block_kernel<uint16_t>), dim3(1), dim3(64), 0, c.stream, (uint16_t*)dd, 8, (const uint16_t*)dt, pri_strength,
                           sec_strength, dir, pri_damping, sec_damping, bw, bh, coeff_shift, (int)subsampling_factor);
    SVT_LAUNCH_CHECK();
    c.down2d(dst8 ? (void*)dst8 : (void*)dst16, (size_t)dstride * px, dd, 8 * px, (size_t)bw * px, bh);
}

static uint64_t cdef_dist_host(const void* dst, int32_t dstride, const void* src, const uint8_t* dlist, int32_t cdef_count, int bsize, int32_t coeff_shift,
                               int32_t pli, uint8_t sub, int px) {
    if (cdef_count <= 0) return 0;
    const int bw = kBlkW[bsize & 3], bh = kBlkH[bsize & 3];
    svthip::HostCall& c = svthip::host_call();
    c.begin();
    const size_t blk = (size_t)bw * bh * px;
    c.reserve(2 * blk * cdef_count + 4096, 3 * blk * cdef_count + 4096);
    uint8_t* dp = (uint8_t*)c.dalloc(blk * cdef_count);
    uint8_t* ds = (uint8_t*)c.dalloc(blk * cdef_count);
    unsigned long long* o = (unsigned long long*)c.dalloc(8);
    // gather the picture blocks named by dlist into the packed layout of `src`
    uint8_t* hp = (uint8_t*)c.palloc(blk * cdef_count);
    for (int bi = 0; bi < cdef_count; bi++) {
        const int by = dlist[2 * bi], bx = dlist[2 * bi + 1];
        for (int i = 0; i < bh; i++)
            memcpy(hp + bi * blk + (size_t)i * bw * px, (const uint8_t*)dst + ((size_t)(by * bh + i) * dstride + bx * bw) * px, (size_t)bw * px);
    }
    HIP_CHECK(hipMemcpyAsync(dp, hp, blk * cdef_count, hipMemcpyHostToDevice, c.stream));
    c.up(ds, src, blk * cdef_count);
    if (px == 2)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(cdef_dist_kernel<uint16_t>), dim3(1), dim3(64), 0, c.stream, (const uint16_t*)dp, (const uint16_t*)ds, cdef_count, bw, bh,
                           coeff_shift, pli, (int)sub, o);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(cdef_dist_kernel<uint8_t>), dim3(1), dim3(64), 0, c.stream, (const uint8_t*)dp, (const uint8_t*)ds, cdef_count, bw, bh,
                           coeff_shift, pli, (int)sub, o);
    SVT_LAUNCH_CHECK();
    uint64_t r;
    c.down(&r, o, 8);
    return r;
}
uint64_t svt_compute_cdef_dist_16bit_hip(const uint16_t* dst, int32_t dstride, const uint16_t* src, const void* dlist, int32_t cdef_count, uint8_t bsize,
                                         int32_t coeff_shift, int32_t pli, uint8_t subsampling_factor) {
    return cdef_dist_host(dst, dstride, src, (const uint8_t*)dlist, cdef_count, bsize, coeff_shift, pli, subsampling_factor, 2);
}
uint64_t svt_compute_cdef_dist_8bit_hip(const uint8_t* dst8, int32_t dstride, const uint8_t* src8, const void* dlist, int32_t cdef_count, uint8_t bsize,
                                        int32_t coeff_shift, int32_t pli, uint8_t subsampling_factor) {
    return cdef_dist_host(dst8, dstride, src8, (const uint8_t*)dlist, cdef_count, bsize, coeff_shift, pli, subsampling_factor, 1);
}
void svt_aom_copy_rect8_8bit_to_16bit_hip(uint16_t* dst, int32_t dstride, const uint8_t* src, int32_t sstride, int32_t v, int32_t h) {
    if (v <= 0 || h <= 0) return;
    svthip::HostCall& c = svthip::host_call();
    c.begin();
    const size_t n = (size_t)v * h;
    c.reserve(n * 3 + 4096, n * 3 + 4096);
    uint8_t*  ds = (uint8_t*)c.dalloc(n);
    uint16_t* dd = (uint16_t*)c.dalloc(n * 2);
    c.up2d(ds, h, src, sstride, h, v);
    hipLaunchKernelGGL(copy_rect8_to_16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c.stream, dd, (const uint8_t*)ds, (int)n);
    SVT_LAUNCH_CHECK();
    c.down2d(dst, (size_t)dstride * 2, dd, (size_t)h * 2, (size_t)h * 2, v);
}

} // extern "C"
