// picprep.hip -- producing ME's inputs on the device: the 2x2 zero-phase decimation that builds the 1/4 and 1/16
// luma planes (downsample_2d -> svt_aom_downsample_2d_c, pic_analysis_process.c:130-160; driver
// svt_aom_downsample_filtering_input_picture, :2138-2200) and picture border replication (svt_aom_generate_padding,
// pic_operators.c:397-441).  HBM-bound byte kernels: every thread produces four horizontally adjacent output pixels.
//
// The decimated planes are written WITH their borders in the same launch: a padded pixel is the decimation evaluated at the clamped
// interior coordinate, which is exactly what "decimate, then replicate edges" leaves there.
#include "svt_hip_common.h"
#include "../../include/svtav1_hip.h"

namespace {

struct __attribute__((packed, aligned(1))) U32A1 { uint32_t v; };
struct __attribute__((packed, aligned(1))) U64A1 { uint32_t lo, hi; };

// out(x, y) = (in[r-1][c-1] + in[r-1][c] + in[r][c-1] + in[r][c] + 2) >> 2 with r = y * step + step / 2, c = x * step + step / 2
__device__ __forceinline__ uint32_t box4(const uint8_t* in, const uint32_t stride, const int x, const int y, const int step) {
    const int      r = y * step + (step >> 1), c = x * step + (step >> 1);
    const uint8_t* p = in + (size_t)(r - 1) * stride + (c - 1);
    return ((uint32_t)p[0] + p[1] + p[stride] + p[stride + 1] + 2u) >> 2;
}

// grid: x = groups of 4 output pixels over the padded width, y = padded rows
__global__ __launch_bounds__(256) void downsample_padded_kernel(const uint8_t* __restrict__ in, const uint32_t in_stride, const int ow, const int oh,
                                                                uint8_t* __restrict__ out, const uint32_t out_stride, const int pad_x, const int pad_y,
                                                                const int step) {
    const int gx = (blockIdx.x * blockDim.x + threadIdx.x) * 4; // first padded-plane column of this thread
    const int py = blockIdx.y;
    const int pw = ow + 2 * pad_x;
    if (gx >= pw) return;
    int y = py - pad_y;
    y     = y < 0 ? 0 : (y >= oh ? oh - 1 : y);
    uint32_t px[4];
    const int x0 = gx - pad_x;
    if (step == 2 && x0 >= 0 && x0 + 4 <= ow) { // interior: 8 input bytes from each of two rows, any alignment
        const uint8_t* p  = in + (size_t)(2 * y) * in_stride + 2 * x0;
        const U64A1    a  = *(const U64A1*)p, b = *(const U64A1*)(p + in_stride);
        // per byte lane: a + b of vertically adjacent pixels (max 510: needs 9 bits) -> widen even / odd bytes separately
        const uint32_t ae0 = a.lo & 0x00ff00ffu, ao0 = (a.lo >> 8) & 0x00ff00ffu, be0 = b.lo & 0x00ff00ffu, bo0 = (b.lo >> 8) & 0x00ff00ffu;
        const uint32_t ae1 = a.hi & 0x00ff00ffu, ao1 = (a.hi >> 8) & 0x00ff00ffu, be1 = b.hi & 0x00ff00ffu, bo1 = (b.hi >> 8) & 0x00ff00ffu;
        const uint32_t s0 = ((ae0 + ao0 + be0 + bo0 + 0x00020002u) >> 2) & 0x00ff00ffu; // outputs 0 (low half) and 1 (high half)
        const uint32_t s1 = ((ae1 + ao1 + be1 + bo1 + 0x00020002u) >> 2) & 0x00ff00ffu; // outputs 2 and 3
        px[0] = s0 & 0xffu; px[1] = s0 >> 16; px[2] = s1 & 0xffu; px[3] = s1 >> 16;
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int x = x0 + k;
            x     = x < 0 ? 0 : (x >= ow ? ow - 1 : x);
            px[k] = box4(in, in_stride, x, y, step);
        }
    }
    uint8_t* o = out + (size_t)py * out_stride + gx;
    if (gx + 4 <= pw) *(U32A1*)o = U32A1{px[0] | (px[1] << 8) | (px[2] << 16) | (px[3] << 24)};
    else
        for (int k = 0; k < pw - gx; k++) o[k] = (uint8_t)px[k];
}

// plain decimation into an unpadded destination (the RTCD single-call form)
__global__ __launch_bounds__(256) void downsample_kernel(const uint8_t* __restrict__ in, const uint32_t in_stride, const int ow, const int oh,
                                                         uint8_t* __restrict__ out, const uint32_t out_stride, const int step) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x < ow && y < oh) out[(size_t)y * out_stride + x] = (uint8_t)box4(in, in_stride, x, y, step);
}

// svt_aom_generate_padding: left / right replication of every picture row, then whole padded rows copied up and down.  One thread per
// border byte; the value is the picture pixel at the clamped coordinate (identical result, no ordering between the two phases needed).
__global__ __launch_bounds__(256) void generate_padding_kernel(uint8_t* __restrict__ base, const uint32_t stride, const int w, const int h, const int pad_x,
                                                               const int pad_y) {
    const int pw = w + 2 * pad_x;
    const int gx = blockIdx.x * blockDim.x + threadIdx.x, gy = blockIdx.y;
    if (gx >= pw) return;
    const int x = gx - pad_x, y = gy - pad_y;
    if (x >= 0 && x < w && y >= 0 && y < h) return; // picture interior: untouched
    const int cx = x < 0 ? 0 : (x >= w ? w - 1 : x), cy = y < 0 ? 0 : (y >= h ? h - 1 : y);
    base[(size_t)gy * stride + gx] = base[(size_t)(cy + pad_y) * stride + cx + pad_x];
}

inline int decim_dim(uint32_t n, uint32_t step) { // number of indices half, half + step, ... < n
    const uint32_t half = step >> 1;
    return n > half ? (int)((n - half + step - 1) / step) : 0;
}

} // namespace

extern "C" {

void svt_hip_downsample_2d_padded(const uint8_t* in_origin, uint32_t in_stride, uint32_t in_width, uint32_t in_height, uint8_t* out_base,
                                  uint32_t out_stride, uint32_t pad_x, uint32_t pad_y, uint32_t step, void* stream) {
    svthip::ensure_device();
    const int ow = decim_dim(in_width, step), oh = decim_dim(in_height, step);
    if (ow <= 0 || oh <= 0) return;
    const int pw = ow + 2 * (int)pad_x, ph = oh + 2 * (int)pad_y;
    hipLaunchKernelGGL(downsample_padded_kernel, dim3(((pw + 3) / 4 + 255) / 256, ph), dim3(256), 0, (hipStream_t)stream, in_origin, in_stride, ow, oh,
                       out_base, out_stride, (int)pad_x, (int)pad_y, (int)step);
    SVT_LAUNCH_CHECK();
}

void svt_hip_generate_padding(uint8_t* base, uint32_t stride, uint32_t width, uint32_t height, uint32_t pad_x, uint32_t pad_y, void* stream) {
    svthip::ensure_device();
    if (width == 0 || height == 0) return;
    const int pw = (int)(width + 2 * pad_x), ph = (int)(height + 2 * pad_y);
    hipLaunchKernelGGL(generate_padding_kernel, dim3((pw + 255) / 256, ph), dim3(256), 0, (hipStream_t)stream, base, stride, (int)width, (int)height,
                       (int)pad_x, (int)pad_y);
    SVT_LAUNCH_CHECK();
}

void svt_aom_downsample_2d_hip(uint8_t* input_samples, uint32_t input_stride, uint32_t input_area_width, uint32_t input_area_height,
                               uint8_t* decim_samples, uint32_t decim_stride, uint32_t decim_step) {
    const int ow = decim_dim(input_area_width, decim_step), oh = decim_dim(input_area_height, decim_step);
    if (ow <= 0 || oh <= 0) return;
    svthip::HostCall& c = svthip::host_call();
    c.begin();
    // rows read: half - 1 .. (oh - 1) * step + half; columns half - 1 .. (ow - 1) * step + half
    const uint32_t half = decim_step >> 1;
    const size_t   iw = (size_t)(ow - 1) * decim_step + half + 1, ih = (size_t)(oh - 1) * decim_step + half + 1;
    const size_t   ip = svthip::align_up(iw, 16), op = svthip::align_up((size_t)ow, 16);
    c.reserve(ip * ih + op * oh + 1024, ip * ih + op * oh + 1024);
    uint8_t* di = (uint8_t*)c.dalloc(ip * ih);
    uint8_t* dout = (uint8_t*)c.dalloc(op * oh);
    c.up2d(di, ip, input_samples, input_stride, iw, ih);
    hipLaunchKernelGGL(downsample_kernel, dim3((ow + 255) / 256, oh), dim3(256), 0, c.stream, di, (uint32_t)ip, ow, oh, dout, (uint32_t)op, (int)decim_step);
    SVT_LAUNCH_CHECK();
    c.down2d(decim_samples, decim_stride, dout, op, ow, oh);
}

} // extern "C"

SVT_HIP_DEFINE_WARM(picprep) // (svt_hip_warmup loads this translation unit's code object at encoder initialisation: svt_hip_common.h)
