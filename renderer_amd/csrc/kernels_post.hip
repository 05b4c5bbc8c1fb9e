// Post-process tail of the frame: colour -> bloom mip 0, five 13-tap
// down-samples, five 3x3 tent up-samples, filmic tonemap + log-luminance,
// centre-crop average luminance.
//
// Reference: Renderer/VulkanRendererBloom.cpp:62-100 (schedule),
// Renderer/shaders/bloomDownsample.comp, bloomUpsample.comp, tonemap.comp,
// averageLuminance.comp; Renderer/VulkanRenderer.cpp:1156-1191.
//
// All of these are HBM-bound: every pixel is touched once per pass, the
// arithmetic per byte is tiny. Threads map to pixels so that a wave reads and
// writes whole rows of 64 consecutive texels (256-B segments); the stencil
// re-reads of neighbours are served by L1/L2.
#include "device_common.h"
#include "kernels.h"

namespace {

__global__ void __launch_bounds__(256) k_hdr_to_bloom0(const uint2* __restrict__ hdr, uint32_t* __restrict__ bloom0, uint32_t n) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    float r, g, b, a;
    unpack_half4(hdr[i], r, g, b, a);
    bloom0[i] = pack_r11g11b10(r, g, b);
}

RDR_DEV int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }

// bloomDownsample.comp:14-33. The LINEAR unnormalized sampler is hit exactly
// on texel centres, so each tap is one texel; taps in shader order.
__global__ void __launch_bounds__(256) k_bloom_down(const uint32_t* __restrict__ src, int sw, int sh, uint32_t* __restrict__ dst,
                                                    int dw, int dh, int dst_pitch) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= dw || y >= dh) return;
    const int dxs[13] = {+1, +1, -1, -1, -2, 0, +2, +2, +2, 0, -2, -2, 0};
    const int dys[13] = {+1, -1, +1, -1, -2, -2, -2, 0, +2, +2, +2, 0, 0};
    const float ws[13] = {0.125f, 0.125f, 0.125f, 0.125f, 0.03125f, 0.0625f, 0.03125f, 0.0625f, 0.03125f, 0.0625f, 0.03125f, 0.0625f, 0.125f};
    float acc0 = 0.0f, acc1 = 0.0f, acc2 = 0.0f;
#pragma unroll
    for (int t = 0; t < 13; t++) {
        const int sx = clampi(2 * x + dxs[t], 0, sw - 1), sy = clampi(2 * y + dys[t], 0, sh - 1);
        float r, g, b;
        unpack_r11g11b10(src[(size_t)sy * sw + sx], r, g, b);
        acc0 = fmaf(ws[t], r, acc0); acc1 = fmaf(ws[t], g, acc1); acc2 = fmaf(ws[t], b, acc2);
    }
    dst[(size_t)y * dst_pitch + x] = pack_r11g11b10(acc0, acc1, acc2);
}

// bloomUpsample.comp:15-29: dst += tent3x3(src) at texel (x/2, y/2), NEAREST taps
__global__ void __launch_bounds__(256) k_bloom_up(const uint32_t* __restrict__ src, int sw, int sh, uint32_t* __restrict__ dst,
                                                  int dw, int dh, int dst_pitch) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= dw || y >= dh) return;
    const int dxs[9] = {-1, 0, +1, -1, 0, +1, -1, 0, +1};
    const int dys[9] = {-1, -1, -1, 0, 0, 0, +1, +1, +1};
    const float ws[9] = {0.0625f, 0.125f, 0.0625f, 0.125f, 0.25f, 0.125f, 0.0625f, 0.125f, 0.0625f};
    float acc0, acc1, acc2;
    unpack_r11g11b10(dst[(size_t)y * dst_pitch + x], acc0, acc1, acc2);
    const int ux = x / 2, uy = y / 2;
#pragma unroll
    for (int t = 0; t < 9; t++) {
        const int sx = clampi(ux + dxs[t], 0, sw - 1), sy = clampi(uy + dys[t], 0, sh - 1);
        float r, g, b;
        unpack_r11g11b10(src[(size_t)sy * sw + sx], r, g, b);
        acc0 = fmaf(ws[t], r, acc0); acc1 = fmaf(ws[t], g, acc1); acc2 = fmaf(ws[t], b, acc2);
    }
    dst[(size_t)y * dst_pitch + x] = pack_r11g11b10(acc0, acc1, acc2);
}

// tonemap.comp:26-28
RDR_DEV float tone(float x) {
    const float shoulderStr = 0.22f, linStr = 0.30f, linAngle = 0.10f, toeStr = 0.20f, toeNum = 0.01f, toeDenom = 0.30f;
    return ((x * (shoulderStr * x + linAngle * linStr) + toeStr * toeNum) / (x * (shoulderStr * x + linStr) + toeStr * toeDenom)) - toeNum / toeDenom;
}

// tonemap.comp:34-53. `scale` = 2^exposure, `white` = tone(11.2), both from the host.
__global__ void __launch_bounds__(256) k_tonemap(const uint2* __restrict__ hdr, const uint32_t* __restrict__ bloom0, uint32_t* __restrict__ out,
                                                 float* __restrict__ loglum, uint32_t n, float scale, float inv_gamma, float white, float log2_6) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    float c[3], a, bl[3];
    unpack_half4(hdr[i], c[0], c[1], c[2], a);
    unpack_r11g11b10(bloom0[i], bl[0], bl[1], bl[2]);
    loglum[i] = __log2f((2.0f * c[0] + 3.0f * c[1]) + c[2]) - log2_6;
    uint32_t px = 0xFF000000u;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float v = c[k] * (1.0f - 0.04f) + bl[k] * 0.04f;
        v = tone(v) / white;
        v = 1.0f - __expf(-v * scale);
        v = __powf(v, inv_gamma);
        px |= float_to_unorm8(v) << (8 * k);
    }
    out[i] = px;
}

// averageLuminance.comp on a 2N x 2N input -> N/8 x N/8 groups, one block per
// group, 64 threads, each fetching a 2x2 quad; the tree adds in the shader's
// order. `crop` != 0: the input is the W x H log-luminance image and texel
// (x,y) is read at the centre-crop offset with clamp-to-edge
// (VulkanRenderer.cpp:1165).
__global__ void __launch_bounds__(64) k_lum_reduce(const float* __restrict__ in, int in_w, int in_h, int ox, int oy, int dim,
                                                   float* __restrict__ out, int groups) {
    __shared__ float a[8][8];  // localAverages[x][y]
    const int lx = threadIdx.x & 7, ly = threadIdx.x >> 3;
    const int gx = blockIdx.x % groups, gy = blockIdx.x / groups;
    const int sx = (gx * 8 + lx) * 2, sy = (gy * 8 + ly) * 2;
    const float f = 1.0f / (8 * 8 * 4);
    float l[4];
    const int offx[4] = {0, 0, 1, 1}, offy[4] = {0, 1, 0, 1};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        int x = sx + offx[k], y = sy + offy[k];
        if (dim == 0) { x = clampi(ox + x, 0, in_w - 1); y = clampi(oy + y, 0, in_h - 1); }
        l[k] = in[(size_t)y * in_w + x];
    }
    a[lx][ly] = ((l[0] * f + l[1] * f) + l[2] * f) + l[3] * f;
    __syncthreads();
    for (int i = 4; i > 0; i >>= 1) {
        float v = 0.0f;
        const bool act = lx < i && ly < i;
        if (act) v = a[lx][ly] + ((a[lx + i][ly + i] + a[lx][ly + i]) + a[lx + i][ly]);
        __syncthreads();
        if (act) a[lx][ly] = v;
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = a[0][0];
}

}  // namespace

void launch_hdr_to_bloom0(hipStream_t s, const void* hdr, uint32_t* bloom0, uint32_t n) {
    hipLaunchKernelGGL(k_hdr_to_bloom0, dim3((n + 255) / 256), dim3(256), 0, s, (const uint2*)hdr, bloom0, n);
}
void launch_bloom_down(hipStream_t s, const uint32_t* src, int sw, int sh, uint32_t* dst, int dw, int dh, int dst_pitch) {
    if (dw <= 0 || dh <= 0) return;
    hipLaunchKernelGGL(k_bloom_down, dim3((dw + 63) / 64, (dh + 3) / 4), dim3(256), 0, s, src, sw, sh, dst, dw, dh, dst_pitch);
}
void launch_bloom_up(hipStream_t s, const uint32_t* src, int sw, int sh, uint32_t* dst, int dw, int dh, int dst_pitch) {
    if (dw <= 0 || dh <= 0) return;
    hipLaunchKernelGGL(k_bloom_up, dim3((dw + 63) / 64, (dh + 3) / 4), dim3(256), 0, s, src, sw, sh, dst, dw, dh, dst_pitch);
}
void launch_tonemap(hipStream_t s, const void* hdr, const uint32_t* bloom0, uint32_t* out, float* loglum, uint32_t n, float scale,
                    float inv_gamma, float white, float log2_6) {
    hipLaunchKernelGGL(k_tonemap, dim3((n + 255) / 256), dim3(256), 0, s, (const uint2*)hdr, bloom0, out, loglum, n, scale, inv_gamma, white, log2_6);
}
void launch_average_luminance(hipStream_t s, const float* loglum, int W, int H, float* l16, float* l1) {
    hipLaunchKernelGGL(k_lum_reduce, dim3(256), dim3(64), 0, s, loglum, W, H, W / 2 - 128, H / 2 - 128, 0, l16, 16);
    hipLaunchKernelGGL(k_lum_reduce, dim3(1), dim3(64), 0, s, (const float*)l16, 16, 16, 0, 0, 16, l1, 1);
}
