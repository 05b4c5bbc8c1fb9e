// Post-process tail of the frame: colour -> bloom mip 0, five 13-tap
// down-samples, five 3x3 tent up-samples, filmic tonemap + log-luminance,
// centre-crop average luminance.
//
// Reference: Renderer/VulkanRendererBloom.cpp:62-100 (schedule),
// Renderer/shaders/bloomDownsample.comp, bloomUpsample.comp, tonemap.comp,
// averageLuminance.comp; Renderer/VulkanRenderer.cpp:1156-1191.
//
// All of these are HBM-bound: every pixel is touched once per pass. What is
// done about it:
//  * bloom mip 0 (the colour image converted to B10G11R11) is never read back
//    from memory: the first down-sample and the last up-sample take the colour
//    image itself and apply the storage truncation in registers;
//  * the last up-sample is fused into the tonemap kernel (tonemap needs only
//    mip 0's final value), which still stores mip 0 for readers of the chain;
//  * the up-sample reads its 3x3 source taps at (x/2, y/2): the four pixels of
//    a 2x2 quad share them, so one thread handles a quad and decodes the nine
//    packed taps once instead of four times;
//  * threads map to pixels so that a wave touches whole rows (coalesced
//    segments); neighbour re-reads are served by L1/L2.
// Arithmetic order (fma chains in tap order, packed-float truncation after
// every pass) is exactly the pass-by-pass order of the reference chain, so
// every stored mip matches the unfused chain bit for bit.
#include "device_common.h"
#include "kernels.h"

namespace {

RDR_DEV int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }

// value of a colour-image texel as bloom mip 0 stores it (B10G11R11, truncated)
RDR_DEV void load_mip0_from_hdr(const uint2* __restrict__ hdr, size_t i, float& r, float& g, float& b) {
    float a;
    unpack_half4(hdr[i], r, g, b, a);
    unpack_r11g11b10(pack_r11g11b10(r, g, b), r, g, b);
}

// bloomDownsample.comp:14-33. The LINEAR unnormalized sampler is hit exactly
// on texel centres, so each tap is one texel; taps in shader order.
// FROM_HDR: the source level is mip 0, read from the colour image.
template <bool FROM_HDR>
__global__ void __launch_bounds__(256) k_bloom_down(const void* __restrict__ src_v, int sw, int sh, uint32_t* __restrict__ dst,
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
        if (FROM_HDR) load_mip0_from_hdr((const uint2*)src_v, (size_t)sy * sw + sx, r, g, b);
        else unpack_r11g11b10(((const uint32_t*)src_v)[(size_t)sy * sw + sx], r, g, b);
        acc0 = fmaf(ws[t], r, acc0); acc1 = fmaf(ws[t], g, acc1); acc2 = fmaf(ws[t], b, acc2);
    }
    dst[(size_t)y * dst_pitch + x] = pack_r11g11b10(acc0, acc1, acc2);
}

// the nine taps of bloomUpsample.comp:17-25 around source texel (ux, uy), decoded once
struct Tent { float r[9], g[9], b[9]; };
RDR_DEV void load_tent(const uint32_t* __restrict__ src, int sw, int sh, int ux, int uy, Tent& t) {
    const int dxs[9] = {-1, 0, +1, -1, 0, +1, -1, 0, +1};
    const int dys[9] = {-1, -1, -1, 0, 0, 0, +1, +1, +1};
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const int sx = clampi(ux + dxs[k], 0, sw - 1), sy = clampi(uy + dys[k], 0, sh - 1);
        unpack_r11g11b10(src[(size_t)sy * sw + sx], t.r[k], t.g[k], t.b[k]);
    }
}
// dst += tent3x3(src): the read-modify-write of one destination texel, fma chain in tap order
RDR_DEV uint32_t tent_accumulate(const Tent& t, float acc0, float acc1, float acc2) {
    const float ws[9] = {0.0625f, 0.125f, 0.0625f, 0.125f, 0.25f, 0.125f, 0.0625f, 0.125f, 0.0625f};
#pragma unroll
    for (int k = 0; k < 9; k++) { acc0 = fmaf(ws[k], t.r[k], acc0); acc1 = fmaf(ws[k], t.g[k], acc1); acc2 = fmaf(ws[k], t.b[k], acc2); }
    return pack_r11g11b10(acc0, acc1, acc2);
}

// bloomUpsample.comp:15-29, one thread per 2x2 quad of the destination level
__global__ void __launch_bounds__(256) k_bloom_up(const uint32_t* __restrict__ src, int sw, int sh, uint32_t* __restrict__ dst,
                                                  int dw, int dh, int dst_pitch) {
    const int qx = blockIdx.x * 64 + (threadIdx.x & 63), qy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (2 * qx >= dw || 2 * qy >= dh) return;
    Tent t;
    load_tent(src, sw, sh, qx, qy, t);
#pragma unroll
    for (int oy = 0; oy < 2; oy++)
#pragma unroll
        for (int ox = 0; ox < 2; ox++) {
            const int x = 2 * qx + ox, y = 2 * qy + oy;
            if (x >= dw || y >= dh) continue;
            float a0, a1, a2;
            unpack_r11g11b10(dst[(size_t)y * dst_pitch + x], a0, a1, a2);
            dst[(size_t)y * dst_pitch + x] = tent_accumulate(t, a0, a1, a2);
        }
}

// tonemap.comp:26-28; the two divisions feed a continuous curve, the hardware reciprocal is enough
RDR_DEV float tone(float x) {
    const float shoulderStr = 0.22f, linStr = 0.30f, linAngle = 0.10f, toeStr = 0.20f, toeNum = 0.01f, toeDenom = 0.30f;
    return ((x * (shoulderStr * x + linAngle * linStr) + toeStr * toeNum) * __builtin_amdgcn_rcpf(x * (shoulderStr * x + linStr) + toeStr * toeDenom)) - toeNum / toeDenom;
}

// tonemap.comp:45-50 for one channel. exp and pow go straight to the hardware exp2 / log2: the argument of the
// first is <= 0 and the second sees [0, 1), so the library's range fix-ups buy nothing at 8-bit output precision.
// `scale2` = 2^exposure * log2(e).
RDR_DEV uint32_t tonemap_channel(float c, float bloom, float scale2, float inv_gamma, float inv_white) {
    float v = c * (1.0f - 0.04f) + bloom * 0.04f;
    v = tone(v) * inv_white;
    v = 1.0f - __builtin_amdgcn_exp2f(-v * scale2);
    v = __builtin_amdgcn_exp2f(inv_gamma * __builtin_amdgcn_logf(v));
    return (uint32_t)(int)(fminf(fmaxf(v, 0.0f), 1.0f) * 255.0f + 0.5f);  // UNORM8, NaN -> 0
}

// Last up-sample (mip 1 -> mip 0, destination read from the colour image) + tonemap.comp:34-53 for a 2x2 quad.
// `inv_white` = 1 / tone(11.2) from the host. VEC: the image width is even, so the two pixels of a quad row are
// one 16-byte load and 8-byte stores.
template <bool VEC>
__global__ void __launch_bounds__(256) k_tonemap(const uint2* __restrict__ hdr, const uint32_t* __restrict__ mip1, int sw, int sh,
                                                 uint32_t* __restrict__ bloom0, uint32_t* __restrict__ out, float* __restrict__ loglum,
                                                 int W, int H, float scale2, float inv_gamma, float inv_white, float log2_6) {
    const int qx = blockIdx.x * 64 + (threadIdx.x & 63), qy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (2 * qx >= W || 2 * qy >= H) return;
    Tent t;
    const bool have_mip1 = sw > 0 && sh > 0;
    if (have_mip1) load_tent(mip1, sw, sh, qx, qy, t);
#pragma unroll
    for (int oy = 0; oy < 2; oy++) {
        const int y = 2 * qy + oy;
        if (y >= H) continue;
        const size_t i0 = (size_t)y * W + 2 * qx;
        uint2 h[2];
        if (VEC) { const uint4 v = *(const uint4*)(hdr + i0); h[0] = make_uint2(v.x, v.y); h[1] = make_uint2(v.z, v.w); }
        else { h[0] = hdr[i0]; h[1] = (2 * qx + 1 < W) ? hdr[i0 + 1] : make_uint2(0u, 0u); }
        uint32_t b0[2], px[2];
        float ll[2];
#pragma unroll
        for (int ox = 0; ox < 2; ox++) {
            float c[3], a, bl[3];
            unpack_half4(h[ox], c[0], c[1], c[2], a);
            b0[ox] = pack_r11g11b10(c[0], c[1], c[2]);
            if (have_mip1) {
                unpack_r11g11b10(b0[ox], bl[0], bl[1], bl[2]);
                b0[ox] = tent_accumulate(t, bl[0], bl[1], bl[2]);
            }
            unpack_r11g11b10(b0[ox], bl[0], bl[1], bl[2]);
            ll[ox] = __builtin_amdgcn_logf((2.0f * c[0] + 3.0f * c[1]) + c[2]) - log2_6;
            px[ox] = 0xFF000000u | tonemap_channel(c[0], bl[0], scale2, inv_gamma, inv_white) |
                     (tonemap_channel(c[1], bl[1], scale2, inv_gamma, inv_white) << 8) |
                     (tonemap_channel(c[2], bl[2], scale2, inv_gamma, inv_white) << 16);
        }
        if (VEC) {
            *(uint2*)(bloom0 + i0) = make_uint2(b0[0], b0[1]);
            *(uint2*)(out + i0) = make_uint2(px[0], px[1]);
            *(float2*)(loglum + i0) = make_float2(ll[0], ll[1]);
        } else {
            bloom0[i0] = b0[0]; out[i0] = px[0]; loglum[i0] = ll[0];
            if (2 * qx + 1 < W) { bloom0[i0 + 1] = b0[1]; out[i0 + 1] = px[1]; loglum[i0 + 1] = ll[1]; }
        }
    }
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

void launch_bloom_down(hipStream_t s, const void* src, bool src_is_hdr, int sw, int sh, uint32_t* dst, int dw, int dh, int dst_pitch) {
    if (dw <= 0 || dh <= 0) return;
    const dim3 grid((dw + 63) / 64, (dh + 3) / 4);
    if (src_is_hdr) hipLaunchKernelGGL(k_bloom_down<true>, grid, dim3(256), 0, s, src, sw, sh, dst, dw, dh, dst_pitch);
    else hipLaunchKernelGGL(k_bloom_down<false>, grid, dim3(256), 0, s, src, sw, sh, dst, dw, dh, dst_pitch);
}
void launch_bloom_up(hipStream_t s, const uint32_t* src, int sw, int sh, uint32_t* dst, int dw, int dh, int dst_pitch) {
    if (dw <= 0 || dh <= 0) return;
    const int qw = (dw + 1) / 2, qh = (dh + 1) / 2;
    hipLaunchKernelGGL(k_bloom_up, dim3((qw + 63) / 64, (qh + 3) / 4), dim3(256), 0, s, src, sw, sh, dst, dw, dh, dst_pitch);
}
void launch_tonemap(hipStream_t s, const void* hdr, const uint32_t* mip1, int sw, int sh, uint32_t* bloom0, uint32_t* out, float* loglum,
                    int W, int H, float scale, float inv_gamma, float white, float log2_6) {
    const int qw = (W + 1) / 2, qh = (H + 1) / 2;
    const float scale2 = scale * 1.44269504088896340736f;  // exp(-x s) = exp2(-x s log2 e)
    const dim3 grid((qw + 63) / 64, (qh + 3) / 4);
    if ((W & 1) == 0)
        hipLaunchKernelGGL(k_tonemap<true>, grid, dim3(256), 0, s, (const uint2*)hdr, mip1, sw, sh, bloom0, out, loglum, W, H, scale2, inv_gamma, 1.0f / white, log2_6);
    else
        hipLaunchKernelGGL(k_tonemap<false>, grid, dim3(256), 0, s, (const uint2*)hdr, mip1, sw, sh, bloom0, out, loglum, W, H, scale2, inv_gamma, 1.0f / white, log2_6);
}
void launch_average_luminance(hipStream_t s, const float* loglum, int W, int H, float* l16, float* l1) {
    hipLaunchKernelGGL(k_lum_reduce, dim3(256), dim3(64), 0, s, loglum, W, H, W / 2 - 128, H / 2 - 128, 0, l16, 16);
    hipLaunchKernelGGL(k_lum_reduce, dim3(1), dim3(64), 0, s, (const float*)l16, 16, 16, 0, 0, 16, l1, 1);
}
