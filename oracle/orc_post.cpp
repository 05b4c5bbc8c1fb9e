// TEST INFRASTRUCTURE — CPU restatement ("oracle"); see orc_types.h.
// Bloom chain, tonemap and average luminance.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "orc_types.h"

namespace orc {

static inline uint32_t mip_dim(uint32_t d, uint32_t level) { return std::max(1u, d >> level); }
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Renderer/VulkanRendererBloom.cpp:62-100 (pre-recorded command buffer):
//   blit colour -> bloom mip 0 (NEAREST, 1:1, converts RGBA16F -> B10G11R11)
//   5 x bloomDownsample.comp, then 5 x bloomUpsample.comp.
static void bloom_chain(Ctx& c) {
    const uint32_t W = c.W, H = c.H;
    c.bloom.resize(6);
    for (uint32_t l = 0; l < 6; l++) c.bloom[l].assign((size_t)mip_dim(W, l) * mip_dim(H, l), 0u);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)W * H; i++) {
        const float* p = &c.hdr[(size_t)i * 4];
        c.bloom[0][i] = pack_r11g11b10(p[0], p[1], p[2]);
    }
    // Renderer/shaders/bloomDownsample.comp:14-33. The sampler is LINEAR with
    // unnormalized coordinates (VulkanRendererBloom.cpp:33); every tap sits at
    // (2x + d - 0.5), i.e. exactly on the centre of texel 2x + d - 1, so the
    // linear filter degenerates to that texel. Taps listed in shader order.
    static const int dn_dx[13] = {+1, +1, -1, -1, -2, 0, +2, +2, +2, 0, -2, -2, 0};
    static const int dn_dy[13] = {+1, -1, +1, -1, -2, -2, -2, 0, +2, +2, +2, 0, 0};
    static const float dn_w[13] = {0.125f, 0.125f, 0.125f, 0.125f, 0.03125f, 0.0625f, 0.03125f,
                                   0.0625f, 0.03125f, 0.0625f, 0.03125f, 0.0625f, 0.125f};
    for (uint32_t i = 0; i < 5; i++) {
        const int sw = (int)mip_dim(W, i), sh = (int)mip_dim(H, i);
        const int dw = (int)(W >> (i + 1)), dh = (int)(H >> (i + 1));  // dispatch extent (…Bloom.cpp:82)
        const std::vector<uint32_t>& src = c.bloom[i];
        std::vector<uint32_t>& dst = c.bloom[i + 1];
        const int dstw = (int)mip_dim(W, i + 1);
#pragma omp parallel for schedule(static)
        for (int y = 0; y < dh; y++)
            for (int x = 0; x < dw; x++) {
                float acc[3] = {0, 0, 0};
                for (int t = 0; t < 13; t++) {
                    int sx = clampi(2 * x + dn_dx[t], 0, sw - 1), sy = clampi(2 * y + dn_dy[t], 0, sh - 1);
                    float r, g, b;
                    unpack_r11g11b10(src[(size_t)sy * sw + sx], r, g, b);
                    acc[0] = std::fmaf(dn_w[t], r, acc[0]);
                    acc[1] = std::fmaf(dn_w[t], g, acc[1]);
                    acc[2] = std::fmaf(dn_w[t], b, acc[2]);
                }
                dst[(size_t)y * dstw + x] = pack_r11g11b10(acc[0], acc[1], acc[2]);
            }
    }
    // Renderer/shaders/bloomUpsample.comp:15-29: NEAREST unnormalized sampler
    // (VulkanRendererBloom.cpp:34), read-modify-write of the destination mip.
    static const int up_dx[9] = {-1, 0, +1, -1, 0, +1, -1, 0, +1};
    static const int up_dy[9] = {-1, -1, -1, 0, 0, 0, +1, +1, +1};
    static const float up_w[9] = {0.0625f, 0.125f, 0.0625f, 0.125f, 0.25f, 0.125f, 0.0625f, 0.125f, 0.0625f};
    for (uint32_t i = 0; i < 5; i++) {
        const uint32_t dl = 4 - i, sl = 5 - i;
        const int sw = (int)mip_dim(W, sl), sh = (int)mip_dim(H, sl);
        const int dw = (int)(W >> dl), dh = (int)(H >> dl);
        const int dstw = (int)mip_dim(W, dl);
        const std::vector<uint32_t>& src = c.bloom[sl];
        std::vector<uint32_t>& dst = c.bloom[dl];
#pragma omp parallel for schedule(static)
        for (int y = 0; y < dh; y++)
            for (int x = 0; x < dw; x++) {
                float acc[3];
                unpack_r11g11b10(dst[(size_t)y * dstw + x], acc[0], acc[1], acc[2]);
                const int ux = x / 2, uy = y / 2;
                for (int t = 0; t < 9; t++) {
                    int sx = clampi(ux + up_dx[t], 0, sw - 1), sy = clampi(uy + up_dy[t], 0, sh - 1);
                    float r, g, b;
                    unpack_r11g11b10(src[(size_t)sy * sw + sx], r, g, b);
                    acc[0] = std::fmaf(up_w[t], r, acc[0]);
                    acc[1] = std::fmaf(up_w[t], g, acc[1]);
                    acc[2] = std::fmaf(up_w[t], b, acc[2]);
                }
                dst[(size_t)y * dstw + x] = pack_r11g11b10(acc[0], acc[1], acc[2]);
            }
    }
}

// Renderer/shaders/tonemap.comp:18-28 (Uncharted-2 filmic curve)
static inline float tone(float x) {
    const float shoulderStr = 0.22f, linStr = 0.30f, linAngle = 0.10f, toeStr = 0.20f, toeNum = 0.01f, toeDenom = 0.30f;
    return ((x * (shoulderStr * x + linAngle * linStr) + toeStr * toeNum) / (x * (shoulderStr * x + linStr) + toeStr * toeDenom)) -
           toeNum / toeDenom;
}

// Renderer/shaders/tonemap.comp:34-53, pushed {exposure, gamma}
// (Renderer/VulkanRenderer.cpp:1094). The output image is the swapchain image:
// R8G8B8A8_UNORM in the SDR path (VulkanRenderer.cpp:673-676); with hdrOutputEnabled
// a float format (VulkanRenderer.cpp:668-673), here R16G16B16A16_SFLOAT, the one of the
// three that presentation surfaces offer: the same values, stored as binary16 (4 x 2 bytes
// per pixel, alpha 1.0 as tonemap.comp:52 writes it).
static void tonemap(Ctx& c, float exposure) {
    const uint32_t W = c.W, H = c.H;
    const float gamma = c.settings.gamma;
    const bool hdr_out = c.settings.hdr_output_enabled != 0;
    c.out.assign((size_t)W * H * (hdr_out ? 8 : 4), 0);
    c.loglum.assign((size_t)W * H, 0.0f);
    const float white = tone(11.2f);
    const float scale = std::pow(2.0f, exposure);
    const float inv_gamma = 1.0f / gamma;
    const float log2_6 = std::log2(6.0f);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)W * H; i++) {
        const float* p = &c.hdr[(size_t)i * 4];
        float col[3] = {p[0], p[1], p[2]};
        float bl[3];
        unpack_r11g11b10(c.bloom[0][i], bl[0], bl[1], bl[2]);
        c.loglum[i] = std::log2((2.0f * col[0] + 3.0f * col[1]) + col[2]) - log2_6;
        for (int k = 0; k < 3; k++) {
            float v = col[k] * (1.0f - 0.04f) + bl[k] * 0.04f;  // mix(color, bloom, 0.04)
            v = tone(v) / white;
            v = 1.0f - std::exp(-v * scale);
            v = std::pow(v, inv_gamma);
            if (hdr_out) { const uint16_t h = float_to_half(v); std::memcpy(&c.out[(size_t)i * 8 + 2 * k], &h, 2); }
            else c.out[(size_t)i * 4 + k] = float_to_unorm8(v);
        }
        if (hdr_out) { const uint16_t one = 0x3C00; std::memcpy(&c.out[(size_t)i * 8 + 6], &one, 2); }
        else c.out[(size_t)i * 4 + 3] = 255;
    }
}

void bloom_and_tonemap(Ctx& c, float exposure) {
    bloom_chain(c);
    tonemap(c, exposure);
}

// Renderer/VulkanRenderer.cpp:1156-1191 + Renderer/shaders/averageLuminance.comp:
// NEAREST blit of the centre 256x256 crop of the log-luminance image, then
// 256^2 -> 16^2 -> 1 with 8x8 groups whose threads each fetch a 2x2 quad.
// The shader lacks a barrier between the first store and the first reduction
// step (averageLuminance.comp:21-25); this computes the intended mean, with
// the shader's summation order. Crops that leave the image (W or H < 256)
// clamp to the edge.
static void reduce_stage(const std::vector<float>& in, int in_dim, std::vector<float>& out) {
    const int groups = in_dim / 16;
    out.assign((size_t)groups * groups, 0.0f);
    const float f = 1.0f / (8 * 8 * 4);
    for (int gy = 0; gy < groups; gy++)
        for (int gx = 0; gx < groups; gx++) {
            float a[8][8];  // localAverages[x][y]
            for (int ly = 0; ly < 8; ly++)
                for (int lx = 0; lx < 8; lx++) {
                    int sx = (gx * 8 + lx) * 2, sy = (gy * 8 + ly) * 2;
                    float l0 = in[(size_t)sy * in_dim + sx], l1 = in[(size_t)(sy + 1) * in_dim + sx];
                    float l2 = in[(size_t)sy * in_dim + sx + 1], l3 = in[(size_t)(sy + 1) * in_dim + sx + 1];
                    a[lx][ly] = ((l0 * f + l1 * f) + l2 * f) + l3 * f;  // dot(lums, vec4(f))
                }
            for (int i = 4; i > 0; i >>= 1)
                for (int ly = 0; ly < i; ly++)
                    for (int lx = 0; lx < i; lx++) a[lx][ly] = a[lx][ly] + ((a[lx + i][ly + i] + a[lx][ly + i]) + a[lx + i][ly]);
            out[(size_t)gy * groups + gx] = a[0][0];
        }
}

float average_luminance(Ctx& c) {
    const int W = (int)c.W, H = (int)c.H;
    std::vector<float> l256((size_t)256 * 256), l16, l1;
    const int ox = W / 2 - 128, oy = H / 2 - 128;
    for (int y = 0; y < 256; y++)
        for (int x = 0; x < 256; x++)
            l256[(size_t)y * 256 + x] = c.loglum[(size_t)clampi(oy + y, 0, H - 1) * W + clampi(ox + x, 0, W - 1)];
    reduce_stage(l256, 256, l16);
    reduce_stage(l16, 16, l1);
    return l1[0];
}

}  // namespace orc
