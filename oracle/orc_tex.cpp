// TEST INFRASTRUCTURE — CPU restatement ("oracle"); see orc_types.h.
//
// 2D material textures: mip-chain generation and filtered sampling. In the
// reference both are done by the Vulkan driver: vkCmdBlitImage with
// VK_FILTER_LINEAR level by level (Renderer/VulkanRenderer.cpp:332-340) and
// `texture()` through the material sampler (LINEAR min/mag/mip, anisotropy on,
// LOD unclamped; Renderer/VulkanRenderer.cpp:41 and :366-380). Restated from
// the Vulkan 1.3 texel-filtering rules with two stated simplifications, shared
// with the HIP path: the LOD is isotropic (anisotropic filtering is an
// implementation-defined refinement) and filter weights are full float.
#include <algorithm>
#include <cmath>

#include "orc_types.h"

namespace orc {

static inline uint32_t mip_dim(uint32_t d, uint32_t level) { return std::max(1u, d >> level); }

// One vkCmdBlitImage(LINEAR) from level l-1 to level l: the centre of each
// destination texel is scaled into the source, filtered bilinearly with
// clamp-to-edge. sRGB images are filtered in linear light and re-encoded;
// alpha is always linear.
static void blit_level(const Ctx& c, const std::vector<uint8_t>& src, int sw, int sh, std::vector<uint8_t>& dst, int dw, int dh, bool srgb) {
    dst.assign((size_t)dw * dh * 4, 0);
    const float scale_x = (float)sw / (float)dw, scale_y = (float)sh / (float)dh;
    for (int y = 0; y < dh; y++)
        for (int x = 0; x < dw; x++) {
            const float u = ((float)x + 0.5f) * scale_x - 0.5f, v = ((float)y + 0.5f) * scale_y - 0.5f;
            const float fu = std::floor(u), fv = std::floor(v);
            const float a = u - fu, b = v - fv;
            const int i0 = std::min(std::max((int)fu, 0), sw - 1), i1 = std::min(std::max((int)fu + 1, 0), sw - 1);
            const int j0 = std::min(std::max((int)fv, 0), sh - 1), j1 = std::min(std::max((int)fv + 1, 0), sh - 1);
            for (int k = 0; k < 4; k++) {
                const uint8_t q00 = src[((size_t)j0 * sw + i0) * 4 + k], q10 = src[((size_t)j0 * sw + i1) * 4 + k];
                const uint8_t q01 = src[((size_t)j1 * sw + i0) * 4 + k], q11 = src[((size_t)j1 * sw + i1) * 4 + k];
                const bool lin = srgb && k < 3;
                const float t00 = lin ? c.srgb.decode[q00] : (float)q00 / 255.0f, t10 = lin ? c.srgb.decode[q10] : (float)q10 / 255.0f;
                const float t01 = lin ? c.srgb.decode[q01] : (float)q01 / 255.0f, t11 = lin ? c.srgb.decode[q11] : (float)q11 / 255.0f;
                const float top = (1.0f - a) * t00 + a * t10, bot = (1.0f - a) * t01 + a * t11;
                const float r = (1.0f - b) * top + b * bot;
                dst[((size_t)y * dw + x) * 4 + k] = lin ? c.srgb.encode(r) : float_to_unorm8(r);
            }
        }
}

// The same blit for an SFLOAT image: texels filtered as they are, rounded to binary16 where the image stores that.
static void blit_level_float(const std::vector<uint8_t>& src_bytes, int sw, int sh, std::vector<uint8_t>& dst_bytes, int dw, int dh, bool to_half) {
    dst_bytes.assign((size_t)dw * dh * 16, 0);
    const float* src = reinterpret_cast<const float*>(src_bytes.data());
    float* dst = reinterpret_cast<float*>(dst_bytes.data());
    const float scale_x = (float)sw / (float)dw, scale_y = (float)sh / (float)dh;
    for (int y = 0; y < dh; y++)
        for (int x = 0; x < dw; x++) {
            const float u = ((float)x + 0.5f) * scale_x - 0.5f, v = ((float)y + 0.5f) * scale_y - 0.5f;
            const float fu = std::floor(u), fv = std::floor(v);
            const float a = u - fu, b = v - fv;
            const int i0 = std::min(std::max((int)fu, 0), sw - 1), i1 = std::min(std::max((int)fu + 1, 0), sw - 1);
            const int j0 = std::min(std::max((int)fv, 0), sh - 1), j1 = std::min(std::max((int)fv + 1, 0), sh - 1);
            for (int k = 0; k < 4; k++) {
                const float t00 = src[((size_t)j0 * sw + i0) * 4 + k], t10 = src[((size_t)j0 * sw + i1) * 4 + k];
                const float t01 = src[((size_t)j1 * sw + i0) * 4 + k], t11 = src[((size_t)j1 * sw + i1) * 4 + k];
                const float top = (1.0f - a) * t00 + a * t10, bot = (1.0f - a) * t01 + a * t11;
                const float r = (1.0f - b) * top + b * bot;
                dst[((size_t)y * dw + x) * 4 + k] = to_half ? quantize_half(r) : r;
            }
        }
}

// Renderer/VulkanRenderer.cpp:311-315: levels = floor(log2(max(w,h))) + 1, capped by maxMipLevels
void build_mips(Ctx& c, Image& img, uint32_t max_levels) {
    uint32_t levels = 1, m = std::max(img.w, img.h);
    if (m > 1) levels = (uint32_t)std::floor(std::log2((double)m)) + 1;
    levels = std::max(1u, std::min(levels, max_levels));
    img.mips.resize(levels);
    for (uint32_t l = 1; l < levels; l++) {
        const int sw = (int)mip_dim(img.w, l - 1), sh = (int)mip_dim(img.h, l - 1), dw = (int)mip_dim(img.w, l), dh = (int)mip_dim(img.h, l);
        if (img.is_float) blit_level_float(img.mips[l - 1], sw, sh, img.mips[l], dw, dh, img.half_texels);
        else blit_level(c, img.mips[l - 1], sw, sh, img.mips[l], dw, dh, img.srgb);
    }
}

static inline int wrap_index(int i, int n, uint32_t mode) {
    if (mode == RDR_WRAP_REPEAT) { int r = i % n; return r < 0 ? r + n : r; }
    if (mode == RDR_WRAP_MIRROR) {
        int period = 2 * n;
        int r = i % period; if (r < 0) r += period;
        return r < n ? r : period - 1 - r;
    }
    return std::min(std::max(i, 0), n - 1);
}

static void bilinear(const Ctx& c, const Image& img, const rdr_sampler& s, uint32_t level, float u, float v, float out[4]) {
    const int w = (int)mip_dim(img.w, level), h = (int)mip_dim(img.h, level);
    const std::vector<uint8_t>& px = img.mips[level];
    const float tu = u * (float)w - 0.5f, tv = v * (float)h - 0.5f;
    const float fu = std::floor(tu), fv = std::floor(tv);
    const float a = tu - fu, b = tv - fv;
    const int i0 = wrap_index((int)fu, w, s.wrap_u), i1 = wrap_index((int)fu + 1, w, s.wrap_u);
    const int j0 = wrap_index((int)fv, h, s.wrap_v), j1 = wrap_index((int)fv + 1, h, s.wrap_v);
    if (img.is_float) {
        const float* fx = reinterpret_cast<const float*>(px.data());
        for (int k = 0; k < 4; k++) {
            const float t00 = fx[((size_t)j0 * w + i0) * 4 + k], t10 = fx[((size_t)j0 * w + i1) * 4 + k];
            const float t01 = fx[((size_t)j1 * w + i0) * 4 + k], t11 = fx[((size_t)j1 * w + i1) * 4 + k];
            const float top = (1.0f - a) * t00 + a * t10, bot = (1.0f - a) * t01 + a * t11;
            out[k] = (1.0f - b) * top + b * bot;
        }
        return;
    }
    for (int k = 0; k < 4; k++) {
        const uint8_t q00 = px[((size_t)j0 * w + i0) * 4 + k], q10 = px[((size_t)j0 * w + i1) * 4 + k];
        const uint8_t q01 = px[((size_t)j1 * w + i0) * 4 + k], q11 = px[((size_t)j1 * w + i1) * 4 + k];
        const bool lin = img.srgb && k < 3;
        const float t00 = lin ? c.srgb.decode[q00] : (float)q00 / 255.0f, t10 = lin ? c.srgb.decode[q10] : (float)q10 / 255.0f;
        const float t01 = lin ? c.srgb.decode[q01] : (float)q01 / 255.0f, t11 = lin ? c.srgb.decode[q11] : (float)q11 / 255.0f;
        const float top = (1.0f - a) * t00 + a * t10, bot = (1.0f - a) * t01 + a * t11;
        out[k] = (1.0f - b) * top + b * bot;
    }
}

// One texel of a level: Vulkan 1.3, 16.8 nearest filtering (i = floor(u) on the unnormalised coordinate) followed by
// the wrapping operation of the address mode.
static void nearest(const Ctx& c, const Image& img, const rdr_sampler& s, uint32_t level, float u, float v, float out[4]) {
    const int w = (int)mip_dim(img.w, level), h = (int)mip_dim(img.h, level);
    const std::vector<uint8_t>& px = img.mips[level];
    const int i = wrap_index((int)std::floor(u * (float)w), w, s.wrap_u), j = wrap_index((int)std::floor(v * (float)h), h, s.wrap_v);
    if (img.is_float) {
        const float* fx = reinterpret_cast<const float*>(px.data());
        for (int k = 0; k < 4; k++) out[k] = fx[((size_t)j * w + i) * 4 + k];
        return;
    }
    for (int k = 0; k < 4; k++) {
        const uint8_t q = px[((size_t)j * w + i) * 4 + k];
        out[k] = (img.srgb && k < 3) ? c.srgb.decode[q] : (float)q / 255.0f;
    }
}

// texture(sampler2D, uv) with screen-space derivatives of uv. The Sampler's three filters as
// Renderer/VulkanRenderer.cpp:366-368 hands them to the driver: magFilter where lambda <= 0, minFilter above; mipmap
// mode NEAREST takes the level ceil(lambda + 0.5) - 1, LINEAR blends floor(lambda) and the next one.
void sample_texture(const Ctx& c, uint32_t tex, float u, float v, float dudx, float dvdx, float dudy, float dvdy, float out[4]) {
    const Texture& t = c.textures[tex];
    const Image& img = c.images[t.image];
    const rdr_sampler& s = t.sampler;
    const uint32_t levels = (uint32_t)img.mips.size();
    float lambda = 0.0f;
    bool minified = false;
    {
        const float ax = dudx * (float)img.w, ay = dvdx * (float)img.h, bx = dudy * (float)img.w, by = dvdy * (float)img.h;
        const float rho = std::max(std::sqrt(ax * ax + ay * ay), std::sqrt(bx * bx + by * by));
        if (rho > 1.0f) { minified = true; lambda = std::min(std::log2(rho), (float)(levels - 1)); }
    }
    const bool near = (minified ? s.min_filter : s.mag_filter) == RDR_FILTER_NEAREST;
    auto level_sample = [&](uint32_t level, float o[4]) { if (near) nearest(c, img, s, level, u, v, o); else bilinear(c, img, s, level, u, v, o); };
    if (s.mipmap_filter == RDR_FILTER_NEAREST) {
        level_sample((uint32_t)std::max(std::ceil(lambda + 0.5f) - 1.0f, 0.0f), out);
        return;
    }
    if (!(lambda > 0.0f)) { level_sample(0, out); return; }
    const float fl = std::floor(lambda);
    const uint32_t l0 = (uint32_t)fl, l1 = std::min(l0 + 1, levels - 1);
    const float f = lambda - fl;
    float c0[4], c1[4];
    level_sample(l0, c0);
    level_sample(l1, c1);
    for (int k = 0; k < 4; k++) out[k] = (1.0f - f) * c0[k] + f * c1[k];
}

}  // namespace orc
