// One-off resource kernels: environment cube upload (RGBA32F -> B10G11R11),
// irradiance and GGX-prefilter bakes, and material-texture mip generation.
//
// Reference: Renderer/VulkanRendererEnvironment.cpp:71-125 (upload/convert),
// :193-250 + shaders/envbakediffuse.frag (irradiance, 256^2 per face),
// :324-391 + shaders/envbakespecular.frag (prefilter, 512^2, 10 levels,
// roughness = level/9), shaders/envbake.vert (direction per pixel);
// Renderer/VulkanRenderer.cpp:332-340 (mip chain by linear blits).
//
// These run once per scene, so they are written for exact agreement with the
// CPU checker rather than speed: correctly rounded divide/sqrt, no
// contraction, and the trigonometric values of the fixed sample angles come
// from host tables.
#include <algorithm>
#include <cstring>

#include "device_common.h"
#include "kernels.h"

namespace {

__global__ void __launch_bounds__(256) k_env_pack(const float4* __restrict__ face, uint32_t* __restrict__ dst, uint32_t n) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float4 c = face[i];
    dst[i] = pack_r11g11b10(c.x, c.y, c.z);
}

// envbake.vert with the face rotations of VulkanRendererEnvironment.cpp:209-217 written out
RDR_DEV F3 bake_direction(int face, int x, int y, int res) {
    const float xn = ((float)x + 0.5f) * (2.0f / (float)res) - 1.0f;
    const float yn = ((float)y + 0.5f) * (2.0f / (float)res) - 1.0f;
    const float vx = xn, vy = -yn, vz = 1.0f;
    switch (face) {
        case 0: return f3(vz, vy, -vx);
        case 1: return f3(-vz, vy, vx);
        case 2: return f3(vx, vz, -vy);
        case 3: return f3(vx, -vz, vy);
        case 4: return f3(vx, vy, vz);
        default: return f3(-vx, vy, -vz);
    }
}

// envbakediffuse.frag:12-40
// `env` (both bakes): the bordered copy of the environment cube
__global__ void __launch_bounds__(256) k_bake_irradiance(DCube env, uint32_t* __restrict__ dst, int res, IrradianceTables tb) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= 6u * res * res) return;
    const int face = (int)(i / (uint32_t)(res * res)), y = (int)((i / (uint32_t)res) % (uint32_t)res), x = (int)(i % (uint32_t)res);
    const float PI = 3.14159265359f;
    const int nSamples = 24;
    const F3 normal = normalize3(bake_direction(face, x, y, res));
    F3 up = f3(0.0f, 1.0f, 0.0f);
    const F3 right = normalize3(cross3(up, normal));
    up = normalize3(cross3(normal, right));
    float irr[3] = {0.0f, 0.0f, 0.0f};
    for (int a = 0; a < nSamples; a++)
        for (int b = 0; b < nSamples / 4; b++) {
            const float st = tb.sin_theta[b], ct = tb.cos_theta[b];
            const float tx = st * tb.cos_phi[a], ty = st * tb.sin_phi[a], tz = ct;
            const F3 sv = right * tx + up * ty + normal * tz;
            float e[3];
            cube_sample_bordered(env, 0, sv.x, sv.y, sv.z, e);
#pragma unroll
            for (int k = 0; k < 3; k++) irr[k] += e[k] * ct * st;
        }
    const float norm = 1.0f / (float)(nSamples * nSamples / 4);
#pragma unroll
    for (int k = 0; k < 3; k++) irr[k] = PI * irr[k] * norm;
    dst[i] = pack_r11g11b10(irr[0], irr[1], irr[2]);
}

RDR_DEV float radical_inverse(uint32_t bits) {
    bits = (bits << 16u) | (bits >> 16u);
    bits = ((bits & 0x55555555u) << 1u) | ((bits & 0xAAAAAAAAu) >> 1u);
    bits = ((bits & 0x33333333u) << 2u) | ((bits & 0xCCCCCCCCu) >> 2u);
    bits = ((bits & 0x0F0F0F0Fu) << 4u) | ((bits & 0xF0F0F0F0u) >> 4u);
    bits = ((bits & 0x00FF00FFu) << 8u) | ((bits & 0xFF00FF00u) >> 8u);
    return (float)bits * 2.3283064365386963e-10f;
}

// envbakespecular.frag:32-86; cos/sin of phi_i = 2 pi i/128 come from `trig` (256 floats: cos then sin)
__global__ void __launch_bounds__(256) k_bake_prefilter(DCube env, uint32_t* __restrict__ dst, int res, float roughness,
                                                        const float* __restrict__ trig) {
    // The half vectors in tangent space depend on the sample and the level's roughness only: the block works them out
    // once (the same operations in the same order as per texel before, so the same bits) instead of once per texel.
    constexpr int nSamples = 128;
    __shared__ float lds_h[3][nSamples];
    if (threadIdx.x < (uint32_t)nSamples) {
        const int s = (int)threadIdx.x;
        const float a = roughness * roughness;
        const float xi_y = radical_inverse((uint32_t)s);
        const float cosTheta = sqrtf((1.0f - xi_y) / (1.0f + (a * a - 1.0f) * xi_y));
        const float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
        lds_h[0][s] = trig[s] * sinTheta; lds_h[1][s] = trig[128 + s] * sinTheta; lds_h[2][s] = cosTheta;
    }
    __syncthreads();
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= 6u * res * res) return;
    const int face = (int)(i / (uint32_t)(res * res)), y = (int)((i / (uint32_t)res) % (uint32_t)res), x = (int)(i % (uint32_t)res);
    const F3 N = normalize3(bake_direction(face, x, y, res));
    const F3 V = N;
    float col[3] = {0.0f, 0.0f, 0.0f}, total = 0.0f;
    const F3 upv = fabsf(N.z) < 0.999f ? f3(0.0f, 0.0f, 1.0f) : f3(1.0f, 0.0f, 0.0f);
    const F3 tangent = normalize3(cross3(upv, N));
    const F3 bitangent = cross3(N, tangent);
    for (int s = 0; s < nSamples; s++) {
        const float hx = lds_h[0][s], hy = lds_h[1][s], hz = lds_h[2][s];
        const F3 H = normalize3(tangent * hx + bitangent * hy + N * hz);
        const F3 L = normalize3(H * (2.0f * dot3(V, H)) - V);
        const float NdotL = fmaxf(dot3(N, L), 0.0f);
        if (NdotL > 0.0f) {
            float e[3];
            cube_sample_bordered(env, 0, L.x, L.y, L.z, e);
#pragma unroll
            for (int k = 0; k < 3; k++) col[k] += e[k] * NdotL;
            total += NdotL;
        }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) col[k] /= total;
    dst[i] = pack_r11g11b10(col[0], col[1], col[2]);
}

// vkCmdBlitImage(LINEAR) from one mip level to the next; sRGB images are
// filtered in linear light. `thr` = 256 encode thresholds (linear value of code n + 0.5).
// One destination texel (x, y):
RDR_DEV uint32_t mip_blit_texel(const uint32_t* __restrict__ src, int sw, int sh, int x, int y, int dw, int dh, int srgb,
                                const float* __restrict__ srgb_decode, const float* __restrict__ unorm_decode, const float* __restrict__ thr) {
    const float scale_x = (float)sw / (float)dw, scale_y = (float)sh / (float)dh;
    const float u = ((float)x + 0.5f) * scale_x - 0.5f, v = ((float)y + 0.5f) * scale_y - 0.5f;
    const float fu = floorf(u), fv = floorf(v);
    const float a = u - fu, b = v - fv;
    const int i0 = min(max((int)fu, 0), sw - 1), i1 = min(max((int)fu + 1, 0), sw - 1);
    const int j0 = min(max((int)fv, 0), sh - 1), j1 = min(max((int)fv + 1, 0), sh - 1);
    const uint32_t q00 = src[j0 * sw + i0], q10 = src[j0 * sw + i1], q01 = src[j1 * sw + i0], q11 = src[j1 * sw + i1];
    uint32_t o = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const bool lin = srgb && k < 3;
        const float* tab = lin ? srgb_decode : unorm_decode;
        const float t00 = tab[(q00 >> (8 * k)) & 255u], t10 = tab[(q10 >> (8 * k)) & 255u];
        const float t01 = tab[(q01 >> (8 * k)) & 255u], t11 = tab[(q11 >> (8 * k)) & 255u];
        const float top = (1.0f - a) * t00 + a * t10, bot = (1.0f - a) * t01 + a * t11;
        const float r = (1.0f - b) * top + b * bot;
        uint32_t code;
        if (lin) {
            if (!(r > 0.0f)) code = 0;
            else {
                int lo = 0, hi = 255;  // number of thresholds <= r among codes 0..254
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (r < thr[mid]) hi = mid; else lo = mid + 1;
                }
                code = (uint32_t)lo;
            }
        } else code = float_to_unorm8(r);
        o |= code << (8 * k);
    }
    return o;
}
__global__ void __launch_bounds__(256) k_mip_blit(const uint32_t* __restrict__ src, int sw, int sh, uint32_t* __restrict__ dst, int dw, int dh,
                                                  int srgb, const float* __restrict__ srgb_decode, const float* __restrict__ unorm_decode,
                                                  const float* __restrict__ thr) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= (uint32_t)(dw * dh)) return;
    dst[i] = mip_blit_texel(src, sw, sh, (int)(i % (uint32_t)dw), (int)(i / (uint32_t)dw), dw, dh, srgb, srgb_decode, unorm_decode, thr);
}
// The small end of a chain in one launch: one block produces levels first + 1 .. last one after the other, each from
// the level before it as stored (a block's own stores are visible to it after the barrier), so that an image costs
// a handful of launches instead of one per level. Levels of at most MIP_TAIL_TEXELS texels are taken this way.
#define MIP_TAIL_TEXELS 4096
struct MipTail {
    uint32_t n;  // levels to produce
    uint32_t off[RDR_MAX_TEX_LEVELS];   // texel offset of the source level, then of every produced level
    uint16_t w[RDR_MAX_TEX_LEVELS], h[RDR_MAX_TEX_LEVELS];
};
__global__ void __launch_bounds__(256) k_mip_blit_tail(uint32_t* __restrict__ base, const MipTail t, int srgb, const float* __restrict__ srgb_decode,
                                                       const float* __restrict__ unorm_decode, const float* __restrict__ thr) {
    for (uint32_t l = 1; l <= t.n; l++) {
        const uint32_t* src = base + t.off[l - 1];
        uint32_t* dst = base + t.off[l];
        const int sw = t.w[l - 1], sh = t.h[l - 1], dw = t.w[l], dh = t.h[l];
        for (uint32_t i = threadIdx.x; i < (uint32_t)(dw * dh); i += 256u)
            dst[i] = mip_blit_texel(src, sw, sh, (int)(i % (uint32_t)dw), (int)(i / (uint32_t)dw), dw, dh, srgb, srgb_decode, unorm_decode, thr);
        __syncthreads();
    }
}

// ---- precompute steps with no counterpart in the reference (it receives six decoded faces and loads its BRDF table
// from a PNG): panorama -> cube faces and the split-sum table.

// atan2 from + - * / and fma only (6-term odd polynomial on [0,1], |error| < 2e-5 rad): the CPU checker evaluates
// the same expression, so both produce the same bits.
RDR_DEV float det_atan2(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    const float a = mx > 0.0f ? mn / mx : 0.0f;
    const float s = a * a;
    float r = fmaf(s, -0.01172120f, 0.05265332f);
    r = fmaf(s, r, -0.11643287f);
    r = fmaf(s, r, 0.19354346f);
    r = fmaf(s, r, -0.33262347f);
    r = fmaf(s, r, 0.99997726f);
    r = r * a;
    if (ay > ax) r = 1.57079637f - r;
    if (x < 0.0f) r = 3.14159274f - r;
    return y < 0.0f ? -r : r;
}

// Direction through the centre of texel (i, j) of a face: the inverse of the face-selection tables of cube_face.
RDR_DEV F3 face_texel_direction(int face, int i, int j, int res) {
    const float sc = ((float)(2 * i + 1)) / (float)res - 1.0f, tc = ((float)(2 * j + 1)) / (float)res - 1.0f;
    switch (face) {
        case 0: return f3(1.0f, -tc, -sc);
        case 1: return f3(-1.0f, -tc, sc);
        case 2: return f3(sc, 1.0f, tc);
        case 3: return f3(sc, -1.0f, -tc);
        case 4: return f3(sc, -tc, 1.0f);
        default: return f3(-sc, -tc, -1.0f);
    }
}

// u = atan2(x, -z) / 2pi + 1/2, v = angle from +Y / pi; bilinear, u wraps, v clamps; writes packed texels
__global__ void __launch_bounds__(256) k_equirect_to_cube(const float4* __restrict__ pano, int W, int H, uint32_t* __restrict__ dst, int res) {
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= 6u * res * res) return;
    const int face = (int)(idx / (uint32_t)(res * res)), j = (int)((idx / (uint32_t)res) % (uint32_t)res), i = (int)(idx % (uint32_t)res);
    const F3 d = face_texel_direction(face, i, j, res);
    const float u = det_atan2(d.x, -d.z) * 0.159154937f + 0.5f;
    const float v = det_atan2(sqrtf(d.x * d.x + d.z * d.z), d.y) * 0.318309873f;
    const float tu = u * (float)W - 0.5f, tv = v * (float)H - 0.5f;
    const float fu = floorf(tu), fv = floorf(tv);
    const float a = tu - fu, b = tv - fv;
    int i0 = (int)fu % W; if (i0 < 0) i0 += W;
    const int i1 = (i0 + 1) % W;
    const int j0 = min(max((int)fv, 0), H - 1), j1 = min(max((int)fv + 1, 0), H - 1);
    const float4 q00 = pano[(size_t)j0 * W + i0], q10 = pano[(size_t)j0 * W + i1];
    const float4 q01 = pano[(size_t)j1 * W + i0], q11 = pano[(size_t)j1 * W + i1];
    const float c00[3] = {q00.x, q00.y, q00.z}, c10[3] = {q10.x, q10.y, q10.z};
    const float c01[3] = {q01.x, q01.y, q01.z}, c11[3] = {q11.x, q11.y, q11.z};
    float o[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float top = (1.0f - a) * c00[k] + a * c10[k], bot = (1.0f - a) * c01[k] + a * c11[k];
        o[k] = (1.0f - b) * top + b * bot;
    }
    dst[idx] = pack_r11g11b10(o[0], o[1], o[2]);
}

// Split-sum BRDF table (Karis 2013) on the Hammersley / GGX helpers of envbakespecular.frag:17-53. Texel (x, y):
// NdotV = (x + .5) / size, roughness = 1 - (y + .5) / size (row 0 = roughness 1, as textures/ibl_brdf_lut.png is
// stored). `trig` = cos then sin of 2 pi s / samples. One lane per texel: the sums are in sample order.
__global__ void __launch_bounds__(256) k_brdf_lut(uint32_t* __restrict__ dst, int size, int samples, const float* __restrict__ trig) {
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= (uint32_t)(size * size)) return;
    const int x = (int)(idx % (uint32_t)size), y = (int)(idx / (uint32_t)size);
    const float nv = ((float)x + 0.5f) / (float)size, rough = 1.0f - ((float)y + 0.5f) / (float)size;
    const float a = rough * rough, k = a * 0.5f;
    const float vx = sqrtf(1.0f - nv * nv), vz = nv;
    float A = 0.0f, B = 0.0f;
    for (int s = 0; s < samples; s++) {
        const float xi = radical_inverse((uint32_t)s);
        const float ct = sqrtf((1.0f - xi) / (1.0f + (a * a - 1.0f) * xi));
        const float st = sqrtf(1.0f - ct * ct);
        const float hx = trig[s] * st, hz = ct;
        float vdh = vx * hx + vz * hz;
        const float lz = 2.0f * vdh * hz - vz;
        const float ndl = fminf(fmaxf(lz, 0.0f), 1.0f), ndh = fminf(fmaxf(hz, 0.0f), 1.0f);
        vdh = fminf(fmaxf(vdh, 0.0f), 1.0f);
        if (ndl > 0.0f) {
            const float g = (ndl / (ndl * (1.0f - k) + k)) * (nv / (nv * (1.0f - k) + k));
            const float gv = g * vdh / fmaxf(ndh * nv, 1e-8f);
            const float om = 1.0f - vdh, om2 = om * om, fc = om2 * om2 * om;
            A += (1.0f - fc) * gv;
            B += fc * gv;
        }
    }
    dst[idx] = float_to_unorm8(A / (float)samples) | (float_to_unorm8(B / (float)samples) << 8) | 0xFF000000u;
}

// One level of a cube with a one-texel border around every face: texel (i, j), i and j in [-1, res], of `face` sits at
// (face * (res + 2) + j + 1) * (res + 2) + i + 1 and holds what seamless filtering reads there (the texel across the
// shared edge; at a corner the clamped-row texel, as cube_fold resolves it). The shading kernels filter from these
// copies: a 2x2 footprint is then always four neighbouring words, whatever face edge it touches.
__global__ void __launch_bounds__(256) k_cube_border(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, int res) {
    const int p = res + 2;
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= 6u * (uint32_t)(p * p)) return;
    const int face = (int)(t / (uint32_t)(p * p)), rem = (int)(t % (uint32_t)(p * p));
    dst[t] = src[cube_texel_offset(face, rem % p - 1, rem / p - 1, res)];
}

// Footprint-major copy of one bordered cube level: entry ((face * (res + 1) + j0 + 1) * (res + 1) + i0 + 1) holds the
// four texels (i0, j0), (i0 + 1, j0), (i0, j0 + 1), (i0 + 1, j0 + 1) of the 2x2 filter footprint whose top-left texel is
// (i0, j0), i0 and j0 in [-1, res - 1]. The fragments of a wave look the image-based term up in directions that the
// normal maps scatter: with one 16-byte entry per look-up every lane touches one cache line where two rows of a
// row-major image cost it two (same texels, same arithmetic, four times the memory: 34 MB for the prefiltered cube).
__global__ void __launch_bounds__(256) k_cube_quads(const uint32_t* __restrict__ bordered, uint4* __restrict__ dst, int res) {
    const int p = res + 2, q = res + 1;
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= 6u * (uint32_t)(q * q)) return;
    const int face = (int)(t / (uint32_t)(q * q)), rem = (int)(t % (uint32_t)(q * q));
    const int I = rem % q, J = rem / q;  // = i0 + 1, j0 + 1: position of the footprint's first texel in the bordered face
    const uint32_t* row = bordered + ((size_t)face * p + J) * p + I;
    dst[t] = make_uint4(row[0], row[1], row[p], row[p + 1]);
}
// The same for the split-sum table, which is addressed with REPEAT: entry (i, j) holds texels (i, j), (i + 1, j),
// (i, j + 1), (i + 1, j + 1) with the indices wrapped.
__global__ void __launch_bounds__(256) k_lut_quads(const uint32_t* __restrict__ src, uint4* __restrict__ dst, int w, int h) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= (uint32_t)(w * h)) return;
    const int i = (int)(t % (uint32_t)w), j = (int)(t / (uint32_t)w);
    const int i1 = i + 1 == w ? 0 : i + 1, j1 = j + 1 == h ? 0 : j + 1;
    dst[t] = make_uint4(src[j * w + i], src[j * w + i1], src[j1 * w + i], src[j1 * w + i1]);
}

// Border of the point-light shadow cubes, in place: face `f` of cube `c` is a (res + 2)^2 image whose interior the
// rasteriser writes; its one-texel border receives the depth across the shared cube edges (cube_fold), so that the
// 2x2 compare footprint of a shadow look-up is four neighbouring words anywhere on the face. One thread per
// (cube, face, side, position along the side); sources are interior texels only, so the order of threads is free.
__global__ void __launch_bounds__(256) k_shadow_cube_border(float* __restrict__ maps, int res, uint32_t n_cubes) {
    const int p = res + 2;
    const uint32_t per_face = 4u * (uint32_t)p, t = blockIdx.x * 256u + threadIdx.x;
    if (t >= n_cubes * 6u * per_face) return;
    const uint32_t cube = t / (6u * per_face), rem = t % (6u * per_face);
    const int face = (int)(rem / per_face), side = (int)((rem % per_face) / (uint32_t)p), k = (int)(rem % (uint32_t)p) - 1;
    const int i = side == 0 ? -1 : (side == 1 ? res : k), j = side == 2 ? -1 : (side == 3 ? res : k);
    int f2, i2, j2;
    cube_fold(face, i, j, res, f2, i2, j2);
    float* base = maps + (size_t)cube * 6u * (size_t)(p * p);
    base[((size_t)face * p + (j + 1)) * p + (i + 1)] = base[((size_t)f2 * p + (j2 + 1)) * p + (i2 + 1)];
}

// fills n floats with v
__global__ void __launch_bounds__(256) k_fill_f32(float* __restrict__ dst, float v, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i < n) dst[i] = v;
}

// The same blit for the SFLOAT formats (one float4 per texel): filtered as it is; `to_half`: the image stores binary16
// (R16G16B16(A16)_SFLOAT), so every level is rounded to it as the blit's store would.
__global__ void __launch_bounds__(256) k_mip_blit_f32(const float4* __restrict__ src, int sw, int sh, float4* __restrict__ dst, int dw, int dh, int to_half) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= (uint32_t)(dw * dh)) return;
    const int x = (int)(i % (uint32_t)dw), y = (int)(i / (uint32_t)dw);
    const float scale_x = (float)sw / (float)dw, scale_y = (float)sh / (float)dh;
    const float u = ((float)x + 0.5f) * scale_x - 0.5f, v = ((float)y + 0.5f) * scale_y - 0.5f;
    const float fu = floorf(u), fv = floorf(v);
    const float a = u - fu, b = v - fv;
    const int i0 = min(max((int)fu, 0), sw - 1), i1 = min(max((int)fu + 1, 0), sw - 1);
    const int j0 = min(max((int)fv, 0), sh - 1), j1 = min(max((int)fv + 1, 0), sh - 1);
    const float4 q00 = src[j0 * sw + i0], q10 = src[j0 * sw + i1], q01 = src[j1 * sw + i0], q11 = src[j1 * sw + i1];
    const float t00[4] = {q00.x, q00.y, q00.z, q00.w}, t10[4] = {q10.x, q10.y, q10.z, q10.w};
    const float t01[4] = {q01.x, q01.y, q01.z, q01.w}, t11[4] = {q11.x, q11.y, q11.z, q11.w};
    float r[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float top = (1.0f - a) * t00[k] + a * t10[k], bot = (1.0f - a) * t01[k] + a * t11[k];
        r[k] = (1.0f - b) * top + b * bot;
        if (to_half) r[k] = quantize_half(r[k]);
    }
    dst[i] = make_float4(r[0], r[1], r[2], r[3]);
}

}  // namespace

void launch_equirect_to_cube(hipStream_t s, const void* pano_rgba32f, int w, int h, uint32_t* dst, int res) {
    hipLaunchKernelGGL(k_equirect_to_cube, dim3((6u * res * res + 255) / 256), dim3(256), 0, s, (const float4*)pano_rgba32f, w, h, dst, res);
}
void launch_brdf_lut(hipStream_t s, uint32_t* dst, int size, int samples, const float* trig) {
    hipLaunchKernelGGL(k_brdf_lut, dim3((size * size + 255) / 256), dim3(256), 0, s, dst, size, samples, trig);
}
void launch_env_pack(hipStream_t s, const void* face_rgba32f, uint32_t* dst, uint32_t n) {
    hipLaunchKernelGGL(k_env_pack, dim3((n + 255) / 256), dim3(256), 0, s, (const float4*)face_rgba32f, dst, n);
}
void launch_shadow_cube_border(hipStream_t s, float* maps, int res, uint32_t n_cubes) {
    const uint32_t n = n_cubes * 6u * 4u * (uint32_t)(res + 2);
    if (n) hipLaunchKernelGGL(k_shadow_cube_border, dim3((n + 255) / 256), dim3(256), 0, s, maps, res, n_cubes);
}
void launch_fill_f32(hipStream_t s, float* dst, float v, size_t n) {
    if (n) hipLaunchKernelGGL(k_fill_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dst, v, n);
}
void launch_cube_border(hipStream_t s, const uint32_t* src, uint32_t* dst, int res) {
    const uint32_t n = 6u * (uint32_t)(res + 2) * (uint32_t)(res + 2);
    hipLaunchKernelGGL(k_cube_border, dim3((n + 255) / 256), dim3(256), 0, s, src, dst, res);
}
void launch_cube_quads(hipStream_t s, const uint32_t* bordered, void* dst, int res) {
    const uint32_t n = 6u * (uint32_t)(res + 1) * (uint32_t)(res + 1);
    hipLaunchKernelGGL(k_cube_quads, dim3((n + 255) / 256), dim3(256), 0, s, bordered, (uint4*)dst, res);
}
void launch_lut_quads(hipStream_t s, const uint32_t* src, void* dst, int w, int h) {
    hipLaunchKernelGGL(k_lut_quads, dim3(((uint32_t)(w * h) + 255) / 256), dim3(256), 0, s, src, (uint4*)dst, w, h);
}
void launch_bake_irradiance(hipStream_t s, const DCube& env, uint32_t* dst, int res, const IrradianceTables& tb) {
    hipLaunchKernelGGL(k_bake_irradiance, dim3((6u * res * res + 255) / 256), dim3(256), 0, s, env, dst, res, tb);
}
void launch_bake_prefilter(hipStream_t s, const DCube& env, uint32_t* dst, int res, float roughness, const float* trig) {
    hipLaunchKernelGGL(k_bake_prefilter, dim3((6u * res * res + 255) / 256), dim3(256), 0, s, env, dst, res, roughness, trig);
}
void launch_mip_chain_f32(hipStream_t s, float* base, const uint32_t* level_offset, uint32_t w, uint32_t h, uint32_t levels, bool to_half) {
    auto dim = [](uint32_t v, uint32_t l) { return (int)std::max(1u, v >> l); };
    float4* texels = (float4*)base;
    for (uint32_t l = 1; l < levels; l++)
        hipLaunchKernelGGL(k_mip_blit_f32, dim3((dim(w, l) * dim(h, l) + 255) / 256), dim3(256), 0, s, texels + level_offset[l - 1], dim(w, l - 1),
                           dim(h, l - 1), texels + level_offset[l], dim(w, l), dim(h, l), to_half ? 1 : 0);
}
void launch_mip_chain(hipStream_t s, uint32_t* base, const uint32_t* level_offset, uint32_t w, uint32_t h, uint32_t levels, int srgb,
                      const float* srgb_decode, const float* unorm_decode, const float* thr) {
    auto dim = [](uint32_t v, uint32_t l) { return (int)std::max(1u, v >> l); };
    uint32_t l = 1;
    // the large levels one launch each, until the source level is small enough for one block
    for (; l < levels && (size_t)dim(w, l - 1) * dim(h, l - 1) > MIP_TAIL_TEXELS; l++)
        hipLaunchKernelGGL(k_mip_blit, dim3((dim(w, l) * dim(h, l) + 255) / 256), dim3(256), 0, s, base + level_offset[l - 1], dim(w, l - 1), dim(h, l - 1),
                           base + level_offset[l], dim(w, l), dim(h, l), srgb, srgb_decode, unorm_decode, thr);
    if (l >= levels) return;
    MipTail t;
    memset(&t, 0, sizeof t);
    t.n = levels - l;
    for (uint32_t k = 0; k <= t.n; k++) { t.off[k] = level_offset[l - 1 + k]; t.w[k] = (uint16_t)dim(w, l - 1 + k); t.h[k] = (uint16_t)dim(h, l - 1 + k); }
    hipLaunchKernelGGL(k_mip_blit_tail, dim3(1), dim3(256), 0, s, base, t, srgb, srgb_decode, unorm_decode, thr);
}
