// TEST INFRASTRUCTURE — CPU restatement ("oracle"); see orc_types.h.
//
// Main colour pass: basic.vert + pbr.frag per fragment, skybox (env.vert +
// env.frag), blended draws, MSAA resolve
// (Renderer/VulkanRenderer.cpp:1043-1065; pipeline state :836-866;
// Renderer/VulkanRendererEnvironment.cpp:40-47).
//
// Attribute interpolation: perspective-correct barycentrics of the UNCLIPPED
// triangle, evaluated at the pixel centre (no sample shading, no centroid)
// from the adjugate of the homogeneous vertex matrix, in double. Quantities
// that feed a discontinuous decision (shadow compares, cascade choice, light
// cut-offs) are computed with correctly rounded +,-,*,/,sqrt,fma in the order
// written here, which the HIP path repeats operation for operation.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "orc_types.h"

namespace orc {

struct F4 { float x, y, z, w; };

static inline F4 xform4(const M4& m, V3 p) {
    F4 r;
    r.x = std::fmaf(m.c[0][0], p.x, std::fmaf(m.c[1][0], p.y, std::fmaf(m.c[2][0], p.z, m.c[3][0])));
    r.y = std::fmaf(m.c[0][1], p.x, std::fmaf(m.c[1][1], p.y, std::fmaf(m.c[2][1], p.z, m.c[3][1])));
    r.z = std::fmaf(m.c[0][2], p.x, std::fmaf(m.c[1][2], p.y, std::fmaf(m.c[2][2], p.z, m.c[3][2])));
    r.w = std::fmaf(m.c[0][3], p.x, std::fmaf(m.c[1][3], p.y, std::fmaf(m.c[2][3], p.z, m.c[3][3])));
    return r;
}
static inline V3 xform_dir(const M4& m, V3 d) {  // (model * vec4(d, 0)).xyz
    V3 r;
    r.x = std::fmaf(m.c[0][0], d.x, std::fmaf(m.c[1][0], d.y, m.c[2][0] * d.z));
    r.y = std::fmaf(m.c[0][1], d.x, std::fmaf(m.c[1][1], d.y, m.c[2][1] * d.z));
    r.z = std::fmaf(m.c[0][2], d.x, std::fmaf(m.c[1][2], d.y, m.c[2][2] * d.z));
    return r;
}
static inline float interp(const float lam[3], float a0, float a1, float a2) {
    return std::fmaf(lam[2], a2, std::fmaf(lam[1], a1, lam[0] * a0));
}
static inline float clampf(float v, float lo, float hi) { return std::min(std::max(v, lo), hi); }

// sampler2DArrayShadow with the shadow sampler of VulkanRendererShadow.cpp:73:
// LINEAR, compare LESS_OR_EQUAL, CLAMP_TO_BORDER, border = opaque white (1.0).
static float shadow_fetch(const float* map, int res, int i, int j, float ref) {
    const float d = (i < 0 || j < 0 || i >= res || j >= res) ? 1.0f : map[(size_t)j * res + i];
    return ref <= d ? 1.0f : 0.0f;
}
static float shadow_pcf(const float* map, int res, float s, float t, float ref) {
    const float tu = s * (float)res - 0.5f, tv = t * (float)res - 0.5f;
    const float fu = std::floor(tu), fv = std::floor(tv);
    const float a = tu - fu, b = tv - fv;
    // keep far-off coordinates inside int range; everything outside is border anyway
    const int i0 = (int)clampf(fu, -2.0f, (float)res + 1.0f), j0 = (int)clampf(fv, -2.0f, (float)res + 1.0f);
    const float r00 = shadow_fetch(map, res, i0, j0, ref), r10 = shadow_fetch(map, res, i0 + 1, j0, ref);
    const float r01 = shadow_fetch(map, res, i0, j0 + 1, ref), r11 = shadow_fetch(map, res, i0 + 1, j0 + 1, ref);
    const float top = (1.0f - a) * r00 + a * r10, bot = (1.0f - a) * r01 + a * r11;
    return (1.0f - b) * top + b * bot;
}
// samplerCubeArrayShadow: face selection per the Vulkan tables, bilinear PCF;
// texels one step outside the face come from the neighbouring face (seamless).
static float shadow_cube(const Ctx& c, uint32_t slot, float rx, float ry, float rz, float ref) {
    const int res = (int)c.settings.point_shadow_map_resolution;
    const float ax = std::fabs(rx), ay = std::fabs(ry), az = std::fabs(rz);
    int face; float sc, tc, ma;
    if (az >= ax && az >= ay) { face = rz < 0.0f ? 5 : 4; sc = rz < 0.0f ? -rx : rx; tc = -ry; ma = az; }
    else if (ay >= ax) { face = ry < 0.0f ? 3 : 2; sc = rx; tc = ry < 0.0f ? -rz : rz; ma = ay; }
    else { face = rx < 0.0f ? 1 : 0; sc = rx < 0.0f ? rz : -rz; tc = -ry; ma = ax; }
    const float s = 0.5f * (sc / ma) + 0.5f, t = 0.5f * (tc / ma) + 0.5f;
    const float* map = c.point_shadow.data() + (size_t)slot * 6 * res * res;  // the six faces of this cube
    const float tu = s * (float)res - 0.5f, tv = t * (float)res - 0.5f;
    const float fu = std::floor(tu), fv = std::floor(tv);
    const float a = tu - fu, b = tv - fv;
    const int i0 = std::min(std::max((int)fu, -1), res - 1), j0 = std::min(std::max((int)fv, -1), res - 1);
    const float r00 = ref <= map[cube_texel_offset(face, i0, j0, res)] ? 1.0f : 0.0f, r10 = ref <= map[cube_texel_offset(face, i0 + 1, j0, res)] ? 1.0f : 0.0f;
    const float r01 = ref <= map[cube_texel_offset(face, i0, j0 + 1, res)] ? 1.0f : 0.0f, r11 = ref <= map[cube_texel_offset(face, i0 + 1, j0 + 1, res)] ? 1.0f : 0.0f;
    const float top = (1.0f - a) * r00 + a * r10, bot = (1.0f - a) * r01 + a * r11;
    return (1.0f - b) * top + b * bot;
}

// pbr.frag:211-269
static const float kPI = 3.14159274f;  // radians(180)
static float DistributionGGX(V3 N, V3 H, float roughness) {
    const float a = roughness * roughness, a2 = a * a;
    const float NdotH = clampf(dot(N, H), 0.0f, 1.0f), NdotH2 = NdotH * NdotH;
    float denom = NdotH2 * (a2 - 1.0f) + 1.0f;
    denom = kPI * denom * denom;
    return a2 / denom;
}
static float GeometrySchlickGGX(float NdotV, float roughness) {
    const float r = roughness + 1.0f, k = (r * r) / 8.0f;
    return NdotV / (NdotV * (1.0f - k) + k);
}
static float GeometrySmith(V3 N, V3 V, V3 L, float roughness) {
    const float NdotV = std::max(dot(N, V), 0.001f), NdotL = std::max(dot(N, L), 0.001f);
    return GeometrySchlickGGX(NdotL, roughness) * GeometrySchlickGGX(NdotV, roughness);
}
static V3 fresnelSchlick(float cosTheta, V3 F0) {
    const float x = 1.0f - clampf(cosTheta, 0.0f, 1.0f);
    const float x2 = x * x, x5 = x2 * x2 * x;  // pow(x, 5.0)
    return {F0.x + (1.0f - F0.x) * x5, F0.y + (1.0f - F0.y) * x5, F0.z + (1.0f - F0.z) * x5};
}
static V3 BRDF(V3 radiance, V3 L, V3 N, V3 V, V3 albedo, float roughness, float metallic, V3 F0) {
    const V3 H = normalize(V + L);
    const float NDF = DistributionGGX(N, H, roughness);
    const float G = GeometrySmith(N, V, L, roughness);
    const V3 F = fresnelSchlick(std::max(dot(H, V), 0.0f), F0);
    const float denominator = std::max(4.0f * std::max(dot(N, V), 1e-3f) * std::max(dot(N, L), 1e-3f), 1e-5f);
    const float NdotL = std::max(dot(N, L), 0.0f);
    const float Fk[3] = {F.x, F.y, F.z}, al[3] = {albedo.x, albedo.y, albedo.z}, rad[3] = {radiance.x, radiance.y, radiance.z};
    float o[3];
    for (int k = 0; k < 3; k++) {
        const float kD = (1.0f - Fk[k]) * (1.0f - metallic);
        const float specular = (NDF * G * Fk[k]) / denominator;
        o[k] = (kD * al[k] / kPI + specular) * rad[k] * NdotL;
    }
    return {o[0], o[1], o[2]};
}

struct ShadeCtx {
    const Ctx& c;
    M4 inv_viewproj;
    V3 cam;
    float inv2W, inv2H;
};

// env.vert:22-26 + env.frag:10-17 at pixel (px,py)
static void shade_sky(const ShadeCtx& sc, int px, int py, float out[4]) {
    const float xn = std::fmaf((float)px + 0.5f, sc.inv2W, -1.0f), yn = std::fmaf((float)py + 0.5f, sc.inv2H, -1.0f);
    const M4& m = sc.inv_viewproj;
    // (invViewProj * vec4(xn, yn, 1, 1)).xyz — no perspective divide
    V3 pos;
    pos.x = std::fmaf(m.c[0][0], xn, std::fmaf(m.c[1][0], yn, m.c[2][0] + m.c[3][0]));
    pos.y = std::fmaf(m.c[0][1], xn, std::fmaf(m.c[1][1], yn, m.c[2][1] + m.c[3][1]));
    pos.z = std::fmaf(m.c[0][2], xn, std::fmaf(m.c[1][2], yn, m.c[2][2] + m.c[3][2]));
    const V3 d = normalize(pos);
    float e[3] = {0, 0, 0};
    if (sc.c.env.res) cube_sample(sc.c.env, 0, d.x, d.y, d.z, e);
    for (int k = 0; k < 3; k++) {
        float v = e[k] / (e[k] + 1.0f);
        out[k] = std::pow(v, 1.0f / 2.2f);
    }
    out[3] = 1.0f;
}

// basic.vert:36-49 for the three vertices + pbr.frag:93-209 at the centre of
// pixel (px,py). Writes outColor.
static void shade_fragment(const ShadeCtx& sc, const Draw& d, uint32_t tri, int px, int py, float out[4]) {
    const Ctx& c = sc.c;
    const Primitive& pr = c.prims[d.prim];
    const uint32_t vi[3] = {pr.idx[3 * tri], pr.idx[3 * tri + 1], pr.idx[3 * tri + 2]};
    F4 cp[3];
    for (int k = 0; k < 3; k++) cp[k] = xform4(d.mvp, pr.pos[vi[k]]);

    // ---- perspective-correct barycentrics at (px,py), (px+1,py), (px,py+1)
    double A[3], B[3], C[3];
    for (int k = 0; k < 3; k++) {
        const F4& p = cp[(k + 1) % 3];
        const F4& q = cp[(k + 2) % 3];
        A[k] = (double)p.y * (double)q.w - (double)q.y * (double)p.w;
        B[k] = (double)q.x * (double)p.w - (double)p.x * (double)q.w;
        C[k] = (double)p.x * (double)q.y - (double)q.x * (double)p.y;
    }
    const float xs[3] = {std::fmaf((float)px + 0.5f, sc.inv2W, -1.0f), std::fmaf((float)px + 1.5f, sc.inv2W, -1.0f), 0.0f};
    const float ys[3] = {std::fmaf((float)py + 0.5f, sc.inv2H, -1.0f), 0.0f, std::fmaf((float)py + 1.5f, sc.inv2H, -1.0f)};
    float lam[3][3];  // [position][vertex]
    for (int e = 0; e < 3; e++) {
        const double xn = e == 1 ? xs[1] : xs[0], yn = e == 2 ? ys[2] : ys[0];
        double l[3];
        for (int k = 0; k < 3; k++) l[k] = std::fma(A[k], xn, std::fma(B[k], yn, C[k]));
        const double inv = 1.0 / ((l[0] + l[1]) + l[2]);
        for (int k = 0; k < 3; k++) lam[e][k] = (float)(l[k] * inv);
    }

    // ---- vertex shader outputs
    V3 wp[3], nrm[3], tan[3], bit[3];
    for (int k = 0; k < 3; k++) {
        const F4 w4 = xform4(d.model, pr.pos[vi[k]]);
        wp[k] = w4.w == 1.0f ? V3{w4.x, w4.y, w4.z} : V3{w4.x / w4.w, w4.y / w4.w, w4.z / w4.w};
        const VertexAttr& a = pr.attr[vi[k]];
        nrm[k] = normalize(xform_dir(d.model, a.normal));
        tan[k] = normalize(xform_dir(d.model, a.tangent));
        bit[k] = normalize(xform_dir(d.model, a.bitangent));
    }
    const V3 P{interp(lam[0], wp[0].x, wp[1].x, wp[2].x), interp(lam[0], wp[0].y, wp[1].y, wp[2].y), interp(lam[0], wp[0].z, wp[1].z, wp[2].z)};
    const V3 Nv{interp(lam[0], nrm[0].x, nrm[1].x, nrm[2].x), interp(lam[0], nrm[0].y, nrm[1].y, nrm[2].y), interp(lam[0], nrm[0].z, nrm[1].z, nrm[2].z)};
    const V3 Tv{interp(lam[0], tan[0].x, tan[1].x, tan[2].x), interp(lam[0], tan[0].y, tan[1].y, tan[2].y), interp(lam[0], tan[0].z, tan[1].z, tan[2].z)};
    const V3 Bv{interp(lam[0], bit[0].x, bit[1].x, bit[2].x), interp(lam[0], bit[0].y, bit[1].y, bit[2].y), interp(lam[0], bit[0].z, bit[1].z, bit[2].z)};
    float uv[3][2];
    for (int e = 0; e < 3; e++) {
        uv[e][0] = interp(lam[e], pr.attr[vi[0]].u, pr.attr[vi[1]].u, pr.attr[vi[2]].u);
        uv[e][1] = interp(lam[e], pr.attr[vi[0]].v, pr.attr[vi[1]].v, pr.attr[vi[2]].v);
    }
    const float dudx = uv[1][0] - uv[0][0], dvdx = uv[1][1] - uv[0][1], dudy = uv[2][0] - uv[0][0], dvdy = uv[2][1] - uv[0][1];
    const rdr_material& mt = pr.mat;
    auto tex = [&](uint32_t id, uint32_t dflt, float o[4]) {
        sample_texture(c, id == RDR_INVALID_ID ? dflt : id, uv[0][0], uv[0][1], dudx, dvdx, dudy, dvdy, o);
    };

    // ---- pbr.frag main()
    float t[4];
    tex(mt.albedo_texture, c.tex_white, t);
    const float albedo[4] = {t[0] * mt.base_color_factor[0], t[1] * mt.base_color_factor[1], t[2] * mt.base_color_factor[2],
                             t[3] * mt.base_color_factor[3]};
    if (mt.alpha_mode == RDR_ALPHA_MASK && albedo[3] < mt.alpha_cutoff) {
        out[0] = out[1] = out[2] = out[3] = 0.0f;
        return;
    }
    tex(mt.metal_roughness_texture, c.tex_white, t);
    const float metallic = clampf(t[2] * mt.metallic_factor, 0.0f, 1.0f);
    const float roughness = clampf(t[1] * mt.roughness_factor, 0.05f, 1.0f);
    tex(mt.ao_texture, c.tex_white, t);
    const float ao = t[0] * mt.ao_factor;
    tex(mt.emissive_texture, c.tex_white, t);
    const float emissive[3] = {t[0] * mt.emissive_factor[0], t[1] * mt.emissive_factor[1], t[2] * mt.emissive_factor[2]};
    tex(mt.normal_texture, c.tex_normal, t);
    const V3 Tn = normalize(Tv), Bn = normalize(Bv), Nn = normalize(Nv);
    const float nx = t[0] * 2.0f - 1.0f, ny = t[1] * 2.0f - 1.0f, nz = t[2] * 2.0f - 1.0f;
    const V3 N = normalize(Tn * nx + Bn * ny + Nn * nz);  // TBN * n
    const V3 V = normalize(sc.cam - P);
    const V3 negV = V * -1.0f;
    const V3 Rv = negV - N * (2.0f * dot(N, negV));  // reflect(-V, N)
    const V3 alb{albedo[0], albedo[1], albedo[2]};
    const V3 F0{0.04f * (1.0f - metallic) + alb.x * metallic, 0.04f * (1.0f - metallic) + alb.y * metallic,
                0.04f * (1.0f - metallic) + alb.z * metallic};
    V3 Lo{0, 0, 0};

    // point lights (pbr.frag:123-160). Two expressions are restated in an algebraically equal form so that the
    // discontinuous tests need one correctly rounded divide instead of a sqrt and five divides:
    //   radiance = I / distance^2 with distance^2 = dot(Lv, Lv); "max(radiance) <= 1e-3" is tested as
    //   max(I) <= 1e-3 * distance^2;
    //   (far - far*near/d) / (far - near) = c0 - c1/d with c0 = far/(far-near), c1 = far*near/(far-near).
    const float far = 100.0f, near = 0.1f;
    const float c0 = far / (far - near), c1 = (far * near) / (far - near);
    static const char* dbg_env = std::getenv("ORC_DEBUG_PIXEL");
    int dbg_x = -1, dbg_y = -1;
    if (dbg_env) std::sscanf(dbg_env, "%d,%d", &dbg_x, &dbg_y);
    const bool dbg = px == dbg_x && py == dbg_y;
    if (dbg) std::fprintf(stderr, "[orc] px %d %d tri %u P %.9g %.9g %.9g N %.9g %.9g %.9g rough %.9g metal %.9g\n", px, py, tri, P.x, P.y, P.z, N.x, N.y, N.z, roughness, metallic);
    if (dbg) for (int k = 0; k < 3; k++) std::fprintf(stderr, "[orc]   vertex %d: normal %.9g %.9g %.9g tangent %.9g %.9g %.9g bitangent %.9g %.9g %.9g -> %.9g %.9g %.9g lam %.9g\n", k, pr.attr[vi[k]].normal.x, pr.attr[vi[k]].normal.y, pr.attr[vi[k]].normal.z, pr.attr[vi[k]].tangent.x, pr.attr[vi[k]].tangent.y, pr.attr[vi[k]].tangent.z, pr.attr[vi[k]].bitangent.x, pr.attr[vi[k]].bitangent.y, pr.attr[vi[k]].bitangent.z, nrm[k].x, nrm[k].y, nrm[k].z, lam[0][k]);
    if (dbg) std::fprintf(stderr, "[orc]   T %.9g %.9g %.9g B %.9g %.9g %.9g Nv %.9g %.9g %.9g normal texel %.9g %.9g %.9g\n", Tv.x, Tv.y, Tv.z, Bv.x, Bv.y, Bv.z, Nv.x, Nv.y, Nv.z, nx, ny, nz);
    for (const PointLight& pl : c.lights) {
        if (dbg) std::fprintf(stderr, "[orc]   Lo %.9g %.9g %.9g (before light at %.3g %.3g %.3g)\n", Lo.x, Lo.y, Lo.z, pl.point.x, pl.point.y, pl.point.z);
        const V3 Lv = pl.point - P;
        const float d2 = dot(Lv, Lv);
        if (std::max(pl.intensity.x, std::max(pl.intensity.y, pl.intensity.z)) <= 1e-3f * d2) continue;
        const V3 radiance = pl.intensity * (1.0f / d2);
        const bool castShadow = pl.flags & RDR_POINT_LIGHT_CAST_SHADOWS, hasShadowMap = pl.flags & RDR_POINT_LIGHT_HAS_SHADOW_MAP;
        if (castShadow && hasShadowMap) {
            float lightDepth = std::max(std::fabs(Lv.x), std::max(std::fabs(Lv.y), std::fabs(Lv.z))) - 0.05f;
            lightDepth = c0 - c1 / lightDepth;
            const float shadowTest = shadow_cube(c, pl.shadow_map_index, -Lv.x, -Lv.y, -Lv.z, lightDepth);
            if (dbg) std::fprintf(stderr, "[orc]     slot %u ref %.9g shadowTest %.9g\n", pl.shadow_map_index, lightDepth, shadowTest);
            if (shadowTest < 1e-3f) continue;
            const V3 r = BRDF(radiance, normalize(Lv), N, V, alb, roughness, metallic, F0);
            Lo = Lo + r * shadowTest;
        } else {
            Lo = Lo + BRDF(radiance, Lv, N, V, alb, roughness, metallic, F0);  // L deliberately not normalised (pbr.frag:158)
        }
    }

    // directional light (pbr.frag:163-183)
    if (c.lights_set) {
        const int ncasc = (int)c.settings.directional_shadow_cascade_levels;
        const int res = (int)c.settings.directional_shadow_map_resolution;
        int split = 0;
        float ls[3] = {0, 0, 0};
        for (; split < ncasc; split++) {
            // basic.vert:45-48, evaluated on the interpolated world position (the light camera is affine)
            const F4 lsp = xform4(c.csm_viewproj[split], P);
            ls[0] = lsp.x * 0.5f + 0.5f; ls[1] = lsp.y * 0.5f + 0.5f; ls[2] = lsp.z;
            if (lsp.w != 1.0f) { ls[0] /= lsp.w; ls[1] /= lsp.w; ls[2] /= lsp.w; }
            if (std::min(ls[0], ls[1]) > 1e-2f && std::max(ls[0], ls[1]) < 1.0f - 1e-2f) break;
        }
        const int layer = std::min(split, ncasc - 1);  // array layer is clamped by the sampler
        const float shadowTest = shadow_pcf(c.dir_shadow.data() + (size_t)layer * res * res, res, ls[0], ls[1], ls[2]);
        if (dbg) std::fprintf(stderr, "[orc]   Lo %.9g %.9g %.9g before dir; layer %d ls %.9g %.9g %.9g shadowTest %.9g\n", Lo.x, Lo.y, Lo.z, layer, ls[0], ls[1], ls[2], shadowTest);
        if (shadowTest > 1e-3f) {
            const Quat q{c.dir_light.orientation_wxyz[0], c.dir_light.orientation_wxyz[1], c.dir_light.orientation_wxyz[2], c.dir_light.orientation_wxyz[3]};
            const V3 L = normalize(rotate(q, V3{0.0f, 0.0f, 1.0f}));
            const V3 I{c.dir_light.intensity[0], c.dir_light.intensity[1], c.dir_light.intensity[2]};
            Lo = Lo + BRDF(I, L, N, V, alb, roughness, metallic, F0) * shadowTest;
        }
    }

    // environment (pbr.frag:186-200)
    if (c.prefiltered.res && c.irradiance.res && c.lut_w) {
        const float NdotV = std::max(dot(N, V), 1e-3f);
        const float lod = std::min(roughness * (float)c.prefiltered.levels, (float)(c.prefiltered.levels - 1));
        const float fl = std::floor(lod), ff = lod - fl;
        const uint32_t l0 = (uint32_t)fl, l1 = std::min(l0 + 1, c.prefiltered.levels - 1);
        float s0[3], s1[3], spec[3];
        cube_sample(c.prefiltered, l0, Rv.x, Rv.y, Rv.z, s0);
        cube_sample(c.prefiltered, l1, Rv.x, Rv.y, Rv.z, s1);
        for (int k = 0; k < 3; k++) spec[k] = (1.0f - ff) * s0[k] + ff * s1[k];
        // textureLod(brdfLUT, vec2(NdotV, roughness), 0): bilinear, REPEAT wrap (material sampler)
        float lut[2];
        {
            const int w = (int)c.lut_w, h = (int)c.lut_h;
            const float tu = NdotV * (float)w - 0.5f, tv = roughness * (float)h - 0.5f;
            const float fu = std::floor(tu), fv = std::floor(tv);
            const float a = tu - fu, b = tv - fv;
            auto wrap = [](int i, int n) { int r = i % n; return r < 0 ? r + n : r; };
            const int i0 = wrap((int)fu, w), i1 = wrap((int)fu + 1, w), j0 = wrap((int)fv, h), j1 = wrap((int)fv + 1, h);
            for (int k = 0; k < 2; k++) {
                const float q00 = (float)c.lut[((size_t)j0 * w + i0) * 4 + k] / 255.0f, q10 = (float)c.lut[((size_t)j0 * w + i1) * 4 + k] / 255.0f;
                const float q01 = (float)c.lut[((size_t)j1 * w + i0) * 4 + k] / 255.0f, q11 = (float)c.lut[((size_t)j1 * w + i1) * 4 + k] / 255.0f;
                const float top = (1.0f - a) * q00 + a * q10, bot = (1.0f - a) * q01 + a * q11;
                lut[k] = (1.0f - b) * top + b * bot;
            }
        }
        const V3 F = fresnelSchlick(NdotV, F0);
        float irr[3];
        cube_sample(c.irradiance, 0, N.x, N.y, N.z, irr);
        const float Fk[3] = {F.x, F.y, F.z};
        float add[3];
        for (int k = 0; k < 3; k++) {
            const float kD = (1.0f - Fk[k]) * (1.0f - metallic);
            const float envSpecular = spec[k] * (Fk[k] * lut[0] + lut[1]);
            const float envDiffuse = kD * irr[k] * albedo[k];
            add[k] = (envDiffuse + envSpecular) * ao;
        }
        Lo = Lo + V3{add[0], add[1], add[2]};
    }

    if (dbg) std::fprintf(stderr, "[orc]   Lo %.9g %.9g %.9g final, emissive %.9g %.9g %.9g\n", Lo.x, Lo.y, Lo.z, emissive[0], emissive[1], emissive[2]);
    out[0] = std::min(Lo.x + emissive[0], 10.0f);
    out[1] = std::min(Lo.y + emissive[1], 10.0f);
    out[2] = std::min(Lo.z + emissive[2], 10.0f);
    out[3] = mt.alpha_mode == RDR_ALPHA_MASK ? 1.0f : albedo[3];
}

static const Draw& find_draw(const std::vector<Draw>& draws, uint32_t id) {
    size_t lo = 0, hi = draws.size();
    while (hi - lo > 1) {
        size_t mid = (lo + hi) / 2;
        if (draws[mid].first_tri <= id) lo = mid; else hi = mid;
    }
    return draws[lo];
}

// Blend pipeline (VulkanRenderer.cpp:865): colour = src*srcA + dst*(1-srcA);
// alpha = 1*srcA - 0*dstA. The colour attachment is RGBA16F.
struct BlendSink : FragmentSink {
    Ctx& c; const ShadeCtx& sc; const std::vector<Draw>& draws;
    int last_px = -1, last_py = -1; uint32_t last_id = RDR_INVALID_ID;
    float col[4];
    BlendSink(Ctx& c_, const ShadeCtx& sc_, const std::vector<Draw>& d) : c(c_), sc(sc_), draws(d) {}
    void fragment(int px, int py, int s, uint32_t id) override {
        if (px != last_px || py != last_py || id != last_id) {
            const Draw& d = find_draw(draws, id);
            shade_fragment(sc, d, id - d.first_tri, px, py, col);
            last_px = px; last_py = py; last_id = id;
        }
        float* dst = &c.sample_color[(((size_t)s * c.H + py) * c.W + px) * 4];
        const float a = col[3];
        for (int k = 0; k < 3; k++) dst[k] = quantize_half(col[k] * a + dst[k] * (1.0f - a));
        dst[3] = quantize_half(a);
    }
};
struct BlendSinkFactory { Ctx* c; const ShadeCtx* sc; const std::vector<Draw>* draws; };
static FragmentSink* make_blend_sink(void* user, int) {
    BlendSinkFactory* f = (BlendSinkFactory*)user;
    return new BlendSink(*f->c, *f->sc, *f->draws);
}

void shade_main(Ctx& c, const std::vector<Draw>& opaque, const std::vector<Draw>& blended, const M4& viewproj, V3 camera_pos) {
    const int W = (int)c.W, H = (int)c.H, S = (int)c.settings.msaa;
    ShadeCtx sc{c, inverse(viewproj), camera_pos, 2.0f / (float)W, 2.0f / (float)H};
    c.sample_color.assign((size_t)S * W * H * 4, 0.0f);
    c.hdr.assign((size_t)W * H * 4, 0.0f);

    // opaque draws, then the skybox where depth is still 1.0 (LESS_OR_EQUAL at z = 1)
#pragma omp parallel for schedule(dynamic, 4)
    for (int py = 0; py < H; py++)
        for (int px = 0; px < W; px++) {
            uint32_t ids[4]; float cols[4][4]; int n_unique = 0;
            bool have_sky = false; float sky[4];
            for (int s = 0; s < S; s++) {
                const size_t o = ((size_t)s * H + py) * W + px;
                const uint32_t id = c.vis[o];
                const float* src;
                if (id == RDR_INVALID_ID || c.depth[o] >= 1.0f) {
                    if (!have_sky) { shade_sky(sc, px, py, sky); have_sky = true; }
                    src = sky;
                } else {
                    int u = 0;
                    for (; u < n_unique; u++) if (ids[u] == id) break;
                    if (u == n_unique) {
                        const Draw& d = find_draw(opaque, id);
                        shade_fragment(sc, d, id - d.first_tri, px, py, cols[u]);
                        ids[u] = id; n_unique++;
                    }
                    src = cols[u];
                }
                float* dst = &c.sample_color[o * 4];
                for (int k = 0; k < 4; k++) dst[k] = quantize_half(src[k]);
            }
        }

    // blended draws, back to front, depth test LESS + depth write (VulkanRenderer.cpp:857,1059-1062)
    if (!blended.empty()) {
        BlendSinkFactory f{&c, &sc, &blended};
        RasterTarget rt{c.W, c.H, (uint32_t)S, c.depth.data(), nullptr, make_blend_sink, &f};
        raster_draws(c, blended, rt, CULL_BACK, &c.stats.triangle_instances, &c.stats.triangles_binned);
    }

    // multisample resolve: average of the samples, stored as RGBA16F
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)W * H; i++)
        for (int k = 0; k < 4; k++) {
            if (S == 1) { c.hdr[(size_t)i * 4 + k] = c.sample_color[(size_t)i * 4 + k]; continue; }
            float sum = 0.0f;
            for (int s = 0; s < S; s++) sum += c.sample_color[((size_t)s * W * H + (size_t)i) * 4 + k];
            c.hdr[(size_t)i * 4 + k] = quantize_half(sum * (1.0f / (float)S));
        }
}

}  // namespace orc
