// Main colour pass from the visibility buffer: basic.vert + pbr.frag per
// pixel, the skybox (env.vert + env.frag), the in-order blended pass
// (k_blend_collect -> k_blend_shade -> k_blend_resolve, described above them)
// and the multisample resolve.
//
// Reference: Renderer/shaders/basic.vert:36-49, pbr.frag:93-269, env.vert:22-26,
// env.frag:10-17; pass order and blend state Renderer/VulkanRenderer.cpp:1043-1065,
// :862-866.
//
// Structure of the opaque pass (k_shade_setup -> fragment lists -> k_shade_frags -> k_frag_resolve):
//  * k_shade_setup, one thread per main-view triangle that survived set-up:
//    runs basic.vert for its three vertices once and stores the outputs plus
//    the perspective-correct barycentric planes in a 384-byte DShadeTri, so a
//    fragment never re-transforms vertices.
//  * With 4x MSAA a pixel is shaded once per DISTINCT triangle among its
//    samples. Those (triangle, pixel) fragments of the whole frame are
//    collected in lists (by the 4x rasteriser, kernels_raster.hip, or by
//    k_frag_classify from the visibility image), k_shade_frags shades the
//    lists densely - every lane a fragment, no tile, no barrier - and
//    k_frag_resolve turns the fragment colours into pixels (described in front
//    of the three kernels below).
// All scene data is read in place from HBM (geometry arena, RGBA8 mip chains,
// packed-float cubes, D32F shadow arrays).
//
// Values that feed a discontinuous decision (shadow compares, cascade choice,
// light cut-offs) are computed with correctly rounded operations in a fixed
// order: barycentrics in double, world position and light-space coordinates
// with explicit fma chains, the shadow reference depth with an IEEE divide.
// Everything downstream of those is continuous in its inputs (bilinear PCF
// weights, filtered colours, the BRDF) and may use rcp / rsq / contraction.
#include "device_common.h"
#include "kernels.h"
#include "raster_common.h"

// Diagnostic builds only: every lane of a gather reads the same place (bit 0 material textures, bit 1 point-light
// shadow maps, bit 2 image-based lighting, bit 3 cascades): the arithmetic stays, the memory path sees one line per
// load. The frames are wrong; the kernel's time tells what the scattered addresses cost.
#ifndef RDR_DIAG_COALESCE
#define RDR_DIAG_COALESCE 0
#endif
#define RDR_DIAG_ZERO(fr) ((uint32_t)(fr).n_cascades >> 16)  // zero, but not to the compiler
#ifndef RDR_BRDF_ONE_RCP
#define RDR_BRDF_ONE_RCP 1  // 0: a reciprocal per denominator of the specular term (measurement aid)
#endif

namespace {

RDR_DEV float interp3(const float lam[3], float a0, float a1, float a2) { return fmaf(lam[2], a2, fmaf(lam[1], a1, lam[0] * a0)); }

// dot product as two fused multiply-adds: for the continuous parts of shading only (the exact paths use dot3)
RDR_DEV float dot3f(F3 a, F3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
RDR_DEV F3 fast_normalize(F3 a) { return a * __builtin_amdgcn_rsqf(dot3f(a, a)); }
RDR_DEV float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// sampler2DArrayShadow: LINEAR + compare LESS_OR_EQUAL + CLAMP_TO_BORDER(white) (VulkanRendererShadow.cpp:73)
typedef float rdr_float2_a4 __attribute__((ext_vector_type(2), aligned(4)));
RDR_DEV void load_pair_f(const RDR_GLOBAL float* p, float& a, float& b) {
    const rdr_float2_a4 v = *(const RDR_GLOBAL rdr_float2_a4*)p;
    a = v.x; b = v.y;
}
// `map` is the cascade's (res + 4)^2 image with two texels of 1.0 around the interior: a footprint that leaves the
// map reads the border value there, as CLAMP_TO_BORDER does, without a test.
RDR_DEV float shadow_pcf(const RDR_GLOBAL float* __restrict__ map, int res, float s, float t, float ref, uint32_t coalesce_mask = ~0u) {
#pragma clang fp contract(fast)
    const float tu = s * (float)res - 0.5f, tv = t * (float)res - 0.5f;
    const float fu = floorf(tu), fv = floorf(tv);
    const float a = tu - fu, b = tv - fv;
    const int i0 = (int)clampf(fu, -2.0f, (float)res), j0 = (int)clampf(fv, -2.0f, (float)res);
    const int pitch = res + 4;
    const RDR_GLOBAL float* row = map + (coalesce_mask & (uint32_t)(mul24(j0 + 2, pitch) + i0 + 2));
    float d00, d10, d01, d11;
    load_pair_f(row, d00, d10);
    load_pair_f(row + pitch, d01, d11);
    const float r00 = ref <= d00 ? 1.0f : 0.0f, r10 = ref <= d10 ? 1.0f : 0.0f, r01 = ref <= d01 ? 1.0f : 0.0f, r11 = ref <= d11 ? 1.0f : 0.0f;
    const float top = (1.0f - a) * r00 + a * r10, bot = (1.0f - a) * r01 + a * r11;
    return (1.0f - b) * top + b * bot;
}
// samplerCubeArrayShadow. Every face is a (res + 2)^2 image whose border holds the depths across the cube edges
// (seamless cube filtering), so the footprint is four neighbouring words wherever it lies.
// A lane with `need` false gets a result it will drop: it reads the first words of the cube instead of its own
// footprint, so that all such lanes of a wave share one cache line.
RDR_DEV float shadow_cube(const DFrame& fr, uint32_t slot, float rx, float ry, float rz, float ref, bool need) {
#pragma clang fp contract(fast)
    const int res = (int)fr.point_res, pitch = res + 2;
    float s, t;
    const int face = cube_face_fast(rx, ry, rz, s, t);  // the PCF result is continuous in (s,t)
    const RDR_GLOBAL float* map = as_global(fr.point_shadow) + (size_t)slot * 6 * pitch * pitch;  // the six faces of this cube
    const float tu = s * (float)res - 0.5f, tv = t * (float)res - 0.5f;
    const float fu = floorf(tu), fv = floorf(tv);
    const float a = tu - fu, b = tv - fv;
    const int i0 = min(max((int)fu, -1), res - 1), j0 = min(max((int)fv, -1), res - 1);
    const RDR_GLOBAL float* row = map + (((RDR_DIAG_COALESCE & 2) ? RDR_DIAG_ZERO(fr) : ~0u) & (uint32_t)(need ? mul24(mul24(face, pitch) + j0 + 1, pitch) + i0 + 1 : 0));
    float d00, d10, d01, d11;
    load_pair_f(row, d00, d10);
    load_pair_f(row + (need ? pitch : 0), d01, d11);
    const float r00 = ref <= d00 ? 1.0f : 0.0f, r10 = ref <= d10 ? 1.0f : 0.0f;
    const float r01 = ref <= d01 ? 1.0f : 0.0f, r11 = ref <= d11 ? 1.0f : 0.0f;
    const float top = (1.0f - a) * r00 + a * r10, bot = (1.0f - a) * r01 + a * r11;
    return (1.0f - b) * top + b * bot;
}

#define kPI 3.14159274f

// What pbr.frag's BRDF (pbr.frag:211-269) needs of the fragment, worked out once: the terms that do not depend on
// the light. Continuous in all arguments: fast math allowed.
struct BrdfFrag {
    F3 N, V, kdiff, F0, omF0;  // kdiff = (1 - metallic) * albedo / pi, omF0 = 1 - F0
    float NV, NV1, a2, a2m1, k, omk, gv, c4;
};
RDR_DEV BrdfFrag brdf_frag(F3 N, F3 V, F3 albedo, float roughness, float metallic, F3 F0) {
#pragma clang fp contract(fast)
    BrdfFrag f;
    f.N = N; f.V = V; f.F0 = F0;
    f.omF0 = f3(1.0f - F0.x, 1.0f - F0.y, 1.0f - F0.z);
    const float kd = (1.0f - metallic) * (1.0f / kPI);
    f.kdiff = albedo * kd;
    f.NV = dot3f(N, V);
    f.NV1 = fmaxf(f.NV, 1e-3f);
    const float a = roughness * roughness;
    f.a2 = a * a; f.a2m1 = f.a2 - 1.0f;
    const float r1 = roughness + 1.0f;
    f.k = (r1 * r1) * 0.125f; f.omk = 1.0f - f.k;
    f.gv = f.NV1 * fast_rcp(f.NV1 * f.omk + f.k);  // GeometrySchlickGGX(NdotV)
    f.c4 = 4.0f * f.NV1;
    return f;
}
typedef float v2f __attribute__((ext_vector_type(2)));
RDR_DEV v2f v2(float a, float b) { v2f r; r.x = a; r.y = b; return r; }
RDR_DEV v2f v2s(float a) { return v2(a, a); }
// brdf_eval below for two lights at once, one per vector component: the sums, products and fused multiply-adds are
// two-component instructions, only the reciprocals, square roots and min / max work on one component at a time.
RDR_DEV void brdf_eval2(const BrdfFrag& f, v2f t, v2f omLL, v2f NL, v2f& r, v2f& g, v2f& b) {
#pragma clang fp contract(fast)
    const v2f rs = v2(__builtin_amdgcn_rsqf(t.x), __builtin_amdgcn_rsqf(t.y));
    v2f NdotH = (f.NV + NL) * rs, HdotV = (0.5f * (t + omLL)) * rs;
    NdotH = v2(clampf(NdotH.x, 0.0f, 1.0f), clampf(NdotH.y, 0.0f, 1.0f));
    HdotV = v2(clampf(HdotV.x, 0.0f, 1.0f), clampf(HdotV.y, 0.0f, 1.0f));
    v2f dd = NdotH * NdotH * f.a2m1 + 1.0f;
    dd = kPI * dd * dd;
    const v2f NL1 = v2(fmaxf(NL.x, 1e-3f), fmaxf(NL.y, 1e-3f));
    const v2f gden = NL1 * f.omk + f.k;
    v2f sden = f.c4 * NL1;
    sden = v2(fmaxf(sden.x, 1e-5f), fmaxf(sden.y, 1e-5f));
#if RDR_BRDF_ONE_RCP
    // D * G / (4 N.V N.L): one reciprocal of the product of the three denominators (each at least 1e-11: no underflow)
    const v2f den = dd * gden * sden;
    const v2f spec = (f.a2 * f.gv) * NL1 * v2(fast_rcp(den.x), fast_rcp(den.y));
#else
    const v2f G = NL1 * v2(fast_rcp(gden.x), fast_rcp(gden.y)) * f.gv;
    const v2f spec = f.a2 * v2(fast_rcp(dd.x), fast_rcp(dd.y)) * G * v2(fast_rcp(sden.x), fast_rcp(sden.y));
#endif
    const v2f x = 1.0f - HdotV;
    const v2f x2 = x * x, x5 = x2 * x2 * x;
    const v2f NdotL = v2(fmaxf(NL.x, 0.0f), fmaxf(NL.y, 0.0f));
    const v2f Fx = f.F0.x + f.omF0.x * x5, Fy = f.F0.y + f.omF0.y * x5, Fz = f.F0.z + f.omF0.z * x5;
    r = ((f.kdiff.x - Fx * f.kdiff.x) + spec * Fx) * NdotL;
    g = ((f.kdiff.y - Fy * f.kdiff.y) + spec * Fy) * NdotL;
    b = ((f.kdiff.z - Fz * f.kdiff.z) + spec * Fz) * NdotL;
}
// BRDF(1, L, ...) of pbr.frag: the outgoing radiance per unit of light radiance, cosine term included. L need not be
// normalised (the reference passes the raw light vector for lights without a shadow map, pbr.frag:157-158):
// t = |V + L|^2, omLL = 1 - L.L, NL = N.L. The half vector itself never materialises: with |V| = 1,
// N.H = (N.V + N.L) / |V + L| and H.V = (1 + V.L) / |V + L| with 1 + V.L = (t + 1 - L.L) / 2. Its length does, from the
// components of V + L: where the
// light is nearly opposite the view direction (the grazing highlight of a surface seen against the light) 1 + 2 V.L + L.L
// cancels to a few float steps while the sum of the vectors keeps its digits, and N.H near 1 needs them.
RDR_DEV F3 brdf_eval(const BrdfFrag& f, float t, float omLL, float NL) {
#pragma clang fp contract(fast)
    const float rs = __builtin_amdgcn_rsqf(t);
    const float NdotH = clampf((f.NV + NL) * rs, 0.0f, 1.0f);
    const float HdotV = clampf((0.5f * (t + omLL)) * rs, 0.0f, 1.0f);
    float dd = NdotH * NdotH * f.a2m1 + 1.0f;
    dd = kPI * dd * dd;
    const float NL1 = fmaxf(NL, 1e-3f);
    const float G = NL1 * fast_rcp(NL1 * f.omk + f.k) * f.gv;
    const float spec = f.a2 * fast_rcp(dd) * G * fast_rcp(fmaxf(f.c4 * NL1, 1e-5f));
    const float x = 1.0f - HdotV;
    const float x2 = x * x, x5 = x2 * x2 * x;
    const float NdotL = fmaxf(NL, 0.0f);
    const float Fx = f.F0.x + f.omF0.x * x5, Fy = f.F0.y + f.omF0.y * x5, Fz = f.F0.z + f.omF0.z * x5;
    F3 o;
    o.x = ((f.kdiff.x - Fx * f.kdiff.x) + spec * Fx) * NdotL;
    o.y = ((f.kdiff.y - Fy * f.kdiff.y) + spec * Fy) * NdotL;
    o.z = ((f.kdiff.z - Fz * f.kdiff.z) + spec * Fz) * NdotL;
    return o;
}

// env.vert:22-26 + env.frag:10-17
RDR_DEV void shade_sky(const DFrame& fr, int px, int py, float out[4]) {
    const float xn = fmaf((float)px + 0.5f, fr.inv2W, -1.0f), yn = fmaf((float)py + 0.5f, fr.inv2H, -1.0f);
    const float* m = fr.inv_viewproj;
    F3 pos;
    pos.x = fmaf(m[0], xn, fmaf(m[4], yn, m[8] + m[12]));
    pos.y = fmaf(m[1], xn, fmaf(m[5], yn, m[9] + m[13]));
    pos.z = fmaf(m[2], xn, fmaf(m[6], yn, m[10] + m[14]));
    const F3 d = normalize3(pos);
    float e[3] = {0.0f, 0.0f, 0.0f};
    if (fr.env.res) cube_sample_fast(fr.env, 0, d.x, d.y, d.z, e);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        // v in [0, 1): the hardware log2 / exp2 are as good as the library's pow here (the result is stored as half)
        const float v = e[k] / (e[k] + 1.0f);
        out[k] = v > 0.0f ? __builtin_amdgcn_exp2f((1.0f / 2.2f) * __builtin_amdgcn_logf(v)) : 0.0f;
    }
    out[3] = 1.0f;
}

// basic.vert for the three vertices of one triangle (basic.vert:36-49) and the
// homogeneous barycentric planes of its clip-space positions.
RDR_DEV void shade_tri_setup(const DFrame& fr, const DDraw& d, uint32_t tri, DShadeTri& o) {
    const DPrim& pr = fr.prims[d.prim];
    const uint32_t* ip = fr.idx + pr.first_index + 3u * tri;
    const uint32_t vi[3] = {pr.first_vertex + ip[0], pr.first_vertex + ip[1], pr.first_vertex + ip[2]};
    float4 op[3];
    F4 cp[3];
#pragma unroll
    for (int k = 0; k < 3; k++) { op[k] = fr.pos[vi[k]]; cp[k] = xform_point(d.mvp, op[k].x, op[k].y, op[k].z); }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const F4 p = cp[(k + 1) % 3], q = cp[(k + 2) % 3];
        o.A[k] = (double)p.y * (double)q.w - (double)q.y * (double)p.w;
        o.B[k] = (double)q.x * (double)p.w - (double)p.x * (double)q.w;
        o.C[k] = (double)p.x * (double)q.y - (double)q.x * (double)p.y;
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const F4 w4 = xform_point(d.model, op[k].x, op[k].y, op[k].z);
        const F3 wp = w4.w == 1.0f ? f3(w4.x, w4.y, w4.z) : f3(w4.x / w4.w, w4.y / w4.w, w4.z / w4.w);
        const DVertexAttr a = fr.attr[vi[k]];
        const F3 n = fast_normalize(xform_dir(d.model, a.nx, a.ny, a.nz));
        const F3 t = fast_normalize(xform_dir(d.model, a.tx, a.ty, a.tz));
        const F3 b = fast_normalize(xform_dir(d.model, a.bx, a.by, a.bz));
        o.wp[k] = wp.x; o.wp[3 + k] = wp.y; o.wp[6 + k] = wp.z;
        o.nrm[k] = n.x; o.nrm[3 + k] = n.y; o.nrm[6 + k] = n.z;
        o.tan[k] = t.x; o.tan[3 + k] = t.y; o.tan[6 + k] = t.z;
        o.bit[k] = b.x; o.bit[3 + k] = b.y; o.bit[6 + k] = b.z;
        o.uv[k] = a.u; o.uv[3 + k] = a.v;
    }
    o.prim = d.prim;
    const DMaterial& mt = pr.mat;
#pragma unroll
    for (int k = 0; k < 4; k++) o.base_color[k] = mt.base_color[k];
#pragma unroll
    for (int k = 0; k < 3; k++) o.emissive[k] = mt.emissive[k];
    o.alpha_cutoff = mt.alpha_cutoff;
    o.metallic = mt.metallic; o.roughness = mt.roughness; o.ao = mt.ao;
    o.alpha_mode = mt.alpha_mode;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        // defaults of the reference: 1x1 white, and 1x1 (128,128,255,0) for the normal map
        const uint32_t ti = mt.tex[k] != 0xFFFFFFFFu ? mt.tex[k] : (k == 1 ? fr.tex_normal : fr.tex_white);
        const DTexture& t = fr.textures[ti];
        const bool pow2sq = t.w == t.h && (t.w & (t.w - 1u)) == 0u && (t.levels == 1u || (1u << (t.levels - 1u)) == t.w);
        const uint32_t flags = (t.flags & 1u) | (pow2sq ? RDR_TEXD_POW2_SQUARE : 0u) | (((t.flags >> 4) & 3u) << 2) | (((t.flags >> 8) & 3u) << 4) |
                               (((t.flags >> 12) & 7u) << RDR_TEXD_NEAREST_SHIFT) | (((t.flags >> 16) & 1u) ? RDR_TEXD_FLOAT : 0u);
        o.tex[k].base = t.base;
        o.tex[k].wh = t.w | (t.h << 16);
        o.tex[k].info = t.levels | (flags << 5) | (ti << 16);
    }
}

// pbr.frag at the centre of pixel (px,py) of the triangle whose vertex-stage outputs are in T.
// MASKED = false for the opaque list, which holds no alpha-masked primitive: the cut-off test and its early
// exit disappear and the five texture fetches can be in flight together.
// `M`: a triangle record of the same primitive as T, where the material block and the texture descriptors are read
// (they are the same in every record of a primitive: a wave whose lanes agree on M reads them from one cache line).
// `park` (PARK): this thread's column of a [12][256] float array in LDS. The world position and the three
// interpolated frame vectors are not needed while the textures are fetched and filtered - the part of the function
// that takes every register there is - and wait there meanwhile (otherwise the compiler spills them, and more, to
// scratch memory: stores that go out to HBM).
template <bool MASKED, bool PARK, bool FLOAT_TEXELS = false>
RDR_DEV void shade_fragment(const DFrame& fr, const DPointLight* __restrict__ lights, const float* srgb, const DShadeTri& T, const DShadeTri& M,
                            int px, int py, float out[4], float* park) {
    // Multiply-add pairs may fuse in this function: what must round as in the CPU checker is written with explicit
    // fma / fmaf calls or sits in the cascade block below, which switches fusing off again.
#pragma clang fp contract(fast)
    // ---- perspective-correct barycentrics at (px,py) in double (the world position they give feeds shadow compares);
    // at (px+1,py) and (px,py+1) in single precision from the same planes: those two only give the texture-coordinate
    // differences the level of detail is chosen from, and filtering is continuous in it
    const float x0 = fmaf((float)px + 0.5f, fr.inv2W, -1.0f), y0 = fmaf((float)py + 0.5f, fr.inv2H, -1.0f);
    float lam[3][3];
    {
        double l[3];
#pragma unroll
        for (int k = 0; k < 3; k++) l[k] = fma(T.A[k], (double)x0, fma(T.B[k], (double)y0, T.C[k]));
        const double inv = 1.0 / ((l[0] + l[1]) + l[2]);
#pragma unroll
        for (int k = 0; k < 3; k++) lam[0][k] = (float)(l[k] * inv);
        float lx[3], ly[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float lf = (float)l[k];
            lx[k] = fmaf((float)T.A[k], fr.inv2W, lf);
            ly[k] = fmaf((float)T.B[k], fr.inv2H, lf);
        }
        const float ix = fast_rcp((lx[0] + lx[1]) + lx[2]), iy = fast_rcp((ly[0] + ly[1]) + ly[2]);
#pragma unroll
        for (int k = 0; k < 3; k++) { lam[1][k] = lx[k] * ix; lam[2][k] = ly[k] * iy; }
    }
    F3 P = f3(interp3(lam[0], T.wp[0], T.wp[1], T.wp[2]), interp3(lam[0], T.wp[3], T.wp[4], T.wp[5]), interp3(lam[0], T.wp[6], T.wp[7], T.wp[8]));
    F3 Nv = f3(interp3(lam[0], T.nrm[0], T.nrm[1], T.nrm[2]), interp3(lam[0], T.nrm[3], T.nrm[4], T.nrm[5]), interp3(lam[0], T.nrm[6], T.nrm[7], T.nrm[8]));
    F3 Tv = f3(interp3(lam[0], T.tan[0], T.tan[1], T.tan[2]), interp3(lam[0], T.tan[3], T.tan[4], T.tan[5]), interp3(lam[0], T.tan[6], T.tan[7], T.tan[8]));
    F3 Bv = f3(interp3(lam[0], T.bit[0], T.bit[1], T.bit[2]), interp3(lam[0], T.bit[3], T.bit[4], T.bit[5]), interp3(lam[0], T.bit[6], T.bit[7], T.bit[8]));
    const float u0 = interp3(lam[0], T.uv[0], T.uv[1], T.uv[2]), v0 = interp3(lam[0], T.uv[3], T.uv[4], T.uv[5]);
    const float dudx = interp3(lam[1], T.uv[0], T.uv[1], T.uv[2]) - u0, dvdx = interp3(lam[1], T.uv[3], T.uv[4], T.uv[5]) - v0;
    const float dudy = interp3(lam[2], T.uv[0], T.uv[1], T.uv[2]) - u0, dvdy = interp3(lam[2], T.uv[3], T.uv[4], T.uv[5]) - v0;
    if (PARK) {
        park[0 * 256] = P.x; park[1 * 256] = P.y; park[2 * 256] = P.z;
        park[3 * 256] = Nv.x; park[4 * 256] = Nv.y; park[5 * 256] = Nv.z;
        park[6 * 256] = Tv.x; park[7 * 256] = Tv.y; park[8 * 256] = Tv.z;
        park[9 * 256] = Bv.x; park[10 * 256] = Bv.y; park[11 * 256] = Bv.z;
        __asm__ volatile("" ::: "memory");  // the values are read back below, not kept
    }
    __builtin_amdgcn_sched_barrier(0);  // the loads below stay below: the record is consumed in stages, not held whole
    const DShadeTri& mt = M;  // material block and texture descriptors travel with the triangle

#ifndef RDR_SHADOW_ALWAYS
#define RDR_SHADOW_ALWAYS 0  // measurement aid: point-light shadow look-ups even where no fragment of the wave needs them
#endif
#ifndef RDR_DIAG
#define RDR_DIAG 0  // diagnostic builds only: bit 0 no textures, bit 1 no directional light, bit 2 no IBL, bit 3 nothing after interpolation, bit 4 no point lights, bit 5 no point-light shadow lookups
#endif
    if (RDR_DIAG & 8) { out[0] = P.x + Nv.x + Tv.y + Bv.z; out[1] = u0 + dudx; out[2] = v0 + dvdy + dudy + dvdx; out[3] = 1.0f; return; }
    // ---- pbr.frag main()
    // The five material textures. Their filter taps come from the albedo texture's geometry. In the common case,
    // decided per wave, all five have that geometry and no footprint wraps around an image edge: the texels of all
    // of them are then loaded in one go (one memory round trip instead of one per texture) and filtered afterwards.
    // Otherwise a texture of another size or wrap mode gets its own taps and the textures are fetched one by one.
    float t[4] = {0.5f, 0.5f, 0.5f, 1.0f};
    float albedo[4] = {0.5f * mt.base_color[0], 0.5f * mt.base_color[1], 0.5f * mt.base_color[2], mt.base_color[3]};
    float metallic = clampf(0.5f * mt.metallic, 0.0f, 1.0f), roughness = clampf(0.5f * mt.roughness, 0.05f, 1.0f), ao = 0.5f * mt.ao;
    float emissive[3] = {0.0f, 0.0f, 0.0f}, nmap[3] = {0.5f, 0.5f, 1.0f};
    // a zero emissive factor (the glTF default) makes the product zero whatever the texel is: no fetch
    const bool has_emissive = mt.emissive[0] != 0.0f || mt.emissive[1] != 0.0f || mt.emissive[2] != 0.0f;
    if (!(RDR_DIAG & 1)) {
        // all five descriptors in one burst of loads, before anything branches on them (the geometry test below uses
        // every one of them, without short-circuit evaluation): one memory round trip instead of one per texture
        DTexDesc td[5];
#pragma unroll
        for (int k = 0; k < 5; k++) td[k] = M.tex[k];
        TexTaps taps0;
        tex_taps(td[0], fr.textures, u0, v0, dudx, dvdx, dudy, dvdy, taps0);
        // (float texels are part of the geometry: a material with an SFLOAT image takes the one-by-one path)
        const bool uniform_taps = (int)tex_pairs_adjacent(taps0) & (int)tex_same_geometry(td[1], td[0]) & (int)tex_same_geometry(td[2], td[0]) &
                                  (int)tex_same_geometry(td[3], td[0]) & (int)tex_same_geometry(td[4], td[0]) &
                                  (int)(!FLOAT_TEXELS || ((td[0].info >> 5) & RDR_TEXD_FLOAT) == 0u);
        if (__all(uniform_taps)) {
            // RDR_TEX_SERIAL: one texture at a time (five memory round trips, few registers: the kernel then fits more
            // waves per SIMD, which hide the round trips); otherwise all loads in flight together (one round trip, forty
            // registers of texels). The empty asm statements keep the compiler from moving the loads back together.
#ifndef RDR_TEX_SERIAL
#define RDR_TEX_SERIAL 0
#endif
#define RDR_TEX_FENCE() do { if (RDR_TEX_SERIAL) asm volatile("" ::: "memory"); } while (0)
            TexRaw r0, r1, r2, r3, r4;
            const bool any_emissive = __any(has_emissive) != 0;
            if (RDR_DIAG_COALESCE & 1) {
                const uint32_t z = RDR_DIAG_ZERO(fr);
                for (int l = 0; l < 2; l++) { taps0.lv[l].o00 &= z; taps0.lv[l].o01 &= z; }
            }
            tex_issue(td[1], taps0, r1);
            if (!RDR_TEX_SERIAL) { tex_issue(td[0], taps0, r0); tex_issue(td[2], taps0, r2); }
            tex_resolve<7>(srgb, td[1], taps0, r1, t);
            nmap[0] = t[0]; nmap[1] = t[1]; nmap[2] = t[2];
            RDR_TEX_FENCE();
            // glTF files usually pack occlusion, roughness and metalness into one image (main.cpp:135-166 reads the two
            // slots independently): when every fragment of the wave has the same image in both slots, it is fetched once
            const bool shared_orm = __all(td[3].base == td[2].base) != 0;
            if (RDR_TEX_SERIAL) tex_issue(td[0], taps0, r0);
            else { if (!shared_orm) tex_issue(td[3], taps0, r3); if (any_emissive) tex_issue(td[4], taps0, r4); }
            tex_resolve<15>(srgb, td[0], taps0, r0, t);
#pragma unroll
            for (int k = 0; k < 4; k++) albedo[k] = t[k] * mt.base_color[k];
            if (MASKED && mt.alpha_mode == 1 && albedo[3] < mt.alpha_cutoff) { out[0] = out[1] = out[2] = out[3] = 0.0f; return; }
            RDR_TEX_FENCE();
            if (RDR_TEX_SERIAL) tex_issue(td[2], taps0, r2);
            tex_resolve<6>(srgb, td[2], taps0, r2, t);
            metallic = clampf(t[2] * mt.metallic, 0.0f, 1.0f);
            roughness = clampf(t[1] * mt.roughness, 0.05f, 1.0f);
            RDR_TEX_FENCE();
            if (RDR_TEX_SERIAL && !shared_orm) tex_issue(td[3], taps0, r3);
            if (shared_orm) tex_resolve<1>(srgb, td[3], taps0, r2, t);  // same geometry (this branch), same texels
            else tex_resolve<1>(srgb, td[3], taps0, r3, t);
            ao = t[0] * mt.ao;
            if (any_emissive) {
                RDR_TEX_FENCE();
                if (RDR_TEX_SERIAL) tex_issue(td[4], taps0, r4);
                tex_resolve<7>(srgb, td[4], taps0, r4, t);
                emissive[0] = t[0] * mt.emissive[0]; emissive[1] = t[1] * mt.emissive[1]; emissive[2] = t[2] * mt.emissive[2];
            }
            RDR_TEX_FENCE();
        } else {
            TexTaps taps;
            tex_fetch<15, FLOAT_TEXELS>(srgb, td[0], taps0, t);
#pragma unroll
            for (int k = 0; k < 4; k++) albedo[k] = t[k] * mt.base_color[k];
            if (MASKED && mt.alpha_mode == 1 && albedo[3] < mt.alpha_cutoff) { out[0] = out[1] = out[2] = out[3] = 0.0f; return; }
#define RDR_SAMPLE_SLOT(slot, mask)                                                                              \
            taps = taps0;                                                                                        \
            if (!__all(tex_same_geometry(td[slot], td[0]))) tex_taps(td[slot], fr.textures, u0, v0, dudx, dvdx, dudy, dvdy, taps); \
            tex_fetch<mask, FLOAT_TEXELS>(srgb, td[slot], taps, t);
            RDR_SAMPLE_SLOT(2, 6)
            metallic = clampf(t[2] * mt.metallic, 0.0f, 1.0f);
            roughness = clampf(t[1] * mt.roughness, 0.05f, 1.0f);
            RDR_SAMPLE_SLOT(3, 1)
            ao = t[0] * mt.ao;
            if (has_emissive) {
                RDR_SAMPLE_SLOT(4, 7)
                emissive[0] = t[0] * mt.emissive[0]; emissive[1] = t[1] * mt.emissive[1]; emissive[2] = t[2] * mt.emissive[2];
            }
            RDR_SAMPLE_SLOT(1, 7)
            nmap[0] = t[0]; nmap[1] = t[1]; nmap[2] = t[2];
#undef RDR_SAMPLE_SLOT
        }
    } else if (MASKED && mt.alpha_mode == 1 && albedo[3] < mt.alpha_cutoff) { out[0] = out[1] = out[2] = out[3] = 0.0f; return; }
    if (PARK) {
        __asm__ volatile("" ::: "memory");
        P = f3(park[0 * 256], park[1 * 256], park[2 * 256]);
        Nv = f3(park[3 * 256], park[4 * 256], park[5 * 256]);
        Tv = f3(park[6 * 256], park[7 * 256], park[8 * 256]);
        Bv = f3(park[9 * 256], park[10 * 256], park[11 * 256]);
    }
    const F3 Tn = fast_normalize(Tv), Bn = fast_normalize(Bv), Nn = fast_normalize(Nv);
    const float nx = nmap[0] * 2.0f - 1.0f, ny = nmap[1] * 2.0f - 1.0f, nz = nmap[2] * 2.0f - 1.0f;
    const F3 N = fast_normalize(Tn * nx + Bn * ny + Nn * nz);
    const F3 cam = f3(fr.cam[0], fr.cam[1], fr.cam[2]);
    const F3 V = fast_normalize(cam - P);
    const F3 negV = V * -1.0f;
    const F3 Rv = negV - N * (2.0f * dot3f(N, negV));
    const F3 alb = f3(albedo[0], albedo[1], albedo[2]);
    const float om = 1.0f - metallic;
    const F3 F0 = f3(0.04f * om + alb.x * metallic, 0.04f * om + alb.y * metallic, 0.04f * om + alb.z * metallic);
    // the sum pbr.frag forms as Lo + emissive, in another order: emissive first, then the image-based term (its
    // inputs - reflection vector, occlusion, table look-up - are then dead before the light loop), then the lights
    F3 Lo = f3(emissive[0], emissive[1], emissive[2]);

    const BrdfFrag bf = brdf_frag(N, V, alb, roughness, metallic, F0);

    // ---- image-based lighting (pbr.frag:186-200)
    auto image_based = [&]() {
    if (fr.prefiltered.res && fr.irradiance.res && fr.lut_w && !(RDR_DIAG & 4)) {
        const float NdotV = bf.NV1;
        const float lod = fminf(roughness * (float)fr.prefiltered.levels, (float)(fr.prefiltered.levels - 1u));
        const float fl = floorf(lod), ff = lod - fl;
        const uint32_t l0 = (uint32_t)fl, l1 = min(l0 + 1u, fr.prefiltered.levels - 1u);
        float s0[3], s1[3], irr[3], lut[2];
        const float dz = (RDR_DIAG_COALESCE & 4) ? (float)RDR_DIAG_ZERO(fr) : 1.0f;  // 0: every lane looks along +x
        cube_sample_quad(fr.prefiltered, l0, (RDR_DIAG_COALESCE & 4) ? 1.0f : Rv.x, Rv.y * dz, Rv.z * dz, s0);
        cube_sample_quad(fr.prefiltered, l1, (RDR_DIAG_COALESCE & 4) ? 1.0f : Rv.x, Rv.y * dz, Rv.z * dz, s1);
        {
            const int w = (int)fr.lut_w, h = (int)fr.lut_h;
            int i0, i1, j0, j1;
            float a, b;
            tex_axis(NdotV, w, 0u, i0, i1, a);
            tex_axis(roughness, h, 0u, j0, j1, b);
            // footprint-major table: the four texels of the (wrapped) footprint at (i0, j0) are one 16-byte entry
            const rdr_uint4_t q = ((const RDR_GLOBAL rdr_uint4_t*)as_global((const uint32_t*)fr.lut))[((RDR_DIAG_COALESCE & 4) ? RDR_DIAG_ZERO(fr) : ~0u) & (uint32_t)(mul24(j0, w) + i0)];
            const uint32_t q00 = q.x, q10 = q.y, q01 = q.z, q11 = q.w;
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const float t00 = (float)((q00 >> (8 * k)) & 255u) * RDR_INV255, t10 = (float)((q10 >> (8 * k)) & 255u) * RDR_INV255;
                const float t01 = (float)((q01 >> (8 * k)) & 255u) * RDR_INV255, t11 = (float)((q11 >> (8 * k)) & 255u) * RDR_INV255;
                const float top = (1.0f - a) * t00 + a * t10, bot = (1.0f - a) * t01 + a * t11;
                lut[k] = (1.0f - b) * top + b * bot;
            }
        }
        cube_sample_quad(fr.irradiance, 0, (RDR_DIAG_COALESCE & 4) ? 1.0f : N.x, N.y * dz, N.z * dz, irr);
        const float x = 1.0f - clampf(NdotV, 0.0f, 1.0f);
        const float x2 = x * x, x5 = x2 * x2 * x;
        const float F[3] = {F0.x + (1.0f - F0.x) * x5, F0.y + (1.0f - F0.y) * x5, F0.z + (1.0f - F0.z) * x5};
        float add[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float spec = (1.0f - ff) * s0[k] + ff * s1[k];
            const float kD = (1.0f - F[k]) * om;
            add[k] = (kD * irr[k] * albedo[k] + spec * (F[k] * lut[0] + lut[1])) * ao;
        }
        Lo = Lo + f3(add[0], add[1], add[2]);
    }
    };
#ifndef RDR_IBL_FIRST
#define RDR_IBL_FIRST 1
#endif
    if (RDR_IBL_FIRST) image_based();

    // ---- point lights (pbr.frag:123-160)
    // depth of the cube-face projection (near .1, far 100): (far - far*near/d) / (far - near) = c0 - c1 / d
    const float c0 = 100.0f / (100.0f - 0.1f), c1 = (100.0f * 0.1f) / (100.0f - 0.1f);
    // Two lights at a time (a DPointLight record), in two phases: first the light vectors and the shadow-map look-ups of
    // the pair (their loads overlap), then both BRDFs in two-component arithmetic. The records are read through a kernel
    // argument with a uniform index: scalar loads into scalar registers. Whether a light casts shadows is uniform, so
    // that branch costs nothing.
    const uint32_t n_pairs = (RDR_DIAG & 16) ? 0u : (fr.n_lights + 1u) >> 1;
    v2f LoR = v2s(0.0f), LoG = v2s(0.0f), LoB = v2s(0.0f);  // what the first / second lights of the pairs add
    for (uint32_t p = 0; p < n_pairs; p++) {
        const DPointLight pl = lights[p];
        const v2f Lx = v2(pl.px[0], pl.px[1]) - P.x, Ly = v2(pl.py[0], pl.py[1]) - P.y, Lz = v2(pl.pz[0], pl.pz[1]) - P.z;
        const v2f d2 = __builtin_elementwise_fma(Lz, Lz, __builtin_elementwise_fma(Ly, Ly, Lx * Lx));  // distance^2
        const v2f nl = __builtin_elementwise_fma(v2s(N.z), Lz, __builtin_elementwise_fma(v2s(N.y), Ly, N.x * Lx));
        // radiance = intensity / distance^2; skipped when its largest channel is <= 1e-3 (pbr.frag:133)
        const bool lit[2] = {!(fmaxf(pl.ix[0], fmaxf(pl.iy[0], pl.iz[0])) <= 1e-3f * d2.x), !(fmaxf(pl.ix[1], fmaxf(pl.iy[1], pl.iz[1])) <= 1e-3f * d2.y)};
        float sh[2];
#pragma unroll
        for (int j = 0; j < 2; j++) {
            // A light behind the shaded surface contributes radiance * max(N.L, 0) = 0 whatever its shadow test says, and
            // one too far away is skipped. The look-up is left out only when no fragment of the wave needs it (a uniform
            // branch; neighbouring fragments mostly face the same way): within a wave there is no per-lane branch, so
            // that the look-ups of the pair sit in one block of code and their loads are in flight together; a lane
            // that does not need the result computes it on a valid address and drops it.
            const bool need = nl[j] > 0.0f && lit[j];
            float shadowTest = 1.0f;
            if ((pl.flags[j] & 3u) == 3u && !(RDR_DIAG & 32) && (RDR_SHADOW_ALWAYS || __any(need))) {  // uniform: casts shadows, has a map, is needed
                const float lx = Lx[j], ly = Ly[j], lz = Lz[j];
                const float lightDepth = fmaxf(fabsf(lx), fmaxf(fabsf(ly), fabsf(lz))) - 0.05f;
                const float ref = c0 - c1 / lightDepth;  // IEEE divide: this value is compared with stored depths
                shadowTest = shadow_cube(fr, pl.slot[j], -lx, -ly, -lz, ref, need);
                shadowTest = shadowTest < 1e-3f ? 0.0f : shadowTest;  // pbr.frag:151
            }
            sh[j] = need ? shadowTest : 0.0f;
        }
        const bool on0 = sh[0] != 0.0f, on1 = sh[1] != 0.0f;
        if (!__any(on0 || on1)) continue;  // uniform
        // a light with a shadow map uses the normalised direction, one without does not (pbr.frag:154-158)
        const bool unit0 = (pl.flags[0] & 3u) == 3u, unit1 = (pl.flags[1] & 3u) == 3u;
        const v2f ls = v2(unit0 ? __builtin_amdgcn_rsqf(d2.x) : 1.0f, unit1 ? __builtin_amdgcn_rsqf(d2.y) : 1.0f);
        v2f r, g, b;
        // |V + L|^2 from the components of the sum (see brdf_eval), one component at a time
        v2f hh, hc = __builtin_elementwise_fma(Lx, ls, v2s(V.x));
        hh = hc * hc;
        hc = __builtin_elementwise_fma(Ly, ls, v2s(V.y));
        hh = __builtin_elementwise_fma(hc, hc, hh);
        hc = __builtin_elementwise_fma(Lz, ls, v2s(V.z));
        hh = __builtin_elementwise_fma(hc, hc, hh);
        brdf_eval2(bf, hh, v2(unit0 ? 0.0f : 1.0f - d2.x, unit1 ? 0.0f : 1.0f - d2.y), nl * ls, r, g, b);
#if RDR_BRDF_ONE_RCP
        const v2f att = v2(unit0 ? ls.x * ls.x : fast_rcp(d2.x), unit1 ? ls.y * ls.y : fast_rcp(d2.y));  // 1 / distance^2
        const v2f w = v2(on0 ? att.x * sh[0] : 0.0f, on1 ? att.y * sh[1] : 0.0f);  // attenuation x shadow
#else
        const v2f w = v2(on0 ? fast_rcp(d2.x) * sh[0] : 0.0f, on1 ? fast_rcp(d2.y) * sh[1] : 0.0f);  // attenuation x shadow
#endif
        LoR += r * (v2(pl.ix[0], pl.ix[1]) * w);
        LoG += g * (v2(pl.iy[0], pl.iy[1]) * w);
        LoB += b * (v2(pl.iz[0], pl.iz[1]) * w);
    }
    Lo = Lo + f3(LoR.x + LoR.y, LoG.x + LoG.y, LoB.x + LoB.y);

    // ---- directional light with cascaded shadow maps (pbr.frag:163-183, basic.vert:45-48)
    // (a surface facing away from the light gets max(N.L, 0) = 0 from it: no cascade look-up, no BRDF)
    const F3 dirL = f3(fr.dir_L[0], fr.dir_L[1], fr.dir_L[2]);
    const float dir_nl = dot3f(N, dirL);
    if (fr.lights_set && !(RDR_DIAG & 2) && dir_nl > 0.0f) {
#pragma clang fp contract(off)  // the light-space coordinates choose the cascade and are compared with stored depths
        const int ncasc = (int)fr.n_cascades, res = (int)fr.dir_res;
        int split = 0;
        float ls0 = 0.0f, ls1 = 0.0f, ls2 = 0.0f;
        for (; split < ncasc; split++) {
            const F4 lsp = xform_point(fr.csm_viewproj[split], P.x, P.y, P.z);
            ls0 = lsp.x * 0.5f + 0.5f; ls1 = lsp.y * 0.5f + 0.5f; ls2 = lsp.z;
            if (lsp.w != 1.0f) { ls0 /= lsp.w; ls1 /= lsp.w; ls2 /= lsp.w; }
            if (fminf(ls0, ls1) > 1e-2f && fmaxf(ls0, ls1) < 1.0f - 1e-2f) break;
        }
        const int layer = min(split, ncasc - 1);
        const float shadowTest = shadow_pcf(as_global(fr.dir_shadow) + (size_t)layer * (res + 4) * (res + 4), res, ls0, ls1, ls2,
                                                (RDR_DIAG_COALESCE & 8) ? RDR_DIAG_ZERO(fr) : ~0u);
        if (shadowTest > 1e-3f) {
            // |V + L|^2 accumulated component by component (one temporary: this is where the kernel's register count peaks)
            float hc = V.x + dirL.x, hh = hc * hc;
            hc = V.y + dirL.y; hh = fmaf(hc, hc, hh);
            hc = V.z + dirL.z; hh = fmaf(hc, hc, hh);
            const F3 r = brdf_eval(bf, hh, 0.0f, dir_nl);
            Lo = Lo + f3(r.x * (fr.dir_I[0] * shadowTest), r.y * (fr.dir_I[1] * shadowTest), r.z * (fr.dir_I[2] * shadowTest));
        }
    }

    if (!RDR_IBL_FIRST) image_based();
    out[0] = fminf(Lo.x, 10.0f);
    out[1] = fminf(Lo.y, 10.0f);
    out[2] = fminf(Lo.z, 10.0f);
    out[3] = (MASKED && mt.alpha_mode == 1) ? 1.0f : albedo[3];
}

RDR_DEV uint32_t find_draw_by_tri(const uint32_t* __restrict__ first_tri, uint32_t n, uint32_t id) {
    uint32_t lo = 0, hi = n;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (first_tri[mid] <= id) lo = mid; else hi = mid;
    }
    return lo;
}

// One thread per triangle of a view's draw list (index = visibility id): vertex stage, stored for the fragments.
// `pair_offset` is the rasteriser's exclusive scan of (triangle, tile) pair counts, starting at this view's first
// triangle and running on for `n_after` instances, with the grand total in *pair_total: a triangle that owns no
// pair (clipped away, culled, degenerate) can never be visible and is skipped.
// The 384-byte records leave through LDS: a lane that stored its own record would send it out as 24 separate 16-byte
// pieces, 384 bytes apart from its neighbours', and the memory system sees 2.4 GB of partial-line writes for 0.8 GB of
// records (ten-million-triangle grid). Half a wave's records are parked at a time, then the whole wave writes them
// out with neighbouring lanes on neighbouring 16-byte pieces: whole lines.
struct alignas(16) ShadeTriSlot { DShadeTri t; uint32_t spread[4]; };  // 400 bytes apart: the lanes' parking stores spread over the banks
__global__ void __launch_bounds__(256) k_shade_setup(const DFrame* __restrict__ frp, const DDraw* __restrict__ draws,
                                                     const uint32_t* __restrict__ first_tri, uint32_t n_draws, uint32_t n_tris,
                                                     const uint32_t* __restrict__ pair_offset, uint32_t n_after,
                                                     const uint32_t* __restrict__ pair_total, DShadeTri* __restrict__ out) {
    __shared__ ShadeTriSlot lds_slot[256 / 64][32];
    const uint32_t id = blockIdx.x * 256u + threadIdx.x;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const bool alive = id < n_tris && (id + 1u < n_after ? pair_offset[id + 1u] : pair_total[0]) != pair_offset[id];
    const unsigned long long live = __ballot(alive);
    if (!live) return;  // uniform
    DShadeTri t;
    if (alive) {
        const uint32_t di = find_draw_by_tri(first_tri, n_draws, id);
        shade_tri_setup(*frp, draws[di], id - first_tri[di], t);
        t._pad[0] = t._pad[1] = t._pad[2] = 0u;
    }
    const uint32_t wave_id0 = id - lane;  // the wave's 64 triangles are consecutive
#pragma unroll
    for (uint32_t half = 0; half < 2u; half++) {
        const unsigned long long part = (live >> (32u * half)) & 0xFFFFFFFFull;
        if (!part) continue;  // uniform
        if (alive && (lane >> 5) == half) lds_slot[wave][lane & 31u].t = t;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // a wave's LDS accesses run in order: its own stores are there
        // 32 records x 24 pieces of 16 bytes, 64 pieces per turn
        for (uint32_t f = lane; f < 32u * RDR_SHADE_TRI_VEC4; f += 64u) {
            const uint32_t r = f / RDR_SHADE_TRI_VEC4, piece = f % RDR_SHADE_TRI_VEC4;
            if ((part >> r) & 1ull) {
                const rdr_uint4_t q = ((const rdr_uint4_t*)&lds_slot[wave][r].t)[piece];
                ((rdr_uint4_t*)(out + (wave_id0 + 32u * half + r)))[piece] = q;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the reads are done before the other half parks its records
    }
}

#define SHADE_NO_FRAGMENT 0xFFu

// ---- Opaque draws + skybox, in three kernels.
//
// With 4x MSAA a pixel is shaded once per DISTINCT triangle among its samples. A 16x16 tile of the benchmark frame
// holds 288 such fragments on average and almost never exactly 256, so a block that shades "its" tile spends a second,
// nearly empty pass on the overhang while three of its four waves wait at a barrier. The fragments of the whole frame
// are therefore compacted into lists first and shaded from the lists, every wave full, no barrier anywhere:
//
//   k_frag_classify  one block per tile: the distinct triangles of every pixel's samples -> fragment records
//                    (triangle, pixel) appended to the list of the tile's screen rectangle; per pixel a state word
//                    (first fragment relative to the tile's, which fragment each sample resolves to); per tile the
//                    position of its first fragment.
//   k_shade_frags    a fixed grid of waves, each taking 64 consecutive fragments of a list at a time: pbr.frag per
//                    fragment, colour stored as RGBA16F beside the record.
//   k_frag_resolve   one lane per pixel: sky where a sample saw no triangle, sample colours from the fragment
//                    colours, multisample resolve (or the four sample colours, in tiles the blended pass goes on with).
//
// The screen is cut into 8 x 8 rectangles, one list each (RDR_FRAG_SEGS lists at fixed positions of one buffer, each as
// large as its rectangle can need, the fill counts a cache line apart: a single list would put one atomic per tile on
// one address). The rectangle in column i, row j of that grid belongs to die (i + j) mod 8, so every die (XCD, with its
// own L2) owns one rectangle of every row and every column of the grid: an even mix of sky and lit regions, and the
// waves of a die, which take the chunks of its lists in order, work on neighbouring tiles at any one time and share
// shadow-map and texture lines in that L2. Workgroups go to the dies round-robin by block index; a wave whose die has
// run dry takes chunks from the next die's lists, so no die idles at the end. List order inside a rectangle is the
// order in which its tiles' blocks arrive; colours do not depend on it.
template <int S>
__global__ void __launch_bounds__(256) k_frag_classify(const DFrame* __restrict__ frp, const uint32_t* __restrict__ vis, FragLists L) {
    __shared__ uint32_t lds_wave_total[4];
    __shared__ uint32_t lds_base;
    const DFrame& fr = *frp;
    const uint32_t tiles_x = (fr.W + RDR_TILE - 1) / RDR_TILE, tiles_y = (fr.H + RDR_TILE - 1) / RDR_TILE;
    // the 64 rectangles advance together, each through its tiles row by row
    const uint32_t rect = blockIdx.x & 63u, in_rect = blockIdx.x >> 6;
    const uint32_t gx = rect & 7u, gy = rect >> 3;
    const uint32_t tx = gx * L.rect_w + in_rect % L.rect_w, ty = gy * L.rect_h + in_rect / L.rect_w;
    if (tx >= tiles_x || ty >= tiles_y) return;
    const uint32_t list = ((gx + gy) & 7u) * 8u + gy;
    // thread t owns pixel (t % 16, t / 16) of the tile: a wave reads and writes whole rows of it
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int px = (int)tx * RDR_TILE + (int)(threadIdx.x & 15u), py = (int)ty * RDR_TILE + (int)(threadIdx.x >> 4);
    const bool in_view = px < (int)fr.W && py < (int)fr.H;
    const size_t plane_px = (size_t)fr.W * fr.H, o = (size_t)py * fr.W + px;

    // ---- the samples of this pixel: sky, or the u-th distinct triangle of the pixel
    uint32_t ids[S];
    uint32_t slot[S];
    uint32_t n_unique = 0;
#pragma unroll
    for (int s = 0; s < S; s++) {
        slot[s] = 15u;
        if (!in_view) continue;
        // a sample that holds a triangle id passed the LESS test against the cleared depth 1.0: no look at its depth
        const uint32_t id = vis[s * plane_px + o];
        if (id == 0xFFFFFFFFu) continue;
        uint32_t u = n_unique;
#pragma unroll
        for (int k = S - 1; k >= 0; k--) if ((uint32_t)k < n_unique && ids[k] == id) u = (uint32_t)k;
        if (u == n_unique) { ids[n_unique] = id; n_unique++; }
        slot[s] = u;
    }
    // ---- exclusive prefix sum of n_unique (0..4) over the block: three ballots per wave, then the wave totals
    uint32_t before = 0, wave_total = 0;
#pragma unroll
    for (int bit = 0; bit < 3; bit++) {
        const unsigned long long m = __ballot((n_unique >> bit) & 1u);
        before += (uint32_t)__popcll(m & ((1ull << lane) - 1ull)) << bit;
        wave_total += (uint32_t)__popcll(m) << bit;
    }
    if (lane == 0) lds_wave_total[wave] = wave_total;
    __syncthreads();
    uint32_t n_frag = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        if (w < wave) before += lds_wave_total[w];
        n_frag += lds_wave_total[w];
    }
    const uint32_t tile = ty * tiles_x + tx;
    if (threadIdx.x == 0) {
        const uint32_t at = n_frag ? atomicAdd(L.count + list * RDR_FRAG_CTR_STRIDE, n_frag) : 0u;  // <= cap - n_frag: the list holds every sample of its rectangle
        lds_base = list * L.cap + at;
        L.tile_first[tile] = list * L.cap + at;
    }
    __syncthreads();
    const uint32_t base = lds_base + before;
    const uint32_t xy = (uint32_t)px | ((uint32_t)py << 16);
#pragma unroll
    for (int k = 0; k < S; k++)
        if ((uint32_t)k < n_unique) L.recs[base + k] = make_uint2(ids[k], xy);
    uint32_t st = before;  // < 1024
#pragma unroll
    for (int s = 0; s < S; s++) st |= slot[s] << (16 + 4 * s);
    L.pixel_state[(size_t)tile * 256u + threadIdx.x] = st;
}

#ifndef RDR_SHADE_WAVES
#define RDR_SHADE_WAVES 4  // waves per SIMD the register allocation is held to (128 VGPRs)
#endif
template <bool FLOAT_TEXELS>
__global__ void __launch_bounds__(256, RDR_SHADE_WAVES) k_shade_frags(const DFrame* __restrict__ frp, const DPointLight* __restrict__ lights,
                                                                      const DShadeTri* __restrict__ tris, FragLists L) {
    __shared__ float lds_srgb[256];
    __shared__ float lds_park[12 * 256];  // see shade_fragment
    const DFrame& fr = *frp;
    lds_srgb[threadIdx.x] = fr.srgb_decode[threadIdx.x];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    // lane l looks after list l: its fill count, and where its 64-fragment chunks lie among those of its die's lists
    const uint32_t cnt = L.count[lane * RDR_FRAG_CTR_STRIDE];
    const uint32_t chunks = (cnt + 63u) >> 6;
    uint32_t incl = chunks;
#pragma unroll
    for (uint32_t d = 1; d < 8u; d <<= 1) {
        const uint32_t v = __shfl_up(incl, d);
        if ((lane & 7u) >= d) incl += v;
    }
    const uint32_t excl = incl - chunks;
    // fragment this lane takes of chunk k of die `die` (its position in recs / cols), or ~0 past the end of the list
    auto locate = [&](uint32_t die, uint32_t k) -> uint32_t {
        const unsigned long long owner = __ballot((lane >> 3) == die && excl <= k && k < incl);  // exactly one lane
        const int ol = __ffsll((long long)owner) - 1;
        const uint32_t local = (k - (uint32_t)__shfl(excl, ol)) * 64u + lane;
        return local < (uint32_t)__shfl(cnt, ol) ? (uint32_t)ol * L.cap + local : 0xFFFFFFFFu;
    };
    for (uint32_t turn = 0; turn < 8u; turn++) {
        const uint32_t die = (blockIdx.x + turn) & 7u;
        const uint32_t total = __shfl(incl, (int)(die * 8u + 7u));
        uint32_t* const claim = L.claim + die * RDR_FRAG_CTR_STRIDE;
        // somebody else's lists: not worth an atomic when they are known to be used up
        if (turn && __hip_atomic_load(claim, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= total) continue;
        // Chunks are claimed several at a time while the die's lists are far from used up (RDR_SHADE_BATCH of them in
        // the first half, then fewer, single chunks over the last eighth, which evens the waves out at the end): a
        // returning atomic on a word that hundreds of waves pull on takes microseconds.
#ifndef RDR_SHADE_BATCH
#define RDR_SHADE_BATCH 4u
#endif
        uint32_t batch = turn ? 1u : RDR_SHADE_BATCH;
        for (;;) {
            uint32_t k = 0;
            if (lane == 0) k = atomicAdd(claim, batch);
            k = __builtin_amdgcn_readfirstlane(k);
            if (k >= total) break;  // every wave gets here: the counter only grows
            const uint32_t k_end = min(k + batch, total);
            for (; k < k_end; k++) {  // uniform
                const uint32_t f = locate(die, k);
                uint2 colour = make_uint2(0u, 0u);
                if (f != 0xFFFFFFFFu) {
                    const uint2 rec = L.recs[f];
                    // One record per primitive among the chunk's fragments stands in for the material of all of them
                    // (the first lane's of each): one turn per distinct primitive, usually one or two.
                    const uint32_t prim = tris[rec.x].prim;
                    uint32_t mat_tri = rec.x;
                    for (unsigned long long todo = __ballot(1); todo;) {
                        const int l = __ffsll((long long)todo) - 1;
                        const uint32_t p = (uint32_t)__builtin_amdgcn_readlane((int)prim, l), t = (uint32_t)__builtin_amdgcn_readlane((int)rec.x, l);
                        const bool same = prim == p;
                        if (same) mat_tri = t;
                        todo &= ~__ballot(same);
                    }
                    float c[4];
                    shade_fragment<false, true, FLOAT_TEXELS>(fr, lights, lds_srgb, tris[rec.x], tris[mat_tri], (int)(rec.y & 0xFFFFu), (int)(rec.y >> 16), c,
                                                lds_park + threadIdx.x);
                    colour = pack_half4(c[0], c[1], c[2], c[3]);
                }
                // Where the colour goes is worked out again (by all lanes: the list's owner lane may hold no fragment of
                // this chunk) rather than kept across the shading: one register less, which lets the register file take
                // a wave of another kernel beside this one's four per SIMD.
                uint32_t k_again = k;
                asm volatile("" : "+s"(k_again));
                const uint32_t f_again = locate(die, k_again);
                if (f_again != 0xFFFFFFFFu) L.cols[f_again] = colour;
            }
            batch = turn ? 1u : (k_end < (total >> 1) ? RDR_SHADE_BATCH : (k_end < total - (total >> 3) ? min(RDR_SHADE_BATCH, 2u) : 1u));
        }
    }
}

// S == 1: writes hdr. S == 4: resolves straight into hdr, except in tiles whose blended-pass triangle list is not
// empty (blend_start < blend_end): those keep their four sample colours in `samples_out` for the blended pass, which
// resolves them afterwards.
template <int S>
__global__ void __launch_bounds__(256) k_frag_resolve(const DFrame* __restrict__ frp, FragLists L, uint2* __restrict__ hdr,
                                                      uint2* __restrict__ samples_out, const uint32_t* __restrict__ blend_start,
                                                      const uint32_t* __restrict__ blend_end) {
    const DFrame& fr = *frp;
    const uint32_t tiles_x = (fr.W + RDR_TILE - 1) / RDR_TILE;
    const uint32_t tile = blockIdx.x;
    const int tx = (int)(tile % tiles_x), ty = (int)(tile / tiles_x);
    const int px = tx * RDR_TILE + (int)(threadIdx.x & 15u), py = ty * RDR_TILE + (int)(threadIdx.x >> 4);  // as in k_frag_classify
    if (!(px < (int)fr.W && py < (int)fr.H)) return;
    const size_t plane_px = (size_t)fr.W * fr.H, o = (size_t)py * fr.W + px;
    const uint32_t state = L.pixel_state[(size_t)tile * 256u + threadIdx.x];
    const uint32_t first = L.tile_first[tile] + (state & 0xFFFFu);
    uint32_t slot[S];
#pragma unroll
    for (int s = 0; s < S; s++) { slot[s] = (state >> (16 + 4 * s)) & 15u; if (slot[s] == 15u) slot[s] = SHADE_NO_FRAGMENT; }
    bool any_sky = false;
#pragma unroll
    for (int s = 0; s < S; s++) any_sky = any_sky || slot[s] == SHADE_NO_FRAGMENT;
    uint2 sc[S];
#pragma unroll
    for (int s = 0; s < S; s++) sc[s] = slot[s] == SHADE_NO_FRAGMENT ? make_uint2(0u, 0u) : L.cols[first + slot[s]];
    if (any_sky) {
        float sky[4];
        shade_sky(fr, px, py, sky);
        const uint2 sky_packed = pack_half4(sky[0], sky[1], sky[2], sky[3]);
#pragma unroll
        for (int s = 0; s < S; s++) if (slot[s] == SHADE_NO_FRAGMENT) sc[s] = sky_packed;
    }
    const bool keep_samples = S > 1 && samples_out && blend_start[tile] < blend_end[tile];
    if (S == 1) {
        hdr[o] = sc[0];
    } else if (keep_samples) {
#pragma unroll
        for (int s = 0; s < S; s++) samples_out[s * plane_px + o] = sc[s];
    } else {
        float sum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int s = 0; s < S; s++) {
            float c[4];
            unpack_half4(sc[s], c[0], c[1], c[2], c[3]);
#pragma unroll
            for (int k = 0; k < 4; k++) sum[k] += c[k];
        }
        hdr[o] = pack_half4(sum[0] * (1.0f / S), sum[1] * (1.0f / S), sum[2] * (1.0f / S), sum[3] * (1.0f / S));
    }
}

// The visibility image (triangle id per sample, 0xFFFFFFFF where nothing covers) from the fragment lists: what a frame
// whose lists were built inside the rasteriser hands to rdr_read_visibility instead of storing the image every frame.
template <int S>
__global__ void __launch_bounds__(256) k_vis_from_lists(FragLists L, uint32_t W, uint32_t H, uint32_t* __restrict__ vis) {
    const uint32_t tiles_x = (W + RDR_TILE - 1) / RDR_TILE;
    const uint32_t tile = blockIdx.x;
    const uint32_t px = (tile % tiles_x) * RDR_TILE + (threadIdx.x & 15u), py = (tile / tiles_x) * RDR_TILE + (threadIdx.x >> 4);
    if (px >= W || py >= H) return;
    const size_t plane_px = (size_t)W * H, o = (size_t)py * W + px;
    const uint32_t state = L.pixel_state[(size_t)tile * 256u + threadIdx.x];
    const uint32_t first = L.tile_first[tile] + (state & 0xFFFFu);
#pragma unroll
    for (int s = 0; s < S; s++) {
        const uint32_t slot = (state >> (16 + 4 * s)) & 15u;
        vis[s * plane_px + o] = slot == 15u ? 0xFFFFFFFFu : L.recs[first + slot].x;
    }
}

#define BLEND_CHUNK 64
#define BLEND_NONE 0xFFFFFFFFu

// Blended (MASK + BLEND) draws, back to front (VulkanRenderer.cpp:862-866, 1059-1062), in three kernels so that the
// expensive part - pbr.frag for every fragment that passes the depth test - runs over the whole GPU at full
// occupancy instead of inside the few tiles that hold blended geometry:
//
//   k_blend_collect  same tile walk as k_raster, one wave per 8x8 quadrant. The order of the fragments of a pixel
//                    is the order in which its wave meets the triangles; depth is tested and written at once (it
//                    does not depend on shading) and every passing fragment is appended to a global list as
//                    (triangle, pixel, sample mask, next fragment of the same pixel). List space is handed out in
//                    runs of RDR_FRAG_RUN slots: one atomic per block for the first run of each of its waves, one
//                    per further run; slots a wave leaves unused are marked empty.
//   k_blend_shade    one lane per list slot, any order: shades the fragment, stores its colour.
//   k_blend_resolve  one lane per pixel of the tiles with blended triangles: walks the pixel's chain in order,
//                    blends into the sample colours the opaque pass left, resolves.
//
// COUNT: a dry run of the same walk that writes nothing but tile_slots[tile] = the list slots the tile's waves would
// take (whole runs, exactly as the real pass hands them out). The host uses it when the upper bound it has for the
// list is larger than it is willing to allocate: it then cuts the tiles into groups whose slots fit the list and runs
// collect / shade / resolve once per group (`tile0` = first tile of the group), so that any number of blended layers
// is rendered in bounded memory.
template <int S, bool COUNT>
__global__ void __launch_bounds__(256) k_blend_collect(const DView* __restrict__ views, const DTriRec* __restrict__ recs,
                                                       const uint32_t* __restrict__ sorted_vals, const uint32_t* __restrict__ tile_start,
                                                       const uint32_t* __restrict__ tile_end, uint4* __restrict__ frags,
                                                       uint32_t* __restrict__ frag_head, uint32_t* __restrict__ frag_used, uint32_t capacity,
                                                       uint32_t tile0, uint32_t* __restrict__ tile_slots) {
    __shared__ uint32_t lds_rec[BLEND_CHUNK * 32];
    __shared__ uint32_t lds_ids[BLEND_CHUNK];
    __shared__ uint32_t lds_mode[BLEND_CHUNK];  // RASTER_LOCAL32: the record was rewritten with 32-bit tile-local edge functions
    __shared__ uint32_t lds_first_run;
    const DView vw = views[0];  // the blended view of the batch; its tile lists start at tile_base
    const uint32_t tile = tile0 + blockIdx.x;
    const uint32_t begin = tile_start[vw.tile_base + tile], end = tile_end[vw.tile_base + tile];
    if (begin >= end) return;
    if (threadIdx.x == 0) {
        if (COUNT) { lds_first_run = 0u; tile_slots[tile] = 4u * RDR_FRAG_RUN; }
        else lds_first_run = atomicAdd(frag_used, 4u * RDR_FRAG_RUN);
    }
    uint32_t extra_runs = 0;  // COUNT: runs this wave takes beyond its first
    const int tx = (int)(tile % vw.tiles_x), ty = (int)(tile / vw.tiles_x);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int qx0 = tx * RDR_TILE + (wave & 1) * 8, qy0 = ty * RDR_TILE + (wave >> 1) * 8;
    const int px = qx0 + (lane & 7), py = qy0 + (lane >> 3);
    const bool in_view = px < (int)vw.w && py < (int)vw.h;
    const size_t plane_px = (size_t)vw.w * vw.h, o = (size_t)py * vw.w + px;
    float z[S];
#pragma unroll
    for (int s = 0; s < S; s++) z[s] = in_view ? vw.depth[s * plane_px + o] : 0.0f;
    uint32_t run_at = 0, run_end = 0;           // wave-uniform: next free slot of this wave's run and its end
    bool have_run = false;
    uint32_t my_first = BLEND_NONE, my_last = BLEND_NONE;

    const int sxo[4] = {96, 224, 32, 160}, syo[4] = {32, 96, 160, 224};
    for (uint32_t base = begin; base < end; base += BLEND_CHUNK) {
        const uint32_t cnt = min((uint32_t)BLEND_CHUNK, end - base);
        __syncthreads();
        if (threadIdx.x < cnt) lds_ids[threadIdx.x] = sorted_vals[base + threadIdx.x];
        __syncthreads();
        for (uint32_t dw = threadIdx.x; dw < cnt * 32u; dw += 256u) lds_rec[dw] = ((const uint32_t*)(recs + lds_ids[dw >> 5]))[dw & 31u];
        __syncthreads();
        if (threadIdx.x < cnt) lds_mode[threadIdx.x] = raster_localise<S>(lds_rec + threadIdx.x * 32, tx * RDR_TILE, ty * RDR_TILE);
        __syncthreads();
        if (!have_run) { run_at = lds_first_run + (uint32_t)wave * RDR_FRAG_RUN; run_end = run_at + RDR_FRAG_RUN; have_run = true; }
        const int qlx = (wave & 1) * 8, qly = (wave >> 1) * 8;  // the quadrant inside the tile
        bool hit = false;
        if ((uint32_t)lane < cnt) {
            if (lds_mode[lane] == RASTER_LOCAL32) {
                const uint32_t b4 = ((const RasterLocal*)(lds_rec + lane * 32))->box4;
                hit = (int)(b4 & 15u) <= qlx + 7 && (int)((b4 >> 8) & 15u) >= qlx && (int)((b4 >> 4) & 15u) <= qly + 7 && (int)((b4 >> 12) & 15u) >= qly;
            } else {
                const DTriRec* r = (const DTriRec*)(lds_rec + lane * 32);
                hit = r->px0 <= qx0 + 7 && r->px1 >= qx0 && r->py0 <= qy0 + 7 && r->py1 >= qy0;
            }
        }
        unsigned long long mask = __ballot(hit);
        while (mask) {
            const int ri = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            uint32_t inside = in_view ? (1u << S) - 1u : 0u;
            double dzdX, dzdY, z00;
            uint32_t rid;
            if (lds_mode[ri] == RASTER_LOCAL32) {
                const RasterLocal* f = (const RasterLocal*)(lds_rec + ri * 32);
                const int u0 = (qlx + (lane & 7)) << 8, v0 = (qly + (lane >> 3)) << 8;
                int Ep[3];
#pragma unroll
                for (int i = 0; i < 3; i++) Ep[i] = f->C[i] + __mul24(f->A[i], u0) + __mul24(f->B[i], v0);  // |A|, |B| <= 2^18
                // a sample is outside when any of its three edge values is negative: the sign bit of their OR
                if (S == 1) { if ((Ep[0] | Ep[1] | Ep[2]) < 0) inside = 0u; }
                else {
                    uint32_t outside = 0u;
#pragma unroll
                    for (int sm = 0; sm < S; sm++)
                        outside |= ((uint32_t)((Ep[0] + f->off[0][sm]) | (Ep[1] + f->off[1][sm]) | (Ep[2] + f->off[2][sm])) >> 31) << sm;
                    inside &= ~outside;
                }
                dzdX = f->dzdX; dzdY = f->dzdY; z00 = f->z00; rid = f->id;
            } else {
                const DTriRec* r = (const DTriRec*)(lds_rec + ri * 32);
                const int n = (int)(r->n_view & 255u);
                for (int i = 0; i < n; i++) {
                    const int j = (i + 1 == n) ? 0 : i + 1;
                    const int Xi = r->X[i], Yi = r->Y[i];
                    const int dX = r->X[j] - Xi, dY = r->Y[j] - Yi;
                    const long long E0 = (long long)dX * (long long)(256 * py - Yi) - (long long)dY * (long long)(256 * px - Xi);
                    const long long bias = ((dY < 0) || (dY == 0 && dX > 0)) ? 0 : 1;
#pragma unroll
                    for (int s = 0; s < S; s++) {
                        const int ax = S == 1 ? 128 : sxo[s], ay = S == 1 ? 128 : syo[s];
                        if (E0 + ((long long)dX * ay - (long long)dY * ax) < bias) inside &= ~(1u << s);
                    }
                }
                dzdX = r->dzdX; dzdY = r->dzdY; z00 = r->z00; rid = r->id;
            }
            uint32_t pass = 0;
#pragma unroll
            for (int s = 0; s < S; s++) {
                if (!(inside & (1u << s))) continue;
                const int Px = 256 * px + (S == 1 ? 128 : sxo[s]), Py = 256 * py + (S == 1 ? 128 : syo[s]);
                float zz = (float)fma(dzdX, (double)Px, fma(dzdY, (double)Py, z00));
                zz = zz < 0.0f ? 0.0f : (zz > 1.0f ? 1.0f : zz);
                if (zz < z[s]) { z[s] = zz; pass |= 1u << s; }
            }
            const unsigned long long pm = __ballot(pass != 0u);
            if (!pm) continue;
            const uint32_t want = (uint32_t)__popcll(pm);
            if (COUNT) {
                if (run_at + want > run_end) { extra_runs++; run_at = 0u; run_end = RDR_FRAG_RUN; }
                run_at += want;
                continue;
            }
            if (run_at + want > run_end) {  // this run cannot take them: mark what is left of it empty, take another
                if (run_at + (uint32_t)lane < run_end) frags[run_at + lane] = make_uint4(BLEND_NONE, 0u, 0u, BLEND_NONE);
                if (run_at + 64u + (uint32_t)lane < run_end) frags[run_at + 64u + lane] = make_uint4(BLEND_NONE, 0u, 0u, BLEND_NONE);
                uint32_t fresh = 0;
                if (lane == 0) fresh = atomicAdd(frag_used, RDR_FRAG_RUN);
                run_at = __shfl(fresh, 0);
                run_end = run_at + RDR_FRAG_RUN;
            }
            if (pass && run_end <= capacity) {
                const uint32_t f = run_at + (uint32_t)__popcll(pm & ((1ull << lane) - 1ull));
                frags[f] = make_uint4(rid, (uint32_t)px | ((uint32_t)py << 16), pass, BLEND_NONE);
                if (my_last != BLEND_NONE) frags[my_last].w = f; else my_first = f;
                my_last = f;
            }
            run_at += want;
        }
    }
    if (COUNT) {
        __syncthreads();  // behind thread 0's store of the four first runs
        if (lane == 0 && extra_runs) atomicAdd(&tile_slots[tile], extra_runs * RDR_FRAG_RUN);
        return;
    }
    // the unused tail of the wave's last run
    if (have_run && run_end <= capacity) {
        if (run_at + (uint32_t)lane < run_end) frags[run_at + lane] = make_uint4(BLEND_NONE, 0u, 0u, BLEND_NONE);
        if (run_at + 64u + (uint32_t)lane < run_end) frags[run_at + 64u + lane] = make_uint4(BLEND_NONE, 0u, 0u, BLEND_NONE);
    }
    if (!in_view) return;
#pragma unroll
    for (int s = 0; s < S; s++) vw.depth[s * plane_px + o] = z[s];
    frag_head[o] = my_first;
}

// One lane per fragment slot; the grid is fixed and strides over the slots in use.
template <bool FLOAT_TEXELS>
__global__ void __launch_bounds__(256, 4) k_blend_shade(const DFrame* __restrict__ frp, const DPointLight* __restrict__ lights,
                                                                      const DShadeTri* __restrict__ tris,
                                                                      const uint4* __restrict__ frags, float4* __restrict__ frag_col,
                                                                      const uint32_t* __restrict__ frag_used, uint32_t capacity) {
    __shared__ float lds_srgb[256];
    __shared__ float lds_park[12 * 256];  // see shade_fragment
    const DFrame& fr = *frp;
    lds_srgb[threadIdx.x] = fr.srgb_decode[threadIdx.x];
    __syncthreads();
    const uint32_t used = min(frag_used[0], capacity);
    for (uint32_t f = blockIdx.x * 256u + threadIdx.x; f < used; f += gridDim.x * 256u) {
        const uint4 m = frags[f];
        if (m.x == BLEND_NONE) continue;
        float c[4];
        shade_fragment<true, true, FLOAT_TEXELS>(fr, lights, lds_srgb, tris[m.x], tris[m.x], (int)(m.y & 0xFFFFu), (int)(m.y >> 16), c, lds_park + threadIdx.x);
        frag_col[f] = make_float4(c[0], c[1], c[2], c[3]);
    }
}

template <int S>
__global__ void __launch_bounds__(256) k_blend_resolve(const DView* __restrict__ views, const uint32_t* __restrict__ tile_start,
                                                       const uint32_t* __restrict__ tile_end, const uint4* __restrict__ frags,
                                                       const float4* __restrict__ frag_col, const uint32_t* __restrict__ frag_head,
                                                       const uint2* __restrict__ color, uint2* __restrict__ hdr, uint32_t tile0) {
    const DView vw = views[0];
    const uint32_t tile = tile0 + blockIdx.x;
    if (tile_start[vw.tile_base + tile] >= tile_end[vw.tile_base + tile]) return;
    const int px = (int)(tile % vw.tiles_x) * RDR_TILE + ((int)threadIdx.x & 15), py = (int)(tile / vw.tiles_x) * RDR_TILE + ((int)threadIdx.x >> 4);
    if (px >= (int)vw.w || py >= (int)vw.h) return;
    const size_t plane_px = (size_t)vw.w * vw.h, o = (size_t)py * vw.w + px;
    float col[S][4];
#pragma unroll
    for (int s = 0; s < S; s++) unpack_half4(color[s * plane_px + o], col[s][0], col[s][1], col[s][2], col[s][3]);
    for (uint32_t f = frag_head[o]; f != BLEND_NONE;) {
        const uint4 m = frags[f];
        const float4 src = frag_col[f];
        const float a = src.w;
#pragma unroll
        for (int s = 0; s < S; s++)
            if (m.z & (1u << s)) {
                col[s][0] = quantize_half(src.x * a + col[s][0] * (1.0f - a));
                col[s][1] = quantize_half(src.y * a + col[s][1] * (1.0f - a));
                col[s][2] = quantize_half(src.z * a + col[s][2] * (1.0f - a));
                col[s][3] = quantize_half(a);
            }
        f = m.w;
    }
    if (S == 1) {
        hdr[o] = pack_half4(col[0][0], col[0][1], col[0][2], col[0][3]);
    } else {  // multisample resolve: the sample colours are RGBA16F values already
        float sum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int s = 0; s < S; s++)
#pragma unroll
            for (int k = 0; k < 4; k++) sum[k] += col[s][k];
        hdr[o] = pack_half4(sum[0] * (1.0f / S), sum[1] * (1.0f / S), sum[2] * (1.0f / S), sum[3] * (1.0f / S));
    }
}

}  // namespace

void launch_shade_setup(hipStream_t s, const DFrame* fr, const DDraw* draws, const uint32_t* first_tri, uint32_t n_draws, uint32_t n_tris,
                        const uint32_t* pair_offset, uint32_t n_after, const uint32_t* pair_total, DShadeTri* out) {
    if (!n_tris) return;
    hipLaunchKernelGGL(k_shade_setup, dim3((n_tris + 255) / 256), dim3(256), 0, s, fr, draws, first_tri, n_draws, n_tris, pair_offset, n_after,
                       pair_total, out);
}
uint32_t shade_frags_blocks_per_cu() { return RDR_SHADE_WAVES; }  // a 256-thread block is one wave per SIMD
size_t shade_frag_slots(uint32_t W, uint32_t H, int samples) {
    const uint32_t tiles_x = (W + RDR_TILE - 1) / RDR_TILE, tiles_y = (H + RDR_TILE - 1) / RDR_TILE;
    return (size_t)RDR_FRAG_SEGS * ((tiles_x + 7u) / 8u) * ((tiles_y + 7u) / 8u) * 256u * (size_t)samples;
}
FragLists make_frag_lists(uint32_t W, uint32_t H, int samples, void* frag_recs, void* frag_cols, uint32_t* pixel_state, uint32_t* tile_first,
                          uint32_t* list_count, uint32_t* list_claim) {
    const uint32_t tiles_x = (W + RDR_TILE - 1) / RDR_TILE, tiles_y = (H + RDR_TILE - 1) / RDR_TILE;
    FragLists L;
    L.rect_w = (tiles_x + 7u) / 8u; L.rect_h = (tiles_y + 7u) / 8u;
    L.cap = L.rect_w * L.rect_h * 256u * (uint32_t)samples;
    L.count = list_count; L.claim = list_claim;
    L.recs = (uint2*)frag_recs; L.cols = (uint2*)frag_cols; L.pixel_state = pixel_state; L.tile_first = tile_first;
    return L;
}
void launch_frag_classify(hipStream_t s, int samples, const DFrame* fr, const uint32_t* vis, const FragLists& L) {
    const uint32_t grid = 64u * L.rect_w * L.rect_h;
    if (samples == 1) hipLaunchKernelGGL(k_frag_classify<1>, dim3(grid), dim3(256), 0, s, fr, vis, L);
    else hipLaunchKernelGGL(k_frag_classify<4>, dim3(grid), dim3(256), 0, s, fr, vis, L);
}
void launch_shade_frags(hipStream_t s, const DFrame* fr, const DPointLight* lights, const DShadeTri* tris, const FragLists& L, uint32_t blocks,
                        bool float_texels) {
    if (float_texels) hipLaunchKernelGGL(k_shade_frags<true>, dim3(blocks), dim3(256), 0, s, fr, lights, tris, L);
    else hipLaunchKernelGGL(k_shade_frags<false>, dim3(blocks), dim3(256), 0, s, fr, lights, tris, L);
}
void launch_vis_from_lists(hipStream_t s, int samples, const FragLists& L, uint32_t W, uint32_t H, uint32_t* vis) {
    const uint32_t tiles = ((W + RDR_TILE - 1) / RDR_TILE) * ((H + RDR_TILE - 1) / RDR_TILE);
    if (samples == 1) hipLaunchKernelGGL(k_vis_from_lists<1>, dim3(tiles), dim3(256), 0, s, L, W, H, vis);
    else hipLaunchKernelGGL(k_vis_from_lists<4>, dim3(tiles), dim3(256), 0, s, L, W, H, vis);
}
void launch_frag_resolve(hipStream_t s, int samples, const DFrame* fr, const FragLists& L, void* hdr, void* samples_out, const uint32_t* blend_start,
                         const uint32_t* blend_end, uint32_t W, uint32_t H) {
    const uint32_t tiles = ((W + RDR_TILE - 1) / RDR_TILE) * ((H + RDR_TILE - 1) / RDR_TILE);
    if (samples == 1) hipLaunchKernelGGL(k_frag_resolve<1>, dim3(tiles), dim3(256), 0, s, fr, L, (uint2*)hdr, (uint2*)samples_out, blend_start, blend_end);
    else hipLaunchKernelGGL(k_frag_resolve<4>, dim3(tiles), dim3(256), 0, s, fr, L, (uint2*)hdr, (uint2*)samples_out, blend_start, blend_end);
}
void launch_raster_blend(hipStream_t s, int samples, const DFrame* fr, const DPointLight* lights, const DView* view, const DShadeTri* tris, const DTriRec* recs,
                         const uint32_t* sorted_vals, const uint32_t* tile_start, const uint32_t* tile_end, uint32_t tile0, uint32_t n_tiles, void* color,
                         void* hdr, void* frags, void* frag_col, uint32_t* frag_head, uint32_t* frag_used, uint32_t capacity, bool float_texels) {
    if (!n_tiles) return;
    if (samples == 1)
        hipLaunchKernelGGL((k_blend_collect<1, false>), dim3(n_tiles), dim3(256), 0, s, view, recs, sorted_vals, tile_start, tile_end, (uint4*)frags, frag_head,
                           frag_used, capacity, tile0, (uint32_t*)nullptr);
    else
        hipLaunchKernelGGL((k_blend_collect<4, false>), dim3(n_tiles), dim3(256), 0, s, view, recs, sorted_vals, tile_start, tile_end, (uint4*)frags, frag_head,
                           frag_used, capacity, tile0, (uint32_t*)nullptr);
    if (float_texels)
        hipLaunchKernelGGL(k_blend_shade<true>, dim3(RDR_BLEND_SHADE_BLOCKS), dim3(256), 0, s, fr, lights, tris, (const uint4*)frags, (float4*)frag_col, frag_used, capacity);
    else
        hipLaunchKernelGGL(k_blend_shade<false>, dim3(RDR_BLEND_SHADE_BLOCKS), dim3(256), 0, s, fr, lights, tris, (const uint4*)frags, (float4*)frag_col, frag_used, capacity);
    if (samples == 1)
        hipLaunchKernelGGL(k_blend_resolve<1>, dim3(n_tiles), dim3(256), 0, s, view, tile_start, tile_end, (const uint4*)frags, (const float4*)frag_col,
                           frag_head, (const uint2*)color, (uint2*)hdr, tile0);
    else
        hipLaunchKernelGGL(k_blend_resolve<4>, dim3(n_tiles), dim3(256), 0, s, view, tile_start, tile_end, (const uint4*)frags, (const float4*)frag_col,
                           frag_head, (const uint2*)color, (uint2*)hdr, tile0);
}
void launch_blend_count(hipStream_t s, int samples, const DView* view, const DTriRec* recs, const uint32_t* sorted_vals, const uint32_t* tile_start,
                        const uint32_t* tile_end, uint32_t n_tiles, uint32_t* tile_slots) {
    if (!n_tiles) return;
    if (samples == 1)
        hipLaunchKernelGGL((k_blend_collect<1, true>), dim3(n_tiles), dim3(256), 0, s, view, recs, sorted_vals, tile_start, tile_end, (uint4*)nullptr,
                           (uint32_t*)nullptr, (uint32_t*)nullptr, 0u, 0u, tile_slots);
    else
        hipLaunchKernelGGL((k_blend_collect<4, true>), dim3(n_tiles), dim3(256), 0, s, view, recs, sorted_vals, tile_start, tile_end, (uint4*)nullptr,
                           (uint32_t*)nullptr, (uint32_t*)nullptr, 0u, 0u, tile_slots);
}
