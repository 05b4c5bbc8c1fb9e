// TEST INFRASTRUCTURE — CPU restatement ("oracle"); see orc_types.h.
//
// Cube-map sampling and the two image-based-lighting bakes
// (Renderer/VulkanRendererEnvironment.cpp:193-250 irradiance, :324-391
// GGX prefilter; shaders envbake.vert, envbakediffuse.frag,
// envbakespecular.frag). Cube face selection and face coordinates follow the
// Vulkan 1.3 "Cube Map Face Selection" tables; filtering is bilinear and
// seamless: a footprint texel that falls outside the selected face is the texel
// across the shared cube edge (cube_fold below; at a corner the row index is
// clamped first), the same choice as the HIP path.
#include <algorithm>
#include <cmath>

#include "orc_types.h"

namespace orc {

// Vulkan cube face selection: major axis = largest magnitude, z wins ties over
// y over x. Returns face (0..5 = +X,-X,+Y,-Y,+Z,-Z) and (s,t) in [0,1].
static inline int cube_face(float rx, float ry, float rz, float& s, float& t) {
    const float ax = std::fabs(rx), ay = std::fabs(ry), az = std::fabs(rz);
    int face; float sc, tc, ma;
    if (az >= ax && az >= ay) { face = rz < 0.0f ? 5 : 4; sc = rz < 0.0f ? -rx : rx; tc = -ry; ma = az; }
    else if (ay >= ax) { face = ry < 0.0f ? 3 : 2; sc = rx; tc = ry < 0.0f ? -rz : rz; ma = ay; }
    else { face = rx < 0.0f ? 1 : 0; sc = rx < 0.0f ? rz : -rz; tc = -ry; ma = ax; }
    s = 0.5f * (sc / ma) + 0.5f;
    t = 0.5f * (tc / ma) + 0.5f;
    return face;
}

// Texel (i, j) of `face` at resolution r, where i or j may be one step outside [0, r): the texel across the cube
// edge. Vulkan filters linearly across cube faces (seamless cube maps are the default), which makes a lookup
// continuous in its direction. Texel centres are odd integers on a grid of half-texel steps, U = 2i + 1 - r in
// (-r, r) on a face at distance r; a step outside is folded over the edge (distance r - 1 on the old major axis,
// r on the crossed one), so the along-edge index carries over exactly. At a corner (both outside) the t index is
// clamped first. Integer arithmetic only.
 void cube_fold(int face, int i, int j, int r, int& face_out, int& i_out, int& j_out) {
    if (i >= 0 && i < r && j >= 0 && j < r) { face_out = face; i_out = i; j_out = j; return; }
    if ((i < 0 || i >= r) && (j < 0 || j >= r)) j = j < 0 ? 0 : r - 1;
    int U = 2 * i + 1 - r, V = 2 * j + 1 - r, M = r;
    if (U > r) { M = 2 * r - U; U = r; } else if (U < -r) { M = 2 * r + U; U = -r; }
    else if (V > r) { M = 2 * r - V; V = r; } else if (V < -r) { M = 2 * r + V; V = -r; }
    int X, Y, Z;  // point on the cube, components in half-texel units
    switch (face) {
        case 0: X = M; Y = -V; Z = -U; break;
        case 1: X = -M; Y = -V; Z = U; break;
        case 2: X = U; Y = M; Z = V; break;
        case 3: X = U; Y = -M; Z = -V; break;
        case 4: X = U; Y = -V; Z = M; break;
        default: X = -U; Y = -V; Z = -M; break;
    }
    const int ax = X < 0 ? -X : X, ay = Y < 0 ? -Y : Y, az = Z < 0 ? -Z : Z;
    int sc, tc;
    if (az >= ax && az >= ay) { face_out = Z < 0 ? 5 : 4; sc = Z < 0 ? -X : X; tc = -Y; }
    else if (ay >= ax) { face_out = Y < 0 ? 3 : 2; sc = X; tc = Y < 0 ? -Z : Z; }
    else { face_out = X < 0 ? 1 : 0; sc = X < 0 ? Z : -Z; tc = -Y; }
    i_out = (sc + r - 1) / 2;
    j_out = (tc + r - 1) / 2;
    i_out = i_out < 0 ? 0 : (i_out > r - 1 ? r - 1 : i_out);
    j_out = j_out < 0 ? 0 : (j_out > r - 1 ? r - 1 : j_out);
}

// face-local texel index of (i, j) folded over the edges, as an offset into one level
size_t cube_texel_offset(int face, int i, int j, int r) {
    int f2, i2, j2;
    cube_fold(face, i, j, r, f2, i2, j2);
    return ((size_t)f2 * r + j2) * r + i2;
}

void cube_sample(const Cube& cube, uint32_t level, float dx, float dy, float dz, float out[3]) {
    float s, t;
    const int face = cube_face(dx, dy, dz, s, t);
    const int r = (int)std::max(1u, cube.res >> level);
    const uint32_t* px = cube.level[level].data();
    const float tu = s * (float)r - 0.5f, tv = t * (float)r - 0.5f;
    const float fu = std::floor(tu), fv = std::floor(tv);
    const float a = tu - fu, b = tv - fv;
    // s, t lie in [0, 1], so the four texels are at most one step outside the face
    const int i0 = std::min(std::max((int)fu, -1), r - 1), j0 = std::min(std::max((int)fv, -1), r - 1);
    float c00[3], c10[3], c01[3], c11[3];
    unpack_r11g11b10(px[cube_texel_offset(face, i0, j0, r)], c00[0], c00[1], c00[2]);
    unpack_r11g11b10(px[cube_texel_offset(face, i0 + 1, j0, r)], c10[0], c10[1], c10[2]);
    unpack_r11g11b10(px[cube_texel_offset(face, i0, j0 + 1, r)], c01[0], c01[1], c01[2]);
    unpack_r11g11b10(px[cube_texel_offset(face, i0 + 1, j0 + 1, r)], c11[0], c11[1], c11[2]);
    for (int k = 0; k < 3; k++) {
        const float top = (1.0f - a) * c00[k] + a * c10[k], bot = (1.0f - a) * c01[k] + a * c11[k];
        out[k] = (1.0f - b) * top + b * bot;
    }
}

// Direction through the centre of texel (x,y) of `face` for a res x res bake.
// envbake.vert:21-24 pushes inverse(proj*view) with proj = scale(1,-1,1) and
// the per-face view rotations of VulkanRendererEnvironment.cpp:209-217; the
// fullscreen quad sits at z = 1, so position = inverse(view) * (xn, -yn, 1).
// With the rotation matrices written out (R^-1 = R^T) that is:
static inline V3 bake_direction(int face, int x, int y, int res) {
    const float xn = ((float)x + 0.5f) * (2.0f / (float)res) - 1.0f;
    const float yn = ((float)y + 0.5f) * (2.0f / (float)res) - 1.0f;
    const float vx = xn, vy = -yn, vz = 1.0f;  // view-space position after undoing proj
    switch (face) {
        case 0: return {vz, vy, -vx};    // view = Ry(-90deg)
        case 1: return {-vz, vy, vx};    // view = Ry(+90deg)
        case 2: return {vx, vz, -vy};    // view = Rx(+90deg)
        case 3: return {vx, -vz, vy};    // view = Rx(-90deg)
        case 4: return {vx, vy, vz};     // identity
        default: return {-vx, vy, -vz};  // view = Ry(180deg)
    }
}

// envbakediffuse.frag:12-40, 256^2 per face (VulkanRenderer.h:220)
void bake_irradiance(Ctx& c) {
    const int res = 256;
    c.irradiance.res = res; c.irradiance.levels = 1;
    c.irradiance.level.assign(1, std::vector<uint32_t>((size_t)6 * res * res));
    const float PI = 3.14159265359f;
    const int nSamples = 24;
    const float stepRads = 2 * PI / nSamples;
#pragma omp parallel for schedule(dynamic, 16) collapse(2)
    for (int face = 0; face < 6; face++)
        for (int y = 0; y < res; y++)
            for (int x = 0; x < res; x++) {
                V3 normal = normalize(bake_direction(face, x, y, res));
                V3 up{0.0f, 1.0f, 0.0f};
                V3 right = normalize(cross(up, normal));
                up = normalize(cross(normal, right));
                float irr[3] = {0, 0, 0};
                for (int i = 0; i < nSamples; i++)
                    for (int j = 0; j < nSamples / 4; j++) {
                        const float phi = i * stepRads, theta = j * stepRads;
                        const float st = std::sin(theta), ct = std::cos(theta);
                        const float tx = st * std::cos(phi), ty = st * std::sin(phi), tz = ct;
                        V3 sv = right * tx + up * ty + normal * tz;
                        float e[3];
                        cube_sample(c.env, 0, sv.x, sv.y, sv.z, e);
                        for (int k = 0; k < 3; k++) irr[k] += e[k] * ct * st;
                    }
                const float norm = 1.0f / (float)(nSamples * nSamples / 4);
                for (int k = 0; k < 3; k++) irr[k] = PI * irr[k] * norm;
                c.irradiance.level[0][((size_t)face * res + y) * res + x] = pack_r11g11b10(irr[0], irr[1], irr[2]);
            }
}

// envbakespecular.frag:17-30
static inline float radical_inverse(uint32_t bits) {
    bits = (bits << 16u) | (bits >> 16u);
    bits = ((bits & 0x55555555u) << 1u) | ((bits & 0xAAAAAAAAu) >> 1u);
    bits = ((bits & 0x33333333u) << 2u) | ((bits & 0xCCCCCCCCu) >> 2u);
    bits = ((bits & 0x0F0F0F0Fu) << 4u) | ((bits & 0xF0F0F0F0u) >> 4u);
    bits = ((bits & 0x00FF00FFu) << 8u) | ((bits & 0xFF00FF00u) >> 8u);
    return (float)bits * 2.3283064365386963e-10f;
}

// envbakespecular.frag:32-86; 512^2 base, 10 levels, roughness = level/9
// (VulkanRenderer.h:225, VulkanRendererEnvironment.cpp:350-360)
void bake_prefiltered(Ctx& c) {
    const int base = 512, levels = 10;
    c.prefiltered.res = base; c.prefiltered.levels = levels;
    c.prefiltered.level.resize(levels);
    const float PI = 3.14159265359f;
    const int nSamples = 128;
    for (int l = 0; l < levels; l++) {
        const int res = base >> l;
        const float roughness = (float)l / 9;
        c.prefiltered.level[l].assign((size_t)6 * res * res, 0u);
#pragma omp parallel for schedule(dynamic, 16) collapse(2)
        for (int face = 0; face < 6; face++)
            for (int y = 0; y < res; y++)
                for (int x = 0; x < res; x++) {
                    const V3 N = normalize(bake_direction(face, x, y, res));
                    const V3 V = N;
                    float col[3] = {0, 0, 0}, total = 0.0f;
                    for (int i = 0; i < nSamples; i++) {
                        const float xi_x = (float)i / (float)nSamples, xi_y = radical_inverse((uint32_t)i);
                        const float a = roughness * roughness;
                        const float phi = 2.0f * PI * xi_x;
                        const float cosTheta = std::sqrt((1.0f - xi_y) / (1.0f + (a * a - 1.0f) * xi_y));
                        const float sinTheta = std::sqrt(1.0f - cosTheta * cosTheta);
                        const V3 Ht{std::cos(phi) * sinTheta, std::sin(phi) * sinTheta, cosTheta};
                        const V3 upv = std::fabs(N.z) < 0.999f ? V3{0, 0, 1} : V3{1, 0, 0};
                        const V3 tangent = normalize(cross(upv, N));
                        const V3 bitangent = cross(N, tangent);
                        const V3 H = normalize(tangent * Ht.x + bitangent * Ht.y + N * Ht.z);
                        const V3 L = normalize(H * (2.0f * dot(V, H)) - V);
                        const float NdotL = std::max(dot(N, L), 0.0f);
                        if (NdotL > 0.0f) {
                            float e[3];
                            cube_sample(c.env, 0, L.x, L.y, L.z, e);
                            for (int k = 0; k < 3; k++) col[k] += e[k] * NdotL;
                            total += NdotL;
                        }
                    }
                    for (int k = 0; k < 3; k++) col[k] /= total;
                    c.prefiltered.level[l][((size_t)face * res + y) * res + x] = pack_r11g11b10(col[0], col[1], col[2]);
                }
    }
}

// ---- precompute steps the reference does not have (it is handed six decoded faces and ships its BRDF table as a
// PNG): equirectangular panorama -> cube faces, and the split-sum BRDF table itself. Defined here so that the HIP
// versions have something to be compared with; the table is pinned by the reference's textures/ibl_brdf_lut.png.

// atan2 from + - * / and fma only (a 6-term odd polynomial on [0,1], |error| < 2e-5 rad), so that both
// implementations produce the same bits; the library atan2f of the two sides do not.
static inline float det_atan2(float y, float x) {
    const float ax = std::fabs(x), ay = std::fabs(y);
    const float mx = std::max(ax, ay), mn = std::min(ax, ay);
    const float a = mx > 0.0f ? mn / mx : 0.0f;
    const float s = a * a;
    float r = std::fmaf(s, -0.01172120f, 0.05265332f);
    r = std::fmaf(s, r, -0.11643287f);
    r = std::fmaf(s, r, 0.19354346f);
    r = std::fmaf(s, r, -0.33262347f);
    r = std::fmaf(s, r, 0.99997726f);
    r = r * a;
    if (ay > ax) r = 1.57079637f - r;
    if (x < 0.0f) r = 3.14159274f - r;
    return y < 0.0f ? -r : r;
}

// Direction through the centre of texel (i, j) of a cube face: the inverse of the Vulkan face-selection tables.
static inline V3 face_texel_direction(int face, int i, int j, int res) {
    const float sc = ((float)(2 * i + 1)) / (float)res - 1.0f, tc = ((float)(2 * j + 1)) / (float)res - 1.0f;
    switch (face) {
        case 0: return {1.0f, -tc, -sc};
        case 1: return {-1.0f, -tc, sc};
        case 2: return {sc, 1.0f, tc};
        case 3: return {sc, -1.0f, -tc};
        case 4: return {sc, -tc, 1.0f};
        default: return {-sc, -tc, -1.0f};
    }
}

// Panorama convention: u = atan2(x, -z) / 2pi + 1/2, v = angle from +Y / pi; bilinear, u wraps, v clamps.
void equirect_to_cube(Ctx& c, const float* pano, uint32_t w, uint32_t h, uint32_t res) {
    c.env.res = res; c.env.levels = 1;
    c.env.level.assign(1, std::vector<uint32_t>((size_t)6 * res * res));
    const int W = (int)w, H = (int)h;
#pragma omp parallel for schedule(static) collapse(2)
    for (int face = 0; face < 6; face++)
        for (int j = 0; j < (int)res; j++)
            for (int i = 0; i < (int)res; i++) {
                const V3 d = face_texel_direction(face, i, j, (int)res);
                const float u = det_atan2(d.x, -d.z) * 0.159154937f + 0.5f;
                const float v = det_atan2(std::sqrt(d.x * d.x + d.z * d.z), d.y) * 0.318309873f;
                const float tu = u * (float)W - 0.5f, tv = v * (float)H - 0.5f;
                const float fu = std::floor(tu), fv = std::floor(tv);
                const float a = tu - fu, b = tv - fv;
                int i0 = (int)fu % W; if (i0 < 0) i0 += W;
                const int i1 = (i0 + 1) % W;
                const int j0 = std::min(std::max((int)fv, 0), H - 1), j1 = std::min(std::max((int)fv + 1, 0), H - 1);
                float o[3];
                for (int k = 0; k < 3; k++) {
                    const float c00 = pano[((size_t)j0 * W + i0) * 4 + k], c10 = pano[((size_t)j0 * W + i1) * 4 + k];
                    const float c01 = pano[((size_t)j1 * W + i0) * 4 + k], c11 = pano[((size_t)j1 * W + i1) * 4 + k];
                    const float top = (1.0f - a) * c00 + a * c10, bot = (1.0f - a) * c01 + a * c11;
                    o[k] = (1.0f - b) * top + b * bot;
                }
                c.env.level[0][((size_t)face * res + j) * res + i] = pack_r11g11b10(o[0], o[1], o[2]);
            }
}

// Split-sum BRDF table (Karis 2013) on the Hammersley / GGX sampling helpers of envbakespecular.frag:17-53.
// Texel (x, y): NdotV = (x + .5) / size, roughness = 1 - (y + .5) / size (row 0 = roughness 1, the orientation of
// the reference's textures/ibl_brdf_lut.png as stored on disk). R = scale, G = bias of F0.
void generate_brdf_lut(Ctx& c, uint32_t size, uint32_t samples) {
    c.lut_w = c.lut_h = size;
    c.lut.assign((size_t)size * size * 4, 0);
    std::vector<float> cs(samples), sn(samples);
    for (uint32_t s = 0; s < samples; s++) {
        const float phi = 2.0f * 3.14159265359f * ((float)s / (float)samples);
        cs[s] = std::cos(phi); sn[s] = std::sin(phi);
    }
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < (int)size; y++)
        for (int x = 0; x < (int)size; x++) {
            const float nv = ((float)x + 0.5f) / (float)size, rough = 1.0f - ((float)y + 0.5f) / (float)size;
            const float a = rough * rough, k = a * 0.5f;
            const float vx = std::sqrt(1.0f - nv * nv), vz = nv;
            float A = 0.0f, Bc = 0.0f;
            for (uint32_t s = 0; s < samples; s++) {
                const float xi = radical_inverse(s);
                const float ct = std::sqrt((1.0f - xi) / (1.0f + (a * a - 1.0f) * xi));
                const float st = std::sqrt(1.0f - ct * ct);
                const float hx = cs[s] * st, hz = ct;
                float vdh = vx * hx + vz * hz;
                const float lz = 2.0f * vdh * hz - vz;
                const float ndl = std::min(std::max(lz, 0.0f), 1.0f), ndh = std::min(std::max(hz, 0.0f), 1.0f);
                vdh = std::min(std::max(vdh, 0.0f), 1.0f);
                if (ndl > 0.0f) {
                    const float g = (ndl / (ndl * (1.0f - k) + k)) * (nv / (nv * (1.0f - k) + k));
                    const float gv = g * vdh / std::max(ndh * nv, 1e-8f);
                    const float om = 1.0f - vdh, om2 = om * om, fc = om2 * om2 * om;
                    A += (1.0f - fc) * gv;
                    Bc += fc * gv;
                }
            }
            uint8_t* px = &c.lut[((size_t)y * size + x) * 4];
            px[0] = float_to_unorm8(A / (float)samples);
            px[1] = float_to_unorm8(Bc / (float)samples);
            px[2] = 0; px[3] = 255;
        }
}

}  // namespace orc
