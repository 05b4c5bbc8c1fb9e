// TEST INFRASTRUCTURE — CPU restatement ("oracle"); see orc_types.h.
//
// C entry points `orc_*` with the same signatures as the product's `rdr_*`
// (include/rdr.h), plus the host side of the frame: camera matrices, frustum
// culling and draw ordering, shadow-view set-up, exposure.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <new>

#include "orc_types.h"

using namespace orc;

struct rdr_ctx : Ctx {};

// ---- Camera (Renderer/Camera.cpp:136-139, 196-216) ------------------------
M4 Camera::view() const {
    M4 r = mul(mul(mul(rotate(-euler.x, v3(1, 0, 0)), rotate(-euler.y, v3(0, 1, 0))), rotate(-euler.z, v3(0, 0, 1))),
               translate(v3(-position.x, -position.y, -position.z)));
    return r;
}
M4 Camera::proj() const {
    const M4 flip = scale(v3(1.0f, -1.0f, 1.0f));
    if (type == RDR_CAMERA_PERSPECTIVE) return mul(flip, perspective(fov_or_height, aspect, near_plane, far_plane));
    return mul(flip, ortho(-fov_or_height * aspect / 2, fov_or_height * aspect / 2, -fov_or_height, fov_or_height, near_plane, far_plane));
}
M4 Camera::viewProj() const { return mul(mul(crop, proj()), view()); }

namespace {

int fail(Ctx* c, int code, const std::string& msg) { if (c) c->err = msg; return code; }

// Renderer/Camera.cpp:245-280 (Gribb-Hartmann), planes pointing inside
void frustum_planes(const M4& M, V4 planes[6]) {
    const float (*m)[4] = M.c;
    planes[0] = {m[0][3] + m[0][0], m[1][3] + m[1][0], m[2][3] + m[2][0], m[3][3] + m[3][0]};
    planes[1] = {m[0][3] - m[0][0], m[1][3] - m[1][0], m[2][3] - m[2][0], m[3][3] - m[3][0]};
    planes[2] = {m[0][3] + m[0][1], m[1][3] + m[1][1], m[2][3] + m[2][1], m[3][3] + m[3][1]};
    planes[3] = {m[0][3] - m[0][1], m[1][3] - m[1][1], m[2][3] - m[2][1], m[3][3] - m[3][1]};
    planes[4] = {m[0][3] - m[0][2], m[1][3] - m[1][2], m[2][3] - m[2][2], m[3][3] - m[3][2]};
    planes[5] = {m[0][2], m[1][2], m[2][2], m[3][2]};
}
// Renderer/VulkanRenderer.cpp:929-943
bool should_cull(const Primitive& p, const M4& model, const M4& viewproj) {
    V4 planes[6];
    frustum_planes(mul(viewproj, model), planes);
    for (int i = 0; i < 6; i++) {
        V3 n = p.bb_min;
        if (planes[i].x >= 0.0f) n.x = p.bb_max.x;
        if (planes[i].y >= 0.0f) n.y = p.bb_max.y;
        if (planes[i].z >= 0.0f) n.z = p.bb_max.z;
        if (dot(n, v3(planes[i].x, planes[i].y, planes[i].z)) + planes[i].w < 0.0f) return true;
    }
    return false;
}

enum SortMode { SORT_NONE, SORT_FRONT_TO_BACK, SORT_BACK_TO_FRONT };

// Renderer/VulkanRenderer.cpp:945-999 / VulkanRendererShadow.cpp:305-331,423-451:
// order by distance(barycenter, eye), then frustum-cull, then one draw per primitive.
std::vector<Draw> build_draws(const Ctx& c, const std::vector<uint32_t>& prims, const M4& viewproj, V3 eye, SortMode mode) {
    std::vector<uint32_t> order(prims);
    if (mode != SORT_NONE) {
        std::vector<float> key(c.prims.size());
        for (uint32_t p : order) key[p] = distance(c.prims[p].barycenter, eye);
        if (mode == SORT_FRONT_TO_BACK) std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return key[a] < key[b]; });
        else std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return key[a] > key[b]; });
    }
    std::vector<Draw> draws;
    uint32_t first = 0;
    for (uint32_t p : order) {
        const Primitive& pr = c.prims[p];
        const M4& model = c.meshes[pr.mesh].model;
        if (should_cull(pr, model, viewproj)) continue;
        Draw d;
        d.prim = p; d.model = model; d.mvp = mul(viewproj, model); d.first_tri = first;
        first += (uint32_t)(pr.idx.size() / 3);
        draws.push_back(d);
    }
    return draws;
}

// glm::quatLookAt (RH) + quat_cast, for the six constant cube-face rotations
// of Renderer/VulkanRendererShadow.cpp:260-267
Quat quat_look_at(V3 direction, V3 up) {
    V3 c2 = direction * -1.0f;
    V3 right = cross(up, c2);
    V3 c0 = right * (1.0f / std::sqrt(std::max(0.00001f, dot(right, right))));
    V3 c1 = cross(c2, c0);
    const float m00 = c0.x, m11 = c1.y, m22 = c2.z;
    const float fx = m00 - m11 - m22, fy = m11 - m00 - m22, fz = m22 - m00 - m11, fw = m00 + m11 + m22;
    int big = 0; float fb = fw;
    if (fx > fb) { fb = fx; big = 1; }
    if (fy > fb) { fb = fy; big = 2; }
    if (fz > fb) { fb = fz; big = 3; }
    const float bv = std::sqrt(fb + 1.0f) * 0.5f, mult = 0.25f / bv;
    // m[col][row]: c0 = column 0 etc.
    const float m01 = c0.y, m02 = c0.z, m10 = c1.x, m12 = c1.z, m20 = c2.x, m21 = c2.y;
    switch (big) {
        case 0: return {bv, (m12 - m21) * mult, (m20 - m02) * mult, (m01 - m10) * mult};
        case 1: return {(m12 - m21) * mult, bv, (m01 + m10) * mult, (m20 + m02) * mult};
        case 2: return {(m20 - m02) * mult, (m01 + m10) * mult, bv, (m12 + m21) * mult};
        default: return {(m01 - m10) * mult, (m20 + m02) * mult, (m12 + m21) * mult, bv};
    }
}

// Renderer/VulkanRendererShadow.cpp:152-175
void compute_cascade_splits(Ctx& c) {
    c.cascade_depths.clear(); c.cascade_camera_depths.clear();
    c.cascade_depths.push_back(c.camera.near_plane);
    c.cascade_camera_depths.push_back(0.0f);
    const float n = c.camera.near_plane, f = c.camera.far_plane;
    const int levels = (int)c.settings.directional_shadow_cascade_levels;
    for (int i = 1; i < levels; i++) {
        const float N = (float)levels;
        const float Cuni = n + (f - n) * i / N;
        const float Clog = n * std::pow(f / n, i / N);
        const float lambda = 0.9f;
        const float d = lambda * Clog + (1 - lambda) * Cuni;
        c.cascade_depths.push_back(d);
        c.cascade_camera_depths.push_back((f - f * n / d) / (f - n));
    }
}

// Renderer/Camera.cpp:218-242
void frustum_vertices(const M4& viewproj, V3 out[8]) {
    static const float ndc[8][3] = {{-1, -1, 0}, {1, -1, 0}, {-1, 1, 0}, {1, 1, 0}, {-1, -1, 1}, {1, -1, 1}, {-1, 1, 1}, {1, 1, 1}};
    const M4 inv = inverse(viewproj);
    for (int i = 0; i < 8; i++) {
        V4 vh = mul(inv, V4{ndc[i][0], ndc[i][1], ndc[i][2], 1.0f});
        out[i] = v3(vh.x / vh.w, vh.y / vh.w, vh.z / vh.w);
    }
}

void clear_depth(float* p, size_t n) { std::fill(p, p + n, 1.0f); }

void render_point_shadows(Ctx& c, V3 camera_pos) {
    const uint32_t count = c.settings.point_shadow_map_count;
    std::vector<uint32_t> casters;
    for (uint32_t i = 0; i < c.lights.size(); i++)
        if (c.lights[i].flags & RDR_POINT_LIGHT_CAST_SHADOWS) casters.push_back(i);
    if (casters.empty() || count == 0) return;
    const uint32_t res = c.settings.point_shadow_map_resolution;
    const size_t face_px = (size_t)res * res;
    if (c.point_shadow.size() != face_px * 6 * count) {
        c.point_shadow.assign(face_px * 6 * count, 1.0f);
        c.point_shadow_static.assign(face_px * 6 * count, 1.0f);
    }
    static const V3 dirs[6] = {{1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
    static const V3 ups[6] = {{0, 1, 0}, {0, 1, 0}, {0, 0, -1}, {0, 0, 1}, {0, 1, 0}, {0, 1, 0}};
    Camera face[6];
    for (int j = 0; j < 6; j++) {
        face[j].type = RDR_CAMERA_PERSPECTIVE;
        face[j].euler = eulerAngles(quat_look_at(dirs[j], ups[j]));
        face[j].near_plane = 0.1f; face[j].far_plane = 100.0f;
        face[j].fov_or_height = 90.0f * 0.01745329251994329576923690768489f;  // glm::radians(90.0f)
        face[j].aspect = 1.0f;
        face[j].crop = scale(v3(-1.0f, 1.0f, 1.0f));
    }
    // slot assignment (VulkanRendererShadow.cpp:278-296)
    std::vector<float> key(c.lights.size());
    for (uint32_t i : casters) key[i] = distance(c.lights[i].point, camera_pos);
    std::stable_sort(casters.begin(), casters.end(), [&](uint32_t a, uint32_t b) { return key[a] < key[b]; });
    std::vector<uint32_t> slotted;
    uint16_t slot = 0;
    for (uint32_t i : casters) {
        PointLight& l = c.lights[i];
        if (slot < count) {
            slotted.push_back(i);
            if (l.shadow_map_index != slot) l.flags &= ~RDR_POINT_LIGHT_STATIC_SHADOW_MAP_RENDERED;
            l.shadow_map_index = slot;
            l.flags |= RDR_POINT_LIGHT_HAS_SHADOW_MAP;
            slot++;
        } else {
            l.shadow_map_index = 0xFFFF;
            l.flags &= ~RDR_POINT_LIGHT_STATIC_SHADOW_MAP_RENDERED;
            l.flags &= ~RDR_POINT_LIGHT_HAS_SHADOW_MAP;
        }
    }
    std::vector<uint32_t> static_prims, dynamic_prims;
    for (uint32_t p = 0; p < c.prims.size(); p++) (c.meshes[c.prims[p].mesh].is_static ? static_prims : dynamic_prims).push_back(p);
    // static maps (…Shadow.cpp:298-342)
    for (uint32_t i : slotted) {
        PointLight& l = c.lights[i];
        if (l.flags & RDR_POINT_LIGHT_STATIC_SHADOW_MAP_RENDERED) continue;
        for (int j = 0; j < 6; j++) {
            face[j].position = l.point;
            const M4 vp = face[j].viewProj();
            float* dst = c.point_shadow_static.data() + ((size_t)l.shadow_map_index * 6 + j) * face_px;
            clear_depth(dst, face_px);
            std::vector<Draw> draws = build_draws(c, static_prims, vp, l.point, SORT_FRONT_TO_BACK);
            RasterTarget rt{res, res, 1, dst, nullptr};
            raster_draws(c, draws, rt, CULL_FRONT, &c.stats.triangle_instances, &c.stats.triangles_binned);
            c.stats.views++;
        }
        l.flags |= RDR_POINT_LIGHT_STATIC_SHADOW_MAP_RENDERED;
        std::memcpy(c.point_shadow.data() + (size_t)l.shadow_map_index * 6 * face_px,
                    c.point_shadow_static.data() + (size_t)l.shadow_map_index * 6 * face_px, 6 * face_px * sizeof(float));
    }
    // dynamic meshes on top of a copy of the static map (…Shadow.cpp:344-388)
    for (uint32_t i : slotted) {
        PointLight& l = c.lights[i];
        for (int j = 0; j < 6; j++) {
            face[j].position = l.point;
            const M4 vp = face[j].viewProj();
            std::vector<Draw> draws = build_draws(c, dynamic_prims, vp, l.point, SORT_FRONT_TO_BACK);
            if (c.settings.dynamic_shadows_enabled && !draws.empty()) {
                float* dst = c.point_shadow.data() + ((size_t)l.shadow_map_index * 6 + j) * face_px;
                std::memcpy(dst, c.point_shadow_static.data() + ((size_t)l.shadow_map_index * 6 + j) * face_px, face_px * sizeof(float));
                RasterTarget rt{res, res, 1, dst, nullptr};
                raster_draws(c, draws, rt, CULL_FRONT, &c.stats.triangle_instances, &c.stats.triangles_binned);
                c.stats.views++;
            }
        }
    }
}

// Renderer/VulkanRendererShadow.cpp:391-456
void render_directional_shadows(Ctx& c) {
    const uint32_t levels = c.settings.directional_shadow_cascade_levels;
    const uint32_t res = c.settings.directional_shadow_map_resolution;
    const size_t px = (size_t)res * res;
    c.dir_shadow.resize(px * levels);
    Camera light;  // DirectionalLight::lightViewCamera (DirectionalLight.cpp:3-5)
    light.type = RDR_CAMERA_ORTHOGRAPHIC;
    light.position = v3(c.dir_light.position[0], c.dir_light.position[1], c.dir_light.position[2]);
    light.euler = eulerAngles(Quat{c.dir_light.orientation_wxyz[0], c.dir_light.orientation_wxyz[1], c.dir_light.orientation_wxyz[2], c.dir_light.orientation_wxyz[3]});
    light.near_plane = 0.1f; light.far_plane = 200.0f; light.fov_or_height = 10.0f; light.aspect = 1.0f;
    const M4 light_vp = light.viewProj();
    std::vector<uint32_t> all(c.prims.size());
    for (uint32_t p = 0; p < all.size(); p++) all[p] = p;
    for (uint32_t i = 0; i < levels; i++) {
        Camera split = c.camera;  // Camera's copy does not carry the crop matrix (Camera.cpp:3-41)
        split.crop = identity();
        split.near_plane = c.cascade_depths[i];
        if (i < levels - 1) split.far_plane = c.cascade_depths[i + 1];
        V3 corners[8];
        frustum_vertices(split.viewProj(), corners);
        // Camera::makeAABBFromVertices in light clip space (Camera.cpp:306-329)
        float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int k = 0; k < 8; k++) {
            V4 h = mul(light_vp, V4{corners[k].x, corners[k].y, corners[k].z, 1.0f});
            const float v[3] = {h.x / h.w, h.y / h.w, h.z / h.w};
            for (int a = 0; a < 3; a++) { if (v[a] < mn[a]) mn[a] = v[a]; if (v[a] > mx[a]) mx[a] = v[a]; }
        }
        const float largest = std::max(mx[0] - mn[0], mx[1] - mn[1]);
        const V3 offset = v3(-((mx[0] + mn[0]) / 2), -((mx[1] + mn[1]) / 2), -0.0f);
        Camera split_pov = light;
        split_pov.crop = mul(scale(v3(1.0f / largest, 1.0f / largest, 1.0f)), translate(offset));
        const M4 vp = split_pov.viewProj();
        c.csm_viewproj[i] = vp;
        float* dst = c.dir_shadow.data() + (size_t)i * px;
        clear_depth(dst, px);
        std::vector<Draw> draws = build_draws(c, all, vp, split_pov.position, SORT_FRONT_TO_BACK);
        RasterTarget rt{res, res, 1, dst, nullptr};
        raster_draws(c, draws, rt, CULL_FRONT, &c.stats.triangle_instances, &c.stats.triangles_binned);
        c.stats.views++;
    }
}

float read_as_float(const uint8_t* p, uint32_t value_type, bool normalized) {
    switch (value_type) {
        case RDR_FLOAT: { float f; std::memcpy(&f, p, 4); return f; }
        case RDR_UINT8: return normalized ? (float)p[0] / 255.0f : (float)p[0];
        case RDR_UINT16: { uint16_t v; std::memcpy(&v, p, 2); return normalized ? (float)v / 65535.0f : (float)v; }
        case RDR_HALF: { uint16_t v; std::memcpy(&v, p, 2); return half_to_float(v); }
        default: return 0.0f;
    }
}
size_t value_size(uint32_t t) {
    switch (t) {
        case RDR_INT8: case RDR_UINT8: return 1;
        case RDR_INT16: case RDR_UINT16: case RDR_HALF: return 2;
        case RDR_INT32: case RDR_UINT32: case RDR_FLOAT: return 4;
        default: return 8;
    }
}

}  // namespace

extern "C" {

int orc_settings_default(rdr_settings* s) {
    if (!s) return RDR_ERR_INVALID;
    *s = rdr_settings{0, 1, 512, 8, 2048, 4, 0, 0.0f, 2.2f, 4};
    return RDR_OK;
}

const char* orc_backend_name(void) { return "cpu:oracle"; }

int orc_create(const rdr_settings* settings, uint32_t width, uint32_t height, int, rdr_ctx** out) {
    if (!settings || !out || width == 0 || height == 0) return RDR_ERR_INVALID;
    if ((settings->msaa != 1 && settings->msaa != 4) || settings->directional_shadow_cascade_levels < 1 || settings->directional_shadow_cascade_levels > 4)
        return RDR_ERR_UNSUPPORTED;
    // what the implementation does not offer is refused, not ignored: shadow maps whose tile boxes would not fit 16-bit
    // coordinates, more cube slots than a frame can reasonably hold
    if (settings->hdr_output_enabled > 1) return RDR_ERR_INVALID;
    if (settings->directional_shadow_map_resolution == 0 || settings->point_shadow_map_resolution == 0) return RDR_ERR_INVALID;
    if (settings->directional_shadow_map_resolution > 16384 || settings->point_shadow_map_resolution > 8192 || settings->point_shadow_map_count > 64)
        return RDR_ERR_UNSUPPORTED;
    rdr_ctx* c = new (std::nothrow) rdr_ctx();
    if (!c) return RDR_ERR_DEVICE;
    c->settings = *settings; c->W = width; c->H = height;
    c->camera.type = RDR_CAMERA_PERSPECTIVE;  // VulkanRenderer.h:117, VulkanRenderer.cpp:50
    c->camera.near_plane = 0.1f; c->camera.far_plane = 60.0f;
    c->camera.fov_or_height = 40.0f * 0.01745329251994329576923690768489f;
    c->camera.aspect = (float)width / (float)height;
    // default textures: 1x1 white (VulkanRenderer.h:28-32) and clear_normal.png = (128,128,255,0)
    const uint8_t white[4] = {255, 255, 255, 255}, nrm[4] = {128, 128, 255, 0};
    for (int k = 0; k < 2; k++) {
        Image img; img.w = img.h = 1; img.srgb = false;
        img.mips.push_back(std::vector<uint8_t>(k == 0 ? white : nrm, (k == 0 ? white : nrm) + 4));
        c->images.push_back(img);
        c->textures.push_back(Texture{(uint32_t)c->images.size() - 1, rdr_sampler{1, 1, 1, 0, 0}});
    }
    c->tex_white = 0; c->tex_normal = 1;
    *out = c;
    return RDR_OK;
}
void orc_destroy(rdr_ctx* c) { delete c; }
const char* orc_last_error(const rdr_ctx* c) { return c ? c->err.c_str() : "null context"; }

int orc_load_buffer(rdr_ctx* c, const void* ptr, size_t size, uint32_t* out_id) {
    if (!c || !ptr || !out_id) return fail(c, RDR_ERR_INVALID, "load_buffer: null argument");
    c->buffers.emplace_back((const uint8_t*)ptr, (const uint8_t*)ptr + size);
    *out_id = (uint32_t)c->buffers.size() - 1;
    return RDR_OK;
}

int orc_load_image(rdr_ctx* c, const void* ptr, size_t size, uint32_t w, uint32_t h, int format, uint32_t max_mips, uint32_t* out_id) {
    if (!c || !ptr || !out_id || w == 0 || h == 0) return fail(c, RDR_ERR_INVALID, "load_image: bad argument");
    int ch;
    bool srgb = false;
    int float_bytes = 0;  // bytes per component of an SFLOAT format (Renderer/Image.h:9-12), 0 for the 8-bit ones
    switch (format) {
        case RDR_FORMAT_R8G8B8_UNORM: ch = 3; break;
        case RDR_FORMAT_R8G8B8_SRGB: ch = 3; srgb = true; break;
        case RDR_FORMAT_R8G8B8A8_UNORM: ch = 4; break;
        case RDR_FORMAT_R8G8B8A8_SRGB: ch = 4; srgb = true; break;
        case RDR_FORMAT_R16G16B16_SFLOAT: ch = 3; float_bytes = 2; break;
        case RDR_FORMAT_R16G16B16A16_SFLOAT: ch = 4; float_bytes = 2; break;
        case RDR_FORMAT_R32G32B32_SFLOAT: ch = 3; float_bytes = 4; break;
        case RDR_FORMAT_R32G32B32A32_SFLOAT: ch = 4; float_bytes = 4; break;
        default: return fail(c, RDR_ERR_INVALID, "load_image: unknown format");
    }
    if (size < (size_t)w * h * ch * (float_bytes ? float_bytes : 1)) return fail(c, RDR_ERR_INVALID, "load_image: size too small");
    if (w > 32768u || h > 32768u) return fail(c, RDR_ERR_UNSUPPORTED, "load_image: images larger than 32768 texels on a side are not supported");
    Image img; img.w = w; img.h = h; img.srgb = srgb;
    const uint8_t* s = (const uint8_t*)ptr;
    if (float_bytes) {
        // one float[4] per texel; a component the format lacks reads as 1 (alpha of the three-component formats)
        img.is_float = true; img.half_texels = float_bytes == 2;
        img.mips.emplace_back((size_t)w * h * 16);
        float* f = reinterpret_cast<float*>(img.mips[0].data());
        for (size_t i = 0; i < (size_t)w * h; i++)
            for (int k = 0; k < 4; k++) {
                float v = 1.0f;
                if (k < ch) {
                    if (float_bytes == 2) { uint16_t hb; std::memcpy(&hb, s + (i * ch + k) * 2, 2); v = half_to_float(hb); }
                    else std::memcpy(&v, s + (i * ch + k) * 4, 4);
                }
                f[i * 4 + k] = v;
            }
    } else {
        img.mips.emplace_back((size_t)w * h * 4);
        for (size_t i = 0; i < (size_t)w * h; i++) {
            for (int k = 0; k < 3; k++) img.mips[0][i * 4 + k] = s[i * ch + k];
            img.mips[0][i * 4 + 3] = ch == 4 ? s[i * ch + 3] : 255;
        }
    }
    build_mips(*c, img, max_mips);
    c->images.push_back(std::move(img));
    *out_id = (uint32_t)c->images.size() - 1;
    return RDR_OK;
}

int orc_make_texture(rdr_ctx* c, uint32_t image, const rdr_sampler* s, uint32_t* out_id) {
    if (!c || !s || !out_id || image >= c->images.size()) return fail(c, RDR_ERR_INVALID, "make_texture: bad argument");
    if (s->mag_filter > RDR_FILTER_LINEAR || s->min_filter > RDR_FILTER_LINEAR || s->mipmap_filter > RDR_FILTER_LINEAR)
        return fail(c, RDR_ERR_INVALID, "make_texture: bad filter");
    if (s->wrap_u > RDR_WRAP_CLAMP || s->wrap_v > RDR_WRAP_CLAMP) return fail(c, RDR_ERR_INVALID, "make_texture: bad wrap mode");
    // VulkanRenderer.cpp:370-371: the V address mode is REPEAT if wrapV says so; otherwise it is chosen by wrapU (the
    // reference tests wrapU for "mirror" in both lines), so (REPEAT, MIRROR) samples V clamped and (MIRROR, CLAMP)
    // samples V mirrored. Kept: it is what the reference's sampler does.
    rdr_sampler eff = *s;
    if (eff.wrap_v != RDR_WRAP_REPEAT) eff.wrap_v = eff.wrap_u == RDR_WRAP_MIRROR ? RDR_WRAP_MIRROR : RDR_WRAP_CLAMP;
    c->textures.push_back(Texture{image, eff});
    *out_id = (uint32_t)c->textures.size() - 1;
    return RDR_OK;
}

int orc_add_mesh(rdr_ctx* c, const rdr_primitive_desc* prims, uint32_t count, const float model[16], int is_static, uint32_t* out_mesh_id) {
    if (!c || !prims || !model || count == 0) return fail(c, RDR_ERR_INVALID, "add_mesh: bad argument");
    Mesh mesh;
    std::memcpy(mesh.model.c, model, 64);
    mesh.is_static = is_static != 0;
    mesh.first_prim = (uint32_t)c->prims.size(); mesh.prim_count = count;
    const uint32_t mesh_id = (uint32_t)c->meshes.size();
    std::vector<Primitive> built;
    for (uint32_t pi = 0; pi < count; pi++) {
        const rdr_primitive_desc& d = prims[pi];
        if (d.mode != 4) return fail(c, RDR_ERR_UNSUPPORTED, "add_mesh: only triangle lists are supported");
        const rdr_vertex_attribute* at[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        for (uint32_t a = 0; a < d.attribute_count; a++) {
            const rdr_vertex_attribute& va = d.attributes[a];
            if (va.name > RDR_ATTR_WEIGHTS_0) continue;
            if (va.buffer >= c->buffers.size()) return fail(c, RDR_ERR_INVALID, "add_mesh: bad buffer id");
            at[va.name] = &va;
        }
        if (!at[RDR_ATTR_POSITION]) return fail(c, RDR_ERR_INVALID, "add_mesh: POSITION missing");
        const size_t nv = (size_t)at[RDR_ATTR_POSITION]->count;
        Primitive p;
        p.pos.resize(nv); p.attr.assign(nv, VertexAttr{v3(0, 0, 0), v3(0, 0, 0), v3(0, 0, 0), 0.0f, 0.0f});
        auto fetch = [&](const rdr_vertex_attribute* va, size_t i, int comps, float* o) -> bool {
            const size_t vs = value_size(va->value_type);
            const int have = (int)va->container + 1;
            const size_t stride = va->stride ? (size_t)va->stride : vs * have;
            const size_t off = (size_t)va->offset + i * stride;
            const std::vector<uint8_t>& buf = c->buffers[va->buffer];
            if (off + vs * std::min(comps, have) > buf.size()) return false;
            const bool norm = (va->name == RDR_ATTR_TEXCOORD_0 || va->name == RDR_ATTR_WEIGHTS_0) && va->value_type != RDR_FLOAT;
            for (int k = 0; k < comps; k++) o[k] = k < have ? read_as_float(&buf[off + k * vs], va->value_type, norm) : 0.0f;
            return true;
        };
        for (size_t i = 0; i < nv; i++) {
            float v[4];
            if (!fetch(at[RDR_ATTR_POSITION], i, 3, v)) return fail(c, RDR_ERR_INVALID, "add_mesh: POSITION out of range");
            p.pos[i] = v3(v[0], v[1], v[2]);
            if (at[RDR_ATTR_NORMAL]) { if (!fetch(at[RDR_ATTR_NORMAL], i, 3, v)) return fail(c, RDR_ERR_INVALID, "add_mesh: NORMAL out of range"); p.attr[i].normal = v3(v[0], v[1], v[2]); }
            if (at[RDR_ATTR_TEXCOORD_0]) { if (!fetch(at[RDR_ATTR_TEXCOORD_0], i, 2, v)) return fail(c, RDR_ERR_INVALID, "add_mesh: TEXCOORD_0 out of range"); p.attr[i].u = v[0]; p.attr[i].v = v[1]; }
            if (at[RDR_ATTR_TANGENT]) {
                if (!fetch(at[RDR_ATTR_TANGENT], i, 4, v)) return fail(c, RDR_ERR_INVALID, "add_mesh: TANGENT out of range");
                p.attr[i].tangent = v3(v[0], v[1], v[2]);
                if (at[RDR_ATTR_BITANGENT]) {
                    float b[4];
                    if (!fetch(at[RDR_ATTR_BITANGENT], i, 3, b)) return fail(c, RDR_ERR_INVALID, "add_mesh: BITANGENT out of range");
                    p.attr[i].bitangent = v3(b[0], b[1], b[2]);
                } else {
                    // main.cpp:93: bitangent = cross(normal, tangent.xyz) * tangent.w
                    p.attr[i].bitangent = cross(p.attr[i].normal, p.attr[i].tangent) * v[3];
                }
            }
        }
        if (d.indices.buffer != RDR_INVALID_ID) {
            if (d.indices.buffer >= c->buffers.size()) return fail(c, RDR_ERR_INVALID, "add_mesh: bad index buffer id");
            const std::vector<uint8_t>& buf = c->buffers[d.indices.buffer];
            const size_t is = value_size(d.indices.index_type);
            if ((size_t)d.indices.offset + is * d.indices.count > buf.size()) return fail(c, RDR_ERR_INVALID, "add_mesh: indices out of range");
            p.idx.resize((size_t)d.indices.count / 3 * 3);
            for (size_t i = 0; i < p.idx.size(); i++) {
                const uint8_t* q = &buf[(size_t)d.indices.offset + i * is];
                uint32_t v = 0;
                if (d.indices.index_type == RDR_UINT8) v = q[0];
                else if (d.indices.index_type == RDR_UINT16) { uint16_t t; std::memcpy(&t, q, 2); v = t; }
                else if (d.indices.index_type == RDR_UINT32) std::memcpy(&v, q, 4);
                else return fail(c, RDR_ERR_UNSUPPORTED, "add_mesh: index type");
                if (v >= nv) return fail(c, RDR_ERR_INVALID, "add_mesh: index beyond vertex count");
                p.idx[i] = v;
            }
        } else {
            p.idx.resize(nv / 3 * 3);
            for (size_t i = 0; i < p.idx.size(); i++) p.idx[i] = (uint32_t)i;
        }
        if (!at[RDR_ATTR_TANGENT]) {
            // MeshPrimitive::calculateTangents, indexed form (Mesh.cpp:5-52)
            for (size_t i = 0; i + 2 < p.idx.size(); i += 3) {
                const uint32_t a = p.idx[i], b = p.idx[i + 1], cc = p.idx[i + 2];
                const V3 e1 = p.pos[b] - p.pos[a], e2 = p.pos[cc] - p.pos[a];
                const float s1 = p.attr[b].u - p.attr[a].u, s2 = p.attr[cc].u - p.attr[a].u;
                const float t1 = p.attr[b].v - p.attr[a].v, t2 = p.attr[cc].v - p.attr[a].v;
                const float f = 1.0f / (s1 * t2 - s2 * t1);
                const V3 sdir = v3((t2 * e1.x - t1 * e2.x) * f, (t2 * e1.y - t1 * e2.y) * f, (t2 * e1.z - t1 * e2.z) * f);
                for (uint32_t vtx : {a, b, cc}) {
                    p.attr[vtx].tangent = sdir;
                    p.attr[vtx].bitangent = cross(p.attr[vtx].normal, sdir);
                }
            }
        }
        p.bb_min = v3(d.bb_min[0], d.bb_min[1], d.bb_min[2]);
        p.bb_max = v3(d.bb_max[0], d.bb_max[1], d.bb_max[2]);
        // calculateBarycenter (Mesh.cpp:77-86)
        V3 bc = v3(0, 0, 0);
        for (size_t i = 0; i < nv; i++) bc = bc + p.pos[i];
        bc = v3(bc.x / (float)nv, bc.y / (float)nv, bc.z / (float)nv);
        V4 bch = mul(mesh.model, V4{bc.x, bc.y, bc.z, 1.0f});
        p.barycenter = v3(bch.x / bch.w, bch.y / bch.w, bch.z / bch.w);
        p.mat = d.material;
        const uint32_t tex_ids[5] = {p.mat.albedo_texture, p.mat.normal_texture, p.mat.metal_roughness_texture, p.mat.ao_texture, p.mat.emissive_texture};
        for (uint32_t t : tex_ids)
            if (t != RDR_INVALID_ID && t >= c->textures.size()) return fail(c, RDR_ERR_INVALID, "add_mesh: bad texture id");
        p.mesh = mesh_id;
        if (at[RDR_ATTR_JOINTS_0] && at[RDR_ATTR_WEIGHTS_0]) {
            const uint32_t jt = at[RDR_ATTR_JOINTS_0]->value_type;
            if (jt != RDR_UINT8 && jt != RDR_UINT16) return fail(c, RDR_ERR_UNSUPPORTED, "add_mesh: JOINTS_0 must be uint8 or uint16");
            if (mesh.is_static) return fail(c, RDR_ERR_INVALID, "add_mesh: a skinned mesh must be added with is_static = 0");
            p.skinned = true;
            p.joints.resize(nv * 4); p.weights.resize(nv * 4);
            for (size_t i = 0; i < nv; i++) {
                float j[4], w[4];
                if (!fetch(at[RDR_ATTR_JOINTS_0], i, 4, j) || !fetch(at[RDR_ATTR_WEIGHTS_0], i, 4, w))
                    return fail(c, RDR_ERR_INVALID, "add_mesh: JOINTS_0 / WEIGHTS_0 out of range");
                for (int k = 0; k < 4; k++) {
                    p.joints[i * 4 + k] = (uint16_t)j[k]; p.weights[i * 4 + k] = w[k];
                    mesh.n_joints = std::max(mesh.n_joints, (uint32_t)p.joints[i * 4 + k] + 1);
                }
            }
            p.rest_pos = p.pos; p.rest_attr = p.attr;
            p.rest_bb_min = p.bb_min; p.rest_bb_max = p.bb_max;
        }
        built.push_back(std::move(p));
    }
    for (auto& p : built) c->prims.push_back(std::move(p));
    c->meshes.push_back(mesh);
    if (out_mesh_id) *out_mesh_id = mesh_id;
    return RDR_OK;
}

int orc_set_mesh_transform(rdr_ctx* c, uint32_t mesh_id, const float model[16]) {
    if (!c || !model || mesh_id >= c->meshes.size()) return fail(c, RDR_ERR_INVALID, "set_mesh_transform: bad argument");
    std::memcpy(c->meshes[mesh_id].model.c, model, 64);
    return RDR_OK;
}

// Linear-blend skinning, evaluated in this fixed order so that the HIP kernel can match it bit for bit:
// S = ((w0*J0 + w1*J1) + w2*J2) + w3*J3 element-wise on the three affine rows, then row . (x, y, z[, 1]).
int orc_set_mesh_joints(rdr_ctx* c, uint32_t mesh_id, const float* jm, uint32_t count) {
    if (!c || !jm || mesh_id >= c->meshes.size()) return fail(c, RDR_ERR_INVALID, "set_mesh_joints: bad argument");
    Mesh& mesh = c->meshes[mesh_id];
    if (mesh.n_joints == 0) return fail(c, RDR_ERR_INVALID, "set_mesh_joints: the mesh has no skinned primitive");
    if (count < mesh.n_joints) return fail(c, RDR_ERR_INVALID, "set_mesh_joints: fewer matrices than the largest joint index in use");
    for (uint32_t pi = mesh.first_prim; pi < mesh.first_prim + mesh.prim_count; pi++) {
        Primitive& p = c->prims[pi];
        if (!p.skinned) continue;
        const size_t nv = p.rest_pos.size();
#pragma omp parallel for schedule(static)
        for (long long i = 0; i < (long long)nv; i++) {
            const uint16_t* j = &p.joints[i * 4];
            const float* w = &p.weights[i * 4];
            float S[3][4];  // S[row][col]
            for (int r = 0; r < 3; r++)
                for (int col = 0; col < 4; col++) {
                    const int e = col * 4 + r;
                    S[r][col] = ((w[0] * jm[j[0] * 16 + e] + w[1] * jm[j[1] * 16 + e]) + w[2] * jm[j[2] * 16 + e]) + w[3] * jm[j[3] * 16 + e];
                }
            auto point = [&](V3 v) { return v3(((S[0][0] * v.x + S[0][1] * v.y) + S[0][2] * v.z) + S[0][3], ((S[1][0] * v.x + S[1][1] * v.y) + S[1][2] * v.z) + S[1][3],
                                               ((S[2][0] * v.x + S[2][1] * v.y) + S[2][2] * v.z) + S[2][3]); };
            auto dir = [&](V3 v) { return v3((S[0][0] * v.x + S[0][1] * v.y) + S[0][2] * v.z, (S[1][0] * v.x + S[1][1] * v.y) + S[1][2] * v.z,
                                             (S[2][0] * v.x + S[2][1] * v.y) + S[2][2] * v.z); };
            p.pos[i] = point(p.rest_pos[i]);
            p.attr[i].normal = dir(p.rest_attr[i].normal);
            p.attr[i].tangent = dir(p.rest_attr[i].tangent);
            p.attr[i].bitangent = dir(p.rest_attr[i].bitangent);
        }
        // culling box: around the rest box carried by every joint (a vertex is a convex combination of those images)
        V3 lo = v3(INFINITY, INFINITY, INFINITY), hi = v3(-INFINITY, -INFINITY, -INFINITY);
        for (uint32_t jn = 0; jn < mesh.n_joints; jn++) {
            const float* m = jm + (size_t)jn * 16;
            for (int k = 0; k < 8; k++) {
                const float x = (k & 1) ? p.rest_bb_max.x : p.rest_bb_min.x, y = (k & 2) ? p.rest_bb_max.y : p.rest_bb_min.y,
                            z = (k & 4) ? p.rest_bb_max.z : p.rest_bb_min.z;
                const float qx = ((m[0] * x + m[4] * y) + m[8] * z) + m[12], qy = ((m[1] * x + m[5] * y) + m[9] * z) + m[13],
                            qz = ((m[2] * x + m[6] * y) + m[10] * z) + m[14];
                lo = v3(std::min(lo.x, qx), std::min(lo.y, qy), std::min(lo.z, qz));
                hi = v3(std::max(hi.x, qx), std::max(hi.y, qy), std::max(hi.z, qz));
            }
        }
        p.bb_min = lo; p.bb_max = hi;
    }
    return RDR_OK;
}

int orc_read_primitive_positions(rdr_ctx* c, uint32_t mesh_id, uint32_t primitive, float* out, uint32_t* out_count) {
    if (!c || mesh_id >= c->meshes.size() || primitive >= c->meshes[mesh_id].prim_count) return fail(c, RDR_ERR_INVALID, "read_primitive_positions: bad id");
    const Primitive& p = c->prims[c->meshes[mesh_id].first_prim + primitive];
    if (out_count) *out_count = (uint32_t)p.pos.size();
    if (out) for (size_t i = 0; i < p.pos.size(); i++) { out[i * 3] = p.pos[i].x; out[i * 3 + 1] = p.pos[i].y; out[i * 3 + 2] = p.pos[i].z; }
    return RDR_OK;
}

int orc_set_lights(rdr_ctx* c, const rdr_point_light* lights, uint32_t count, const rdr_directional_light* dl) {
    if (!c || !dl || (count && !lights)) return fail(c, RDR_ERR_INVALID, "set_lights: bad argument");
    c->lights.clear();
    for (uint32_t i = 0; i < count; i++) {
        PointLight l;
        l.point = v3(lights[i].point[0], lights[i].point[1], lights[i].point[2]);
        l.intensity = v3(lights[i].intensity[0], lights[i].intensity[1], lights[i].intensity[2]);
        l.flags = lights[i].flags & RDR_POINT_LIGHT_CAST_SHADOWS;
        l.shadow_map_index = 0xFFFF;
        c->lights.push_back(l);
    }
    c->dir_light = *dl;
    c->lights_set = true;
    c->point_shadow.clear(); c->point_shadow_static.clear();
    compute_cascade_splits(*c);
    return RDR_OK;
}

int orc_set_environment_map(rdr_ctx* c, const float* const faces[6], uint32_t w, uint32_t h) {
    if (!c || !faces || w == 0 || w != h) return fail(c, RDR_ERR_INVALID, "set_environment_map: faces must be square");
    c->env.res = w; c->env.levels = 1;
    c->env.level.assign(1, std::vector<uint32_t>((size_t)6 * w * w));
    for (int f = 0; f < 6; f++) {
        if (!faces[f]) return fail(c, RDR_ERR_INVALID, "set_environment_map: null face");
        for (size_t i = 0; i < (size_t)w * w; i++)
            c->env.level[0][(size_t)f * w * w + i] = pack_r11g11b10(faces[f][i * 4], faces[f][i * 4 + 1], faces[f][i * 4 + 2]);
    }
    bake_irradiance(*c);
    bake_prefiltered(*c);
    return RDR_OK;
}

int orc_set_environment_equirect(rdr_ctx* c, const float* pano, uint32_t w, uint32_t h, uint32_t face_res) {
    if (!c || !pano || w == 0 || h == 0 || face_res == 0 || face_res > 8192) return fail(c, RDR_ERR_INVALID, "set_environment_equirect: bad argument");
    equirect_to_cube(*c, pano, w, h, face_res);
    bake_irradiance(*c);
    bake_prefiltered(*c);
    return RDR_OK;
}

int orc_generate_brdf_lut(rdr_ctx* c, uint32_t size, uint32_t samples) {
    if (!c || size == 0 || size > 4096 || samples == 0 || samples > 65536) return fail(c, RDR_ERR_INVALID, "generate_brdf_lut: bad argument");
    generate_brdf_lut(*c, size, samples);
    return RDR_OK;
}

int orc_read_brdf_lut(rdr_ctx* c, uint8_t* out, uint32_t* ow, uint32_t* oh) {
    if (!c || !c->lut_w) return fail(c, RDR_ERR_INVALID, "read_brdf_lut: no table");
    if (ow) *ow = c->lut_w;
    if (oh) *oh = c->lut_h;
    if (out) std::memcpy(out, c->lut.data(), c->lut.size());
    return RDR_OK;
}

int orc_set_brdf_lut(rdr_ctx* c, const uint8_t* px, uint32_t w, uint32_t h, uint32_t channels) {
    if (!c || !px || w == 0 || h == 0 || (channels != 3 && channels != 4)) return fail(c, RDR_ERR_INVALID, "set_brdf_lut: bad argument");
    c->lut_w = w; c->lut_h = h;
    c->lut.resize((size_t)w * h * 4);
    for (size_t i = 0; i < (size_t)w * h; i++) {
        for (int k = 0; k < 3; k++) c->lut[i * 4 + k] = px[i * channels + k];
        c->lut[i * 4 + 3] = channels == 4 ? px[i * 4 + 3] : 255;
    }
    return RDR_OK;
}

int orc_set_camera(rdr_ctx* c, const rdr_camera* cam) {
    if (!c || !cam) return fail(c, RDR_ERR_INVALID, "set_camera: null");
    c->camera.type = cam->type;
    c->camera.position = v3(cam->position[0], cam->position[1], cam->position[2]);
    c->camera.euler = v3(cam->euler[0], cam->euler[1], cam->euler[2]);
    c->camera.near_plane = cam->near_plane; c->camera.far_plane = cam->far_plane;
    c->camera.fov_or_height = cam->fov_or_height; c->camera.aspect = cam->aspect_ratio;
    return RDR_OK;
}

int orc_set_exposure_bias(rdr_ctx* c, float bias, float gamma) {
    if (!c) return RDR_ERR_INVALID;
    c->settings.exposure_bias = bias; c->settings.gamma = gamma;
    return RDR_OK;
}

// exposure recurrence of Renderer/VulkanRenderer.cpp:1075-1082
static float compute_exposure(Ctx& c, const rdr_frame_params* fp) {
    if (fp && fp->exposure_mode == 2) return fp->exposure;
    if (fp && fp->exposure_mode == 1) c.temporal_luminance = fp->temporal_luminance;
    else {
        const double dt = fp ? (double)fp->delta_time : 1.0 / 60.0;
        const float avg = std::min(c.last_avg_log_lum, 10.0f);
        c.temporal_luminance = (float)(c.temporal_luminance + (std::exp2(avg) - c.temporal_luminance) * (1 - std::exp(-dt)));
    }
    float exposure = 1.03f - 2 / (std::log2(c.temporal_luminance + 1.0f) + 2);
    exposure = 0.18f / exposure + c.settings.exposure_bias;
    return exposure;
}

// Frames are rendered to completion, one at a time. What the HIP library promises a caller that keeps two frames in
// flight (include/rdr.h, rdr_render_frame) is reproduced for the statistics: after a call with a "device" destination
// (here simply memory) rdr_get_frame_stats still describes the frame before it, until the next frame or a synchronize.
int orc_render_frame(rdr_ctx* c, const rdr_frame_params* fp, uint8_t* out_rgba8, int out_is_device, float* out_avg) {
    if (!c) return RDR_ERR_INVALID;
    if (c->stats_deferred) { c->done_stats = c->stats; c->stats_deferred = false; }
    const uint32_t W = c->W, H = c->H, S = c->settings.msaa;
    const uint64_t frame_index = c->stats.frame_index + 1;
    c->stats = rdr_frame_stats{};
    c->stats.frame_index = frame_index;
    // per-pass wall-clock times of this CPU run (what bench.py reports beside the GPU's pass times)
    const auto now = [] { return std::chrono::steady_clock::now(); };
    const auto ms_between = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<float, std::milli>(b - a).count();
    };
    const auto t_begin = now();
    const V3 cam_pos = c->camera.position;
    const M4 viewproj = c->camera.viewProj();
    if (c->lights_set) {
        render_point_shadows(*c, cam_pos);
        render_directional_shadows(*c);
    }
    // main pass (VulkanRenderer.cpp:1043-1065)
    std::vector<uint32_t> opaque, non_opaque;
    for (uint32_t p = 0; p < c->prims.size(); p++) (c->prims[p].mat.alpha_mode == RDR_ALPHA_OPAQUE ? opaque : non_opaque).push_back(p);
    std::vector<Draw> od = build_draws(*c, opaque, viewproj, cam_pos, SORT_FRONT_TO_BACK);
    std::vector<Draw> bd = build_draws(*c, non_opaque, viewproj, cam_pos, SORT_BACK_TO_FRONT);
    c->stats.draws_main = (uint32_t)(od.size() + bd.size());
    c->depth.assign((size_t)S * W * H, 1.0f);
    c->vis.assign((size_t)S * W * H, RDR_INVALID_ID);
    RasterTarget rt{W, H, S, c->depth.data(), c->vis.data()};
    raster_draws(*c, od, rt, CULL_BACK, &c->stats.triangle_instances, &c->stats.triangles_binned);
    c->stats.views++;
    const auto t_geometry = now();
    shade_main(*c, od, bd, viewproj, cam_pos);
    const auto t_shade = now();
    const float exposure = compute_exposure(*c, fp);
    bloom_and_tonemap(*c, exposure);
    const float avg = average_luminance(*c);
    const auto t_post = now();
    c->stats.ms_geometry = ms_between(t_begin, t_geometry);
    c->stats.ms_shade = ms_between(t_geometry, t_shade);
    c->stats.ms_bloom = ms_between(t_shade, t_post);  // bloom chain + tonemap + luminance are one function here
    c->stats.ms_total = ms_between(t_begin, t_post);
    c->last_avg_log_lum = avg;
    c->stats.exposure = exposure; c->stats.avg_log_luminance = avg;
    if (out_rgba8) std::memcpy(out_rgba8, c->out.data(), c->out.size());
    if (out_avg) *out_avg = avg;
    if (out_is_device && out_rgba8) c->stats_deferred = true;
    else c->done_stats = c->stats;
    return RDR_OK;
}
int orc_synchronize(rdr_ctx* c) {
    if (c && c->stats_deferred) { c->done_stats = c->stats; c->stats_deferred = false; }
    return RDR_OK;
}
int orc_wait_frame(rdr_ctx* c, uint64_t frame_index) {
    if (!c || frame_index > c->stats.frame_index) return RDR_ERR_INVALID;
    if (c->stats_deferred && c->stats.frame_index <= frame_index) { c->done_stats = c->stats; c->stats_deferred = false; }
    return RDR_OK;
}
int orc_get_frame_stats(rdr_ctx* c, rdr_frame_stats* out) { if (!c || !out) return RDR_ERR_INVALID; *out = c->done_stats; return RDR_OK; }

int orc_read_depth(rdr_ctx* c, float* out) { if (!c || !out || c->depth.empty()) return RDR_ERR_INVALID; std::memcpy(out, c->depth.data(), c->depth.size() * 4); return RDR_OK; }
int orc_read_visibility(rdr_ctx* c, uint32_t* out) { if (!c || !out || c->vis.empty()) return RDR_ERR_INVALID; std::memcpy(out, c->vis.data(), c->vis.size() * 4); return RDR_OK; }
int orc_read_directional_shadow(rdr_ctx* c, uint32_t index, float* out) {
    const size_t px = (size_t)c->settings.directional_shadow_map_resolution * c->settings.directional_shadow_map_resolution;
    if (!c || !out || (index + 1) * px > c->dir_shadow.size()) return RDR_ERR_INVALID;
    std::memcpy(out, c->dir_shadow.data() + index * px, px * 4);
    return RDR_OK;
}
int orc_read_point_shadow(rdr_ctx* c, uint32_t slot, uint32_t face, float* out) {
    const size_t px = (size_t)c->settings.point_shadow_map_resolution * c->settings.point_shadow_map_resolution;
    const size_t idx = (size_t)slot * 6 + face;
    if (!c || !out || face >= 6 || (idx + 1) * px > c->point_shadow.size()) return RDR_ERR_INVALID;
    std::memcpy(out, c->point_shadow.data() + idx * px, px * 4);
    return RDR_OK;
}
int orc_read_hdr(rdr_ctx* c, float* out) { if (!c || !out || c->hdr.empty()) return RDR_ERR_INVALID; std::memcpy(out, c->hdr.data(), c->hdr.size() * 4); return RDR_OK; }
int orc_keep_intermediates(rdr_ctx* c, uint32_t) { return c ? RDR_OK : RDR_ERR_INVALID; }  // every image is always kept here
int orc_read_bloom(rdr_ctx* c, uint32_t level, float* out) {
    if (!c || !out || level >= c->bloom.size()) return RDR_ERR_INVALID;
    for (size_t i = 0; i < c->bloom[level].size(); i++) unpack_r11g11b10(c->bloom[level][i], out[i * 3], out[i * 3 + 1], out[i * 3 + 2]);
    return RDR_OK;
}
int orc_read_log_luminance(rdr_ctx* c, float* out) { if (!c || !out || c->loglum.empty()) return RDR_ERR_INVALID; std::memcpy(out, c->loglum.data(), c->loglum.size() * 4); return RDR_OK; }
int orc_read_env_cube(rdr_ctx* c, int which, uint32_t level, float* out) {
    if (!c || !out) return RDR_ERR_INVALID;
    const Cube& cube = which == 0 ? c->env : (which == 1 ? c->irradiance : c->prefiltered);
    if (level >= cube.levels) return RDR_ERR_INVALID;
    for (size_t i = 0; i < cube.level[level].size(); i++) unpack_r11g11b10(cube.level[level][i], out[i * 3], out[i * 3 + 1], out[i * 3 + 2]);
    return RDR_OK;
}
int orc_read_cascades(rdr_ctx* c, float* vp, float* depth) {
    if (!c || !vp || !depth || !c->lights_set) return RDR_ERR_INVALID;
    for (uint32_t i = 0; i < 4; i++) {
        if (i < c->settings.directional_shadow_cascade_levels) { std::memcpy(vp + 16 * i, c->csm_viewproj[i].c, 64); depth[i] = c->cascade_camera_depths[i]; }
        else { std::memset(vp + 16 * i, 0, 64); depth[i] = 0.0f; }
    }
    return RDR_OK;
}
int orc_read_image_level(rdr_ctx* c, uint32_t image, uint32_t level, uint8_t* out, uint32_t* ow, uint32_t* oh) {
    if (!c || image >= c->images.size() || level >= c->images[image].mips.size()) return RDR_ERR_INVALID;
    const Image& img = c->images[image];
    if (ow) *ow = std::max(1u, img.w >> level);
    if (oh) *oh = std::max(1u, img.h >> level);
    if (out) std::memcpy(out, img.mips[level].data(), img.mips[level].size());
    return RDR_OK;
}

int orc_post_process(rdr_ctx* c, const float* hdr, float exposure, uint8_t* out_rgba8, float* out_avg) {
    if (!c || !hdr) return RDR_ERR_INVALID;
    c->hdr.resize((size_t)c->W * c->H * 4);
    for (size_t i = 0; i < c->hdr.size(); i++) c->hdr[i] = quantize_half(hdr[i]);
    bloom_and_tonemap(*c, exposure);
    const float avg = average_luminance(*c);
    if (out_rgba8) std::memcpy(out_rgba8, c->out.data(), c->out.size());
    if (out_avg) *out_avg = avg;
    return RDR_OK;
}

}  // extern "C"

// Test hook (not part of the rdr.h ABI): texel (i, j) of a cube face, folded over the cube edge when one step outside.
extern "C" void orc_test_cube_fold(int face, int i, int j, int r, int out[3]) {
    const size_t o = orc::cube_texel_offset(face, i, j, r);
    out[0] = (int)(o / ((size_t)r * r)); out[1] = (int)(o % r); out[2] = (int)((o / r) % r);
}
