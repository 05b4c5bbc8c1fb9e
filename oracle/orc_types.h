// TEST INFRASTRUCTURE — CPU restatement ("oracle") of the reference's
// per-frame pixel pipeline. Not part of the product: only tests/,
// __graft_entry__.smoke() and the cpu_baseline leg of bench.py may use it.
// parity unpinned: the reference has no tests, golden images or vectors for
// this path, and its Vulkan/lavapipe execution is unavailable here; this
// restatement follows the reference's GLSL shaders and pipeline-state code
// (cited per function) plus the Vulkan 1.3 specification for fixed-function
// behaviour, and is checked against hand-derived known-answer tests.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../include/rdr.h"
#include "orc_formats.h"
#include "orc_math.h"

namespace orc {

struct Image {
    uint32_t w = 0, h = 0;
    bool srgb = false;
    bool is_float = false;     // an SFLOAT format of Renderer/Image.h:9-12: a level holds one float[4] per texel instead of RGBA8
    bool half_texels = false;  // ... stored as binary16 in the reference's image: every level holds values rounded to it
    std::vector<std::vector<uint8_t>> mips;  // per level: RGBA8, or the bytes of float[4] texels
};
struct Texture {
    uint32_t image;
    rdr_sampler sampler;
};
struct VertexAttr {  // reference Vertex (Renderer/Vertex.h:8-12) minus the position
    V3 normal, tangent, bitangent;
    float u, v;
};
struct Primitive {
    std::vector<V3> pos;
    std::vector<VertexAttr> attr;
    std::vector<uint32_t> idx;  // always expanded to a triangle list
    V3 bb_min, bb_max;
    V3 barycenter;  // world space, fixed at load (Renderer/Mesh.cpp:77-86)
    rdr_material mat;
    uint32_t mesh;
    // skinned primitives keep their rest pose; pos/attr/bb_* above then hold the current pose
    bool skinned = false;
    std::vector<V3> rest_pos;
    std::vector<VertexAttr> rest_attr;
    std::vector<uint16_t> joints;  // 4 per vertex
    std::vector<float> weights;    // 4 per vertex
    V3 rest_bb_min, rest_bb_max;
};
struct Mesh {
    M4 model;
    bool is_static;
    uint32_t first_prim, prim_count;
    uint32_t n_joints = 0;  // largest joint index in use + 1; 0 = not skinned
};
struct Camera {  // Renderer/Camera.h:61-82
    int type = RDR_CAMERA_PERSPECTIVE;
    V3 position{0, 0, 0};
    V3 euler{0, 0, 0};
    float near_plane = 0.1f, far_plane = 100.0f, fov_or_height = 35.0f, aspect = 16.0f / 9.0f;
    M4 crop = identity();
    M4 view() const;
    M4 proj() const;
    M4 viewProj() const;
};
struct PointLight {
    V3 point, intensity;
    uint32_t flags;
    uint16_t shadow_map_index;
};
struct Cube {  // B10G11R11 packed, level-major then face-major
    uint32_t res = 0, levels = 0;
    std::vector<std::vector<uint32_t>> level;  // level[l][face*r*r + y*r + x]
};
struct Draw {
    uint32_t prim;
    M4 mvp, model;
    uint32_t first_tri;  // running triangle index within the pass
};

enum CullMode { CULL_NONE = 0, CULL_FRONT = 1, CULL_BACK = 2 };

struct FragmentSink {  // called for every sample that passed the depth test, in submission order
    virtual void fragment(int px, int py, int s, uint32_t id) = 0;
    virtual ~FragmentSink() {}
};
struct RasterTarget {
    uint32_t w, h, samples;
    float* depth;     // [samples][h][w]
    uint32_t* vis;    // optional, same layout
    FragmentSink* (*make_sink)(void* user, int band) = nullptr;  // optional: one sink per band (thread)
    void* sink_user = nullptr;
};

struct Ctx {
    rdr_settings settings;
    uint32_t W, H;
    std::string err;
    SrgbTables srgb;

    std::vector<std::vector<uint8_t>> buffers;
    std::vector<Image> images;
    std::vector<Texture> textures;
    std::vector<Primitive> prims;
    std::vector<Mesh> meshes;
    uint32_t tex_white, tex_normal;  // reference default textures

    Camera camera;
    std::vector<PointLight> lights;
    rdr_directional_light dir_light{};
    bool lights_set = false;
    std::vector<float> cascade_depths;             // view-space split distances
    std::vector<float> cascade_camera_depths;      // (f - f n/d)/(f-n)
    M4 csm_viewproj[4];

    Cube env, irradiance, prefiltered;
    uint32_t lut_w = 0, lut_h = 0;
    std::vector<uint8_t> lut;  // RGBA8

    // per-frame images
    std::vector<float> depth;      // [samples][H][W]
    std::vector<uint32_t> vis;     // [samples][H][W] after opaque
    std::vector<float> sample_color;  // [samples][H][W][4], only when blended prims exist or msaa>1
    std::vector<float> hdr;        // [H][W][4], values representable in fp16
    std::vector<std::vector<uint32_t>> bloom;  // 6 mips packed
    std::vector<uint8_t> out;      // RGBA8
    std::vector<float> loglum;     // [H][W]
    std::vector<float> dir_shadow;   // [4][res][res]
    std::vector<float> point_shadow, point_shadow_static;  // [count*6][res][res]

    float temporal_luminance = 0.0f;
    float last_avg_log_lum = 0.0f;
    rdr_frame_stats stats{};       // of the frame rendered last
    rdr_frame_stats done_stats{};  // what rdr_get_frame_stats hands out (see orc_render_frame)
    bool stats_deferred = false;
};

// orc_post.cpp
void bloom_and_tonemap(Ctx& c, float exposure);
float average_luminance(Ctx& c);
// orc_raster.cpp
void raster_draws(const Ctx& c, const std::vector<Draw>& draws, const RasterTarget& rt, CullMode cull,
                  uint64_t* tri_instances, uint64_t* tri_binned);
// orc_shade.cpp
void shade_main(Ctx& c, const std::vector<Draw>& opaque, const std::vector<Draw>& blended, const M4& viewproj,
                V3 camera_pos);
// orc_ibl.cpp
void bake_irradiance(Ctx& c);
void bake_prefiltered(Ctx& c);
void cube_sample(const Cube& cube, uint32_t level, float dx, float dy, float dz, float out[3]);
void equirect_to_cube(Ctx& c, const float* pano, uint32_t w, uint32_t h, uint32_t res);
void generate_brdf_lut(Ctx& c, uint32_t size, uint32_t samples);
size_t cube_texel_offset(int face, int i, int j, int r);  // texel (i,j) of a face, one step outside folds over the cube edge
// orc_tex.cpp
void build_mips(Ctx& c, Image& img, uint32_t max_levels);
void sample_texture(const Ctx& c, uint32_t tex, float u, float v, float dudx, float dvdx, float dudy, float dvdy,
                    float out[4]);

}  // namespace orc
