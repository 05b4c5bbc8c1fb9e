/*
 * rdr.h — C ABI of the MI355X-native per-frame pixel pipeline.
 *
 * The reference (Samlibardi/Renderer) has no plugin/FFI seam: its GPU path is
 * the private half of `class VulkanRenderer`. The seam this library cuts is
 * that class's PUBLIC surface (Renderer/VulkanRenderer.h:94-110). Every entry
 * point below names the reference member it stands in for. The C++ façade in
 * include/renderer.hpp re-creates the reference's own class names on top of it.
 *
 * Conventions
 *  - plain C types only; no exceptions cross the boundary; every call returns
 *    an int status (RDR_OK == 0) and leaves a message for rdr_last_error().
 *  - all matrices are column-major float[16] (glm layout: m[col*4+row]).
 *  - every load/set call COPIES the caller's memory before it returns (the
 *    reference does a staging memcpy + blocking fence:
 *    Renderer/VulkanRenderer.cpp:281-294, 306-355).
 *  - resources are opaque uint32_t ids; 0xFFFFFFFF is invalid
 *    (Renderer/Handle.h:3-19).
 *  - a context is externally synchronised (one thread at a time) and is bound
 *    to one GPU; contexts are independent of each other.
 */
#ifndef RDR_H
#define RDR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RDR_OK 0
#define RDR_ERR_INVALID 1   /* bad argument / bad id / wrong state          */
#define RDR_ERR_DEVICE 2    /* HIP runtime error (message has the HIP text) */
#define RDR_ERR_CAPACITY 3  /* a fixed-capacity device buffer would overflow */
#define RDR_ERR_UNSUPPORTED 4

#define RDR_INVALID_ID 0xFFFFFFFFu

typedef struct rdr_ctx rdr_ctx;

/* Mirrors RendererSettings (Renderer/VulkanRenderer.h:44-59) field for field;
 * bools widened to uint32_t so the struct has no padding surprises. */
typedef struct rdr_settings {
    uint32_t hdr_output_enabled;               /* default 0: the output image is R8G8B8A8_UNORM. 1: R16G16B16A16_SFLOAT (the float
                                                * format of VulkanRenderer.cpp:668-673 that presentation surfaces offer; there is no
                                                * surface to ask here): tonemap.comp's values unquantised, 8 bytes per pixel */
    uint32_t dynamic_shadows_enabled;          /* default 1   */
    uint32_t point_shadow_map_resolution;      /* default 512;  1 .. 8192  */
    uint32_t point_shadow_map_count;           /* default 8;    0 .. 64    */
    uint32_t directional_shadow_map_resolution;/* default 2048; 1 .. 16384 */
    uint32_t directional_shadow_cascade_levels;/* default 4 (max 4) */
    uint32_t vsync;                            /* ignored (headless) */
    float exposure_bias;                       /* default 0   */
    float gamma;                               /* default 2.2 */
    uint32_t msaa;                             /* 1 or 4; default 4 */
} rdr_settings;

/* Image formats: Renderer/Image.h:5-14 (same order, same values). Input texels are tightly packed in the named
 * format (3, 4, 6, 8, 12 or 16 bytes each). The three-component formats read alpha as 1. */
enum rdr_image_format {
    RDR_FORMAT_R8G8B8_UNORM = 0,
    RDR_FORMAT_R8G8B8_SRGB = 1,
    RDR_FORMAT_R8G8B8A8_UNORM = 2,
    RDR_FORMAT_R8G8B8A8_SRGB = 3,
    RDR_FORMAT_R16G16B16_SFLOAT = 4,
    RDR_FORMAT_R16G16B16A16_SFLOAT = 5,
    RDR_FORMAT_R32G32B32_SFLOAT = 6,
    RDR_FORMAT_R32G32B32A32_SFLOAT = 7,
};

/* Sampler description consumed by makeTexture
 * (Renderer/VulkanRenderer.cpp:363-385; Sampler.h itself is missing from the
 * reference tree, the fields are the ones that function reads). */
enum rdr_sampler_filter { RDR_FILTER_NEAREST = 0, RDR_FILTER_LINEAR = 1 };
enum rdr_sampler_wrap { RDR_WRAP_REPEAT = 0, RDR_WRAP_MIRROR = 1, RDR_WRAP_CLAMP = 2 };
typedef struct rdr_sampler {
    uint32_t mag_filter;     /* within a level where the level of detail lambda <= 0 */
    uint32_t min_filter;     /* within a level where lambda > 0 */
    uint32_t mipmap_filter;  /* LINEAR: the two levels around lambda blended; NEAREST: level ceil(lambda + 0.5) - 1 */
    uint32_t wrap_u;
    uint32_t wrap_v;   /* as the reference reads it (VulkanRenderer.cpp:370-371): REPEAT if it says REPEAT; otherwise
                        * MIRROR when wrap_u is MIRROR and CLAMP when it is not, whatever wrap_v itself names */
} rdr_sampler;

/* Vertex attribute / index descriptions: Renderer/Mesh.h:12-80. */
enum rdr_value_type {
    RDR_INT8 = 0, RDR_INT16, RDR_INT32, RDR_INT64,
    RDR_UINT8, RDR_UINT16, RDR_UINT32, RDR_UINT64,
    RDR_HALF, RDR_FLOAT, RDR_DOUBLE
};
enum rdr_container_type { RDR_SCALAR = 0, RDR_VEC2, RDR_VEC3, RDR_VEC4 };
enum rdr_attribute_name {
    RDR_ATTR_POSITION = 0,   /* "POSITION"   vec3 float               */
    RDR_ATTR_NORMAL = 1,     /* "NORMAL"     vec3 float               */
    RDR_ATTR_TANGENT = 2,    /* "TANGENT"    vec4 float (w = handedness) */
    RDR_ATTR_TEXCOORD_0 = 3, /* "TEXCOORD_0" vec2 float / unorm8 / unorm16 */
    RDR_ATTR_BITANGENT = 4,  /* legacy interleaved Vertex only (Vertex.h:11) */
    RDR_ATTR_JOINTS_0 = 5,   /* "JOINTS_0"   vec4 uint8 / uint16; not read by the reference (main.cpp:61-177) */
    RDR_ATTR_WEIGHTS_0 = 6,  /* "WEIGHTS_0"  vec4 float / unorm8 / unorm16; a primitive with both is skinned */
};
typedef struct rdr_vertex_attribute {
    uint32_t name;      /* rdr_attribute_name */
    uint32_t buffer;    /* id from rdr_load_buffer */
    uint64_t offset;    /* bytes into the buffer */
    int32_t stride;     /* bytes between elements; 0 = tightly packed */
    uint32_t container; /* rdr_container_type */
    uint64_t count;
    uint32_t value_type;/* rdr_value_type */
    uint32_t _pad;
} rdr_vertex_attribute;
typedef struct rdr_index_buffer {
    uint32_t buffer;    /* RDR_INVALID_ID = non-indexed */
    uint32_t index_type;/* RDR_UINT8 / RDR_UINT16 / RDR_UINT32 */
    uint64_t offset;
    uint64_t count;
} rdr_index_buffer;

/* PBRInfo + AlphaInfo + the five material textures. The structs are used at
 * Renderer/main.cpp:162-165 and Renderer/VulkanRenderer.cpp:414-470 and read
 * by Renderer/shaders/pbr.frag:21-37. */
enum rdr_alpha_mode { RDR_ALPHA_OPAQUE = 0, RDR_ALPHA_MASK = 1, RDR_ALPHA_BLEND = 2 };
typedef struct rdr_material {
    float base_color_factor[4];
    float emissive_factor[4];
    float normal_scale;     /* carried, never read by pbr.frag */
    float metallic_factor;
    float roughness_factor;
    float ao_factor;
    int32_t alpha_mode;     /* rdr_alpha_mode */
    float alpha_cutoff;
    /* texture ids from rdr_make_texture; RDR_INVALID_ID selects the
     * reference's defaults: 1x1 white (VulkanRenderer.h:28-32) and, for the
     * normal map, 1x1 (128,128,255,0) (main.cpp:149, textures/clear_normal.png) */
    uint32_t albedo_texture;        /* sampled as sRGB (main.cpp:138) */
    uint32_t normal_texture;
    uint32_t metal_roughness_texture;
    uint32_t ao_texture;
    uint32_t emissive_texture;
    uint32_t _pad;
} rdr_material;

typedef struct rdr_primitive_desc {
    const rdr_vertex_attribute* attributes;
    uint32_t attribute_count;
    uint32_t mode;          /* must be 4 (triangles, Mesh.h:54-62) */
    rdr_index_buffer indices;
    float bb_min[3];
    float bb_max[3];
    rdr_material material;
} rdr_primitive_desc;

/* PointLight (Renderer/PointLight.h:19-28). flags/shadow_map_index are
 * outputs of slot assignment and are ignored on input except CAST_SHADOWS. */
#define RDR_POINT_LIGHT_CAST_SHADOWS 0x01u
#define RDR_POINT_LIGHT_HAS_SHADOW_MAP 0x02u
#define RDR_POINT_LIGHT_STATIC_SHADOW_MAP_RENDERED 0x04u
typedef struct rdr_point_light {
    float point[3];
    float intensity[3];
    uint32_t flags;
} rdr_point_light;

/* DirectionalLight (Renderer/DirectionalLight.h:10-20). orientation is a
 * quaternion (w,x,y,z); the shader direction is orientation * (0,0,1)
 * (VulkanRenderer.cpp:263) and the light camera is
 * Ortographic(position, eulerAngles(orientation), .1, 200, 10, 1)
 * (DirectionalLight.cpp:3-5). */
typedef struct rdr_directional_light {
    float position[3];
    float intensity[3];
    float orientation_wxyz[4];
} rdr_directional_light;

/* Camera state (Renderer/Camera.h:61-82). */
enum rdr_camera_type { RDR_CAMERA_PERSPECTIVE = 0, RDR_CAMERA_ORTHOGRAPHIC = 1 };
typedef struct rdr_camera {
    int32_t type;
    float position[3];
    float euler[3];
    float near_plane;
    float far_plane;
    float fov_or_height;
    float aspect_ratio;
} rdr_camera;

/* Per-frame inputs. The reference derives exposure from wall-clock deltaTime
 * and a luminance readback that is some frames old
 * (VulkanRenderer.cpp:1019-1023, 1075-1082). Both are made explicit here:
 *  exposure_mode 0: run the reference recurrence with `delta_time`, using the
 *                   average log-luminance produced by the PREVIOUS
 *                   rdr_render_frame call on this context (0 for the first).
 *  exposure_mode 1: `temporal_luminance` is taken as the adapted luminance
 *                   for this frame (frames become order-independent).
 *  exposure_mode 2: `exposure` is pushed to the tonemap kernel as is. */
typedef struct rdr_frame_params {
    int32_t exposure_mode;
    float delta_time;
    float temporal_luminance;
    float exposure;
} rdr_frame_params;

/* Counters and per-pass device times of the last rendered frame. */
typedef struct rdr_frame_stats {
    uint64_t triangle_instances;  /* triangles submitted over all views */
    uint64_t triangles_binned;    /* survived clip/cull/bbox */
    uint64_t tile_pairs;          /* (triangle, tile) pairs sorted */
    uint32_t views;               /* raster views this frame */
    uint32_t draws_main;
    float ms_geometry;            /* setup + bin + sort + raster, all views */
    float ms_shade;               /* opaque shading + sky + blended pass */
    float ms_bloom;               /* colour->mip0 + 5 down + 5 up */
    float ms_tonemap;             /* tonemap + luminance reduction */
    float ms_total;
    float exposure;               /* value pushed to tonemap */
    float avg_log_luminance;
    float ms_shade_kernel;        /* the opaque+sky shading kernel alone */
    float ms_raster_kernel;       /* the tile raster kernel(s) of the main batch alone */
    uint64_t chunks_culled;       /* 64-triangle chunks whose bounding box the GPU culling pre-pass found outside a
                                   * view (their triangles skip set-up); 0 from the CPU checker, which has no such
                                   * pre-pass. The reference culls whole primitives only (VulkanRenderer.cpp:929-943). */
    uint64_t frame_index;         /* which rdr_render_frame call of this context these numbers belong to, counted from 1 */
    uint32_t blend_groups;        /* 0: the blended pass ran over all tiles at once; n > 0: its fragment list would have exceeded
                                   * the context's budget and the tiles were rendered in n groups (same result, bounded memory) */
    float host_ms;                /* host time rdr_render_frame spent building and enqueueing this frame (culling, ordering, uploads) */
    uint32_t views_reused;        /* shadow views (cascades, point-light faces) not rendered because nothing that goes into them
                                   * changed since they were last rendered; they are not part of `views` and the counters above.
                                   * The reference redraws them every frame (VulkanRendererShadow.cpp:348, 424); 0 from the CPU checker */
    uint32_t draw_lists_on_device; /* 1: culling and ordering of the primitives per view ran on the GPU (scenes with >= 2048
                                   * primitives), 0: in the host loop that follows drawMeshes (VulkanRenderer.cpp:945-999) */
} rdr_frame_stats;

/* ---- lifetime ---------------------------------------------------------- */

/* RendererSettings{} defaults (VulkanRenderer.h:44-59). */
int rdr_settings_default(rdr_settings* out);

/* VulkanRenderer::VulkanRenderer(window, settings) (VulkanRenderer.cpp:19-81);
 * the window becomes (width,height); `device` is the HIP device ordinal.
 * Camera starts as Perspective(0,0,.1,60,radians(40),width/height)
 * (VulkanRenderer.h:117, VulkanRenderer.cpp:50). */
int rdr_create(const rdr_settings* settings, uint32_t width, uint32_t height, int device, rdr_ctx** out);
/* VulkanRenderer::~VulkanRenderer (VulkanRenderer.cpp:83-188). */
void rdr_destroy(rdr_ctx* ctx);
/* Replaces the C++ exceptions of the reference (VulkanRenderer.cpp:38,593). */
const char* rdr_last_error(const rdr_ctx* ctx);
/* Name of the back end: "hip:gfx950". */
const char* rdr_backend_name(void);

/* ---- resources --------------------------------------------------------- */

/* VulkanRenderer::loadBuffer (VulkanRenderer.cpp:273-299). */
int rdr_load_buffer(rdr_ctx* ctx, const void* ptr, size_t size, uint32_t* out_id);
/* VulkanRenderer::loadImage (VulkanRenderer.cpp:305-361): uploads level 0 and
 * builds the mip chain by successive linear blits (sRGB formats filter in linear
 * light; the 16-bit float formats round every level to binary16). */
int rdr_load_image(rdr_ctx* ctx, const void* ptr, size_t size, uint32_t width, uint32_t height,
                   int format, uint32_t max_mip_levels, uint32_t* out_id);
/* VulkanRenderer::makeTexture (VulkanRenderer.cpp:363-385). */
int rdr_make_texture(rdr_ctx* ctx, uint32_t image, const rdr_sampler* sampler, uint32_t* out_id);
/* VulkanRenderer::addMesh / setMeshes (VulkanRenderer.cpp:301-303, 387-492):
 * one Mesh = `count` primitives sharing one node. `model` is
 * node->modelMatrix() (Node.cpp:5-7); `is_static` is node->isStatic()
 * (Node.h:63). Returns the mesh id used by rdr_set_mesh_transform. */
int rdr_add_mesh(rdr_ctx* ctx, const rdr_primitive_desc* primitives, uint32_t count,
                 const float model[16], int is_static, uint32_t* out_mesh_id);
/* drawMeshes reads mesh->node->modelMatrix() every frame
 * (VulkanRenderer.cpp:975); this is how a caller moves a node. */
int rdr_set_mesh_transform(rdr_ctx* ctx, uint32_t mesh_id, const float model[16]);
/* No counterpart in the reference, which has node animation only (Node.cpp:21-38) and reads no skins. Sets the
 * joint-matrix palette of a mesh whose primitives carry JOINTS_0 + WEIGHTS_0: `count` column-major 4x4 matrices in
 * the mesh's model space (glTF: inverse(meshGlobal) * jointGlobal * inverseBind). Before the next frame every vertex
 * becomes S * rest with S = w0*J[j0] + w1*J[j1] + w2*J[j2] + w3*J[j3] (positions with w = 1; normal, tangent and
 * bitangent by the upper 3x3, as the vertex stage treats `model`, basic.vert:41-44); the mesh's model matrix applies
 * afterwards as usual. The culling box of each primitive becomes the box around its rest box carried by every joint.
 * `count` must cover the largest joint index in use; the mesh must have been added with is_static == 0. Until the
 * first call the palette is the identity (rest pose). */
int rdr_set_mesh_joints(rdr_ctx* ctx, uint32_t mesh_id, const float* joint_matrices, uint32_t count);
/* VulkanRenderer::setLights (VulkanRenderer.cpp:204-246). Also fixes the PSSM
 * cascade splits from the camera's near/far AT THIS MOMENT
 * (VulkanRendererShadow.cpp:152-175). */
int rdr_set_lights(rdr_ctx* ctx, const rdr_point_light* lights, uint32_t count,
                   const rdr_directional_light* directional);
/* VulkanRenderer::setEnvironmentMap (VulkanRendererEnvironment.cpp:71-125):
 * six RGBA32F faces in the order +X,-X,+Y,-Y,+Z,-Z (main.cpp:434-441); runs the
 * irradiance and GGX-prefilter bakes (…Environment.cpp:193-250, 324-391).
 * The cubes and the BRDF table exist once per context: this call, rdr_set_environment_equirect, rdr_set_brdf_lut and
 * rdr_generate_brdf_lut first wait for the frames in flight (which read them) to finish on the GPU; the finished
 * frames' results are picked up as usual afterwards. */
int rdr_set_environment_map(rdr_ctx* ctx, const float* const faces_rgba32f[6], uint32_t width, uint32_t height);
/* The reference loads textures/ibl_brdf_lut.png in its constructor
 * (VulkanRenderer.cpp:44-45); here the caller hands over the decoded RGB(A)8
 * pixels, row 0 first. `channels` is 3 or 4. */
int rdr_set_brdf_lut(rdr_ctx* ctx, const uint8_t* pixels, uint32_t width, uint32_t height, uint32_t channels);
/* No counterpart in the reference, which is handed six decoded cube faces (main.cpp:434-441): converts an
 * equirectangular RGBA32F panorama (row 0 = the +Y pole, column 0 = the -Z direction, u grows towards +X) into six
 * face_res^2 faces by bilinear look-up (u wraps, v clamps), then does what rdr_set_environment_map does. */
int rdr_set_environment_equirect(rdr_ctx* ctx, const float* rgba32f, uint32_t width, uint32_t height, uint32_t face_res);
/* No counterpart in the reference, which ships the table as textures/ibl_brdf_lut.png: integrates the split-sum
 * BRDF table (GGX importance sampling on the Hammersley points of envbakespecular.frag:17-53; R = scale, G = bias
 * of F0) and installs it like rdr_set_brdf_lut. Layout as the shipped file: column = NdotV, row 0 = roughness 1. */
int rdr_generate_brdf_lut(rdr_ctx* ctx, uint32_t size, uint32_t samples);
/* No counterpart in the reference (which times nothing, VulkanRenderer.cpp:1019-1023 aside). Device time in ms of the
 * kernels of the last environment precompute on this context: out[0] panorama -> cube faces
 * (rdr_set_environment_equirect), out[1] irradiance bake, out[2] GGX prefilter bake (ten levels), out[3] BRDF table
 * (rdr_generate_brdf_lut); uploads and allocation are not in these numbers. HIP library only. */
int rdr_get_precompute_ms(rdr_ctx* ctx, float out_ms[4]);
/* renderer.camera() mutators (VulkanRenderer.h:108, Camera.h:35-54) collapsed
 * into one state upload. */
int rdr_set_camera(rdr_ctx* ctx, const rdr_camera* camera);
/* renderer.settings().exposureBias / gamma are the two settings the render
 * loop re-reads every frame (VulkanRenderer.cpp:1082,1094). */
int rdr_set_exposure_bias(rdr_ctx* ctx, float exposure_bias, float gamma);

/* ---- the hot path ------------------------------------------------------ */

/* One iteration of VulkanRenderer::renderLoop (VulkanRenderer.cpp:1013-1110):
 * shadow maps -> main pass (opaque, skybox, blended) -> bloom -> tonemap ->
 * average luminance. `out_rgba8` receives width*height*4 bytes, row 0 first
 * (width*height*8 bytes of RGBA16F when the context was created with
 * hdr_output_enabled; the same holds for the out_rgba8 of rdr_post_process
 * and rdr_post_process_noalloc, whose destination must then be 16-byte aligned);
 * it is a host pointer, or a device pointer when out_is_device != 0 (then the
 * call only enqueues work on the context's stream; use rdr_synchronize).
 * NULL skips the copy.
 *
 * Frames in flight (the reference keeps two, VulkanRenderer.cpp:1013-1030): with a device destination the next
 * rdr_render_frame may be called without rdr_synchronize in between. It builds and enqueues its frame while the GPU
 * finishes the previous one, and by the time it returns the previous frame is complete: its image is in its
 * destination and rdr_get_frame_stats describes it (frame_index tells which frame the numbers belong to). With
 * RDR_FRAMES_IN_FLIGHT=3 in the environment of rdr_create three frames are in flight and it is the frame two calls
 * back that is complete when a call returns; a destination buffer must then not be handed to another frame before the
 * frame it was given to is complete (three destinations in rotation, or rdr_wait_frame). rdr_wait_frame waits for
 * one frame, rdr_synchronize for everything enqueued; after either rdr_get_frame_stats describes the newest complete
 * frame. Results are handed over at those points only (never by rdr_get_frame_stats itself), so which frame the
 * statistics describe does not depend on timing. */
int rdr_render_frame(rdr_ctx* ctx, const rdr_frame_params* params, uint8_t* out_rgba8, int out_is_device,
                     float* out_avg_log_luminance);
int rdr_synchronize(rdr_ctx* ctx);
/* Waits until the frame with this frame_index (rdr_frame_stats::frame_index, counted from 1) is complete and hands
 * over the results of every frame up to and including it, oldest first; rdr_get_frame_stats then describes it.
 * Returns at once for a frame whose results were handed over before; RDR_ERR_INVALID for a frame not yet enqueued.
 * (The reference waits on the fence of a frame in flight the same way, VulkanRenderer.cpp:1016-1019.) */
int rdr_wait_frame(rdr_ctx* ctx, uint64_t frame_index);
int rdr_get_frame_stats(rdr_ctx* ctx, rdr_frame_stats* out);

/* ---- debug read-backs (needed for bit-exact parity tests) -------------- */

/* main-pass depth after the whole pass, sample-major: out[s*W*H + y*W + x],
 * `samples` = settings.msaa. A 4x frame stores depth only where its blended
 * pass reads it; the first call after such a frame waits for the frame and
 * rasterises the remaining tiles again (valid until the next render call). */
int rdr_read_depth(rdr_ctx* ctx, float* out);
/* main-pass visibility after the OPAQUE draws: id of the covering triangle
 * (draw-order index of the primitive in the high 8+ bits is NOT encoded; the
 * value is the running triangle index over the frustum-culled, sorted opaque
 * draw list), RDR_INVALID_ID where nothing covers. Sample-major like depth. */
int rdr_read_visibility(rdr_ctx* ctx, uint32_t* out);
/* cascade `index` of the directional shadow array (res*res floats). */
int rdr_read_directional_shadow(rdr_ctx* ctx, uint32_t index, float* out);
/* face `face` of cube `slot` of the point shadow cube array (res*res floats). */
int rdr_read_point_shadow(rdr_ctx* ctx, uint32_t slot, uint32_t face, float* out);
/* resolved HDR colour (RGBA, float32 values that are exactly representable
 * in the RGBA16F colour attachment), W*H*4 floats. */
int rdr_read_hdr(rdr_ctx* ctx, float* out);
/* Two images the reference's tonemap pass writes are read by nothing later in the frame: the accumulated bloom mip 0
 * (Renderer/shaders/tonemap.comp:37-41 keeps it only as the bloom input of the same invocation) and, outside the
 * 256 x 256 centre crop that the average-luminance pass reads (VulkanRenderer.cpp:1165), the log-luminance image
 * (tonemap.comp:52). By default the HIP library does not store them (a third of the pass's memory traffic);
 * enable != 0 makes every later frame store both so that rdr_read_bloom(ctx, 0, ..) and rdr_read_log_luminance
 * work. The frame and its average luminance are the same either way. */
int rdr_keep_intermediates(rdr_ctx* ctx, uint32_t enable);
/* bloom mip `level` (0..5) decoded from B10G11R11 to RGB float32;
 * out holds (W>>level)*(H>>level)*3 floats. Level 0 needs rdr_keep_intermediates. */
int rdr_read_bloom(rdr_ctx* ctx, uint32_t level, float* out);
/* per-pixel log-luminance image written by the tonemap pass, W*H floats. Needs rdr_keep_intermediates. */
int rdr_read_log_luminance(rdr_ctx* ctx, float* out);
/* irradiance cube (level must be 0) or prefiltered cube level, face-major,
 * decoded to RGB float32: 6*res*res*3 floats with res = base>>level. */
int rdr_read_env_cube(rdr_ctx* ctx, int which /*0 env,1 irradiance,2 prefiltered*/, uint32_t level, float* out);
/* 4 cascade matrices (64 floats) + 4 split depths of the last frame
 * (CSMSplitShaderData, VulkanRenderer.h:82-85). */
int rdr_read_cascades(rdr_ctx* ctx, float* out_viewproj_4x16, float* out_depth_4);
/* model-space positions of primitive `primitive` of mesh `mesh_id` as the next frame will rasterise them (after
 * skinning): count*3 floats; out may be NULL to query the count. */
int rdr_read_primitive_positions(rdr_ctx* ctx, uint32_t mesh_id, uint32_t primitive, float* out_xyz, uint32_t* out_count);
/* the installed BRDF table as RGBA8, row 0 first; out may be NULL to query the size. */
int rdr_read_brdf_lut(rdr_ctx* ctx, uint8_t* out, uint32_t* out_w, uint32_t* out_h);
/* mip `level` of image `image` as stored, for mip-chain parity: RGBA8 (4 bytes per texel), or for an image of one of
 * the SFLOAT formats one float[4] per texel (16 bytes; the 16-bit formats hold values representable in binary16). */
int rdr_read_image_level(rdr_ctx* ctx, uint32_t image, uint32_t level, uint8_t* out, uint32_t* out_w, uint32_t* out_h);

/* ---- stand-alone post-process entry point ------------------------------ */

/* Runs only the tail of renderLoop on a caller-supplied HDR image:
 * bloom chain (VulkanRendererBloom.cpp:62-100) + tonemap.comp + luminance
 * (VulkanRenderer.cpp:1084-1109, 1156-1191). `hdr_rgba_f32` is W*H*4 host
 * floats (quantised to RGBA16F on upload, like the colour attachment). */
int rdr_post_process(rdr_ctx* ctx, const float* hdr_rgba_f32, float exposure, uint8_t* out_rgba8,
                     float* out_avg_log_luminance);

/* No counterpart in the reference. A variant of rdr_post_process for callers that cannot allocate device memory:
 * code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc. It calls no hipMalloc, hipFree
 * or other allocating routine: the bloom chain, the log-luminance image and the partial means of the luminance
 * reduction live in a fixed array inside the library's code object (about 21 MB per GPU, held until the process
 * ends), which is why the image is limited to RDR_NOALLOC_MAX_PIXELS pixels (RDR_ERR_CAPACITY beyond that).
 * A second limit applies to very narrow or very tall images: mip l is max(1, width >> l) x max(1, height >> l), so
 * the six mips of e.g. a 2 x 1044480 image hold more than 4/3 of mip 0. The six mips (each rounded up to a multiple
 * of four texels) plus second copies of mips 3 and 4 must fit 1.5 x RDR_NOALLOC_MAX_PIXELS texels, else
 * RDR_ERR_CAPACITY; any image with both sides >= 32 and at most RDR_NOALLOC_MAX_PIXELS pixels fits.
 *   hdr_rgba16f_device  width*height RGBA16F pixels in device memory, 16-byte aligned (the colour attachment as is)
 *   out_rgba8_device    width*height RGBA8 pixels in device memory, 8-byte aligned
 *   out_avg_device      one float in device memory: the average log-luminance of the centre 256x256 window
 *   stream              a hipStream_t, or NULL for the context's own stream
 * The call only enqueues on `stream` (synchronise that stream, or call rdr_synchronize when it is NULL); width and
 * height need not be the context's. Gamma is the context's setting. All calls on one GPU share the array, which keeps
 * the contents of the last call: calls that overlap in time on one GPU (two streams, two contexts, two threads) race.
 * Results are bit for bit those of rdr_post_process on the same image. */
#define RDR_NOALLOC_MAX_PIXELS (1920u * 1088u)
int rdr_post_process_noalloc(rdr_ctx* ctx, const void* hdr_rgba16f_device, uint32_t width, uint32_t height, float exposure,
                             uint8_t* out_rgba8_device, float* out_avg_device, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RDR_H */
