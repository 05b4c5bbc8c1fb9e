// C ABI of include/rdr.h on HIP: resource management, the host side of
// renderLoop (matrices, culling, draw ordering, shadow-view set-up, exposure)
// and the launch sequence of one frame.
//
// Reference for the host logic: Renderer/VulkanRenderer.cpp:929-1111
// (shouldCullMesh, drawMeshes, renderLoop), Renderer/VulkanRendererShadow.cpp:
// 152-175 (cascade splits), :255-389 (point shadow maps), :391-456 (cascades).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <functional>
#include <new>
#include <string>
#include <vector>

#include "../../include/rdr.h"
#include "kernels.h"
#include "../../include/rdr_math.h"

using namespace rdr;

namespace {

// d_counters: 64 bytes of scalar counters, the spread counters, then (from RDR_COUNTER_LISTS on, never read back) the fill
// counts of the opaque pass's fragment lists and the chunk counters of the eight dies, 128 bytes apart. h_read (pinned)
// holds three copies of the first two parts:
// [s * RDR_COUNTER_BYTES, +RDR_COUNTER_BYTES) for s = 0, 1 is what the end of a frame in slot s reads back (average
// luminance in word RDR_COUNTER_LUM, spread counters from byte 64), [RDR_READ_MID, +RDR_COUNTER_BYTES) what a batch
// reads back between its pair scan and its sort.
#define RDR_COUNTER_BYTES (64u + RDR_BINNED_SLOTS * 128u)
#define RDR_COUNTER_LISTS RDR_COUNTER_BYTES  // byte offset of the list counters inside a slot's block
#define RDR_COUNTER_SLOT_BYTES (RDR_COUNTER_BYTES + (RDR_FRAG_SEGS + 8u) * RDR_FRAG_CTR_STRIDE * 4u)  // what a frame zeroes
#define RDR_COUNTER_LUM 4u  // word of the scalar counters that the luminance reduction writes its result to
#define RDR_CHAINED_MAX (1u << 21)   // elements up to which the one-launch scan and sort passes are used (see bin_batch)
#define RDR_COUNTER_SORT_FAILED 6u  // word the pair sort raises when one of its blocks gave up waiting for another
#define RDR_COUNTER_FRAGS 5u  // word the blended pass counts its fragment-list slots in (zero after the frame's fill)
#define RDR_FRAME_SLOTS 3  // frames that can be in flight: each has its own buffer set, events, counters and read-back block
#define RDR_READ_MID ((unsigned)RDR_FRAME_SLOTS * RDR_COUNTER_BYTES)
#define RDR_READ_BYTES ((unsigned)(RDR_FRAME_SLOTS + 1) * RDR_COUNTER_BYTES)

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes, bool keep = false, hipStream_t s = nullptr) {
        if (bytes <= cap) return hipSuccess;
        size_t want = std::max(bytes, cap + cap / 2);
        want = (want + 255) & ~(size_t)255;
        void* q = nullptr;
        hipError_t e = hipMalloc(&q, want);
        if (e != hipSuccess) return e;
        if (p) {
            if (keep) { e = hipMemcpyAsync(q, p, cap, hipMemcpyDeviceToDevice, s); if (e == hipSuccess) e = hipStreamSynchronize(s); }
            (void)hipFree(p);
            if (e != hipSuccess) { (void)hipFree(q); p = nullptr; cap = 0; return e; }
        }
        p = q; cap = want;
        return hipSuccess;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() const { return (T*)p; }
};

// A typed window into somebody else's DevBuf: several small tables that are uploaded, cleared or read back together
// live in one allocation, so that they move with one copy / one fill instead of one each (every small copy or fill is
// its own command on the stream, a few microseconds the kernels behind it wait for).
struct DevSpan {
    void* p = nullptr;
    template <class T> T* as() const { return (T*)p; }
};

struct PinBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
        const size_t want = std::max(bytes, (size_t)4096) * 2;
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() const { return (T*)p; }
};

struct HostImage {
    DevBuf pixels;
    uint32_t w = 0, h = 0, levels = 0;
    bool srgb = false;
    bool is_float = false;   // an SFLOAT format of Image.h: one float4 per texel instead of RGBA8
    bool half_texels = false;  // ... whose texels are binary16 in the image (every level holds values rounded to it)
    uint32_t level_offset[RDR_MAX_TEX_LEVELS] = {0};
};
struct HostTexture { uint32_t image; rdr_sampler sampler; };
struct HostPrim {
    uint32_t first_vertex, first_index, tri_count, n_vertices;
    vec3 bb_min, bb_max, barycenter;
    rdr_material mat;
    uint32_t mesh;
    bool skinned = false;
    uint32_t skin_first = 0;  // first entry of this primitive in the skin arrays
    vec3 rest_bb_min, rest_bb_max;
};
struct HostMesh {
    mat4 model; bool is_static;
    uint32_t first_prim = 0, prim_count = 0;
    uint32_t n_joints = 0;       // largest joint index in use + 1; 0 = not skinned
    uint32_t palette_base = 0;   // first matrix of this mesh in the frame's palette
    std::vector<float> joints;   // n_joints x 16, column-major; identity until rdr_set_mesh_joints
};
struct HostLight { vec3 point, intensity; uint32_t flags; uint16_t slot; };

// One batch of views binned, sorted and rasterised together. Its draws come either from the host (`draws`, filled by
// append_draws) or from the draw-list kernels (`lists`, one per view that has candidates); bin_batch leaves in
// `info[v]` where the draws and triangle instances of view v sit in the batch's tables on either path.
struct Batch {
    std::vector<DView> views;
    std::vector<DDraw> draws;
    std::vector<DDrawList> lists;
    std::vector<DDrawListInfo> info;
    uint32_t views_dropped = 0;  // views whose list came out empty and that are therefore not rendered (DDrawList::drop_empty)
};

// Scratch of rdr_post_process_noalloc: bloom mips 0..5, the log-luminance image and the 16x16 partial means of the
// luminance reduction for an image of at most RDR_NOALLOC_MAX_PIXELS pixels. Every GPU the library is used on holds
// its own copy of this array from the moment the code object is loaded until the process ends. In host code the
// name only stands for the symbol: the address on a device comes from hipGetSymbolAddress (rdr_create).
#define NOALLOC_BLOOM_WORDS (RDR_NOALLOC_MAX_PIXELS + RDR_NOALLOC_MAX_PIXELS / 2u)  // sum of the six mips < 4/3 of mip 0; ragged sizes round down
#define NOALLOC_SCRATCH_BYTES ((size_t)(NOALLOC_BLOOM_WORDS + RDR_NOALLOC_MAX_PIXELS + 512u) * 4u)
static __device__ __align__(256) unsigned char scratch[NOALLOC_SCRATCH_BYTES];

}  // namespace

struct rdr_ctx {
    rdr_settings settings;
    uint32_t W = 0, H = 0;
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;

    // scene, host side
    std::vector<std::vector<uint8_t>> buffers;
    std::vector<HostImage> images;
    std::vector<HostTexture> textures;
    std::vector<HostPrim> prims;
    std::vector<HostMesh> meshes;
    std::vector<float> h_pos;          // float4 per vertex
    std::vector<DVertexAttr> h_attr;
    std::vector<uint32_t> h_idx;
    // skinned vertices: arena index + owning mesh, joints (4 x u16) and weights (4 x f32); the arena keeps the rest pose on the host
    std::vector<uint32_t> h_skin_dst, h_skin_mesh, h_skin_joints;
    std::vector<float> h_skin_weights;
    uint32_t palette_matrices = 0;
    bool geometry_dirty = true, textures_dirty = true, skin_dirty = false;
    Camera camera;
    std::vector<HostLight> lights;
    rdr_directional_light dir_light;
    bool lights_set = false;
    std::vector<float> cascade_depths, cascade_camera_depths;
    mat4 csm_viewproj[4];

    // scene, device side
    DevBuf d_pos, d_attr, d_idx, d_prims, d_textures, d_tables, d_trig;
    DevBuf d_rest_pos, d_rest_attr, d_skin_joints, d_skin_weights, d_skin_dst_pal, d_palette;
    DevBuf d_chunk_bb;  // bounding boxes of the chunks of RDR_CHUNK_TRIS triangles (GPU-side culling), see upload_scene
    PinBuf h_palette;
    DevBuf d_env, d_irradiance, d_prefiltered, d_lut;
    DCube env{}, irradiance{}, prefiltered{};
    // the three cubes again with a one-texel border per face (k_cube_border): what the shading kernels filter from
    DevBuf d_env_b, d_irradiance_b, d_prefiltered_b;
    DCube env_b{}, irradiance_b{}, prefiltered_b{};
    // ... and footprint-major copies of the two the fragments look up in scattered directions, and of the BRDF table
    // (k_cube_quads, k_lut_quads): one 16-byte entry per 2x2 footprint
    DevBuf d_irradiance_q, d_prefiltered_q, d_lut_q;
    DCube irradiance_q{}, prefiltered_q{};
    uint32_t lut_w = 0, lut_h = 0;

    // frame targets
    bool float_images = false;  // an image of an SFLOAT format exists: the shading kernels are launched in the build that samples those
    DevBuf d_depth, d_vis, d_samples, d_hdr, d_bloom, d_out, d_loglum, d_lum, d_dir_shadow, d_point_shadow, d_point_static;
    uint32_t bloom_offset[6], bloom_w[6], bloom_h[6], bloom_fin_offset[2];
    bool keep_intermediates = false;  // rdr_keep_intermediates: the tonemap kernel also stores mip 0 and the whole log-luminance image
    bool post_kept = false;           // ... and did so in the last post-process pass
    bool bloom_fin_separate = false;  // the last chain left the accumulated mips 3 and 4 at bloom_fin_offset
    bool have_frame = false, have_post = false;

    // raster scratch
    DevBuf d_chunk_flags;
    DevBuf d_batch_tables;  // views | draws | draw_first | first_tri | chunk_first of the batch being rasterised, one upload
    DevSpan d_views, d_draws, d_draw_first, d_first_tri, d_chunk_first;
    DevBuf d_recs, d_pair_count, d_scan, d_hist;
    DevBuf d_tile_start;    // tile_start[n_tiles] | tile_end[n_tiles], one fill
    DevSpan d_tile_end;
    DevBuf d_counters;      // per frame slot: 64 bytes of scalar counters | RDR_BINNED_SLOTS x 128 bytes of spread counters, one fill, one read-back
    DevBuf d_frame;         // DFrame | point lights, one upload
    DevSpan d_lights;
    DevBuf d_keys[2], d_vals[2];
    DevBuf d_shade_tris;  // DShadeTri per triangle of the view being shaded
    PinBuf h_stage, h_read;
    // Draw lists built on the GPU (kernels_draws.hip) once a scene has enough primitives for the host loop over
    // views x primitives to matter; RDR_DEVICE_DRAW_LISTS=0/1 forces either path (tests run both).
    bool device_draws = false;       // this frame's choice
    uint64_t cull_tables_frame = 0;  // frame the tables below were last uploaded for
    DevBuf d_prim_cull, d_models, d_cand, d_draw_scratch;
    PinBuf h_cull;
    uint32_t cand_first[4] = {0, 0, 0, 0}, cand_count[4] = {0, 0, 0, 0};  // opaque, non-opaque, static, dynamic primitives in d_cand
    uint64_t cand_tris[5] = {0, 0, 0, 0, 0};                                // triangles of those four lists and of all primitives
    uint32_t batch_instances = 0;   // triangle instances of the last raster batch
    uint64_t blend_pairs = 0, blend_box_pixels = 0;  // of the last batch: (triangle, tile) pairs / box pixels of blended-view triangles
    DevBuf d_frags, d_frag_col, d_frag_head, d_tile_slots;
    // opaque pass: fragment records and colours (one list per screen rectangle), per-pixel state, first fragment per tile
    DevBuf d_ofrag_recs, d_ofrag_cols, d_pixel_state, d_tile_first;
    DevBuf d_sort_work;      // digit counts + status words of the pair sort (launch_radix_sort_onesweep)
    DevBuf d_scan_status;    // status words of the one-launch scans: zero between launches (launch_exclusive_scan_chained)
    uint32_t sort_epoch = 0; // its running pass number
    bool sort_failed = false;
    // The last frame's 4x rasteriser stored depth only in the tiles the blended pass works on (nothing else in a frame
    // reads the main view's depth): rdr_read_depth first runs the rasteriser over the other tiles again, from the
    // frame's triangle records and sorted tile lists, which stay as they are until the slot's next frame.
    bool depth_partial = false;
    DView depth_view{};
    RasterRange depth_range{};
    int depth_cur = 0;
    uint32_t depth_blend_base = 0xFFFFFFFFu;
    bool vis_in_lists = false;  // the last frame's visibility lives in its fragment lists (rdr_read_visibility unpacks it)
    uint32_t shade_blocks = 1024;  // grid of the fragment shading kernel: four 256-thread blocks per compute unit
    uint64_t blend_budget_slots = 48ull << 20;  // fragment-list slots (32 B each) a frame may use at once; RDR_BLEND_BUDGET_SLOTS overrides (tests)  // the count of used fragment slots is word RDR_COUNTER_FRAGS of d_counters

    // Shadow views whose inputs did not change are not rendered again. `geometry_version` counts every call that can
    // move or add geometry (rdr_add_mesh, rdr_set_mesh_transform, rdr_set_mesh_joints), `lights_version` every
    // rdr_set_lights. A cascade array (one per frame slot) remembers the version and the cascade matrices it was
    // rendered with; the point-light faces remember the version, the lights and the slot assignment of the last pass
    // over the moving primitives.
    uint64_t geometry_version = 1, lights_version = 1;
    struct ShadowKey { bool valid = false; uint64_t geometry = 0, lights = 0, hash = 0; };
    ShadowKey cascades_key[RDR_FRAME_SLOTS], point_dynamic_key;
    uint32_t views_reused = 0;          // shadow views of the frame being enqueued that were left as they are
    bool point_borders_stale = false;   // a face of a point-light cube was (or is about to be) re-rendered: the borders follow
    uint32_t point_cubes_in_use = 0;
    unsigned long long frame_pairs_before = 0;  // (triangle, tile) pairs of the batches of this frame binned so far
    double host_wait_s = 0.0;           // time the current rdr_render_frame call has spent blocked on the GPU
    // RDR_TRACE_FRAME=1: where a frame's time goes, printed to stderr when the frame is picked up. Device times (events
    // behind set-up and behind the front end, beside the pass events) and the host's clock at the same points of the call.
    bool trace = false;
    hipEvent_t tr_ev[RDR_FRAME_SLOTS][2] = {};
    double tr_host[RDR_FRAME_SLOTS][6] = {};
    float temporal_luminance = 0.0f, last_avg_log_lum = 0.0f;
    // Two frames are in flight, three on request: a frame rendered into a device destination is only enqueued, and the
    // next call (the next two, with `lag` = 2) builds and enqueues its own while the GPU still works on this one. Each
    // keeps its own events and its own block of the pinned read-back area until its results are picked up (harvest()).
    rdr_frame_stats stats{};       // host-side counters of the frame being enqueued
    rdr_frame_stats done_stats{};  // the newest frame whose results have been picked up: what rdr_get_frame_stats returns
    struct PendingFrame { bool on = false; rdr_frame_stats host{}; } pending[RDR_FRAME_SLOTS];
    uint64_t frames = 0;           // frames enqueued so far
    int slot = 0;                  // frame slot of the frame being enqueued (frames enqueued before it, modulo RDR_FRAME_SLOTS)
    bool frame_claimed = false;    // the current rdr_render_frame call got as far as taking `slot` and a frame index for its frame
    bool tail_split = true;        // resolve and blended pass on the tail stream (RDR_TAIL_ON_SHADING_STREAM=1: on the shading stream)
    int lag = 1;                   // a call returns once the frame `lag` calls back has finished and been picked up (RDR_FRAMES_IN_FLIGHT - 1)
    hipEvent_t ev_sets[RDR_FRAME_SLOTS][10] = {};
    hipEvent_t* ev = ev_sets[0];   // the events of the frame being enqueued
    hipEvent_t ev_pairs = nullptr;  // the pair count of the batch being binned has reached the host
    // The two tile rasterisers of a batch (single-sampled shadow views, 4x view) are independent and neither fills the
    // GPU's issue slots (their tiles mostly wait): the 4x one runs on a second stream beside the other.
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_raster1 = nullptr;  // the rasteriser launches of `stream` of the last batch are done
    hipEvent_t ev_vertex_in = nullptr, ev_vertex_out = nullptr;  // the main view's vertex stage (k_shade_setup on stream3): its inputs are there / it is done
    bool raster_side = false;         // ... and its 4x rasteriser ran on stream3 (ev_join)
    hipStream_t raster_stream = nullptr;  // stream the shadow-view rasteriser of the last batch was enqueued on
    // Geometry and shading of consecutive frames overlap: `stream` carries the geometry phase of a frame (set-up, binning,
    // sort; `stream3` takes the 4x rasteriser), `stream2` the shadow-view rasteriser and the shading kernel, `stream4`
    // what follows it (resolve, blended pass, post chain). What the geometry phase writes and the shading reads exists twice (`alt` holds the set of the
    // other frame slot; the two are swapped at the start of every frame), so the geometry phase of frame k+1 runs
    // while frame k is shaded. The point-light shadow maps exist once (a face nothing moves in keeps what it held):
    // their rasteriser launch waits for the previous frame's shading (ev_shaded).
    hipStream_t stream3 = nullptr;
    hipStream_t stream4 = nullptr;  // the tail of a frame (resolve, blended pass, bloom chain, tonemap, luminance, read-backs), beside the next frame's rasterisers and shading
    hipEvent_t ev_shaded[RDR_FRAME_SLOTS] = {};
    struct AltBuffers { DevBuf batch_tables, frame, shade_tris, recs, keys[2], vals[2], tile_start, depth, vis, dir_shadow, hdr, ofrag_recs, pixel_state, tile_first; } alt[RDR_FRAME_SLOTS - 1];
    PinBuf h_frame;                 // staging of DFrame + lights, one half per frame slot (h_stage is busy with the batch tables at that point)
    unsigned char* noalloc_scratch = nullptr;  // address of `scratch` on this context's device
    float precompute_ms[4] = {0.0f, 0.0f, 0.0f, 0.0f};  // device time of the last conversion to cube faces, irradiance bake, prefilter bake, BRDF table
};

namespace {

// measurement switches: set and not 0
bool env_on(const char* name) { const char* v = getenv(name); return v && atoi(v) != 0; }

double host_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int fail(rdr_ctx* c, int code, const std::string& msg) { if (c) c->err = msg; return code; }
int hip_fail(rdr_ctx* c, const char* what, hipError_t e) {
    char buf[512];
    snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    return fail(c, RDR_ERR_DEVICE, buf);
}
#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return hip_fail(c, #expr, e_); } while (0)
#define RDR_TRY(expr) do { int r_ = (expr); if (r_ != RDR_OK) return r_; } while (0)

// counters block of the frame being enqueued (each frame slot has its own: the tail of a frame still
// writes its block while the next frame has cleared and is filling the other)
uint32_t* frame_counters(const rdr_ctx* c) { return (uint32_t*)(c->d_counters.as<uint8_t>() + (size_t)c->slot * RDR_COUNTER_SLOT_BYTES); }

double srgb_eotf(double v) { return v <= 0.04045 ? v / 12.92 : pow((v + 0.055) / 1.055, 2.4); }

// Renderer/Camera.cpp:245-280, planes pointing into the frustum
void frustum_planes(const mat4& M, vec4 pl[6]) {
    pl[0] = {M.at(0, 3) + M.at(0, 0), M.at(1, 3) + M.at(1, 0), M.at(2, 3) + M.at(2, 0), M.at(3, 3) + M.at(3, 0)};
    pl[1] = {M.at(0, 3) - M.at(0, 0), M.at(1, 3) - M.at(1, 0), M.at(2, 3) - M.at(2, 0), M.at(3, 3) - M.at(3, 0)};
    pl[2] = {M.at(0, 3) + M.at(0, 1), M.at(1, 3) + M.at(1, 1), M.at(2, 3) + M.at(2, 1), M.at(3, 3) + M.at(3, 1)};
    pl[3] = {M.at(0, 3) - M.at(0, 1), M.at(1, 3) - M.at(1, 1), M.at(2, 3) - M.at(2, 1), M.at(3, 3) - M.at(3, 1)};
    pl[4] = {M.at(0, 3) - M.at(0, 2), M.at(1, 3) - M.at(1, 2), M.at(2, 3) - M.at(2, 2), M.at(3, 3) - M.at(3, 2)};
    pl[5] = {M.at(0, 2), M.at(1, 2), M.at(2, 2), M.at(3, 2)};
}
// Renderer/VulkanRenderer.cpp:929-943
bool should_cull(const HostPrim& p, const mat4& model, const mat4& viewproj) {
    vec4 pl[6];
    frustum_planes(mat_mul(viewproj, model), pl);
    for (int i = 0; i < 6; i++) {
        vec3 n = p.bb_min;
        if (pl[i].x >= 0.0f) n.x = p.bb_max.x;
        if (pl[i].y >= 0.0f) n.y = p.bb_max.y;
        if (pl[i].z >= 0.0f) n.z = p.bb_max.z;
        if (dot(n, make3(pl[i].x, pl[i].y, pl[i].z)) + pl[i].w < 0.0f) return true;
    }
    return false;
}

enum SortMode { SORT_NONE, SORT_FRONT_TO_BACK, SORT_BACK_TO_FRONT };

// drawMeshes (VulkanRenderer.cpp:945-999): sort by distance(barycenter, eye), frustum-cull, one draw per primitive.
void append_draws(const rdr_ctx* c, const std::vector<uint32_t>& prim_list, const mat4& viewproj, vec3 eye, SortMode mode, uint32_t view,
                  std::vector<DDraw>& out) {
    std::vector<uint32_t> order(prim_list);
    if (mode != SORT_NONE) {
        std::vector<float> key(c->prims.size());
        for (uint32_t p : order) key[p] = distance(c->prims[p].barycenter, eye);
        if (mode == SORT_FRONT_TO_BACK) std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return key[a] < key[b]; });
        else std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return key[a] > key[b]; });
    }
    uint32_t first_tri = 0;
    for (uint32_t p : order) {
        const HostPrim& pr = c->prims[p];
        const mat4& model = c->meshes[pr.mesh].model;
        if (should_cull(pr, model, viewproj)) continue;
        DDraw d;
        const mat4 mvp = mat_mul(viewproj, model);
        memcpy(d.mvp, mvp.m, 64);
        memcpy(d.model, model.m, 64);
        d.view = view; d.prim = p; d.first_instance = 0; d.first_tri_id = first_tri;
        first_tri += pr.tri_count;
        out.push_back(d);
    }
}

uint64_t hash_words(uint64_t h, const void* data, size_t bytes) {  // 64-bit FNV-1a over 32-bit words
    const uint32_t* w = (const uint32_t*)data;
    for (size_t i = 0; i < bytes / 4; i++) { h ^= w[i]; h *= 1099511628211ull; }
    return h;
}

enum CandKind { CAND_OPAQUE = 0, CAND_NON_OPAQUE = 1, CAND_STATIC = 2, CAND_DYNAMIC = 3, CAND_EVERY = 4 };

// The draws of one view: appended on the host, or left to the draw-list kernels as one list of candidates (`kind`
// names the same primitives as `prim_list`). `view` must be the index the view has, or is about to get, in b.views.
void add_view_draws(rdr_ctx* c, Batch& b, CandKind kind, const std::vector<uint32_t>& prim_list, const mat4& viewproj, vec3 eye, SortMode mode,
                    uint32_t view, bool drop_empty) {
    if (!c->device_draws) { append_draws(c, prim_list, viewproj, eye, mode, view, b.draws); return; }
    DDrawList l;
    memset(&l, 0, sizeof l);
    memcpy(l.viewproj, viewproj.m, 64);
    l.eye[0] = eye.x; l.eye[1] = eye.y; l.eye[2] = eye.z;
    l.sort = mode == SORT_FRONT_TO_BACK ? 1u : (mode == SORT_BACK_TO_FRONT ? 2u : 0u);
    l.cand_first = kind == CAND_EVERY ? RDR_CAND_ALL : c->cand_first[kind];
    l.cand_count = kind == CAND_EVERY ? (uint32_t)c->prims.size() : c->cand_count[kind];
    l.view = view;
    l.drop_empty = drop_empty ? 1u : 0u;
    l.kind = (uint32_t)kind;
    b.lists.push_back(l);
}

// Tables the draw-list kernels read: box / barycentre / size of every primitive, the model matrix of every mesh and
// the four candidate lists. Rebuilt once per frame (boxes of skinned primitives and model matrices move).
int upload_cull_tables(rdr_ctx* c) {
    if (c->cull_tables_frame == c->frames) return RDR_OK;
    c->cull_tables_frame = c->frames;
    const size_t np = c->prims.size(), nm = c->meshes.size();
    auto round16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const size_t sz_p = round16(np * sizeof(DPrimCull)), sz_m = round16(nm * 64), sz_c = round16(2 * np * 4);
    HIP_TRY(c->h_cull.ensure(sz_p + sz_m + sz_c + 64));
    uint8_t* st = c->h_cull.as<uint8_t>();
    DPrimCull* pc = (DPrimCull*)st;
    float* mm = (float*)(st + sz_p);
    uint32_t* cand = (uint32_t*)(st + sz_p + sz_m);
    uint32_t n_kind[4] = {0, 0, 0, 0};
    for (int k = 0; k < 5; k++) c->cand_tris[k] = 0;
    for (size_t p = 0; p < np; p++) {
        const HostPrim& hp = c->prims[p];
        DPrimCull& d = pc[p];
        d.bb_min[0] = hp.bb_min.x; d.bb_min[1] = hp.bb_min.y; d.bb_min[2] = hp.bb_min.z;
        d.bb_max[0] = hp.bb_max.x; d.bb_max[1] = hp.bb_max.y; d.bb_max[2] = hp.bb_max.z;
        d.barycenter[0] = hp.barycenter.x; d.barycenter[1] = hp.barycenter.y; d.barycenter[2] = hp.barycenter.z;
        d.mesh = hp.mesh; d.tri_count = hp.tri_count; d.n_chunks = (hp.tri_count + RDR_CHUNK_TRIS - 1u) / RDR_CHUNK_TRIS;
        n_kind[hp.mat.alpha_mode == RDR_ALPHA_OPAQUE ? CAND_OPAQUE : CAND_NON_OPAQUE]++;
        n_kind[c->meshes[hp.mesh].is_static ? CAND_STATIC : CAND_DYNAMIC]++;
    }
    // opaque | non-opaque share the first np entries, static | dynamic the second np
    c->cand_first[CAND_OPAQUE] = 0; c->cand_first[CAND_NON_OPAQUE] = n_kind[CAND_OPAQUE];
    c->cand_first[CAND_STATIC] = (uint32_t)np; c->cand_first[CAND_DYNAMIC] = (uint32_t)np + n_kind[CAND_STATIC];
    uint32_t fill[4] = {0, 0, 0, 0};
    for (size_t p = 0; p < np; p++) {
        const HostPrim& hp = c->prims[p];
        const int a = hp.mat.alpha_mode == RDR_ALPHA_OPAQUE ? CAND_OPAQUE : CAND_NON_OPAQUE, m = c->meshes[hp.mesh].is_static ? CAND_STATIC : CAND_DYNAMIC;
        cand[c->cand_first[a] + fill[a]++] = (uint32_t)p;
        cand[c->cand_first[m] + fill[m]++] = (uint32_t)p;
        c->cand_tris[a] += hp.tri_count; c->cand_tris[m] += hp.tri_count; c->cand_tris[CAND_EVERY] += hp.tri_count;
    }
    for (int k = 0; k < 4; k++) c->cand_count[k] = n_kind[k];
    for (size_t m = 0; m < nm; m++) memcpy(mm + m * 16, c->meshes[m].model.m, 64);
    HIP_TRY(c->d_prim_cull.ensure(std::max<size_t>(sz_p, 16)));
    HIP_TRY(c->d_models.ensure(std::max<size_t>(sz_m, 16)));
    HIP_TRY(c->d_cand.ensure(std::max<size_t>(sz_c, 16)));
    if (sz_p) HIP_TRY(hipMemcpyAsync(c->d_prim_cull.p, pc, sz_p, hipMemcpyHostToDevice, c->stream));
    if (sz_m) HIP_TRY(hipMemcpyAsync(c->d_models.p, mm, sz_m, hipMemcpyHostToDevice, c->stream));
    if (sz_c) HIP_TRY(hipMemcpyAsync(c->d_cand.p, cand, sz_c, hipMemcpyHostToDevice, c->stream));
    return RDR_OK;
}

// VulkanRendererShadow.cpp:152-175
void compute_cascade_splits(rdr_ctx* c) {
    c->cascade_depths.clear(); c->cascade_camera_depths.clear();
    c->cascade_depths.push_back(c->camera.near_plane);
    c->cascade_camera_depths.push_back(0.0f);
    const float n = c->camera.near_plane, f = c->camera.far_plane;
    const int levels = (int)c->settings.directional_shadow_cascade_levels;
    for (int i = 1; i < levels; i++) {
        const float N = (float)levels;
        const float Cuni = n + (f - n) * i / N;
        const float Clog = n * powf(f / n, i / N);
        const float lambda = 0.9f;
        const float d = lambda * Clog + (1 - lambda) * Cuni;
        c->cascade_depths.push_back(d);
        c->cascade_camera_depths.push_back((f - f * n / d) / (f - n));
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
float half_bits_to_float(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1Fu, m = h & 0x3FFu;
    uint32_t bits;
    if (e == 0) { float v = (float)m * 5.9604644775390625e-08f; return sign ? -v : v; }
    if (e == 31) bits = sign | 0x7F800000u | (m << 13);
    else bits = sign | ((e + 112u) << 23) | (m << 13);
    float f; memcpy(&f, &bits, 4); return f;
}
float component_as_float(const uint8_t* p, uint32_t value_type, bool normalized) {
    switch (value_type) {
        case RDR_FLOAT: { float f; memcpy(&f, p, 4); return f; }
        case RDR_UINT8: return normalized ? (float)p[0] / 255.0f : (float)p[0];
        case RDR_UINT16: { uint16_t v; memcpy(&v, p, 2); return normalized ? (float)v / 65535.0f : (float)v; }
        case RDR_HALF: { uint16_t v; memcpy(&v, p, 2); return half_bits_to_float(v); }
        default: return 0.0f;
    }
}

// Re-poses the skinned vertices of the arena from the current palettes (enqueued on the context's stream).
int run_skinning(rdr_ctx* c) {
    const size_t ns = c->h_skin_dst.size();
    if (!ns || !c->skin_dirty) return RDR_OK;
    const size_t bytes = (size_t)c->palette_matrices * 64;
    // the pinned block is free to rewrite: every caller synchronises the stream after this copy and before it can
    // come back here (the frame does when it reads the pair count back, the read-back before its copy)
    HIP_TRY(c->h_palette.ensure(bytes));
    HIP_TRY(c->d_palette.ensure(bytes));
    for (const HostMesh& m : c->meshes)
        if (m.n_joints) memcpy(c->h_palette.as<float>() + (size_t)m.palette_base * 16, m.joints.data(), (size_t)m.n_joints * 64);
    HIP_TRY(hipMemcpyAsync(c->d_palette.p, c->h_palette.p, bytes, hipMemcpyHostToDevice, c->stream));
    launch_skin(c->stream, c->d_rest_pos.p, c->d_rest_attr.p, c->d_skin_joints.p, c->d_skin_weights.p, c->d_skin_dst_pal.p, c->d_palette.as<float>(),
                c->d_pos.p, c->d_attr.p, (uint32_t)ns);
    c->skin_dirty = false;
    return RDR_OK;
}

int upload_scene(rdr_ctx* c) {
    if (c->geometry_dirty) {
        HIP_TRY(c->d_pos.ensure(std::max<size_t>(c->h_pos.size() * 4, 16)));
        HIP_TRY(c->d_attr.ensure(std::max<size_t>(c->h_attr.size() * sizeof(DVertexAttr), 48)));
        HIP_TRY(c->d_idx.ensure(std::max<size_t>(c->h_idx.size() * 4, 16)));
        if (!c->h_pos.empty()) {
            HIP_TRY(hipMemcpy(c->d_pos.p, c->h_pos.data(), c->h_pos.size() * 4, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(c->d_attr.p, c->h_attr.data(), c->h_attr.size() * sizeof(DVertexAttr), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(c->d_idx.p, c->h_idx.data(), c->h_idx.size() * 4, hipMemcpyHostToDevice));
        }
        // boxes of every RDR_CHUNK_TRIS consecutive triangles of the rigid primitives, in model space (k_chunk_cull)
        std::vector<float> chunk_bb;
        std::vector<uint32_t> first_chunk(c->prims.size(), RDR_NO_CHUNKS);
        for (size_t i = 0; i < c->prims.size(); i++) {
            const HostPrim& p = c->prims[i];
            if (p.skinned) continue;
            first_chunk[i] = (uint32_t)(chunk_bb.size() / 8);
            for (uint32_t t0 = 0; t0 < p.tri_count; t0 += RDR_CHUNK_TRIS) {
                float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
                const uint32_t t1 = std::min(p.tri_count, t0 + RDR_CHUNK_TRIS);
                for (size_t k = (size_t)p.first_index + 3u * t0; k < (size_t)p.first_index + 3u * t1; k++) {
                    const float* v = &c->h_pos[((size_t)p.first_vertex + c->h_idx[k]) * 4];
                    for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], v[a]); hi[a] = std::max(hi[a], v[a]); }
                }
                const float box[8] = {lo[0], lo[1], lo[2], 0.0f, hi[0], hi[1], hi[2], 0.0f};
                chunk_bb.insert(chunk_bb.end(), box, box + 8);
            }
        }
        HIP_TRY(c->d_chunk_bb.ensure(std::max<size_t>(chunk_bb.size() * 4, 32)));
        if (!chunk_bb.empty()) HIP_TRY(hipMemcpy(c->d_chunk_bb.p, chunk_bb.data(), chunk_bb.size() * 4, hipMemcpyHostToDevice));
        std::vector<DPrim> dp(std::max<size_t>(c->prims.size(), 1));
        for (size_t i = 0; i < c->prims.size(); i++) {
            const HostPrim& p = c->prims[i];
            DPrim& d = dp[i];
            memset(&d, 0, sizeof d);
            d.first_vertex = p.first_vertex; d.first_index = p.first_index; d.tri_count = p.tri_count; d.first_chunk = first_chunk[i];
            memcpy(d.mat.base_color, p.mat.base_color_factor, 16);
            memcpy(d.mat.emissive, p.mat.emissive_factor, 16);
            d.mat.metallic = p.mat.metallic_factor; d.mat.roughness = p.mat.roughness_factor; d.mat.ao = p.mat.ao_factor;
            d.mat.alpha_cutoff = p.mat.alpha_cutoff; d.mat.alpha_mode = p.mat.alpha_mode;
            d.mat.tex[0] = p.mat.albedo_texture; d.mat.tex[1] = p.mat.normal_texture; d.mat.tex[2] = p.mat.metal_roughness_texture;
            d.mat.tex[3] = p.mat.ao_texture; d.mat.tex[4] = p.mat.emissive_texture;
        }
        HIP_TRY(c->d_prims.ensure(dp.size() * sizeof(DPrim)));
        HIP_TRY(hipMemcpy(c->d_prims.p, dp.data(), dp.size() * sizeof(DPrim), hipMemcpyHostToDevice));
        const size_t ns = c->h_skin_dst.size();
        if (ns) {
            // rest pose of the skinned vertices, gathered from the arena; palette slots in mesh order
            uint32_t base = 0;
            for (HostMesh& m : c->meshes) { m.palette_base = base; base += m.n_joints; }
            c->palette_matrices = base;
            std::vector<float> rp(ns * 4);
            std::vector<DVertexAttr> ra(ns);
            std::vector<uint32_t> dst_pal(ns * 2);
            for (size_t i = 0; i < ns; i++) {
                const uint32_t v = c->h_skin_dst[i];
                memcpy(&rp[i * 4], &c->h_pos[(size_t)v * 4], 16);
                ra[i] = c->h_attr[v];
                dst_pal[i * 2] = v; dst_pal[i * 2 + 1] = c->meshes[c->h_skin_mesh[i]].palette_base;
            }
            HIP_TRY(c->d_rest_pos.ensure(ns * 16));
            HIP_TRY(c->d_rest_attr.ensure(ns * sizeof(DVertexAttr)));
            HIP_TRY(c->d_skin_joints.ensure(ns * 8));
            HIP_TRY(c->d_skin_weights.ensure(ns * 16));
            HIP_TRY(c->d_skin_dst_pal.ensure(ns * 8));
            HIP_TRY(hipMemcpy(c->d_rest_pos.p, rp.data(), ns * 16, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(c->d_rest_attr.p, ra.data(), ns * sizeof(DVertexAttr), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(c->d_skin_joints.p, c->h_skin_joints.data(), ns * 8, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(c->d_skin_weights.p, c->h_skin_weights.data(), ns * 16, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(c->d_skin_dst_pal.p, dst_pal.data(), ns * 8, hipMemcpyHostToDevice));
            c->skin_dirty = true;
        }
        c->geometry_dirty = false;
    }
    if (c->textures_dirty) {
        std::vector<DTexture> dt(c->textures.size());
        for (size_t i = 0; i < c->textures.size(); i++) {
            const HostTexture& t = c->textures[i];
            const HostImage& img = c->images[t.image];
            DTexture& d = dt[i];
            memset(&d, 0, sizeof d);
            d.base = (const uint8_t*)img.pixels.p;
            d.w = img.w; d.h = img.h; d.levels = img.levels;
            const uint32_t nearest = (t.sampler.mag_filter == RDR_FILTER_NEAREST ? 1u : 0u) | (t.sampler.min_filter == RDR_FILTER_NEAREST ? 2u : 0u) |
                                     (t.sampler.mipmap_filter == RDR_FILTER_NEAREST ? 4u : 0u);
            d.flags = (img.srgb ? 1u : 0u) | (t.sampler.wrap_u << 4) | (t.sampler.wrap_v << 8) | (nearest << 12) | (img.is_float ? 1u << 16 : 0u);
            memcpy(d.level_offset, img.level_offset, sizeof d.level_offset);
        }
        HIP_TRY(c->d_textures.ensure(dt.size() * sizeof(DTexture)));
        HIP_TRY(hipMemcpy(c->d_textures.p, dt.data(), dt.size() * sizeof(DTexture), hipMemcpyHostToDevice));
        c->textures_dirty = false;
    }
    return RDR_OK;
}

// `texels`: level 0, RGBA8 (4 bytes per texel) or, with is_float, float4 (16 bytes per texel)
int create_image(rdr_ctx* c, const void* texels0, uint32_t w, uint32_t h, bool srgb, uint32_t max_mips, uint32_t* out_id, bool is_float = false,
                 bool half_texels = false) {
    HostImage img;
    img.w = w; img.h = h; img.srgb = srgb && !is_float; img.is_float = is_float; img.half_texels = half_texels;
    uint32_t levels = 1;
    const uint32_t m = std::max(w, h);
    if (m > 1) levels = (uint32_t)floor(log2((double)m)) + 1;  // VulkanRenderer.cpp:311-315
    levels = std::max(1u, std::min(std::min(levels, max_mips), (uint32_t)RDR_MAX_TEX_LEVELS));
    img.levels = levels;
    size_t texels = 0;
    for (uint32_t l = 0; l < levels; l++) {
        img.level_offset[l] = (uint32_t)texels;
        texels += (size_t)std::max(1u, w >> l) * std::max(1u, h >> l);
    }
    const size_t texel_bytes = is_float ? 16u : 4u;
    HIP_TRY(img.pixels.ensure(texels * texel_bytes + 16));  // the shading kernels read texel pairs: one texel of slack after the last level
    HIP_TRY(hipMemcpyAsync(img.pixels.p, texels0, (size_t)w * h * texel_bytes, hipMemcpyHostToDevice, c->stream));
    if (is_float) launch_mip_chain_f32(c->stream, img.pixels.as<float>(), img.level_offset, w, h, levels, half_texels);
    else {
        const float* tables = c->d_tables.as<float>();
        launch_mip_chain(c->stream, img.pixels.as<uint32_t>(), img.level_offset, w, h, levels, srgb ? 1 : 0, tables, tables + 256, tables + 512);
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipGetLastError());
    c->images.push_back(img);
    if (is_float) c->float_images = true;
    *out_id = (uint32_t)c->images.size() - 1;
    return RDR_OK;
}

// Picks up the results of the frame in `slot` (it must have finished on the GPU): luminance, device-side counters, pass
// times. Idempotent.
void harvest(rdr_ctx* c, int slot) {
    if (!c->pending[slot].on) return;
    rdr_frame_stats st = c->pending[slot].host;
    const uint8_t* rd = c->h_read.as<uint8_t>() + (size_t)slot * RDR_COUNTER_BYTES;
    memcpy(&c->last_avg_log_lum, rd + 4 * RDR_COUNTER_LUM, 4);
    uint32_t sort_word = 0; memcpy(&sort_word, rd + 4 * RDR_COUNTER_SORT_FAILED, 4);
    if (sort_word) c->sort_failed = true;
    uint32_t binned = 0, chunks_culled = 0;
    for (uint32_t k = 0; k < RDR_BINNED_SLOTS; k++) {
        const uint32_t* w = (const uint32_t*)(rd + 64 + k * 128);
        binned += w[0];
        chunks_culled += w[4];
    }
    st.triangles_binned = binned;
    st.chunks_culled = chunks_culled;
    st.avg_log_luminance = c->last_avg_log_lum;
    hipEvent_t* ev = c->ev_sets[slot];
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, ev[0], ev[1]) == hipSuccess) st.ms_geometry = ms;
    if (hipEventElapsedTime(&ms, ev[1], ev[2]) == hipSuccess) st.ms_shade = ms;
    if (hipEventElapsedTime(&ms, ev[2], ev[3]) == hipSuccess) st.ms_bloom = ms;
    if (hipEventElapsedTime(&ms, ev[3], ev[4]) == hipSuccess) st.ms_tonemap = ms;
    if (hipEventElapsedTime(&ms, ev[0], ev[5]) == hipSuccess) st.ms_total = ms;
    // ev[1] closes the geometry phase right behind the tile rasterisers (the shading stream waits for it); ev[6] and
    // ev[7] bracket the shading kernel on its own stream
    if (hipEventElapsedTime(&ms, ev[6], ev[7]) == hipSuccess) st.ms_shade_kernel = ms;
    if (hipEventElapsedTime(&ms, ev[8], ev[1]) == hipSuccess) st.ms_raster_kernel = ms;
    if (c->trace) {
        auto at = [&](hipEvent_t e) { float t = -1.0f; return hipEventElapsedTime(&t, ev[0], e) == hipSuccess ? t * 1e3f : -1.0f; };
        const int before = (slot + RDR_FRAME_SLOTS - 1) % RDR_FRAME_SLOTS;
        float since = -1.0f;
        if (hipEventElapsedTime(&since, c->ev_sets[before][0], ev[0]) != hipSuccess) since = -1.0f;
        const double* h = c->tr_host[slot];
        fprintf(stderr, "[rdr trace] frame start +%.0f us after the one before | device: set-up %.0f front end %.0f raster %.0f..%.0f shading %.0f..%.0f "
                        "post %.0f end %.0f | host: pair wait %.0f..%.0f front end enqueued %.0f all enqueued %.0f return %.0f\n",
                since * 1e3f, at(c->tr_ev[slot][0]), at(c->tr_ev[slot][1]), at(ev[8]), at(ev[1]), at(ev[6]), at(ev[7]), at(ev[2]), at(ev[5]),
                (h[1] - h[0]) * 1e6, (h[2] - h[0]) * 1e6, (h[3] - h[0]) * 1e6, (h[4] - h[0]) * 1e6, (h[5] - h[0]) * 1e6);
        (void)hipGetLastError();  // an event of the first frames that was never recorded
    }
    c->done_stats = st;
    c->pending[slot].on = false;
}

// The status words of the one-launch scans start out zero and every scan leaves them zero: cleared when the buffer grows.
int ensure_scan_status(rdr_ctx* c, size_t words) {
    if (c->d_scan_status.cap < words * 4) {
        HIP_TRY(c->d_scan_status.ensure(words * 4 + 4096));
        HIP_TRY(hipMemsetAsync(c->d_scan_status.p, 0, c->d_scan_status.cap, c->stream));
    }
    return RDR_OK;
}

// Runs set-up -> bin -> sort for one batch of views. On return the sorted
// triangle lists of every tile are in d_vals[*cur] / d_tile_start / d_tile_end.
int bin_batch(rdr_ctx* c, Batch& b, uint32_t* out_tiles, int* out_cur, const std::function<int()>& after_scan) {
    uint32_t n_tiles = 0;
    for (DView& v : b.views) {
        v.tiles_x = (v.w + RDR_TILE - 1) / RDR_TILE; v.tiles_y = (v.h + RDR_TILE - 1) / RDR_TILE;
        v.tile_base = n_tiles;
        n_tiles += v.tiles_x * v.tiles_y;
    }
    auto round16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
    uint32_t n_draws = 0, n_inst = 0, n_chunks = 0;  // n_chunks: one flag per RDR_CHUNK_TRIS triangle instances of a draw
    b.info.assign(b.views.size(), DDrawListInfo{0, 0, 0, 0});
    b.views_dropped = 0;
    if (!c->device_draws) {
        n_draws = (uint32_t)b.draws.size();
        std::vector<uint32_t> draw_first(n_draws + 1, 0), first_tri(n_draws + 1, 0), chunk_first(n_draws + 1, 0);
        uint64_t inst64 = 0;
        for (uint32_t i = 0; i < n_draws; i++) {
            DDraw& d = b.draws[i];
            DDrawListInfo& vi = b.info[d.view];
            if (vi.n_draws == 0) { vi.first_draw = i; vi.first_instance = (uint32_t)inst64; }
            vi.n_draws++;
            vi.n_tris += c->prims[d.prim].tri_count;
            d.first_instance = (uint32_t)inst64;
            draw_first[i] = (uint32_t)inst64;
            first_tri[i] = d.first_tri_id;
            chunk_first[i] = n_chunks;
            inst64 += c->prims[d.prim].tri_count;
            n_chunks += (c->prims[d.prim].tri_count + RDR_CHUNK_TRIS - 1u) / RDR_CHUNK_TRIS;
        }
        chunk_first[n_draws] = n_chunks;
        if (inst64 >= 0xFFFFFFF0ull) return fail(c, RDR_ERR_CAPACITY, "more than 2^32 triangle instances in one batch");
        n_inst = (uint32_t)inst64;
        draw_first[n_draws] = n_inst;
        // stage the tables in pinned memory and upload
        const size_t sz_views = round16(b.views.size() * sizeof(DView)), sz_draws = round16(std::max<size_t>(n_draws, 1) * sizeof(DDraw)),
                     sz_first = round16((n_draws + 1) * 4);
        const size_t sz_tables = sz_views + sz_draws + 3 * sz_first;
        HIP_TRY(c->h_stage.ensure(sz_tables + 64));
        uint8_t* st = c->h_stage.as<uint8_t>();
        memcpy(st, b.views.data(), b.views.size() * sizeof(DView));
        if (n_draws) memcpy(st + sz_views, b.draws.data(), n_draws * sizeof(DDraw));
        memcpy(st + sz_views + sz_draws, draw_first.data(), (n_draws + 1) * 4);
        memcpy(st + sz_views + sz_draws + sz_first, first_tri.data(), (n_draws + 1) * 4);
        memcpy(st + sz_views + sz_draws + 2 * sz_first, chunk_first.data(), (n_draws + 1) * 4);
        HIP_TRY(c->d_batch_tables.ensure(sz_tables));
        uint8_t* dt = c->d_batch_tables.as<uint8_t>();
        c->d_views.p = dt; c->d_draws.p = dt + sz_views; c->d_draw_first.p = dt + sz_views + sz_draws;
        c->d_first_tri.p = dt + sz_views + sz_draws + sz_first; c->d_chunk_first.p = dt + sz_views + sz_draws + 2 * sz_first;
        HIP_TRY(hipMemcpyAsync(dt, st, sz_tables, hipMemcpyHostToDevice, c->stream));
    } else {
        // the draw-list kernels fill the same tables; the host uploads views and lists and reads back the sizes
        const uint32_t n_lists = (uint32_t)b.lists.size();
        uint64_t slots64 = 0, tris64 = 0;
        uint32_t longest_sorted = 0;
        for (DDrawList& l : b.lists) {
            l.slot_base = (uint32_t)slots64;
            slots64 += l.cand_count;
            if (l.sort) longest_sorted = std::max(longest_sorted, l.cand_count);
            tris64 += c->cand_tris[l.kind];  // an upper bound: every candidate survives
        }
        if (slots64 >= 0xFFFFFFF0ull) return fail(c, RDR_ERR_CAPACITY, "more than 2^32 (view, primitive) candidates in one batch");
        if (tris64 >= 0xFFFFFFF0ull) return fail(c, RDR_ERR_CAPACITY, "more than 2^32 triangle instances in one batch");
        const uint32_t n_slots = (uint32_t)slots64;
        const size_t sz_views = round16(b.views.size() * sizeof(DView)), sz_lists = round16(std::max<size_t>(n_lists, 1) * sizeof(DDrawList)),
                     sz_draws = round16(std::max<size_t>(n_slots, 1) * sizeof(DDraw)), sz_first = round16(((size_t)n_slots + 1) * 4),
                     sz_info = round16(std::max<size_t>(n_lists, 1) * sizeof(DDrawListInfo)) + 64;
        const size_t sz_tables = sz_views + sz_lists + sz_draws + 3 * sz_first + sz_info;
        HIP_TRY(c->h_stage.ensure(sz_views + sz_lists + 64));
        uint8_t* st = c->h_stage.as<uint8_t>();
        memcpy(st, b.views.data(), b.views.size() * sizeof(DView));
        if (n_lists) memcpy(st + sz_views, b.lists.data(), n_lists * sizeof(DDrawList));
        HIP_TRY(c->d_batch_tables.ensure(sz_tables));
        uint8_t* dt = c->d_batch_tables.as<uint8_t>();
        c->d_views.p = dt;
        uint8_t* d_lists = dt + sz_views;
        c->d_draws.p = d_lists + sz_lists; c->d_draw_first.p = (uint8_t*)c->d_draws.p + sz_draws;
        c->d_first_tri.p = (uint8_t*)c->d_draw_first.p + sz_first; c->d_chunk_first.p = (uint8_t*)c->d_first_tri.p + sz_first;
        uint8_t* d_info = (uint8_t*)c->d_chunk_first.p + sz_first;                                   // n_lists records, then ...
        uint32_t* d_totals = (uint32_t*)(d_info + sz_info - 64);                                     // ... {draws, instances, chunks}
        HIP_TRY(hipMemcpyAsync(dt, st, sz_views + sz_lists, hipMemcpyHostToDevice, c->stream));
        std::vector<DDrawListInfo> info(n_lists);
        uint32_t totals[3] = {0, 0, 0};
        if (n_lists) {
            HIP_TRY(c->d_draw_scratch.ensure(draw_list_scratch_words(n_slots, longest_sorted) * 4));
            HIP_TRY(c->d_hist.ensure((sort_hist_words(std::max(longest_sorted, 1u)) + 16) * 4));
            HIP_TRY(c->d_scan.ensure((std::max(scan_scratch_words(std::max(n_slots, 1u)), scan_scratch_words((uint32_t)sort_hist_words(std::max(longest_sorted, 1u)))) + 16) * 4));
            launch_draw_lists(c->stream, b.lists.data(), n_lists, (const DDrawList*)d_lists, n_slots, c->d_prim_cull.as<DPrimCull>(), c->d_models.as<float>(),
                              c->d_cand.as<uint32_t>(), c->d_draw_scratch.as<uint32_t>(), c->d_hist.as<uint32_t>(), c->d_scan.as<uint32_t>(),
                              c->d_draws.as<DDraw>(), c->d_draw_first.as<uint32_t>(), c->d_first_tri.as<uint32_t>(), c->d_chunk_first.as<uint32_t>(),
                              (DDrawListInfo*)d_info, c->d_views.as<DView>(), d_totals);
            // the sizes of everything downstream: one small copy, and the host waits for it
            std::vector<uint8_t> back(sz_info);
            HIP_TRY(hipMemcpyAsync(back.data(), d_info, sz_info, hipMemcpyDeviceToHost, c->stream));
            const double t_wait = host_now();
            HIP_TRY(hipStreamSynchronize(c->stream));
            c->host_wait_s += host_now() - t_wait;
            memcpy(info.data(), back.data(), n_lists * sizeof(DDrawListInfo));
            memcpy(totals, back.data() + sz_info - 64, 12);
        }
        n_draws = totals[0]; n_inst = totals[1]; n_chunks = totals[2];
        for (uint32_t l = 0; l < n_lists; l++) {
            b.info[b.lists[l].view] = info[l];
            if (b.lists[l].drop_empty && info[l].n_draws == 0) b.views_dropped++;
        }
    }
    c->batch_instances = n_inst;
    HIP_TRY(c->d_chunk_flags.ensure(std::max<size_t>(n_chunks, 16)));
    HIP_TRY(c->d_recs.ensure(std::max<size_t>(n_inst, 1) * sizeof(DTriRec)));
    HIP_TRY(c->d_pair_count.ensure(std::max<size_t>(n_inst, 1) * 4));
    HIP_TRY(c->d_tile_start.ensure(std::max<size_t>(n_tiles, 1) * 8));
    c->d_tile_end.p = c->d_tile_start.as<uint32_t>() + n_tiles;
    HIP_TRY(c->d_scan.ensure((scan_scratch_words(n_inst) + 16) * 4));
    uint32_t* counters = frame_counters(c);  // [0] pair total, [2] scratch total
    launch_setup(c->stream, c->d_views.as<DView>(), c->d_draws.as<DDraw>(), c->d_draw_first.as<uint32_t>(), n_draws, n_inst,
                 c->d_prims.as<DPrim>(), c->d_pos.p, c->d_idx.as<uint32_t>(), c->d_recs.as<DTriRec>(), c->d_pair_count.as<uint32_t>(),
                 counters + 16, c->d_chunk_first.as<uint32_t>(), n_chunks, c->d_chunk_bb.p, c->d_chunk_flags.as<uint8_t>());
    HIP_TRY(c->h_read.ensure(RDR_READ_BYTES));
    // The host sizes the pair buffers from the sums set-up leaves in the spread counters, so the copy goes out in front
    // of the scan: the scan runs while the count travels to the host and the host gets the next launches ready.
    // (It lands beside the block the end of the frame is read into.)
    HIP_TRY(hipMemcpyAsync(c->h_read.as<uint8_t>() + RDR_READ_MID, counters, RDR_COUNTER_BYTES, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipEventRecord(c->ev_pairs, c->stream));
    if (c->trace) (void)hipEventRecord(c->tr_ev[c->slot][0], c->stream);
    // One launch up to RDR_CHAINED_MAX elements (RDR_SCAN_CLASSIC=1: always the two-kernel scan, a measurement aid). The
    // chained kernels pass a running total from block to block through memory, a few microseconds per step between
    // dies: fewer launches is what counts for a frame's million triangles, the chain is what counts for fifty million
    // (0.54 ms against 0.29 ms for the multi-kernel scan).
    if (n_inst <= RDR_CHAINED_MAX && !env_on("RDR_SCAN_CLASSIC")) {
        RDR_TRY(ensure_scan_status(c, scan_chained_words(n_inst)));
        launch_exclusive_scan_chained(c->stream, c->d_pair_count.as<uint32_t>(), c->d_pair_count.as<uint32_t>(), n_inst, c->d_scan_status.as<uint32_t>(), counters);
    } else {
        launch_exclusive_scan(c->stream, c->d_pair_count.as<uint32_t>(), c->d_pair_count.as<uint32_t>(), n_inst, c->d_scan.as<uint32_t>(), counters);
    }
    if (after_scan) RDR_TRY(after_scan());
    {
        const double t_wait = host_now();
        HIP_TRY(hipEventSynchronize(c->ev_pairs));
        c->host_wait_s += host_now() - t_wait;
        c->tr_host[c->slot][1] = t_wait; c->tr_host[c->slot][2] = host_now();
    }
    const uint32_t* mid = (const uint32_t*)(c->h_read.as<uint8_t>() + RDR_READ_MID);
    // what set-up summed over the triangles of the blended view (none in other batches): bounds for its fragment list
    c->blend_pairs = 0; c->blend_box_pixels = 0;
    unsigned long long pairs64 = 0;
    for (uint32_t k = 0; k < RDR_BINNED_SLOTS; k++) {
        const uint32_t* slot = mid + 16 + k * 32;
        c->blend_pairs += slot[1];
        unsigned long long a; memcpy(&a, slot + 2, 8);
        c->blend_box_pixels += a;
        memcpy(&a, slot + 6, 8);
        pairs64 += a;
    }
    // pair positions are 32-bit (the counters run on over the batches of a frame: this batch's share is what was added
    // since the batch before)
    const unsigned long long batch_pairs = pairs64 - c->frame_pairs_before;
    c->frame_pairs_before = pairs64;
    const uint32_t n_pairs = (uint32_t)batch_pairs;
    if (batch_pairs >= 0xFFFFFFF0ull)
        return fail(c, RDR_ERR_CAPACITY, "more than 2^32 (triangle, tile) pairs in one batch: too many large triangles over too many views");
    for (int k = 0; k < 2; k++) {
        HIP_TRY(c->d_keys[k].ensure(std::max<size_t>(n_pairs, 1) * 4));
        HIP_TRY(c->d_vals[k].ensure(std::max<size_t>(n_pairs, 1) * 4));
    }
    uint32_t key_bits = 1;
    while ((1ull << key_bits) <= n_tiles) key_bits++;
    uint32_t* keys[2] = {c->d_keys[0].as<uint32_t>(), c->d_keys[1].as<uint32_t>()};
    uint32_t* vals[2] = {c->d_vals[0].as<uint32_t>(), c->d_vals[1].as<uint32_t>()};
    int cur;
    // Pair expansion, then the sort: one launch per digit behind one that counts the digits (and clears the tile ranges
    // on the way), five launches from the pair count to the sorted lists. RDR_SORT_CLASSIC=1: the four-kernel passes
    // (count, two-level scan, scatter), a measurement aid and the path for more than RDR_CHAINED_MAX pairs (6 M pairs:
    // 0.40 ms per chained pass against 0.28).
    if (n_pairs <= RDR_CHAINED_MAX && !env_on("RDR_SORT_CLASSIC")) {
        const size_t need = sort_onesweep_work_bytes(n_pairs);
        if (c->d_sort_work.cap < need || c->sort_epoch > 0xFFFF0000u) {
            HIP_TRY(c->d_sort_work.ensure(need));
            HIP_TRY(hipMemsetAsync(c->d_sort_work.p, 0, c->d_sort_work.cap, c->stream));  // no status word of an earlier sort
            c->sort_epoch = 0u;
        }
        launch_emit(c->stream, c->d_views.as<DView>(), c->d_recs.as<DTriRec>(), c->d_pair_count.as<uint32_t>(), n_inst, n_pairs,
                    keys[0], vals[0], c->d_sort_work.as<uint32_t>(), RDR_SORT_ONESWEEP_HIST_WORDS);
        cur = launch_radix_sort_onesweep(c->stream, keys, vals, n_pairs, key_bits, c->d_sort_work.p, &c->sort_epoch, c->d_tile_start.as<uint32_t>(),
                                         n_tiles * 2u, counters + RDR_COUNTER_SORT_FAILED, n_pairs != 0u);
        launch_ranges(c->stream, keys[cur], n_pairs, c->d_tile_start.as<uint32_t>(), n_tiles, true);
    } else {
        HIP_TRY(c->d_hist.ensure((sort_hist_words(n_pairs) + 16) * 4));
        HIP_TRY(c->d_scan.ensure((scan_scratch_words((uint32_t)sort_hist_words(n_pairs)) + scan_scratch_words(n_inst) + 16) * 4));
        launch_emit(c->stream, c->d_views.as<DView>(), c->d_recs.as<DTriRec>(), c->d_pair_count.as<uint32_t>(), n_inst, n_pairs, keys[0], vals[0]);
        cur = launch_radix_sort(c->stream, keys, vals, n_pairs, key_bits, c->d_hist.as<uint32_t>(), c->d_scan.as<uint32_t>(), counters + 2);
        launch_ranges(c->stream, keys[cur], n_pairs, c->d_tile_start.as<uint32_t>(), n_tiles);
    }
    if (c->trace) { (void)hipEventRecord(c->tr_ev[c->slot][1], c->stream); c->tr_host[c->slot][3] = host_now(); }
    c->stats.triangle_instances += n_inst;
    c->stats.tile_pairs += n_pairs;
    c->stats.views += (uint32_t)b.views.size() - b.views_dropped;
    *out_tiles = n_tiles;
    *out_cur = cur;
    return RDR_OK;
}

// `after_scan`, if given, runs once set-up and the pair scan are enqueued and before the host waits for the pair
// count: work enqueued there keeps the GPU busy while the host sizes the pair buffers.
// `classify`: fragment lists the 4x rasteriser fills for the opaque pass (the frame's main batch), or null.
int prev_slot(const rdr_ctx* c) { return (c->slot + RDR_FRAME_SLOTS - 1) % RDR_FRAME_SLOTS; }

// `main_batch`: the frame's own batch. Its shadow-view rasteriser goes to the shading stream (in front of the frame's
// shading, which needs it, and behind the previous frame's, which reads the point-light maps it writes) instead of
// following the front end on `stream`: `stream` is then free for the front end of the next frame, which the host
// enqueues while this frame is rasterised, and does not queue behind the rasterisers.
int raster_batch(rdr_ctx* c, Batch& b, int* out_cur = nullptr, const std::function<int()>& after_scan = nullptr, const FragLists* classify = nullptr,
                 bool main_batch = false) {
    if (b.views.empty()) return RDR_OK;
    uint32_t n_tiles = 0;
    int cur = 0;
    RDR_TRY(bin_batch(c, b, &n_tiles, &cur, after_scan));
    if (out_cur) *out_cur = cur;
    // tile range per sample count (views are laid out in order, so each is one interval); tiles of views this
    // kernel does not rasterise (load == 2) are left out
    // three groups: single-sampled views that start from a clear (cascades, static maps), single-sampled views that
    // start from another map (point-light faces with moving meshes), the 4x view
    RasterRange rr[3] = {{0xFFFFFFFFu, 0u, 0u}, {0xFFFFFFFFu, 0u, 0u}, {0xFFFFFFFFu, 0u, 0u}};
    for (uint32_t vi = 0; vi < b.views.size(); vi++) {
        const DView& v = b.views[vi];
        if (v.load == 2u) continue;
        RasterRange& r = rr[v.samples == 4 ? 2 : (v.load == 1u ? 1 : 0)];
        if (r.first_tile == 0xFFFFFFFFu) { r.first_tile = v.tile_base; r.first_view = vi; }
        r.end_tile = v.tile_base + v.tiles_x * v.tiles_y;
    }
    hipStream_t rs = (main_batch && !getenv("RDR_RASTER_ON_FRONT_END_STREAM")) ? c->stream2 : c->stream;
    c->raster_stream = rs;
    HIP_TRY(hipEventRecord(c->ev_fork, c->stream));  // the sorted tile lists (and everything else the front end leaves) are there
    if (rs != c->stream) HIP_TRY(hipStreamWaitEvent(rs, c->ev_fork, 0));
    HIP_TRY(hipEventRecord(c->ev[8], rs));
    const RasterRange none = {0xFFFFFFFFu, 0u, 0u};
    // a frame whose fragment lists come from the rasteriser stores the main view's depth only where the blended pass
    // needs it (RDR_FULL_DEPTH=1: everywhere, a measurement aid)
    uint32_t depth_mode = 0u, blend_base = 0xFFFFFFFFu;
    if (main_batch && classify && rr[2].first_tile < rr[2].end_tile && !getenv("RDR_FULL_DEPTH")) {
        depth_mode = 1u;
        const DView& mv = b.views[rr[2].first_view];
        for (const DView& v : b.views)
            if (v.load == 2u && v.samples == 4u && v.depth == mv.depth) blend_base = v.tile_base;
        c->depth_partial = true;
        c->depth_view = mv; c->depth_view.vis = nullptr;
        c->depth_range = rr[2]; c->depth_range.first_view = 0u;
        c->depth_cur = cur;
        c->depth_blend_base = blend_base;
    } else if (main_batch) {
        c->depth_partial = false;
    }
    auto launch = [&](hipStream_t st, const RasterRange& s1, const RasterRange& s4) {
        launch_raster(st, b.views.data(), (uint32_t)b.views.size(), c->d_views.as<DView>(), c->d_recs.as<DTriRec>(), c->d_vals[cur].as<uint32_t>(),
                      c->d_tile_start.as<uint32_t>(), c->d_tile_end.as<uint32_t>(), s1, 1);
        launch_raster(st, b.views.data(), (uint32_t)b.views.size(), c->d_views.as<DView>(), c->d_recs.as<DTriRec>(), c->d_vals[cur].as<uint32_t>(),
                      c->d_tile_start.as<uint32_t>(), c->d_tile_end.as<uint32_t>(), s4, 4, classify, depth_mode, blend_base);
    };
    // RDR_SERIAL_RASTER: measurement aid, the rasterisers one after the other so that a profile shows each alone
    const bool side = rr[2].first_tile < rr[2].end_tile && (rr[0].first_tile < rr[0].end_tile || rr[1].first_tile < rr[1].end_tile) && !getenv("RDR_SERIAL_RASTER");
    if (side) {
        HIP_TRY(hipStreamWaitEvent(c->stream3, c->ev_fork, 0));
        launch(c->stream3, none, rr[2]);
        HIP_TRY(hipEventRecord(c->ev_join, c->stream3));
    } else {
        launch(rs, none, rr[2]);
    }
    launch(rs, rr[0], none);
    if (rr[1].first_tile < rr[1].end_tile) {
        // the point-light shadow maps exist once: the previous frame's shading may still be reading them
        // (the faces with moving meshes in one launch with the views that start from a clear: 1 % fewer frames/s)
        if (c->pending[prev_slot(c)].on) HIP_TRY(hipStreamWaitEvent(rs, c->ev_shaded[prev_slot(c)], 0));
        launch(rs, rr[1], none);
    }
    if (c->point_borders_stale && c->point_cubes_in_use) {
        // behind every rasteriser launch of this stream that writes point-light faces (static maps were copied in before)
        if (c->pending[prev_slot(c)].on) HIP_TRY(hipStreamWaitEvent(rs, c->ev_shaded[prev_slot(c)], 0));
        launch_shadow_cube_border(rs, c->d_point_shadow.as<float>(), (int)c->settings.point_shadow_map_resolution, c->point_cubes_in_use);
        c->point_borders_stale = false;
    }
    // what comes behind the batch on another stream waits for these two itself: one hop from the rasteriser that ends
    // last instead of two (stream3 -> stream -> the other stream, some ten microseconds each)
    HIP_TRY(hipEventRecord(c->ev_raster1, rs));
    c->raster_side = side;
    if (side) HIP_TRY(hipStreamWaitEvent(rs, c->ev_join, 0));
    HIP_TRY(hipGetLastError());
    return RDR_OK;
}

DView make_view(float* depth, uint32_t* vis, uint32_t w, uint32_t h, uint32_t samples, uint32_t cull, uint32_t load, const float* init = nullptr,
                uint32_t pitch = 0) {
    DView v;
    memset(&v, 0, sizeof v);
    v.depth = depth; v.init = init ? init : depth; v.vis = vis; v.w = w; v.h = h; v.samples = samples; v.cull = cull; v.load = load;
    v.pitch = pitch ? pitch : w;
    return v;
}

// Shadow targets carry a border the shading kernels filter across without edge cases. A point-light cube face is
// (res + 2)^2 with one texel of border (the depths across the cube edges, k_shadow_cube_border); a cascade is
// (res + 4)^2 with two texels of 1.0 (the sampler's opaque-white border, VulkanRendererShadow.cpp:73).
size_t point_face_texels(const rdr_ctx* c) { const size_t p = (size_t)c->settings.point_shadow_map_resolution + 2; return p * p; }
size_t point_face_origin(const rdr_ctx* c) { return (size_t)c->settings.point_shadow_map_resolution + 2 + 1; }
size_t cascade_texels(const rdr_ctx* c) { const size_t p = (size_t)c->settings.directional_shadow_map_resolution + 4; return p * p; }
size_t cascade_origin(const rdr_ctx* c) { return 2 * ((size_t)c->settings.directional_shadow_map_resolution + 4) + 2; }

// recordPointShadowMapsCommands (VulkanRendererShadow.cpp:255-389)
int point_shadow_views(rdr_ctx* c, vec3 camera_pos, Batch& dynamic_batch) {
    const uint32_t count = c->settings.point_shadow_map_count;
    std::vector<uint32_t> casters;
    for (uint32_t i = 0; i < c->lights.size(); i++)
        if (c->lights[i].flags & RDR_POINT_LIGHT_CAST_SHADOWS) casters.push_back(i);
    if (casters.empty() || count == 0) return RDR_OK;
    const uint32_t res = c->settings.point_shadow_map_resolution, pitch = res + 2;
    const size_t face_px = point_face_texels(c), origin = point_face_origin(c), total_bytes = face_px * 6 * count * 4;
    if (c->d_point_shadow.cap < total_bytes) {
        c->point_dynamic_key.valid = false;
        HIP_TRY(c->d_point_shadow.ensure(total_bytes));
        HIP_TRY(c->d_point_static.ensure(total_bytes));
        launch_fill_f32(c->stream, c->d_point_shadow.as<float>(), 1.0f, face_px * 6 * count);
        launch_fill_f32(c->stream, c->d_point_static.as<float>(), 1.0f, face_px * 6 * count);
    }
    static const vec3 dirs[6] = {{1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
    static const vec3 ups[6] = {{0, 1, 0}, {0, 1, 0}, {0, 0, -1}, {0, 0, 1}, {0, 1, 0}, {0, 1, 0}};
    Camera face[6];
    for (int j = 0; j < 6; j++) {
        face[j].type = 0;
        face[j].euler = quat_euler(quat_look_at(dirs[j], ups[j]));
        face[j].near_plane = 0.1f; face[j].far_plane = 100.0f;
        face[j].fov_or_height = 90.0f * 0.01745329251994329576923690768489f;
        face[j].aspect = 1.0f;
        face[j].crop = mat_scale(make3(-1.0f, 1.0f, 1.0f));
    }
    std::vector<float> key(c->lights.size());
    for (uint32_t i : casters) key[i] = distance(c->lights[i].point, camera_pos);
    std::stable_sort(casters.begin(), casters.end(), [&](uint32_t a, uint32_t b) { return key[a] < key[b]; });
    std::vector<uint32_t> slotted;
    uint16_t slot = 0;
    for (uint32_t i : casters) {
        HostLight& l = c->lights[i];
        if (slot < count) {
            slotted.push_back(i);
            if (l.slot != slot) l.flags &= ~RDR_POINT_LIGHT_STATIC_SHADOW_MAP_RENDERED;
            l.slot = slot;
            l.flags |= RDR_POINT_LIGHT_HAS_SHADOW_MAP;
            slot++;
        } else {
            l.slot = 0xFFFF;
            l.flags &= ~(RDR_POINT_LIGHT_STATIC_SHADOW_MAP_RENDERED | RDR_POINT_LIGHT_HAS_SHADOW_MAP);
        }
    }
    std::vector<uint32_t> static_prims, dynamic_prims;
    for (uint32_t p = 0; p < c->prims.size(); p++) (c->meshes[c->prims[p].mesh].is_static ? static_prims : dynamic_prims).push_back(p);
    // static maps of the lights whose slot is stale, all faces in one batch
    Batch sb;
    std::vector<uint32_t> refreshed;
    for (uint32_t i : slotted) {
        HostLight& l = c->lights[i];
        if (l.flags & RDR_POINT_LIGHT_STATIC_SHADOW_MAP_RENDERED) continue;
        for (int j = 0; j < 6; j++) {
            face[j].position = l.point;
            const mat4 vp = face[j].view_proj();
            float* dst = c->d_point_static.as<float>() + ((size_t)l.slot * 6 + j) * face_px + origin;
            sb.views.push_back(make_view(dst, nullptr, res, res, 1, 1, 0, nullptr, pitch));
            add_view_draws(c, sb, CAND_STATIC, static_prims, vp, l.point, SORT_NONE, (uint32_t)sb.views.size() - 1, false);
        }
        l.flags |= RDR_POINT_LIGHT_STATIC_SHADOW_MAP_RENDERED;
        refreshed.push_back(i);
    }
    if (!sb.views.empty()) {
        // d_point_static exists once: the previous frame's rasteriser of the faces with moving meshes (stream2, in front of
        // that frame's shading) may still be reading the static faces this batch is about to overwrite
        if (c->pending[prev_slot(c)].on) HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_shaded[prev_slot(c)], 0));
        RDR_TRY(raster_batch(c, sb));
        // (the copies into the maps the shaders read come behind the same wait)
        for (uint32_t i : refreshed) {
            const size_t off = (size_t)c->lights[i].slot * 6 * face_px;
            HIP_TRY(hipMemcpyAsync(c->d_point_shadow.as<float>() + off, c->d_point_static.as<float>() + off, 6 * face_px * 4,
                                   hipMemcpyDeviceToDevice, c->stream));
        }
        c->point_borders_stale = true;
    }
    c->point_cubes_in_use = (uint32_t)slotted.size();
    // The pass over the moving primitives is a function of the lights in the slots, the static maps and the geometry:
    // if none of them changed since the last frame, the faces already hold what it would write.
    {
        rdr_ctx::ShadowKey& key = c->point_dynamic_key;
        uint64_t h = hash_words(14695981039346656037ull, &res, 4);
        for (uint32_t i : slotted) { h = hash_words(h, &i, 4); h = hash_words(h, &c->lights[i].point, 12); }
        const uint32_t dyn = c->settings.dynamic_shadows_enabled;
        h = hash_words(h, &dyn, 4);
        if (refreshed.empty() && key.valid && key.geometry == c->geometry_version && key.lights == c->lights_version && key.hash == h) {
            c->views_reused += 6 * (uint32_t)slotted.size();
            return RDR_OK;
        }
        key.valid = true; key.geometry = c->geometry_version; key.lights = c->lights_version; key.hash = h;
    }
    // dynamic meshes on top of the static face: the rasteriser starts every tile of such a face from the cached static
    // map (DView::init) and writes the result to the map the shaders read, so the static map is never copied
    for (uint32_t i : slotted) {
        HostLight& l = c->lights[i];
        for (int j = 0; j < 6; j++) {
            face[j].position = l.point;
            const mat4 vp = face[j].view_proj();
            if (!c->settings.dynamic_shadows_enabled || dynamic_prims.empty()) continue;
            // a face is rendered again only if a moving primitive is in its frustum (VulkanRendererShadow.cpp:337-342);
            // otherwise it keeps what it holds
            const size_t n_before = dynamic_batch.draws.size();
            add_view_draws(c, dynamic_batch, CAND_DYNAMIC, dynamic_prims, vp, l.point, SORT_NONE, (uint32_t)dynamic_batch.views.size(), true);
            if (c->device_draws || dynamic_batch.draws.size() > n_before) {
                const size_t off = ((size_t)l.slot * 6 + j) * face_px + origin;
                dynamic_batch.views.push_back(make_view(c->d_point_shadow.as<float>() + off, nullptr, res, res, 1, 1, 1,
                                                        c->d_point_static.as<float>() + off, pitch));
                c->point_borders_stale = true;
            }
        }
    }
    return RDR_OK;
}

// recordDirectionalShadowMapsCommands (VulkanRendererShadow.cpp:391-456)
int cascade_views(rdr_ctx* c, Batch& b) {
    const uint32_t levels = c->settings.directional_shadow_cascade_levels, res = c->settings.directional_shadow_map_resolution;
    const size_t px = cascade_texels(c);
    if (c->d_dir_shadow.cap < px * levels * 4) {
        c->cascades_key[c->slot].valid = false;
        HIP_TRY(c->d_dir_shadow.ensure(px * levels * 4));
        launch_fill_f32(c->stream, c->d_dir_shadow.as<float>(), 1.0f, px * levels);  // the border keeps this value; the rasteriser writes the interior
    }
    Camera light;  // DirectionalLight::lightViewCamera (DirectionalLight.cpp:3-5)
    light.type = 1;
    light.position = make3(c->dir_light.position[0], c->dir_light.position[1], c->dir_light.position[2]);
    const quat q = {c->dir_light.orientation_wxyz[0], c->dir_light.orientation_wxyz[1], c->dir_light.orientation_wxyz[2], c->dir_light.orientation_wxyz[3]};
    light.euler = quat_euler(q);
    light.near_plane = 0.1f; light.far_plane = 200.0f; light.fov_or_height = 10.0f; light.aspect = 1.0f;
    light.crop = mat_identity();
    const mat4 light_vp = light.view_proj();
    std::vector<uint32_t> all(c->prims.size());
    for (uint32_t p = 0; p < all.size(); p++) all[p] = p;
    static const float ndc[8][3] = {{-1, -1, 0}, {1, -1, 0}, {-1, 1, 0}, {1, 1, 0}, {-1, -1, 1}, {1, -1, 1}, {-1, 1, 1}, {1, 1, 1}};
    vec3 eye[RDR_MAX_CASCADES];
    for (uint32_t i = 0; i < levels; i++) {
        Camera split = c->camera;  // the reference's Camera copy drops the crop matrix (Camera.cpp:3-41)
        split.crop = mat_identity();
        split.near_plane = c->cascade_depths[i];
        if (i < levels - 1) split.far_plane = c->cascade_depths[i + 1];
        // Camera::getFrustumVertices (Camera.cpp:218-242) then makeAABBFromVertices in light clip space (:306-329)
        const mat4 inv = mat_inverse(split.view_proj());
        float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int k = 0; k < 8; k++) {
            const vec4 vh = mat_mul_vec(inv, vec4{ndc[k][0], ndc[k][1], ndc[k][2], 1.0f});
            const vec4 h = mat_mul_vec(light_vp, vec4{vh.x / vh.w, vh.y / vh.w, vh.z / vh.w, 1.0f});
            const float v[3] = {h.x / h.w, h.y / h.w, h.z / h.w};
            for (int a = 0; a < 3; a++) { if (v[a] < mn[a]) mn[a] = v[a]; if (v[a] > mx[a]) mx[a] = v[a]; }
        }
        const float largest = std::max(mx[0] - mn[0], mx[1] - mn[1]);
        const vec3 offset = make3(-((mx[0] + mn[0]) / 2), -((mx[1] + mn[1]) / 2), -0.0f);
        Camera split_pov = light;
        split_pov.crop = mat_mul(mat_scale(make3(1.0f / largest, 1.0f / largest, 1.0f)), mat_translate(offset));
        c->csm_viewproj[i] = split_pov.view_proj();
        eye[i] = split_pov.position;
    }
    // The reference draws every cascade every frame (VulkanRendererShadow.cpp:424). The array of this frame slot
    // still holds the cascades of RDR_FRAME_SLOTS frames ago: if nothing that goes into them has changed since (same matrices,
    // no mesh added or moved), they are what this frame would render, and are left alone.
    rdr_ctx::ShadowKey& key = c->cascades_key[c->slot];
    uint64_t h = hash_words(14695981039346656037ull, &levels, 4);
    h = hash_words(h, &res, 4);
    for (uint32_t i = 0; i < levels; i++) h = hash_words(h, c->csm_viewproj[i].m, 64);
    if (key.valid && key.geometry == c->geometry_version && key.hash == h) { c->views_reused += levels; return RDR_OK; }
    key.valid = true; key.geometry = c->geometry_version; key.hash = h;
    for (uint32_t i = 0; i < levels; i++) {
        b.views.push_back(make_view(c->d_dir_shadow.as<float>() + (size_t)i * px + cascade_origin(c), nullptr, res, res, 1, 1, 0, nullptr, res + 4));
        add_view_draws(c, b, CAND_EVERY, all, c->csm_viewproj[i], eye[i], SORT_NONE, (uint32_t)b.views.size() - 1, false);
    }
    return RDR_OK;
}

// exposure recurrence (VulkanRenderer.cpp:1075-1082)
float compute_exposure(rdr_ctx* c, const rdr_frame_params* fp) {
    if (fp && fp->exposure_mode == 2) return fp->exposure;
    if (fp && fp->exposure_mode == 1) c->temporal_luminance = fp->temporal_luminance;
    else {
        const double dt = fp ? (double)fp->delta_time : 1.0 / 60.0;
        const float avg = std::min(c->last_avg_log_lum, 10.0f);
        c->temporal_luminance = (float)(c->temporal_luminance + (exp2f(avg) - c->temporal_luminance) * (1 - exp(-dt)));
    }
    float exposure = 1.03f - 2 / (log2f(c->temporal_luminance + 1.0f) + 2);
    exposure = 0.18f / exposure + c->settings.exposure_bias;
    return exposure;
}

float tone_curve(float x) {  // tonemap.comp:26-28
    const float shoulderStr = 0.22f, linStr = 0.30f, linAngle = 0.10f, toeStr = 0.20f, toeNum = 0.01f, toeDenom = 0.30f;
    return ((x * (shoulderStr * x + linAngle * linStr) + toeStr * toeNum) / (x * (shoulderStr * x + linStr) + toeStr * toeDenom)) - toeNum / toeDenom;
}

// The output image: R8G8B8A8_UNORM, or R16G16B16A16_SFLOAT with hdrOutputEnabled (VulkanRenderer.cpp:668-676: the
// float format a presentation surface offers). The tonemap kernel stores two pixels at a time.
size_t out_pixel_bytes(const rdr_ctx* c) { return c->settings.hdr_output_enabled ? 8u : 4u; }
uintptr_t out_align(const rdr_ctx* c) { return c->settings.hdr_output_enabled ? 16u : 8u; }

int ensure_post_targets(rdr_ctx* c) {
    const size_t n = (size_t)c->W * c->H;
    HIP_TRY(c->d_hdr.ensure(n * 8));
    size_t off = 0;
    for (uint32_t l = 0; l < 6; l++) {
        c->bloom_w[l] = std::max(1u, c->W >> l); c->bloom_h[l] = std::max(1u, c->H >> l);
        c->bloom_offset[l] = (uint32_t)off;
        off += ((size_t)c->bloom_w[l] * c->bloom_h[l] + 3u) & ~(size_t)3u;  // every level starts on a 16-byte boundary
    }
    // the accumulated mips 3 and 4 of the fused up-sample kernel (launch_bloom_chain)
    c->bloom_fin_offset[0] = (uint32_t)off; off += ((size_t)c->bloom_w[3] * c->bloom_h[3] + 3u) & ~(size_t)3u;
    c->bloom_fin_offset[1] = (uint32_t)off; off += (size_t)c->bloom_w[4] * c->bloom_h[4];
    // A bloom level whose dispatch has no groups (the reference's extent is W >> l x H >> l, VulkanRendererBloom.cpp:82,
    // while the level itself is never smaller than 1 x 1) is never written and is read by the up-sample above it. The
    // reference leaves its contents undefined; here it reads as zero: the buffer is cleared when it is created, and
    // nothing ever writes such a level.
    if (c->d_bloom.cap < off * 4) {
        HIP_TRY(c->d_bloom.ensure(off * 4));
        HIP_TRY(hipMemset(c->d_bloom.p, 0, c->d_bloom.cap));
    }
    HIP_TRY(c->d_out.ensure(n * out_pixel_bytes(c)));
    HIP_TRY(c->d_loglum.ensure(n * 4));
    HIP_TRY(c->d_lum.ensure((256 + 16) * 4));
    return RDR_OK;
}

// bloom chain + tonemap + average luminance (VulkanRendererBloom.cpp:62-100, VulkanRenderer.cpp:1084-1109, 1156-1191)
// `direct_out`: a device destination the tonemap kernel writes the finished frame to instead of d_out (aligned to
// out_align(c) bytes)
void enqueue_post(rdr_ctx* c, hipStream_t st, float exposure, uint32_t* direct_out = nullptr) {
    uint32_t* bloom = c->d_bloom.as<uint32_t>();
    (void)hipEventRecord(c->ev[2], st);
    // mip 0 is the colour image itself until the tonemap kernel stores its final value
    uint32_t* mip[6];
    int mw[6], mh[6];
    for (int k = 0; k < 6; k++) { mip[k] = bloom + c->bloom_offset[k]; mw[k] = (int)c->bloom_w[k]; mh[k] = (int)c->bloom_h[k]; }
    c->bloom_fin_separate = launch_bloom_chain(st, c->d_hdr.p, mip, mw, mh, (int)c->W, (int)c->H, bloom + c->bloom_fin_offset[0], bloom + c->bloom_fin_offset[1]);
    (void)hipEventRecord(c->ev[3], st);
    launch_tonemap(st, c->d_hdr.p, bloom + c->bloom_offset[1], (int)c->bloom_w[1], (int)c->bloom_h[1], bloom, direct_out ? direct_out : c->d_out.as<uint32_t>(),
                   c->d_loglum.as<float>(), (int)c->W, (int)c->H, powf(2.0f, exposure), 1.0f / c->settings.gamma, tone_curve(11.2f), log2f(6.0f),
                   c->keep_intermediates, c->settings.hdr_output_enabled != 0);
    c->post_kept = c->keep_intermediates;
    launch_average_luminance(st, c->d_loglum.as<float>(), (int)c->W, (int)c->H, c->d_lum.as<float>(), (float*)frame_counters(c) + RDR_COUNTER_LUM);
    (void)hipEventRecord(c->ev[4], st);
    c->have_post = true;
}

}  // namespace

extern "C" {

int rdr_settings_default(rdr_settings* s) {
    if (!s) return RDR_ERR_INVALID;
    const rdr_settings d = {0, 1, 512, 8, 2048, 4, 0, 0.0f, 2.2f, 4};
    *s = d;
    return RDR_OK;
}

const char* rdr_backend_name(void) { return "hip:gfx950"; }

int rdr_create(const rdr_settings* settings, uint32_t width, uint32_t height, int device, rdr_ctx** out) {
    if (!settings || !out || width == 0 || height == 0 || width > 16384 || height > 16384) return RDR_ERR_INVALID;
    if ((settings->msaa != 1 && settings->msaa != 4) || settings->directional_shadow_cascade_levels < 1 ||
        settings->directional_shadow_cascade_levels > RDR_MAX_CASCADES)
        return RDR_ERR_UNSUPPORTED;
    // what the implementation does not offer is refused, not ignored: shadow maps whose tile boxes would not fit 16-bit
    // coordinates, more cube slots than a frame can reasonably hold
    if (settings->hdr_output_enabled > 1) return RDR_ERR_INVALID;
    if (settings->directional_shadow_map_resolution == 0 || settings->point_shadow_map_resolution == 0) return RDR_ERR_INVALID;
    if (settings->directional_shadow_map_resolution > 16384 || settings->point_shadow_map_resolution > 8192 || settings->point_shadow_map_count > 64)
        return RDR_ERR_UNSUPPORTED;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev) return RDR_ERR_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return RDR_ERR_DEVICE;
    rdr_ctx* c = new (std::nothrow) rdr_ctx();
    if (!c) return RDR_ERR_DEVICE;
    c->settings = *settings; c->W = width; c->H = height; c->device = device;
    memset(&c->dir_light, 0, sizeof c->dir_light);
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) c->shade_blocks = (uint32_t)cus * shade_frags_blocks_per_cu();
    }
    bool ok = hipStreamCreate(&c->stream) == hipSuccess;
    for (int i = 0; i < 10 * RDR_FRAME_SLOTS && ok; i++) ok = hipEventCreate(&c->ev_sets[i / 10][i % 10]) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&c->ev_pairs, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipStreamCreate(&c->stream2) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&c->ev_raster1, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&c->ev_vertex_in, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&c->ev_vertex_out, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipStreamCreate(&c->stream3) == hipSuccess && hipStreamCreate(&c->stream4) == hipSuccess;
    for (int i = 0; i < RDR_FRAME_SLOTS && ok; i++) ok = hipEventCreateWithFlags(&c->ev_shaded[i], hipEventDisableTiming) == hipSuccess;
    // decode tables: [0..255] sRGB EOTF, [256..511] i/255, [512..767] sRGB encode thresholds
    std::vector<float> tables(768);
    for (int i = 0; i < 256; i++) {
        tables[i] = (float)srgb_eotf((double)i / 255.0);
        tables[256 + i] = (float)i / 255.0f;
        tables[512 + i] = (float)srgb_eotf(((double)i + 0.5) / 255.0);
    }
    ok = ok && c->d_tables.ensure(768 * 4) == hipSuccess && hipMemcpy(c->d_tables.p, tables.data(), 768 * 4, hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && c->d_counters.ensure(RDR_FRAME_SLOTS * RDR_COUNTER_SLOT_BYTES) == hipSuccess && hipMemset(c->d_counters.p, 0, RDR_FRAME_SLOTS * RDR_COUNTER_SLOT_BYTES) == hipSuccess;
    ok = ok && c->d_frame.ensure(sizeof(DFrame) + 64) == hipSuccess;
    // this device's copy of the no-allocation scratch array (the device is current: hipSetDevice above)
    ok = ok && hipGetSymbolAddress((void**)&c->noalloc_scratch, HIP_SYMBOL(scratch)) == hipSuccess && c->noalloc_scratch != nullptr;
    if (!ok) { rdr_destroy(c); return RDR_ERR_DEVICE; }
    if (const char* e = getenv("RDR_BLEND_BUDGET_SLOTS")) { const long long v = atoll(e); if (v > 0) c->blend_budget_slots = (uint64_t)v; }
    // 2 or 3 frames in flight (2 by default: a frame's results are there one call later; 3 gave +1 % frames/s on the
    // benchmark frame and a shading kernel 2 % longer for what runs beside it)
    if (getenv("RDR_TAIL_ON_SHADING_STREAM")) c->tail_split = false;
    if (env_on("RDR_TRACE_FRAME")) {
        c->trace = true;
        for (int i = 0; i < RDR_FRAME_SLOTS * 2; i++) c->trace = c->trace && hipEventCreate(&c->tr_ev[i / 2][i % 2]) == hipSuccess;
    }
    if (const char* e = getenv("RDR_FRAMES_IN_FLIGHT")) { const int v = atoi(e); if (v == 2 || v == 3) c->lag = v - 1; }
    // measurement aid: size of the fragment shading kernel's grid (a multiple of 8, one share per die)
    if (const char* e = getenv("RDR_SHADE_BLOCKS")) { const long v = atol(e); if (v >= 8 && v <= 65536) c->shade_blocks = (uint32_t)v & ~7u; }
    c->camera.type = 0;  // VulkanRenderer.h:117, VulkanRenderer.cpp:50
    c->camera.position = make3(0, 0, 0); c->camera.euler = make3(0, 0, 0);
    c->camera.near_plane = 0.1f; c->camera.far_plane = 60.0f;
    c->camera.fov_or_height = 40.0f * 0.01745329251994329576923690768489f;
    c->camera.aspect = (float)width / (float)height;
    c->camera.crop = mat_identity();
    // default textures: 1x1 white (VulkanRenderer.h:28-32), 1x1 (128,128,255,0) (textures/clear_normal.png)
    const uint8_t white[4] = {255, 255, 255, 255}, nrm[4] = {128, 128, 255, 0};
    uint32_t id;
    const rdr_sampler smp = {RDR_FILTER_LINEAR, RDR_FILTER_LINEAR, RDR_FILTER_LINEAR, RDR_WRAP_REPEAT, RDR_WRAP_REPEAT};
    if (create_image(c, white, 1, 1, false, 1, &id) != RDR_OK || create_image(c, nrm, 1, 1, false, 1, &id) != RDR_OK) { rdr_destroy(c); return RDR_ERR_DEVICE; }
    c->textures.push_back(HostTexture{0, smp});
    c->textures.push_back(HostTexture{1, smp});
    *out = c;
    return RDR_OK;
}

void rdr_destroy(rdr_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->stream2) (void)hipStreamSynchronize(c->stream2);
    if (c->stream3) (void)hipStreamSynchronize(c->stream3);
    if (c->stream4) (void)hipStreamSynchronize(c->stream4);
    DevBuf* bufs[] = {&c->d_pos, &c->d_attr, &c->d_idx, &c->d_prims, &c->d_textures, &c->d_tables, &c->d_trig, &c->d_env,
                      &c->d_irradiance, &c->d_prefiltered, &c->d_env_b, &c->d_irradiance_b, &c->d_prefiltered_b, &c->d_lut, &c->d_depth, &c->d_vis, &c->d_samples, &c->d_hdr, &c->d_bloom, &c->d_out,
                      &c->d_loglum, &c->d_lum, &c->d_dir_shadow, &c->d_point_shadow, &c->d_point_static, &c->d_batch_tables,
                      &c->d_recs, &c->d_pair_count, &c->d_scan, &c->d_hist, &c->d_tile_start,
                      &c->d_counters, &c->d_sort_work, &c->d_scan_status, &c->d_ofrag_recs, &c->d_ofrag_cols, &c->d_pixel_state, &c->d_tile_first, &c->d_frags, &c->d_frag_col, &c->d_frag_head, &c->d_tile_slots, &c->d_frame, &c->d_shade_tris, &c->d_rest_pos, &c->d_rest_attr, &c->d_skin_joints,
                      &c->d_skin_weights, &c->d_skin_dst_pal, &c->d_palette, &c->d_chunk_bb, &c->d_chunk_flags, &c->d_keys[0], &c->d_keys[1], &c->d_vals[0], &c->d_vals[1],
                      &c->d_prim_cull, &c->d_models, &c->d_cand, &c->d_draw_scratch, &c->d_irradiance_q, &c->d_prefiltered_q, &c->d_lut_q};
    for (DevBuf* b : bufs) b->release();
    for (rdr_ctx::AltBuffers& a : c->alt) {
        DevBuf* alts[] = {&a.batch_tables, &a.frame, &a.shade_tris, &a.recs, &a.keys[0], &a.keys[1], &a.vals[0], &a.vals[1], &a.tile_start, &a.depth, &a.vis,
                          &a.dir_shadow, &a.hdr, &a.ofrag_recs, &a.pixel_state, &a.tile_first};
        for (DevBuf* b : alts) b->release();
    }
    for (HostImage& img : c->images) img.pixels.release();
    c->h_stage.release(); c->h_read.release(); c->h_palette.release(); c->h_cull.release();
    for (int i = 0; i < 10 * RDR_FRAME_SLOTS; i++) if (c->ev_sets[i / 10][i % 10]) (void)hipEventDestroy(c->ev_sets[i / 10][i % 10]);
    for (int i = 0; i < RDR_FRAME_SLOTS * 2; i++) if (c->tr_ev[i / 2][i % 2]) (void)hipEventDestroy(c->tr_ev[i / 2][i % 2]);
    if (c->ev_pairs) (void)hipEventDestroy(c->ev_pairs);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->ev_raster1) (void)hipEventDestroy(c->ev_raster1);
    if (c->ev_vertex_in) (void)hipEventDestroy(c->ev_vertex_in);
    if (c->ev_vertex_out) (void)hipEventDestroy(c->ev_vertex_out);
    for (int i = 0; i < RDR_FRAME_SLOTS; i++) if (c->ev_shaded[i]) (void)hipEventDestroy(c->ev_shaded[i]);
    if (c->stream4) (void)hipStreamDestroy(c->stream4);
    if (c->stream3) (void)hipStreamDestroy(c->stream3);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    c->h_frame.release();
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* rdr_last_error(const rdr_ctx* c) { return c ? c->err.c_str() : "null context"; }

int rdr_load_buffer(rdr_ctx* c, const void* ptr, size_t size, uint32_t* out_id) {
    if (!c || !ptr || !out_id) return fail(c, RDR_ERR_INVALID, "load_buffer: null argument");
    c->buffers.emplace_back((const uint8_t*)ptr, (const uint8_t*)ptr + size);
    *out_id = (uint32_t)c->buffers.size() - 1;
    return RDR_OK;
}

int rdr_load_image(rdr_ctx* c, const void* ptr, size_t size, uint32_t w, uint32_t h, int format, uint32_t max_mips, uint32_t* out_id) {
    if (!c || !ptr || !out_id || w == 0 || h == 0) return fail(c, RDR_ERR_INVALID, "load_image: bad argument");
    int ch; bool srgb = false; int float_bytes = 0;  // bytes per component of an SFLOAT format, 0 for the 8-bit ones
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
    HIP_TRY(hipSetDevice(c->device));
    if (float_bytes) {
        // the SFLOAT formats (Image.h:9-12) are held as one float4 per texel; a missing alpha reads as 1 (the Vulkan rule
        // for a format without that component)
        std::vector<float> f4((size_t)w * h * 4);
        for (size_t i = 0; i < (size_t)w * h; i++) {
            for (int k = 0; k < 4; k++) {
                float v = 1.0f;
                if (k < ch) {
                    if (float_bytes == 2) { _Float16 hv; memcpy(&hv, (const uint8_t*)ptr + (i * ch + k) * 2, 2); v = (float)hv; }
                    else memcpy(&v, (const uint8_t*)ptr + (i * ch + k) * 4, 4);
                }
                f4[i * 4 + k] = v;
            }
        }
        return create_image(c, f4.data(), w, h, false, max_mips, out_id, true, float_bytes == 2);
    }
    const uint8_t* s = (const uint8_t*)ptr;
    std::vector<uint8_t> rgba;
    if (ch == 3) {
        rgba.resize((size_t)w * h * 4);
        for (size_t i = 0; i < (size_t)w * h; i++) { rgba[i * 4] = s[i * 3]; rgba[i * 4 + 1] = s[i * 3 + 1]; rgba[i * 4 + 2] = s[i * 3 + 2]; rgba[i * 4 + 3] = 255; }
        s = rgba.data();
    }
    return create_image(c, s, w, h, srgb, max_mips, out_id);
}

int rdr_make_texture(rdr_ctx* c, uint32_t image, const rdr_sampler* s, uint32_t* out_id) {
    if (!c || !s || !out_id || image >= c->images.size()) return fail(c, RDR_ERR_INVALID, "make_texture: bad argument");
    if (s->mag_filter > RDR_FILTER_LINEAR || s->min_filter > RDR_FILTER_LINEAR || s->mipmap_filter > RDR_FILTER_LINEAR)
        return fail(c, RDR_ERR_INVALID, "make_texture: bad filter");
    if (s->wrap_u > RDR_WRAP_CLAMP || s->wrap_v > RDR_WRAP_CLAMP) return fail(c, RDR_ERR_INVALID, "make_texture: bad wrap mode");
    // VulkanRenderer.cpp:370-371: the V address mode is REPEAT if wrapV says so; otherwise it is chosen by wrapU (the
    // reference tests wrapU for "mirror" in both lines), so (REPEAT, MIRROR) samples V clamped and (MIRROR, CLAMP)
    // samples V mirrored. Kept: it is what the reference's sampler does.
    rdr_sampler eff = *s;
    if (eff.wrap_v != RDR_WRAP_REPEAT) eff.wrap_v = eff.wrap_u == RDR_WRAP_MIRROR ? RDR_WRAP_MIRROR : RDR_WRAP_CLAMP;
    c->textures.push_back(HostTexture{image, eff});
    c->textures_dirty = true;
    *out_id = (uint32_t)c->textures.size() - 1;
    return RDR_OK;
}

int rdr_add_mesh(rdr_ctx* c, const rdr_primitive_desc* descs, uint32_t count, const float model[16], int is_static, uint32_t* out_mesh_id) {
    if (!c || !descs || !model || count == 0) return fail(c, RDR_ERR_INVALID, "add_mesh: bad argument");
    HostMesh mesh;
    memcpy(mesh.model.m, model, 64);
    mesh.is_static = is_static != 0;
    mesh.first_prim = (uint32_t)c->prims.size(); mesh.prim_count = count;
    const uint32_t mesh_id = (uint32_t)c->meshes.size();
    const size_t keep_pos = c->h_pos.size(), keep_attr = c->h_attr.size(), keep_idx = c->h_idx.size(), keep_prims = c->prims.size();
    const size_t keep_skin = c->h_skin_dst.size();
    auto rollback = [&](int code, const char* msg) {
        c->h_pos.resize(keep_pos); c->h_attr.resize(keep_attr); c->h_idx.resize(keep_idx); c->prims.resize(keep_prims);
        c->h_skin_dst.resize(keep_skin); c->h_skin_mesh.resize(keep_skin); c->h_skin_joints.resize(keep_skin * 2); c->h_skin_weights.resize(keep_skin * 4);
        return fail(c, code, msg);
    };
    for (uint32_t pi = 0; pi < count; pi++) {
        const rdr_primitive_desc& d = descs[pi];
        if (d.mode != 4) return rollback(RDR_ERR_UNSUPPORTED, "add_mesh: only triangle lists are supported");
        const rdr_vertex_attribute* at[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        for (uint32_t a = 0; a < d.attribute_count; a++) {
            const rdr_vertex_attribute& va = d.attributes[a];
            if (va.name > RDR_ATTR_WEIGHTS_0) continue;
            if (va.buffer >= c->buffers.size()) return rollback(RDR_ERR_INVALID, "add_mesh: bad buffer id");
            at[va.name] = &va;
        }
        if (!at[RDR_ATTR_POSITION]) return rollback(RDR_ERR_INVALID, "add_mesh: POSITION missing");
        const size_t nv = (size_t)at[RDR_ATTR_POSITION]->count;
        HostPrim hp;
        hp.first_vertex = (uint32_t)(c->h_pos.size() / 4);
        hp.first_index = (uint32_t)c->h_idx.size();
        hp.n_vertices = (uint32_t)nv;
        c->h_pos.resize(c->h_pos.size() + nv * 4);
        c->h_attr.resize(c->h_attr.size() + nv);
        float* pos = c->h_pos.data() + (size_t)hp.first_vertex * 4;
        DVertexAttr* attr = c->h_attr.data() + hp.first_vertex;
        memset(attr, 0, nv * sizeof(DVertexAttr));
        // reads `comps` components of element i of an attribute as floats
        auto fetch = [&](const rdr_vertex_attribute* va, size_t i, int comps, float* o) -> bool {
            const size_t vs = value_size(va->value_type);
            const int have = (int)va->container + 1;
            const size_t stride = va->stride ? (size_t)va->stride : vs * have;
            const size_t off = (size_t)va->offset + i * stride;
            const std::vector<uint8_t>& buf = c->buffers[va->buffer];
            if (off + vs * std::min(comps, have) > buf.size()) return false;
            const bool norm = (va->name == RDR_ATTR_TEXCOORD_0 || va->name == RDR_ATTR_WEIGHTS_0) && va->value_type != RDR_FLOAT;
            for (int k = 0; k < comps; k++) o[k] = k < have ? component_as_float(&buf[off + k * vs], va->value_type, norm) : 0.0f;
            return true;
        };
        for (size_t i = 0; i < nv; i++) {
            float v[4];
            if (!fetch(at[RDR_ATTR_POSITION], i, 3, v)) return rollback(RDR_ERR_INVALID, "add_mesh: POSITION out of range");
            pos[i * 4] = v[0]; pos[i * 4 + 1] = v[1]; pos[i * 4 + 2] = v[2]; pos[i * 4 + 3] = 1.0f;
            if (at[RDR_ATTR_NORMAL]) {
                if (!fetch(at[RDR_ATTR_NORMAL], i, 3, v)) return rollback(RDR_ERR_INVALID, "add_mesh: NORMAL out of range");
                attr[i].nx = v[0]; attr[i].ny = v[1]; attr[i].nz = v[2];
            }
            if (at[RDR_ATTR_TEXCOORD_0]) {
                if (!fetch(at[RDR_ATTR_TEXCOORD_0], i, 2, v)) return rollback(RDR_ERR_INVALID, "add_mesh: TEXCOORD_0 out of range");
                attr[i].u = v[0]; attr[i].v = v[1];
            }
            if (at[RDR_ATTR_TANGENT]) {
                if (!fetch(at[RDR_ATTR_TANGENT], i, 4, v)) return rollback(RDR_ERR_INVALID, "add_mesh: TANGENT out of range");
                attr[i].tx = v[0]; attr[i].ty = v[1]; attr[i].tz = v[2];
                if (at[RDR_ATTR_BITANGENT]) {
                    float b[4];
                    if (!fetch(at[RDR_ATTR_BITANGENT], i, 3, b)) return rollback(RDR_ERR_INVALID, "add_mesh: BITANGENT out of range");
                    attr[i].bx = b[0]; attr[i].by = b[1]; attr[i].bz = b[2];
                } else {  // main.cpp:93
                    const vec3 bt = mulf(cross(make3(attr[i].nx, attr[i].ny, attr[i].nz), make3(v[0], v[1], v[2])), v[3]);
                    attr[i].bx = bt.x; attr[i].by = bt.y; attr[i].bz = bt.z;
                }
            }
        }
        size_t n_idx;
        if (d.indices.buffer != RDR_INVALID_ID) {
            if (d.indices.buffer >= c->buffers.size()) return rollback(RDR_ERR_INVALID, "add_mesh: bad index buffer id");
            const std::vector<uint8_t>& buf = c->buffers[d.indices.buffer];
            const uint32_t it = d.indices.index_type;
            if (it != RDR_UINT8 && it != RDR_UINT16 && it != RDR_UINT32) return rollback(RDR_ERR_UNSUPPORTED, "add_mesh: index type");
            const size_t is = value_size(it);
            if ((size_t)d.indices.offset + is * d.indices.count > buf.size()) return rollback(RDR_ERR_INVALID, "add_mesh: indices out of range");
            n_idx = (size_t)d.indices.count / 3 * 3;
            c->h_idx.resize(hp.first_index + n_idx);
            for (size_t i = 0; i < n_idx; i++) {
                const uint8_t* q = &buf[(size_t)d.indices.offset + i * is];
                uint32_t v = 0;
                if (it == RDR_UINT8) v = q[0];
                else if (it == RDR_UINT16) { uint16_t t; memcpy(&t, q, 2); v = t; }
                else memcpy(&v, q, 4);
                if (v >= nv) return rollback(RDR_ERR_INVALID, "add_mesh: index beyond vertex count");
                c->h_idx[hp.first_index + i] = v;
            }
        } else {
            n_idx = nv / 3 * 3;
            c->h_idx.resize(hp.first_index + n_idx);
            for (size_t i = 0; i < n_idx; i++) c->h_idx[hp.first_index + i] = (uint32_t)i;
        }
        hp.tri_count = (uint32_t)(n_idx / 3);
        // re-take the pointers: h_idx growth does not move them, but keep it obvious
        pos = c->h_pos.data() + (size_t)hp.first_vertex * 4;
        attr = c->h_attr.data() + hp.first_vertex;
        if (!at[RDR_ATTR_TANGENT]) {
            // MeshPrimitive::calculateTangents, indexed form (Mesh.cpp:5-52)
            const uint32_t* ix = c->h_idx.data() + hp.first_index;
            for (size_t i = 0; i + 2 < n_idx; i += 3) {
                const uint32_t a = ix[i], b = ix[i + 1], cc = ix[i + 2];
                const vec3 pa = make3(pos[a * 4], pos[a * 4 + 1], pos[a * 4 + 2]);
                const vec3 e1 = sub(make3(pos[b * 4], pos[b * 4 + 1], pos[b * 4 + 2]), pa), e2 = sub(make3(pos[cc * 4], pos[cc * 4 + 1], pos[cc * 4 + 2]), pa);
                const float s1 = attr[b].u - attr[a].u, s2 = attr[cc].u - attr[a].u, t1 = attr[b].v - attr[a].v, t2 = attr[cc].v - attr[a].v;
                const float f = 1.0f / (s1 * t2 - s2 * t1);
                const vec3 sdir = make3((t2 * e1.x - t1 * e2.x) * f, (t2 * e1.y - t1 * e2.y) * f, (t2 * e1.z - t1 * e2.z) * f);
                const uint32_t vs[3] = {a, b, cc};
                for (uint32_t vtx : vs) {
                    attr[vtx].tx = sdir.x; attr[vtx].ty = sdir.y; attr[vtx].tz = sdir.z;
                    const vec3 bt = cross(make3(attr[vtx].nx, attr[vtx].ny, attr[vtx].nz), sdir);
                    attr[vtx].bx = bt.x; attr[vtx].by = bt.y; attr[vtx].bz = bt.z;
                }
            }
        }
        hp.bb_min = make3(d.bb_min[0], d.bb_min[1], d.bb_min[2]);
        hp.bb_max = make3(d.bb_max[0], d.bb_max[1], d.bb_max[2]);
        vec3 bc = make3(0, 0, 0);  // calculateBarycenter (Mesh.cpp:77-86)
        for (size_t i = 0; i < nv; i++) bc = add(bc, make3(pos[i * 4], pos[i * 4 + 1], pos[i * 4 + 2]));
        bc = make3(bc.x / (float)nv, bc.y / (float)nv, bc.z / (float)nv);
        const vec4 bch = mat_mul_vec(mesh.model, vec4{bc.x, bc.y, bc.z, 1.0f});
        hp.barycenter = make3(bch.x / bch.w, bch.y / bch.w, bch.z / bch.w);
        hp.mat = d.material;
        const uint32_t tex_ids[5] = {hp.mat.albedo_texture, hp.mat.normal_texture, hp.mat.metal_roughness_texture, hp.mat.ao_texture, hp.mat.emissive_texture};
        for (uint32_t t : tex_ids)
            if (t != RDR_INVALID_ID && t >= c->textures.size()) return rollback(RDR_ERR_INVALID, "add_mesh: bad texture id");
        hp.mesh = mesh_id;
        if (at[RDR_ATTR_JOINTS_0] && at[RDR_ATTR_WEIGHTS_0]) {
            const uint32_t jt = at[RDR_ATTR_JOINTS_0]->value_type;
            if (jt != RDR_UINT8 && jt != RDR_UINT16) return rollback(RDR_ERR_UNSUPPORTED, "add_mesh: JOINTS_0 must be uint8 or uint16");
            if (mesh.is_static) return rollback(RDR_ERR_INVALID, "add_mesh: a skinned mesh must be added with is_static = 0");
            hp.skinned = true;
            hp.skin_first = (uint32_t)c->h_skin_dst.size();
            hp.rest_bb_min = hp.bb_min; hp.rest_bb_max = hp.bb_max;
            for (size_t i = 0; i < nv; i++) {
                float j[4], w[4];
                if (!fetch(at[RDR_ATTR_JOINTS_0], i, 4, j) || !fetch(at[RDR_ATTR_WEIGHTS_0], i, 4, w))
                    return rollback(RDR_ERR_INVALID, "add_mesh: JOINTS_0 / WEIGHTS_0 out of range");
                uint32_t ji[4];
                for (int k = 0; k < 4; k++) { ji[k] = (uint32_t)(uint16_t)j[k]; mesh.n_joints = std::max(mesh.n_joints, ji[k] + 1); }
                c->h_skin_dst.push_back(hp.first_vertex + (uint32_t)i);
                c->h_skin_mesh.push_back(mesh_id);
                c->h_skin_joints.push_back(ji[0] | (ji[1] << 16));
                c->h_skin_joints.push_back(ji[2] | (ji[3] << 16));
                for (int k = 0; k < 4; k++) c->h_skin_weights.push_back(w[k]);
            }
        }
        c->prims.push_back(hp);
    }
    mesh.joints.assign((size_t)mesh.n_joints * 16, 0.0f);
    for (uint32_t j = 0; j < mesh.n_joints; j++) mesh.joints[j * 16] = mesh.joints[j * 16 + 5] = mesh.joints[j * 16 + 10] = mesh.joints[j * 16 + 15] = 1.0f;
    c->meshes.push_back(mesh);
    c->geometry_version++;
    c->geometry_dirty = true;
    if (out_mesh_id) *out_mesh_id = mesh_id;
    return RDR_OK;
}

int rdr_set_mesh_transform(rdr_ctx* c, uint32_t mesh_id, const float model[16]) {
    if (!c || !model || mesh_id >= c->meshes.size()) return fail(c, RDR_ERR_INVALID, "set_mesh_transform: bad argument");
    memcpy(c->meshes[mesh_id].model.m, model, 64);
    c->geometry_version++;
    return RDR_OK;
}

int rdr_set_mesh_joints(rdr_ctx* c, uint32_t mesh_id, const float* jm, uint32_t count) {
    if (!c || !jm || mesh_id >= c->meshes.size()) return fail(c, RDR_ERR_INVALID, "set_mesh_joints: bad argument");
    HostMesh& mesh = c->meshes[mesh_id];
    if (mesh.n_joints == 0) return fail(c, RDR_ERR_INVALID, "set_mesh_joints: the mesh has no skinned primitive");
    if (count < mesh.n_joints) return fail(c, RDR_ERR_INVALID, "set_mesh_joints: fewer matrices than the largest joint index in use");
    memcpy(mesh.joints.data(), jm, (size_t)mesh.n_joints * 64);
    c->skin_dirty = true;
    c->geometry_version++;
    // culling box of each skinned primitive: around its rest box carried by every joint
    for (uint32_t pi = mesh.first_prim; pi < mesh.first_prim + mesh.prim_count; pi++) {
        HostPrim& p = c->prims[pi];
        if (!p.skinned) continue;
        vec3 lo = make3(INFINITY, INFINITY, INFINITY), hi = make3(-INFINITY, -INFINITY, -INFINITY);
        for (uint32_t jn = 0; jn < mesh.n_joints; jn++) {
            const float* m = jm + (size_t)jn * 16;
            for (int k = 0; k < 8; k++) {
                const float x = (k & 1) ? p.rest_bb_max.x : p.rest_bb_min.x, y = (k & 2) ? p.rest_bb_max.y : p.rest_bb_min.y,
                            z = (k & 4) ? p.rest_bb_max.z : p.rest_bb_min.z;
                const float qx = ((m[0] * x + m[4] * y) + m[8] * z) + m[12], qy = ((m[1] * x + m[5] * y) + m[9] * z) + m[13],
                            qz = ((m[2] * x + m[6] * y) + m[10] * z) + m[14];
                lo = make3(std::min(lo.x, qx), std::min(lo.y, qy), std::min(lo.z, qz));
                hi = make3(std::max(hi.x, qx), std::max(hi.y, qy), std::max(hi.z, qz));
            }
        }
        p.bb_min = lo; p.bb_max = hi;
    }
    return RDR_OK;
}

int rdr_read_primitive_positions(rdr_ctx* c, uint32_t mesh_id, uint32_t primitive, float* out, uint32_t* out_count) {
    if (!c || mesh_id >= c->meshes.size() || primitive >= c->meshes[mesh_id].prim_count) return fail(c, RDR_ERR_INVALID, "read_primitive_positions: bad id");
    const HostPrim& p = c->prims[c->meshes[mesh_id].first_prim + primitive];
    if (out_count) *out_count = p.n_vertices;
    if (!out) return RDR_OK;
    HIP_TRY(hipSetDevice(c->device));
    RDR_TRY(upload_scene(c));
    RDR_TRY(run_skinning(c));
    std::vector<float> tmp((size_t)p.n_vertices * 4);
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(tmp.data(), c->d_pos.as<float>() + (size_t)p.first_vertex * 4, tmp.size() * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < p.n_vertices; i++) { out[i * 3] = tmp[i * 4]; out[i * 3 + 1] = tmp[i * 4 + 1]; out[i * 3 + 2] = tmp[i * 4 + 2]; }
    return RDR_OK;
}

int rdr_set_lights(rdr_ctx* c, const rdr_point_light* lights, uint32_t count, const rdr_directional_light* dl) {
    if (!c || !dl || (count && !lights)) return fail(c, RDR_ERR_INVALID, "set_lights: bad argument");
    c->lights.clear();
    for (uint32_t i = 0; i < count; i++) {
        HostLight l;
        l.point = make3(lights[i].point[0], lights[i].point[1], lights[i].point[2]);
        l.intensity = make3(lights[i].intensity[0], lights[i].intensity[1], lights[i].intensity[2]);
        l.flags = lights[i].flags & RDR_POINT_LIGHT_CAST_SHADOWS;
        l.slot = 0xFFFF;
        c->lights.push_back(l);
    }
    c->dir_light = *dl;
    c->lights_set = true;
    c->lights_version++;
    compute_cascade_splits(c);
    return RDR_OK;
}

// Device time of a group of launches on the context's stream (the precompute calls synchronise anyway).
struct StreamTimer {
    hipEvent_t a = nullptr, b = nullptr;
    hipStream_t st;
    explicit StreamTimer(hipStream_t s) : st(s) {
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess || hipEventRecord(a, st) != hipSuccess) { release(); }
    }
    float stop() {  // waits for the launches; 0 if the events could not be made
        float ms = 0.0f;
        if (a && b && hipEventRecord(b, st) == hipSuccess && hipEventSynchronize(b) == hipSuccess) (void)hipEventElapsedTime(&ms, a, b);
        release();
        return ms;
    }
    void release() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); a = b = nullptr; }
    ~StreamTimer() { release(); }
};

// Both bakes from the packed cube in d_env (the second half of setEnvironmentMap).
// The environment cubes and the BRDF table exist once and every frame's shading reads them: a call that replaces them
// first lets the frames in flight finish (their results stay where they are until they are picked up as usual).
static int finish_frames_in_flight(rdr_ctx* c) {
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream3));
    HIP_TRY(hipStreamSynchronize(c->stream2));
    HIP_TRY(hipStreamSynchronize(c->stream4));
    return RDR_OK;
}

static int bake_environment(rdr_ctx* c, uint32_t w) {
    memset(&c->env, 0, sizeof c->env);
    c->env.texels = c->d_env.as<uint32_t>(); c->env.res = w; c->env.levels = 1;
    // the bordered copy first: the bakes read it too (four neighbouring texels per tap instead of folding the footprint
    // over the cube's edges tap by tap: the same texels, the same sums)
    {
        c->env_b = c->env;
        c->env_b.level_offset[0] = 0;
        HIP_TRY(c->d_env_b.ensure((size_t)6 * (w + 2) * (w + 2) * 4 + 16));  // the shading kernels read texel pairs: one texel of slack
        launch_cube_border(c->stream, c->env.texels, c->d_env_b.as<uint32_t>(), (int)w);
        c->env_b.texels = c->d_env_b.as<uint32_t>();
    }
    // irradiance: 256^2 per face (VulkanRenderer.h:220)
    {
        const int res = 256;
        HIP_TRY(c->d_irradiance.ensure((size_t)6 * res * res * 4));
        IrradianceTables tb;
        const float PI = 3.14159265359f;
        const int nSamples = 24;
        const float stepRads = 2 * PI / nSamples;
        for (int i = 0; i < 24; i++) { const float phi = i * stepRads; tb.cos_phi[i] = cosf(phi); tb.sin_phi[i] = sinf(phi); }
        for (int j = 0; j < 6; j++) { const float theta = j * stepRads; tb.sin_theta[j] = sinf(theta); tb.cos_theta[j] = cosf(theta); }
        StreamTimer t(c->stream);
        launch_bake_irradiance(c->stream, c->env_b, c->d_irradiance.as<uint32_t>(), res, tb);
        c->precompute_ms[1] = t.stop();
        memset(&c->irradiance, 0, sizeof c->irradiance);
        c->irradiance.texels = c->d_irradiance.as<uint32_t>(); c->irradiance.res = res; c->irradiance.levels = 1;
    }
    // GGX prefilter: 512^2, 10 levels, roughness = level / 9 (VulkanRenderer.h:225, …Environment.cpp:350-360)
    {
        const int base = 512, levels = 10;
        size_t off = 0;
        memset(&c->prefiltered, 0, sizeof c->prefiltered);
        for (int l = 0; l < levels; l++) { c->prefiltered.level_offset[l] = (uint32_t)off; off += (size_t)6 * (base >> l) * (base >> l); }
        HIP_TRY(c->d_prefiltered.ensure(off * 4));
        std::vector<float> trig(256);
        const float PI = 3.14159265359f;
        for (int i = 0; i < 128; i++) {
            const float xi_x = (float)i / (float)128;
            const float phi = 2.0f * PI * xi_x;
            trig[i] = cosf(phi); trig[128 + i] = sinf(phi);
        }
        HIP_TRY(c->d_trig.ensure(256 * 4));
        HIP_TRY(hipMemcpyAsync(c->d_trig.p, trig.data(), 256 * 4, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        StreamTimer t(c->stream);
        for (int l = 0; l < levels; l++)
            launch_bake_prefilter(c->stream, c->env_b, c->d_prefiltered.as<uint32_t>() + c->prefiltered.level_offset[l], base >> l, (float)l / 9,
                                  c->d_trig.as<float>());
        c->precompute_ms[2] = t.stop();
        c->prefiltered.texels = c->d_prefiltered.as<uint32_t>(); c->prefiltered.res = base; c->prefiltered.levels = levels;
    }
    // bordered copies for the shading kernels
    auto add_border = [&](const DCube& src, DevBuf& buf, DCube& dst) -> int {
        dst = src;
        size_t off = 0;
        for (uint32_t l = 0; l < src.levels; l++) {
            const size_t r = std::max(1u, src.res >> l);
            dst.level_offset[l] = (uint32_t)off;
            off += 6 * (r + 2) * (r + 2);
        }
        HIP_TRY(buf.ensure(off * 4 + 16));  // the shading kernels read texel pairs: one texel of slack
        for (uint32_t l = 0; l < src.levels; l++)
            launch_cube_border(c->stream, src.texels + src.level_offset[l], buf.as<uint32_t>() + dst.level_offset[l], (int)std::max(1u, src.res >> l));
        dst.texels = buf.as<uint32_t>();
        return RDR_OK;
    };
    RDR_TRY(add_border(c->irradiance, c->d_irradiance_b, c->irradiance_b));
    RDR_TRY(add_border(c->prefiltered, c->d_prefiltered_b, c->prefiltered_b));
    auto add_quads = [&](const DCube& bordered, DevBuf& buf, DCube& dst) -> int {
        dst = bordered;
        size_t off = 0;
        for (uint32_t l = 0; l < bordered.levels; l++) {
            const size_t r = std::max(1u, bordered.res >> l);
            dst.level_offset[l] = (uint32_t)off;
            off += 6 * (r + 1) * (r + 1);
        }
        HIP_TRY(buf.ensure(off * 16));
        for (uint32_t l = 0; l < bordered.levels; l++)
            launch_cube_quads(c->stream, bordered.texels + bordered.level_offset[l], buf.as<uint8_t>() + (size_t)dst.level_offset[l] * 16,
                              (int)std::max(1u, bordered.res >> l));
        dst.texels = buf.as<uint32_t>();
        return RDR_OK;
    };
    RDR_TRY(add_quads(c->irradiance_b, c->d_irradiance_q, c->irradiance_q));
    RDR_TRY(add_quads(c->prefiltered_b, c->d_prefiltered_q, c->prefiltered_q));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipGetLastError());
    return RDR_OK;
}

int rdr_set_environment_map(rdr_ctx* c, const float* const faces[6], uint32_t w, uint32_t h) {
    if (!c || !faces || w == 0 || w != h) return fail(c, RDR_ERR_INVALID, "set_environment_map: faces must be square");
    for (int f = 0; f < 6; f++) if (!faces[f]) return fail(c, RDR_ERR_INVALID, "set_environment_map: null face");
    HIP_TRY(hipSetDevice(c->device));
    RDR_TRY(finish_frames_in_flight(c));
    const size_t face_px = (size_t)w * w;
    HIP_TRY(c->d_env.ensure(6 * face_px * 4));
    DevBuf staging;
    HIP_TRY(staging.ensure(face_px * 16));
    for (int f = 0; f < 6; f++) {
        HIP_TRY(hipMemcpyAsync(staging.p, faces[f], face_px * 16, hipMemcpyHostToDevice, c->stream));
        launch_env_pack(c->stream, staging.p, c->d_env.as<uint32_t>() + f * face_px, (uint32_t)face_px);
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    staging.release();
    return bake_environment(c, w);
}

int rdr_set_environment_equirect(rdr_ctx* c, const float* pano, uint32_t w, uint32_t h, uint32_t face_res) {
    if (!c || !pano || w == 0 || h == 0 || w > 32768 || h > 32768 || face_res == 0 || face_res > 8192)
        return fail(c, RDR_ERR_INVALID, "set_environment_equirect: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    RDR_TRY(finish_frames_in_flight(c));
    HIP_TRY(c->d_env.ensure((size_t)6 * face_res * face_res * 4));
    DevBuf staging;
    HIP_TRY(staging.ensure((size_t)w * h * 16));
    HIP_TRY(hipMemcpyAsync(staging.p, pano, (size_t)w * h * 16, hipMemcpyHostToDevice, c->stream));
    {
        StreamTimer t(c->stream);
        launch_equirect_to_cube(c->stream, staging.p, (int)w, (int)h, c->d_env.as<uint32_t>(), (int)face_res);
        c->precompute_ms[0] = t.stop();
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    staging.release();
    return bake_environment(c, face_res);
}

// footprint-major copy of the BRDF table in d_lut (what the shading kernels read)
static int make_lut_quads(rdr_ctx* c) {
    HIP_TRY(c->d_lut_q.ensure((size_t)c->lut_w * c->lut_h * 16));
    launch_lut_quads(c->stream, c->d_lut.as<uint32_t>(), c->d_lut_q.p, (int)c->lut_w, (int)c->lut_h);
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipGetLastError());
    return RDR_OK;
}

int rdr_generate_brdf_lut(rdr_ctx* c, uint32_t size, uint32_t samples) {
    if (!c || size == 0 || size > 4096 || samples == 0 || samples > 65536) return fail(c, RDR_ERR_INVALID, "generate_brdf_lut: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    RDR_TRY(finish_frames_in_flight(c));
    std::vector<float> trig(2 * (size_t)samples);
    for (uint32_t s = 0; s < samples; s++) {
        const float phi = 2.0f * 3.14159265359f * ((float)s / (float)samples);
        trig[s] = cosf(phi); trig[samples + s] = sinf(phi);
    }
    DevBuf d_trig;
    HIP_TRY(d_trig.ensure(trig.size() * 4));
    HIP_TRY(hipMemcpyAsync(d_trig.p, trig.data(), trig.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c->d_lut.ensure((size_t)size * size * 4 + 16));  // spare words: the shading kernels read texel pairs
    {
        StreamTimer t(c->stream);
        launch_brdf_lut(c->stream, c->d_lut.as<uint32_t>(), (int)size, (int)samples, d_trig.as<float>());
        c->precompute_ms[3] = t.stop();
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipGetLastError());
    d_trig.release();
    c->lut_w = c->lut_h = size;
    return make_lut_quads(c);
}

int rdr_get_precompute_ms(rdr_ctx* c, float out[4]) {
    if (!c || !out) return RDR_ERR_INVALID;
    memcpy(out, c->precompute_ms, sizeof c->precompute_ms);
    return RDR_OK;
}

int rdr_read_brdf_lut(rdr_ctx* c, uint8_t* out, uint32_t* ow, uint32_t* oh) {
    if (!c || !c->lut_w) return fail(c, RDR_ERR_INVALID, "read_brdf_lut: no table");
    if (ow) *ow = c->lut_w;
    if (oh) *oh = c->lut_h;
    if (out) {
        HIP_TRY(hipSetDevice(c->device));
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipMemcpy(out, c->d_lut.p, (size_t)c->lut_w * c->lut_h * 4, hipMemcpyDeviceToHost));
    }
    return RDR_OK;
}

int rdr_set_brdf_lut(rdr_ctx* c, const uint8_t* px, uint32_t w, uint32_t h, uint32_t channels) {
    if (!c || !px || w == 0 || h == 0 || (channels != 3 && channels != 4)) return fail(c, RDR_ERR_INVALID, "set_brdf_lut: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    RDR_TRY(finish_frames_in_flight(c));
    std::vector<uint8_t> rgba((size_t)w * h * 4);
    for (size_t i = 0; i < (size_t)w * h; i++) {
        for (int k = 0; k < 3; k++) rgba[i * 4 + k] = px[i * channels + k];
        rgba[i * 4 + 3] = channels == 4 ? px[i * 4 + 3] : 255;
    }
    HIP_TRY(c->d_lut.ensure(rgba.size() + 16));  // spare words: the shading kernels read texel pairs
    HIP_TRY(hipMemcpy(c->d_lut.p, rgba.data(), rgba.size(), hipMemcpyHostToDevice));
    c->lut_w = w; c->lut_h = h;
    return make_lut_quads(c);
}

int rdr_set_camera(rdr_ctx* c, const rdr_camera* cam) {
    if (!c || !cam) return fail(c, RDR_ERR_INVALID, "set_camera: null");
    c->camera.type = cam->type;
    c->camera.position = make3(cam->position[0], cam->position[1], cam->position[2]);
    c->camera.euler = make3(cam->euler[0], cam->euler[1], cam->euler[2]);
    c->camera.near_plane = cam->near_plane; c->camera.far_plane = cam->far_plane;
    c->camera.fov_or_height = cam->fov_or_height; c->camera.aspect = cam->aspect_ratio;
    return RDR_OK;
}

int rdr_set_exposure_bias(rdr_ctx* c, float bias, float gamma) {
    if (!c) return RDR_ERR_INVALID;
    c->settings.exposure_bias = bias; c->settings.gamma = gamma;
    return RDR_OK;
}

static int render_frame_impl(rdr_ctx* c, const rdr_frame_params* fp, uint8_t* out_rgba8, int out_is_device, float* out_avg) {
    const double t_enter = host_now();
    c->host_wait_s = 0.0;
    c->frame_claimed = false;
    HIP_TRY(hipSetDevice(c->device));
    const uint32_t W = c->W, H = c->H, S = c->settings.msaa;
    const size_t n = (size_t)W * H;
    // The previous frame's vertex stage (k_shade_setup on stream3) reads the vertex arena and the pair counts, which
    // exist once: nothing of this frame (geometry uploaded again, skinned vertices, set-up) may overtake it.
    if (c->frames > 0) HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_vertex_out, 0));
    RDR_TRY(upload_scene(c));
    RDR_TRY(run_skinning(c));
    RDR_TRY(ensure_post_targets(c));
    {
        const int slot = (int)(c->frames % RDR_FRAME_SLOTS);
        if (c->pending[slot].on) {  // the last frame of this slot was never picked up: it has long finished
            HIP_TRY(hipEventSynchronize(c->ev_sets[slot][5]));
            harvest(c, slot);
        }
        // from here on `slot` is this frame's (rdr_render_frame's error path drops the frame in c->slot only then)
        c->slot = slot;
        c->ev = c->ev_sets[slot];
        c->frame_claimed = true;
        c->tr_host[c->slot][0] = t_enter;
    }
    // this frame writes the buffer set the last frame of its slot used: the sets rotate, the oldest comes to the front
    for (int a = 0; a < RDR_FRAME_SLOTS - 1; a++) {
        rdr_ctx::AltBuffers& o = c->alt[a];
        std::swap(c->d_batch_tables, o.batch_tables); std::swap(c->d_frame, o.frame); std::swap(c->d_shade_tris, o.shade_tris);
        std::swap(c->d_recs, o.recs); std::swap(c->d_tile_start, o.tile_start); std::swap(c->d_depth, o.depth); std::swap(c->d_vis, o.vis);
        std::swap(c->d_dir_shadow, o.dir_shadow); std::swap(c->d_hdr, o.hdr);
        // the 4x rasteriser of this frame fills the fragment lists while the frames before are still shaded from their own
        std::swap(c->d_ofrag_recs, o.ofrag_recs); std::swap(c->d_pixel_state, o.pixel_state); std::swap(c->d_tile_first, o.tile_first);
        for (int k = 0; k < 2; k++) { std::swap(c->d_keys[k], o.keys[k]); std::swap(c->d_vals[k], o.vals[k]); }
    }
    HIP_TRY(c->d_depth.ensure(n * S * 4));
    // The fragment lists of the opaque pass are filled by the 4x rasteriser from the tiles' samples while they are in
    // LDS (RDR_FUSED_CLASSIFY=0: by k_frag_classify from the visibility image, a measurement aid). The rasteriser then
    // does not store the visibility image at all: rdr_read_visibility rebuilds it from the lists.
    bool fused_classify = S == 4;
    if (const char* e = getenv("RDR_FUSED_CLASSIFY")) fused_classify = fused_classify && atoi(e) != 0;
    if (!fused_classify) HIP_TRY(c->d_vis.ensure(n * S * 4));
    c->vis_in_lists = fused_classify;
    HIP_TRY(c->d_hdr.ensure(n * 8));
    c->stats = rdr_frame_stats{};
    c->stats.frame_index = ++c->frames;
    c->frame_pairs_before = 0;
    c->views_reused = 0;
    {
        const char* e = getenv("RDR_DEVICE_DRAW_LISTS");
        c->device_draws = e ? atoi(e) != 0 : c->prims.size() >= 2048;
    }
    HIP_TRY(hipMemsetAsync(frame_counters(c), 0, RDR_COUNTER_SLOT_BYTES, c->stream));
    if (c->device_draws) RDR_TRY(upload_cull_tables(c));  // the candidate lists the views below refer to
    HIP_TRY(hipEventRecord(c->ev[0], c->stream));
    const vec3 cam_pos = c->camera.position;
    const mat4 viewproj = c->camera.view_proj();

    // ---- shadow views, the main opaque view and the blended draws of the main view, binned as one batch.
    // The blended "view" shares the main depth target; k_raster leaves its tiles alone (load == 2) and
    // k_blend_collect walks them after the opaque shading.
    Batch b;
    if (c->lights_set) {
        RDR_TRY(point_shadow_views(c, cam_pos, b));
        RDR_TRY(cascade_views(c, b));
    }
    std::vector<uint32_t> opaque, non_opaque;
    for (uint32_t p = 0; p < c->prims.size(); p++) (c->prims[p].mat.alpha_mode == RDR_ALPHA_OPAQUE ? opaque : non_opaque).push_back(p);
    const uint32_t main_view = (uint32_t)b.views.size();
    b.views.push_back(make_view(c->d_depth.as<float>(), fused_classify ? nullptr : c->d_vis.as<uint32_t>(), W, H, S, 2, 0));
    // the order of these two lists shows: visibility ids follow it, the earlier draw keeps a depth tie, blending is
    // back to front (VulkanRenderer.cpp:945-963)
    add_view_draws(c, b, CAND_OPAQUE, opaque, viewproj, cam_pos, SORT_FRONT_TO_BACK, main_view, false);
    const size_t draws_before_blend = b.draws.size();
    if (!non_opaque.empty()) add_view_draws(c, b, CAND_NON_OPAQUE, non_opaque, viewproj, cam_pos, SORT_BACK_TO_FRONT, main_view + 1, false);
    // the blended "view" exists when blended draws do (host path) or may (device path: their number comes back later)
    const bool blend_view = c->device_draws ? !non_opaque.empty() : b.draws.size() > draws_before_blend;
    if (blend_view) b.views.push_back(make_view(c->d_depth.as<float>(), nullptr, W, H, S, 2, 2));
    // Frame constants, then the vertex stage of the main view's triangles: enqueued from inside the batch, right after
    // its pair scan (which is all k_shade_setup needs from it), so that they run while the host waits for the pair count.
    size_t first_main_draw = 0, n_main_draws = 0, first_blend_draw = 0, n_blend_draws = 0;
    bool have_blend = false;
    uint32_t main_first_instance = 0, n_main_tris = 0, blend_first_instance = 0, n_blend_tris = 0;
    DShadeTri *tris_main = nullptr, *tris_blend = nullptr;
    auto enqueue_shade_setup = [&]() -> int {
        // where the batch's tables hold the two lists (triangle records of a view are contiguous in d_recs, in
        // visibility-id order)
        const DDrawListInfo& mi = b.info[main_view];
        first_main_draw = mi.first_draw; n_main_draws = mi.n_draws; main_first_instance = mi.first_instance; n_main_tris = mi.n_tris;
        if (blend_view) {
            const DDrawListInfo& bi = b.info[main_view + 1];
            first_blend_draw = bi.first_draw; n_blend_draws = bi.n_draws; blend_first_instance = bi.first_instance; n_blend_tris = bi.n_tris;
        }
        have_blend = n_blend_draws != 0;
        c->stats.draws_main = (uint32_t)(n_main_draws + n_blend_draws);
        DFrame fr;
        memset(&fr, 0, sizeof fr);
        fr.W = W; fr.H = H; fr.samples = S; fr.n_lights = (uint32_t)c->lights.size();
        fr.inv2W = 2.0f / (float)W; fr.inv2H = 2.0f / (float)H;
        fr.cam[0] = cam_pos.x; fr.cam[1] = cam_pos.y; fr.cam[2] = cam_pos.z;
        const mat4 inv = mat_inverse(viewproj);
        memcpy(fr.inv_viewproj, inv.m, 64);
        fr.n_cascades = c->settings.directional_shadow_cascade_levels;
        for (uint32_t i = 0; i < fr.n_cascades; i++) memcpy(fr.csm_viewproj[i], c->csm_viewproj[i].m, 64);
        fr.dir_res = c->settings.directional_shadow_map_resolution; fr.point_res = c->settings.point_shadow_map_resolution;
        fr.lights_set = c->lights_set ? 1u : 0u;
        if (c->lights_set) {
            const quat q = {c->dir_light.orientation_wxyz[0], c->dir_light.orientation_wxyz[1], c->dir_light.orientation_wxyz[2], c->dir_light.orientation_wxyz[3]};
            const vec3 L = normalize(quat_rotate(q, make3(0.0f, 0.0f, 1.0f)));  // VulkanRenderer.cpp:263 + pbr.frag:179-180
            fr.dir_L[0] = L.x; fr.dir_L[1] = L.y; fr.dir_L[2] = L.z;
            memcpy(fr.dir_I, c->dir_light.intensity, 12);
        }
        fr.dir_shadow = c->d_dir_shadow.as<float>(); fr.point_shadow = c->d_point_shadow.as<float>();
        std::vector<DPointLight> dl(std::max<size_t>((c->lights.size() + 1) / 2, 1));
        memset(dl.data(), 0, dl.size() * sizeof(DPointLight));
        for (size_t i = 0; i < c->lights.size(); i++) {
            const HostLight& l = c->lights[i];
            DPointLight& d = dl[i / 2];
            const int j = (int)(i & 1);
            d.px[j] = l.point.x; d.py[j] = l.point.y; d.pz[j] = l.point.z;
            d.ix[j] = l.intensity.x; d.iy[j] = l.intensity.y; d.iz[j] = l.intensity.z;
            d.flags[j] = l.flags; d.slot[j] = l.slot;
            if (j == 0) { d.px[1] = d.px[0]; d.py[1] = d.py[0]; d.pz[1] = d.pz[0]; }  // stands in for a missing second light
        }
        const size_t sz_l = dl.size() * sizeof(DPointLight), off_l = (sizeof(DFrame) + 15) & ~(size_t)15;
        HIP_TRY(c->d_frame.ensure(off_l + sz_l));
        c->d_lights.p = c->d_frame.as<uint8_t>() + off_l;
        fr.lights = c->d_lights.as<DPointLight>();
        fr.env = c->env_b; fr.irradiance = c->irradiance_q; fr.prefiltered = c->prefiltered_q;  // bordered / footprint-major copies
        fr.lut = c->d_lut_q.as<uint8_t>(); fr.lut_w = c->lut_w; fr.lut_h = c->lut_h;
        fr.textures = c->d_textures.as<DTexture>(); fr.prims = c->d_prims.as<DPrim>();
        fr.pos = c->d_pos.as<float4>(); fr.attr = c->d_attr.as<DVertexAttr>(); fr.idx = c->d_idx.as<uint32_t>();
        fr.srgb_decode = c->d_tables.as<float>(); fr.unorm_decode = c->d_tables.as<float>() + 256;
        fr.tex_white = 0; fr.tex_normal = 1;
        // its own pinned block, one half per frame slot: the copy out of a half was enqueued two frames ago, in front
        // of work the previous frame's pair-count wait has since seen finished
        const size_t half = (off_l + sz_l + 63) & ~(size_t)63;
        HIP_TRY(c->h_frame.ensure(RDR_FRAME_SLOTS * half));
        uint8_t* st = c->h_frame.as<uint8_t>() + (size_t)c->slot * half;
        memcpy(st, &fr, sizeof fr);
        memcpy(st + off_l, dl.data(), sz_l);
        HIP_TRY(hipMemcpyAsync(c->d_frame.p, st, off_l + sz_l, hipMemcpyHostToDevice, c->stream));

        // d_draws / d_first_tri hold the whole batch; a view's draws are a slice
        HIP_TRY(c->d_shade_tris.ensure(std::max<size_t>((size_t)n_main_tris + n_blend_tris, 1) * sizeof(DShadeTri)));
        tris_main = c->d_shade_tris.as<DShadeTri>();
        tris_blend = tris_main + n_main_tris;
        // The vertex stage needs the pair scan and the frame constants, and only the shading needs the vertex stage: it
        // goes to the stream the 4x rasteriser will take later (idle until the sort is done), beside the pair expansion
        // and the sort instead of in front of them (RDR_VERTEX_STAGE_IN_LINE=1: on the front-end stream, a measurement aid).
        hipStream_t vs = env_on("RDR_VERTEX_STAGE_IN_LINE") ? c->stream : c->stream3;
        if (vs != c->stream) {
            HIP_TRY(hipEventRecord(c->ev_vertex_in, c->stream));
            HIP_TRY(hipStreamWaitEvent(vs, c->ev_vertex_in, 0));
        }
        launch_shade_setup(vs, c->d_frame.as<DFrame>(), c->d_draws.as<DDraw>() + first_main_draw, c->d_first_tri.as<uint32_t>() + first_main_draw,
                           (uint32_t)n_main_draws, n_main_tris, c->d_pair_count.as<uint32_t>() + main_first_instance,
                           c->batch_instances - main_first_instance, frame_counters(c), tris_main);
        if (have_blend)
            launch_shade_setup(vs, c->d_frame.as<DFrame>(), c->d_draws.as<DDraw>() + first_blend_draw,
                               c->d_first_tri.as<uint32_t>() + first_blend_draw, (uint32_t)n_blend_draws, n_blend_tris,
                               c->d_pair_count.as<uint32_t>() + blend_first_instance, c->batch_instances - blend_first_instance,
                               frame_counters(c), tris_blend);
        HIP_TRY(hipEventRecord(c->ev_vertex_out, vs));
        return RDR_OK;
    };
    const size_t frag_slots = shade_frag_slots(W, H, (int)S);
    const uint32_t n_screen_tiles = ((W + RDR_TILE - 1) / RDR_TILE) * ((H + RDR_TILE - 1) / RDR_TILE);
    if (frag_slots >= (1ull << 32)) return fail(c, RDR_ERR_CAPACITY, "render_frame: the image has more than 2^32 samples");
    HIP_TRY(c->d_ofrag_recs.ensure(frag_slots * 8));
    HIP_TRY(c->d_ofrag_cols.ensure(frag_slots * 8));
    HIP_TRY(c->d_pixel_state.ensure((size_t)n_screen_tiles * 256 * 4));
    HIP_TRY(c->d_tile_first.ensure((size_t)n_screen_tiles * 4));
    uint32_t* const list_count = frame_counters(c) + RDR_COUNTER_LISTS / 4u;
    const FragLists L = make_frag_lists(W, H, (int)S, c->d_ofrag_recs.p, c->d_ofrag_cols.p, c->d_pixel_state.as<uint32_t>(), c->d_tile_first.as<uint32_t>(),
                                  list_count, list_count + RDR_FRAG_SEGS * RDR_FRAG_CTR_STRIDE);
    int cur = 0;
    RDR_TRY(raster_batch(c, b, &cur, enqueue_shade_setup, fused_classify ? &L : nullptr, true));
    if (blend_view) c->stats.views -= 1;  // the blended draws belong to the main view, already counted
    HIP_TRY(hipEventRecord(c->ev[1], c->raster_stream ? c->raster_stream : c->stream));
    const uint32_t blend_tile_base = blend_view ? b.views.back().tile_base : 0u;

    // ---- opaque shading + skybox
    if (S == 4 && have_blend) HIP_TRY(c->d_samples.ensure(n * 4 * 8));
    // from here on the frame runs on the shading stream, behind the tile rasterisers of its own geometry phase and, in stream order,
    // behind the previous frame's shading and tail
    HIP_TRY(hipStreamWaitEvent(c->stream2, c->ev_raster1, 0));
    if (c->raster_side) HIP_TRY(hipStreamWaitEvent(c->stream2, c->ev_join, 0));
    HIP_TRY(hipStreamWaitEvent(c->stream2, c->ev_vertex_out, 0));
    // What follows the shading kernel (resolve, blended pass) goes to the tail stream in front of the post chain: the
    // shading stream then holds only shadow-view rasterisers and shading kernels, and the next frame's rasteriser does
    // not queue behind this frame's blended pass. The fragment colours exist once, so the shading kernel waits for
    // the resolve of the frame before (ev[9] of its slot).
    hipStream_t ts = c->tail_split ? c->stream4 : c->stream2;
    {
        if (!fused_classify) launch_frag_classify(c->stream2, (int)S, c->d_frame.as<DFrame>(), c->d_vis.as<uint32_t>(), L);
        if (ts != c->stream2 && c->pending[prev_slot(c)].on) HIP_TRY(hipStreamWaitEvent(c->stream2, c->ev_sets[prev_slot(c)][9], 0));
        HIP_TRY(hipEventRecord(c->ev[6], c->stream2));
        launch_shade_frags(c->stream2, c->d_frame.as<DFrame>(), c->d_lights.as<DPointLight>(), tris_main, L, c->shade_blocks, c->float_images);
        HIP_TRY(hipEventRecord(c->ev[7], c->stream2));
        if (ts != c->stream2) HIP_TRY(hipStreamWaitEvent(ts, c->ev[7], 0));
        launch_frag_resolve(ts, (int)S, c->d_frame.as<DFrame>(), L, c->d_hdr.p, (S == 4 && have_blend) ? c->d_samples.p : nullptr,
                            c->d_tile_start.as<uint32_t>() + blend_tile_base, c->d_tile_end.as<uint32_t>() + blend_tile_base, W, H);
        HIP_TRY(hipEventRecord(c->ev[9], ts));
    }
    HIP_TRY(hipGetLastError());

    // ---- blended draws, back to front; resolves its tiles into hdr
    if (have_blend) {
        const uint32_t n_tiles = ((W + RDR_TILE - 1) / RDR_TILE) * ((H + RDR_TILE - 1) / RDR_TILE);
        // Fragment list slots: every block with blended triangles (at most one per pair, at most n_tiles) starts with
        // one run per wave; a wave moves to a fresh run only when fewer than 64 slots are left in its run, so the runs
        // it leaves behind are at least half full; fragments are at most the pixels of the triangles' boxes.
        const uint64_t blocks = std::min<uint64_t>(n_tiles, c->blend_pairs);
        const uint64_t bound = blocks * 4u * RDR_FRAG_RUN + 2u * c->blend_box_pixels + RDR_FRAG_RUN;
        HIP_TRY(c->d_frag_head.ensure(n * 4));
        auto run_group = [&](uint32_t tile0, uint32_t count, uint32_t capacity) {
            launch_raster_blend(ts, (int)S, c->d_frame.as<DFrame>(), c->d_lights.as<DPointLight>(), c->d_views.as<DView>() + (main_view + 1),
                                tris_blend, c->d_recs.as<DTriRec>(), c->d_vals[cur].as<uint32_t>(), c->d_tile_start.as<uint32_t>(),
                                c->d_tile_end.as<uint32_t>(), tile0, count, S == 4 ? c->d_samples.p : c->d_hdr.p, c->d_hdr.p, c->d_frags.p,
                                c->d_frag_col.p, c->d_frag_head.as<uint32_t>(), frame_counters(c) + RDR_COUNTER_FRAGS, capacity, c->float_images);
        };
        if (bound <= c->blend_budget_slots) {
            HIP_TRY(c->d_frags.ensure(bound * 16));
            HIP_TRY(c->d_frag_col.ensure(bound * 16));
            run_group(0u, n_tiles, (uint32_t)bound);
        } else {
            // The bound is more than this context allocates for the list (many screen-filling layers: the bound counts
            // every triangle's whole box): a dry run counts the slots each tile really takes, and the tiles are then
            // rendered in groups that fit the list, one collect / shade / resolve sequence per group. Blending is per
            // pixel, so the order of the groups does not matter.
            HIP_TRY(c->d_tile_slots.ensure((size_t)n_tiles * 4));
            HIP_TRY(hipMemsetAsync(c->d_tile_slots.p, 0, (size_t)n_tiles * 4, ts));
            launch_blend_count(ts, (int)S, c->d_views.as<DView>() + (main_view + 1), c->d_recs.as<DTriRec>(), c->d_vals[cur].as<uint32_t>(),
                               c->d_tile_start.as<uint32_t>(), c->d_tile_end.as<uint32_t>(), n_tiles, c->d_tile_slots.as<uint32_t>());
            std::vector<uint32_t> slots(n_tiles);
            HIP_TRY(hipMemcpyAsync(slots.data(), c->d_tile_slots.p, (size_t)n_tiles * 4, hipMemcpyDeviceToHost, ts));
            HIP_TRY(hipStreamSynchronize(ts));
            uint64_t largest = 0;
            for (uint32_t t = 0; t < n_tiles; t++) largest = std::max<uint64_t>(largest, slots[t]);
            const uint64_t list = std::max<uint64_t>(c->blend_budget_slots, largest) + RDR_FRAG_RUN;
            if (list >= (1ull << 31)) return fail(c, RDR_ERR_CAPACITY, "blended pass: one 16x16 tile alone needs more than 2^31 fragment slots");
            HIP_TRY(c->d_frags.ensure(list * 16));
            HIP_TRY(c->d_frag_col.ensure(list * 16));
            for (uint32_t t0 = 0; t0 < n_tiles;) {
                uint64_t sum = 0;
                uint32_t t1 = t0;
                while (t1 < n_tiles && (t1 == t0 || sum + slots[t1] <= c->blend_budget_slots)) sum += slots[t1++];
                if (sum) {
                    if (t0) HIP_TRY(hipMemsetAsync(frame_counters(c) + RDR_COUNTER_FRAGS, 0, 4, ts));
                    run_group(t0, t1 - t0, (uint32_t)list);
                    c->stats.blend_groups++;
                }
                t0 = t1;
            }
        }
        HIP_TRY(hipGetLastError());
    }

    // ---- bloom, tonemap, luminance
    if (c->have_frame && (!fp || fp->exposure_mode == 0)) {  // previous frame's luminance
        const double t_wait = host_now();
        RDR_TRY(rdr_synchronize(c));
        c->host_wait_s += host_now() - t_wait;
    }
    const float exposure = compute_exposure(c, fp);
    // a device destination receives the frame straight from the tonemap kernel (its two-pixel stores need the alignment)
    uint32_t* direct_out = (out_rgba8 && out_is_device && ((uintptr_t)out_rgba8 & (out_align(c) - 1u)) == 0) ? (uint32_t*)out_rgba8 : nullptr;
    HIP_TRY(hipEventRecord(c->ev_shaded[c->slot], ts));  // the shadow maps and the geometry phase's outputs have been read
    HIP_TRY(hipStreamWaitEvent(c->stream4, c->ev_shaded[c->slot], 0));  // the tail has its own stream: the next frame's shading does not queue behind it
    enqueue_post(c, c->stream4, exposure, direct_out);
    HIP_TRY(hipGetLastError());
    c->have_frame = true;
    c->stats.exposure = exposure;
    HIP_TRY(c->h_read.ensure(RDR_READ_BYTES));
    // the average luminance (word RDR_COUNTER_LUM) and the spread counters, one copy
    HIP_TRY(hipMemcpyAsync(c->h_read.as<uint8_t>() + (size_t)c->slot * RDR_COUNTER_BYTES, frame_counters(c), RDR_COUNTER_BYTES,
                           hipMemcpyDeviceToHost, c->stream4));
    if (out_rgba8 && !direct_out)
        HIP_TRY(hipMemcpyAsync(out_rgba8, c->d_out.p, n * out_pixel_bytes(c), out_is_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream4));
    HIP_TRY(hipEventRecord(c->ev[5], c->stream4));
    // host time of this frame: everything this call did up to here, less the time it sat waiting for the GPU
    c->stats.host_ms = (float)((host_now() - t_enter - c->host_wait_s) * 1e3);
    c->stats.views_reused = c->views_reused;
    c->stats.draw_lists_on_device = c->device_draws ? 1u : 0u;
    c->pending[c->slot].on = true;
    c->pending[c->slot].host = c->stats;
    // The frame `lag` calls back has had the whole set-up of the frames since to finish its tail: its results are
    // picked up now, so that it is complete (image, luminance, counters) when this call returns.
    c->tr_host[c->slot][4] = c->tr_host[c->slot][5] = host_now();
    for (int age = RDR_FRAME_SLOTS - 1; age >= c->lag; age--) {
        const int s = (c->slot + RDR_FRAME_SLOTS - age) % RDR_FRAME_SLOTS;
        if (!c->pending[s].on) continue;
        HIP_TRY(hipEventSynchronize(c->ev_sets[s][5]));
        c->tr_host[c->slot][5] = host_now();
        harvest(c, s);
    }
    if (c->sort_failed) {
        c->sort_failed = false;
        return fail(c, RDR_ERR_DEVICE, "render_frame: the pair sort of a finished frame gave up waiting for one of its own blocks; that frame is not valid");
    }
    // With a device destination the call only enqueues; rdr_synchronize(), or the next frame, picks up the luminance
    // and the counters.
    if (out_is_device && out_rgba8) return RDR_OK;
    RDR_TRY(rdr_synchronize(c));
    if (out_avg) *out_avg = c->last_avg_log_lum;
    return RDR_OK;
}

int rdr_render_frame(rdr_ctx* c, const rdr_frame_params* fp, uint8_t* out_rgba8, int out_is_device, float* out_avg) {
    if (!c) return RDR_ERR_INVALID;
    const int rc = render_frame_impl(c, fp, out_rgba8, out_is_device, out_avg);
    if (rc != RDR_OK) {
        // The frame failed part-way (a buffer could not grow, a capacity limit): what it had enqueued is drained and
        // it is dropped, so that the context is left as between two frames. An older frame still in flight keeps its
        // results; the failed frame's index is not reused.
        const std::string keep = c->err;
        (void)hipStreamSynchronize(c->stream); (void)hipStreamSynchronize(c->stream3);
        (void)hipStreamSynchronize(c->stream2); (void)hipStreamSynchronize(c->stream4);
        (void)hipGetLastError();
        // A frame that failed before it took its slot (upload, skinning, target allocation) leaves c->slot naming the
        // frame before it, which is intact and is handed over like the other older ones.
        if (c->frame_claimed) c->pending[c->slot].on = false;
        if (c->d_scan_status.p) (void)hipMemset(c->d_scan_status.p, 0, c->d_scan_status.cap);  // a scan cut short leaves its status words behind
        for (rdr_ctx::ShadowKey& k : c->cascades_key) k.valid = false;  // targets may be half written
        c->point_dynamic_key.valid = false;
        for (int age = RDR_FRAME_SLOTS - 1; age >= (c->frame_claimed ? 1 : 0); age--) harvest(c, (c->slot + RDR_FRAME_SLOTS - age) % RDR_FRAME_SLOTS);
        c->err = keep;
    }
    return rc;
}

int rdr_synchronize(rdr_ctx* c) {
    if (!c) return RDR_ERR_INVALID;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream3));
    HIP_TRY(hipStreamSynchronize(c->stream2));
    HIP_TRY(hipStreamSynchronize(c->stream4));
    // the oldest frame first, so that done_stats ends up holding the newest
    for (int age = RDR_FRAME_SLOTS - 1; age >= 0; age--) harvest(c, (c->slot + RDR_FRAME_SLOTS - age) % RDR_FRAME_SLOTS);
    return RDR_OK;
}

int rdr_wait_frame(rdr_ctx* c, uint64_t frame_index) {
    if (!c) return RDR_ERR_INVALID;
    if (frame_index > c->frames) return fail(c, RDR_ERR_INVALID, "wait_frame: that frame has not been enqueued");
    HIP_TRY(hipSetDevice(c->device));
    for (int age = RDR_FRAME_SLOTS - 1; age >= 0; age--) {  // oldest first
        const int s = (c->slot + RDR_FRAME_SLOTS - age) % RDR_FRAME_SLOTS;
        if (!c->pending[s].on || c->pending[s].host.frame_index > frame_index) continue;
        HIP_TRY(hipEventSynchronize(c->ev_sets[s][5]));
        harvest(c, s);
    }
    return RDR_OK;
}

int rdr_get_frame_stats(rdr_ctx* c, rdr_frame_stats* out) {
    if (!c || !out) return RDR_ERR_INVALID;
    *out = c->done_stats;
    return RDR_OK;
}

static int read_back(rdr_ctx* c, void* dst, const void* src, size_t bytes) {
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream3));
    HIP_TRY(hipStreamSynchronize(c->stream2));
    HIP_TRY(hipStreamSynchronize(c->stream4));
    HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return RDR_OK;
}

// rows of `w` floats, `pitch` floats apart on the device, packed on the host
static int read_back_2d(rdr_ctx* c, float* dst, const float* src, size_t pitch, size_t w, size_t h) {
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream3));
    HIP_TRY(hipStreamSynchronize(c->stream2));
    HIP_TRY(hipStreamSynchronize(c->stream4));
    HIP_TRY(hipMemcpy2D(dst, w * 4, src, pitch * 4, w * 4, h, hipMemcpyDeviceToHost));
    return RDR_OK;
}

int rdr_read_depth(rdr_ctx* c, float* out) {
    if (!c || !out || !c->have_frame) return fail(c, RDR_ERR_INVALID, "read_depth: no frame");
    if (c->depth_partial) {
        HIP_TRY(hipSetDevice(c->device));
        RDR_TRY(rdr_synchronize(c));
        launch_raster(c->stream2, &c->depth_view, 1u, nullptr, c->d_recs.as<DTriRec>(), c->d_vals[c->depth_cur].as<uint32_t>(), c->d_tile_start.as<uint32_t>(),
                      c->d_tile_end.as<uint32_t>(), c->depth_range, 4, nullptr, 2u, c->depth_blend_base);
        HIP_TRY(hipGetLastError());
        c->depth_partial = false;  // the image is whole now
    }
    return read_back(c, out, c->d_depth.p, (size_t)c->W * c->H * c->settings.msaa * 4);
}
int rdr_read_visibility(rdr_ctx* c, uint32_t* out) {
    if (!c || !out || !c->have_frame) return fail(c, RDR_ERR_INVALID, "read_visibility: no frame");
    if (c->vis_in_lists) {
        const uint32_t S = c->settings.msaa;
        HIP_TRY(hipSetDevice(c->device));
        HIP_TRY(c->d_vis.ensure((size_t)c->W * c->H * S * 4));
        const FragLists L = make_frag_lists(c->W, c->H, (int)S, c->d_ofrag_recs.p, c->d_ofrag_cols.p, c->d_pixel_state.as<uint32_t>(),
                                            c->d_tile_first.as<uint32_t>(), nullptr, nullptr);
        HIP_TRY(hipStreamSynchronize(c->stream3));  // the rasteriser that filled the lists
        launch_vis_from_lists(c->stream2, (int)S, L, c->W, c->H, c->d_vis.as<uint32_t>());
        HIP_TRY(hipGetLastError());
    }
    return read_back(c, out, c->d_vis.p, (size_t)c->W * c->H * c->settings.msaa * 4);
}
int rdr_read_directional_shadow(rdr_ctx* c, uint32_t index, float* out) {
    if (!c || !out || !c->have_frame || !c->lights_set || index >= c->settings.directional_shadow_cascade_levels) return fail(c, RDR_ERR_INVALID, "read_directional_shadow");
    const uint32_t res = c->settings.directional_shadow_map_resolution;
    return read_back_2d(c, out, c->d_dir_shadow.as<float>() + index * cascade_texels(c) + cascade_origin(c), res + 4, res, res);
}
int rdr_read_point_shadow(rdr_ctx* c, uint32_t slot, uint32_t face, float* out) {
    if (!c || !out || face >= 6 || slot >= c->settings.point_shadow_map_count || !c->d_point_shadow.p) return fail(c, RDR_ERR_INVALID, "read_point_shadow");
    const uint32_t res = c->settings.point_shadow_map_resolution;
    return read_back_2d(c, out, c->d_point_shadow.as<float>() + ((size_t)slot * 6 + face) * point_face_texels(c) + point_face_origin(c), res + 2, res, res);
}
int rdr_read_hdr(rdr_ctx* c, float* out) {
    if (!c || !out || !c->have_post) return fail(c, RDR_ERR_INVALID, "read_hdr: nothing rendered");
    const size_t n = (size_t)c->W * c->H;
    std::vector<uint16_t> tmp(n * 4);
    RDR_TRY(read_back(c, tmp.data(), c->d_hdr.p, n * 8));
    for (size_t i = 0; i < n * 4; i++) out[i] = half_bits_to_float(tmp[i]);
    return RDR_OK;
}
static float ufloat_decode(uint32_t v, int mb) {
    const uint32_t e = v >> mb, m = v & ((1u << mb) - 1u);
    uint32_t bits;
    if (e == 0) return (float)m * ldexpf(1.0f, -14 - mb);
    if (e == 31) bits = 0x7F800000u | (m << (23 - mb));
    else bits = ((e + 112u) << 23) | (m << (23 - mb));
    float f; memcpy(&f, &bits, 4); return f;
}
static void unpack_rgb(const std::vector<uint32_t>& src, float* out) {
    for (size_t i = 0; i < src.size(); i++) {
        out[i * 3] = ufloat_decode(src[i] & 0x7FFu, 6);
        out[i * 3 + 1] = ufloat_decode((src[i] >> 11) & 0x7FFu, 6);
        out[i * 3 + 2] = ufloat_decode(src[i] >> 22, 5);
    }
}
int rdr_keep_intermediates(rdr_ctx* c, uint32_t enable) {
    if (!c) return RDR_ERR_INVALID;
    c->keep_intermediates = enable != 0;
    return RDR_OK;
}

int rdr_read_bloom(rdr_ctx* c, uint32_t level, float* out) {
    if (!c || !out || !c->have_post || level >= 6) return fail(c, RDR_ERR_INVALID, "read_bloom");
    if (level == 0 && !c->post_kept) return fail(c, RDR_ERR_INVALID, "read_bloom: mip 0 is only stored after rdr_keep_intermediates(ctx, 1)");
    std::vector<uint32_t> tmp((size_t)c->bloom_w[level] * c->bloom_h[level]);
    const uint32_t at = (c->bloom_fin_separate && (level == 3 || level == 4)) ? c->bloom_fin_offset[level - 3] : c->bloom_offset[level];
    RDR_TRY(read_back(c, tmp.data(), c->d_bloom.as<uint32_t>() + at, tmp.size() * 4));
    unpack_rgb(tmp, out);
    return RDR_OK;
}
int rdr_read_log_luminance(rdr_ctx* c, float* out) {
    if (!c || !out || !c->have_post) return fail(c, RDR_ERR_INVALID, "read_log_luminance");
    if (!c->post_kept) return fail(c, RDR_ERR_INVALID, "read_log_luminance: the whole image is only stored after rdr_keep_intermediates(ctx, 1)");
    return read_back(c, out, c->d_loglum.p, (size_t)c->W * c->H * 4);
}
int rdr_read_env_cube(rdr_ctx* c, int which, uint32_t level, float* out) {
    if (!c || !out) return RDR_ERR_INVALID;
    const DCube& cube = which == 0 ? c->env : (which == 1 ? c->irradiance : c->prefiltered);
    if (!cube.res || level >= cube.levels) return fail(c, RDR_ERR_INVALID, "read_env_cube");
    const uint32_t r = std::max(1u, cube.res >> level);
    std::vector<uint32_t> tmp((size_t)6 * r * r);
    RDR_TRY(read_back(c, tmp.data(), cube.texels + cube.level_offset[level], tmp.size() * 4));
    unpack_rgb(tmp, out);
    return RDR_OK;
}
int rdr_read_cascades(rdr_ctx* c, float* vp, float* depth) {
    if (!c || !vp || !depth || !c->lights_set || !c->have_frame) return fail(c, RDR_ERR_INVALID, "read_cascades");
    for (uint32_t i = 0; i < 4; i++) {
        if (i < c->settings.directional_shadow_cascade_levels) { memcpy(vp + 16 * i, c->csm_viewproj[i].m, 64); depth[i] = c->cascade_camera_depths[i]; }
        else { memset(vp + 16 * i, 0, 64); depth[i] = 0.0f; }
    }
    return RDR_OK;
}
int rdr_read_image_level(rdr_ctx* c, uint32_t image, uint32_t level, uint8_t* out, uint32_t* ow, uint32_t* oh) {
    if (!c || image >= c->images.size() || level >= c->images[image].levels) return fail(c, RDR_ERR_INVALID, "read_image_level");
    const HostImage& img = c->images[image];
    const uint32_t w = std::max(1u, img.w >> level), h = std::max(1u, img.h >> level);
    if (ow) *ow = w;
    if (oh) *oh = h;
    const size_t texel_bytes = img.is_float ? 16u : 4u;
    if (out) return read_back(c, out, img.pixels.as<uint8_t>() + (size_t)img.level_offset[level] * texel_bytes, (size_t)w * h * texel_bytes);
    return RDR_OK;
}

int rdr_post_process(rdr_ctx* c, const float* hdr, float exposure, uint8_t* out_rgba8, float* out_avg) {
    if (!c || !hdr) return fail(c, RDR_ERR_INVALID, "post_process: null");
    RDR_TRY(rdr_synchronize(c));  // frames in flight finish and hand over their results before their events are reused
    RDR_TRY(ensure_post_targets(c));
    const size_t n = (size_t)c->W * c->H;
    // quantise to RGBA16F on the host (round to nearest even) exactly like the colour attachment would store it
    std::vector<uint16_t> h16(n * 4);
    for (size_t i = 0; i < n * 4; i++) {
        const _Float16 q = (_Float16)hdr[i];
        memcpy(&h16[i], &q, 2);
    }
    HIP_TRY(hipMemcpyAsync(c->d_hdr.p, h16.data(), n * 8, hipMemcpyHostToDevice, c->stream));
    for (int i = 0; i < 10; i++) HIP_TRY(hipEventRecord(c->ev[i], c->stream));
    enqueue_post(c, c->stream, exposure);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (out_rgba8) HIP_TRY(hipMemcpy(out_rgba8, c->d_out.p, n * out_pixel_bytes(c), hipMemcpyDeviceToHost));
    if (out_avg) HIP_TRY(hipMemcpy(out_avg, (float*)frame_counters(c) + RDR_COUNTER_LUM, 4, hipMemcpyDeviceToHost));
    return RDR_OK;
}

int rdr_post_process_noalloc(rdr_ctx* c, const void* hdr_rgba16f, uint32_t width, uint32_t height, float exposure, uint8_t* out_rgba8,
                             float* out_avg, void* stream) {
    if (!c || !hdr_rgba16f || !out_rgba8 || !out_avg) return fail(c, RDR_ERR_INVALID, "post_process_noalloc: null argument");
    if (width == 0 || height == 0) return fail(c, RDR_ERR_INVALID, "post_process_noalloc: empty image");
    if ((uint64_t)width * height > RDR_NOALLOC_MAX_PIXELS)
        return fail(c, RDR_ERR_CAPACITY, "post_process_noalloc: more than RDR_NOALLOC_MAX_PIXELS pixels; use rdr_post_process");
    if (((uintptr_t)hdr_rgba16f & 15u) || ((uintptr_t)out_rgba8 & (out_align(c) - 1u)) || ((uintptr_t)out_avg & 3u))
        return fail(c, RDR_ERR_INVALID, "post_process_noalloc: hdr must be 16-byte, out 8-byte (16-byte for an RGBA16F output), avg 4-byte aligned");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    // carve the scratch array: six bloom mips | log-luminance image | 256 + 16 partial means
    uint32_t* bloom = (uint32_t*)c->noalloc_scratch;
    float* loglum = (float*)(bloom + NOALLOC_BLOOM_WORDS);
    float* lum = loglum + RDR_NOALLOC_MAX_PIXELS;
    uint32_t off[6], bw[6], bh[6];
    size_t words = 0;
    for (uint32_t l = 0; l < 6; l++) {
        bw[l] = std::max(1u, width >> l); bh[l] = std::max(1u, height >> l);
        off[l] = (uint32_t)words;
        words += ((size_t)bw[l] * bh[l] + 3u) & ~(size_t)3u;
    }
    const size_t fin4_at = words + (((size_t)bw[3] * bh[3] + 3u) & ~(size_t)3u);
    if (fin4_at + (size_t)bw[4] * bh[4] > NOALLOC_BLOOM_WORDS)
        return fail(c, RDR_ERR_CAPACITY, "post_process_noalloc: the six bloom mips of this image (every level at least one texel wide and high) need "
                                         "more than 1.5 x RDR_NOALLOC_MAX_PIXELS texels; use rdr_post_process");
    // the array keeps the last call's contents: a level no down-sample writes (a side shorter than 2^l) must read as zero
    for (uint32_t l = 1; l < 6; l++)
        if ((width >> l) == 0u || (height >> l) == 0u) HIP_TRY(hipMemsetAsync(bloom + off[l], 0, (size_t)bw[l] * bh[l] * 4u, st));
    // the same launches, in the same order, as enqueue_post
    uint32_t* mip[6];
    int mw[6], mh[6];
    for (int k = 0; k < 6; k++) { mip[k] = bloom + off[k]; mw[k] = (int)bw[k]; mh[k] = (int)bh[k]; }
    launch_bloom_chain(st, hdr_rgba16f, mip, mw, mh, (int)width, (int)height, bloom + words, bloom + fin4_at);
    launch_tonemap(st, hdr_rgba16f, bloom + off[1], (int)bw[1], (int)bh[1], bloom, (uint32_t*)out_rgba8, loglum, (int)width, (int)height,
                   powf(2.0f, exposure), 1.0f / c->settings.gamma, tone_curve(11.2f), log2f(6.0f), false, c->settings.hdr_output_enabled != 0);
    launch_average_luminance(st, loglum, (int)width, (int)height, lum, out_avg);
    HIP_TRY(hipGetLastError());
    return RDR_OK;
}

}  // extern "C"
