// Plain-data structures shared by the host side of the library and its HIP
// kernels: everything a kernel reads from HBM that is not a pixel or a vertex.
#pragma once
#include <stdint.h>

#define RDR_TILE 16             // raster tile edge in pixels (one 256-thread block, 4 waves of 8x8)
#define RDR_MAX_POLY 9          // a triangle cut by 6 planes has at most 9 vertices
#define RDR_MAX_CASCADES 4
#define RDR_MAX_TEX_LEVELS 16
#define RDR_PAIR_INVALID 0xFFFFFFFFu
#define RDR_BINNED_SLOTS 64u     // counters (128 bytes apart) the binned-triangle statistic is spread over
#define RDR_CHUNK_TRIS 64u       // consecutive triangles of a primitive that share one bounding box for GPU-side culling
#define RDR_NO_CHUNKS 0xFFFFFFFFu  // DPrim::first_chunk of a primitive without chunk boxes (skinned: its vertices move)

// Material block of one primitive (PBRInfo + AlphaInfo of the reference,
// read by pbr.frag:21-37) plus its five texture slots.
struct DMaterial {
    float base_color[4];
    float emissive[4];
    float metallic, roughness, ao;
    float alpha_cutoff;
    int32_t alpha_mode;
    uint32_t tex[5];  // albedo, normal, metalRough, ao, emissive (indices into DTexture table)
    uint32_t _pad[2];
};

// One mesh primitive in the geometry arena.
struct DPrim {
    uint32_t first_vertex;  // into pos[] / attr[]
    uint32_t first_index;   // into idx[]
    uint32_t tri_count;
    uint32_t first_chunk;   // into the chunk boxes (two float4 each: min, max in model space), or RDR_NO_CHUNKS
    DMaterial mat;
};

// Per-vertex shading attributes, the reference's Vertex (Vertex.h:8-12) without the position.
struct DVertexAttr {
    float nx, ny, nz;
    float tx, ty, tz;
    float bx, by, bz;
    float u, v;
    float _pad;
};  // 48 bytes

// One texture = image (all mip levels, level-major; RGBA8, or for the float formats of Image.h one float4 per texel) + sampler state.
struct DTexture {
    const uint8_t* base;
    uint32_t w, h, levels;
    uint32_t flags;  // bit0 sRGB, bits 4..5 wrap_u, bits 8..9 wrap_v, bits 12..14 NEAREST for mag / min / mipmap mode, bit 16 float texels
    uint32_t level_offset[RDR_MAX_TEX_LEVELS];  // in texels
};

// One raster view (a shadow-map face, a cascade, or the main camera).
struct DView {
    float* depth;        // [samples][h][w]
    const float* init;   // load == 1: where the depth the view starts from is read (same layout); may be `depth` itself
    uint32_t* vis;       // same layout, or null for depth-only views
    uint32_t w, h;
    uint32_t samples;    // 1 or 4
    uint32_t cull;       // 0 none, 1 front, 2 back
    uint32_t tiles_x, tiles_y;
    uint32_t tile_base;  // first global tile id of this view
    uint32_t load;       // 0: start from depth 1.0, 1: start from what `init` holds, 2: binned only (the blended pass rasterises it),
                         // 3: skipped (set on the GPU for a view whose draw list came out empty, DDrawList::drop_empty)
    uint32_t pitch;      // texels from one row of depth / init / vis to the next (>= w: shadow targets carry a border the shaders filter across)
};

// One draw = one primitive seen from one view.
struct DDraw {
    float mvp[16];
    float model[16];
    uint32_t view;
    uint32_t prim;
    uint32_t first_instance;  // global index of its first triangle instance in this batch
    uint32_t first_tri_id;    // running triangle index inside its view (the visibility id)
};

// What the draw-list kernels need of a primitive: the box and barycentre drawMeshes culls and orders by
// (Renderer/VulkanRenderer.cpp:929-999), its mesh (for the model matrix) and how much it adds to the batch.
struct DPrimCull {
    float bb_min[3], bb_max[3], barycenter[3];
    uint32_t mesh;
    uint32_t tri_count;
    uint32_t n_chunks;  // ceil(tri_count / RDR_CHUNK_TRIS)
};  // 48 bytes

#define RDR_CAND_ALL 0xFFFFFFFFu  // DDrawList::cand_first: the candidates are all primitives, in index order

// One view's candidate primitives: which of them survive the frustum test, and in what order they are drawn, is
// decided on the GPU. Candidate j of list l owns slot slot_base + j of the batch.
struct DDrawList {
    float viewproj[16];
    float eye[3];
    uint32_t sort;        // 0: candidate order, 1: nearest barycentre first, 2: farthest first (both stable)
    uint32_t cand_first;  // first entry in the candidate array, or RDR_CAND_ALL
    uint32_t cand_count;
    uint32_t slot_base;
    uint32_t view;        // the view (index into the batch's DView table) the surviving draws belong to
    uint32_t drop_empty;  // 1: a view without a surviving draw is not rendered at all (its target keeps what it holds)
    uint32_t kind;        // host bookkeeping: which candidate list this is
    uint32_t _pad[2];
};  // 112 bytes

// per list, written by the draw-list kernels and read back by the host
struct DDrawListInfo { uint32_t first_draw, n_draws, first_instance, n_tris; };

// Output of triangle set-up: the snapped clip polygon and its depth plane.
struct DTriRec {
    int32_t X[RDR_MAX_POLY];  // 1/256 px, oriented so the interior has positive edge functions
    int32_t Y[RDR_MAX_POLY];
    uint32_t n_view;          // n (low 8 bits, 0 = dropped) | view << 8
    uint32_t id;              // visibility id
    int16_t px0, py0, px1, py1;  // pixel bounding box, inclusive, clipped to the view
    uint32_t _pad;
    double dzdX, dzdY, z00;   // z(P) = dzdX*Px + dzdY*Py + z00, P in 1/256 px
    uint32_t _pad2[2];
};  // 128 bytes

// Compact texture descriptor inside a DShadeTri: everything a fetch needs without another table look-up.
#define RDR_TEXD_SRGB 1u        // flags bit 0
#define RDR_TEXD_POW2_SQUARE 2u  // flags bit 1: w == h == 2^n with a complete chain, level offsets are arithmetic
#define RDR_TEXD_NEAREST_SHIFT 6u  // flags bits 6..8: the Sampler's magFilter / minFilter / mipmapFilter is NEAREST
#define RDR_TEXD_FLOAT 512u       // flags bit 9: texels are float4 (the SFLOAT formats of Image.h), not RGBA8
struct DTexDesc {
    const uint8_t* base;  // texels, all levels: RGBA8, or float4 with RDR_TEXD_FLOAT
    uint32_t wh;          // w | h << 16
    uint32_t info;        // levels (5 bits) | flags << 5 (srgb, pow2-square, wrap_u << 2, wrap_v << 4, nearest filters << 6, float << 9) | texture index << 16
};  // 16 bytes

// Everything the fragments of one triangle share: the vertex-stage outputs
// (basic.vert), the material block (pbr.frag:21-37) and its five texture
// descriptors. Written once per frame by the shading set-up kernel; every fragment
// reads the record of its triangle in place (neighbouring fragments share it through the caches).
struct alignas(16) DShadeTri {
    double A[3], B[3], C[3];  // homogeneous barycentric planes: l_k(x,y) = A_k x + B_k y + C_k in NDC
    float wp[9];              // world position, component-major: x0 x1 x2 y0 y1 y2 z0 z1 z2
    float nrm[9], tan[9], bit[9];
    float uv[6];              // u0 u1 u2 v0 v1 v2
    float base_color[4];
    float emissive[3];
    float alpha_cutoff;
    float metallic, roughness, ao;
    int32_t alpha_mode;
    DTexDesc tex[5];          // albedo, normal, metalRough, ao, emissive
    uint32_t prim;
    uint32_t _pad[3];
};  // 384 bytes = 24 x 16
#define RDR_SHADE_TRI_VEC4 24

// Point lights 2k and 2k + 1 side by side, component by component: the shading kernels read the record with scalar
// loads and work on both lights of a pair at once (two-component vector instructions take a component pair straight
// from two neighbouring scalar registers). The second light of the last pair of an odd number of lights has zero
// intensity (which the light cut-off of pbr.frag:133 drops), flags 0 and the position of the first.
struct DPointLight {
    float px[2], py[2], pz[2];
    float ix[2], iy[2], iz[2];
    uint32_t flags[2];
    uint32_t slot[2];
};  // 64 bytes

struct DCube {
    const uint32_t* texels;  // B10G11R11 packed, level-major, face-major inside a level
    uint32_t res, levels;
    uint32_t level_offset[12];
};

// Everything the shading kernels need for one frame.
struct DFrame {
    uint32_t W, H, samples, n_lights;
    float inv2W, inv2H;
    float cam[3];
    float inv_viewproj[16];
    float csm_viewproj[RDR_MAX_CASCADES][16];
    uint32_t n_cascades, dir_res, point_res, lights_set;
    float dir_L[3];          // normalised direction towards... exactly directionalLight.direction normalised
    float dir_I[3];
    const float* dir_shadow;    // [cascade][res + 4][res + 4]: two texels of border holding 1.0 (clamp-to-border, white)
    const float* point_shadow;  // [slot*6+face][res + 2][res + 2]: one texel of border holding the texels across the cube edges
    const DPointLight* lights;
    DCube env;                      // bordered faces, (res + 2)^2 words each (the sky's look-ups are coherent)
    DCube irradiance, prefiltered;  // footprint-major: `texels` points at 16-byte entries of four texels, level_offset counts entries
    const uint8_t* lut;             // footprint-major RGBA8 table: lut_w x lut_h entries of four texels
    uint32_t lut_w, lut_h;
    const DTexture* textures;
    const DPrim* prims;
    const float4* pos;
    const DVertexAttr* attr;
    const uint32_t* idx;
    const float* srgb_decode;   // 256 floats
    const float* unorm_decode;  // 256 floats: i / 255
    uint32_t tex_white, tex_normal;
};
