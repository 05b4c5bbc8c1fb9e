// Host-callable launchers of the HIP kernels. Every launcher only enqueues
// work on the given stream.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "rdr_device.h"

struct IrradianceTables {  // sin/cos of the 24 azimuth and 6 polar angles of envbakediffuse.frag:18-26
    float cos_phi[24], sin_phi[24], sin_theta[6], cos_theta[6];
};

// kernels_post.hip
void launch_bloom_down(hipStream_t s, const void* src, bool src_is_hdr, int sw, int sh, uint32_t* dst, int dw, int dh, int dst_pitch);
void launch_bloom_up(hipStream_t s, const uint32_t* src, int sw, int sh, uint32_t* dst, int dw, int dh, int dst_pitch);
// the whole chain before the tonemap kernel (five down-samples, up-samples down to mip 1); level k is max(1, W >> k) x max(1, H >> k)
// returns true when the accumulated mips 3 and 4 were written to fin3 / fin4 instead of in place
bool launch_bloom_chain(hipStream_t s, const void* hdr, uint32_t* const mip[6], const int w[6], const int h[6], int W, int H, uint32_t* fin3,
                        uint32_t* fin4);
// last up-sample (mip 1 -> mip 0) fused with the tonemap; sw/sh = 0 when the image has no mip 1. `out`: RGBA8, 4 bytes
// per pixel, or with hdr16 RGBA16F, 8 bytes per pixel
void launch_tonemap(hipStream_t s, const void* hdr, const uint32_t* mip1, int sw, int sh, uint32_t* bloom0, uint32_t* out, float* loglum,
                    int W, int H, float scale, float inv_gamma, float white, float log2_6, bool keep, bool hdr16);
void launch_average_luminance(hipStream_t s, const float* loglum, int W, int H, float* l16, float* l1);

// kernels_sort.hip
size_t scan_scratch_words(uint32_t n);
void launch_exclusive_scan(hipStream_t s, const uint32_t* in, uint32_t* out, uint32_t n, uint32_t* scratch, uint32_t* d_total);
// the same in one launch; `status` holds scan_chained_words(n) words (8-byte aligned) that are zero before and after
size_t scan_chained_words(uint32_t n);
void launch_exclusive_scan_chained(hipStream_t s, const uint32_t* in, uint32_t* out, uint32_t n, uint32_t* status, uint32_t* d_total);
uint32_t sort_chunks(uint32_t n);
size_t sort_hist_words(uint32_t n);
int launch_radix_sort(hipStream_t s, uint32_t* keys[2], uint32_t* vals[2], uint32_t n, uint32_t key_bits, uint32_t* hist,
                      uint32_t* scan_scratch, uint32_t* d_dummy_total);
// the same order from one launch per digit (and one in front that counts the digits and clears `clear` words from
// `to_clear` on): see kernels_sort.hip. n < 2^30. `ghist_zeroed`: the first 4 x 512 words of `work` are already zero.
size_t sort_onesweep_work_bytes(uint32_t n);
#define RDR_SORT_ONESWEEP_HIST_WORDS (4u * 512u)
int launch_radix_sort_onesweep(hipStream_t s, uint32_t* keys[2], uint32_t* vals[2], uint32_t n, uint32_t key_bits, void* work, uint32_t* epoch,
                               uint32_t* to_clear, uint32_t clear, uint32_t* error, bool ghist_zeroed);

// kernels_draws.hip
size_t draw_list_scratch_words(uint32_t n_slots, uint32_t longest_sorted);
void launch_draw_lists(hipStream_t s, const DDrawList* lists, uint32_t n_lists, const DDrawList* d_lists, uint32_t n_slots,
                       const DPrimCull* prims, const float* models, const uint32_t* cand, uint32_t* scratch, uint32_t* sort_hist,
                       uint32_t* scan_scratch, DDraw* draws, uint32_t* draw_first, uint32_t* first_tri, uint32_t* chunk_first,
                       DDrawListInfo* info, DView* views, uint32_t* totals);

// kernels_raster.hip
void launch_setup(hipStream_t s, const DView* views, const DDraw* draws, const uint32_t* draw_first, uint32_t n_draws, uint32_t n_inst,
                  const DPrim* prims, const void* pos, const uint32_t* idx, DTriRec* recs, uint32_t* pair_count, uint32_t* n_binned,
                  const uint32_t* chunk_first, uint32_t n_chunks, const void* chunk_bb, uint8_t* chunk_flags);
// `zero` / `n_zero` (at most 4096 words): cleared by the first block on the way, for the kernel behind this one
void launch_emit(hipStream_t s, const DView* views, const DTriRec* recs, const uint32_t* pair_offset, uint32_t n_inst, uint32_t n_pairs,
                 uint32_t* keys, uint32_t* vals, uint32_t* zero = nullptr, uint32_t n_zero = 0);
void launch_ranges(hipStream_t s, const uint32_t* keys, uint32_t n_pairs, uint32_t* tile_ranges, uint32_t n_tiles, bool cleared = false);
// tiles [first_tile, end_tile) hold the views of one sample count, the first of which is view first_view
struct RasterRange { uint32_t first_tile, end_tile, first_view; };
// the views of one rasteriser launch, passed by value in the kernel arguments
#define RDR_RASTER_MAX_VIEWS 32
struct RasterViews {
    uint32_t n, first_tile, end_tile;
    // Which tiles of the launch store their depth: 0 all; 1 only those with a blended triangle (the blended pass reads
    // and writes their depth: what the frame itself needs); 2 only those without (what rdr_read_depth asks for later:
    // a second pass over the same lists, for the tiles the first left out). `blend_base`: first entry of the blended
    // view's tile lists in tile_start / tile_end, 0xFFFFFFFF when there is no such view.
    uint32_t depth_mode, blend_base, _pad[3];
    uint32_t base[RDR_RASTER_MAX_VIEWS];  // first tile of view k (0xFFFFFFFF past the last)
    DView v[RDR_RASTER_MAX_VIEWS];
};
// `classify`: for the 4x view of a frame, the fragment lists of the opaque pass, filled from the tiles' samples while
// they are still in LDS (then launch_frag_classify is not needed); null otherwise
struct FragLists;
void launch_raster(hipStream_t s, const DView* views, uint32_t n_views, const DView* d_views, const DTriRec* recs, const uint32_t* sorted_vals,
                   const uint32_t* tile_start, const uint32_t* tile_end, RasterRange r, int samples, const FragLists* classify = nullptr,
                   uint32_t depth_mode = 0u, uint32_t blend_base = 0xFFFFFFFFu);

// kernels_shade.hip
void launch_shade_setup(hipStream_t s, const DFrame* fr, const DDraw* draws, const uint32_t* first_tri, uint32_t n_draws, uint32_t n_tris,
                        const uint32_t* pair_offset, uint32_t n_after, const uint32_t* pair_total, DShadeTri* out);
// opaque pass: fragment lists (one per screen rectangle), see kernels_shade.hip
#define RDR_FRAG_SEGS 64u
#define RDR_FRAG_CTR_STRIDE 32u  // words between two list counters (128 bytes)

struct FragLists {
    uint32_t rect_w, rect_h;  // rectangle size in tiles
    uint32_t cap;             // slots per list
    uint32_t* count;          // fill count of list s at count[s * RDR_FRAG_CTR_STRIDE]; list s = die * 8 + grid row
    uint32_t* claim;          // chunks handed out of die d's lists at claim[d * RDR_FRAG_CTR_STRIDE]
    uint2* recs;              // [RDR_FRAG_SEGS][cap]: triangle id, x | y << 16
    uint2* cols;              // same layout: RGBA16F
    uint32_t* pixel_state;    // [tile][256]: first fragment of the pixel, relative to the tile's | slot of sample s << (16 + 4 s)
    uint32_t* tile_first;     // [tile]: position (in recs / cols) of the tile's first fragment
};

size_t shade_frag_slots(uint32_t W, uint32_t H, int samples);  // slots of recs / cols
uint32_t shade_frags_blocks_per_cu();  // blocks of k_shade_frags that fit a compute unit
FragLists make_frag_lists(uint32_t W, uint32_t H, int samples, void* frag_recs, void* frag_cols, uint32_t* pixel_state, uint32_t* tile_first,
                          uint32_t* list_count, uint32_t* list_claim);
void launch_frag_classify(hipStream_t s, int samples, const DFrame* fr, const uint32_t* vis, const FragLists& L);
// float_texels: some image of the context has an SFLOAT format (the build of the kernel that can sample those)
void launch_shade_frags(hipStream_t s, const DFrame* fr, const DPointLight* lights, const DShadeTri* tris, const FragLists& L, uint32_t blocks,
                        bool float_texels);
// the visibility image rebuilt from the lists (frames whose rasteriser built the lists do not store it)
void launch_vis_from_lists(hipStream_t s, int samples, const FragLists& L, uint32_t W, uint32_t H, uint32_t* vis);
void launch_frag_resolve(hipStream_t s, int samples, const DFrame* fr, const FragLists& L, void* hdr, void* samples_out, const uint32_t* blend_start,
                         const uint32_t* blend_end, uint32_t W, uint32_t H);
// blended pass: fragment list slots are handed out in runs of RDR_FRAG_RUN; `capacity` slots exist (see blend_capacity);
// *frag_used counts the slots handed out and must be zero when launch_raster_blend is called
#define RDR_FRAG_RUN 128u
#define RDR_BLEND_SHADE_BLOCKS 2048u
void launch_raster_blend(hipStream_t s, int samples, const DFrame* fr, const DPointLight* lights, const DView* view, const DShadeTri* tris, const DTriRec* recs,
                         const uint32_t* sorted_vals, const uint32_t* tile_start, const uint32_t* tile_end, uint32_t tile0, uint32_t n_tiles, void* color,
                         void* hdr, void* frags, void* frag_col, uint32_t* frag_head, uint32_t* frag_used, uint32_t capacity, bool float_texels);
// dry run of the collect pass over all tiles: tile_slots[tile] = list slots the tile would take (0 for tiles without blended triangles; zero the array first)
void launch_blend_count(hipStream_t s, int samples, const DView* view, const DTriRec* recs, const uint32_t* sorted_vals, const uint32_t* tile_start,
                        const uint32_t* tile_end, uint32_t n_tiles, uint32_t* tile_slots);

// kernels_ibl.hip
// per-frame skinning of the vertex arena (kernels_skin.hip): rest pose + joints/weights + palette -> pos/attr
void launch_skin(hipStream_t s, const void* rest_pos, const void* rest_attr, const void* joints, const void* weights, const void* dst_pal,
                 const float* palette, void* pos, void* attr, uint32_t n);
void launch_equirect_to_cube(hipStream_t s, const void* pano_rgba32f, int w, int h, uint32_t* dst, int res);
void launch_brdf_lut(hipStream_t s, uint32_t* dst, int size, int samples, const float* trig);
void launch_env_pack(hipStream_t s, const void* face_rgba32f, uint32_t* dst, uint32_t n);
// point-light shadow cubes stored as n_cubes x 6 faces of (res + 2)^2: fills every face's one-texel border in place
void launch_shadow_cube_border(hipStream_t s, float* maps, int res, uint32_t n_cubes);
void launch_fill_f32(hipStream_t s, float* dst, float v, size_t n);
// one level of a cube (6 x res^2 words) -> the same level with a one-texel border per face (6 x (res + 2)^2 words)
void launch_cube_border(hipStream_t s, const uint32_t* src, uint32_t* dst, int res);
// footprint-major copies for the shading kernels: one bordered cube level -> 6 x (res + 1)^2 entries of four texels;
// the w x h BRDF table -> w x h entries of four texels (REPEAT)
void launch_cube_quads(hipStream_t s, const uint32_t* bordered, void* dst, int res);
void launch_lut_quads(hipStream_t s, const uint32_t* src, void* dst, int w, int h);
void launch_bake_irradiance(hipStream_t s, const DCube& env, uint32_t* dst, int res, const IrradianceTables& tb);
void launch_bake_prefilter(hipStream_t s, const DCube& env, uint32_t* dst, int res, float roughness, const float* trig);
// the same for an image of float4 texels; to_half: every level rounded to binary16
void launch_mip_chain_f32(hipStream_t s, float* base, const uint32_t* level_offset, uint32_t w, uint32_t h, uint32_t levels, bool to_half);
// levels 1 .. levels - 1 of an image whose level 0 (w x h texels) is at base + level_offset[0], each blitted from the one before
void launch_mip_chain(hipStream_t s, uint32_t* base, const uint32_t* level_offset, uint32_t w, uint32_t h, uint32_t levels, int srgb,
                      const float* srgb_decode, const float* unorm_decode, const float* thr);
