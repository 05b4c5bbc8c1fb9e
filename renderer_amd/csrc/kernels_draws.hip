// Draw lists built on the GPU: which primitives a view draws, and in what order.
//
// Reference: drawMeshes / shouldCullMesh (Renderer/VulkanRenderer.cpp:929-999) and the per-face / per-cascade loops
// that call it for every shadow view of every frame (Renderer/VulkanRendererShadow.cpp:305, 348, 424): each view
// sorts the primitives by the distance of their barycentre to the eye, tests each bounding box against the six planes
// of viewproj * model and issues one draw per survivor. On one host thread that is views x primitives matrix products
// per frame, which overtakes the GPU's frame time at a few thousand primitives. Here one thread takes one
// (view, primitive) candidate:
//   k_draw_cull    the frustum test, in the host's float operations and their order (same decisions, bit for bit),
//                  and the sort key of the candidate where the view's order matters
//   (stable radix sort of those keys, kernels_sort.hip)
//   k_draw_gather  keep flag / triangles / cull chunks of every slot in drawing order   -> three exclusive scans
//   k_draw_write   the survivors' DDraw records and the per-draw offset tables set-up reads, compacted in order
// Only views with a visibility target need an order (ids follow it, and the earlier draw keeps a depth tie); a
// depth-only view ends with the minimum depth per sample whatever the order, so its candidates stay in index order.
#include "device_common.h"
#include "kernels.h"

namespace {

// viewproj * model as include/rdr_math.h mat_mul forms it: column c, row r = ((a0r b0 + a1r b1) + a2r b2) + a3r b3
RDR_DEV void mat_mul_cm(const float* a, const float* b, float* r) {
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const float b0 = b[c * 4], b1 = b[c * 4 + 1], b2 = b[c * 4 + 2], b3 = b[c * 4 + 3];
#pragma unroll
        for (int row = 0; row < 4; row++) {
            float s = a[row] * b0 + a[4 + row] * b1;
            s = s + a[8 + row] * b2;
            s = s + a[12 + row] * b3;
            r[c * 4 + row] = s;
        }
    }
}

// shouldCullMesh (VulkanRenderer.cpp:929-943) with the planes of Camera.cpp:245-280: the box corner farthest along a
// plane's normal lies behind it
RDR_DEV bool frustum_culls(const float* M, const DPrimCull& p) {
#define M_AT(c, r) M[(c) * 4 + (r)]
    float pl[6][4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        pl[0][k] = M_AT(k, 3) + M_AT(k, 0); pl[1][k] = M_AT(k, 3) - M_AT(k, 0);
        pl[2][k] = M_AT(k, 3) + M_AT(k, 1); pl[3][k] = M_AT(k, 3) - M_AT(k, 1);
        pl[4][k] = M_AT(k, 3) - M_AT(k, 2); pl[5][k] = M_AT(k, 2);
    }
#undef M_AT
    bool culled = false;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        const float nx = pl[i][0] >= 0.0f ? p.bb_max[0] : p.bb_min[0];
        const float ny = pl[i][1] >= 0.0f ? p.bb_max[1] : p.bb_min[1];
        const float nz = pl[i][2] >= 0.0f ? p.bb_max[2] : p.bb_min[2];
        if (((nx * pl[i][0] + ny * pl[i][1]) + nz * pl[i][2]) + pl[i][3] < 0.0f) culled = true;
    }
    return culled;
}

RDR_DEV uint32_t candidate(const DDrawList& l, const uint32_t* __restrict__ cand, uint32_t j) {
    return l.cand_first == RDR_CAND_ALL ? j : cand[l.cand_first + j];
}

// grid.y = list, one thread per candidate
__global__ void __launch_bounds__(256) k_draw_cull(const DDrawList* __restrict__ lists, const DPrimCull* __restrict__ prims,
                                                   const float* __restrict__ models, const uint32_t* __restrict__ cand,
                                                   uint32_t* __restrict__ keep, uint32_t* __restrict__ key, uint32_t* __restrict__ val) {
    const DDrawList& l = lists[blockIdx.y];
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j >= l.cand_count) return;
    const DPrimCull p = prims[candidate(l, cand, j)];
    float mvp[16];
    mat_mul_cm(l.viewproj, models + (size_t)p.mesh * 16, mvp);
    const uint32_t slot = l.slot_base + j;
    keep[slot] = frustum_culls(mvp, p) ? 0u : 1u;
    if (l.sort) {
        // distance(barycentre, eye) as include/rdr_math.h evaluates it; a non-negative float orders like its bits
        const float dx = l.eye[0] - p.barycenter[0], dy = l.eye[1] - p.barycenter[1], dz = l.eye[2] - p.barycenter[2];
        const uint32_t bits = __float_as_uint(sqrtf((dx * dx + dy * dy) + dz * dz));
        key[slot] = l.sort == 1u ? bits : ~bits;
        val[slot] = j;
    }
}

// slot i of a list in drawing order: candidate order[i] (the sorted candidates) or i itself
__global__ void __launch_bounds__(256) k_draw_gather(const DDrawList* __restrict__ lists, const DPrimCull* __restrict__ prims,
                                                     const uint32_t* __restrict__ cand, const uint32_t* __restrict__ keep,
                                                     const uint32_t* __restrict__ order, uint32_t* __restrict__ flag,
                                                     uint32_t* __restrict__ tris, uint32_t* __restrict__ chunks) {
    const DDrawList& l = lists[blockIdx.y];
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= l.cand_count) return;
    const uint32_t j = l.sort ? order[l.slot_base + i] : i;
    const uint32_t k = keep[l.slot_base + j];
    const uint32_t pi = candidate(l, cand, j);
    flag[l.slot_base + i] = k;
    tris[l.slot_base + i] = k ? prims[pi].tri_count : 0u;
    chunks[l.slot_base + i] = k ? prims[pi].n_chunks : 0u;
}

// scan_* = exclusive scans of flag / tris / chunks over the batch's slots, totals = their three sums
__global__ void __launch_bounds__(256) k_draw_write(const DDrawList* __restrict__ lists, uint32_t n_lists, uint32_t n_slots,
                                                    const DPrimCull* __restrict__ prims, const float* __restrict__ models,
                                                    const uint32_t* __restrict__ cand, const uint32_t* __restrict__ order,
                                                    const uint32_t* __restrict__ flag, const uint32_t* __restrict__ scan_flag,
                                                    const uint32_t* __restrict__ scan_tris, const uint32_t* __restrict__ scan_chunks,
                                                    const uint32_t* __restrict__ totals, DDraw* __restrict__ draws,
                                                    uint32_t* __restrict__ draw_first, uint32_t* __restrict__ first_tri,
                                                    uint32_t* __restrict__ chunk_first, DDrawListInfo* __restrict__ info, DView* views) {
    const uint32_t li = blockIdx.y;
    const DDrawList& l = lists[li];
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i == 0u) {
        // what the host reads back about this list; a list without candidates owns no slot
        const uint32_t end = l.slot_base + l.cand_count;
        const uint32_t d0 = l.slot_base < n_slots ? scan_flag[l.slot_base] : totals[0], d1 = end < n_slots ? scan_flag[end] : totals[0];
        const uint32_t t0 = l.slot_base < n_slots ? scan_tris[l.slot_base] : totals[1], t1 = end < n_slots ? scan_tris[end] : totals[1];
        DDrawListInfo o;
        o.first_draw = d0; o.n_draws = d1 - d0; o.first_instance = t0; o.n_tris = t1 - t0;
        info[li] = o;
        if (l.drop_empty && d1 == d0) views[l.view].load = 3u;
        if (li == 0u) { draw_first[totals[0]] = totals[1]; chunk_first[totals[0]] = totals[2]; }
    }
    if (i >= l.cand_count) return;
    const uint32_t slot = l.slot_base + i;
    if (!flag[slot]) return;
    const uint32_t j = l.sort ? order[slot] : i;
    const uint32_t pi = candidate(l, cand, j);
    const float* model = models + (size_t)prims[pi].mesh * 16;
    const uint32_t d = scan_flag[slot];
    DDraw o;
    mat_mul_cm(l.viewproj, model, o.mvp);
#pragma unroll
    for (int k = 0; k < 16; k++) o.model[k] = model[k];
    o.view = l.view; o.prim = pi;
    o.first_instance = scan_tris[slot];
    o.first_tri_id = scan_tris[slot] - scan_tris[l.slot_base];
    draws[d] = o;
    draw_first[d] = o.first_instance;
    first_tri[d] = o.first_tri_id;
    chunk_first[d] = scan_chunks[slot];
}

}  // namespace

size_t draw_list_scratch_words(uint32_t n_slots, uint32_t longest_sorted) {
    // keep | key | val | flag | tris | chunks | three scans | second key / value buffers of the sort | totals
    return (size_t)n_slots * 9 + (size_t)longest_sorted * 2 + 16;
}

// `lists`: host copy of the table at `d_lists` (list l owns slots [slot_base, slot_base + cand_count)). Leaves the
// compacted draws and offset tables in draws / draw_first / first_tri / chunk_first, the per-list summary in `info`
// and {draws, triangle instances, cull chunks} of the batch in `totals` (device, 3 words).
void launch_draw_lists(hipStream_t s, const DDrawList* lists, uint32_t n_lists, const DDrawList* d_lists, uint32_t n_slots,
                       const DPrimCull* prims, const float* models, const uint32_t* cand, uint32_t* scratch, uint32_t* sort_hist,
                       uint32_t* scan_scratch, DDraw* draws, uint32_t* draw_first, uint32_t* first_tri, uint32_t* chunk_first,
                       DDrawListInfo* info, DView* views, uint32_t* totals) {
    uint32_t longest = 0, longest_sorted = 0;
    for (uint32_t l = 0; l < n_lists; l++) {
        longest = max(longest, lists[l].cand_count);
        if (lists[l].sort) longest_sorted = max(longest_sorted, lists[l].cand_count);
    }
    uint32_t* keep = scratch;
    uint32_t* key = keep + n_slots;
    uint32_t* val = key + n_slots;
    uint32_t* flag = val + n_slots;
    uint32_t* tris = flag + n_slots;
    uint32_t* chunks = tris + n_slots;
    uint32_t* scan_flag = chunks + n_slots;
    uint32_t* scan_tris = scan_flag + n_slots;
    uint32_t* scan_chunks = scan_tris + n_slots;
    uint32_t* key2 = scan_chunks + n_slots;
    uint32_t* val2 = key2 + longest_sorted;
    const dim3 grid((max(longest, 1u) + 255u) / 256u, n_lists);
    if (n_slots) {
        hipLaunchKernelGGL(k_draw_cull, grid, dim3(256), 0, s, d_lists, prims, models, cand, keep, key, val);
        for (uint32_t l = 0; l < n_lists; l++) {
            if (!lists[l].sort || lists[l].cand_count < 2u) continue;
            uint32_t* keys[2] = {key + lists[l].slot_base, key2};
            uint32_t* vals[2] = {val + lists[l].slot_base, val2};
            const int cur = launch_radix_sort(s, keys, vals, lists[l].cand_count, 32u, sort_hist, scan_scratch, totals + 8);
            if (cur) (void)hipMemcpyAsync(vals[0], vals[1], (size_t)lists[l].cand_count * 4, hipMemcpyDeviceToDevice, s);
        }
        hipLaunchKernelGGL(k_draw_gather, grid, dim3(256), 0, s, d_lists, prims, cand, keep, val, flag, tris, chunks);
    }
    launch_exclusive_scan(s, flag, scan_flag, n_slots, scan_scratch, totals);
    launch_exclusive_scan(s, tris, scan_tris, n_slots, scan_scratch, totals + 1);
    launch_exclusive_scan(s, chunks, scan_chunks, n_slots, scan_scratch, totals + 2);
    hipLaunchKernelGGL(k_draw_write, grid, dim3(256), 0, s, d_lists, n_lists, n_slots, prims, models, cand, val, flag, scan_flag, scan_tris,
                       scan_chunks, totals, draws, draw_first, first_tri, chunk_first, info, views);
}
