// Software rasteriser: replaces the fixed-function half of the reference's
// depth-tested triangle-list draws (pipeline state:
// Renderer/VulkanRenderer.cpp:836-866 main pass,
// Renderer/VulkanRendererShadow.cpp:72-118 shadow passes; vertex stage:
// Renderer/shaders/shadowmap.vert:10-12, basic.vert:37).
//
//   k_chunk_cull  one thread per chunk of RDR_CHUNK_TRIS consecutive triangles of a
//             draw: bounding box of the chunk against the view's clip planes
//             -> one flag per chunk; set-up skips the triangles of culled chunks
//   k_setup   one thread per triangle instance: transform, clip to a polygon
//             (near, far, +-16384 px guard band), snap to 1/256 px, facing +
//             cull, pixel/tile bounding box, depth plane   -> DTriRec
//   k_emit    one thread per (triangle, tile) candidate: exact tile test,
//             key = global tile id                        -> pairs
//   (stable radix sort of the pairs by tile: kernels_sort.hip)
//   k_ranges  list boundaries per tile
//   k_raster  one block per 16x16 tile: records staged through LDS 64 at a
//             time, the (triangle, pixel) items of their pixel boxes spread
//             over the threads; edge functions are exact integers (32-bit
//             relative to the tile where that is provably the same, 64-bit
//             otherwise), depth is a double-precision plane rounded to float,
//             the LESS test is an LDS atomic min on (depth, triangle id).
//
// Bit-exactness with the CPU checker rests on three things: the fma chains
// are written out, the edge functions are integer, and double arithmetic is
// IEEE on both sides. This file must be compiled with -ffp-contract=off.
#include <cstddef>

#include "device_common.h"
#include "kernels.h"
#include "raster_common.h"

namespace {

struct ClipV { float x, y, z, w; };

RDR_DEV float plane_dist(const ClipV& v, int plane, float kx, float ky) {
    switch (plane) {
        case 0: return v.z;
        case 1: return v.w - v.z;
        case 2: return fmaf(kx, v.w, v.x);
        case 3: return fmaf(kx, v.w, -v.x);
        case 4: return fmaf(ky, v.w, v.y);
        default: return fmaf(ky, v.w, -v.y);
    }
}

RDR_DEV uint32_t find_draw(const uint32_t* __restrict__ draw_first, uint32_t n_draws, uint32_t inst) {
    uint32_t lo = 0, hi = n_draws;  // largest d with draw_first[d] <= inst
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (draw_first[mid] <= inst) lo = mid; else hi = mid;
    }
    return lo;
}

// Second half of triangle set-up, from the snapped polygon on: facing and cull, pixel box, depth plane, the
// record. FIXED = 3 unrolls everything for an unclipped triangle; FIXED = 0 takes n at run time. Returns the number
// of tiles the pixel box touches, 0 if the polygon is dropped.
static_assert(offsetof(DTriRec, Y) == 36 && offsetof(DTriRec, n_view) == 72 && offsetof(DTriRec, id) == 76 && offsetof(DTriRec, px0) == 80 &&
              offsetof(DTriRec, dzdX) == 96 && offsetof(DTriRec, z00) == 112, "finish_record writes the record as 16-byte pieces");

template <int FIXED>
RDR_DEV uint32_t finish_record(const int32_t* X, const int32_t* Y, int n_in, const ClipV (&cv)[3], const DView& vw, const DDraw& d, uint32_t tri,
                               DTriRec& rec, uint32_t& box_area) {
    const int n = FIXED ? FIXED : n_in;
    const int W = (int)vw.w, H = (int)vw.h, S = (int)vw.samples;
    // ---- facing: Vulkan area a = -1/2 sum(x_i y_i+1 - x_i+1 y_i), CCW front => front iff sum < 0
    long long area2 = 0;
#pragma unroll
    for (int i = 0; i < n; i++) {
        const int j = (i + 1 == n) ? 0 : i + 1;
        area2 += (long long)X[i] * Y[j] - (long long)X[j] * Y[i];
    }
    if (area2 == 0) return 0;
    const bool front = area2 < 0;
    if ((vw.cull == 2u && !front) || (vw.cull == 1u && front)) return 0;

    // ---- pixel bounding box
    int32_t minX = X[0], maxX = X[0], minY = Y[0], maxY = Y[0];
#pragma unroll
    for (int i = 1; i < n; i++) {
        minX = min(minX, X[i]); maxX = max(maxX, X[i]);
        minY = min(minY, Y[i]); maxY = max(maxY, Y[i]);
    }
    int ix0, ix1, iy0, iy1;
    if (S == 1) { ix0 = (minX - 128 + 255) >> 8; ix1 = (maxX - 128) >> 8; iy0 = (minY - 128 + 255) >> 8; iy1 = (maxY - 128) >> 8; }
    else { ix0 = minX >> 8; ix1 = maxX >> 8; iy0 = minY >> 8; iy1 = maxY >> 8; }
    ix0 = max(ix0, 0); iy0 = max(iy0, 0);
    ix1 = min(ix1, W - 1); iy1 = min(iy1, H - 1);
    if (ix0 > ix1 || iy0 > iy1) return 0;

    // ---- a triangle whose box is a few pixels may cover no sample at all (sub-pixel triangles of dense meshes,
    // slivers of surfaces seen edge-on): the rasteriser's own test, on the at most 16 pixels of the box, settles it
    // here, before the triangle costs a record, tile pairs and a place in the tile lists.
    // With the triangle itself at most 32 pixels across every term below is under 2^28, so 32-bit integers hold them.
    if (FIXED == 3 && (ix1 - ix0 + 1) * (iy1 - iy0 + 1) <= 16 && maxX - minX <= 8192 && maxY - minY <= 8192) {
        const bool flip = area2 < 0;
        int E0[3], ex[3], ey[3], es[3][4];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            // edge i of the polygon oriented so that its interior has positive edge functions
            const int a = flip ? 2 - i : i, b2 = flip ? (2 - i + 2) % 3 : (i + 1) % 3;
            const int dX = X[b2] - X[a], dY = Y[b2] - Y[a];
            const int bias = ((dY < 0) || (dY == 0 && dX > 0)) ? 0 : 1;
            E0[i] = dX * (256 * iy0 - Y[a]) - dY * (256 * ix0 - X[a]) - bias;
            ex[i] = -dY * 256; ey[i] = dX * 256;
            const int sxo[4] = {96, 224, 32, 160}, syo[4] = {32, 96, 160, 224};
#pragma unroll
            for (int sm = 0; sm < 4; sm++) es[i][sm] = S == 1 ? dX * 128 - dY * 128 : dX * syo[sm] - dY * sxo[sm];
        }
        bool covered = false;
        const int ns = S == 1 ? 1 : 4;
        for (int py = 0; py <= iy1 - iy0 && !covered; py++)
            for (int px = 0; px <= ix1 - ix0 && !covered; px++) {
                const int e0 = E0[0] + ex[0] * px + ey[0] * py, e1 = E0[1] + ex[1] * px + ey[1] * py, e2 = E0[2] + ex[2] * px + ey[2] * py;
                for (int sm = 0; sm < ns; sm++)
                    if (((e0 + es[0][sm]) | (e1 + es[1][sm]) | (e2 + es[2][sm])) >= 0) covered = true;
            }
        if (!covered) return 0;
    }

    // ---- depth plane a x + b y + c w = z through the unclipped vertices
    const double x0 = cv[0].x, y0 = cv[0].y, z0 = cv[0].z, w0 = cv[0].w;
    const double x1 = cv[1].x, y1 = cv[1].y, z1 = cv[1].z, w1 = cv[1].w;
    const double x2 = cv[2].x, y2 = cv[2].y, z2 = cv[2].z, w2 = cv[2].w;
    const double m_yw = y1 * w2 - y2 * w1, m_xw = x1 * w2 - x2 * w1, m_xy = x1 * y2 - x2 * y1;
    const double m_zw = z1 * w2 - z2 * w1, m_zy = z1 * y2 - z2 * y1, m_xz = x1 * z2 - x2 * z1;
    const double m_yz = y1 * z2 - y2 * z1;
    const double det = (x0 * m_yw - y0 * m_xw) + w0 * m_xy;
    if (det == 0.0 || !isfinite(det)) return 0;
    const double pa = ((z0 * m_yw - y0 * m_zw) + w0 * m_zy) / det;
    const double pb = ((x0 * m_zw - z0 * m_xw) + w0 * m_xz) / det;
    const double pc = ((x0 * m_yz - y0 * m_xz) + z0 * m_xy) / det;
    const double sx = 2.0 / (256.0 * (double)W), sy = 2.0 / (256.0 * (double)H);
    const double dzdX = pa * sx, dzdY = pb * sy, z00 = (pc - pa) - pb;
    if (!isfinite(dzdX) || !isfinite(dzdY) || !isfinite(z00)) return 0;

    // ---- write the record, oriented so that the interior has positive edge functions. Neighbouring lanes write
    // records 128 bytes apart, so every store instruction touches 64 cache lines: the record goes out as 16-byte
    // pieces (six for a triangle) rather than field by field.
    const uint32_t n_view = (uint32_t)n | (d.view << 8), id = d.first_tri_id + tri;
    const uint32_t box0 = (uint32_t)(ix0 & 0xFFFF) | ((uint32_t)iy0 << 16), box1 = (uint32_t)(ix1 & 0xFFFF) | ((uint32_t)iy1 << 16);
    uint4* q = (uint4*)&rec;
    if (FIXED == 3) {
        const bool flip = area2 < 0;
        const int32_t xa = flip ? X[2] : X[0], xc = flip ? X[0] : X[2], ya = flip ? Y[2] : Y[0], yc = flip ? Y[0] : Y[2];
        q[0] = make_uint4((uint32_t)xa, (uint32_t)X[1], (uint32_t)xc, 0u);             // X[0..3]
        q[1] = make_uint4(0u, 0u, 0u, 0u);                                             // (the whole 128-byte line is written: no partial-line write-back)
        q[2] = make_uint4(0u, (uint32_t)ya, (uint32_t)Y[1], (uint32_t)yc);             // X[8], Y[0..2]
        q[3] = make_uint4(0u, 0u, 0u, 0u);
        q[4] = make_uint4(0u, 0u, n_view, id);                                         // Y[7], Y[8], n_view, id
    } else {
        if (area2 < 0) { for (int i = 0; i < n; i++) { rec.X[i] = X[n - 1 - i]; rec.Y[i] = Y[n - 1 - i]; } }
        else { for (int i = 0; i < n; i++) { rec.X[i] = X[i]; rec.Y[i] = Y[i]; } }
        rec.n_view = n_view; rec.id = id;
    }
    q[5] = make_uint4(box0, box1, 0u, 0u);                                             // px0 py0 | px1 py1 | pad | alignment
    const unsigned long long ux = (unsigned long long)__double_as_longlong(dzdX), uy = (unsigned long long)__double_as_longlong(dzdY),
                             uz = (unsigned long long)__double_as_longlong(z00);
    q[6] = make_uint4((uint32_t)ux, (uint32_t)(ux >> 32), (uint32_t)uy, (uint32_t)(uy >> 32));
    q[7] = make_uint4((uint32_t)uz, (uint32_t)(uz >> 32), 0u, 0u);
    box_area = (uint32_t)(ix1 - ix0 + 1) * (uint32_t)(iy1 - iy0 + 1);
    return (uint32_t)((ix1 / RDR_TILE - ix0 / RDR_TILE + 1) * (iy1 / RDR_TILE - iy0 / RDR_TILE + 1));
}

// GPU-side culling below the granularity of a draw. The reference culls whole primitives against the frustum on the
// host (VulkanRenderer.cpp:929-943) and leaves the rest to the fixed-function clipper; here a pre-pass tests the
// bounding box of every chunk of RDR_CHUNK_TRIS consecutive triangles, so that a ten-million-triangle primitive seen
// through a shadow cascade that covers a corner of it costs set-up work only for that corner.
//
// A chunk is culled only if set-up would drop every triangle in it anyway, so nothing downstream changes. Two ways:
//  * all of the box lies outside one of set-up's six clip planes (near, far, guard band): each of its triangles has
//    its three vertices outside that plane;
//  * all of the box lies in front of the eye (w > 0) and more than two pixels outside one side of the viewport: a
//    polygon clipped from such a triangle keeps w > 0 and stays beyond that side, so its pixel box is empty.
// With the same float matrix as coefficients a plane distance is a linear function of the position, so over the box
// it is largest at a corner; the float evaluation at a vertex differs from the exact value by a few ulp of the sum
// of the absolute values of its terms (A below, taken at the largest coordinates of the box), and the tests leave a
// margin of 2^-16 A, hundreds of times that.
__global__ void __launch_bounds__(256) k_chunk_cull(const DView* __restrict__ views, const DDraw* __restrict__ draws,
                                                    const uint32_t* __restrict__ chunk_first, uint32_t n_draws, uint32_t n_chunks,
                                                    const DPrim* __restrict__ prims, const float4* __restrict__ chunk_bb,
                                                    uint8_t* __restrict__ flags, uint32_t* __restrict__ n_binned) {
    const uint32_t ci = blockIdx.x * 256u + threadIdx.x;
    const bool valid = ci < n_chunks;  // lanes past the end stay for the wave-wide count at the bottom
    const uint32_t di = find_draw(chunk_first, n_draws, valid ? ci : 0u);
    const DDraw& d = draws[di];
    const DPrim& pr = prims[d.prim];
    uint8_t culled = 0;
    if (valid && pr.first_chunk != RDR_NO_CHUNKS) {
        const DView& vw = views[d.view];
        const float kx = 32768.0f / (float)vw.w, ky = 32768.0f / (float)vw.h;
        const size_t at = 2 * ((size_t)pr.first_chunk + (ci - chunk_first[di]));
        const float4 lo = chunk_bb[at], hi = chunk_bb[at + 1];
        const float mx = fmaxf(fabsf(lo.x), fabsf(hi.x)), my = fmaxf(fabsf(lo.y), fabsf(hi.y)), mz = fmaxf(fabsf(lo.z), fabsf(hi.z));
        const float* m = d.mvp;
        float A[4];  // bound on the sum of absolute terms of clip x, y, z, w over the box
#pragma unroll
        for (int r = 0; r < 4; r++) A[r] = fabsf(m[r]) * mx + fabsf(m[4 + r]) * my + fabsf(m[8 + r]) * mz + fabsf(m[12 + r]);
        const float eps = 1.52587890625e-05f;  // 2^-16
        const float vx = 1.0f + 4.0f / (float)vw.w, vy = 1.0f + 4.0f / (float)vw.h;  // viewport sides, two pixels out
        // planes 0..5 as in plane_dist; 6..9 the viewport's -x, +x, -y, +y sides
        const float slack[10] = {eps * A[2], eps * (A[3] + A[2]), eps * (kx * A[3] + A[0]), eps * (kx * A[3] + A[0]), eps * (ky * A[3] + A[1]),
                                 eps * (ky * A[3] + A[1]), eps * (vx * A[3] + A[0]), eps * (vx * A[3] + A[0]), eps * (vy * A[3] + A[1]),
                                 eps * (vy * A[3] + A[1])};
        uint32_t all_outside = 1023u;
        bool finite = true, in_front = true;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const F4 q = xform_point(m, (k & 1) ? hi.x : lo.x, (k & 2) ? hi.y : lo.y, (k & 4) ? hi.z : lo.z);
            const ClipV cv = {q.x, q.y, q.z, q.w};
            finite = finite && isfinite(q.x) && isfinite(q.y) && isfinite(q.z) && isfinite(q.w);
            in_front = in_front && q.w - eps * A[3] > 0.0f;
            float dist[10];
#pragma unroll
            for (int plane = 0; plane < 6; plane++) dist[plane] = plane_dist(cv, plane, kx, ky);
            dist[6] = fmaf(vx, q.w, q.x); dist[7] = fmaf(vx, q.w, -q.x); dist[8] = fmaf(vy, q.w, q.y); dist[9] = fmaf(vy, q.w, -q.y);
#pragma unroll
            for (int plane = 0; plane < 10; plane++)
                if (!(dist[plane] + slack[plane] < 0.0f)) all_outside &= ~(1u << plane);
        }
        bool ok = finite;
#pragma unroll
        for (int plane = 0; plane < 10; plane++) ok = ok && isfinite(slack[plane]);
        if (!in_front) all_outside &= 63u;
        culled = (ok && all_outside) ? 1 : 0;
    }
    if (valid) flags[ci] = culled;
    // statistic: culled chunks, summed per wave into word 4 of the spread counters (see k_setup)
    const unsigned long long m = __ballot(culled != 0);
    if (m && (threadIdx.x & 63u) == 0u)
        atomicAdd(n_binned + (((blockIdx.x << 2) | (threadIdx.x >> 6)) & (RDR_BINNED_SLOTS - 1u)) * 32u + 4u, (uint32_t)__popcll(m));
}

#ifndef RDR_SETUP_DIAG_NO_CLIP
#define RDR_SETUP_DIAG_NO_CLIP 0  // diagnostic builds only: triangles that need clipping are dropped
#endif
__global__ void __launch_bounds__(256) k_setup(const DView* __restrict__ views, const DDraw* __restrict__ draws,
                                               const uint32_t* __restrict__ draw_first, uint32_t n_draws, uint32_t n_inst,
                                               const DPrim* __restrict__ prims, const float4* __restrict__ pos,
                                               const uint32_t* __restrict__ idx, DTriRec* __restrict__ recs,
                                               uint32_t* __restrict__ pair_count, uint32_t* __restrict__ n_binned,
                                               const uint32_t* __restrict__ chunk_first, const uint8_t* __restrict__ chunk_culled) {
    const uint32_t inst = blockIdx.x * 256u + threadIdx.x;
    const bool valid = inst < n_inst;  // lanes past the end stay for the wave-wide sums at the bottom
    uint32_t count = 0;  // records of dropped triangles are never read (they own no pair) and are left unwritten
    uint32_t box_area = 0;
    bool blended_view = false;
    do {
        if (!valid) break;
        const uint32_t di = find_draw(draw_first, n_draws, inst);
        const DDraw& d = draws[di];
        const DView& vw = views[d.view];
        const DPrim& pr = prims[d.prim];
        const uint32_t tri = inst - d.first_instance;
        if (chunk_culled[chunk_first[di] + tri / RDR_CHUNK_TRIS]) break;  // every triangle of the chunk lies outside one clip plane
        DTriRec& rec = recs[inst];
        blended_view = vw.load == 2u;
        const uint32_t* ip = idx + pr.first_index + 3u * tri;
        const float4 p0 = pos[pr.first_vertex + ip[0]], p1 = pos[pr.first_vertex + ip[1]], p2 = pos[pr.first_vertex + ip[2]];
        ClipV cv[3];
        {
            const F4 a = xform_point(d.mvp, p0.x, p0.y, p0.z), b = xform_point(d.mvp, p1.x, p1.y, p1.z), c = xform_point(d.mvp, p2.x, p2.y, p2.z);
            cv[0] = {a.x, a.y, a.z, a.w}; cv[1] = {b.x, b.y, b.z, b.w}; cv[2] = {c.x, c.y, c.z, c.w};
        }
        bool finite = true;
#pragma unroll
        for (int i = 0; i < 3; i++) finite = finite && isfinite(cv[i].x) && isfinite(cv[i].y) && isfinite(cv[i].z) && isfinite(cv[i].w);
        if (!finite) break;
        const int W = (int)vw.w, H = (int)vw.h;
        const float kx = 32768.0f / (float)W, ky = 32768.0f / (float)H;
        const float halfW256 = 128.0f * (float)W, halfH256 = 128.0f * (float)H;

        // ---- clip: a plane is applied only if an input vertex lies outside it
        uint32_t need = 0;
        bool rejected = false;
#pragma unroll
        for (int plane = 0; plane < 6; plane++) {
            int outside = 0;
#pragma unroll
            for (int i = 0; i < 3; i++) outside += plane_dist(cv[i], plane, kx, ky) < 0.0f ? 1 : 0;
            if (outside == 3) rejected = true;
            if (outside) need |= 1u << plane;
        }
        if (rejected) break;
        // The unclipped case (nearly every triangle) is kept apart so that its three vertices stay in registers; the
        // polygon arrays of the clipped case live in scratch memory.
        if (need == 0) {
            int32_t X[3], Y[3];
            bool ok = true;
#pragma unroll
            for (int i = 0; i < 3; i++) {
                const float xn = cv[i].x / cv[i].w, yn = cv[i].y / cv[i].w;
                const float fx = rintf(fmaf(xn, halfW256, halfW256)), fy = rintf(fmaf(yn, halfH256, halfH256));
                if (!(fabsf(fx) <= 16777216.0f) || !(fabsf(fy) <= 16777216.0f)) ok = false;
                X[i] = (int32_t)fx; Y[i] = (int32_t)fy;
            }
            if (!ok) break;
            count = finish_record<3>(X, Y, 3, cv, vw, d, tri, rec, box_area);
        } else if (!RDR_SETUP_DIAG_NO_CLIP) {
            int32_t X[RDR_MAX_POLY], Y[RDR_MAX_POLY];
            int n = 3;
            bool ok = true;
            ClipV poly[12], tmp[12];
            poly[0] = cv[0]; poly[1] = cv[1]; poly[2] = cv[2];
            for (int plane = 0; plane < 6 && ok; plane++) {
                if (!(need & (1u << plane))) continue;
                int m = 0;
                for (int i = 0; i < n; i++) {
                    const ClipV a = poly[i];
                    const ClipV b = poly[(i + 1) % n];
                    const float da = plane_dist(a, plane, kx, ky), db = plane_dist(b, plane, kx, ky);
                    const bool ina = da >= 0.0f, inb = db >= 0.0f;
                    if (ina) tmp[m++] = a;
                    if (ina != inb) {
                        const ClipV in = ina ? a : b;
                        const ClipV out = ina ? b : a;
                        const float din = ina ? da : db, dout = ina ? db : da;
                        const float tt = din / (din - dout);
                        ClipV q;
                        q.x = fmaf(tt, out.x - in.x, in.x);
                        q.y = fmaf(tt, out.y - in.y, in.y);
                        q.z = fmaf(tt, out.z - in.z, in.z);
                        q.w = fmaf(tt, out.w - in.w, in.w);
                        tmp[m++] = q;
                    }
                }
                n = m;
                for (int i = 0; i < n; i++) poly[i] = tmp[i];
                if (n < 3) ok = false;
            }
            if (n > RDR_MAX_POLY) ok = false;
            if (ok) {
                for (int i = 0; i < n; i++) {
                    const float xn = poly[i].x / poly[i].w, yn = poly[i].y / poly[i].w;
                    const float fx = rintf(fmaf(xn, halfW256, halfW256)), fy = rintf(fmaf(yn, halfH256, halfH256));
                    if (!(fabsf(fx) <= 16777216.0f) || !(fabsf(fy) <= 16777216.0f)) ok = false;
                    X[i] = (int32_t)fx; Y[i] = (int32_t)fy;
                }
            }
            if (!ok) break;
            count = finish_record<0>(X, Y, n, cv, vw, d, tri, rec, box_area);
        }
    } while (false);
    if (valid) pair_count[inst] = count;
    // One atomic per wave, spread over RDR_BINNED_SLOTS counters a cache line apart (the host adds them up): atomics
    // on one address are served one after the other, and a kernel is not over until its last one has landed, which
    // for one shared counter took three times longer than all the set-up work.
    const unsigned long long alive = __ballot(count != 0u);
    uint32_t* slot = n_binned + (((blockIdx.x << 2) | (threadIdx.x >> 6)) & (RDR_BINNED_SLOTS - 1u)) * 32u;
    const bool leader = (threadIdx.x & 63u) == 0u;
    if (alive) {
        // (triangle, tile) pairs of the wave, summed in 64 bits: the scan that follows works in 32 bits, and the host
        // refuses the batch when the total would not fit instead of letting it wrap
        unsigned long long pairs64 = count;
#pragma unroll
        for (int dlt = 32; dlt >= 1; dlt >>= 1)
            pairs64 += ((unsigned long long)(uint32_t)__shfl_xor((int)(pairs64 >> 32), dlt) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)pairs64, dlt);
        if (leader) { atomicAdd(slot, (uint32_t)__popcll(alive)); atomicAdd((unsigned long long*)(slot + 6), pairs64); }
    }
    // Triangles of the blended view (load == 2): their (triangle, tile) pairs and the pixels of their boxes bound
    // the fragment list of the blended pass from above; the host sizes that list from these sums.
    const bool blended = count != 0u && blended_view;
    if (__ballot(blended)) {
        unsigned long long area = blended ? (unsigned long long)box_area : 0ull;
        uint32_t pairs = blended ? count : 0u;
#pragma unroll
        for (int dlt = 32; dlt >= 1; dlt >>= 1) {
            area += ((unsigned long long)(uint32_t)__shfl_xor((int)(area >> 32), dlt) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)area, dlt);
            pairs += (uint32_t)__shfl_xor((int)pairs, dlt);
        }
        if (leader) { atomicAdd(slot + 1, pairs); atomicAdd((unsigned long long*)(slot + 2), area); }
    }
}

// Largest i in [0, n) with offset[i] <= p (offset[0] <= p), found by a whole wave: 64 probes per round, so a million
// entries take four dependent rounds of loads where a binary search takes twenty.
RDR_DEV uint32_t wave_upper_index(const uint32_t* __restrict__ offset, uint32_t n, uint32_t p, uint32_t lane) {
    uint32_t a = 0, len = n;
    while (len > 1u) {
        const uint32_t step = (len + 63u) >> 6;
        const uint32_t i = a + lane * step;
        const bool le = lane * step < len && offset[i] <= p;   // true for a prefix of the lanes: the offsets are sorted
        const uint32_t cnt = (uint32_t)__popcll(__ballot(le));  // >= 1: lane 0 probes offset[a]
        const uint32_t adv = (cnt - 1u) * step;
        a += adv;
        len = min(step, len - adv);
    }
    return a;
}

// One thread per candidate pair: pair p belongs to the triangle instance i with pair_offset[i] <= p < pair_offset[i + 1]
// (pair_offset = exclusive scan of pair_count) and is that triangle's (p - pair_offset[i])-th tile. The owners of the
// 256 pairs of a block are a run of instances: two waves find its ends with the wave-wide search, the block copies
// that run of offsets to LDS and every thread finds its owner there. (Runs longer than EMIT_SPAN - stretches of
// instances without pairs - are searched in global memory, between the two ends.) The chain of dependent loads per
// pair is 5 long instead of 20, which matters most when this kernel runs beside the previous frame's shading.
#define EMIT_SPAN 2048u
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8))) k_emit(const DView* __restrict__ views, const DTriRec* __restrict__ recs,
                                              const uint32_t* __restrict__ pair_offset, uint32_t n_inst, uint32_t n_pairs,
                                              uint32_t* __restrict__ keys, uint32_t* __restrict__ vals, uint32_t* __restrict__ zero, uint32_t n_zero) {
    __shared__ uint32_t lds_off[EMIT_SPAN];
    if (blockIdx.x == 0u)
        for (uint32_t i = threadIdx.x; i < n_zero; i += 256u) zero[i] = 0u;
    __shared__ uint32_t lds_end[2];
    const uint32_t p0 = blockIdx.x * 256u, p = p0 + threadIdx.x;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    if (wave < 2u) {
        const uint32_t target = wave == 0u ? p0 : min(p0 + 255u, n_pairs - 1u);
        const uint32_t e = wave_upper_index(pair_offset, n_inst, target, lane);
        if (lane == 0u) lds_end[wave] = e;
    }
    __syncthreads();
    const uint32_t first = lds_end[0], span = lds_end[1] - first + 1u;
    const bool in_lds = span <= EMIT_SPAN;
    if (in_lds)
        for (uint32_t i = threadIdx.x; i < span; i += 256u) lds_off[i] = pair_offset[first + i];
    __syncthreads();
    if (p >= n_pairs) return;
    uint32_t lo = 0, hi = span;  // largest i in the run with offset <= p
    if (in_lds) {
        while (hi - lo > 1u) {
            const uint32_t mid = (lo + hi) >> 1;
            if (lds_off[mid] <= p) lo = mid; else hi = mid;
        }
    } else {
        while (hi - lo > 1u) {
            const uint32_t mid = (lo + hi) >> 1;
            if (pair_offset[first + mid] <= p) lo = mid; else hi = mid;
        }
    }
    lo += first;
    const DTriRec& r = recs[lo];
    const uint32_t local = p - pair_offset[lo];
    const int n = (int)(r.n_view & 255u);
    const DView& vw = views[r.n_view >> 8];
    const int tx0 = r.px0 / RDR_TILE, ty0 = r.py0 / RDR_TILE, tx1 = r.px1 / RDR_TILE;
    const int bw = tx1 - tx0 + 1;
    const int tx = tx0 + (int)(local % (uint32_t)bw), ty = ty0 + (int)(local / (uint32_t)bw);
    // sample-position extent of the part of the tile inside the triangle's pixel box
    const int bx0 = max(tx * RDR_TILE, (int)r.px0), bx1 = min(tx * RDR_TILE + RDR_TILE - 1, (int)r.px1);
    const int by0 = max(ty * RDR_TILE, (int)r.py0), by1 = min(ty * RDR_TILE + RDR_TILE - 1, (int)r.py1);
    const int lo_off = vw.samples == 1u ? 128 : 32, hi_off = vw.samples == 1u ? 128 : 224;
    const long long xmin = 256ll * bx0 + lo_off, xmax = 256ll * bx1 + hi_off, ymin = 256ll * by0 + lo_off, ymax = 256ll * by1 + hi_off;
    bool keep = true;
    for (int i = 0; i < n; i++) {
        const int j = (i + 1 == n) ? 0 : i + 1;
        const long long dX = r.X[j] - r.X[i], dY = r.Y[j] - r.Y[i];
        const long long Px = dY > 0 ? xmin : xmax, Py = dX > 0 ? ymax : ymin;  // corner where the edge function is largest
        const long long E = dX * (Py - r.Y[i]) - dY * (Px - r.X[i]);
        const long long bias = ((dY < 0) || (dY == 0 && dX > 0)) ? 0 : 1;
        if (E < bias) keep = false;
    }
    keys[p] = keep ? vw.tile_base + (uint32_t)ty * vw.tiles_x + (uint32_t)tx : RDR_PAIR_INVALID;
    vals[p] = lo;
}

__global__ void __launch_bounds__(256) k_ranges(const uint32_t* __restrict__ keys, uint32_t n_pairs, uint32_t* __restrict__ tile_start,
                                                uint32_t* __restrict__ tile_end) {
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= n_pairs) return;
    // tile_end follows tile_start: a key that is no tile of the batch (RDR_PAIR_INVALID, the dropped pairs at the end
    // of the order) owns no range
    const uint32_t n_tiles = (uint32_t)(tile_end - tile_start);
    const uint32_t k = keys[p];
    const uint32_t prev = p ? keys[p - 1] : RDR_PAIR_INVALID;
    if (k != prev || p == 0) {
        if (k < n_tiles) tile_start[k] = p;
        if (p && prev < n_tiles) tile_end[prev] = p;
    }
    if (p == n_pairs - 1 && k < n_tiles) tile_end[k] = n_pairs;
}

// threads per tile and records staged per pass: single-sampled views (shadow maps), 4x multisampled view
#ifndef RDR_RASTER1_WAVES
#define RDR_RASTER1_WAVES 6  // waves per SIMD the single-sample instantiation is compiled for
#endif
#ifndef RDR_RASTER_T1
#define RDR_RASTER_T1 64
#endif
#ifndef RDR_RASTER_C1
#define RDR_RASTER_C1 32
#endif
#ifndef RDR_RASTER_T4
#define RDR_RASTER_T4 128
#endif
#ifndef RDR_RASTER_C4
#define RDR_RASTER_C4 32
#endif
// consecutive tiles per block
#ifndef RDR_RASTER_K1
#define RDR_RASTER_K1 1
#endif
#ifndef RDR_RASTER_DIAG
#define RDR_RASTER_DIAG 0  // diagnostic builds only: bit 0 no per-pixel items, bit 1 every tile taken as empty
#endif
#ifndef RDR_RASTER_K4
#define RDR_RASTER_K4 1
#endif

// One block of T threads per 16x16 tile, staging the tile's triangle records C at a time. A tile spends most of its
// life waiting for two dependent rounds of loads per chunk (the sorted triangle ids, then the records they point at;
// cycle counters around the phases of a typical tile of the benchmark frame: staging 65 % of the block's time for the
// main view, 85 % for the shadow views, set-up 6 %, the pixel work 13 %), so what counts is how many tiles a CU has in
// flight: one wave and 6 KB of LDS per shadow-view tile, two waves and 12.5 KB per tile of the 4x view. Measured on
// that frame: shadow views (T, C) = (128, 64) 176 us, (64, 64) 189, (64, 32) 158, (64, 16) 211; 4x view (256, 64)
// 271 us, (128, 64) 221, (128, 32) 216, (128, 16) 263, (64, 32) 234.
// Work is spread over (triangle, pixel) items, not over triangles: for every chunk of records the block
// sums the areas of their pixel boxes inside the tile (prefix sum), and thread k takes items k, k+T, ... - so
// sixteen 2x2-pixel triangles keep one wave busy where a loop over triangles would leave 60 of its 64 lanes idle
// on each.
//
// Items of different triangles hit the same sample in any order, so the depth
// test is an LDS 64-bit atomic min on (depth bits << 32 | triangle id): depths
// are in [0,1], where float order is unsigned-integer order, and triangle ids
// grow along the tile's list, so the minimum is exactly what a sequential LESS
// test in list order leaves (the earlier triangle keeps a tie). The key starts
// as (initial depth << 32 | 0): a fragment at the initial depth never beats
// it, as LESS demands.
// A point-light face that moving meshes are drawn on top of starts from its cached static map (DView::init, load == 1
// with init != depth). The map is copied to the target in one bandwidth-bound sweep before the face is rasterised, so
// the tiles without triangles - nearly all of such a face - cost the rasteriser one load of their list bounds.
// Such a target is a bordered face (pitch = w + 2, one texel around the h rows, both maps laid out alike): the sweep
// takes the whole padded face, rows and border in one run of 16-byte words where the face's size and place allow it
// (the border is written again from the neighbouring faces once the rasteriser is through, k_shadow_cube_border).
// grid: (blocks over the padded face, 1, views); `views` is the batch's table in device memory (a view the draw-list
// kernels dropped, load == 3, is left alone).
__global__ void __launch_bounds__(256) k_view_init(const DView* __restrict__ views, uint32_t first_view) {
    const DView& vw = views[first_view + blockIdx.z];
    if (vw.samples != 1u || vw.load != 1u || vw.init == vw.depth) return;
    if (vw.pitch != vw.w + 2u) {  // not a bordered face: the rows alone
        const size_t i0 = ((size_t)blockIdx.x * 256u + threadIdx.x) * 4u;
        for (size_t k = i0; k < min((size_t)vw.w * vw.h, i0 + 4u); k++) {
            const size_t at = k / vw.w * vw.pitch + k % vw.w;
            vw.depth[at] = vw.init[at];
        }
        return;
    }
    const size_t lead = (size_t)vw.pitch + 1u, n = (size_t)vw.pitch * (vw.h + 2u);
    float* dst = vw.depth - lead;
    const float* src = vw.init - lead;
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if ((((uintptr_t)dst | (uintptr_t)src) & 15u) == 0u && (n & 3u) == 0u) {
        if (i * 4u < n) ((rdr_uint4_t*)dst)[i] = ((const rdr_uint4_t*)src)[i];
    } else {
        for (size_t k = i * 4u; k < min(n, i * 4u + 4u); k++) dst[k] = src[k];
    }
}

// CLASSIFY (the 4x view of a frame): the tile's samples are also turned into the fragment lists the opaque shading works
// from, while they are still in LDS - the distinct triangles of every pixel's samples -> (triangle, pixel) records
// appended to the list of the tile's screen rectangle, a state word per pixel, the position of the tile's first
// fragment (FragLists, kernels.h; k_frag_classify in kernels_shade.hip does the same from the visibility image).
template <int S, int T, int RASTER_CHUNK, int TILES, bool CLASSIFY>
__global__ void __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu(S == 1 ? RDR_RASTER1_WAVES : 6))) k_raster(const RasterViews tbl, const DTriRec* __restrict__ recs, const uint32_t* __restrict__ sorted_vals,
                                              const uint32_t* __restrict__ tile_start, const uint32_t* __restrict__ tile_end, const FragLists fl) {
    static_assert(!CLASSIFY || S == 4, "fragment lists are built for the multisampled main view");
    __shared__ uint32_t lds_cls[CLASSIFY ? (256 / 64) + 1 : 1];  // fragments per (pixel group, wave), then the tile's first fragment
    constexpr int Q = 256 / T;  // pixels of the tile per thread: pixel q*T + tid
    static_assert(RASTER_CHUNK <= 64 && (RASTER_CHUNK & (RASTER_CHUNK - 1)) == 0 && RASTER_CHUNK <= T, "one lane of wave 0 per staged record");
    __shared__ __attribute__((aligned(16))) uint32_t lds_rec[RASTER_CHUNK * 32];  // records of 128 bytes
    __shared__ uint32_t lds_ids[RASTER_CHUNK];
    __shared__ uint32_t lds_prefix[RASTER_CHUNK + 1];
    __shared__ uint32_t lds_cover[RASTER_CHUNK];     // RASTER_EXACT64 / RASTER_COVER / RASTER_LOCAL32, see below
    __shared__ unsigned long long lds_key[S * 256];  // [sample][pixel of the tile, row-major]
    // A tile's life is a chain of dependent loads, and what a CU gets done is set by the length of that chain. The
    // views of the launch travel in the kernel arguments (constant memory, no round trip to HBM). A block takes
    // TILES consecutive tiles, one after the other: the list bounds of the next tile are fetched while this one is
    // worked on, and its first triangle ids while this one's last chunk is rasterised, so that only the first tile of
    // a block waits for them.
    const int tid = (int)threadIdx.x;
    const uint32_t tile0 = tbl.first_tile + blockIdx.x * (uint32_t)TILES;
    uint32_t nbegin = tile_start[tile0], nend = tile_end[tile0];
    uint32_t next_id = 0u;   // triangle id this thread stages for the coming chunk
    bool have_ids = false;   // ... already fetched for the tile about to start
    for (int t = 0; t < TILES; t++) {
    const uint32_t tile = tile0 + (uint32_t)t;
    if (tile >= tbl.end_tile) break;
    const uint32_t begin = nbegin, end = nend;
    nbegin = nend = 0u;
    if (t + 1 < TILES && tile + 1u < tbl.end_tile) { nbegin = tile_start[tile + 1u]; nend = tile_end[tile + 1u]; }
    if (!have_ids && tid < RASTER_CHUNK && begin + (uint32_t)tid < end) next_id = sorted_vals[begin + tid];
    have_ids = false;
    // view of this tile: the last one whose first tile is <= tile (views are laid out in order of their tiles)
    uint32_t v = 0;
    for (uint32_t k = 1; k < tbl.n; k++) if (tbl.base[k] <= tile) v = k;
    v = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
    const DView vw = tbl.v[v];
    const uint32_t local_tile = tile - vw.tile_base;
    if (local_tile >= vw.tiles_x * vw.tiles_y) continue;
    const int ox0 = (int)(local_tile % vw.tiles_x) * RDR_TILE, oy0 = (int)(local_tile / vw.tiles_x) * RDR_TILE;
    const size_t plane_px = (size_t)vw.pitch * vw.h;
    bool store_depth = true;
    if (tbl.depth_mode) {
        const bool has_blend = tbl.blend_base != 0xFFFFFFFFu && tile_start[tbl.blend_base + local_tile] < tile_end[tbl.blend_base + local_tile];
        store_depth = (tbl.depth_mode == 1u) == has_blend;
        if (tbl.depth_mode == 2u && !store_depth) continue;  // stored by the first pass (and changed by the blended pass since)
    }
    if (begin >= end || (RDR_RASTER_DIAG & 2)) {  // nothing drawn here: the tile is just cleared ...
        if (S == 1 && vw.load) continue;  // ... or, for a view that adds to what its target holds, left as it is
#pragma unroll
        for (int q = 0; q < Q; q++) {
            const int pix = q * T + tid, cx = ox0 + (pix & 15), cy = oy0 + (pix >> 4);
            if (cx < (int)vw.w && cy < (int)vw.h) {
                const size_t at = (size_t)cy * vw.pitch + cx;
#pragma unroll
                for (int s = 0; s < S; s++) {
                    if (store_depth) vw.depth[s * plane_px + at] = 1.0f;
                    if (vw.vis) vw.vis[s * plane_px + at] = RDR_PAIR_INVALID;
                }
            }
            if (CLASSIFY) fl.pixel_state[(size_t)local_tile * 256u + (uint32_t)pix] = 0xFFFF0000u;  // no fragment, every sample sky
        }
        if (CLASSIFY && tid == 0) fl.tile_first[local_tile] = 0u;
        continue;
    }

    // bits of the initial depth; the 4x view always starts from a clear (nothing to keep in registers)
    constexpr uint32_t ONE = 0x3F800000u;
    uint32_t zi1[Q];
#pragma unroll
    for (int q = 0; q < Q; q++) {
        const int pix = q * T + tid, px = ox0 + (pix & 15), py = oy0 + (pix >> 4);
        const bool in_view = px < (int)vw.w && py < (int)vw.h;
        zi1[q] = (S == 1 && vw.load && in_view) ? __float_as_uint(vw.init[(size_t)py * vw.pitch + px]) : ONE;
#pragma unroll
        for (int s = 0; s < S; s++) lds_key[s * 256 + pix] = (unsigned long long)(S == 1 ? zi1[q] : ONE) << 32;
    }
    const int sxo[4] = {96, 224, 32, 160}, syo[4] = {32, 96, 160, 224};
    // Smallest upper depth bound of the polygons seen so far that contain the whole tile (kept by wave 0). A record
    // whose lower depth bound over the tile is strictly greater loses the LESS test at every sample, whatever the
    // order, so it is dropped before any per-pixel work. Bounds are taken at the tile's corners in double, rounded to
    // float and widened by one ulp, which dwarfs the rounding of the per-sample evaluation.
    float tile_far = 2.0f;
    //
    // Edge functions. E(P) = dX (Py - Yi) - dY (Px - Xi) needs 64 bits in general, but inside one tile it is
    // E(O) + A u + B v with O the tile's origin, A = -dY, B = dX and 0 <= u, v < 4096 + 224 sub-pixels. For an
    // unclipped triangle whose edges all have |A| + |B| <= 2^18 (1024 px) the variable part stays below 2^30, so the
    // setup lane folds E(O) - bias into a 32-bit constant C (saturated at +-2^30, where the sign no longer depends on
    // u, v) and rewrites the staged record in place as a RasterLocal: the per-item test is then C + A u + B v >= 0 in
    // 32-bit multiply-adds, with the same sign as the 64-bit expression. Everything else (clipped polygons, very long
    // edges) keeps the 64-bit path; a polygon containing the whole tile needs no test at all.
    for (uint32_t base = begin; base < end; base += RASTER_CHUNK) {
        const uint32_t cnt = min((uint32_t)RASTER_CHUNK, end - base);
        __syncthreads();  // the previous chunk's items are done with lds_rec (first pass: the keys are initialised)
        if ((uint32_t)tid < cnt) lds_ids[tid] = next_id;
        __syncthreads();
        // eight consecutive threads copy one 128-byte record, 16 bytes each
        for (uint32_t q = threadIdx.x; q < cnt * 8u; q += (uint32_t)T)
            ((uint4*)lds_rec)[q] = ((const uint4*)(recs + lds_ids[q >> 3]))[q & 7u];
        // the ids of the chunk after this one - or of the next tile's first chunk: in flight while this one is set up
        // and rasterised
        if (base + RASTER_CHUNK < end) {
            if (tid < RASTER_CHUNK && base + RASTER_CHUNK + (uint32_t)tid < end) next_id = sorted_vals[base + RASTER_CHUNK + tid];
        } else if (nbegin < nend) {
            if (tid < RASTER_CHUNK && nbegin + (uint32_t)tid < nend) next_id = sorted_vals[nbegin + tid];
            have_ids = true;
        }
        __syncthreads();
        // wave 0: area of each record's pixel box inside the tile, inclusive prefix sum over its lanes
        if (tid < 64) {
            uint32_t area = 0;
            bool cover = false;
            float z_lo = 0.0f, z_hi = 2.0f;
            if ((uint32_t)tid < cnt) {
                const DTriRec* r = (const DTriRec*)(lds_rec + tid * 32);
                const int bx0 = max((int)r->px0, ox0), bx1 = min((int)r->px1, ox0 + RDR_TILE - 1);
                const int by0 = max((int)r->py0, oy0), by1 = min((int)r->py1, oy0 + RDR_TILE - 1);
                if (bx0 <= bx1 && by0 <= by1) {
                    area = (uint32_t)((bx1 - bx0 + 1) * (by1 - by0 + 1));
                    const int X0 = 256 * ox0, X1 = X0 + 256 * RDR_TILE, Y0 = 256 * oy0, Y1 = Y0 + 256 * RDR_TILE;
                    const int n = (int)(r->n_view & 255u);
                    cover = area == (uint32_t)(RDR_TILE * RDR_TILE);
                    for (int i = 0; i < n && cover; i++) {  // every edge function is positive at its worst tile corner
                        const int j = (i + 1 == n) ? 0 : i + 1;
                        const int Xi = r->X[i], Yi = r->Y[i];
                        const int dX = r->X[j] - Xi, dY = r->Y[j] - Yi;
                        const long long Emin = (long long)dX * (long long)((dX >= 0 ? Y0 : Y1) - Yi) - (long long)dY * (long long)((dY >= 0 ? X1 : X0) - Xi);
                        if (Emin < 1) cover = false;
                    }
                    const double dzdX = r->dzdX, dzdY = r->dzdY, z00 = r->z00;
                    const double lo = fma(dzdX, (double)(dzdX >= 0.0 ? X0 : X1), fma(dzdY, (double)(dzdY >= 0.0 ? Y0 : Y1), z00));
                    const double hi = fma(dzdX, (double)(dzdX >= 0.0 ? X1 : X0), fma(dzdY, (double)(dzdY >= 0.0 ? Y1 : Y0), z00));
                    float flo = (float)lo, fhi = (float)hi;
                    flo = flo < 0.0f ? 0.0f : (flo > 1.0f ? 1.0f : flo);
                    fhi = fhi < 0.0f ? 0.0f : (fhi > 1.0f ? 1.0f : fhi);
                    z_lo = flo > 0.0f ? __uint_as_float(__float_as_uint(flo) - 1u) : 0.0f;  // one ulp down / up: depths are in [0, 1]
                    z_hi = __uint_as_float(__float_as_uint(fhi) + 1u);
                }
            }
            float chunk_far = cover ? z_hi : 2.0f;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) chunk_far = fminf(chunk_far, __shfl_xor(chunk_far, d));
            tile_far = fminf(tile_far, chunk_far);
            if (z_lo > tile_far) { area = 0; cover = false; }
            uint32_t mode = cover ? RASTER_COVER : RASTER_EXACT64;
            if (area && !cover) mode = raster_localise<S>(lds_rec + tid * 32, ox0, oy0);
            if ((uint32_t)tid < cnt) lds_cover[tid] = mode;
            uint32_t incl = area;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t up = __shfl_up(incl, d);
                if (tid >= d) incl += up;
            }
            if (tid < RASTER_CHUNK) lds_prefix[tid + 1] = incl;
            if (tid == 0) lds_prefix[0] = 0u;
        }
        __syncthreads();
        const uint32_t total = (RDR_RASTER_DIAG & 1) ? 0u : lds_prefix[RASTER_CHUNK];
        for (uint32_t wi = (uint32_t)tid; wi < total; wi += (uint32_t)T) {
            // largest lo with prefix[lo] <= wi; entries past the last record hold the total, which is > wi
            uint32_t lo = 0;
#pragma unroll
            for (int step = RASTER_CHUNK / 2; step >= 1; step >>= 1) lo += lds_prefix[lo + step] <= wi ? (uint32_t)step : 0u;
            const uint32_t local = wi - lds_prefix[lo];
            const uint32_t mode = lds_cover[lo];
            uint32_t inside = (1u << S) - 1u;
            int x, y;
            double dzdX, dzdY, z00;
            uint32_t rid;
            if (mode == RASTER_LOCAL32) {
                const RasterLocal* f = (const RasterLocal*)(lds_rec + lo * 32);
                const uint32_t box = f->box;
                const uint32_t bw = ((box >> 8) & 15u) + 1u;
                const uint32_t row = __umul24(local, box >> 12) >> 16;   // local / bw, exact for local < 256
                const int xl = (int)(box & 15u) + (int)(local - __umul24(row, bw)), yl = (int)((box >> 4) & 15u) + (int)row;
                x = ox0 + xl; y = oy0 + yl;
                int Ep[3];
#pragma unroll
                for (int i = 0; i < 3; i++) Ep[i] = f->C[i] + __mul24(f->A[i], xl << 8) + __mul24(f->B[i], yl << 8);  // |A|, |B| <= 2^18
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
                const DTriRec* r = (const DTriRec*)(lds_rec + lo * 32);
                const int bx0 = max((int)r->px0, ox0), bx1 = min((int)r->px1, ox0 + RDR_TILE - 1);
                const int by0 = max((int)r->py0, oy0);
                const uint32_t bw = (uint32_t)(bx1 - bx0 + 1);               // 1..16
                const uint32_t row = (local * (65536u / bw + 1u)) >> 16;     // local / bw, exact for local < 256
                x = bx0 + (int)(local - row * bw); y = by0 + (int)row;
                const int n = (int)(r->n_view & 255u);
                const int n_edges = mode == RASTER_COVER ? 0 : n;
                for (int i = 0; i < n_edges; i++) {
                    const int j = (i + 1 == n) ? 0 : i + 1;
                    const int Xi = r->X[i], Yi = r->Y[i];
                    const int dX = r->X[j] - Xi, dY = r->Y[j] - Yi;
                    const long long E0 = (long long)dX * (long long)(256 * y - Yi) - (long long)dY * (long long)(256 * x - Xi);
                    const long long bias = ((dY < 0) || (dY == 0 && dX > 0)) ? 0 : 1;
#pragma unroll
                    for (int s = 0; s < S; s++) {
                        const int ax = S == 1 ? 128 : sxo[s], ay = S == 1 ? 128 : syo[s];
                        const long long E = E0 + ((long long)dX * ay - (long long)dY * ax);
                        if (E < bias) inside &= ~(1u << s);
                    }
                }
                dzdX = r->dzdX; dzdY = r->dzdY; z00 = r->z00; rid = r->id;
            }
            if (inside) {
                const int Px0 = 256 * x, Py0 = 256 * y;
                const int pix = (y - oy0) * RDR_TILE + (x - ox0);
#pragma unroll
                for (int s = 0; s < S; s++) {
                    if (!(inside & (1u << s))) continue;
                    const int Px = Px0 + (S == 1 ? 128 : sxo[s]), Py = Py0 + (S == 1 ? 128 : syo[s]);
                    float zz = (float)fma(dzdX, (double)Px, fma(dzdY, (double)Py, z00));
                    zz = zz < 0.0f ? 0.0f : (zz > 1.0f ? 1.0f : zz);
                    atomicMin(&lds_key[s * 256 + pix], ((unsigned long long)__float_as_uint(zz) << 32) | rid);
                }
            }
        }
    }
    __syncthreads();
    uint32_t cls_ids[CLASSIFY ? Q : 1][S], cls_state[CLASSIFY ? Q : 1], cls_n[CLASSIFY ? Q : 1];
#pragma unroll
    for (int q = 0; q < Q; q++) {
        const int pix = q * T + tid, px = ox0 + (pix & 15), py = oy0 + (pix >> 4);
        const bool in_view = px < (int)vw.w && py < (int)vw.h;
        if (CLASSIFY) { cls_n[q] = 0u; cls_state[q] = 0xFFFF0000u; }
        if (!in_view) continue;
        const size_t o = (size_t)py * vw.pitch + px;
        unsigned long long key[S];
#pragma unroll
        for (int s = 0; s < S; s++) key[s] = lds_key[s * 256 + pix];
        // every pixel is stored, changed or not: whole sectors go out, none has to be merged in memory
        if (store_depth) {
#pragma unroll
            for (int s = 0; s < S; s++) vw.depth[s * plane_px + o] = __uint_as_float((uint32_t)(key[s] >> 32));
        }
        uint32_t vid[S];
#pragma unroll
        for (int s = 0; s < S; s++) vid[s] = key[s] != ((unsigned long long)(S == 1 ? zi1[q] : ONE) << 32) ? (uint32_t)key[s] : RDR_PAIR_INVALID;
        if (vw.vis) {
#pragma unroll
            for (int s = 0; s < S; s++) vw.vis[s * plane_px + o] = vid[s];
        }
        if constexpr (CLASSIFY) {
            // the u-th distinct triangle of the pixel, in order of first appearance (15: the sample saw none), from the
            // six comparisons between its four samples; two equal ids are both valid, so only the sample itself is checked
            const bool v0 = vid[0] != RDR_PAIR_INVALID, v1 = vid[1] != RDR_PAIR_INVALID, v2 = vid[2] != RDR_PAIR_INVALID, v3 = vid[3] != RDR_PAIR_INVALID;
            const bool e10 = vid[1] == vid[0], e20 = vid[2] == vid[0], e21 = vid[2] == vid[1];
            const bool e30 = vid[3] == vid[0], e31 = vid[3] == vid[1], e32 = vid[3] == vid[2];
            const bool n1 = v1 && !e10, n2 = v2 && !e20 && !e21, n3 = v3 && !e30 && !e31 && !e32;  // first appearances (sample 0: v0)
            const uint32_t r1 = v0 ? 1u : 0u, r2 = r1 + (n1 ? 1u : 0u), r3 = r2 + (n2 ? 1u : 0u);  // distinct triangles before the sample
            const uint32_t n_unique = r3 + (n3 ? 1u : 0u);
            const uint32_t u0 = v0 ? 0u : 15u;
            const uint32_t u1 = !v1 ? 15u : (e10 ? u0 : r1);
            const uint32_t u2 = !v2 ? 15u : (e20 ? u0 : (e21 ? u1 : r2));
            const uint32_t u3 = !v3 ? 15u : (e30 ? u0 : (e31 ? u1 : (e32 ? u2 : r3)));
            const uint32_t st = (u0 << 16) | (u1 << 20) | (u2 << 24) | (u3 << 28);
            cls_ids[q][0] = v0 ? vid[0] : (n1 ? vid[1] : (n2 ? vid[2] : vid[3]));
            cls_ids[q][1] = (n1 && r1 == 1u) ? vid[1] : ((n2 && r2 == 1u) ? vid[2] : vid[3]);
            cls_ids[q][2] = (n2 && r2 == 2u) ? vid[2] : vid[3];
            cls_ids[q][3] = vid[3];
            cls_n[q] = n_unique; cls_state[q] = st;
        }
    }
    if (CLASSIFY) {
        // exclusive prefix sum of the fragment counts (0..4) in pixel order: pixel q * T + tid, so group q before q + 1,
        // inside a group wave after wave; three ballots per group and wave, then the totals through LDS
        constexpr int NW = T / 64;
        const int wave = tid >> 6, lane = tid & 63;
        uint32_t before[Q];
#pragma unroll
        for (int q = 0; q < Q; q++) {
            uint32_t b = 0u, tot = 0u;
#pragma unroll
            for (int bit = 0; bit < 3; bit++) {
                const unsigned long long m = __ballot((cls_n[q] >> bit) & 1u);
                b += (uint32_t)__popcll(m & ((1ull << lane) - 1ull)) << bit;
                tot += (uint32_t)__popcll(m) << bit;
            }
            before[q] = b;
            if (lane == 0) lds_cls[q * NW + wave] = tot;
        }
        __syncthreads();
        uint32_t n_frag = 0u;
#pragma unroll
        for (int j = 0; j < Q * NW; j++) {
            const uint32_t v = lds_cls[j];
#pragma unroll
            for (int q = 0; q < Q; q++) if (j < q * NW + wave) before[q] += v;
            n_frag += v;
        }
        const uint32_t tx = local_tile % vw.tiles_x, ty = local_tile / vw.tiles_x;
        const uint32_t gx = tx / fl.rect_w, gy = ty / fl.rect_h;
        const uint32_t list = ((gx + gy) & 7u) * 8u + gy;
        __syncthreads();  // everybody has read the totals: the last word now carries the tile's first fragment
        if (tid == 0) {
            const uint32_t at = n_frag ? atomicAdd(fl.count + list * RDR_FRAG_CTR_STRIDE, n_frag) : 0u;  // <= cap - n_frag: the list holds every sample of its rectangle
            lds_cls[256 / 64] = list * fl.cap + at;
            fl.tile_first[local_tile] = list * fl.cap + at;
        }
        __syncthreads();
        const uint32_t first = lds_cls[256 / 64];
#pragma unroll
        for (int q = 0; q < Q; q++) {
            const int pix = q * T + tid, px = ox0 + (pix & 15), py = oy0 + (pix >> 4);
            const uint32_t xy = (uint32_t)px | ((uint32_t)py << 16);
#pragma unroll
            for (int k = 0; k < S; k++)
                if ((uint32_t)k < cls_n[q]) fl.recs[first + before[q] + k] = make_uint2(cls_ids[q][k], xy);
            fl.pixel_state[(size_t)local_tile * 256u + (uint32_t)pix] = cls_state[q] | before[q];  // before < 1024
        }
    }
    }  // tiles of the block
}

}  // namespace

void launch_setup(hipStream_t s, const DView* views, const DDraw* draws, const uint32_t* draw_first, uint32_t n_draws, uint32_t n_inst,
                  const DPrim* prims, const void* pos, const uint32_t* idx, DTriRec* recs, uint32_t* pair_count, uint32_t* n_binned,
                  const uint32_t* chunk_first, uint32_t n_chunks, const void* chunk_bb, uint8_t* chunk_flags) {
    if (!n_inst) return;
    hipLaunchKernelGGL(k_chunk_cull, dim3((n_chunks + 255) / 256), dim3(256), 0, s, views, draws, chunk_first, n_draws, n_chunks, prims,
                       (const float4*)chunk_bb, chunk_flags, n_binned);
    hipLaunchKernelGGL(k_setup, dim3((n_inst + 255) / 256), dim3(256), 0, s, views, draws, draw_first, n_draws, n_inst, prims,
                       (const float4*)pos, idx, recs, pair_count, n_binned, chunk_first, chunk_flags);
}
void launch_emit(hipStream_t s, const DView* views, const DTriRec* recs, const uint32_t* pair_offset, uint32_t n_inst, uint32_t n_pairs,
                 uint32_t* keys, uint32_t* vals, uint32_t* zero, uint32_t n_zero) {
    if (!n_pairs) return;
    hipLaunchKernelGGL(k_emit, dim3((n_pairs + 255) / 256), dim3(256), 0, s, views, recs, pair_offset, n_inst, n_pairs, keys, vals, zero, n_zero);
}
// `tile_ranges` = tile_start[n_tiles] followed by tile_end[n_tiles]: cleared with one fill
void launch_ranges(hipStream_t s, const uint32_t* keys, uint32_t n_pairs, uint32_t* tile_ranges, uint32_t n_tiles, bool cleared) {
    if (!cleared || !n_pairs) (void)hipMemsetAsync(tile_ranges, 0, (size_t)n_tiles * 8, s);
    if (!n_pairs) return;
    hipLaunchKernelGGL(k_ranges, dim3((n_pairs + 255) / 256), dim3(256), 0, s, keys, n_pairs, tile_ranges, tile_ranges + n_tiles);
}
// Rasterises the tiles [r.first_tile, r.end_tile): the views from r.first_view on that have the sample count
// `samples`, RDR_RASTER_MAX_VIEWS of them per launch. `views` is the host copy of the batch's view table, `d_views` the
// device copy (single-sampled targets are first brought to their initial state from it, k_view_init).
void launch_raster(hipStream_t s, const DView* views, uint32_t n_views, const DView* d_views, const DTriRec* recs, const uint32_t* sorted_vals,
                   const uint32_t* tile_start, const uint32_t* tile_end, RasterRange r, int samples, const FragLists* classify, uint32_t depth_mode,
                   uint32_t blend_base) {
    if (r.first_tile >= r.end_tile) return;
    uint32_t last = r.first_view;
    while (last + 1 < n_views && views[last + 1].tile_base < r.end_tile) last++;
    if (samples == 1) {
        size_t texels = 0;
        for (uint32_t v = r.first_view; v <= last; v++)
            if (views[v].load == 1u && views[v].init != views[v].depth) texels = max(texels, (size_t)views[v].pitch * (views[v].h + 2u));
        if (texels) hipLaunchKernelGGL(k_view_init, dim3((uint32_t)((texels + 1023u) / 1024u), 1, last - r.first_view + 1u), dim3(256), 0, s, d_views, r.first_view);
    }
    for (uint32_t v0 = r.first_view; v0 <= last; v0 += RDR_RASTER_MAX_VIEWS) {
        RasterViews tbl;
        tbl.n = min((uint32_t)RDR_RASTER_MAX_VIEWS, last - v0 + 1u);
        for (uint32_t k = 0; k < tbl.n; k++) { tbl.v[k] = views[v0 + k]; tbl.base[k] = views[v0 + k].tile_base; }
        for (uint32_t k = tbl.n; k < RDR_RASTER_MAX_VIEWS; k++) tbl.base[k] = 0xFFFFFFFFu;
        tbl.first_tile = tbl.base[0];
        tbl.depth_mode = samples == 4 ? depth_mode : 0u; tbl.blend_base = blend_base;
        tbl._pad[0] = tbl._pad[1] = tbl._pad[2] = 0u;
        const DView& lv = tbl.v[tbl.n - 1];
        tbl.end_tile = lv.tile_base + lv.tiles_x * lv.tiles_y;
        const uint32_t n_tiles = tbl.end_tile - tbl.first_tile;
        const FragLists none{};
        if (samples == 1)
            hipLaunchKernelGGL((k_raster<1, RDR_RASTER_T1, RDR_RASTER_C1, RDR_RASTER_K1, false>), dim3((n_tiles + RDR_RASTER_K1 - 1) / RDR_RASTER_K1), dim3(RDR_RASTER_T1), 0, s,
                               tbl, recs, sorted_vals, tile_start, tile_end, none);
        else if (classify && tbl.n == 1u)  // the lists are laid out for one view: the frame's main view
            hipLaunchKernelGGL((k_raster<4, RDR_RASTER_T4, RDR_RASTER_C4, RDR_RASTER_K4, true>), dim3((n_tiles + RDR_RASTER_K4 - 1) / RDR_RASTER_K4), dim3(RDR_RASTER_T4), 0, s,
                               tbl, recs, sorted_vals, tile_start, tile_end, *classify);
        else
            hipLaunchKernelGGL((k_raster<4, RDR_RASTER_T4, RDR_RASTER_C4, RDR_RASTER_K4, false>), dim3((n_tiles + RDR_RASTER_K4 - 1) / RDR_RASTER_K4), dim3(RDR_RASTER_T4), 0, s,
                               tbl, recs, sorted_vals, tile_start, tile_end, none);
    }
}
